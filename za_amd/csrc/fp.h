// BN254 prime-field arithmetic for host and gfx950 device code.
//
// Memory form (HBM, C ABI): Montgomery residue x*2^256 mod p as 4 x u64 little-endian limbs
// (identical bytes to 8 x u32 LE) -- the layout pairing_ce's Fq/Fr use in memory, i.e. what a
// Rust caller of the FFI already holds (reference call sites: prover/src/groth16/prover.rs:10-17).
// Register form: 8 x u32.  gfx950 has no carry-chained multiply-add; the primitive is
// v_mad_u64_u32 (32x32+64 -> 64), which tools/ubench_valu.hip measures at the same issue rate as a
// 32-bit add, so the multiply is written as CIOS over u32 limbs with u64 column sums.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZG_HD __host__ __device__ __forceinline__
#define ZG_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define ZG_HD inline
#define ZG_HD_NOINLINE __attribute__((noinline))
#endif

namespace zg {

struct FqParams {
    static constexpr uint32_t MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t NINV = 0xe4866389u;  // -p^-1 mod 2^32
    // 2^256 mod p
    static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                        0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    // 2^512 mod p
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};

struct FrParams {
    static constexpr uint32_t MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t NINV = 0xefffffffu;
    static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                        0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

template <class P>
struct alignas(16) Fp {
    uint32_t l[8];

    static ZG_HD Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = 0;
        return r;
    }
    static ZG_HD Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = P::ONE[i];
        return r;
    }
    ZG_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i];
        return o == 0;
    }
    ZG_HD bool operator==(const Fp& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    ZG_HD bool operator!=(const Fp& b) const { return !(*this == b); }

    // r = t - p if t >= p else t   (t < 2p)
    static ZG_HD Fp reduce_once(const uint32_t t[8]) {
        uint32_t s[8];
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)t[i] - P::MOD[i] - br;
            s[i] = (uint32_t)d;
            br = (uint32_t)(d >> 63);
        }
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = br ? t[i] : s[i];
        return r;
    }

    friend ZG_HD Fp operator+(const Fp& a, const Fp& b) {
        uint32_t t[8];
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t s = (uint64_t)a.l[i] + b.l[i] + c;
            t[i] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        return reduce_once(t);  // p < 2^254 so a+b < 2^255: no carry out of the top limb
    }
    friend ZG_HD Fp operator-(const Fp& a, const Fp& b) {
        uint32_t t[8];
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)a.l[i] - b.l[i] - br;
            t[i] = (uint32_t)d;
            br = (uint32_t)(d >> 63);
        }
        uint32_t mask = 0u - br;  // add p back when a < b
        uint32_t c = 0;
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t s = (uint64_t)t[i] + (P::MOD[i] & mask) + c;
            r.l[i] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        return r;
    }
    ZG_HD Fp neg() const {
        if (is_zero()) return *this;
        return zero() - *this;
    }
    ZG_HD Fp dbl() const { return *this + *this; }

    // Montgomery product a*b*2^-256 mod p.  CIOS with the two carry chains merged: the top limb of
    // p is < 2^30, so the running 33-bit carries never overflow.
    friend ZG_HD Fp operator*(const Fp& a, const Fp& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
        return mul_host64(a, b);
#else
        uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t A = (uint64_t)a.l[0] * b.l[i] + t[0];
            uint32_t m = (uint32_t)A * P::NINV;
            uint64_t C = (uint64_t)m * P::MOD[0] + (uint32_t)A;
            A >>= 32;
            C >>= 32;
#pragma unroll
            for (int j = 1; j < 8; j++) {
                A += (uint64_t)a.l[j] * b.l[i] + t[j];
                C += (uint64_t)m * P::MOD[j] + (uint32_t)A;
                t[j - 1] = (uint32_t)C;
                A >>= 32;
                C >>= 32;
            }
            t[7] = (uint32_t)(A + C);
        }
        return reduce_once(t);
#endif
    }
    ZG_HD Fp sqr() const { return *this * *this; }

#if !defined(__HIP_DEVICE_COMPILE__)
    // Host code (proof combination, Horner passes of the multiexp jobs, pairing, set-up glue) multiplies with 4 x 64-bit
    // limbs and 128-bit products: the same Montgomery residue, a third of the time of the 8 x 32 form above.
    static constexpr uint64_t mod64(int i) { return (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32); }
    static constexpr uint64_t ninv64() {
        uint64_t inv = 1;   // Newton: p0 * inv == 1 (mod 2^64)
        for (int i = 0; i < 6; i++) inv *= 2 - mod64(0) * inv;
        return 0 - inv;
    }
    static inline Fp mul_host64(const Fp& a, const Fp& b) {
        typedef unsigned __int128 u128;
        constexpr uint64_t NI = ninv64();
        constexpr uint64_t p0 = mod64(0), p1 = mod64(1), p2 = mod64(2), p3 = mod64(3);
        uint64_t x[4], y[4];
        for (int i = 0; i < 4; i++) {
            x[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
            y[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
        }
        uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        for (int i = 0; i < 4; i++) {
            u128 c = (u128)x[0] * y[i] + t0;
            const uint64_t u0 = (uint64_t)c;
            c = (c >> 64) + (u128)x[1] * y[i] + t1;
            const uint64_t u1 = (uint64_t)c;
            c = (c >> 64) + (u128)x[2] * y[i] + t2;
            const uint64_t u2 = (uint64_t)c;
            c = (c >> 64) + (u128)x[3] * y[i] + t3;
            const uint64_t u3 = (uint64_t)c;
            const uint64_t u4 = (uint64_t)(c >> 64);
            const uint64_t m = u0 * NI;
            c = (u128)m * p0 + u0;
            c = (c >> 64) + (u128)m * p1 + u1;
            t0 = (uint64_t)c;
            c = (c >> 64) + (u128)m * p2 + u2;
            t1 = (uint64_t)c;
            c = (c >> 64) + (u128)m * p3 + u3;
            t2 = (uint64_t)c;
            t3 = (uint64_t)(c >> 64) + u4;   // operands below p < 2^254: the running value stays below 2p
        }
        // t - p if t >= p
        u128 d = (u128)t0 - p0;
        const uint64_t s0 = (uint64_t)d;
        d = (u128)t1 - p1 - (uint64_t)((d >> 64) & 1);
        const uint64_t s1 = (uint64_t)d;
        d = (u128)t2 - p2 - (uint64_t)((d >> 64) & 1);
        const uint64_t s2 = (uint64_t)d;
        d = (u128)t3 - p3 - (uint64_t)((d >> 64) & 1);
        const uint64_t s3 = (uint64_t)d;
        const bool below = (uint64_t)(d >> 64) & 1;
        const uint64_t r0 = below ? t0 : s0, r1 = below ? t1 : s1, r2 = below ? t2 : s2, r3 = below ? t3 : s3;
        Fp r;
        r.l[0] = (uint32_t)r0; r.l[1] = (uint32_t)(r0 >> 32);
        r.l[2] = (uint32_t)r1; r.l[3] = (uint32_t)(r1 >> 32);
        r.l[4] = (uint32_t)r2; r.l[5] = (uint32_t)(r2 >> 32);
        r.l[6] = (uint32_t)r3; r.l[7] = (uint32_t)(r3 >> 32);
        return r;
    }
#endif

    ZG_HD Fp to_mont() const {
        Fp r2;
#pragma unroll
        for (int i = 0; i < 8; i++) r2.l[i] = P::R2[i];
        return *this * r2;
    }
    ZG_HD Fp from_mont() const {
        Fp o = zero();
        o.l[0] = 1;
        return *this * o;
    }
    // exponent: 8 x u32 little-endian, canonical integer
    ZG_HD Fp pow(const uint32_t e[8]) const {
        Fp acc = one(), base = *this;
        for (int i = 0; i < 256; i++) {
            if ((e[i >> 5] >> (i & 31)) & 1) acc = acc * base;
            base = base.sqr();
        }
        return acc;
    }
    ZG_HD Fp pow_u64(uint64_t e) const {
        Fp acc = one(), base = *this;
        while (e) {
            if (e & 1) acc = acc * base;
            base = base.sqr();
            e >>= 1;
        }
        return acc;
    }
    ZG_HD Fp inv() const {  // Fermat; 0 -> 0
        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = P::MOD[i];
        e[0] -= 2;  // low limb of both moduli is >= 2
        return pow(e);
    }
    static ZG_HD Fp from_u64(uint64_t v) {  // canonical small integer -> Montgomery
        Fp r = zero();
        r.l[0] = (uint32_t)v;
        r.l[1] = (uint32_t)(v >> 32);
        return r.to_mont();
    }
};

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

// ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2 + 1)
struct Fq2 {
    Fq c0, c1;
    static ZG_HD Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
    static ZG_HD Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
    ZG_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZG_HD bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
    ZG_HD bool operator!=(const Fq2& b) const { return !(*this == b); }
    friend ZG_HD Fq2 operator+(const Fq2& a, const Fq2& b) { return Fq2{a.c0 + b.c0, a.c1 + b.c1}; }
    friend ZG_HD Fq2 operator-(const Fq2& a, const Fq2& b) { return Fq2{a.c0 - b.c0, a.c1 - b.c1}; }
    ZG_HD Fq2 neg() const { return Fq2{c0.neg(), c1.neg()}; }
    ZG_HD Fq2 dbl() const { return Fq2{c0.dbl(), c1.dbl()}; }
    friend ZG_HD Fq2 operator*(const Fq2& a, const Fq2& b) {  // Karatsuba, 3 base multiplications
        Fq aa = a.c0 * b.c0;
        Fq bb = a.c1 * b.c1;
        Fq o = (a.c0 + a.c1) * (b.c0 + b.c1);
        return Fq2{aa - bb, o - aa - bb};
    }
    ZG_HD Fq2 sqr() const {  // (c0+c1)(c0-c1), 2 c0 c1
        Fq p = c0 * c1;
        return Fq2{(c0 + c1) * (c0 - c1), p + p};
    }
    ZG_HD Fq2 inv() const {
        Fq d = (c0.sqr() + c1.sqr()).inv();
        return Fq2{c0 * d, (c1 * d).neg()};
    }
};

}  // namespace zg
