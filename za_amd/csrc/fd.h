// Device-side BN254 field: 9 x 29-bit limbs, Montgomery radix 2^261, lazy reduction.
//
// Why: on gfx950 v_mad_u64_u32 (32x32+64 -> 64) issues at the same rate as a 32-bit add
// (tools/ubench_valu.hip), so a multiply is bound by instruction count.  With 29-bit limbs a 64-bit
// column absorbs all 18 partial products of a Montgomery multiply without carry handling:
// 162 v_mad_u64_u32 + 9 v_mul_lo + 18 shift/and pairs, versus 128 mads + 122 64-bit adds + ~330 moves
// for the 8 x 32-bit CIOS in fp.h.
//
// Representation
//   value V = sum l[i] * 2^(29 i),  every limb < 2^29 after every operation ("normalised"),
//   V is only reduced lazily: V < B*p with a bound B the CALLER tracks (B*p < 2^261 ~ 168.9 p).
//   Montgomery form: V == x * 2^261 (mod p).
//   mul/sqr:  out = (A*B + M*p) / 2^261  <  A*B/2^261 + p       (bound  Ba*Bb/168.9 + 1)
//   add:      bounds add.   sub<K>(a, b) = a + K*p - b  needs K >= bound(b);  bound(a) + K.
//   reduce(): any V < 2^261  ->  V' == V (mod p), V' < 2.01 p   (~40 instructions)
// Memory form ("M261"): the same residue packed into 8 x u32 (needs V < 2^256 ~ 5.29 p: callers
// reduce() before storing).  Conversion from/to the ABI's 2^256 Montgomery form is a multiplication
// by the constant 32 resp. 1/32.
//
// Lazy LIMBS (round 2): a limb has three spare bits in its 32-bit register, and a 64-bit column of the multiplier
// holds 9 products a_i*b_j + 9 products m_i*p_j as long as 9*max(a_i)*max(b_j) + 9*2^58 + 2^36 < 2^64.  So sums and
// differences that only feed a multiplication need no carry propagation at all:
//   add_lazy(a, b)        limb-wise sum (9 instructions instead of 33)
//   sub_lazy<K,S>(a, b)   a + C - b with C = K*p written with limbs >= 2^S - 2^(S-29) (needs limbs of b below that)
//   neg_lazy<K,S>(b)      C - b
//   norm()                carry propagation back to limbs < 2^29 (25 instructions)
//   mac2(a0,b0,a1,b1)     (a0*b0 + a1*b1) / 2^261 with ONE Montgomery reduction: 243 instead of 324 multiply-adds
// Multiplication results are always normalised.  The "lm" bound (largest limb) is tracked next to the value bound.
//
// ZG_FD_DEBUG (host unit test only) carries an upper bound in units of p and the largest-limb bound through every
// operation and asserts the K constants and the column capacity, so the bounds in ecd.h are machine-checked rather
// than hand-waved.
#pragma once
#include "fp.h"
#ifdef ZG_FD_DEBUG
#include <cassert>
#include <cstdio>
#endif

namespace zg {

struct Fq29 {
    using Wide = FqParams;
    static constexpr uint32_t P[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                                      0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t NINV = 0x04866389u;  // -p^-1 mod 2^29
    static constexpr uint32_t ONE[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                                        0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};  // 2^261 mod p
    static constexpr uint32_t R2[9] = {0x059bac10u, 0x0d1503a3u, 0x018016b8u, 0x10ab0ca8u, 0x02632639u,
                                       0x02c0169fu, 0x169bfd53u, 0x11869d4cu, 0x002a11a6u};   // 2^522 mod p
    static constexpr uint32_t C_TO261[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u,
                                            0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};  // 2^266 mod p
    static constexpr uint32_t C_TO256[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u,
                                            0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};  // 2^256 mod p
};
struct Fr29 {
    using Wide = FrParams;
    static constexpr uint32_t P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                      0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t NINV = 0x0fffffffu;
    static constexpr uint32_t ONE[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu,
                                        0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    static constexpr uint32_t R2[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu,
                                       0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};
    static constexpr uint32_t C_TO261[9] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau,
                                            0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
    static constexpr uint32_t C_TO256[9] = {0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu,
                                            0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
};

template <class P>
struct alignas(4) Fd {
    static constexpr uint32_t MASK = (1u << 29) - 1u;
    uint32_t l[9];
    static constexpr double NORM = 536870911.0;   // largest limb of a normalised value
#ifdef ZG_FD_DEBUG
    double ub = 0;  // upper bound of the integer value in units of p
    double lm = 0;  // upper bound of every limb (NORM when normalised)
    static constexpr double CAP = 160.0;
#define FD_SET_UB(x, v) ((x).ub = (v), (x).lm = Fd::NORM, assert((x).ub < Fd::CAP))
#define FD_SET_LM(x, v) ((x).lm = (v), assert((x).lm < 4294967288.0))
#define FD_NEED_NORM(x) assert((x).lm <= Fd::NORM)
    // capacity of a multiplier column: 9 products of each listed pair + 9 m*p products + the carry from below
    static bool column_fits(double sum_of_products) {
        return 9.0 * sum_of_products + 9.0 * NORM * NORM + 68719476736.0 < 18446744073709551615.0;
    }
#else
#define FD_SET_UB(x, v) ((void)0)
#define FD_SET_LM(x, v) ((void)0)
#define FD_NEED_NORM(x) ((void)0)
#endif

    static ZG_HD Fd zero() {
        Fd r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = 0;
        FD_SET_UB(r, 0.0);
        return r;
    }
    // exact-zero test (identity markers, canonical inputs); NOT a test modulo p
    ZG_HD bool is_exact_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) o |= l[i];
        return o == 0;
    }

    // ---- memory form: 8 x u32, value < 2^256
    static ZG_HD Fd unpack(const uint32_t w[8]) {
        Fd r;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int bit = 29 * i, j = bit >> 5, sh = bit & 31;
            uint32_t lo = w[j] >> sh;
            if (sh > 3) lo |= w[j + 1] << (32 - sh);  // limb spans two words when sh + 29 > 32
            r.l[i] = lo & MASK;
        }
        r.l[8] = w[7] >> 8;
        FD_SET_UB(r, 5.3);
        return r;
    }
    ZG_HD void pack(uint32_t w[8]) const {
#ifdef ZG_FD_DEBUG
        assert(ub <= 5.28);
        assert(lm <= NORM);
#endif
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int bit = 32 * j, i = bit / 29, o = bit - 29 * i;
            w[j] = (l[i] >> o) | (l[i + 1] << (29 - o));
        }
    }
    static ZG_HD Fd load(const Fp<typename P::Wide>* p) { return unpack(p->l); }
    ZG_HD void store(Fp<typename P::Wide>* p) const { pack(p->l); }

    // ---- carry propagation of signed 32-bit limb values d[i] in (-2^30, 2^31)
    static ZG_HD Fd normalise(const int32_t d[9]) {
        Fd r;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int32_t t = d[i] + c;
            r.l[i] = (uint32_t)t & MASK;
            c = t >> 29;
        }
        r.l[8] = (uint32_t)(d[8] + c);
        return r;
    }

    friend ZG_HD Fd operator+(const Fd& a, const Fd& b) {
#ifdef ZG_FD_DEBUG
        assert(a.lm + b.lm < 2147483647.0);
#endif
        int32_t d[9];
#pragma unroll
        for (int i = 0; i < 9; i++) d[i] = (int32_t)(a.l[i] + b.l[i]);
        Fd r = normalise(d);
        FD_SET_UB(r, a.ub + b.ub);
        return r;
    }
    ZG_HD Fd dbl() const { return *this + *this; }

    // K*p as normalised limbs (K <= 31), computed at compile time after unrolling
    template <int K>
    static ZG_HD void kp(uint32_t out[9]) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            uint64_t t64 = (uint64_t)P::P[i] * (uint32_t)K + c;
            out[i] = i < 8 ? (uint32_t)(t64 & MASK) : (uint32_t)t64;
            c = (uint32_t)(t64 >> 29);
        }
    }
    // a + K*p - b,  requires value(b) <= K*p
    template <int K>
    static ZG_HD Fd sub(const Fd& a, const Fd& b) {
#ifdef ZG_FD_DEBUG
        if (!(b.ub <= (double)K)) {
            fprintf(stderr, "Fd::sub<%d>: subtrahend bound %.3f\n", K, b.ub);
            assert(false);
        }
#endif
#ifdef ZG_FD_DEBUG
        assert(a.lm + NORM < 2147483647.0 && b.lm < 1073741824.0);
#endif
        uint32_t kpv[9];
        kp<K>(kpv);
        int32_t d[9];
#pragma unroll
        for (int i = 0; i < 9; i++) d[i] = (int32_t)(a.l[i] + kpv[i]) - (int32_t)b.l[i];
        Fd r = normalise(d);
        FD_SET_UB(r, a.ub + K);
        return r;
    }

    // ---- lazy limbs ---------------------------------------------------------------------------------------------
    // limb-wise sum, no carries
    static ZG_HD Fd add_lazy(const Fd& a, const Fd& b) {
        Fd r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
#ifdef ZG_FD_DEBUG
        r.ub = a.ub + b.ub;
        assert(r.ub < CAP);
        FD_SET_LM(r, a.lm + b.lm);
#endif
        return r;
    }
    // K*p with every limb but the top one raised by 2^S (and the next one lowered by 2^(S-29) to compensate):
    // limbs 0..7 are >= 2^S - 2^(S-29), so c - b needs no borrow for limbs of b up to that
    template <int K, int S>
    static ZG_HD void kp_spread(uint32_t out[9]) {
        uint32_t base[9];
        kp<K>(base);
        constexpr uint32_t UP = 1u << S, DOWN = 1u << (S - 29);
#pragma unroll
        for (int i = 0; i < 9; i++) out[i] = base[i] + (i < 8 ? UP : 0u) - (i > 0 ? DOWN : 0u);
    }
    // a + K*p - b  without carry propagation; needs value(b) < K*p (with the top-limb margin asserted below) and
    // every limb of b at most 2^S - 2^(S-29)
    template <int K, int S>
    static ZG_HD Fd sub_lazy(const Fd& a, const Fd& b) {
        uint32_t c[9];
        kp_spread<K, S>(c);
        Fd r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (c[i] - b.l[i]);
#ifdef ZG_FD_DEBUG
        if (!(b.lm <= (double)(1u << S) - (double)(1u << (S - 29)))) {
            fprintf(stderr, "Fd::sub_lazy<%d,%d>: subtrahend limb bound %.0f\n", K, S, b.lm);
            assert(false);
        }
        // top limb: floor(V_b / 2^232) <= ub(b) * (P8 + 1) must not exceed c[8]
        if (!(b.ub * ((double)P::P[8] + 1.0) <= (double)c[8])) {
            fprintf(stderr, "Fd::sub_lazy<%d,%d>: subtrahend bound %.3f exceeds the top limb of K*p\n", K, S, b.ub);
            assert(false);
        }
        r.ub = a.ub + K;
        assert(r.ub < CAP);
        FD_SET_LM(r, a.lm + (double)(1u << S) + NORM);
#endif
        return r;
    }
    template <int K, int S>
    ZG_HD Fd neg_lazy() const {
        uint32_t c[9];
        kp_spread<K, S>(c);
        Fd r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = c[i] - l[i];
#ifdef ZG_FD_DEBUG
        assert(lm <= (double)(1u << S) - (double)(1u << (S - 29)));
        assert(ub * ((double)P::P[8] + 1.0) <= (double)c[8]);
        r.ub = K;
        FD_SET_LM(r, (double)(1u << S) + NORM);
#endif
        return r;
    }
    // carry propagation: limbs back below 2^29 (the value is unchanged)
    ZG_HD Fd norm() const {
        Fd r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t t = l[i] + c;
            r.l[i] = t & MASK;
            c = t >> 29;
        }
        r.l[8] = l[8] + c;
#ifdef ZG_FD_DEBUG
        r.ub = ub;
        r.lm = NORM;
#endif
        return r;
    }
    // cheap necessary condition for value == 0 (mod p): V = k*p with k <= KMAX means the Montgomery quotient digit of
    // the lowest limb is -k mod 2^29.  False positives have probability ~KMAX / 2^29: follow with is_zero_mod_p().
    template <int KMAX>
    ZG_HD bool maybe_zero_mod_p() const {
        const uint32_t k = (0u - l[0] * P::NINV) & MASK;   // = V * p^-1 mod 2^29
        return k <= (uint32_t)KMAX;
    }
    template <int K>
    ZG_HD Fd neg() const { return sub<K>(zero(), *this); }

    // Montgomery product, product scanning with 64-bit columns (no carries inside a column)
    friend ZG_HD Fd operator*(const Fd& a, const Fd& b) {
#ifdef ZG_FD_DEBUG
        if (!column_fits(a.lm * b.lm)) {
            fprintf(stderr, "Fd::mul: limb bounds %.0f x %.0f overflow a column\n", a.lm, b.lm);
            assert(false);
        }
#endif
        uint32_t m[9];
        Fd r;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
            m[k] = ((uint32_t)acc * P::NINV) & MASK;
            acc += (uint64_t)m[k] * P::P[0];
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * P::P[k - i];
            r.l[k - 9] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        FD_SET_UB(r, a.ub * b.ub / 168.8 + 1.0);
        return r;
    }
    ZG_HD Fd sqr() const {
#ifdef ZG_FD_DEBUG
        assert(column_fits(lm * lm) && lm < 2147483648.0);
#endif
        uint32_t m[9], a2[9];
#pragma unroll
        for (int i = 0; i < 9; i++) a2[i] = l[i] << 1;
        Fd r;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a2[i] * l[k - i];
            if ((k & 1) == 0) acc += (uint64_t)l[k >> 1] * l[k >> 1];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
            m[k] = ((uint32_t)acc * P::NINV) & MASK;
            acc += (uint64_t)m[k] * P::P[0];
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; 2 * i < k; i++) acc += (uint64_t)a2[i] * l[k - i];
            if ((k & 1) == 0) acc += (uint64_t)l[k >> 1] * l[k >> 1];
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * P::P[k - i];
            r.l[k - 9] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        FD_SET_UB(r, ub * ub / 168.8 + 1.0);
        return r;
    }

    // (a0*b0 + a1*b1) / 2^261 with one Montgomery reduction (162 + 81 multiply-adds instead of 2 x 162); the result is
    // normalised and below (A0 B0 + A1 B1) / 2^261 + p
    static ZG_HD Fd mac2(const Fd& a0, const Fd& b0, const Fd& a1, const Fd& b1) {
#ifdef ZG_FD_DEBUG
        if (!column_fits(a0.lm * b0.lm + a1.lm * b1.lm)) {
            fprintf(stderr, "Fd::mac2: limb bounds %.0f x %.0f + %.0f x %.0f overflow a column\n", a0.lm, b0.lm, a1.lm, b1.lm);
            assert(false);
        }
#endif
        uint32_t m[9];
        Fd r;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a0.l[i] * b0.l[k - i];
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a1.l[i] * b1.l[k - i];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
            m[k] = ((uint32_t)acc * P::NINV) & MASK;
            acc += (uint64_t)m[k] * P::P[0];
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (uint64_t)a0.l[i] * b0.l[k - i];
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (uint64_t)a1.l[i] * b1.l[k - i];
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * P::P[k - i];
            r.l[k - 9] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        FD_SET_UB(r, (a0.ub * b0.ub + a1.ub * b1.ub) / 168.8 + 1.0);
        return r;
    }

    // V < 2^261  ->  same residue, < 2.01 p.  q = floor(top / (p_top + 1)) never over-estimates V / p.
    ZG_HD Fd reduce() const {
        FD_NEED_NORM(*this);
        constexpr uint32_t PTOP1 = P::P[8] + 1u;                       // (p >> 232) + 1
        constexpr uint32_t RECIP = (uint32_t)(0xFFFFFFFFull / PTOP1);  // floor((2^32 - 1) / PTOP1)
        uint32_t q = (uint32_t)(((uint64_t)l[8] * RECIP) >> 32);       // <= floor(l8 / PTOP1)
        Fd r;
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int64_t t = (int64_t)l[i] - (int64_t)((uint64_t)q * P::P[i]) + c;
            r.l[i] = (uint32_t)t & MASK;
            c = t >> 29;
        }
        r.l[8] = (uint32_t)((int64_t)l[8] - (int64_t)((uint64_t)q * P::P[8]) + c);
        FD_SET_UB(r, 2.01);
        return r;
    }
    // value == 0 (mod p)?   Works for any bound below the cap.
    ZG_HD bool is_zero_mod_p() const {
        Fd r = reduce();  // < 2.01 p: candidates 0, p, 2p
        uint32_t d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) d0 |= r.l[i];
        if (d0 == 0) return true;
        uint32_t twop[9];
        kp<2>(twop);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            d1 |= r.l[i] ^ P::P[i];
            d2 |= r.l[i] ^ twop[i];
        }
        return d1 == 0 || d2 == 0;
    }
    // canonical representative in [0, p)
    ZG_HD Fd canonical() const {
        Fd r = reduce();
        for (int k = 0; k < 2; k++) {  // at most two subtractions of p
            int32_t d[9];
#pragma unroll
            for (int i = 0; i < 9; i++) d[i] = (int32_t)r.l[i] - (int32_t)P::P[i];
            Fd t = normalise(d);
            if ((int32_t)t.l[8] >= 0) r = t;  // no borrow out: r >= p
        }
        FD_SET_UB(r, 1.0);
        return r;
    }

    // ---- constants and conversions (R = 2^261)
    static ZG_HD Fd from_limbs(const uint32_t c[9]) {
        Fd r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = c[i];
        FD_SET_UB(r, 1.0);
        return r;
    }
    static ZG_HD Fd one() { return from_limbs(P::ONE); }
    static ZG_HD Fd r2() { return from_limbs(P::R2); }
    // memory loaders: canonical residues (< p) and reduce()d values (< 2.01 p)
    static ZG_HD Fd load_canon(const Fp<typename P::Wide>* p) {
        Fd r = unpack(p->l);
        FD_SET_UB(r, 1.0);
        return r;
    }
    static ZG_HD Fd load_red(const Fp<typename P::Wide>* p) {
        Fd r = unpack(p->l);
        FD_SET_UB(r, 2.01);
        return r;
    }
    // integer x < p (canonical scalar)  ->  x * 2^261
    static ZG_HD Fd to_mont(const Fd& x_int) { return x_int * r2(); }
    // ABI form x*2^256 (canonical)  ->  x*2^261, and back (result < 1.03 p; canonical() it before exporting)
    ZG_HD Fd m256_to_m261() const { return *this * from_limbs(P::C_TO261); }
    ZG_HD Fd m261_to_m256() const { return *this * from_limbs(P::C_TO256); }
    // x * 2^261 -> x  (multiply by the integer 1), canonical
    ZG_HD Fd from_mont() const {
        Fd one_int = zero();
        one_int.l[0] = 1;
        FD_SET_UB(one_int, 1.0);
        return (*this * one_int).canonical();
    }
    ZG_HD Fd pow(const uint32_t e[8]) const {
        Fd acc = one(), base = *this;
        for (int i = 0; i < 256; i++) {
            if ((e[i >> 5] >> (i & 31)) & 1) acc = acc * base;
            base = base.sqr();
        }
        return acc;
    }
    ZG_HD Fd pow_u64(uint64_t e) const {
        Fd acc = one(), base = *this;
        while (e) {
            if (e & 1) acc = acc * base;
            base = base.sqr();
            e >>= 1;
        }
        return acc;
    }
    ZG_HD Fd inv() const {  // Fermat; 0 -> 0
        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = P::Wide::MOD[i];
        e[0] -= 2;
        return pow(e);
    }
};

}  // namespace zg
