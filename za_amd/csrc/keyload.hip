// Parameters::<Bn256>::read(reader, checked) on the device (reference call site prover/src/groth16/format.rs:285, paid by every
// prove() of the reference: helper.rs:91-98 <- binding/c/native/src/lib.rs:81-99).
//
// A proving key is five arrays of uncompressed points (pairing_ce G1Uncompressed / G2Uncompressed: big-endian
// coordinates, byte 0 bit 7 = compression flag, bit 6 = identity), 1.4 GB at 2^22 constraints.  The host keeps the
// container walk and the few verifying-key points (formats.hip parse_bellman_header); the five queries go to the GPU as
// they are and ONE kernel per query does what `read` does per point:
//   flags (compression / identity encoding)  ->  coordinates < q  ->  Montgomery form (here: the device's M261 residues,
//   written straight into the resident query)  ->  y^2 = x^3 + b  ->  (G2) membership in the order-r subgroup
// and reports the first point (lowest index) that fails, with the class of the failure, in one 64-bit word per query.
// The host parser of formats.hip stays as the cross-check (zg_params_inspect; tests/test_gpu_keyload.py compares the
// two verdicts and messages on mutated keys).
//
// G2 membership.  pairing_ce's into_affine() tests [r]P = O: 254 doublings + 127 additions over Fq2 per point -- 0.57 ms
// per point on a host core, 30-40 s for the 2 M b_g2 points of a 2^22 key on 32 threads.  Here: the endomorphism form
//     P in G2  <=>  [u + 1]P + psi([u]P) + psi^2([u]P) = psi^3([2u]P),      u = 4965661367192848881 (the BN parameter)
// with psi = twist o Frobenius o untwist: 62 doublings + 27 mixed additions for [u]P, three psi (two Fq2 products each)
// and five more group operations.  It accepts exactly the points [r]P = O accepts: the argument
// (psi acts as q on G2 and f(q) = 0 mod r; the norm of f(psi) is coprime to the cofactor 2q - r) is written out in
// DESIGN.md section 7 and tests/test_oracle_python.py checks its two integer facts and both tests on points in and
// outside the subgroup.
// The kernel runs Fq2 on lane pairs (ecd.h Fq2PL): ~95 group operations per point, 2 waves per SIMD.
#include "common.h"
#include "formats.h"

namespace zg {

namespace {

struct KeyConsts {
    Fq b_g1;          // 3, M261
    Fq2 b_g2;         // 3 / (9 + u), M261
    Fq2 psi_x, psi_y; // xi^((q-1)/3), xi^((q-1)/2), M261
};

constexpr uint64_t BN_U = 4965661367192848881ull;

// big-endian 32 bytes (two 16-byte loads) -> 8 x u32 little-endian words
__device__ __forceinline__ void be_words(const uint4& a, const uint4& b, uint32_t w[8]) {
    w[7] = __builtin_bswap32(a.x); w[6] = __builtin_bswap32(a.y); w[5] = __builtin_bswap32(a.z); w[4] = __builtin_bswap32(a.w);
    w[3] = __builtin_bswap32(b.x); w[2] = __builtin_bswap32(b.y); w[1] = __builtin_bswap32(b.z); w[0] = __builtin_bswap32(b.w);
}
__device__ __forceinline__ uint32_t or_words(const uint32_t w[8]) {
    return w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7];
}
// integer below q?  (limbs normalised)
__device__ __forceinline__ bool below_q(const FqD& v) {
    int32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = (int32_t)v.l[i] - (int32_t)Fq29::P[i];
    const FqD t = FqD::normalise(d);
    return (int32_t)t.l[8] < 0;   // borrow out of the top limb
}
__device__ __forceinline__ void report(unsigned long long* verdict, uint64_t index, uint32_t st) {
    atomicMin(verdict, (unsigned long long)((index << 3) | st));
}

// One thread per G1 point.  raw: n points of 64 bytes (16-byte aligned), `base` = index of raw[0] in the query; the
// points with lo <= index < hi are written to out[index - lo].
__global__ void __launch_bounds__(256) k_key_decode_g1(const uint4* __restrict__ raw, uint64_t n, uint64_t base, uint64_t lo,
                                                       uint64_t hi, G1Affine* __restrict__ out, int checked, KeyConsts kc,
                                                       unsigned long long* verdict) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint4* p = raw + 4 * t;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    uint32_t xw[8], yw[8];
    be_words(a, b, xw);
    be_words(c, d, yw);
    const uint64_t idx = base + t;
    const uint32_t flags = xw[7] >> 30;
    if (flags & 2u) return report(verdict, idx, KEY_PT_COMPRESSED);
    if (flags & 1u) {
        xw[7] &= 0x3fffffffu;
        return report(verdict, idx, (or_words(xw) | or_words(yw)) ? KEY_PT_BAD_INFINITY : KEY_PT_INFINITY);
    }
    const FqD xi = FqD::unpack(xw), yi = FqD::unpack(yw);
    if (!below_q(xi) || !below_q(yi)) return report(verdict, idx, KEY_PT_NOT_IN_FIELD);
    const FqD x = FqD::to_mont(xi), y = FqD::to_mont(yi);
    if (checked) {
        const FqD rhs = x.sqr() * x + FqD::load_canon(&kc.b_g1);
        if (!FqD::sub<3>(y.sqr(), rhs).is_zero_mod_p()) return report(verdict, idx, KEY_PT_NOT_ON_CURVE);
    }
    if (idx >= lo && idx < hi) {
        G1Affine* o = out + (idx - lo);
        x.canonical().store(&o->x);
        y.canonical().store(&o->y);
    }
}

using L2 = Fq2PL;
using P2 = XYZZd<Fq2PL>;
// conj() on a lane pair: the c1 lane negates (a value below K p)
template <int K>
__device__ __forceinline__ FqD conj_pl(const FqD& v) {
    return L2::sel(L2::hi(), FqD::sub<K>(FqD::zero(), v), v);
}
__device__ __forceinline__ P2 psi_pl(const P2& q, const FqD& gx, const FqD& gy) {
    if (q.is_inf()) return q;   // (the conjugate of an exact zero would be K p on the c1 lane)
    return P2{L2::mul(conj_pl<P2::BX>(q.x), gx), L2::mul(conj_pl<P2::BY>(q.y), gy), conj_pl<P2::MB>(q.zz).reduce(),
              conj_pl<P2::MB>(q.zzz).reduce()};
}

// One LANE PAIR per G2 point (lane 2i: the c0 components, lane 2i + 1: c1).  raw: n points of 128 bytes,
// x.c1 | x.c0 | y.c1 | y.c0 big-endian.
__global__ void __launch_bounds__(256, 2) k_key_decode_g2(const uint4* __restrict__ raw, uint64_t n, uint64_t base, uint64_t lo,
                                                       uint64_t hi, G2Affine* __restrict__ out, int checked, KeyConsts kc,
                                                       unsigned long long* verdict) {
    const uint64_t t = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 1;
    if (t >= n) return;   // (both lanes of a pair leave together)
    const bool h = L2::hi();
    const uint4* p = raw + 8 * t + (h ? 0 : 2);
    const uint4 a = p[0], b = p[1], c = p[4], d = p[5];
    uint32_t xw[8], yw[8];
    be_words(a, b, xw);
    be_words(c, d, yw);
    const uint64_t idx = base + t;
    const uint32_t own_flags = h ? xw[7] >> 30 : 0u;
    const uint32_t flags = own_flags | L2::swap_u32(own_flags);
    uint32_t st = KEY_PT_OK;
    if (flags & 2u) {
        st = KEY_PT_COMPRESSED;
    } else if (flags & 1u) {
        if (h) xw[7] &= 0x3fffffffu;
        const uint32_t rest = or_words(xw) | or_words(yw);
        st = (rest | L2::swap_u32(rest)) ? KEY_PT_BAD_INFINITY : KEY_PT_INFINITY;
    }
    if (st != KEY_PT_OK) {   // pair-uniform
        if (!h) report(verdict, idx, st);
        return;
    }
    const FqD xi = FqD::unpack(xw), yi = FqD::unpack(yw);
    if (!L2::both(below_q(xi) && below_q(yi))) {
        if (!h) report(verdict, idx, KEY_PT_NOT_IN_FIELD);
        return;
    }
    const FqD x = FqD::to_mont(xi).canonical(), y = FqD::to_mont(yi).canonical();
    if (checked) {
        const FqD bc = FqD::load_canon(h ? &kc.b_g2.c1 : &kc.b_g2.c0);
        const FqD rhs = L2::mul(L2::sqr(x), x) + bc;
        if (!L2::is_zero_mod_p(FqD::sub<4>(L2::sqr(y), rhs))) {
            if (!h) report(verdict, idx, KEY_PT_NOT_ON_CURVE);
            return;
        }
        // [u]P, most significant bit first
        AffD<L2> P;
        P.x = x;
        P.y = y;
        P.inf = false;
        P2 q = P2::from_affine(P);
#pragma unroll 1
        for (int i = 61; i >= 0; i--) {
            q = q.dbl();
            if ((BN_U >> i) & 1ull) q = q.add_mixed(P);
        }
        const FqD gx = FqD::load_canon(h ? &kc.psi_x.c1 : &kc.psi_x.c0), gy = FqD::load_canon(h ? &kc.psi_y.c1 : &kc.psi_y.c0);
        // [u + 1]P + psi([u]P) + psi^2([u]P) - psi^3([2u]P), one psi image alive at a time
        P2 acc = q.add_mixed(P);
        q = psi_pl(q, gx, gy);
        acc = acc.add(q);
        q = psi_pl(q, gx, gy);
        acc = acc.add(q);
        q = psi_pl(q, gx, gy);
        const P2 diff = acc.add(q.dbl().neg());
        if (!diff.is_inf()) {
            if (!h) report(verdict, idx, KEY_PT_NOT_IN_SUBGROUP);
            return;
        }
    }
    if (idx >= lo && idx < hi) {
        G2Affine* o = out + (idx - lo);
        x.store(h ? &o->x.c1 : &o->x.c0);
        y.store(h ? &o->y.c1 : &o->y.c0);
    }
}

// ---- the way back: Parameters::write (prover/src/groth16/format.rs:250).  Resident points (device form, identity = all
// zero) -> pairing_ce's uncompressed encoding: canonical coordinates, big-endian, G2 as x.c1 | x.c0 | y.c1 | y.c0, the
// identity as 0x40 followed by zeros.
__device__ __forceinline__ void be_store(uint4* dst, const uint32_t w[8]) {   // 8 x u32 little-endian words -> 32 bytes big-endian
    dst[0] = make_uint4(__builtin_bswap32(w[7]), __builtin_bswap32(w[6]), __builtin_bswap32(w[5]), __builtin_bswap32(w[4]));
    dst[1] = make_uint4(__builtin_bswap32(w[3]), __builtin_bswap32(w[2]), __builtin_bswap32(w[1]), __builtin_bswap32(w[0]));
}
template <int NC>   // coordinates per point: 2 (G1) or 4 (G2)
__global__ void __launch_bounds__(256) k_key_encode(const Fq* __restrict__ pts, uint64_t n, uint4* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;   // one thread per coordinate
    if (t >= n * NC) return;
    const uint64_t i = t / NC;
    const uint32_t k = (uint32_t)(t % NC);
    // file order of the coordinates of a G2 point: x.c1, x.c0, y.c1, y.c0 (memory: x.c0, x.c1, y.c0, y.c1)
    const uint32_t src = NC == 4 ? (k ^ 1u) : k;
    uint32_t any = 0;
    for (int q = 0; q < NC; q++) {
        const Fq& c = pts[i * NC + q];
#pragma unroll
        for (int j = 0; j < 8; j++) any |= c.l[j];
    }
    uint32_t w[8];
    if (any == 0) {   // the identity
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = 0;
        if (k == 0) w[7] = 0x40000000u;
    } else {
        Fq c;
        FqD::load_canon(&pts[i * NC + src]).from_mont().store(&c);
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = c.l[j];
    }
    be_store(out + 2 * t, w);
}

Fq2 fq2_pow(Fq2 a, const uint32_t e[8]) {
    Fq2 r{Fq::one(), Fq::zero()};
    for (int i = 255; i >= 0; i--) {
        r = r * r;
        if ((e[i >> 5] >> (i & 31)) & 1) r = r * a;
    }
    return r;
}
// (q - 1) / d for d = 2, 3 (both divide q - 1)
void q_minus_1_over(uint32_t d, uint32_t out[8]) {
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = FqParams::MOD[i];
    w[0] -= 1;
    uint64_t rem = 0;
    for (int i = 7; i >= 0; i--) {
        const uint64_t cur = (rem << 32) | w[i];
        out[i] = (uint32_t)(cur / d);
        rem = cur % d;
    }
}
const KeyConsts& key_consts() {
    static const KeyConsts kc = [] {
        KeyConsts k;
        k.b_g1 = host_to_m261(Fq::from_u64(3));
        k.b_g2 = host_to_m261(g2_coeff_b());
        const Fq2 xi{Fq::from_u64(9), Fq::from_u64(1)};
        uint32_t e[8];
        q_minus_1_over(3, e);
        k.psi_x = host_to_m261(fq2_pow(xi, e));
        q_minus_1_over(2, e);
        k.psi_y = host_to_m261(fq2_pow(xi, e));
        return k;
    }();
    return kc;
}

}  // namespace

const char* key_query_name(int which) {
    static const char* names[5] = {"h", "l", "a", "b_g1", "b_g2"};
    return which >= 0 && which < 5 ? names[which] : "?";
}

// Queue the decoding (and, checked, the validation) of one query on ctx->stream.  host_raw: the query's n uncompressed
// points as they stand in the key; d_out receives the points [lo, hi) in device form; *d_verdict (device, preset to ~0)
// takes the minimum of (index << 3 | class) over the points that fail.  Points outside [lo, hi) are only uploaded when
// they are to be validated.  stage: two device buffers the raw bytes pass through (grown on demand).
int key_decode_query_dev(Ctx* ctx, int which, const uint8_t* host_raw, uint64_t n, bool checked, uint64_t lo, uint64_t hi,
                         void* d_out, unsigned long long* d_verdict, KeyStage* stage) {
    const bool g2 = which == 4;
    const size_t elem = g2 ? 128 : 64;
    uint64_t first = checked ? 0 : lo, last = checked ? n : hi;   // the points that travel
    if (last > n) last = n;
    if (first >= last) return ZG_OK;
    const uint64_t chunk_pts = KeyStage::CHUNK_BYTES / elem;
    const KeyConsts& kc = key_consts();
    for (uint64_t at = first; at < last; at += chunk_pts) {
        const uint64_t cnt = std::min<uint64_t>(chunk_pts, last - at);
        const int s = stage->next;
        stage->next ^= 1;
        if (!stage->ev[s]) ZG_HIP(ctx, hipEventCreateWithFlags(&stage->ev[s], hipEventDisableTiming));
        ZG_HIP(ctx, stage->buf[s].reserve(KeyStage::CHUNK_BYTES));
        // (the copy below may return before the engine has read a pageable source only after it has staged it: safe; what
        // must not happen is the NEXT copy into this buffer overtaking the kernel that still reads it -- same stream, so
        // it cannot)
        ZG_HIP(ctx, hipMemcpyAsync(stage->buf[s].p, host_raw + at * elem, cnt * elem, hipMemcpyHostToDevice, ctx->stream));
        if (g2)
            ZG_LAUNCH(ctx, k_key_decode_g2, dim3((unsigned)((2 * cnt + 255) / 256)), dim3(256), 0, ctx->stream,
                      stage->buf[s].as<uint4>(), cnt, at, lo, hi, (G2Affine*)d_out, checked ? 1 : 0, kc, d_verdict);
        else
            ZG_LAUNCH(ctx, k_key_decode_g1, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream,
                      stage->buf[s].as<uint4>(), cnt, at, lo, hi, (G1Affine*)d_out, checked ? 1 : 0, kc, d_verdict);
        ZG_HIP(ctx, hipGetLastError());
    }
    return ZG_OK;
}

// Queue-and-copy: the n resident points of one query (device form) -> their uncompressed encoding in host_out (n * 64 or
// n * 128 bytes).  Blocking.
int key_encode_query_dev(Ctx* ctx, int which, const void* d_points, uint64_t n, uint8_t* host_out, KeyStage* stage) {
    const bool g2 = which == 4;
    const size_t elem = g2 ? 128 : 64;
    const uint64_t chunk_pts = KeyStage::CHUNK_BYTES / elem;
    for (uint64_t at = 0; at < n; at += chunk_pts) {
        const uint64_t cnt = std::min<uint64_t>(chunk_pts, n - at);
        const int s = stage->next;
        stage->next ^= 1;
        ZG_HIP(ctx, stage->buf[s].reserve(KeyStage::CHUNK_BYTES));
        const Fq* src = (const Fq*)((const uint8_t*)d_points + at * elem);
        if (g2)
            ZG_LAUNCH(ctx, (k_key_encode<4>), dim3((unsigned)((cnt * 4 + 255) / 256)), dim3(256), 0, ctx->stream, src, cnt, stage->buf[s].as<uint4>());
        else
            ZG_LAUNCH(ctx, (k_key_encode<2>), dim3((unsigned)((cnt * 2 + 255) / 256)), dim3(256), 0, ctx->stream, src, cnt, stage->buf[s].as<uint4>());
        ZG_HIP(ctx, hipGetLastError());
        ZG_HIP(ctx, hipMemcpyAsync(host_out + at * elem, stage->buf[s].p, cnt * elem, hipMemcpyDeviceToHost, ctx->stream));
    }
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZG_OK;
}

// the verdict words of the five queries -> bellman's error (the first query in file order that has a bad point, its
// lowest bad index), worded like the host parser's
int key_verdict_to_error(Ctx* ctx, const unsigned long long verdict[5]) {
    for (int which = 0; which < 5; which++) {
        if (verdict[which] == ~0ull) continue;
        return key_point_error(ctx, key_query_name(which), (int)(verdict[which] & 7u));
    }
    return ZG_OK;
}

}  // namespace zg
