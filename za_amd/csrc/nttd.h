// Butterflies of the Fr number-theoretic transform over the lazy 9x29 field (fd.h), shared by the kernels of
// ntt.hip and by the host bound checker (tests/cpp/test_fd.cpp, -DZG_FD_DEBUG).
//
// Replaces the inner loops of bellman_ce::domain::{serial_fft, parallel_fft} (called from create_proof, reference
// call site prover/src/groth16/prover.rs:173): same radix-2 butterflies, two stages at a time on four elements held
// in registers.  Sums and differences that feed a twiddle multiplication, or the next stage's sum, are formed limb-wise
// without carry propagation ("lazy limbs", fd.h); every value written back to the LDS tile is normalised again, so the
// tile invariant is: limbs < 2^29, value bound as stated per function.
#pragma once
#include "ecd.h"

namespace zg {

// twiddle tables hold the 9 limbs of the canonical residue as they are (no unpacking in the butterfly loop), padded
// to 48 bytes so that an entry is three aligned 16-byte loads
struct alignas(16) FrLimbs {
    uint32_t l[12];
};
static ZG_HD FrD load_limbs(const FrLimbs* p) {
    FrD r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = p->l[i];
#ifdef ZG_FD_DEBUG
    r.ub = 1.0;
    r.lm = FrD::NORM;
#endif
    return r;
}
static ZG_HD void store_limbs(FrLimbs* p, const FrD& canonical) {
#pragma unroll
    for (int i = 0; i < 9; i++) p->l[i] = canonical.l[i];
    p->l[9] = p->l[10] = p->l[11] = 0;
}

// Decimation in time (Cooley-Tukey), two stages: (u, v) -> (u + v w, u - v w).
// In: four normalised values with bound B (any B up to ~140 p).  Out: normalised, bound B + 4 p.
// w1 serves both low-stage butterflies, w2 / w3 the two high-stage ones.
static ZG_HD void ntt_dit4(FrD& x0, FrD& x1, FrD& x2, FrD& x3, const FrD& w1, const FrD& w2, const FrD& w3) {
    const FrD ta = x1 * w1, tb = x3 * w1;                       // < 1.9 p whatever the inputs' bound
    const FrD a0 = FrD::add_lazy(x0, ta), a1 = FrD::sub_lazy<2, 29>(x0, ta);
    const FrD a2 = FrD::add_lazy(x2, tb), a3 = FrD::sub_lazy<2, 29>(x2, tb);
    const FrD tc = a2 * w2, td = a3 * w3;                       // factors with limbs < 2^30 resp. 3 * 2^29
    x0 = FrD::add_lazy(a0, tc).norm();
    x2 = FrD::sub_lazy<2, 29>(a0, tc).norm();
    x1 = FrD::add_lazy(a1, td).norm();
    x3 = FrD::sub_lazy<2, 29>(a1, td).norm();
}
// one stage
static ZG_HD void ntt_dit2(FrD& u, FrD& v, const FrD& w) {
    const FrD t = v * w;
    v = FrD::sub<2>(u, t);
    u = u + t;
}

// Decimation in frequency (Gentleman-Sande), two stages: (u, v) -> (u + v, (u - v) w).
// In: normalised values below 4.02 p (below 2.01 p when red_hi is false).  The sum path doubles its bound per stage:
// red_hi reduces the high-stage sums, red_lo the low-stage ones (the caller's schedule keeps what goes back to the
// tile below 4.02 p).  w2 / w3 are the high-stage twiddles, w1 the low-stage one.
static ZG_HD void ntt_dif4(FrD& x0, FrD& x1, FrD& x2, FrD& x3, const FrD& w1, const FrD& w2, const FrD& w3, bool red_hi,
                           bool red_lo) {
    FrD b0 = FrD::add_lazy(x0, x2), b1 = FrD::add_lazy(x1, x3);
    const FrD b2 = FrD::sub_lazy<5, 29>(x0, x2) * w2;
    const FrD b3 = FrD::sub_lazy<5, 29>(x1, x3) * w3;
    if (red_hi) {
        b0 = b0.norm().reduce();
        b1 = b1.norm().reduce();
    }
    const FrD y00 = FrD::add_lazy(b0, b1).norm(), y10 = FrD::add_lazy(b2, b3).norm();
    x1 = FrD::sub_lazy<9, 30>(b0, b1) * w1;                     // b1 < 8.04 p with limbs < 2^30
    x3 = FrD::sub_lazy<2, 29>(b2, b3) * w1;
    x0 = red_lo ? y00.reduce() : y00;
    x2 = red_lo ? y10.reduce() : y10;
}
static ZG_HD void ntt_dif2(FrD& u, FrD& v, const FrD& w, bool red) {
    const FrD sm = FrD::add_lazy(u, v).norm();
    v = FrD::sub_lazy<5, 29>(u, v) * w;
    u = red ? sm.reduce() : sm;
}

// ---- one LDS tile's worth of butterfly stages, shared by every transform kernel of ntt.hip and by the host bound
// checker (tests/cpp/test_fd.cpp runs the same function with one "thread" and bound tracking).
// A tile holds 2^(k+j) elements of a 2^log_m vector: stages s_lo .. s_lo+k-1.  "Strided" form (s_lo >= j): 2^k rows
// (the butterfly bits) of 2^j adjacent columns; "contiguous" form (s_lo == 0): 2^(k+j) consecutive elements.
struct TileGeom {
    int log_m, s_lo, k, j, contiguous;
    size_t hbase, lblock;
    ZG_HD uint32_t elems() const { return 1u << (k + j); }
    ZG_HD size_t gidx(uint32_t li) const {   // local index -> index in the vector
        if (contiguous) return hbase + li;
        return hbase | ((size_t)(li >> j) << s_lo) | lblock | (li & ((1u << j) - 1u));
    }
    static ZG_HD TileGeom make(int log_m, int s_lo, int k, int j, int contiguous, size_t tile) {
        TileGeom g;
        g.log_m = log_m; g.s_lo = s_lo; g.k = k; g.j = j; g.contiguous = contiguous;
        if (contiguous) {
            g.hbase = tile << (k + j);
            g.lblock = 0;
        } else {
            const size_t n_lblocks = (size_t)1 << (s_lo - j);
            g.hbase = (tile / n_lblocks) << (s_lo + k);
            g.lblock = (tile % n_lblocks) << j;
        }
        return g;
    }
};

//   DIT = false: decimation in frequency (Gentleman-Sande), stages high -> low, natural in / bit-reversed out:
//                (u, v) -> (u + v, (u - v) w).  In: normalised, below 2.01 p.  The sum path doubles its bound every
//                stage: reduce() every 2nd stage and after the last one; out: below 2.01 p... 4.02 p (see ntt_dif4).
//   DIT = true : decimation in time (Cooley-Tukey), stages low -> high, bit-reversed in / natural out:
//                (u, v) -> (u + v w, u - v w).  v always passes through a multiplication, so bounds grow by +2 p per
//                stage only: no reduction inside; in: bound B, out: B + 2 k.
// Tile: get(li) -> FrD, put(li, FrD).  `tid` / `nthreads`: this thread and the workgroup size; sync(): the barrier.
// Two stages per tile round trip: a thread owns the four elements that differ in local bits (bit, bit + 1) and runs two
// radix-2 stages on them in registers (3 twiddle loads per 4 butterflies: both low-stage butterflies share a twiddle,
// the two high-stage ones differ by omega^(m/4)); an odd stage count starts with one radix-2 stage.
template <bool DIT, class Tile, class Sync>
static ZG_HD void ntt_tile_stages(Tile& t, const TileGeom& G, const FrLimbs* __restrict__ tw, uint32_t tid, uint32_t nthreads,
                                  Sync sync) {
    const int k = G.k, s_lo = G.s_lo, log_m = G.log_m;
    const int sh = G.contiguous ? 0 : G.j;  // position of the butterfly bits inside the local index
    const uint32_t tile_elems = G.elems();
    const uint32_t half = tile_elems >> 1, quarter = tile_elems >> 2;
    // DIF: reduce the sums every 2nd stage (and after the last one); `it` counts stages in execution order
    auto reduces = [&](int it) { return (it & 1) != 0 || it == k - 1; };
    int it = 0;
    if (k & 1) {
        const int q = DIT ? 0 : k - 1;
        const int s = s_lo + q;          // global stage: butterfly span 2^s
        const int bit = sh + q;
        const bool red = reduces(0);
        for (uint32_t b = tid; b < half; b += nthreads) {
            const uint32_t i0 = ((b >> bit) << (bit + 1)) | (b & ((1u << bit) - 1u));
            const uint32_t i1 = i0 | (1u << bit);
            const size_t e0 = G.gidx(i0);
            const size_t texp = (e0 & (((size_t)1 << s) - 1)) << (log_m - 1 - s);
            FrD u = t.get(i0), v = t.get(i1);
            const FrD w = load_limbs(&tw[texp]);
            if (DIT) ntt_dit2(u, v, w);
            else ntt_dif2(u, v, w, red);
            t.put(i0, u);
            t.put(i1, v);
        }
        sync();
        it = 1;
    }
    for (; it < k; it += 2) {
        const int qlo = DIT ? it : k - 2 - it;
        const int s = s_lo + qlo;        // low stage: span 2^s; high stage: span 2^(s+1)
        const int bit = sh + qlo;
        for (uint32_t g = tid; g < quarter; g += nthreads) {
            const uint32_t i00 = ((g >> bit) << (bit + 2)) | (g & ((1u << bit) - 1u));
            const uint32_t i01 = i00 | (1u << bit), i10 = i00 | (2u << bit), i11 = i01 | i10;
            const size_t e00 = G.gidx(i00);
            const size_t lowbits = e00 & (((size_t)1 << s) - 1);
            const size_t t1 = lowbits << (log_m - 1 - s);          // low stage, both butterflies
            const size_t t2 = lowbits << (log_m - 2 - s);          // high stage, butterfly (00, 10)
            const size_t t3 = t2 + ((size_t)1 << (log_m - 2));     // high stage, butterfly (01, 11)
            FrD x0 = t.get(i00), x1 = t.get(i01), x2 = t.get(i10), x3 = t.get(i11);
            const FrD w1 = load_limbs(&tw[t1]), w2 = load_limbs(&tw[t2]), w3 = load_limbs(&tw[t3]);
            if (DIT) ntt_dit4(x0, x1, x2, x3, w1, w2, w3);
            else ntt_dif4(x0, x1, x2, x3, w1, w2, w3, reduces(it), reduces(it + 1));
            t.put(i00, x0);
            t.put(i01, x1);
            t.put(i10, x2);
            t.put(i11, x3);
        }
        sync();
    }
}

}  // namespace zg
