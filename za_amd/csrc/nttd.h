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

}  // namespace zg
