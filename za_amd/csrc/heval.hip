// The h query in EVALUATION form: a one-time transform of a proving key's h bases that removes the seventh
// transform (bellman's icoset_fft) from every proof.
//
// Replaces nothing in the reference by name -- it re-associates bellman_ce::groth16::create_proof's last two steps
// (reference call site prover/src/groth16/prover.rs:173):
//     h = icoset_fft(e);   answer += multiexp(params.h, h)           e = ((A B - C) / Z) on the coset g <omega>
// icoset_fft is linear, h_i = (g^-i / m) sum_j omega^(-i j) e_j, and so is the multiexp:
//     sum_i h_i H_i = sum_j e_j H'_j,      H'_j = sum_i omega^(-i j) P_i,   P_i = (g^-i / m) H_i   (P_(m-1) = identity)
// i.e. H' is the discrete Fourier transform OVER THE GROUP of the scaled h query.  The group element a proof adds is
// the same, so proof bytes do not change; the transform costs (m / 2) log2(m) scalar multiplications once per key
// (about 2 s at 2^22) and every proof saves one of its seven size-m field transforms.
// Layout: decimation in frequency, natural order in, bit-reversed out -- out[p] = H'_(bitrev(p)) -- which is the order
// h_eval_dev (ntt.hip) leaves the scalars e in.
//
// The same transform WITHOUT the coset factor gives the bases of the "C folded" form (c_fold; h_eval_dev with d_c == NULL,
// prover.hip ensure_c_fold).  e is linear in c, and c's values on the coset are the transform of its values c_j on the
// domain, so
//     sum_p e_p H'_p = sum_p (a b / Z)_p H'_p + sum_j c_j C_j,      C_j = -(1 / (m Z(g))) sum_i omega^(-i j) H_i
// (i < m - 1 in both sums -- the points the h query has; bellman truncates h there too).  C_j is [L_j(tau) Z(tau) /
// delta] scaled by -1/Z(g) and truncated: the c_j are the R1CS rows' C sides as evaluated, in row order -- out[j] = C_j,
// followed by out[m + j] = 2 C_j (a row that evaluates to 2 -- an XOR gate written (2a) b = a + b - c with a = b = 1 -- is a
// ONE on the doubled base: a quarter of a bit-level circuit's rows, which as digits would all land in one bucket).
#include "common.h"
#include "nttd.h"

namespace zg {

// k * P for a plain 256-bit integer k (8 words), binary double-and-add from the top bit.  The 64 lanes of a wave run
// in lockstep, so a sparser digit set would not save anything: some lane always adds.
template <class L>
__device__ __forceinline__ XYZZd<L> scalar_mul_xyzz(const XYZZd<L>& p, const uint32_t k[8]) {
    XYZZd<L> acc = XYZZd<L>::inf();
    for (int bit = 253; bit >= 0; bit--) {
        acc = acc.dbl();
        if ((k[bit >> 5] >> (bit & 31)) & 1u) acc = acc.add(p);
    }
    return acc;
}

__device__ __forceinline__ void integer_words(const FrD& mont, uint32_t out[8]) {
    Fr w;
    mont.from_mont().store(&w);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = w.l[i];
}

// D[i] = (g^-i / m) H_i for i < n (the first m - 1 points of the query), identity above
// (use_konst: D[i] = konst * H_i, konst a plain integer -- the c_fold form)
__global__ void __launch_bounds__(256) k_heval_init(const G1Affine* __restrict__ h, uint64_t n, uint64_t m,
                                                    const Fr* __restrict__ gi_lo, const Fr* __restrict__ gi_hi_int, int lo_bits,
                                                    G1XYZZ* __restrict__ D, int use_konst, Fr konst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (i >= n) {
        G1L::inf().store(&D[i]);
        return;
    }
    // an M261 residue times a plain integer is the plain product
    Fr sw = konst;
    if (!use_konst) {
        const FrD s = FrD::load_canon(&gi_lo[i & (((uint64_t)1 << lo_bits) - 1)]) * FrD::load_canon(&gi_hi_int[i >> lo_bits]);
        s.canonical().store(&sw);
    }
    const G1L p = G1L::from_affine(AffD<FqL>::load(&h[i]).y_canonical());
    scalar_mul_xyzz<FqL>(p, sw.l).store(&D[i]);
}

// one decimation-in-frequency stage over the group: (U, V) -> (U + V, omega^-e (U - V)), span 2^s
__global__ void __launch_bounds__(256) k_heval_stage(G1XYZZ* __restrict__ D, int log_m, int s, const FrLimbs* __restrict__ tw_inv) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ((uint64_t)1 << (log_m - 1))) return;
    const uint64_t low = t & (((uint64_t)1 << s) - 1);
    const uint64_t i0 = ((t >> s) << (s + 1)) | low, i1 = i0 | ((uint64_t)1 << s);
    const G1L u = G1L::load(&D[i0]), v = G1L::load(&D[i1]);
    u.add(v).store(&D[i0]);
    G1L d = u.add(v.neg());
    const uint64_t texp = low << (log_m - 1 - s);
    if (texp != 0) {   // (the whole stage s = 0, and the first butterfly of every block, have the twiddle 1)
        uint32_t k[8];
        integer_words(load_limbs(&tw_inv[texp]), k);
        d = scalar_mul_xyzz<FqL>(d, k);
    }
    d.store(&D[i1]);
}

// rev_bits > 0: D is in bit-reversed order, out in natural order, and out[m + j] = 2 out[j]
__global__ void __launch_bounds__(256) k_heval_affine(const G1XYZZ* __restrict__ D, uint64_t m, G1Affine* __restrict__ out, int rev_bits) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint64_t dst = rev_bits ? (uint64_t)(__brev((uint32_t)i) >> (32 - rev_bits)) : i;
    const G1L p = G1L::load(&D[i]);
    p.to_affine_mem(&out[dst]);
    if (rev_bits) p.dbl().to_affine_mem(&out[m + dst]);
}

int h_eval_bases_build(Ctx* ctx, const G1Affine* d_h, uint64_t n_h, uint32_t log_m, G1Affine* d_out, bool c_fold) {
    const uint64_t m = (uint64_t)1 << log_m;
    if (log_m == 0 || log_m >= 28) return fail(ctx, ZG_E_INVALID_ARG, "h_eval: domain 2^%u", log_m);
    if (n_h < m - 1) return fail(ctx, ZG_E_QUERY_SIZE, "h query has %llu points, need %llu", (unsigned long long)n_h,
                                 (unsigned long long)(m - 1));
    const FrLimbs* tw_inv;
    const Fr *gi_lo, *gi_hi_int;
    int lo_bits;
    Fr konst;
    ZG_TRY(ntt_tables_for_heval(ctx, log_m, &tw_inv, &gi_lo, &gi_hi_int, &lo_bits, &konst));
    DevBuf work;
    ZG_HIP(ctx, work.alloc(m * sizeof(G1XYZZ)));
    G1XYZZ* D = work.as<G1XYZZ>();
    hipStream_t st = ctx->stream;
    ZG_LAUNCH(ctx, k_heval_init, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d_h, m - 1, m, gi_lo, gi_hi_int, lo_bits, D,
              c_fold ? 1 : 0, konst);
    for (int s = (int)log_m - 1; s >= 0; s--)
        ZG_LAUNCH(ctx, k_heval_stage, dim3((unsigned)((m / 2 + 255) / 256)), dim3(256), 0, st, D, (int)log_m, s, tw_inv);
    ZG_LAUNCH(ctx, k_heval_affine, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, D, m, d_out, c_fold ? (int)log_m : 0);
    ZG_HIP(ctx, hipGetLastError());
    ZG_HIP(ctx, hipStreamSynchronize(st));   // `work` is freed on return
    return ZG_OK;
}

}  // namespace zg
