// Radix-2 number-theoretic transform over BN254 Fr and the H-polynomial pipeline for gfx950.
//
// Replaces bellman_ce::domain::EvaluationDomain::{fft, ifft, coset_fft, icoset_fft, mul_assign,
// sub_assign, divide_by_z_on_coset} as sequenced inside create_proof (reference call site
// prover/src/groth16/prover.rs:173).  Bellman does an in-place bit-reversal + log2(m) butterfly
// sweeps over the whole vector per transform; here each HBM pass performs up to NTT_KMAX butterfly
// stages on an LDS-resident tile (decimation in frequency, natural order in, bit-reversed out),
// followed by one bit-reversal permutation, so a 2^22 transform touches HBM 3+1 times instead of 22.
// omega = root_of_unity^(2^(28-log_m)), root_of_unity = 7^((r-1)/2^28); coset generator g = 7.
// Algorithmic HBM bytes: 64 B per element per transform (one read + one write).
// Field work uses the lazy 9x29-limb arithmetic (fd.h): HBM holds M261 residues (x*2^261) packed in
// 8 x u32; the LDS tile holds 9 limbs per element (36-byte stride: conflict-free for ds_read_b32).
#include <cstring>
#include <vector>

#include "common.h"
#include "nttd.h"

namespace zg {

static constexpr int NTT_KMAX = 8;        // butterfly stages fused per pass
static constexpr int NTT_THREADS = 256;   // one butterfly quad per thread and round; four 36 KB tiles per CU (4 waves per SIMD)
static constexpr int NTT_LOG_TILE = 10;   // 2^10 elements * 36 B = 36 KiB of LDS per workgroup: four workgroups per CU in different phases
                                          // (load / butterflies / store) overlap where two 72 KB tiles of 512 threads did not:
                                          // 2^22 H pipeline alone 4.79 -> 4.44 ms (2^9 tiles / 128 threads: 4.72)
static constexpr int GEOM_LO = 10;        // low-table bits for the g^i two-level power tables

struct NttTables {
    uint32_t log_m = 0;
    DevBuf tw_fwd, tw_inv;              // omega^i, omega^-i, i < m/2   (canonical M261 residues as 9 limbs, FrLimbs)
    DevBuf g_lo, g_hi, g_hi_minv;       // g^j (j < 2^lo), g^(k 2^lo), g^(k 2^lo)/m       (M261)
    DevBuf gi_lo, gi_hi, gi_hi_canon;   // g^-j, g^-(k 2^lo)/m (Montgomery) and the same hi table in canonical form
    Fr minv;                            // 1/m          (M261, for kernels)
    Fr zinv;                            // 1/(g^m - 1)  (M261, for kernels)
    Fr minv_host;                       // 1/m in the host's 2^256 form
    int lo_bits = 0;
};

// ------------------------------------------------------------------------------------------------ table kernels
// out[i] = scale * base^i for i < n (M261, canonical); each thread owns a run of `run` consecutive powers
__global__ void __launch_bounds__(256) k_powers(Fr* __restrict__ out, size_t n, Fr base_m, Fr scale_m, uint32_t run) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = t * run;
    if (lo >= n) return;
    size_t hi = lo + run < n ? lo + run : n;
    const FrD base = FrD::load_canon(&base_m);
    FrD cur = base.pow_u64(lo) * FrD::load_canon(&scale_m);
    for (size_t i = lo; i < hi; i++) {
        cur.canonical().store(&out[i]);
        cur = cur * base;
    }
}
// the same powers in limb form (twiddle tables: nttd.h FrLimbs)
__global__ void __launch_bounds__(256) k_powers_limbs(FrLimbs* __restrict__ out, size_t n, Fr base_m, uint32_t run) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = t * run;
    if (lo >= n) return;
    size_t hi = lo + run < n ? lo + run : n;
    const FrD base = FrD::load_canon(&base_m);
    FrD cur = base.pow_u64(lo);
    for (size_t i = lo; i < hi; i++) {
        store_limbs(&out[i], cur.canonical());
        cur = cur * base;
    }
}
// M261 residue -> plain integer (canonical)
__global__ void __launch_bounds__(256) k_from_mont(Fr* __restrict__ data, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) FrD::load_canon(&data[i]).from_mont().store(&data[i]);
}
// ABI Montgomery form (x*2^256) <-> device form (x*2^261), in place; which: 0 Fq, 1 Fr
template <class P29>
__global__ void __launch_bounds__(256) k_convert(Fp<typename P29::Wide>* __restrict__ data, size_t n, int dir) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fd<P29> v = Fd<P29>::load_canon(&data[i]);
    v = dir == 0 ? v.m256_to_m261() : v.m261_to_m256();
    v.canonical().store(&data[i]);
}

static int launch_powers(Ctx* ctx, Fr* out, size_t n, const Fr& base, const Fr& scale) {
    uint32_t run = 64;
    size_t threads = (n + run - 1) / run;
    hipLaunchKernelGGL(k_powers, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, out, n,
                       host_to_m261(base), host_to_m261(scale), run);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

static int launch_powers_limbs(Ctx* ctx, FrLimbs* out, size_t n, const Fr& base) {
    uint32_t run = 64;
    size_t threads = (n + run - 1) / run;
    hipLaunchKernelGGL(k_powers_limbs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, out, n,
                       host_to_m261(base), run);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

static Fr fr_root_of_unity() {
    // 7^((r-1)/2^28)
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = FrParams::MOD[i];
    e[0] -= 1;
    uint32_t t[8];
    for (int i = 0; i < 8; i++) t[i] = (e[i] >> 28) | (i < 7 ? e[i + 1] << 4 : 0);
    return Fr::from_u64(7).pow(t);
}

static int get_tables(Ctx* ctx, uint32_t log_m, NttTables** out) {
    if (log_m >= 28) return fail(ctx, ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE, "domain 2^%u >= 2^28", log_m);
    if (ctx->ntt[log_m]) {
        *out = ctx->ntt[log_m];
        return ZG_OK;
    }
    NttTables* T = new NttTables();
    T->log_m = log_m;
    const size_t m = (size_t)1 << log_m;
    Fr omega = fr_root_of_unity();
    for (uint32_t i = log_m; i < 28; i++) omega = omega.sqr();
    Fr omega_inv = omega.inv();
    Fr g = Fr::from_u64(7), g_inv = g.inv();
    T->minv_host = Fr::from_u64(m).inv();
    T->minv = host_to_m261(T->minv_host);
    T->zinv = host_to_m261((g.pow_u64(m) - Fr::one()).inv());
    const size_t half = m > 1 ? m / 2 : 1;
    int rc = ZG_OK;
    auto A = [&](DevBuf& b, size_t n) {
        if (rc == ZG_OK && b.alloc(n * sizeof(Fr)) != hipSuccess) rc = fail(ctx, ZG_E_HIP, "hipMalloc of NTT table failed");
    };
    T->lo_bits = (int)log_m < GEOM_LO ? (int)log_m : GEOM_LO;
    const size_t nlo = (size_t)1 << T->lo_bits, nhi = m >> T->lo_bits;
    if (T->tw_fwd.alloc(half * sizeof(FrLimbs)) != hipSuccess || T->tw_inv.alloc(half * sizeof(FrLimbs)) != hipSuccess)
        rc = fail(ctx, ZG_E_HIP, "hipMalloc of NTT table failed");
    A(T->g_lo, nlo); A(T->g_hi, nhi); A(T->g_hi_minv, nhi); A(T->gi_lo, nlo); A(T->gi_hi, nhi); A(T->gi_hi_canon, nhi);
    if (rc != ZG_OK) { delete T; return rc; }
    rc = launch_powers_limbs(ctx, T->tw_fwd.as<FrLimbs>(), half, omega);
    if (!rc) rc = launch_powers_limbs(ctx, T->tw_inv.as<FrLimbs>(), half, omega_inv);
    if (!rc) rc = launch_powers(ctx, T->g_lo.as<Fr>(), nlo, g, Fr::one());
    if (!rc) rc = launch_powers(ctx, T->g_hi.as<Fr>(), nhi, g.pow_u64(nlo), Fr::one());
    if (!rc) rc = launch_powers(ctx, T->g_hi_minv.as<Fr>(), nhi, g.pow_u64(nlo), T->minv_host);
    if (!rc) rc = launch_powers(ctx, T->gi_lo.as<Fr>(), nlo, g_inv, Fr::one());
    if (!rc) rc = launch_powers(ctx, T->gi_hi.as<Fr>(), nhi, g_inv.pow_u64(nlo), T->minv_host);
    if (!rc) rc = launch_powers(ctx, T->gi_hi_canon.as<Fr>(), nhi, g_inv.pow_u64(nlo), T->minv_host);
    if (!rc) {
        hipLaunchKernelGGL(k_from_mont, dim3((unsigned)((nhi + 255) / 256)), dim3(256), 0, ctx->stream,
                           T->gi_hi_canon.as<Fr>(), nhi);
        if (hipGetLastError() != hipSuccess) rc = fail(ctx, ZG_E_HIP, "k_from_mont launch failed");
    }
    if (rc) { delete T; return rc; }
    ctx->ntt[log_m] = T;
    *out = T;
    return ZG_OK;
}

void ntt_tables_free(Ctx* ctx) {
    for (int i = 0; i < 28; i++) {
        delete ctx->ntt[i];
        ctx->ntt[i] = nullptr;
    }
}

// ------------------------------------------------------------------------------------------------ transform pass
// One in-place pass of k butterfly stages s_lo .. s_lo+k-1 over data[0..2^log_m) on an LDS tile.
// A workgroup owns 2^(k+j) elements.  If s_lo >= j ("strided" form) they are 2^k rows (the k butterfly
// bits) of 2^j adjacent columns, so global accesses are 2^j*32 B contiguous runs; otherwise
// (s_lo == 0, "contiguous" form) the tile is 2^(k+j) consecutive elements.
//   DIT = false: decimation in frequency (Gentleman-Sande), stages high -> low, natural in / bit-reversed out:
//                (u, v) -> (u + v, (u - v) w).  The sum path doubles its bound every stage: reduce() every 2nd.
//   DIT = true : decimation in time (Cooley-Tukey), stages low -> high, bit-reversed in / natural out:
//                (u, v) -> (u + v w, u - v w).  v always passes through a multiplication, so bounds grow by
//                +2 p per stage only: no reduction inside the pass.
// Fused element-wise work (uniform switches):
//   pre  1: x *= hi[i >> lo_bits] * lo[i & mask]  (coset / 1/m distribution by natural index i)
//        2: x = (a*b - c) * zinv                  (the pointwise step of H; a is `data`)
//   post 1: x *= s                                (1/m)
//        2: x *= hi[i >> lo_bits] * lo[i & mask]
//        3: out[i] = integer(x * lo[..] * hi_int[..])   (final scaling fused with leaving Montgomery form)
struct PassFuse {
    int pre, post, lo_bits;
    const Fr *hi, *lo;      // pre tables
    const Fr *b, *c;        // pre 2
    Fr s;                   // pre 2: zinv; post 1: constant
    const Fr *phi, *plo;    // post tables
    Fr* out;                // post 3
};

template <bool DIT>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_pass(Fr* __restrict__ data, const FrLimbs* __restrict__ tw, int log_m, int s_lo,
                                                  int k, int j, int contiguous, PassFuse fz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* lds = reinterpret_cast<uint32_t*>(smem);   // 9 limbs per element
    const int s_hi = s_lo + k - 1;
    const uint32_t tile_elems = 1u << (k + j);
    const size_t tile = blockIdx.x;
    size_t hbase, lblock;
    if (contiguous) {
        hbase = tile << (k + j);
        lblock = 0;
    } else {
        const size_t n_lblocks = (size_t)1 << (s_lo - j);
        hbase = (tile / n_lblocks) << (s_hi + 1);
        lblock = (tile % n_lblocks) << j;
    }
    auto gidx = [&](uint32_t li) -> size_t {
        if (contiguous) return hbase + li;
        return hbase | ((size_t)(li >> j) << s_lo) | lblock | (li & ((1u << j) - 1u));
    };
    auto lds_get = [&](uint32_t li) {
        FrD v;
#pragma unroll
        for (int q = 0; q < 9; q++) v.l[q] = lds[9 * li + q];
        return v;
    };
    auto lds_put = [&](uint32_t li, const FrD& v) {
#pragma unroll
        for (int q = 0; q < 9; q++) lds[9 * li + q] = v.l[q];
    };
    const size_t lomask = ((size_t)1 << fz.lo_bits) - 1;
    auto ingest = [&](uint32_t li, size_t g, const Fr& raw) {
        FrD x = FrD::load_red(&raw);
        if (fz.pre == 1) {
            x = x * (FrD::load_canon(&fz.hi[g >> fz.lo_bits]) * FrD::load_canon(&fz.lo[g & lomask]));
        } else if (fz.pre == 2) {
            FrD ab = x * FrD::load_red(&fz.b[g]);
            x = FrD::sub<3>(ab, FrD::load_red(&fz.c[g])) * FrD::load_canon(&fz.s);
        }
        lds_put(li, x);
    };
    if (tile_elems == 4u * NTT_THREADS) {
        // the usual case: four elements per thread.  All four 32-byte loads are issued before the first is used, so the
        // tile's load phase costs one HBM latency instead of four.
        Fr raw[4];
        size_t g4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            g4[q] = gidx(threadIdx.x + q * NTT_THREADS);
            raw[q] = data[g4[q]];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) ingest(threadIdx.x + q * NTT_THREADS, g4[q], raw[q]);
    } else {
        for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) {
            const size_t g = gidx(li);
            ingest(li, g, data[g]);
        }
    }
    __syncthreads();
    const int sh = contiguous ? 0 : j;  // position of the butterfly bits inside the local index
    const uint32_t half = tile_elems >> 1;
    // DIF: reduce the sums every 2nd stage (and after the last one); `it` counts stages in execution order
    auto reduces = [&](int it) { return (it & 1) != 0 || it == k - 1; };
    int it = 0;
    if (k & 1) {   // an odd stage count: one radix-2 stage first, radix-4 pairs after it
        const int q = DIT ? 0 : k - 1;
        const int s = s_lo + q;          // global stage: butterfly span 2^s
        const int bit = sh + q;
        const bool red = reduces(0);
        for (uint32_t b = threadIdx.x; b < half; b += blockDim.x) {
            uint32_t i0 = ((b >> bit) << (bit + 1)) | (b & ((1u << bit) - 1u));
            uint32_t i1 = i0 | (1u << bit);
            size_t e0 = gidx(i0);
            size_t texp = (e0 & (((size_t)1 << s) - 1)) << (log_m - 1 - s);
            FrD u = lds_get(i0), v = lds_get(i1);
            const FrD w = load_limbs(&tw[texp]);
            if (DIT) ntt_dit2(u, v, w);          // bounds grow by at most 2 p per stage: <= 2.01 + 2k
            else ntt_dif2(u, v, w, red);         // v <= 4.02 p on entry
            lds_put(i0, u);
            lds_put(i1, v);
        }
        __syncthreads();
        it = 1;
    }
    // Two stages per LDS round trip: a thread owns the four elements that differ in local bits (bit, bit + 1) and
    // runs the two radix-2 stages on them in registers -- the same arithmetic, element by element, as two separate
    // stages (so the lazy bounds above still hold), with half the LDS traffic and barriers and 3 twiddle loads
    // instead of 4: both low-stage butterflies share a twiddle, the two high-stage ones differ by omega^(m/4).
    const uint32_t quarter = tile_elems >> 2;
    for (; it < k; it += 2) {
        const int qlo = DIT ? it : k - 2 - it;
        const int s = s_lo + qlo;        // low stage: span 2^s; high stage: span 2^(s+1)
        const int bit = sh + qlo;
        for (uint32_t g = threadIdx.x; g < quarter; g += blockDim.x) {
            const uint32_t i00 = ((g >> bit) << (bit + 2)) | (g & ((1u << bit) - 1u));
            const uint32_t i01 = i00 | (1u << bit), i10 = i00 | (2u << bit), i11 = i01 | i10;
            const size_t e00 = gidx(i00);
            const size_t lowbits = e00 & (((size_t)1 << s) - 1);
            const size_t t1 = lowbits << (log_m - 1 - s);          // low stage, both butterflies
            const size_t t2 = lowbits << (log_m - 2 - s);          // high stage, butterfly (00, 10)
            const size_t t3 = t2 + ((size_t)1 << (log_m - 2));     // high stage, butterfly (01, 11)
            FrD x0 = lds_get(i00), x1 = lds_get(i01), x2 = lds_get(i10), x3 = lds_get(i11);
            const FrD w1 = load_limbs(&tw[t1]), w2 = load_limbs(&tw[t2]), w3 = load_limbs(&tw[t3]);
            if (DIT) ntt_dit4(x0, x1, x2, x3, w1, w2, w3);
            else ntt_dif4(x0, x1, x2, x3, w1, w2, w3, reduces(it), reduces(it + 1));
            lds_put(i00, x0);
            lds_put(i01, x1);
            lds_put(i10, x2);
            lds_put(i11, x3);
        }
        __syncthreads();
    }
    for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) {
        const size_t g = gidx(li);
        FrD x = lds_get(li);
        if (fz.post == 0) {
            x.reduce().store(&data[g]);
        } else if (fz.post == 1) {
            (x * FrD::load_canon(&fz.s)).reduce().store(&data[g]);
        } else if (fz.post == 2) {
            (x * (FrD::load_canon(&fz.phi[g >> fz.lo_bits]) * FrD::load_canon(&fz.plo[g & lomask]))).reduce().store(&data[g]);
        } else {
            FrD t = x * FrD::load_canon(&fz.plo[g & lomask]);
            (t * FrD::load_canon(&fz.phi[g >> fz.lo_bits])).canonical().store(&fz.out[g]);
        }
    }
}

__global__ void __launch_bounds__(256) k_bitrev(Fr* __restrict__ data, int log_m) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << log_m)) return;
    size_t r = (size_t)(__brev((uint32_t)i) >> (32 - log_m));
    if (i < r) {
        Fr a = data[i], b = data[r];
        data[i] = b;
        data[r] = a;
    }
}

// data[i] *= hi[i >> lo_bits] * lo[i & mask]          (tables canonical M261)
__global__ void __launch_bounds__(256) k_scale_geom(Fr* __restrict__ data, size_t n, const Fr* __restrict__ hi,
                                                    const Fr* __restrict__ lo, int lo_bits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FrD s = FrD::load_canon(&hi[i >> lo_bits]) * FrD::load_canon(&lo[i & (((size_t)1 << lo_bits) - 1)]);
    (FrD::load_red(&data[i]) * s).reduce().store(&data[i]);
}
// out[i] = integer( data[i] * lo[i & mask] * hi_int[i >> lo_bits] )  -- leaves Montgomery form:
// an M261 residue times a plain integer gives the plain product
__global__ void __launch_bounds__(256) k_scale_geom_to_repr(const Fr* __restrict__ data, Fr* __restrict__ out, size_t n,
                                                            const Fr* __restrict__ hi_int, const Fr* __restrict__ lo,
                                                            int lo_bits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FrD t = FrD::load_red(&data[i]) * FrD::load_canon(&lo[i & (((size_t)1 << lo_bits) - 1)]);
    (t * FrD::load_canon(&hi_int[i >> lo_bits])).canonical().store(&out[i]);
}
__global__ void __launch_bounds__(256) k_scale_const(Fr* __restrict__ data, size_t n, Fr s_m) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) (FrD::load_red(&data[i]) * FrD::load_canon(&s_m)).reduce().store(&data[i]);
}
// a[i] = (a[i]*b[i] - c[i]) * zinv
__global__ void __launch_bounds__(256) k_mul_sub_scale(Fr* __restrict__ a, const Fr* __restrict__ b,
                                                       const Fr* __restrict__ c, size_t n, Fr zinv_m) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FrD ab = FrD::load_red(&a[i]) * FrD::load_red(&b[i]);
    (FrD::sub<3>(ab, FrD::load_red(&c[i])) * FrD::load_canon(&zinv_m)).reduce().store(&a[i]);
}

static inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

static PassFuse no_fuse() {
    PassFuse f;
    memset(&f, 0, sizeof(f));
    return f;
}

// All passes of one transform.  DIF: natural in -> bit-reversed out (stages high -> low); DIT: bit-reversed in ->
// natural out (stages low -> high).  `first` / `last` carry the element-wise work fused into the first pass's
// loads / the last pass's stores.
template <bool DIT>
static int run_passes(Ctx* ctx, Fr* d, const FrLimbs* tw, uint32_t log_m, const PassFuse& first, const PassFuse& last) {
    static bool lds_attr_set = false;
    if (!lds_attr_set) {
        ZG_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_pass<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        36 << NTT_LOG_TILE));
        ZG_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_pass<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        36 << NTT_LOG_TILE));
        lds_attr_set = true;
    }
    // split the log_m stages into passes of (nearly) equal size <= NTT_KMAX
    int np = ((int)log_m + NTT_KMAX - 1) / NTT_KMAX;
    if (np < 1) np = 1;
    std::vector<int> ks;
    int remaining = (int)log_m;
    for (int p = np; p > 0; p--) {
        int k = (remaining + p - 1) / p;
        ks.push_back(k);
        remaining -= k;
    }
    // DIF walks the stages from the top, DIT from the bottom
    int s_next = DIT ? 0 : (int)log_m;
    for (size_t pi = 0; pi < ks.size(); pi++) {
        const int k = ks[pi];
        const int s_lo = DIT ? s_next : s_next - k;
        s_next = DIT ? s_next + k : s_next - k;
        int j, contiguous;
        if (s_lo == 0) {
            contiguous = 1;
            j = NTT_LOG_TILE - k;
            if (k + j > (int)log_m) j = (int)log_m - k;
        } else {
            contiguous = 0;
            j = NTT_LOG_TILE - k;
            if (j > s_lo) j = s_lo;
        }
        if (j < 0) j = 0;
        PassFuse fz = no_fuse();
        if (pi == 0) {
            fz.pre = first.pre; fz.hi = first.hi; fz.lo = first.lo; fz.b = first.b; fz.c = first.c; fz.lo_bits = first.lo_bits;
            if (first.pre == 2) fz.s = first.s;
        }
        if (pi + 1 == ks.size()) {
            fz.post = last.post; fz.phi = last.phi; fz.plo = last.plo; fz.out = last.out;
            if (last.post) fz.lo_bits = last.lo_bits;
            if (last.post == 1) fz.s = last.s;
        }
        if (fz.pre == 2 && fz.post == 1) return fail(ctx, ZG_E_INVALID_ARG, "internal: PassFuse.s used twice");
        size_t tiles = ((size_t)1 << log_m) >> (k + j);
        size_t lds_bytes = (size_t)36 << (k + j);   // 9 u32 limbs per element
        ProfScope ps(ctx, PROF_NTT_PASS, (uint64_t)64 << log_m, nullptr, ((uint64_t)k << log_m) >> 1);  // units: butterflies
        hipLaunchKernelGGL((k_ntt_pass<DIT>), dim3((unsigned)tiles), dim3(NTT_THREADS), lds_bytes, ctx->stream, d, tw, (int)log_m,
                           s_lo, k, j, contiguous, fz);
        ZG_HIP(ctx, hipGetLastError());
    }
    return ZG_OK;
}

// natural order in -> bit-reversed order out
static int dif_passes(Ctx* ctx, Fr* d, const FrLimbs* tw, uint32_t log_m) {
    return run_passes<false>(ctx, d, tw, log_m, no_fuse(), no_fuse());
}

static int ntt_core(Ctx* ctx, NttTables* T, Fr* d, bool inverse, bool coset) {
    const uint32_t log_m = T->log_m;
    const size_t m = (size_t)1 << log_m;
    hipStream_t st = ctx->stream;
    if (!inverse && coset) {
        hipLaunchKernelGGL(k_scale_geom, dim3(blocks_for(m)), dim3(256), 0, st, d, m, T->g_hi.as<Fr>(), T->g_lo.as<Fr>(),
                           T->lo_bits);
    }
    if (log_m > 0) {
        ZG_TRY(dif_passes(ctx, d, inverse ? T->tw_inv.as<FrLimbs>() : T->tw_fwd.as<FrLimbs>(), log_m));
        hipLaunchKernelGGL(k_bitrev, dim3(blocks_for(m)), dim3(256), 0, st, d, (int)log_m);
    }
    if (inverse) {
        if (coset)
            hipLaunchKernelGGL(k_scale_geom, dim3(blocks_for(m)), dim3(256), 0, st, d, m, T->gi_hi.as<Fr>(),
                               T->gi_lo.as<Fr>(), T->lo_bits);
        else
            hipLaunchKernelGGL(k_scale_const, dim3(blocks_for(m)), dim3(256), 0, st, d, m, T->minv);
    }
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

// out[j] = L_j(tau) (Lagrange basis of the 2^log_m domain at tau) as plain integers, natural order:
// the inverse transform of the powers of tau, exactly how bellman's generator obtains them.
int lagrange_at_tau_dev(Ctx* ctx, uint32_t log_m, const Fr& tau_host, Fr* d_out) {
    NttTables* T;
    ZG_TRY(get_tables(ctx, log_m, &T));
    const size_t m = (size_t)1 << log_m;
    ZG_TRY(launch_powers(ctx, d_out, m, tau_host, Fr::one()));
    ZG_TRY(ntt_core(ctx, T, d_out, /*inverse=*/true, /*coset=*/false));
    hipLaunchKernelGGL(k_from_mont, dim3(blocks_for(m)), dim3(256), 0, ctx->stream, d_out, m);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}
// out[i] = integer(scale * base^i), i < n
int powers_int_dev(Ctx* ctx, Fr* d_out, size_t n, const Fr& base_host, const Fr& scale_host) {
    if (n == 0) return ZG_OK;
    ZG_TRY(launch_powers(ctx, d_out, n, base_host, scale_host));
    hipLaunchKernelGGL(k_from_mont, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, d_out, n);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

int bitrev_dev(Ctx* ctx, Fr* d_data, uint32_t log_m) {
    if (log_m == 0) return ZG_OK;
    hipLaunchKernelGGL(k_bitrev, dim3(blocks_for((size_t)1 << log_m)), dim3(256), 0, ctx->stream, d_data, (int)log_m);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

int convert_mont_dev(Ctx* ctx, void* d_elems, size_t n, int which, int dir, hipStream_t st) {
    if (n == 0) return ZG_OK;
    if (!st) st = ctx->stream;
    if (which)
        hipLaunchKernelGGL((k_convert<Fr29>), dim3(blocks_for(n)), dim3(256), 0, st, (Fr*)d_elems, n, dir);
    else
        hipLaunchKernelGGL((k_convert<Fq29>), dim3(blocks_for(n)), dim3(256), 0, st, (Fq*)d_elems, n, dir);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

// ABI entry: data in the caller's 2^256 Montgomery form, converted to the device form and back
int ntt_fr_dev(Ctx* ctx, Fr* d_data, uint32_t log_m, bool inverse, bool coset) {
    NttTables* T;
    ZG_TRY(get_tables(ctx, log_m, &T));
    const size_t m = (size_t)1 << log_m;
    ZG_TRY(convert_mont_dev(ctx, d_data, m, 1, 0));
    ZG_TRY(ntt_core(ctx, T, d_data, inverse, coset));
    return convert_mont_dev(ctx, d_data, m, 1, 1);
}

// H = (A*B - C)/Z.  a, b, c: device form (M261, reduce()d) with row i stored at index bitrev(i)
// (k_r1cs_eval scatters them that way), zero padded.  d_h_out receives the m coefficients as plain integers in
// natural order.  Bellman's sequence ifft, coset_fft (x3), mul/sub/divide-by-Z, icoset_fft becomes
//   DIT(omega^-1)            bit-reversed evaluations -> natural coefficients       [x3]
//   DIF(omega), pre-scaled   coefficients * g^i / m   -> bit-reversed coset evaluations   [x3]
//   DIT(omega^-1), pre = (a*b - c)/Z(g), post = * g^-i / m -> integer coefficients of H, natural order
// i.e. 21 LDS passes and no separate permutation or scaling pass.
int h_poly_dev(Ctx* ctx, Fr* d_a, Fr* d_b, Fr* d_c, uint32_t log_m, Fr* d_h_out) {
    NttTables* T;
    ZG_TRY(get_tables(ctx, log_m, &T));
    Fr* v[3] = {d_a, d_b, d_c};
    PassFuse pre_g = no_fuse();   // coefficients * g^i * (1/m): the 1/m of the preceding inverse transform rides along
    pre_g.pre = 1; pre_g.hi = T->g_hi_minv.as<Fr>(); pre_g.lo = T->g_lo.as<Fr>(); pre_g.lo_bits = T->lo_bits;
    // (one launch per pass for all of a, b, c was measured: no gain, 4.46 -> 4.57 ms for the pipeline alone)
    for (int k = 0; k < 3; k++) {
        ZG_TRY(run_passes<true>(ctx, v[k], T->tw_inv.as<FrLimbs>(), log_m, no_fuse(), no_fuse()));
        ZG_TRY(run_passes<false>(ctx, v[k], T->tw_fwd.as<FrLimbs>(), log_m, pre_g, no_fuse()));
    }
    PassFuse pre_m = no_fuse();
    pre_m.pre = 2; pre_m.b = d_b; pre_m.c = d_c; pre_m.s = T->zinv;
    PassFuse post = no_fuse();
    post.post = 3; post.phi = T->gi_hi_canon.as<Fr>(); post.plo = T->gi_lo.as<Fr>(); post.lo_bits = T->lo_bits;
    post.out = d_h_out;
    if (log_m == 0) return fail(ctx, ZG_E_INVALID_ARG, "H needs a domain of at least 2 rows");
    ZG_TRY(run_passes<true>(ctx, d_a, T->tw_inv.as<FrLimbs>(), log_m, pre_m, post));
    return ZG_OK;
}

}  // namespace zg
