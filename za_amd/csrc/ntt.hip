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
#include <vector>

#include "common.h"

namespace zg {

static constexpr int NTT_KMAX = 8;        // butterfly stages fused per pass
static constexpr int NTT_LOG_TILE = 11;   // 2^11 elements * 32 B = 64 KiB of LDS per workgroup
static constexpr int GEOM_LO = 10;        // low-table bits for the g^i two-level power tables

struct NttTables {
    uint32_t log_m = 0;
    DevBuf tw_fwd, tw_inv;              // omega^i, omega^-i, i < m/2   (Montgomery)
    DevBuf g_lo, g_hi;                  // g^j (j < 2^lo), g^(k 2^lo)                    (Montgomery)
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
    A(T->tw_fwd, half); A(T->tw_inv, half);
    A(T->g_lo, nlo); A(T->g_hi, nhi); A(T->gi_lo, nlo); A(T->gi_hi, nhi); A(T->gi_hi_canon, nhi);
    if (rc != ZG_OK) { delete T; return rc; }
    rc = launch_powers(ctx, T->tw_fwd.as<Fr>(), half, omega, Fr::one());
    if (!rc) rc = launch_powers(ctx, T->tw_inv.as<Fr>(), half, omega_inv, Fr::one());
    if (!rc) rc = launch_powers(ctx, T->g_lo.as<Fr>(), nlo, g, Fr::one());
    if (!rc) rc = launch_powers(ctx, T->g_hi.as<Fr>(), nhi, g.pow_u64(nlo), Fr::one());
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
// One in-place pass of decimation-in-frequency stages s_hi, s_hi-1, ..., s_hi-k+1 over data[0..2^log_m).
// A workgroup owns 2^(k+j) elements.  If s_lo = s_hi-k+1 >= j ("strided" form) they are 2^k rows (the
// k butterfly bits) of 2^j adjacent columns, so global accesses are 2^j*32 B contiguous runs; otherwise
// (s_lo == 0, "contiguous" form) the tile is 2^(k+j) consecutive elements.
__global__ void __launch_bounds__(256) k_ntt_dif_pass(Fr* __restrict__ data, const Fr* __restrict__ tw, int log_m,
                                                      int s_hi, int k, int j, int contiguous) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* lds = reinterpret_cast<uint32_t*>(smem);   // 9 limbs per element
    const int s_lo = s_hi - k + 1;
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
    for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) lds_put(li, FrD::load_red(&data[gidx(li)]));
    __syncthreads();
    const int sh = contiguous ? 0 : j;  // position of the butterfly bits inside the local index
    const uint32_t half = tile_elems >> 1;
    for (int q = k - 1; q >= 0; q--) {
        const int s = s_lo + q;          // global stage: butterfly span 2^s
        const int bit = sh + q;
        for (uint32_t b = threadIdx.x; b < half; b += blockDim.x) {
            uint32_t i0 = ((b >> bit) << (bit + 1)) | (b & ((1u << bit) - 1u));
            uint32_t i1 = i0 | (1u << bit);
            size_t e0 = gidx(i0);
            size_t texp = (e0 & (((size_t)1 << s) - 1)) << (log_m - 1 - s);
            FrD u = lds_get(i0), v = lds_get(i1);   // both < 2.01 p
            lds_put(i0, (u + v).reduce());
            lds_put(i1, FrD::sub<3>(u, v) * FrD::load_canon(&tw[texp]));
        }
        __syncthreads();
    }
    for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) lds_get(li).reduce().store(&data[gidx(li)]);
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

// natural order in -> bit-reversed order out
static int dif_passes(Ctx* ctx, Fr* d, const Fr* tw, uint32_t log_m) {
    static bool lds_attr_set = false;
    if (!lds_attr_set) {
        ZG_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_dif_pass, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        36 << NTT_LOG_TILE));
        lds_attr_set = true;
    }
    int remaining = (int)log_m, s_hi = (int)log_m - 1;
    while (remaining > 0) {
        int np = (remaining + NTT_KMAX - 1) / NTT_KMAX;
        int k = (remaining + np - 1) / np;
        int s_lo = s_hi - k + 1;
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
        size_t tiles = ((size_t)1 << log_m) >> (k + j);
        size_t lds_bytes = (size_t)36 << (k + j);   // 9 u32 limbs per element
        ProfScope ps(ctx, PROF_NTT_PASS, (uint64_t)64 << log_m, nullptr, ((uint64_t)k << log_m) >> 1);  // units: butterflies
        hipLaunchKernelGGL(k_ntt_dif_pass, dim3((unsigned)tiles), dim3(256), lds_bytes, ctx->stream, d, tw, (int)log_m,
                           s_hi, k, j, contiguous);
        ZG_HIP(ctx, hipGetLastError());
        s_hi -= k;
        remaining -= k;
    }
    return ZG_OK;
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
        ZG_TRY(dif_passes(ctx, d, inverse ? T->tw_inv.as<Fr>() : T->tw_fwd.as<Fr>(), log_m));
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

// a, b, c in the device form (M261, reduce()d); d_h_out receives plain integers
int h_poly_dev(Ctx* ctx, Fr* d_a, Fr* d_b, Fr* d_c, uint32_t log_m, Fr* d_h_out) {
    NttTables* T;
    ZG_TRY(get_tables(ctx, log_m, &T));
    const size_t m = (size_t)1 << log_m;
    hipStream_t st = ctx->stream;
    Fr* v[3] = {d_a, d_b, d_c};
    for (int k = 0; k < 3; k++) {
        ZG_TRY(ntt_core(ctx, T, v[k], /*inverse=*/true, /*coset=*/false));
        ZG_TRY(ntt_core(ctx, T, v[k], /*inverse=*/false, /*coset=*/true));
    }
    hipLaunchKernelGGL(k_mul_sub_scale, dim3(blocks_for(m)), dim3(256), 0, st, d_a, d_b, d_c, m, T->zinv);
    // icoset_fft with the final g^-i/m scaling fused with the Montgomery -> canonical conversion
    if (log_m > 0) {
        ZG_TRY(dif_passes(ctx, d_a, T->tw_inv.as<Fr>(), log_m));
        hipLaunchKernelGGL(k_bitrev, dim3(blocks_for(m)), dim3(256), 0, st, d_a, (int)log_m);
    }
    hipLaunchKernelGGL(k_scale_geom_to_repr, dim3(blocks_for(m)), dim3(256), 0, st, d_a, d_h_out, m,
                       T->gi_hi_canon.as<Fr>(), T->gi_lo.as<Fr>(), T->lo_bits);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

}  // namespace zg
