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
#include <algorithm>
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
    Fr zinv_int;                        // 1/(g^m - 1) as a plain integer (k_ntt_tail3 / k_h_quotient leave Montgomery form with it)
    Fr cfold_int;                       // -1 / (m (g^m - 1)) as a plain integer: the scale of the folded C bases (heval.hip)
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
    ZG_LAUNCH(ctx, k_powers, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, out, n,
                       host_to_m261(base), host_to_m261(scale), run);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

static int launch_powers_limbs(Ctx* ctx, FrLimbs* out, size_t n, const Fr& base) {
    uint32_t run = 64;
    size_t threads = (n + run - 1) / run;
    ZG_LAUNCH(ctx, k_powers_limbs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, out, n,
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
    T->zinv_int = (g.pow_u64(m) - Fr::one()).inv().from_mont();
    T->cfold_int = (Fr::zero() - (g.pow_u64(m) - Fr::one()).inv() * T->minv_host).from_mont();
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
        ZG_LAUNCH(ctx, k_from_mont, dim3((unsigned)((nhi + 255) / 256)), dim3(256), 0, ctx->stream,
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

// the LDS tile: 9 limbs per element (36-byte stride: conflict-free for unit-stride rounds)
struct LdsTile {
    uint32_t* lds;
    __device__ __forceinline__ FrD get(uint32_t li) const {
        FrD v;
#pragma unroll
        for (int q = 0; q < 9; q++) v.l[q] = lds[9 * li + q];
        return v;
    }
    __device__ __forceinline__ void put(uint32_t li, const FrD& v) const {
#pragma unroll
        for (int q = 0; q < 9; q++) lds[9 * li + q] = v.l[q];
    }
};
struct BlockSync {
    __device__ __forceinline__ void operator()() const { __syncthreads(); }
};

// Which of a launch's vectors is c, and the device word that says whether THIS proof transforms it: the folded form (C rows
// as scalars of the l multiexp, prover.hip ensure_c_fold) is decided per proof ON THE DEVICE from the digits its rows
// evaluate to (k_fold_decide), after the launches are queued -- so c's passes are always queued and leave at once when the
// word says "folded", and the last kernel takes the two- or three-vector form by it.  flag == NULL: no guard.
struct SkipC {
    const uint32_t* flag = nullptr;   // *flag != 0: the C rows are folded, c is not transformed
    int y = -1;                       // blockIdx.y of c in this launch (-1: c is not in it)
    __device__ __forceinline__ bool drops(uint32_t by) const { return flag && (int)by == y && *flag != 0u; }
};

// (data1, data2: two more vectors for the workgroups with blockIdx.y = 1, 2 -- the a, b, c of a small proof's H pipeline as ONE
// launch per pass: their passes are chains of short launches, and three vectors one after the other make the chain three
// times as long)
template <bool DIT>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_pass(Fr* __restrict__ data0, const FrLimbs* __restrict__ tw, int log_m, int s_lo,
                                                  int k, int j, int contiguous, PassFuse fz, Fr* __restrict__ data1,
                                                  Fr* __restrict__ data2, SkipC skip) {
    if (skip.drops(blockIdx.y)) return;   // (c's transforms of a proof whose C rows went into the l multiexp: SkipC)
    Fr* __restrict__ data = blockIdx.y == 0 ? data0 : blockIdx.y == 1 ? data1 : data2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsTile T{reinterpret_cast<uint32_t*>(smem)};
    const TileGeom G = TileGeom::make(log_m, s_lo, k, j, contiguous, blockIdx.x);
    const uint32_t tile_elems = G.elems();
    const size_t lomask = ((size_t)1 << fz.lo_bits) - 1;
    auto ingest = [&](uint32_t li, size_t g, const Fr& raw) {
        FrD x = FrD::load_red(&raw);
        if (fz.pre == 1) {
            x = x * (FrD::load_canon(&fz.hi[g >> fz.lo_bits]) * FrD::load_canon(&fz.lo[g & lomask]));
        } else if (fz.pre == 2) {
            FrD ab = x * FrD::load_red(&fz.b[g]);
            x = FrD::sub<3>(ab, FrD::load_red(&fz.c[g])) * FrD::load_canon(&fz.s);
        }
        T.put(li, x);
    };
    if (tile_elems == 4u * NTT_THREADS) {
        // the usual case: four elements per thread.  All four 32-byte loads are issued before the first is used, so the
        // tile's load phase costs one HBM latency instead of four.
        Fr raw[4];
        size_t g4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            g4[q] = G.gidx(threadIdx.x + q * NTT_THREADS);
            raw[q] = data[g4[q]];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) ingest(threadIdx.x + q * NTT_THREADS, g4[q], raw[q]);
    } else {
        for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) {
            const size_t g = G.gidx(li);
            ingest(li, g, data[g]);
        }
    }
    __syncthreads();
    ntt_tile_stages<DIT>(T, G, tw, threadIdx.x, blockDim.x, BlockSync{});
    for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) {
        const size_t g = G.gidx(li);
        FrD x = T.get(li);
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

// ---- fused passes of the evaluation-form H pipeline (h_eval_dev)
// The LAST pass of an inverse transform (decimation in time, top stages) and the FIRST pass of the following coset
// transform (decimation in frequency, the same top stages) work on the same strided tile: one kernel runs both with
// the coset scaling g^i / m applied in LDS in between -- one load and one store where two passes made two of each.
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_mid(Fr* __restrict__ data0, const FrLimbs* __restrict__ tw_inv,
                                                         const FrLimbs* __restrict__ tw_fwd, int log_m, int s_lo, int k, int j,
                                                         int contiguous, const Fr* __restrict__ hi, const Fr* __restrict__ lo,
                                                         int lo_bits, Fr* __restrict__ data1, Fr* __restrict__ data2, SkipC skip) {
    if (skip.drops(blockIdx.y)) return;
    Fr* __restrict__ data = blockIdx.y == 0 ? data0 : blockIdx.y == 1 ? data1 : data2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsTile T{reinterpret_cast<uint32_t*>(smem)};
    const TileGeom G = TileGeom::make(log_m, s_lo, k, j, contiguous, blockIdx.x);
    const uint32_t tile_elems = G.elems();
    const size_t lomask = ((size_t)1 << lo_bits) - 1;
    if (tile_elems == 4u * NTT_THREADS) {
        Fr raw[4];
#pragma unroll
        for (int q = 0; q < 4; q++) raw[q] = data[G.gidx(threadIdx.x + q * NTT_THREADS)];
#pragma unroll
        for (int q = 0; q < 4; q++) T.put(threadIdx.x + q * NTT_THREADS, FrD::load_red(&raw[q]));
    } else {
        for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) T.put(li, FrD::load_red(&data[G.gidx(li)]));
    }
    __syncthreads();
    ntt_tile_stages<true>(T, G, tw_inv, threadIdx.x, blockDim.x, BlockSync{});   // bounds: 2.01 + 2 k
    for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) {
        const size_t g = G.gidx(li);
        const FrD sc = FrD::load_canon(&hi[g >> lo_bits]) * FrD::load_canon(&lo[g & lomask]);
        T.put(li, T.get(li) * sc);   // a product: below 1.2 p whatever k was
    }
    __syncthreads();
    ntt_tile_stages<false>(T, G, tw_fwd, threadIdx.x, blockDim.x, BlockSync{});
    for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) T.get(li).reduce().store(&data[G.gidx(li)]);
}

// The LAST pass of the three coset transforms (contiguous tiles, low stages, bit-reversed out) and the quotient
//     e = (a b - c) / Z(g)     as a plain integer (the h multiexp's scalar)
// in one kernel: the tiles of c, b, a go through the same LDS buffer one after the other, every thread keeps its
// four results of c and b in registers (packed, 2 x 32) until a's are there.  full_k > 0 (domains of one tile, 2^10 and
// below): the whole H pipeline of a tile -- inverse transform (full_k DIT stages), coset scaling, coset transform -- runs
// in front, so such a proof's H is ONE launch.
// A thread parks its results of c and b in its own elements of those vectors (scratch by then; the same thread reads
// them back, so no barrier) -- holding them in registers instead was tried: 174 VGPRs, two workgroups per CU.
__global__ void __launch_bounds__(NTT_THREADS, 4) k_ntt_tail3(const Fr* __restrict__ a, Fr* __restrict__ b,
                                                           Fr* __restrict__ c_in, const FrLimbs* __restrict__ tw_fwd,
                                                           const FrLimbs* __restrict__ tw_inv, int log_m, int k, int j, int full,
                                                           const Fr* __restrict__ hi, const Fr* __restrict__ lo, int lo_bits,
                                                           Fr zinv_int, Fr* __restrict__ out, uint32_t tile0, const uint32_t* skip_c) {
    Fr* __restrict__ c = skip_c && *skip_c != 0u ? nullptr : c_in;   // (SkipC: this proof's C rows are in its l multiexp)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsTile T{reinterpret_cast<uint32_t*>(smem)};
    const TileGeom G = TileGeom::make(log_m, 0, k, j, 1, (size_t)blockIdx.x + tile0);
    const uint32_t tile_elems = G.elems();
    const size_t lomask = ((size_t)1 << lo_bits) - 1;
#pragma unroll 1
    for (int v = c ? 0 : 1; v < 3; v++) {   // (c == NULL: the C term is folded into the l multiexp, see h_eval_dev)
        const Fr* d = v == 0 ? c : v == 1 ? b : a;
        if (tile_elems == 4u * NTT_THREADS) {
            Fr raw[4];
#pragma unroll
            for (int q = 0; q < 4; q++) raw[q] = d[G.hbase + threadIdx.x + q * NTT_THREADS];
#pragma unroll
            for (int q = 0; q < 4; q++) T.put(threadIdx.x + q * NTT_THREADS, FrD::load_red(&raw[q]));
        } else {
            for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) T.put(li, FrD::load_red(&d[G.hbase + li]));
        }
        __syncthreads();
        if (full) {
            ntt_tile_stages<true>(T, G, tw_inv, threadIdx.x, blockDim.x, BlockSync{});
            for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) {
                const size_t g = G.hbase + li;
                const FrD sc = FrD::load_canon(&hi[g >> lo_bits]) * FrD::load_canon(&lo[g & lomask]);
                T.put(li, T.get(li) * sc);
            }
            __syncthreads();
        }
        ntt_tile_stages<false>(T, G, tw_fwd, threadIdx.x, blockDim.x, BlockSync{});
        if (v < 2) {
            Fr* parkv = v == 0 ? c : b;
            for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) T.get(li).reduce().store(&parkv[G.hbase + li]);
            __syncthreads();   // the next vector's loads overwrite the tile
        }
    }
    const FrD z = FrD::load_canon(&zinv_int);
    for (uint32_t li = threadIdx.x; li < tile_elems; li += blockDim.x) {
        const size_t g = G.hbase + li;
        const FrD ab = T.get(li).reduce() * FrD::load_red(&b[g]);
        // an M261 residue times a plain integer is the plain product: (a b - c) / Z leaves Montgomery form here
        if (c)
            (FrD::sub<3>(ab, FrD::load_red(&c[g])) * z).canonical().store(&out[g]);
        else
            (ab * z).canonical().store(&out[g]);
    }
}

// e = (a b - c) / Z(g) as a plain integer, element-wise (the unfused form of k_ntt_tail3's last step)
__global__ void __launch_bounds__(256) k_h_quotient(const Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c,
                                                    size_t n, Fr zinv_int, Fr* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FrD ab = FrD::load_red(&a[i]) * FrD::load_red(&b[i]);
    (FrD::sub<3>(ab, FrD::load_red(&c[i])) * FrD::load_canon(&zinv_int)).canonical().store(&out[i]);
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

// The transform kernels' dynamic-LDS limit is a per-device function attribute: set once per context (a host may open
// several devices; a process-wide flag would leave the second device without it).
static int ntt_kernel_attributes(Ctx* ctx) {
    if (ctx->ntt_attr_set) return ZG_OK;
    const int bytes = 36 << NTT_LOG_TILE;
    ZG_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_pass<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    ZG_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_pass<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    ZG_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_mid, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    ZG_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_tail3, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    ctx->ntt_attr_set = true;
    return ZG_OK;
}

// All passes of one transform.  DIF: natural in -> bit-reversed out (stages high -> low); DIT: bit-reversed in ->
// natural out (stages low -> high).  `first` / `last` carry the element-wise work fused into the first pass's
// loads / the last pass's stores.
template <bool DIT>
static int run_passes(Ctx* ctx, Fr* d, const FrLimbs* tw, uint32_t log_m, const PassFuse& first, const PassFuse& last) {
    ZG_TRY(ntt_kernel_attributes(ctx));
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
        ZG_LAUNCH(ctx, (k_ntt_pass<DIT>), dim3((unsigned)tiles), dim3(NTT_THREADS), lds_bytes, ctx->stream, d, tw, (int)log_m,
                           s_lo, k, j, contiguous, fz, (Fr*)nullptr, (Fr*)nullptr, SkipC{});
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
        ZG_LAUNCH(ctx, k_scale_geom, dim3(blocks_for(m)), dim3(256), 0, st, d, m, T->g_hi.as<Fr>(), T->g_lo.as<Fr>(),
                           T->lo_bits);
    }
    if (log_m > 0) {
        ZG_TRY(dif_passes(ctx, d, inverse ? T->tw_inv.as<FrLimbs>() : T->tw_fwd.as<FrLimbs>(), log_m));
        ZG_LAUNCH(ctx, k_bitrev, dim3(blocks_for(m)), dim3(256), 0, st, d, (int)log_m);
    }
    if (inverse) {
        if (coset)
            ZG_LAUNCH(ctx, k_scale_geom, dim3(blocks_for(m)), dim3(256), 0, st, d, m, T->gi_hi.as<Fr>(),
                               T->gi_lo.as<Fr>(), T->lo_bits);
        else
            ZG_LAUNCH(ctx, k_scale_const, dim3(blocks_for(m)), dim3(256), 0, st, d, m, T->minv);
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
    ZG_LAUNCH(ctx, k_from_mont, dim3(blocks_for(m)), dim3(256), 0, ctx->stream, d_out, m);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}
// out[i] = integer(scale * base^i), i < n
int powers_int_dev(Ctx* ctx, Fr* d_out, size_t n, const Fr& base_host, const Fr& scale_host) {
    if (n == 0) return ZG_OK;
    ZG_TRY(launch_powers(ctx, d_out, n, base_host, scale_host));
    ZG_LAUNCH(ctx, k_from_mont, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, d_out, n);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

int bitrev_dev(Ctx* ctx, Fr* d_data, uint32_t log_m) {
    if (log_m == 0) return ZG_OK;
    ZG_LAUNCH(ctx, k_bitrev, dim3(blocks_for((size_t)1 << log_m)), dim3(256), 0, ctx->stream, d_data, (int)log_m);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

int convert_mont_dev(Ctx* ctx, void* d_elems, size_t n, int which, int dir, hipStream_t st) {
    if (n == 0) return ZG_OK;
    if (!st) st = ctx->stream;
    if (which)
        ZG_LAUNCH(ctx, (k_convert<Fr29>), dim3(blocks_for(n)), dim3(256), 0, st, (Fr*)d_elems, n, dir);
    else
        ZG_LAUNCH(ctx, (k_convert<Fq29>), dim3(blocks_for(n)), dim3(256), 0, st, (Fq*)d_elems, n, dir);
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

// ---- H in EVALUATION form ---------------------------------------------------------------------------------------
// bellman ends with icoset_fft and multiplies the COEFFICIENTS h_i into the h query.  The multiexp is linear, so the
// last transform can be moved into the bases once per key (heval.hip: h_eval_i = sum_j M_ij H_j, a DFT over the group):
//     sum_i h_i H_i  =  sum_p e[p] H_eval[p],      e[p] = ((A B - C) / Z)(g omega^bitrev(p))
// and a proof needs six transforms instead of seven, with no pass over H's coefficients at all.  Per vector:
//     DIT passes 0 .. L-2 | k_ntt_mid (last DIT pass + coset scaling + first DIF pass) | DIF passes L-2 .. 1
// then ONE k_ntt_tail3 for the last DIF pass of a, b and c with the quotient in LDS: 4 L - 3 launches (13 at 2^22) where
// the coefficient form takes 7 L (21).  Domains of one tile (2^10 and below): one launch.
// The stage partition is listed low -> high; the top pass is the largest (k_ntt_mid runs it twice per load).
struct StagePlan {
    int n;
    int k[8], s_lo[8];
};
static StagePlan stage_plan(uint32_t log_m) {
    StagePlan P{};
    P.n = ((int)log_m + NTT_KMAX - 1) / NTT_KMAX;
    if (P.n < 1) P.n = 1;
    int remaining = (int)log_m;
    for (int p = P.n; p > 0; p--) {   // from the top: ceil first
        const int k = (remaining + p - 1) / p;
        P.k[p - 1] = k;
        remaining -= k;
    }
    int s = 0;
    for (int p = 0; p < P.n; p++) {
        P.s_lo[p] = s;
        s += P.k[p];
    }
    return P;
}
static void pass_geometry(uint32_t log_m, int s_lo, int k, int* j, int* contiguous) {
    if (s_lo == 0) {
        *contiguous = 1;
        *j = NTT_LOG_TILE - k;
        if (k + *j > (int)log_m) *j = (int)log_m - k;
    } else {
        *contiguous = 0;
        *j = NTT_LOG_TILE - k;
        if (*j > s_lo) *j = s_lo;
    }
    if (*j < 0) *j = 0;
}
template <bool DIT>
static int run_one_pass(Ctx* ctx, Fr* d, const FrLimbs* tw, uint32_t log_m, int s_lo, int k, Fr* d1 = nullptr, Fr* d2 = nullptr,
                        SkipC skip = SkipC{}) {
    int j, contiguous;
    pass_geometry(log_m, s_lo, k, &j, &contiguous);
    const size_t tiles = ((size_t)1 << log_m) >> (k + j);
    const unsigned nv = d2 ? 3u : d1 ? 2u : 1u;   // (one vector, or a, b (, c) together)
    ProfScope ps(ctx, PROF_NTT_PASS, (uint64_t)nv * 64 << log_m, nullptr, ((uint64_t)nv * k << log_m) >> 1);
    ZG_LAUNCH(ctx, (k_ntt_pass<DIT>), dim3((unsigned)tiles, nv), dim3(NTT_THREADS), (size_t)36 << (k + j), ctx->stream, d, tw,
                       (int)log_m, s_lo, k, j, contiguous, no_fuse(), d1, d2, skip);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

// a, b, c: device form (M261, reduce()d), row i at index bitrev(i), zero padded; clobbered.  d_e_out: m plain integers,
// e[p] = the quotient's value at g * omega^bitrev(p) -- the scalars that go with the key's h_eval bases.
// d_c == NULL ("C folded", prover.hip ensure_c_fold): FOUR transforms.  The quotient is linear in c,
//     sum_p e[p] H_eval[p] = sum_p (a b / Z)[p] H_eval[p]  -  (1 / Z) sum_p c(g omega^bitrev(p)) H_eval[p],
// and the second sum is sum_j c_j [ (1/m) sum_i omega^(-i j) H_i ] over c's VALUES ON THE DOMAIN (the R1CS rows as they
// are evaluated: no transform at all) against a second per-key transformed basis (heval.hip, c_fold) -- so e = a b / Z
// here and the caller adds the multiexp of the rows' c values, which for a circuit of bit signals are bits themselves.
// nsplit > 0: the last kernel runs in nsplit launches over consecutive ranges of e -- range k covers the positions
// below min(m, (k + 1) * ceil(m / nsplit)) -- with ev_split[k] recorded behind launch k: part k of the h multiexp starts
// its digit sort while the rest of e is still being computed.
// d_skip_c (with d_c): a device word written before these launches run (prover.hip k_fold_decide): non-zero = this proof's
// C rows are folded -- c's passes leave at once and the last kernel takes the two-vector form; zero = six transforms.
int h_eval_dev(Ctx* ctx, Fr* d_a, Fr* d_b, Fr* d_c, uint32_t log_m, Fr* d_e_out, int nsplit, const hipEvent_t* ev_split,
               const uint32_t* d_skip_c) {
    if (!d_c) d_skip_c = nullptr;
    NttTables* T;
    ZG_TRY(get_tables(ctx, log_m, &T));
    if (log_m == 0) return fail(ctx, ZG_E_INVALID_ARG, "H needs a domain of at least 2 rows");
    ZG_TRY(ntt_kernel_attributes(ctx));
    const FrLimbs *twi = T->tw_inv.as<FrLimbs>(), *twf = T->tw_fwd.as<FrLimbs>();
    const Fr *hi = T->g_hi_minv.as<Fr>(), *lo = T->g_lo.as<Fr>();
    const size_t m = (size_t)1 << log_m;
    if (log_m <= (uint32_t)NTT_LOG_TILE) {   // the whole domain is one tile
        const uint64_t nt = d_c ? 6 : 4;
        ProfScope ps(ctx, PROF_NTT_PASS, nt * 64 << log_m, nullptr, (nt * log_m << log_m) >> 1);
        ZG_LAUNCH(ctx, k_ntt_tail3, dim3(1), dim3(NTT_THREADS), (size_t)36 << log_m, ctx->stream, d_a, d_b, d_c, twf, twi,
                           (int)log_m, (int)log_m, 0, 1, hi, lo, T->lo_bits, T->zinv_int, d_e_out, 0u, d_skip_c);
        for (int k = 0; k < nsplit; k++) ZG_HIP(ctx, hipEventRecord(ev_split[k], ctx->stream));
        ZG_HIP(ctx, hipGetLastError());
        return ZG_OK;
    }
    const StagePlan P = stage_plan(log_m);
    const int L = P.n;   // >= 2 here
    Fr* v[3] = {d_a, d_b, d_c};
    // Up to 2^opt_ntt_batch_log_m rows (default 2^21) the passes of a, b and c are ONE launch each (blockIdx.y picks the
    // vector): the pipeline is a chain of launches on the proof's critical path, sharing the command processor with the
    // witness multiexps' chains -- 7 launches instead of 15 at 2^14, 9 instead of 21 at 2^20 (a 2^18 proof 1.69 -> 1.40 ms, 2^20
    // 3.64 -> 3.40).  At 2^22 a pass fills the GPU for 0.17 ms on its own and nothing is gained.
    const bool together = (int)log_m <= ctx->opt_ntt_batch_log_m;
    const int n_vec = d_c ? 3 : 2;
    for (int q = 0; q < (together ? 1 : n_vec); q++) {
        Fr *v1 = together ? v[1] : nullptr, *v2 = together ? v[2] : nullptr;   // (v[2] == NULL: a and b only)
        const unsigned nv = together ? (unsigned)n_vec : 1u;
        SkipC skip;   // where c is in these launches: blockIdx.y = 2 of a joint launch, 0 of its own
        if (d_skip_c && (together || q == 2)) {
            skip.flag = d_skip_c;
            skip.y = together ? 2 : 0;
        }
        for (int p = 0; p + 1 < L; p++) ZG_TRY(run_one_pass<true>(ctx, v[q], twi, log_m, P.s_lo[p], P.k[p], v1, v2, skip));
        {
            int j, contiguous;
            pass_geometry(log_m, P.s_lo[L - 1], P.k[L - 1], &j, &contiguous);
            const size_t tiles = m >> (P.k[L - 1] + j);
            ProfScope ps(ctx, PROF_NTT_PASS, (uint64_t)nv * 64 << log_m, nullptr, ((uint64_t)nv * 2 * P.k[L - 1] << log_m) >> 1);
            ZG_LAUNCH(ctx, k_ntt_mid, dim3((unsigned)tiles, nv), dim3(NTT_THREADS), (size_t)36 << (P.k[L - 1] + j), ctx->stream,
                               v[q], twi, twf, (int)log_m, P.s_lo[L - 1], P.k[L - 1], j, contiguous, hi, lo, T->lo_bits, v1, v2, skip);
            ZG_HIP(ctx, hipGetLastError());
        }
        for (int p = L - 2; p >= 1; p--) ZG_TRY(run_one_pass<false>(ctx, v[q], twf, log_m, P.s_lo[p], P.k[p], v1, v2, skip));
    }
    {
        int j, contiguous;
        pass_geometry(log_m, 0, P.k[0], &j, &contiguous);
        const size_t tiles = m >> (P.k[0] + j);
        ProfScope ps(ctx, PROF_NTT_PASS, (uint64_t)(n_vec + 1) * 32 << log_m, nullptr, ((uint64_t)n_vec * P.k[0] << log_m) >> 1);
        const int ns = nsplit > 0 ? nsplit : 1;
        const size_t chunk = (m + ns - 1) / ns, tile_elems = (size_t)1 << (P.k[0] + j);
        size_t t0 = 0;
        for (int q = 0; q < ns; q++) {
            const size_t upto = std::min(m, (size_t)(q + 1) * chunk);
            const size_t t1 = q + 1 == ns ? tiles : std::min(tiles, (upto + tile_elems - 1) / tile_elems);
            if (t1 > t0)
                ZG_LAUNCH(ctx, k_ntt_tail3, dim3((unsigned)(t1 - t0)), dim3(NTT_THREADS), (size_t)36 << (P.k[0] + j), ctx->stream,
                                   d_a, d_b, d_c, twf, twi, (int)log_m, P.k[0], j, 0, hi, lo, T->lo_bits, T->zinv_int, d_e_out,
                                   (uint32_t)t0, d_skip_c);
            if (nsplit > 0) ZG_HIP(ctx, hipEventRecord(ev_split[q], ctx->stream));
            t0 = std::max(t0, t1);
        }
        ZG_HIP(ctx, hipGetLastError());
    }
    return ZG_OK;
}

// What heval.hip needs of the tables: omega^-i in limb form (i < m/2) and the two-level tables of g^-i / m
int ntt_tables_for_heval(Ctx* ctx, uint32_t log_m, const FrLimbs** tw_inv, const Fr** gi_lo, const Fr** gi_hi_int, int* lo_bits,
                         Fr* cfold_int) {
    NttTables* T;
    ZG_TRY(get_tables(ctx, log_m, &T));
    if (cfold_int) *cfold_int = T->cfold_int;
    *tw_inv = T->tw_inv.as<FrLimbs>();
    *gi_lo = T->gi_lo.as<Fr>();
    *gi_hi_int = T->gi_hi_canon.as<Fr>();
    *lo_bits = T->lo_bits;
    return ZG_OK;
}

}  // namespace zg
