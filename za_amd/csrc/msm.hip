// Multi-scalar multiplication over BN254 G1 / G2 for gfx950.
//
// Replaces bellman_ce::multiexp::multiexp (the 8 calls inside create_proof, reference call site
// prover/src/groth16/prover.rs:173).  Bellman runs one CPU thread per c-bit window with 2^c-1
// Jacobian buckets; the result is a group element, so any bucket schedule gives the same point.
// GPU schedule (all kernels on ctx->stream):
//   1. k_count      signed c-bit digits of every scalar -> per-bucket counts (global atomics,
//                   wave-aggregated for the wave's leading key so 0/1-heavy witnesses do not
//                   serialise on one address -- bellman special-cases exponent 0 and 1 for the same reason)
//   2. scan         exclusive prefix sums: bucket offsets and "item" offsets (an item = up to T
//                   points of one bucket; heavy buckets are split so work per thread is bounded)
//   3. k_scatter    counting-sort scatter of (base index | sign) into bucket order
//   4. k_accumulate one thread per item: gather affine bases (64 B / 128 B, dwordx4 loads), XYZZ mixed adds
//   5. k_reduce_chunks  per window, weighted running-sum over runs of items:
//                   sum_i w_i P_i = sum_i (w_i - w_{i-1}) * (P_i + P_{i+1} + ...)
//   6. k_reduce_windows one workgroup per window, LDS tree over the chunk results
//   7. host: Horner over <= 64 window sums (c doublings each).
// Algorithmic HBM bytes: 64 B base + 32 B scalar per point (G1), 128 + 32 (G2).  The kernel is
// integer-ALU bound (about 10 field multiplications per point per window), not HBM bound.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <memory>

#include "common.h"

namespace zg {

static constexpr int MAXWIN = 128;

struct MsmScratch {
    DevBuf counts, offsets, item_off, cursor, sorted, item_key, partial, chunk_out, win_out, bsum, meta;
};

struct MsmPlan {
    int c, nwin;
    uint32_t B;      // buckets per window = 2^(c-1)
    uint32_t nkeys;  // nwin * B
    uint32_t T;      // max points per item
};

static MsmPlan plan_msm(size_t n, int elem_cost /*1 for G1, 3 for G2: same plan*/) {
    (void)elem_cost;
    MsmPlan best{};
    double best_cost = 1e300;
    for (int c = 2; c <= 22; c++) {
        int nwin = (255 + c - 1) / c;
        if (nwin > MAXWIN) continue;
        double B = (double)(1u << (c - 1));
        // stage-1 mixed adds + stage-2 full adds (2 per bucket, ~1.5x a mixed add) + fixed launch work
        double cost = (double)n * nwin + 3.0 * nwin * B + 2000.0 * nwin;
        // sorted-index footprint must stay below 2^32 entries
        if ((double)n * nwin >= 4.0e9) continue;
        if (cost < best_cost) {
            best_cost = cost;
            best.c = c;
            best.nwin = nwin;
            best.B = 1u << (c - 1);
        }
    }
    best.nkeys = (uint32_t)best.nwin * best.B;
    size_t mean = (n + best.B - 1) / best.B;
    size_t T = 2 * mean;
    if (T < 16) T = 16;
    if (T > 512) T = 512;
    best.T = (uint32_t)T;
    return best;
}

// ------------------------------------------------------------------------------------------------
struct Scalar256 {
    uint32_t w[8];
};

__device__ __forceinline__ Scalar256 load_scalar(const uint32_t* scalars, size_t idx) {
    const uint4* p = reinterpret_cast<const uint4*>(scalars + idx * 8);
    uint4 a = p[0], b = p[1];
    Scalar256 s;
    s.w[0] = a.x; s.w[1] = a.y; s.w[2] = a.z; s.w[3] = a.w;
    s.w[4] = b.x; s.w[5] = b.y; s.w[6] = b.z; s.w[7] = b.w;
    return s;
}

// Next signed digit of the scalar held in s (consumes c low bits).  Digits lie in
// [-2^(c-1)+1, 2^(c-1)]; nwin*c >= 255 guarantees the last carry is zero for scalars < 2^254.
__device__ __forceinline__ int32_t next_digit(Scalar256& s, int c, uint32_t& carry) {
    uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    uint32_t v = (s.w[0] & mask) + carry;
#pragma unroll
    for (int i = 0; i < 7; i++) s.w[i] = __funnelshift_r(s.w[i], s.w[i + 1], c);
    s.w[7] >>= c;
    if (v > half) {
        carry = 1;
        return (int32_t)v - (int32_t)(1u << c);
    }
    carry = 0;
    return (int32_t)v;
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// mode 0: histogram (counts[key] += 1); mode 1: scatter (sorted[cursor[key]++] = i | sign<<31)
template <int MODE>
__global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scalars,
                                                const uint32_t* __restrict__ index, size_t n, int c, int nwin,
                                                uint32_t* __restrict__ counts_or_cursor,
                                                uint32_t* __restrict__ sorted) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n_round = (n + 63) & ~(size_t)63;
    const uint32_t B = 1u << (c - 1);
    const uint32_t lane = lane_id();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool in = i < n;
        Scalar256 s;
#pragma unroll
        for (int k = 0; k < 8; k++) s.w[k] = 0;
        if (in) s = load_scalar(scalars, index ? (size_t)index[i] : i);
        uint32_t carry = 0;
        for (int w = 0; w < nwin; w++) {
            int32_t d = next_digit(s, c, carry);
            const bool valid = d != 0;  // lanes past n hold a zero scalar: never valid
            const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
            const uint32_t key = (uint32_t)w * B + mag - 1u;
            const uint32_t val = (uint32_t)i | (d < 0 ? 0x80000000u : 0u);
            const uint64_t vm = __ballot(valid);
            if (vm == 0) continue;  // wave-uniform
            // Aggregate the lanes that share the leading lane's key into one atomic.
            const int leader = __ffsll((unsigned long long)vm) - 1;
            const uint32_t k0 = __shfl(key, leader);
            const bool same = valid && key == k0;
            const uint64_t sm = __ballot(same);
            const uint32_t cnt = (uint32_t)__popcll(sm);
            if (MODE == 0) {
                if (same) {
                    if ((int)lane == leader) atomicAdd(&counts_or_cursor[k0], cnt);
                } else if (valid) {
                    atomicAdd(&counts_or_cursor[key], 1u);
                }
            } else {
                uint32_t base = 0;
                if (same && (int)lane == leader) base = atomicAdd(&counts_or_cursor[k0], cnt);
                base = __shfl(base, leader);
                if (same) {
                    sorted[base + mbcnt(sm)] = val;
                } else if (valid) {
                    uint32_t pos = atomicAdd(&counts_or_cursor[key], 1u);
                    sorted[pos] = val;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ scans
static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_PER_THREAD = 16;
static constexpr int SCAN_TILE = SCAN_THREADS * SCAN_PER_THREAD;

// exclusive scan of one value per thread across the block; returns the block total through `total`
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63) lds[wid] = x;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
    for (uint32_t k = 0; k < nw; k++) {
        uint32_t s = lds[k];
        if (k < wid) wave_off += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return wave_off + x - v;
}

__device__ __forceinline__ uint32_t items_of(uint32_t cnt, uint32_t T) { return (cnt + T - 1) / T; }

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_reduce(const uint32_t* __restrict__ counts, uint32_t nkeys,
                                                              uint32_t T, uint32_t* __restrict__ bsum /*[2][nb]*/,
                                                              uint32_t nb) {
    __shared__ uint32_t lds[8];
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t sc = 0, si = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        uint32_t idx = base + k;
        uint32_t cnt = idx < nkeys ? counts[idx] : 0;
        sc += cnt;
        si += items_of(cnt, T);
    }
    uint32_t tc, ti;
    block_excl_scan(sc, lds, tc);
    block_excl_scan(si, lds, ti);
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = tc;
        bsum[nb + blockIdx.x] = ti;
    }
}

// single block: in-place exclusive scan of both block-sum arrays
__global__ void __launch_bounds__(1024) k_scan_bsums(uint32_t* bsum, uint32_t nb) {
    __shared__ uint32_t lds[16];
    for (int arr = 0; arr < 2; arr++) {
        uint32_t* a = bsum + (size_t)arr * nb;
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nb; base += 1024) {
            uint32_t idx = base + threadIdx.x;
            uint32_t v = idx < nb ? a[idx] : 0;
            uint32_t tot;
            uint32_t ex = block_excl_scan(v, lds, tot);
            if (idx < nb) a[idx] = carry + ex;
            carry += tot;
        }
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_final(const uint32_t* __restrict__ counts, uint32_t nkeys,
                                                             uint32_t T, const uint32_t* __restrict__ bsum, uint32_t nb,
                                                             uint32_t* __restrict__ offsets,
                                                             uint32_t* __restrict__ item_off,
                                                             uint32_t* __restrict__ cursor) {
    __shared__ uint32_t lds[8];
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t cnt[SCAN_PER_THREAD];
    uint32_t sc = 0, si = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        uint32_t idx = base + k;
        cnt[k] = idx < nkeys ? counts[idx] : 0;
        sc += cnt[k];
        si += items_of(cnt[k], T);
    }
    uint32_t tc, ti;
    uint32_t ec = block_excl_scan(sc, lds, tc) + bsum[blockIdx.x];
    uint32_t ei = block_excl_scan(si, lds, ti) + bsum[nb + blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        uint32_t idx = base + k;
        if (idx <= nkeys) {  // entry nkeys holds the totals
            offsets[idx] = ec;
            item_off[idx] = ei;
            if (idx < nkeys) cursor[idx] = ec;
        }
        ec += cnt[k];
        ei += items_of(cnt[k], T);
    }
}

// meta[0] = total entries, meta[1] = total items, meta[2 + w] = item_off[w * B] for w = 0..nwin
__global__ void k_meta(const uint32_t* offsets, const uint32_t* item_off, uint32_t nkeys, uint32_t B, int nwin,
                       uint32_t* meta) {
    int t = threadIdx.x;
    if (t == 0) {
        meta[0] = offsets[nkeys];
        meta[1] = item_off[nkeys];
    }
    if (t <= nwin) meta[2 + t] = item_off[(size_t)t * B];
}

__global__ void __launch_bounds__(256) k_build_items(const uint32_t* __restrict__ item_off, uint32_t nkeys,
                                                     uint32_t* __restrict__ item_key) {
    uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= nkeys) return;
    uint32_t lo = item_off[key], hi = item_off[key + 1];
    for (uint32_t t = lo; t < hi; t++) item_key[t] = key;
}

// ------------------------------------------------------------------------------------------------ stage 1
template <class F>
__global__ void __launch_bounds__(256) k_accumulate(const Affine<F>* __restrict__ bases,
                                                    const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ offsets,
                                                    const uint32_t* __restrict__ item_off,
                                                    const uint32_t* __restrict__ item_key, uint32_t total_items,
                                                    uint32_t T, XYZZ<F>* __restrict__ partial) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_items) return;
    uint32_t key = item_key[t];
    uint32_t seg = t - item_off[key];
    uint32_t start = offsets[key] + seg * T;
    uint32_t end = offsets[key + 1];
    if (end > start + T) end = start + T;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t k = start; k < end; k++) {
        uint32_t e = sorted[k];
        Affine<F> p = bases[e & 0x7fffffffu];
        if (e >> 31) p.y = p.y.neg();
        acc = acc.add_mixed(p);
    }
    partial[t] = acc;
}

// ------------------------------------------------------------------------------------------------ stage 2
struct WinTable {
    uint32_t item_lo[MAXWIN + 1];   // first item of each window (and one past the end)
    uint32_t chunk_lo[MAXWIN + 1];  // first chunk (thread) of each window
    uint32_t per_chunk[MAXWIN];     // items per chunk
};

template <class F>
__global__ void __launch_bounds__(128) k_reduce_chunks(const XYZZ<F>* __restrict__ partial,
                                                       const uint32_t* __restrict__ item_key, WinTable tab, int nwin,
                                                       uint32_t B, XYZZ<F>* __restrict__ chunk_out) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= tab.chunk_lo[nwin]) return;
    int w = 0;
    while (w + 1 < nwin && g >= tab.chunk_lo[w + 1]) w++;
    uint32_t j = g - tab.chunk_lo[w];
    uint32_t L = tab.per_chunk[w];
    uint32_t lo = tab.item_lo[w] + j * L;
    uint32_t hi = tab.item_lo[w + 1];
    if (hi > lo + L) hi = lo + L;
    const uint32_t key0 = (uint32_t)w * B;
    XYZZ<F> running = XYZZ<F>::inf(), acc = XYZZ<F>::inf();
    for (uint32_t i = hi; i-- > lo;) {
        uint32_t wi = item_key[i] - key0 + 1u;
        uint32_t wprev = i > lo ? item_key[i - 1] - key0 + 1u : 0u;
        running = running.add(partial[i]);
        uint32_t gap = wi - wprev;
        if (gap == 1u)
            acc = acc.add(running);
        else if (gap > 1u)
            acc = acc.add(running.mul_small(gap));
    }
    chunk_out[g] = acc;
}

template <class F>
__global__ void __launch_bounds__(128) k_reduce_windows(const XYZZ<F>* __restrict__ chunk_out, WinTable tab,
                                                        XYZZ<F>* __restrict__ win_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    XYZZ<F>* lds = reinterpret_cast<XYZZ<F>*>(smem);
    const int w = blockIdx.x;
    const uint32_t lo = tab.chunk_lo[w], hi = tab.chunk_lo[w + 1];
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) acc = acc.add(chunk_out[i]);
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            acc = acc.add(lds[threadIdx.x + s]);
            lds[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) win_out[w] = acc;
}

// ------------------------------------------------------------------------------------------------ host driver
template <class F>
static int msm_run(Ctx* ctx, const Affine<F>* d_bases, const uint32_t* d_scalars, const uint32_t* d_index, size_t n,
                   XYZZ<F>* out) {
    *out = XYZZ<F>::inf();
    if (n == 0) return ZG_OK;
    if (n >= (1ull << 31)) return fail(ctx, ZG_E_INVALID_ARG, "msm: n = %zu exceeds 2^31-1", n);
    if (!ctx->msm_scratch) ctx->msm_scratch = new MsmScratch();
    MsmScratch& S = *ctx->msm_scratch;
    hipStream_t st = ctx->stream;
    const MsmPlan P = plan_msm(n, 1);
    const uint32_t nkeys = P.nkeys;

    ZG_HIP(ctx, S.counts.reserve((size_t)(nkeys + 1) * 4));
    ZG_HIP(ctx, S.offsets.reserve((size_t)(nkeys + 1) * 4));
    ZG_HIP(ctx, S.item_off.reserve((size_t)(nkeys + 1) * 4));
    ZG_HIP(ctx, S.cursor.reserve((size_t)(nkeys + 1) * 4));
    ZG_HIP(ctx, S.meta.reserve((size_t)(MAXWIN + 4) * 4));
    ZG_HIP(ctx, hipMemsetAsync(S.counts.p, 0, (size_t)(nkeys + 1) * 4, st));

    constexpr bool IS_G1 = sizeof(F) == sizeof(Fq);
    constexpr uint64_t ALG_BYTES = sizeof(Affine<F>) + 32;  // base + scalar, each read once
    int grid = (int)std::min<size_t>((n + 255) / 256, (size_t)ctx->num_cu * 8);
    std::unique_ptr<ProfScope> ps(new ProfScope(ctx, PROF_MSM_SORT, n * 32));
    hipLaunchKernelGGL(k_digits<0>, dim3(grid), dim3(256), 0, st, d_scalars, d_index, n, P.c, P.nwin,
                       S.counts.as<uint32_t>(), (uint32_t*)nullptr);

    // nkeys + 1 entries are scanned so that entry nkeys receives the totals
    const uint32_t nb = (nkeys + 1 + SCAN_TILE - 1) / SCAN_TILE;
    ZG_HIP(ctx, S.bsum.reserve((size_t)nb * 2 * 4));
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_THREADS), 0, st, S.counts.as<uint32_t>(), nkeys, P.T,
                       S.bsum.as<uint32_t>(), nb);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(1024), 0, st, S.bsum.as<uint32_t>(), nb);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(SCAN_THREADS), 0, st, S.counts.as<uint32_t>(), nkeys, P.T,
                       S.bsum.as<uint32_t>(), nb, S.offsets.as<uint32_t>(), S.item_off.as<uint32_t>(),
                       S.cursor.as<uint32_t>());
    hipLaunchKernelGGL(k_meta, dim3(1), dim3(MAXWIN + 1), 0, st, S.offsets.as<uint32_t>(), S.item_off.as<uint32_t>(),
                       nkeys, P.B, P.nwin, S.meta.as<uint32_t>());
    ps.reset();
    uint32_t meta[MAXWIN + 4];
    ZG_HIP(ctx, hipMemcpyAsync(meta, S.meta.p, (size_t)(P.nwin + 3) * 4, hipMemcpyDeviceToHost, st));
    ZG_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t total_entries = meta[0], total_items = meta[1];
    if (total_items == 0) return ZG_OK;  // all scalars zero

    ZG_HIP(ctx, S.sorted.reserve((size_t)total_entries * 4));
    ZG_HIP(ctx, S.item_key.reserve((size_t)total_items * 4));
    ZG_HIP(ctx, S.partial.reserve((size_t)total_items * sizeof(XYZZ<F>)));
    ps.reset(new ProfScope(ctx, PROF_MSM_SORT, n * 32));
    hipLaunchKernelGGL(k_digits<1>, dim3(grid), dim3(256), 0, st, d_scalars, d_index, n, P.c, P.nwin,
                       S.cursor.as<uint32_t>(), S.sorted.as<uint32_t>());
    hipLaunchKernelGGL(k_build_items, dim3((nkeys + 255) / 256), dim3(256), 0, st, S.item_off.as<uint32_t>(), nkeys,
                       S.item_key.as<uint32_t>());
    ps.reset(new ProfScope(ctx, IS_G1 ? PROF_MSM_ACC_G1 : PROF_MSM_ACC_G2, n * ALG_BYTES));
    hipLaunchKernelGGL((k_accumulate<F>), dim3((total_items + 255) / 256), dim3(256), 0, st, d_bases,
                       S.sorted.as<uint32_t>(), S.offsets.as<uint32_t>(), S.item_off.as<uint32_t>(),
                       S.item_key.as<uint32_t>(), total_items, P.T, S.partial.as<XYZZ<F>>());
    ps.reset();

    // stage 2 geometry: ~16 items per thread
    WinTable tab;
    memset(&tab, 0, sizeof(tab));
    uint32_t chunks = 0;
    for (int w = 0; w < P.nwin; w++) {
        tab.item_lo[w] = meta[2 + w];
        uint32_t items = meta[2 + w + 1] - meta[2 + w];
        uint32_t L = 16;
        uint32_t nch = (items + L - 1) / L;
        tab.per_chunk[w] = L;
        tab.chunk_lo[w] = chunks;
        chunks += nch;
    }
    tab.item_lo[P.nwin] = meta[2 + P.nwin];
    tab.chunk_lo[P.nwin] = chunks;
    ZG_HIP(ctx, S.chunk_out.reserve((size_t)chunks * sizeof(XYZZ<F>)));
    ZG_HIP(ctx, S.win_out.reserve((size_t)P.nwin * sizeof(XYZZ<F>)));
    ps.reset(new ProfScope(ctx, PROF_MSM_REDUCE, 0));
    hipLaunchKernelGGL((k_reduce_chunks<F>), dim3((chunks + 127) / 128), dim3(128), 0, st, S.partial.as<XYZZ<F>>(),
                       S.item_key.as<uint32_t>(), tab, P.nwin, P.B, S.chunk_out.as<XYZZ<F>>());
    hipLaunchKernelGGL((k_reduce_windows<F>), dim3(P.nwin), dim3(128), 128 * sizeof(XYZZ<F>), st,
                       S.chunk_out.as<XYZZ<F>>(), tab, S.win_out.as<XYZZ<F>>());
    ps.reset();
    std::vector<XYZZ<F>> win(P.nwin);
    ZG_HIP(ctx, hipMemcpyAsync(win.data(), S.win_out.p, (size_t)P.nwin * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
    ZG_HIP(ctx, hipStreamSynchronize(st));
    ZG_HIP(ctx, hipGetLastError());

    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = P.nwin - 1; w >= 0; w--) {
        for (int k = 0; k < P.c; k++) acc = acc.dbl();
        acc = acc.add(win[w]);
    }
    *out = acc;
    return ZG_OK;
}

int msm_g1(Ctx* ctx, const G1Affine* d_bases, const uint32_t* d_scalars, const uint32_t* d_index, size_t n,
           G1XYZZ* out) {
    return msm_run<Fq>(ctx, d_bases, d_scalars, d_index, n, out);
}
int msm_g2(Ctx* ctx, const G2Affine* d_bases, const uint32_t* d_scalars, const uint32_t* d_index, size_t n,
           G2XYZZ* out) {
    return msm_run<Fq2>(ctx, d_bases, d_scalars, d_index, n, out);
}
void msm_scratch_free(Ctx* ctx) {
    delete ctx->msm_scratch;
    ctx->msm_scratch = nullptr;
}

}  // namespace zg
