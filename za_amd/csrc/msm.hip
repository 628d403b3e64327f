// Multi-scalar multiplication over BN254 G1 / G2 for gfx950.
//
// Replaces bellman_ce::multiexp::multiexp (the 8 calls inside create_proof, reference call site
// prover/src/groth16/prover.rs:173).  Bellman runs one CPU thread per c-bit window with 2^c-1
// Jacobian buckets, skips exponent 0 and adds exponent-1 bases straight into the accumulator; the
// result is a group element, so any bucket schedule gives the same point.
// GPU schedule (one job = one stream; the h job of a proof runs as two dependent parts, see msm_batch5):
//   0. k_classify   counts scalars equal to 0, equal to 1, and the rest; the window size c is planned
//                   from the number of "real" scalars (a 97%-bit witness needs far fewer buckets)
//   1. k_part<0/1>, k_part_local   signed c-bit digits of every scalar, sorted by bucket with a two-level LDS
//                   partition (no global atomics); scalars == 1 go to one extra "ones" bucket, as bellman
//                   special-cases them.  With a fixed-base window table (plan_fbw_for) all digit positions of a
//                   scalar share ONE bucket set and the position selects the table row.
//   2. scans        exclusive prefix sums: bucket offsets and "item" offsets (an item = up to T points
//                   of one bucket, so work per thread is bounded for any scalar distribution)
//   3. k_item_*     items ordered by length (counting sort, longest first): the 64 lanes of a wave run equal loops
//   4. k_accumulate one thread per item: gather affine bases (dwordx4 loads), XYZZ mixed adds   [dominant]
//   5. k_heavy_fold, k_bits_level   bucket reduction by bit planes: per window S = sum of buckets and T_j = sum of
//                   the buckets with index bit j set; plain sums only, two lanes per group, two dependent adds per
//                   level; a level also folds the bit-plane rows of the levels before it
//   6. k_rowsum     sums the partial sums of the "ones" bucket
//   7. host: one Horner pass over the 254 bit positions (window value = S + sum_j 2^j T_j) + the ones sum, on a
//                   helper thread while other jobs are in flight.
// Bases live in HBM in the device field's M261 form (fd.h); field work uses the lazy 9x29-limb arithmetic.
// Algorithmic HBM bytes: 64 B base + 32 B scalar per point (G1), 128 + 32 (G2).  The kernel is
// integer-ALU bound (about 10 field multiplications per point per window), not HBM bound.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <memory>
#include <thread>
#include <array>
#include <atomic>
#include <functional>
#include <condition_variable>
#include <deque>
#include <future>

#include "common.h"

namespace zg {

static constexpr int MAXWIN = 128;
// jobs up to this many points are latency bound: queued whole, planned from n (MsmJob::eager_small), and the key gets
// window tables for all of its queries whatever the witness looks like (prover.hip ensure_tables)
// Two thresholds.  A DENSE job (the h query: every scalar a field element) stops being latency bound early -- 2^19
// points x 13 digits is 0.5 ms of additions at full rate, and the latency-oriented plan costs more of them (h shard of an
// 8-way 2^22 proof: 1.23 -> 1.97 ms when treated as small): 2^16.  A WITNESS job of a bit-heavy circuit has few real
// scalars and stays a chain of short launches much longer (same shard's witness phase: 3.4 -> 1.8 ms; a 2^18 proof 3.1 ->
// 2.6 ms; at 2^20 positions per query nothing is gained, at 2^21 it costs 0.8 ms): about 2^19.
// (ZG_OPT_SMALL_LOG_N / ZG_OPT_SMALL_W_LOG_N override; 0 turns the mode off.  Per context: tests switch them.)
size_t msm_small_n(const Ctx* ctx) { return ctx->opt_small_log_n <= 0 ? 0 : (size_t)1 << std::min(ctx->opt_small_log_n, 24); }
size_t msm_small_witness_n(const Ctx* ctx) {
    if (ctx->opt_small_log_n <= 0 && ctx->opt_small_w_log_n < 0) return 0;
    if (ctx->opt_small_w_log_n < 0) return 600000;   // "2^19 and a bit": a 1/8 shard of a 2^22 query is 2^19 + 8 positions
    return ctx->opt_small_w_log_n == 0 ? 0 : (size_t)1 << std::min(ctx->opt_small_w_log_n, 24);
}
int trace_level() {
    static const int lvl = getenv("ZA_AMD_TRACE") ? atoi(getenv("ZA_AMD_TRACE")) : 0;
    return lvl;
}
static constexpr int TREE_LAMBDA = 3;   // tree fan-in 2^3 = 8
static constexpr uint32_t T_ONES = 64;  // points per item of the ones bucket (upper limit; see item_sizes)

struct MsmScratch {
    DevBuf counts, offsets, item_off, tile_tab, abuf, sorted, item_key, perm, ihist, partial, lvl[2], ones[2], bsum, meta, heavy;
    DevBuf heavy_tmp;          // k_heavy_fold: the unit sums of buckets folded by several workgroups
    DevBuf abuf2, tab2, p2m;   // staged partition: level-2 output, its [segment][sub-bin][chunk] table, its plan
    static constexpr size_t HOST_BYTES = 256 * 1024;
    unsigned char* host = nullptr;  // pinned read-back buffer (class counts, totals, window sums)
};

struct MsmPlan {
    int c, nwin;     // digit width, digits per scalar
    int kw;          // windows in the key space: nwin, or 1 when a fixed-base window table folds them
    uint32_t B;      // buckets per window = 2^(c-1)
    uint32_t nkeys;  // kw * B
    uint32_t T;      // max points per item
};

// n_real: scalars that are neither 0 nor 1; n_ones: scalars equal to 1
static MsmPlan plan_msm(size_t n_real, size_t n_ones) {
    MsmPlan best{};
    double best_cost = 1e300;
    for (int c = 2; c <= 22; c++) {
        int nwin = (255 + c - 1) / c;
        if (nwin > MAXWIN) continue;
        double B = (double)(1u << (c - 1));
        // stage-1 mixed adds + tree reduction (~3.2 full adds per bucket, a full add ~1.5 mixed adds) + launches
        double cost = (double)n_real * nwin + 4.8 * nwin * B + 3000.0 * ((c - 1 + TREE_LAMBDA - 1) / TREE_LAMBDA);
        if ((double)n_real * nwin + (double)n_ones >= 4.0e9) continue;  // sorted-index footprint < 2^32 entries
        if (cost < best_cost) {
            best_cost = cost;
            best.c = c;
            best.nwin = nwin;
            best.B = 1u << (c - 1);
        }
    }
    best.kw = best.nwin;
    best.nkeys = (uint32_t)best.nwin * best.B;
    size_t mean = (n_real + best.B - 1) / best.B;
    size_t T = 2 * mean;
    if (T < 16) T = 16;
    if (T > 512) T = 512;
    best.T = (uint32_t)T;
    return best;
}

// Fixed-base window table (fbw): the table holds 2^(c*w) * P_i for every digit position w, so ALL digits of a
// scalar fall into ONE set of 2^(c-1) buckets.  The bucket reduction is paid once instead of once per window,
// which moves the optimum to a wider digit (c = 20 instead of 16 at n = 2^22: 13 instead of 16 mixed adds per
// scalar).  Costs nwin x the key's HBM footprint -- the trade 288 GB of HBM3E is there for.
// The last digit of a canonical scalar (< 2^254) has only t = 254 - c*(nwin-1) bits.  Folded into the shared
// bucket set it piles n entries onto the lowest 2^t buckets, and with a small t one k_part_local workgroup sorts
// all of them (measured: c = 19, t = 7 cost 1.26 ms of a 2^20 proof).  Large jobs only use plans with t >= 10.
// (Spreading the short digit over the bucket range with a rescaled last table row was tried: it balances the sort
// but mixes 360-point and 104-point buckets inside every wave of k_accumulate, +2 ms at 2^22.)
static int fbw_top_bits(int c, int nwin) { return 254 - c * (nwin - 1); }
static MsmPlan plan_fbw_for(int c, int nwin, size_t n) {
    MsmPlan p{};
    p.c = c;
    p.nwin = nwin;
    p.kw = 1;
    p.B = 1u << (c - 1);
    p.nkeys = p.B;
    size_t mean = (n * (size_t)nwin + p.B - 1) / p.B;
    size_t T = 2 * mean;
    if (T < 16) T = 16;
    if (T > 512) T = 512;
    p.T = (uint32_t)T;
    return p;
}
// n_real scalars (neither 0 nor 1) out of n_table points.  Returns false when no table fits the budget or when
// it would not beat the plain per-window plan by at least 8%.
// force: take the best table plan even if it does not beat the per-window plan by 8 % (small jobs: what a table buys them
// is one bucket set and a host pass of c - 1 doublings instead of 254)
// always: take the best table plan that fits even when the per-window plan is within 8 % of it (at 2^27 points c = 22 is
// within reach of both: the table then saves one bucket reduction out of twelve, not digits)
bool msm_fbw_plan(size_t n_real, size_t n_table, size_t max_bytes, size_t elem_bytes, int* c_out, int* nwin_out, bool force,
                  bool always) {
    if (force) {
        // A small job is bound by the DEPTH of its chain of dependent additions (6 us each in G1, 19 us in G2), not by
        // their number: about T + 1.25 mean / T in the accumulation and its bucket sums (T ~ sqrt(2 mean), see
        // MsmJob::phase_b) plus two per reduction level, (c - 1) / 2 levels.  Wide digits win: few points per bucket.
        double best = 1e300;
        int bc = 0, bw = 0;
        for (int c = 4; c <= 20; c++) {
            const int nwin = (255 + c - 1) / c;
            if ((double)n_table * nwin * (double)elem_bytes > (double)max_bytes) continue;
            const double mean = (double)n_real * nwin / (double)(1u << (c - 1));
            const double t = std::min(64.0, std::max(4.0, std::ceil(std::sqrt(2.0 * mean))));
            // (+ the leaf level's pass over all 2^(c-1) buckets -- four rows of XYZZ points written and read back --
            // in the same unit: about 1e-5 additions' time per bucket)
            double depth = std::min(t, mean + 1.0) + 1.25 * mean / t + 2.0 * ((c - 1 + 1) / 2)
                           + 1e-5 * (double)(1u << (c - 1));
            // The last digit has only `top` significant bits, and with a window table its entries share the bucket set of
            // all the others: n entries pile onto 2^top buckets.  When that is more than a bucket may hold before it counts
            // as heavy, ONE workgroup folds each pile (k_heavy_fold: its items over 128 lanes, then a 7-level tree) -- at
            // n = 2^16 the depth model alone picked c = 18, whose last digit has 2 bits: three buckets of ~16 000 entries,
            // 0.375 ms of a 1.06 ms chain (c = 17: 16 bits, no pile).
            const int top = fbw_top_bits(c, nwin);
            if (top < c - 1) {
                const double pile_items = (double)n_real / (double)(1u << (top > 0 ? top : 0)) / t;
                if (pile_items > 2.5 * mean / t) depth += pile_items / 128.0 + 7.0;
            }
            if (depth < best) {
                best = depth;
                bc = c;
                bw = nwin;
            }
        }
        if (!bc) return false;
        *c_out = bc;
        *nwin_out = bw;
        return true;
    }
    double best_cost = 1e300;
    int bc = 0, bw = 0;
    for (int c = 4; c <= 23; c++) {
        int nwin = (255 + c - 1) / c;
        if (nwin > MAXWIN) continue;
        if ((double)n_table * nwin >= 2.0e9) continue;                 // value = w * n + i must fit 31 bits
        if ((double)n_table * nwin * (double)elem_bytes > (double)max_bytes) continue;
        if (n_real >= (1u << 17) && fbw_top_bits(c, nwin) < 10) continue;   // small jobs: the imbalance is harmless
        double cost = (double)n_real * nwin + 4.8 * (double)(1u << (c - 1)) + 3000.0 * ((c - 1 + TREE_LAMBDA - 1) / TREE_LAMBDA);
        if (cost < best_cost) {
            best_cost = cost;
            bc = c;
            bw = nwin;
        }
    }
    if (!bc) return false;
    const MsmPlan plain = plan_msm(n_real, 0);
    const double plain_cost = (double)n_real * plain.nwin + 4.8 * plain.nwin * (double)plain.B +
                              3000.0 * ((plain.c - 1 + TREE_LAMBDA - 1) / TREE_LAMBDA);
    if (best_cost > 0.92 * plain_cost && !always) return false;
    *c_out = bc;
    *nwin_out = bw;
    return true;
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

// zeros / ones / other scalars (through the index indirection), one atomic per wave and class
// rank (or NULL): the job's compact window table (MsmJob::cw_rank); cls[2] counts the scalars that are neither 0 nor 1
// at positions the table does not cover -- any such scalar sends the job down the per-window path for this proof
__global__ void __launch_bounds__(256) k_classify(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ index,
                                                  const uint8_t* __restrict__ wcls, size_t n,
                                                  uint32_t* __restrict__ cls /*[3]*/, const uint32_t* __restrict__ rank) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n_round = (n + 63) & ~(size_t)63;
    uint32_t zeros = 0, ones = 0, unranked = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        uint32_t c = 0;
        if (i < n && wcls) {
            c = wcls[index ? (size_t)index[i] : i];
        } else if (i < n) {
            Scalar256 s = load_scalar(scalars, index ? (size_t)index[i] : i);
            uint32_t hi = s.w[1] | s.w[2] | s.w[3] | s.w[4] | s.w[5] | s.w[6] | s.w[7];
            c = (hi | s.w[0]) == 0 ? 0u : hi == 0 && s.w[0] == 1 ? 1u : 2u;
        }
        if (i < n) {
            zeros += c == 0;
            ones += c == 1;
            if (rank && c == 2) unranked += rank[i] == 0xffffffffu;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        zeros += __shfl_down(zeros, d);
        ones += __shfl_down(ones, d);
        unranked += __shfl_down(unranked, d);
    }
    // one set of device-scope atomics per WORKGROUP: per wave they were 16 k serialised read-modify-writes on two
    // addresses for a 4 M-position query (0.2 - 0.35 ms of a kernel that reads 20 MB)
    __shared__ uint32_t part[3][4];
    if (lane_id() == 0) {
        part[0][threadIdx.x >> 6] = zeros;
        part[1][threadIdx.x >> 6] = ones;
        part[2][threadIdx.x >> 6] = unranked;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t z = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        const uint32_t o = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        const uint32_t u = part[2][0] + part[2][1] + part[2][2] + part[2][3];
        if (z) atomicAdd(&cls[0], z);
        if (o) atomicAdd(&cls[1], o);
        if (u) atomicAdd(&cls[2], u);
    }
}

// ------------------------------------------------------------------------------------------------ digit sort
// (point, window) pairs are sorted by bucket with a two-level partition that never touches global
// atomics (device-scope atomics from 8 XCDs are fabric transactions on MI355X; the first version of
// this file spent more time in them than in curve arithmetic):
//   level 1  k_part_hist / k_part_scatter: tiles of PART_TP points; per-tile LDS histogram over
//            (window, high bucket bits) bins [+ one bin for scalars == 1]; an exclusive scan over
//            [bin][tile] gives every tile a private output range per bin, ranks come from LDS atomics.
//   level 2  k_part_local: one workgroup per bin sorts its segment by the low bucket bits in LDS,
//            writes the per-bucket counts and the final (base index | sign) order.
static constexpr int PART_TP = 4096;        // points per tile
static constexpr int PART_TP_SMALL = 256;   // small jobs: one point per thread (a 4096-point job would otherwise be ONE
                                            // workgroup walking 16 points x nwin digits per thread: 40 us instead of 5)

__device__ __forceinline__ bool scalar_is_zero(const Scalar256& s) {
    return (s.w[0] | s.w[1] | s.w[2] | s.w[3] | s.w[4] | s.w[5] | s.w[6] | s.w[7]) == 0;
}
__device__ __forceinline__ bool scalar_is_one(const Scalar256& s) {
    return s.w[0] == 1 && (s.w[1] | s.w[2] | s.w[3] | s.w[4] | s.w[5] | s.w[6] | s.w[7]) == 0;
}

// MODE 0: tile_tab[bin * ntiles + tile] = number of the tile's pairs in `bin`
// MODE 1: tile_tab holds the exclusive scan of that table; pairs are written to A[pos] = {val, low bits}
template <int MODE>
__global__ void __launch_bounds__(256) k_part(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ index,
                                              size_t n, int c, int nwin, int kw, uint32_t vstride, uint32_t quad_n, int hb,
                                              int lb, uint32_t ntiles, uint32_t tp, uint32_t* __restrict__ tile_tab,
                                              uint2* __restrict__ A, uint32_t* __restrict__ sorted, int use_ones_bin,
                                              const uint8_t* __restrict__ wcls, int qw, const uint32_t* __restrict__ rank,
                                              uint32_t rank_off) {
    extern __shared__ uint32_t lds_bins[];
    const uint32_t nbins = ((uint32_t)kw << hb) + 1u;  // last bin: scalars == 1
    const uint32_t wsel = kw == 1 ? 0u : ~0u;            // folded windows (fbw table): the window goes into the value
    const uint32_t tile = blockIdx.x;
    for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x)
        lds_bins[b] = MODE == 0 ? 0u : tile_tab[(size_t)b * ntiles + tile];
    __syncthreads();
    const size_t base = (size_t)tile * tp;
    const uint32_t lomask = (1u << lb) - 1u;
    for (uint32_t k = threadIdx.x; k < tp; k += blockDim.x) {
        const size_t i = base + k;
        const bool live = i < n;
        Scalar256 s;
#pragma unroll
        for (int q = 0; q < 8; q++) s.w[q] = 0;
        // use_ones_bin == 0 (dense jobs, which queue the accumulation without reading the ones bucket's size back):
        // a scalar equal to 1 takes the digit path like any other value (digit 1 of window 0)
        bool one, zero;
        if (wcls) {   // class bytes: only a "real" scalar is loaded
            const size_t si = live ? (index ? (size_t)index[i] : i) : 0;
            const uint32_t c = live ? wcls[si] : 0u;
            zero = c == 0;
            one = use_ones_bin && c == 1;
            if (c == 2 || (c == 1 && !use_ones_bin)) s = load_scalar(scalars, si);
        } else {
            if (live) s = load_scalar(scalars, index ? (size_t)index[i] : i);
            one = use_ones_bin && live && scalar_is_one(s);
            zero = !live || scalar_is_zero(s);
        }
        if (quad_n) {
            // Subset-sum table (msm_quad_build): the qw (4 or 8) lanes of an aligned group of positions emit ONE entry for
            // all of the group's scalars equal to 1 -- the table's sum over the positions holding one -- whatever the other
            // positions hold (a field element among them takes its digits as usual).
            const uint64_t mo = __ballot(one);
            const uint32_t qb = lane_id() & ~((uint32_t)qw - 1u), gm = (1u << qw) - 1u;
            const uint32_t q_ones = (uint32_t)(mo >> qb) & gm;
            if (one) {
                if ((lane_id() - qb) == (uint32_t)__builtin_ctz(q_ones)) {
                    uint32_t pos = atomicAdd(&lds_bins[nbins - 1], 1u);
                    if (MODE == 1) sorted[pos] = quad_n + gm * (uint32_t)(i / (uint32_t)qw) + q_ones - 1u;
                }
                continue;
            }
        }
        if (zero) continue;
        if (one) {
            uint32_t pos = atomicAdd(&lds_bins[nbins - 1], 1u);
            if (MODE == 1) sorted[pos] = (uint32_t)i;   // the ones bin is one bucket and the last bin: final order
            continue;
        }
        // (rank: the point's row-0 entry in the compact window table that follows the subset sums, MsmJob::cw_rank)
        const uint32_t vi = rank ? rank_off + rank[i] : (uint32_t)i;
        uint32_t carry = 0;
        for (int w = 0; w < nwin; w++) {
            int32_t d = next_digit(s, c, carry);
            if (d == 0) continue;
            const uint32_t mag = (uint32_t)(d < 0 ? -d : d) - 1u;
            uint32_t pos = atomicAdd(&lds_bins[(((uint32_t)w & wsel) << hb) | (mag >> lb)], 1u);
            if (MODE == 1)
                A[pos] = make_uint2((vi + (uint32_t)w * vstride) | (d < 0 ? 0x80000000u : 0u), mag & lomask);
        }
    }
    if (MODE == 0) {
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) tile_tab[(size_t)b * ntiles + tile] = lds_bins[b];
    }
}

// ------------------------------------------------------------------------------------------------ staged partition
// Large jobs (MsmJob::phase_b picks per job): every level of the partition leaves a workgroup as RUNS of consecutive
// entries per bin.  The unstaged k_part<1> above writes each 8-byte entry where its LDS-atomic rank says: with 4097 bins a
// tile's entries for one bin are a dozen, written one at a time by different waves, and the lines reach HBM partly
// filled -- 275 MB written for 89 MB of entries at 2^22, 11.5 GB for 3.2 GB at 2^26 (profiles/r03_pmc_traffic*.json).
// Here the key (bucket index, c - 1 bits, times the window when windows are not folded) is split three ways:
//   level 1  k_part<0> counts as before, with FEW bins (2^7 or so); k_part_staged scatters: a tile's points are taken in
//            rounds of at most PS_CAP entries, counting-sorted by bin in LDS and written out bin by bin -- a round's run
//            for one bin is ~32 consecutive entries (256 B);
//   level 2  k_p2<0/1>: every level-1 segment is cut into chunks of P2_CHUNK entries; per chunk a histogram over the
//            next key bits, one exclusive scan over [segment][sub-bin][chunk] (which IS the sorted order), then the same
//            LDS-staged scatter.  Skipped when the key space is small enough for two levels;
//   level 3  k_part_local as before: one workgroup per (level-1 bin, level-2 sub-bin) segment, at most 2^7 buckets.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds, uint32_t& total);   // (with the scans below)
__device__ __forceinline__ void p2_plan(const uint32_t* __restrict__ tile_tab, uint32_t ntiles, uint32_t nb1,
                                        uint32_t* __restrict__ p2m, uint32_t* lds);
static constexpr uint32_t PS_CAP = 4096;      // staged entries per round (8 bytes of LDS each)
static constexpr uint32_t P2_CHUNK = 4096;    // entries per level-2 chunk (16 per thread)

// Signed digit w of the scalar at `sw` (8 words) on its own: the same value next_digit() reaches after w steps.  The
// carry into window w is 1 iff the raw digit below exceeds 2^(c-1), or equals it and has a carry itself (then the walk
// goes one window further down: probability 2^-c per step).
__device__ __forceinline__ uint32_t raw_digit(const uint32_t* sw, int w, int c) {
    const int o = w * c, wi = o >> 5, sh = o & 31;   // o < 256: w < nwin = ceil(255 / c)
    const uint64_t lo = sw[wi], hi = wi + 1 < 8 ? sw[wi + 1] : 0u;
    return (uint32_t)(((hi << 32) | lo) >> sh) & ((1u << c) - 1u);
}
__device__ __forceinline__ int32_t digit_lds(const uint32_t* sw, int w, int c) {
    const uint32_t half = 1u << (c - 1);
    uint32_t carry = 0;
    for (int k = w - 1; k >= 0; k--) {
        const uint32_t r = raw_digit(sw, k, c);
        if (r != half) {
            carry = r > half ? 1u : 0u;
            break;
        }
        if (k == 0) carry = 0;   // r == half all the way down: v = half is not above half
    }
    const uint32_t v = raw_digit(sw, w, c) + carry;
    return v > half ? (int32_t)v - (int32_t)(1u << c) : (int32_t)v;
}

// (key, value) of every entry of point i -- the one place the digit walk of k_part lives for the staged kernel
template <class Emit>
__device__ __forceinline__ void walk_digits(Scalar256 s, uint32_t i, int c, int nwin, uint32_t wsel, uint32_t vstride, int hb,
                                            int lb, uint32_t lomask, Emit&& emit) {
    uint32_t carry = 0;
    for (int w = 0; w < nwin; w++) {
        const int32_t d = next_digit(s, c, carry);
        if (d == 0) continue;
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d) - 1u;
        emit((((uint32_t)w & wsel) << hb) | (mag >> lb), (i + (uint32_t)w * vstride) | (d < 0 ? 0x80000000u : 0u), mag & lomask);
    }
}

// level-1 scatter.  Same arguments and bins as k_part<1>; tile_tab holds the exclusive scan of k_part<0>'s table.
__global__ void __launch_bounds__(256) k_part_staged(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ index,
                                                     size_t n, int c, int nwin, int kw, uint32_t vstride, uint32_t quad_n, int hb,
                                                     int lb, uint32_t ntiles, uint32_t tp, const uint32_t* __restrict__ tile_tab,
                                                     uint2* __restrict__ A, uint32_t* __restrict__ sorted, int use_ones_bin,
                                                     const uint8_t* __restrict__ wcls, uint32_t* __restrict__ p2m, int qw,
                                                     const uint32_t* __restrict__ rank, uint32_t rank_off) {
    extern __shared__ uint32_t lds_s[];
    const uint32_t nbins = ((uint32_t)kw << hb) + 1u;   // last bin: scalars == 1 (final order, goes to `sorted`)
    uint32_t* hist = lds_s;                  // [nbins] the round's counts
    uint32_t* lscan = hist + nbins;          // [nbins] their exclusive scan
    uint32_t* cursor = lscan + nbins;        // [nbins] placement cursors
    uint32_t* gbase = cursor + nbins;        // [nbins] where the tile's next entry of a bin goes in A / sorted
    uint32_t* scratch = gbase + nbins;       // [16]
    uint32_t* pfx = scratch + 16;            // [tp + 1] class << 30 | exclusive prefix of the points' entry bounds
    uint32_t* st_v = pfx + tp + 1;           // [PS_CAP]
    uint16_t* st_l = reinterpret_cast<uint16_t*>(st_v + PS_CAP);   // [PS_CAP] low key bits
    uint16_t* st_b = st_l + PS_CAP;                                  // [PS_CAP] bin
    const uint32_t wsel = kw == 1 ? 0u : ~0u;
    const uint32_t tile = blockIdx.x;
    const size_t base = (size_t)tile * tp;
    const uint32_t lomask = (1u << lb) - 1u;
    if (p2m && blockIdx.x == 0) {   // the level-2 plan (for the kernels after this one)
        p2_plan(tile_tab, ntiles, nbins - 1u, p2m, scratch);
        __syncthreads();
    }
    for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) gbase[b] = tile_tab[(size_t)b * ntiles + tile];
    // ---- entry bound of every point (0, 1 or nwin) and its class, prefix-summed over the tile
    {
        const uint32_t per = tp / blockDim.x;   // tp is a multiple of 256
        const uint32_t k0 = threadIdx.x * per;
        uint32_t sum = 0;
        for (uint32_t j = 0; j < per; j++) {
            const size_t i = base + k0 + j;
            uint32_t cls = 0;
            if (i < n) cls = wcls ? wcls[index ? (size_t)index[i] : i] : 2u;
            const uint32_t ub = cls == 0 ? 0u : (cls == 1 && use_ones_bin) ? 1u : (uint32_t)nwin;
            pfx[k0 + j] = (cls << 30) | ub;   // (the bound; replaced by the prefix below)
            sum += ub;
        }
        uint32_t tot;
        uint32_t ex = block_excl_scan(sum, scratch, tot);
        for (uint32_t j = 0; j < per; j++) {
            const uint32_t v = pfx[k0 + j];
            pfx[k0 + j] = (v & 0xC0000000u) | ex;
            ex += v & 0x3FFFFFFFu;
        }
        if (threadIdx.x == blockDim.x - 1) pfx[tp] = ex;
    }
    __syncthreads();
    // one point: class / scalar / subset-sum vote exactly as in k_part, entries handed to emit(bin, value, low)
    auto point = [&](uint32_t k, bool in_round, auto&& emit) {
        const size_t i = base + k;
        const bool live = in_round && i < n;
        Scalar256 s;
#pragma unroll
        for (int q = 0; q < 8; q++) s.w[q] = 0;
        bool one, zero;
        if (wcls) {
            const uint32_t cl = live ? pfx[k] >> 30 : 0u;
            zero = cl == 0;
            one = use_ones_bin && cl == 1;
            if (cl == 2 || (cl == 1 && !use_ones_bin)) s = load_scalar(scalars, index ? (size_t)index[i] : i);
        } else {
            if (live) s = load_scalar(scalars, index ? (size_t)index[i] : i);
            one = use_ones_bin && live && scalar_is_one(s);
            zero = !live || scalar_is_zero(s);
        }
        if (quad_n) {
            const uint64_t mo = __ballot(one);
            const uint32_t qb = lane_id() & ~((uint32_t)qw - 1u), gm = (1u << qw) - 1u;
            const uint32_t q_ones = (uint32_t)(mo >> qb) & gm;
            if (one) {
                if ((lane_id() - qb) == (uint32_t)__builtin_ctz(q_ones))
                    emit(nbins - 1, quad_n + gm * (uint32_t)(i / (uint32_t)qw) + q_ones - 1u, 0u);
                return;
            }
        }
        if (zero) return;
        if (one) {
            emit(nbins - 1, (uint32_t)i, 0u);
            return;
        }
        walk_digits(s, rank ? rank_off + rank[i] : (uint32_t)i, c, nwin, wsel, vstride, hb, lb, lomask, emit);
    };
    // ---- a tile in which every point may contribute all of its nwin digits (the h job, a uniform witness).  A round is
    // PS_CAP / nwin whole points: their scalars come in by coalesced 16-byte loads (issued a round ahead, parked in
    // registers) and sit in LDS -- aliased with the stage, which is only written once the digits are out -- and the round's
    // entries are (point, window) pairs dealt out sixteen per thread: every lane busy whatever nwin is, each digit cut
    // straight out of the scalar's words with its carry from the window below (digit_lds), ranked by ONE returning LDS
    // atomic and kept in registers across the scan.  One global-memory latency per round, hidden behind the round before;
    // the general rounds below make two passes over the scalars with a thread walking its points' digits one by one.
    {
        const uint32_t live = base >= n ? 0u : (uint32_t)std::min<size_t>(tp, n - base);
        const bool full = (pfx[tp] & 0x3FFFFFFFu) == live * (uint32_t)nwin && (wcls || !use_ones_bin) && !quad_n && !rank && nwin >= 8;
        if (full) {
            constexpr uint32_t PER = PS_CAP / 256, PRE = 4;   // nwin >= 8: at most 512 points = 1024 16-byte units per round
            const uint32_t ppr = PS_CAP / (uint32_t)nwin;
            const uint32_t magic = 0xFFFFFFFFu / (uint32_t)nwin + 1u;   // s / nwin = umulhi(s, magic) for s < 2^16
            uint32_t* sc_lds = st_v;   // [ppr * 8] words, aliased with the stage
            uint4 pre[PRE];
            auto fetch = [&](uint32_t p0) {   // the scalars of points [p0, p0 + ppr) of the tile, two 16-byte halves each
                const uint32_t npts = p0 >= live ? 0u : std::min(ppr, live - p0);
#pragma unroll
                for (uint32_t q = 0; q < PRE; q++) {
                    const uint32_t u = q * 256 + threadIdx.x;
                    if (u < 2 * npts) {
                        const size_t i = base + p0 + (u >> 1);
                        pre[q] = reinterpret_cast<const uint4*>(scalars + (index ? (size_t)index[i] : i) * 8)[u & 1u];
                    }
                }
            };
            fetch(0);
            for (uint32_t p0 = 0; p0 < live; p0 += ppr) {
                const uint32_t npts = std::min(ppr, live - p0), nslots = npts * (uint32_t)nwin;
#pragma unroll
                for (uint32_t q = 0; q < PRE; q++) {
                    const uint32_t u = q * 256 + threadIdx.x;
                    if (u < 2 * npts) reinterpret_cast<uint4*>(sc_lds)[u] = pre[q];
                }
                for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) hist[b] = 0;
                __syncthreads();
                fetch(p0 + ppr);   // in flight while this round runs
                uint32_t ev[PER], el[PER], er[PER];   // value | low key and bin | rank in the bin
#pragma unroll
                for (uint32_t e = 0; e < PER; e++) {
                    const uint32_t sl = e * 256 + threadIdx.x;
                    el[e] = 0xffffffffu;
                    if (sl < nslots) {
                        const uint32_t kl = __umulhi(sl, magic), w = sl - kl * (uint32_t)nwin;
                        const int32_t d = digit_lds(sc_lds + 8 * kl, (int)w, c);
                        if (d != 0) {
                            const uint32_t i = (uint32_t)base + p0 + kl;
                            const uint32_t mag = (uint32_t)(d < 0 ? -d : d) - 1u;
                            const uint32_t bin = ((w & wsel) << hb) | (mag >> lb);
                            ev[e] = (i + w * vstride) | (d < 0 ? 0x80000000u : 0u);
                            el[e] = (mag & lomask) | (bin << 16);
                            er[e] = atomicAdd(&hist[bin], 1u);
                        }
                    }
                }
                __syncthreads();   // (every digit is out of sc_lds: the stage may be written)
                uint32_t n_round;
                {
                    const uint32_t per = (nbins + blockDim.x - 1) / blockDim.x, b0 = threadIdx.x * per;
                    uint32_t sum = 0;
                    for (uint32_t q = 0; q < per; q++) sum += b0 + q < nbins ? hist[b0 + q] : 0u;
                    uint32_t ex = block_excl_scan(sum, scratch, n_round);
                    for (uint32_t q = 0; q < per; q++) {
                        if (b0 + q < nbins) {
                            lscan[b0 + q] = ex;
                            ex += hist[b0 + q];
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (uint32_t e = 0; e < PER; e++) {
                    if (el[e] != 0xffffffffu) {
                        const uint32_t bin = el[e] >> 16, pos = lscan[bin] + er[e];
                        st_v[pos] = ev[e];
                        st_l[pos] = (uint16_t)el[e];
                        st_b[pos] = (uint16_t)bin;
                    }
                }
                __syncthreads();
                for (uint32_t pos = threadIdx.x; pos < n_round; pos += blockDim.x) {
                    const uint32_t b = st_b[pos];
                    A[gbase[b] + (pos - lscan[b])] = make_uint2(st_v[pos], st_l[pos]);   // (no ones bin on this path)
                }
                __syncthreads();
                for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) gbase[b] += hist[b];
            }
            return;
        }
    }
    // ---- rounds: points [k_lo, k_hi), both multiples of eight (a subset-sum group never straddles a round), whose
    // entry bounds sum to at most PS_CAP
    for (uint32_t k_lo = 0; k_lo < tp;) {
        const uint32_t p_lo = pfx[k_lo] & 0x3FFFFFFFu;
        uint32_t lo = k_lo / 8 + 1, hi = tp / 8;   // group indices; the answer is the largest g with pfx[8 g] - p_lo <= PS_CAP
        while (lo < hi) {                          // (a group's bound is at most 8 nwin <= 1024)
            const uint32_t mid = (lo + hi + 1) / 2;
            if ((pfx[8 * mid] & 0x3FFFFFFFu) - p_lo <= PS_CAP) lo = mid; else hi = mid - 1;
        }
        const uint32_t k_hi = 8 * lo;
        const uint32_t bound = (pfx[k_hi] & 0x3FFFFFFFu) - p_lo;
        if (bound == 0) {   // nothing but zeros up to k_hi (uniform: every thread reads the same words)
            k_lo = k_hi;
            continue;
        }
        for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        const uint32_t j_lo = k_lo / blockDim.x, j_hi = (k_hi + blockDim.x - 1) / blockDim.x;
        for (uint32_t j = j_lo; j < j_hi; j++) {
            const uint32_t k = j * blockDim.x + threadIdx.x;
            point(k, k >= k_lo && k < k_hi, [&](uint32_t bin, uint32_t, uint32_t) { atomicAdd(&hist[bin], 1u); });
        }
        __syncthreads();
        uint32_t n_round;
        {   // exclusive scan of the round's counts
            const uint32_t per = (nbins + blockDim.x - 1) / blockDim.x, b0 = threadIdx.x * per;
            uint32_t sum = 0;
            for (uint32_t q = 0; q < per; q++) sum += b0 + q < nbins ? hist[b0 + q] : 0u;
            uint32_t ex = block_excl_scan(sum, scratch, n_round);
            for (uint32_t q = 0; q < per; q++) {
                if (b0 + q < nbins) {
                    lscan[b0 + q] = ex;
                    cursor[b0 + q] = ex;
                    ex += hist[b0 + q];
                }
            }
        }
        __syncthreads();
        for (uint32_t j = j_lo; j < j_hi; j++) {
            const uint32_t k = j * blockDim.x + threadIdx.x;
            point(k, k >= k_lo && k < k_hi, [&](uint32_t bin, uint32_t val, uint32_t low) {
                const uint32_t pos = atomicAdd(&cursor[bin], 1u);
                st_v[pos] = val;
                st_l[pos] = (uint16_t)low;
                st_b[pos] = (uint16_t)bin;
            });
        }
        __syncthreads();
        for (uint32_t pos = threadIdx.x; pos < n_round; pos += blockDim.x) {
            const uint32_t b = st_b[pos];
            const uint32_t g = gbase[b] + (pos - lscan[b]);
            if (b == nbins - 1) sorted[g] = st_v[pos];
            else A[g] = make_uint2(st_v[pos], st_l[pos]);
        }
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) gbase[b] += hist[b];
        k_lo = k_hi;
    }
}

// level-2 plan (one workgroup -- the first of k_part_staged: the scanned level-1 table is all it needs):
// p2m = [nb1 + 2] segment starts (regular bins, the ones bin, the total) | [nb1 + 1] exclusive prefix of the segments' chunk counts
__device__ __forceinline__ void p2_plan(const uint32_t* __restrict__ tile_tab, uint32_t ntiles, uint32_t nb1,
                                        uint32_t* __restrict__ p2m, uint32_t* lds /*[16]*/) {
    uint32_t* seg_off = p2m;
    uint32_t* cb = p2m + nb1 + 2;
    for (uint32_t s = threadIdx.x; s < nb1 + 2; s += blockDim.x) seg_off[s] = tile_tab[(size_t)s * ntiles];
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t s0 = 0; s0 <= nb1; s0 += blockDim.x) {
        const uint32_t sg = s0 + threadIdx.x;
        const uint32_t len = sg < nb1 ? seg_off[sg + 1] - seg_off[sg] : 0u;
        const uint32_t nch = (len + P2_CHUNK - 1) / P2_CHUNK;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(nch, lds, tot);
        if (sg <= nb1) cb[sg] = carry + ex;
        carry += tot;
    }
}
// this workgroup's (segment, chunk): binary search in the chunk prefix
__device__ __forceinline__ bool p2_locate(const uint32_t* __restrict__ p2m, uint32_t nb1, uint32_t* sg, uint32_t* chunk,
                                          uint32_t* nch, uint32_t* start, uint32_t* end, uint32_t* cbase) {
    const uint32_t* seg_off = p2m;
    const uint32_t* cb = p2m + nb1 + 2;
    if (blockIdx.x >= cb[nb1]) return false;
    uint32_t lo = 0, hi = nb1 - 1;   // largest s with cb[s] <= blockIdx.x (empty segments share a value: the search lands
    while (lo < hi) {                // on the last of them, which is the one that owns the chunk)
        const uint32_t mid = (lo + hi + 1) / 2;
        if (cb[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    *sg = lo;
    *cbase = cb[lo];
    *chunk = blockIdx.x - cb[lo];
    *nch = cb[lo + 1] - cb[lo];
    *start = seg_off[lo] + *chunk * P2_CHUNK;
    const uint32_t e = seg_off[lo + 1];
    *end = e - *start < P2_CHUNK ? e : *start + P2_CHUNK;
    return true;
}
// MODE 0: tab2[nsub * cb[seg] + sub * nch[seg] + chunk] = the chunk's entries with sub-bin `sub` (bits [b3, b3 + b2) of
// the low key); MODE 1: tab2 scanned; the chunk is counting-sorted by sub-bin in LDS and written to `out` as runs,
// the low key cut down to its b3 bits
template <int MODE>
__global__ void __launch_bounds__(256) k_p2(const uint2* __restrict__ in, const uint32_t* __restrict__ p2m, uint32_t nb1,
                                            int b2, int b3, uint32_t* __restrict__ tab2, uint2* __restrict__ out) {
    __shared__ uint32_t hist[128], lscan[128], gb[128], scratch[16];
    __shared__ uint32_t st_v[MODE ? P2_CHUNK : 1];
    __shared__ uint16_t st_l[MODE ? P2_CHUNK : 1];
    __shared__ uint8_t st_b[MODE ? P2_CHUNK : 1];
    uint32_t sg, chunk, nch, start, end, cbase;
    const uint32_t nsub = 1u << b2, submask = nsub - 1u, lowmask = (1u << b3) - 1u;
    if (!p2_locate(p2m, nb1, &sg, &chunk, &nch, &start, &end, &cbase)) {
        // the grid is sized from a bound on the entries; the table entries of the chunks that do not exist -- a dense tail
        // after the nsub * chunks used ones -- must read zero for the scan (no memset launch)
        if (MODE == 0 && threadIdx.x < nsub) tab2[(size_t)blockIdx.x * nsub + threadIdx.x] = 0;
        return;
    }
    const size_t idx0 = (size_t)nsub * cbase + chunk;
    constexpr uint32_t PER = P2_CHUNK / 256;
    uint2 e[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t i = start + k * 256 + threadIdx.x;
        e[k] = i < end ? in[i] : make_uint2(0u, 0u);
    }
    if (MODE == 0) {
        if (threadIdx.x < nsub) hist[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; k++)
            if (start + k * 256 + threadIdx.x < end) atomicAdd(&hist[(e[k].y >> b3) & submask], 1u);
        __syncthreads();
        if (threadIdx.x < nsub) tab2[idx0 + (size_t)threadIdx.x * nch] = hist[threadIdx.x];
        return;
    }
    uint32_t cnt = 0;
    if (threadIdx.x < nsub) {
        const size_t ix = idx0 + (size_t)threadIdx.x * nch;
        const uint32_t g = tab2[ix];
        gb[threadIdx.x] = g;
        cnt = tab2[ix + 1] - g;   // the next table entry in scan order: the next chunk of this sub-bin, or what follows it
    }
    uint32_t n_t;
    const uint32_t ex = block_excl_scan(cnt, scratch, n_t);
    if (threadIdx.x < nsub) {
        lscan[threadIdx.x] = ex;
        hist[threadIdx.x] = ex;   // placement cursor
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        if (start + k * 256 + threadIdx.x < end) {
            const uint32_t sb = (e[k].y >> b3) & submask;
            const uint32_t pos = atomicAdd(&hist[sb], 1u);
            st_v[pos] = e[k].x;
            st_l[pos] = (uint16_t)(e[k].y & lowmask);
            st_b[pos] = (uint8_t)sb;
        }
    }
    __syncthreads();
    for (uint32_t pos = threadIdx.x; pos < n_t; pos += 256) {
        const uint32_t b = st_b[pos];
        out[gb[b] + (pos - lscan[b])] = make_uint2(st_v[pos], st_l[pos]);
    }
}
// start of level-3 segment `bin` = (level-1 bin, sub-bin) in the order of the scanned level-2 table; bin == nb1 << b2 is
// the ones bin, one more the total
__device__ __forceinline__ uint32_t p3_start(const uint32_t* __restrict__ p2m, const uint32_t* __restrict__ tab2, uint32_t nb1,
                                             int b2, uint32_t bin) {
    const uint32_t sg = bin >> b2, sub = bin & ((1u << b2) - 1u);
    if (sg >= nb1) return p2m[nb1 + sub];   // seg_off[nb1] (ones), seg_off[nb1 + 1] (total)
    const uint32_t* cb = p2m + nb1 + 2;
    const uint32_t c0 = cb[sg], nch = cb[sg + 1] - c0;
    if (nch == 0) return p2m[sg];           // an empty level-1 segment
    return tab2[((size_t)c0 << b2) + (size_t)sub * nch];
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

// The kernels take both item sizes in one word: points per item of a regular bucket | points per item of the ones
// bucket << 16 (item_sizes() on the host).
__device__ __forceinline__ uint32_t items_of(uint32_t cnt, uint32_t T, uint32_t idx, uint32_t nkeys) {
    const uint32_t t = idx + 1 == nkeys ? T >> 16 : T & 0xffffu;
    return (cnt + t - 1) / t;
}

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
        si += items_of(cnt, T, idx, nkeys);
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
// (256 threads: a 1024-thread workgroup needs a CU with 16 free wave slots and waited up to 3 ms behind the
// long-running accumulation waves of the other jobs)
__global__ void __launch_bounds__(256) k_scan_bsums(uint32_t* bsum, uint32_t nb) {
    __shared__ uint32_t lds[16];
    for (int arr = 0; arr < 2; arr++) {
        uint32_t* a = bsum + (size_t)arr * nb;
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nb; base += blockDim.x) {
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
        si += items_of(cnt[k], T, idx, nkeys);
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
            if (cursor && idx < nkeys) cursor[idx] = ec;
        }
        ec += cnt[k];
        ei += items_of(cnt[k], T, idx, nkeys);
    }
}

// ---- plain exclusive scan of a u32 array (in place), entry n receives the total
__global__ void __launch_bounds__(SCAN_THREADS) k_xscan_reduce(const uint32_t* __restrict__ v, size_t n,
                                                               uint32_t* __restrict__ bsum) {
    __shared__ uint32_t lds[8];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_PER_THREAD;
    uint32_t sc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) sc += base + k < n ? v[base + k] : 0u;
    uint32_t tot;
    block_excl_scan(sc, lds, tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) k_xscan_bsums(uint32_t* bsum, uint32_t nb) {
    __shared__ uint32_t lds[16];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += blockDim.x) {
        uint32_t idx = base + threadIdx.x;
        uint32_t val = idx < nb ? bsum[idx] : 0;
        uint32_t tot;
        uint32_t ex = block_excl_scan(val, lds, tot);
        if (idx < nb) bsum[idx] = carry + ex;
        carry += tot;
    }
}
__global__ void __launch_bounds__(SCAN_THREADS) k_xscan_final(uint32_t* __restrict__ v, size_t n,
                                                              const uint32_t* __restrict__ bsum) {
    __shared__ uint32_t lds[8];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_PER_THREAD;
    uint32_t x[SCAN_PER_THREAD];
    uint32_t sc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        x[k] = base + k < n ? v[base + k] : 0u;
        sc += x[k];
    }
    uint32_t tot;
    uint32_t e = block_excl_scan(sc, lds, tot) + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        if (base + k <= n) v[base + k] = e;   // v has n + 1 entries
        e += x[k];
    }
}

// ---- the same two scans as ONE single-block launch each, for tables of at most SCAN_ONE_MAX entries: a small job is a
// chain of dependent launches (~5 us each whatever they do), and three launches to scan 2000 words are two too many
// (16 tiles: the 2^15 + 1 bucket counts of a c = 16 small job, a 257-bin x 64-tile partition table -- 16 trips of one
// workgroup, ~10 us, against three dependent launches)
static constexpr size_t SCAN_ONE_MAX = 4 * (size_t)SCAN_TILE;
// (Four times that for jobs in small-job mode -- the 2^15 + 1 bucket counts of a c = 16 plan in 16 trips of one workgroup
// instead of three launches -- was measured and is slower: 2^14 proof 0.92 against 0.87 ms.)   // jobs in small-job mode (a chain of launches is what they cost)
__global__ void __launch_bounds__(SCAN_THREADS) k_xscan_one(uint32_t* __restrict__ v, size_t n) {
    __shared__ uint32_t lds[8];
    uint32_t carry = 0;
    for (size_t t0 = 0; t0 <= n; t0 += SCAN_TILE) {   // v has n + 1 entries
        const size_t base = t0 + (size_t)threadIdx.x * SCAN_PER_THREAD;
        uint32_t x[SCAN_PER_THREAD];
        uint32_t sc = 0;
#pragma unroll
        for (int k = 0; k < SCAN_PER_THREAD; k++) {
            x[k] = base + k < n ? v[base + k] : 0u;
            sc += x[k];
        }
        uint32_t tot;
        uint32_t e = block_excl_scan(sc, lds, tot) + carry;
#pragma unroll
        for (int k = 0; k < SCAN_PER_THREAD; k++) {
            if (base + k <= n) v[base + k] = e;
            e += x[k];
        }
        carry += tot;
    }
}
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_one(const uint32_t* __restrict__ counts, uint32_t nkeys, uint32_t T,
                                                           uint32_t* __restrict__ offsets, uint32_t* __restrict__ item_off) {
    __shared__ uint32_t lds[8];
    uint32_t cc = 0, ci = 0;
    for (uint32_t t0 = 0; t0 <= nkeys; t0 += SCAN_TILE) {   // entry nkeys holds the totals
        const uint32_t base = t0 + threadIdx.x * SCAN_PER_THREAD;
        uint32_t cnt[SCAN_PER_THREAD];
        uint32_t sc = 0, si = 0;
#pragma unroll
        for (int k = 0; k < SCAN_PER_THREAD; k++) {
            const uint32_t idx = base + k;
            cnt[k] = idx < nkeys ? counts[idx] : 0;
            sc += cnt[k];
            si += items_of(cnt[k], T, idx, nkeys);
        }
        uint32_t tc, ti;
        uint32_t ec = block_excl_scan(sc, lds, tc) + cc;
        uint32_t ei = block_excl_scan(si, lds, ti) + ci;
#pragma unroll
        for (int k = 0; k < SCAN_PER_THREAD; k++) {
            const uint32_t idx = base + k;
            if (idx <= nkeys) {
                offsets[idx] = ec;
                item_off[idx] = ei;
            }
            ec += cnt[k];
            ei += items_of(cnt[k], T, idx, nkeys);
        }
        cc += tc;
        ci += ti;
    }
}

// level 2: one workgroup per bin; segment [tile_off[bin*ntiles], tile_off[(bin+1)*ntiles]) of A
// A long segment (a 2^26-point job: ~10^5 entries per bin) is scattered TILE entries at a time through LDS: the tile is
// counting-sorted there and leaves as runs of consecutive words per bucket.  Writing sorted[cursor++] entry by entry, as
// the short-segment path does, fills every 64-byte line of a 400 KB output region over the workgroup's whole lifetime --
// 200 MB of half-written lines in flight across the chip, far beyond the L2s: 20.5 ms for the 403 M entries of one part of
// a 2^26-point multiexp (19.7 G entries/s against 117 G/s on the 6.6 K-entry segments of a 2^22 proof, whose 218 MB of
// entries also still sit in the 256 MB Infinity Cache when this kernel reads them).  Staged, with the counting pass's loads
// eight deep and the next tile prefetched: 7.5 ms (counting 1.8, tiles 3.7, writes 2.0); the multiexp 103 -> 86 ms.
static constexpr uint32_t PART_STAGE_TILE = 8192;
// ctr[b] += 1 for every lane that is `on`, with ONE LDS atomic per distinct bucket among the wave's lanes; returns the lane's
// rank (the value an atomicAdd of its own would have returned, in lane order).  For SKEWED segments only: when one bucket
// holds most of a segment (a quarter of a witness equal to one constant: a million entries in one bucket) the 64 lanes of
// every wave take turns on its counter -- 2.5 ms for such a segment; on evenly spread entries the loop below would run once
// per lane instead.  Executed by whole waves (the shuffles read every lane).
__device__ __forceinline__ uint32_t wave_agg_add(uint32_t* ctr, uint32_t b, bool on) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t r = 0;
    unsigned long long todo = __ballot(on);
    while (todo) {   // (wave-uniform)
        const int first = __ffsll((long long)todo) - 1;
        const uint32_t lead = __shfl(b, first);
        const unsigned long long same = __ballot(on && b == lead) & todo;
        uint32_t base = 0;
        if ((int)lane == first) base = atomicAdd(&ctr[lead], (uint32_t)__popcll(same));
        base = __shfl(base, first);
        if ((same >> lane) & 1ull) r = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return r;
}
// (p2m != nullptr: the segments are those of the staged three-level partition, see p3_start)
__global__ void __launch_bounds__(1024) k_part_local(const uint2* __restrict__ A, const uint32_t* __restrict__ tile_off,
                                                    uint32_t ntiles, uint32_t nbins, int lb, uint32_t stage_min,
                                                    uint32_t* __restrict__ counts, uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ p2m, const uint32_t* __restrict__ tab2,
                                                    uint32_t nb1, int b2, uint32_t single_max) {
    extern __shared__ uint32_t lds_l[];   // [2^lb] histogram / cursors, 16 words of scan scratch; staged: see below
    const uint32_t bin = blockIdx.x;
    const int bits = bin + 1 == nbins ? 0 : lb;   // the ones bin is a single bucket
    const uint32_t nl = 1u << bits;
    uint32_t* scratch = lds_l + (1u << lb);
    const uint32_t start = p2m ? p3_start(p2m, tab2, nb1, b2, bin) : tile_off[(size_t)bin * ntiles];
    const uint32_t end = p2m ? p3_start(p2m, tab2, nb1, b2, bin + 1)
                             : tile_off[(size_t)(bin + 1) * ntiles];   // the scanned table has nbins*ntiles + 1 entries
    if (bin + 1 == nbins) {   // scalars == 1: k_part<1> already wrote them in final order
        if (threadIdx.x == 0) counts[bin << lb] = end - start;
        return;
    }
    for (uint32_t b = threadIdx.x; b < nl; b += blockDim.x) lds_l[b] = 0;
    __syncthreads();
    if (single_max && end - start <= single_max && nl <= blockDim.x) {
        // A segment that fits the stage goes through in ONE pass: its entries are read once (eight per thread, in
        // registers), ranked by one returning LDS atomic, placed in LDS by bucket and written out as consecutive words --
        // where the general path below reads the segment twice and writes sorted[cursor++] entry by entry.
        // [2^lb] histogram | 16 scratch | [2^lb] bucket starts | [single_max] values
        constexpr uint32_t PER1 = PART_STAGE_TILE / 1024;
        uint32_t* bstart = scratch + 16;
        uint32_t* stv = bstart + (1u << lb);
        const uint32_t len = end - start;
        uint2 e1[PER1];
        uint32_t r1[PER1];
#pragma unroll
        for (uint32_t k = 0; k < PER1; k++) {
            const uint32_t i = k * 1024 + threadIdx.x;
            if (i < len) {
                e1[k] = A[start + i];
                r1[k] = atomicAdd(&lds_l[e1[k].y], 1u);
            }
        }
        __syncthreads();
        uint32_t tot1;
        const uint32_t cnt1 = threadIdx.x < nl ? lds_l[threadIdx.x] : 0u;
        const uint32_t ex1 = block_excl_scan(cnt1, scratch, tot1);
        if (threadIdx.x < nl) {
            bstart[threadIdx.x] = ex1;
            counts[(bin << lb) + threadIdx.x] = cnt1;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER1; k++)
            if (k * 1024 + threadIdx.x < len) stv[bstart[e1[k].y] + r1[k]] = e1[k].x;
        __syncthreads();
        for (uint32_t pos = threadIdx.x; pos < len; pos += 1024) sorted[start + pos] = stv[pos];
        return;
    }
    // Is one bucket most of this segment?  1024 samples, each compared with its wave's first: from a quarter of them equal the
    // segment takes the wave-aggregated counters (wave_agg_add) in its three passes; every other segment the plain atomics.
    bool skew = false;
    if (end - start >= 16u * 1024u) {
        // (four runs of 256 consecutive entries at the quarter points: 8 KB in whole lines -- 1024 scattered samples were 64 KB of
        // lines for a segment of 128 KB, 0.05 ms per sort of a SHA-256 proof)
        const uint32_t y0 = A[start + (uint32_t)(((uint64_t)(end - start - 256u) * (threadIdx.x >> 8)) / 3u) + (threadIdx.x & 255u)].y;
        const uint32_t eq = (uint32_t)__popcll(__ballot(y0 == __shfl(y0, 0)));
        if ((threadIdx.x & 63u) == 0) scratch[threadIdx.x >> 6] = eq;
        __syncthreads();
        uint32_t tot_eq = 0;
        for (uint32_t k = 0; k < blockDim.x / 64; k++) tot_eq += scratch[k];
        skew = tot_eq >= blockDim.x / 4;
        __syncthreads();
    }
    if (skew) {   // (whole-block rounds: the ballots need whole waves)
        for (uint32_t i0 = start; i0 < end; i0 += blockDim.x) {
            const uint32_t i = i0 + threadIdx.x;
            const bool on = i < end;
            (void)wave_agg_add(lds_l, on ? A[i].y : 0u, on);
        }
    } else {   // (eight loads in flight per thread: a long segment is otherwise a chain of 96 load -> atomic round trips)
        constexpr uint32_t U = 8;
        uint32_t i = start + threadIdx.x;
        for (; i + (U - 1) * blockDim.x < end; i += U * blockDim.x) {
            uint32_t y[U];
#pragma unroll
            for (uint32_t k = 0; k < U; k++) y[k] = A[i + k * blockDim.x].y;
#pragma unroll
            for (uint32_t k = 0; k < U; k++) atomicAdd(&lds_l[y[k]], 1u);
        }
        for (; i < end; i += blockDim.x) atomicAdd(&lds_l[A[i].y], 1u);
    }
    __syncthreads();
    // exclusive prefix over the nl local buckets: each thread owns a run of consecutive buckets
    const uint32_t per = (nl + blockDim.x - 1) / blockDim.x;
    const uint32_t b0 = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; k++) sum += b0 + k < nl ? lds_l[b0 + k] : 0u;
    uint32_t tot;
    uint32_t ex = block_excl_scan(sum, scratch, tot);
    const uint32_t key0 = bin << lb;
    for (uint32_t k = 0; k < per; k++) {
        if (b0 + k < nl) {
            uint32_t cnt = lds_l[b0 + k];
            counts[key0 + b0 + k] = cnt;
            lds_l[b0 + k] = start + ex;   // becomes the bucket's write cursor
            ex += cnt;
        }
    }
    __syncthreads();
    if (end - start < stage_min || nl > blockDim.x) {
        if (skew) {
            for (uint32_t i0 = start; i0 < end; i0 += blockDim.x) {
                const uint32_t i = i0 + threadIdx.x;
                const bool on = i < end;
                const uint2 e = on ? A[i] : make_uint2(0u, 0u);
                const uint32_t pos = wave_agg_add(lds_l, e.y, on);
                if (on) sorted[pos] = e.x;
            }
            return;
        }
        for (uint32_t i = start + threadIdx.x; i < end; i += blockDim.x) {
            uint2 e = A[i];
            sorted[atomicAdd(&lds_l[e.y], 1u)] = e.x;
        }
        return;
    }
    // staged: [2^lb] cursors | 16 scratch | [2^lb] tile counts | [2^lb] tile offsets | [TILE] values | [TILE] buckets (u16)
    constexpr uint32_t TILE = PART_STAGE_TILE, PER = TILE / 1024;
    uint32_t* cur = lds_l;
    uint32_t* th = scratch + 16;
    uint32_t* toff = th + (1u << lb);
    uint32_t* stage_v = toff + (1u << lb);
    uint16_t* stage_b = reinterpret_cast<uint16_t*>(stage_v + TILE);
    if (threadIdx.x < nl) th[threadIdx.x] = 0;
    __syncthreads();
    // (the next tile's entries are loaded while this one goes through LDS: a tile is otherwise one memory round trip with
    // nothing to overlap it -- the barriers keep the workgroup's waves in step)
    uint2 e[PER], nx[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t i = start + k * 1024 + threadIdx.x;
        nx[k] = i < end ? A[i] : make_uint2(0u, 0u);
    }
    for (uint32_t t0 = start; t0 < end; t0 += TILE) {
        const uint32_t n_t = end - t0 < TILE ? end - t0 : TILE;
        uint32_t rk[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            e[k] = nx[k];
            const uint32_t i = t0 + TILE + k * 1024 + threadIdx.x;
            if (i < end) nx[k] = A[i];
        }
        if (skew) {
#pragma unroll
            for (uint32_t k = 0; k < PER; k++) rk[k] = wave_agg_add(th, e[k].y, k * 1024 + threadIdx.x < n_t);
        } else {
#pragma unroll
            for (uint32_t k = 0; k < PER; k++)
                if (k * 1024 + threadIdx.x < n_t) rk[k] = atomicAdd(&th[e[k].y], 1u);
        }
        __syncthreads();
        uint32_t tot_t;
        const uint32_t ex_t = block_excl_scan(threadIdx.x < nl ? th[threadIdx.x] : 0u, scratch, tot_t);
        if (threadIdx.x < nl) toff[threadIdx.x] = ex_t;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            if (k * 1024 + threadIdx.x < n_t) {
                const uint32_t pos = toff[e[k].y] + rk[k];
                stage_v[pos] = e[k].x;
                stage_b[pos] = (uint16_t)e[k].y;
            }
        }
        __syncthreads();
        for (uint32_t pos = threadIdx.x; pos < n_t; pos += 1024) {
            const uint32_t b = stage_b[pos];
            sorted[cur[b] + (pos - toff[b])] = stage_v[pos];
        }
        __syncthreads();
        if (threadIdx.x < nl) {
            cur[threadIdx.x] += th[threadIdx.x];
            th[threadIdx.x] = 0;
        }
        __syncthreads();
    }
}

// meta[0] = total entries, meta[1] = total items, meta[2], meta[3] = item range of the ones bucket (key nkeys-1)
__global__ void k_meta(const uint32_t* offsets, const uint32_t* item_off, uint32_t nkeys, uint32_t* meta) {
    if (threadIdx.x == 0) {
        meta[0] = offsets[nkeys];
        meta[1] = item_off[nkeys];
        meta[2] = item_off[nkeys - 1];
        meta[3] = item_off[nkeys];
    }
}

// item -> key map for the regular keys (the ones key, last, is filled by a memset)
// (with_last: small jobs queued without read-backs have no host-side range for that memset: the grid writes the ones
// bucket's items too)
__global__ void __launch_bounds__(256) k_build_items(const uint32_t* __restrict__ item_off, uint32_t nkeys,
                                                     uint32_t* __restrict__ item_key, int with_last,
                                                     uint32_t* __restrict__ zeroed, uint32_t nzero) {
    uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    // (also clears the words the kernels after this one count in: item-size histogram, its scan, heavy-bucket counter)
    for (uint32_t i = key; i < nzero; i += gridDim.x * blockDim.x) zeroed[i] = 0;
    if (with_last) {   // all threads share the ones bucket's items (a bit-heavy witness: tens of thousands of them)
        const uint32_t lo = item_off[nkeys - 1], hi = item_off[nkeys];
        for (uint32_t t = lo + key; t < hi; t += gridDim.x * blockDim.x) item_key[t] = nkeys - 1;
    }
    uint32_t lo = 0, hi = 0;
    if (key + 1u < nkeys) {
        lo = item_off[key];
        hi = item_off[key + 1];
    }
    // A bucket of a few items is written by its own thread; a big one (skewed scalars: a quarter of the witness holding one
    // constant puts a million entries into one bucket -- one thread then wrote 10^5 items, 1.1 ms at the head of the job's
    // chain) by the 64 lanes of its wave, one such bucket after the other.
    constexpr uint32_t OWN = 16;
    if (hi - lo <= OWN)
        for (uint32_t t = lo; t < hi; t++) item_key[t] = key;
    unsigned long long big = __ballot(hi - lo > OWN);
    const uint32_t lane = threadIdx.x & 63u;
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t blo = __shfl(lo, src), bhi = __shfl(hi, src), bkey = __shfl(key, src);
        for (uint32_t t = blo + lane; t < bhi; t += 64) item_key[t] = bkey;
    }
}

// Items sorted by their number of points (counting sort, longest first): a wave of k_accumulate then holds 64
// items of (nearly) the same length.  Bucket loads of a dense job are Poisson(~104): in bucket order the longest of
// 64 lanes is ~1.23x the mean and every lane waits for it.
static constexpr uint32_t ITEM_CLASSES = 513;   // item sizes 0 .. 512
__device__ __forceinline__ uint32_t item_size(uint32_t t, const uint32_t* __restrict__ item_key,
                                              const uint32_t* __restrict__ item_off, const uint32_t* __restrict__ offsets,
                                              uint32_t T, uint32_t ones_key) {
    const uint32_t key = item_key[t];
    const uint32_t Tk = key == ones_key ? T >> 16 : T & 0xffffu;
    const uint32_t start = offsets[key] + (t - item_off[key]) * Tk;
    const uint32_t left = offsets[key + 1] - start;
    return left < Tk ? left : Tk;
}
__global__ void __launch_bounds__(256) k_item_hist(const uint32_t* __restrict__ item_key, const uint32_t* __restrict__ item_off,
                                                   const uint32_t* __restrict__ offsets, uint32_t nkeys, uint32_t T,
                                                   uint32_t ones_key, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[ITEM_CLASSES];
    for (uint32_t i = threadIdx.x; i < ITEM_CLASSES; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint32_t total = item_off[nkeys];
    // (grid-stride in whole-block rounds: the ballots below need whole waves.  A wave whose items all have one size --
    // the items of a heavy bucket are all T points long -- makes ONE atomic instead of 64 on one word)
    for (uint32_t t0 = blockIdx.x * blockDim.x; t0 < total; t0 += gridDim.x * blockDim.x) {
        const uint32_t t = t0 + threadIdx.x;
        const bool on = t < total;
        const uint32_t sz = on ? item_size(t, item_key, item_off, offsets, T, ones_key) : 0u;
        const unsigned long long act = __ballot(on);
        const int first = __ffsll((long long)act) - 1;
        const uint32_t lead = __shfl(sz, first < 0 ? 0 : first);
        if (act && __ballot(on && sz == lead) == act) {
            if ((int)(threadIdx.x & 63u) == first) atomicAdd(&h[lead], (uint32_t)__popcll(act));
        } else if (on) {
            atomicAdd(&h[sz], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ITEM_CLASSES; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}
// `hist` is complete (k_item_hist); `taken` (zeroed, ITEM_CLASSES words) counts what the workgroups have claimed per size.
// Every workgroup derives the classes' starts itself -- first[s] = number of items longer than s (longest first): a
// 513-word suffix sum -- instead of a single-thread kernel between the two (one launch less in every job's chain).
__global__ void __launch_bounds__(256) k_item_scatter(const uint32_t* __restrict__ item_key, const uint32_t* __restrict__ item_off,
                                                      const uint32_t* __restrict__ offsets, uint32_t nkeys, uint32_t T,
                                                      uint32_t ones_key, const uint32_t* __restrict__ hist,
                                                      uint32_t* __restrict__ taken, uint32_t* __restrict__ perm) {
    __shared__ uint32_t h[ITEM_CLASSES], base[ITEM_CLASSES], first[ITEM_CLASSES + 3], sc[8];
    const uint32_t total = item_off[nkeys];
    {   // suffix sums of hist: thread t owns sizes 3t, 3t + 1, 3t + 2 (ITEM_CLASSES = 513 <= 3 * 256), taken from the top
        const int hi = (int)ITEM_CLASSES - 1 - 3 * (int)threadIdx.x;   // this thread's largest size
        uint32_t v[3], sum = 0;
        for (int q = 0; q < 3; q++) {
            v[q] = hi - q >= 0 ? hist[hi - q] : 0u;
            sum += v[q];
        }
        uint32_t tot;
        uint32_t ex = block_excl_scan(sum, sc, tot);   // items of sizes above this thread's
        for (int q = 0; q < 3; q++) {
            if (hi - q >= 0) first[hi - q] = ex;
            ex += v[q];
        }
        __syncthreads();
    }
    // grid-stride in whole-block chunks so that every thread of a block sees the same number of rounds
    for (uint32_t t0 = blockIdx.x * blockDim.x; t0 < total; t0 += gridDim.x * blockDim.x) {
        for (uint32_t i = threadIdx.x; i < ITEM_CLASSES; i += blockDim.x) h[i] = 0;
        __syncthreads();
        const uint32_t t = t0 + threadIdx.x;
        uint32_t sz = 0, rank = 0;
        {   // (as k_item_hist: one atomic for a wave of equal sizes, ranks by lane order)
            const bool on = t < total;
            if (on) sz = item_size(t, item_key, item_off, offsets, T, ones_key);
            const unsigned long long act = __ballot(on);
            const int first = __ffsll((long long)act) - 1;
            const uint32_t lead = __shfl(sz, first < 0 ? 0 : first);
            if (act && __ballot(on && sz == lead) == act) {
                uint32_t base0 = 0;
                if ((int)(threadIdx.x & 63u) == first) base0 = atomicAdd(&h[lead], (uint32_t)__popcll(act));
                base0 = __shfl(base0, first);
                rank = base0 + (uint32_t)__popcll(act & ((1ull << (threadIdx.x & 63u)) - 1ull));
            } else if (on) {
                rank = atomicAdd(&h[sz], 1u);
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < ITEM_CLASSES; i += blockDim.x)
            if (h[i]) base[i] = first[i] + atomicAdd(&taken[i], h[i]);
        __syncthreads();
        if (t < total) perm[base[sz] + rank] = t;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ stage 1
// L / LP: the lazy arithmetic and the lanes per point -- (Fq2PL, 2) runs a G2 point on a lane pair (ecd.h): half the
// dependent instruction stream per addition, which is what a latency-bound job's chain is made of.
// Shared digit sort (MsmJobBase::src): the witness queries l, a, b_g1, b_g2 multiply the SAME scalars w_j, so their
// (position, digit) pairs are sorted ONCE, over the witness positions, and every query accumulates over that one list.  An
// entry then names (position j, table row w) -- SH_POS_BITS bits of position, the row above them, the sign in bit 31 --
// and the query maps j to its own dense index: through its rank map (a, b: position -> index in the query, ~0 = the
// variable is not in the query's density: the entry is skipped) or by an offset (l: the aux variables in order).
static constexpr int SH_POS_BITS = 26;
struct ShQuery {
    const uint32_t* rank;   // [positions of the sort] or NULL (identity)
    uint32_t sub, nq;       // index = rank[j] - sub, valid below nq (= the query's points and its table's row stride)
};
__device__ __forceinline__ uint32_t sh_index(uint32_t e, const ShQuery& q) {
    const uint32_t j = e & ((1u << SH_POS_BITS) - 1u), w = (e >> SH_POS_BITS) & 31u;
    uint32_t r = q.rank ? q.rank[j] : j;
    r -= q.sub;   // (~0 - sub stays above nq: sub + nq < 2^31)
    return r < q.nq ? w * q.nq + r : 0xffffffffu;
}

// Several queries in ONE launch (gridDim.y of them): the followers of a shared sort accumulate over the same items, fold the
// same heavy buckets and run the same reduction levels -- only the table, the position map and the buffers differ -- so the
// G1 queries of a small proof make one chain of launches instead of three (the accumulation, the heavy-bucket fold and the
// levels take the per-query pointers from this table, indexed by blockIdx.y; a job on its own is a batch of one).
static constexpr int BATCH_MAX = 3;
template <class F>
struct BatchQ {
    const Affine<F>* bases;     // k_accumulate: the query's bases / table
    XYZZ<F>* partial;           // item sums
    const XYZZ<F>* lvl_in;      // k_bits_level: this level's input rows (NULL at the leaf level) ...
    XYZZ<F>* lvl_out;           // ... and output rows
    XYZZ<F>* ones_out;          // k_heavy_fold: where the folded ones bucket goes (or NULL)
    ShQuery shq;
};
template <class F>
struct Batch {
    BatchQ<F> q[BATCH_MAX];
};

template <class F, class L = typename LazyOf<F>::L, int LP = 1, bool SH = false>
__global__ void __launch_bounds__(256) k_accumulate(Batch<F> batch,
                                                    const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ offsets,
                                                    const uint32_t* __restrict__ item_off,
                                                    const uint32_t* __restrict__ item_key,
                                                    const uint32_t* __restrict__ perm, uint32_t nkeys,
                                                    uint32_t T, uint32_t ones_key) {
    const Affine<F>* __restrict__ bases = batch.q[blockIdx.y].bases;
    XYZZ<F>* __restrict__ partial = batch.q[blockIdx.y].partial;
    const ShQuery shq = batch.q[blockIdx.y].shq;
    uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / LP;
    if (t >= item_off[nkeys]) return;   // the grid may be sized from an upper bound (dense jobs skip the read-back)
    if (perm) t = perm[t];              // items in order of decreasing length
    uint32_t key = item_key[t];
    uint32_t Tk = key == ones_key ? T >> 16 : T & 0xffffu;
    uint32_t seg = t - item_off[key];
    uint32_t start = offsets[key] + seg * Tk;
    uint32_t end = offsets[key + 1];
    if (end > start + Tk) end = start + Tk;
    XYZZd<L> acc = XYZZd<L>::inf();
    if constexpr (SH && sizeof(F) == sizeof(Fq) && LP == 1) {
        // the same software pipeline one stage deeper: entry k + 3 and the rank of entry k + 2 are in flight with the base of
        // entry k + 1 while point k is added
        if (start < end) {
            uint32_t e0 = sorted[start];
            uint32_t e1 = start + 1 < end ? sorted[start + 1] : 0u;
            uint32_t e2 = start + 2 < end ? sorted[start + 2] : 0u;
            uint32_t i0 = sh_index(e0, shq);
            uint32_t i1 = start + 1 < end ? sh_index(e1, shq) : 0xffffffffu;
            Affine<F> cur = bases[i0 != 0xffffffffu ? i0 : 0u];
            for (uint32_t k = start; k < end; k++) {
                Affine<F> nxt = cur;
                if (i1 != 0xffffffffu) nxt = bases[i1];
                const uint32_t i2 = k + 2 < end ? sh_index(e2, shq) : 0xffffffffu;
                const uint32_t e3 = k + 3 < end ? sorted[k + 3] : 0u;
                if (i0 != 0xffffffffu) {
                    AffD<L> p = AffD<L>::load(&cur);
                    if (e0 >> 31) p = p.negated_lazy();
                    acc = acc.add_mixed(p);
                }
                cur = nxt;
                i0 = i1;
                i1 = i2;
                e0 = e1;
                e1 = e2;
                e2 = e3;
            }
        }
    } else if constexpr (SH) {
        for (uint32_t k = start; k < end; k++) {
            const uint32_t e = sorted[k];
            const uint32_t i = sh_index(e, shq);
            if (i == 0xffffffffu) continue;   // (uniform over the LP lanes of a point: they hold the same entry)
            AffD<L> p = AffD<L>::load(&bases[i]);
            if (e >> 31) p = p.negated_lazy();
            acc = acc.add_mixed(p);
        }
    } else if constexpr (sizeof(F) == sizeof(Fq)) {
        // Software pipeline: the gather of point k+1 (index two steps ahead) is in flight while point k is added, so the
        // ~1 us of dependent load latency (sorted[k] -> bases[...]) hides behind the ~5 us mixed addition instead of
        // parking the wave (SQ_WAIT_ANY was 13 % of the wave cycles).  16 more VGPRs; G2 has none to spare.
        if (start < end) {
            uint32_t e = sorted[start];
            uint32_t e_next = start + 1 < end ? sorted[start + 1] : 0u;
            Affine<F> cur = bases[e & 0x7fffffffu];
            for (uint32_t k = start; k < end; k++) {
                Affine<F> nxt = cur;
                if (k + 1 < end) nxt = bases[e_next & 0x7fffffffu];
                const uint32_t e_nn = k + 2 < end ? sorted[k + 2] : 0u;
                AffD<L> p = AffD<L>::load(&cur);
                if (e >> 31) p = p.negated_lazy();   // -y is only multiplied inside add_mixed: no carry propagation
                acc = acc.add_mixed(p);
                cur = nxt;
                e = e_next;
                e_next = e_nn;
            }
        }
    } else {
        for (uint32_t k = start; k < end; k++) {
            uint32_t e = sorted[k];
            AffD<L> p = AffD<L>::load(&bases[e & 0x7fffffffu]);
            if (e >> 31) p = p.negated_lazy();
            acc = acc.add_mixed(p);
        }
    }
    acc.store(&partial[t]);
}

// ------------------------------------------------------------------------------------------------ stage 2: bucket tree
// Buckets split into more than HEAVY_ITEMS items (the top window of a small-c plan, or skewed
// scalars) are folded by a whole workgroup first: partial[first item] receives the bucket sum, and the
// leaf kernel reads only that.  heavy_list / heavy_cnt are filled by k_heavy_collect.
// heavy_items: the per-job threshold (4 for throughput-bound jobs; small jobs, whose items are short, sum up to about
// twice the mean bucket load serially, see MsmJob::phase_b).  force_key (or ~0u): a key listed whenever it holds an item --
// the ones bucket of a job queued without read-backs, whose fold k_heavy_fold also copies to ones_out.
static constexpr uint32_t HEAVY_ITEMS = 4;
// A heavy bucket is folded in UNITS of HEAVY_CH items, one workgroup each: a bucket of a million entries (skewed scalars) was
// ONE workgroup's 10^5 dependent-ish additions, 1.5 ms at the tail of its job.  A bucket of several units: every unit's sum
// goes to fold_tmp, and the workgroup that finishes LAST (a counter per bucket) folds those -- one launch, no host round trip.
// unit = {key, chunk, chunks of the key, first slot of the key in fold_tmp / its counter in hdone}.
// heavy_cnt[0] counts units, heavy_cnt[1] the slots handed out to buckets of more than one unit.
static constexpr uint32_t HEAVY_CH = 1024;
__global__ void __launch_bounds__(256) k_heavy_collect(const uint32_t* __restrict__ item_off, uint32_t nkeys_listed,
                                                       uint32_t heavy_items, uint32_t force_key,
                                                       uint32_t* __restrict__ heavy_cnt, uint4* __restrict__ units,
                                                       uint32_t* __restrict__ hdone, uint32_t nb, uint32_t tmp_stride) {
    uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= nkeys_listed) return;
    const uint32_t items = item_off[key + 1] - item_off[key];
    if (!(items > heavy_items || (key == force_key && items > 0))) return;
    const uint32_t nch = (items + HEAVY_CH - 1) / HEAVY_CH;
    const uint32_t u0 = atomicAdd(&heavy_cnt[0], nch);
    uint32_t t0 = 0;
    if (nch > 1) {
        t0 = atomicAdd(&heavy_cnt[1], nch);
        for (uint32_t y = 0; y < nb; y++) hdone[(size_t)y * tmp_stride + t0] = 0;
    }
    for (uint32_t c = 0; c < nch; c++) units[u0 + c] = make_uint4(key, c, nch, t0);
}
template <class F, class L = typename LazyOf<F>::L, int LP = 1>
__global__ void __launch_bounds__(128 * LP) k_heavy_fold(Batch<F> batch, const uint32_t* __restrict__ item_off,
                                                         const uint32_t* __restrict__ heavy_cnt,
                                                         const uint4* __restrict__ units, uint32_t ones_key,
                                                         XYZZ<F>* __restrict__ fold_tmp, uint32_t* __restrict__ hdone,
                                                         uint32_t tmp_stride) {
    XYZZ<F>* __restrict__ partial = batch.q[blockIdx.y].partial;
    XYZZ<F>* __restrict__ ones_out = batch.q[blockIdx.y].ones_out;
    XYZZ<F>* __restrict__ ftmp = fold_tmp + (size_t)blockIdx.y * tmp_stride;
    uint32_t* __restrict__ dn = hdone + (size_t)blockIdx.y * tmp_stride;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
    XYZZ<F>* lds = reinterpret_cast<XYZZ<F>*>(smem_h);
    __shared__ uint32_t last_unit;
    const uint32_t tid = threadIdx.x / LP, bdim = blockDim.x / LP;   // points, not lanes
    const uint32_t cnt = heavy_cnt[0];
    for (uint32_t slot = blockIdx.x; slot < cnt; slot += gridDim.x) {
        const uint4 u = units[slot];
        const uint32_t key = u.x, nch = u.z, t0 = u.w;
        const uint32_t lo0 = item_off[key], hi0 = item_off[key + 1];
        // ONE copy of the summation in the code (the second round, for the workgroup that arrives last at a bucket of several
        // units, re-enters it with the units' sums as its source: two inlined copies cost 248 registers and 344 bytes of scratch
        // against 166 and none, and a normal proof's fold 0.2 ms instead of 0.08)
        const XYZZ<F>* src = partial;
        uint32_t lo = lo0 + u.y * HEAVY_CH, hi = lo + HEAVY_CH < hi0 ? lo + HEAVY_CH : hi0;
        bool final_round = nch == 1;
        for (;;) {
            XYZZd<L> acc = XYZZd<L>::inf();
            for (uint32_t i = lo + tid; i < hi; i += bdim) acc = acc.add(XYZZd<L>::load(&src[i]));
            acc.store(&lds[tid]);
            __syncthreads();
            for (uint32_t st = bdim >> 1; st > 0; st >>= 1) {
                if (tid < st) {
                    acc = acc.add(XYZZd<L>::load(&lds[tid + st]));
                    acc.store(&lds[tid]);
                }
                __syncthreads();
            }
            if (final_round) {
                if (tid == 0) {
                    acc.store(&partial[lo0]);
                    if (key == ones_key) acc.store(ones_out);
                }
                break;
            }
            if (tid == 0) acc.store(&ftmp[t0 + u.y]);
            __threadfence();   // the unit's sum is out before its arrival is counted (the units of a bucket run on any XCD)
            __syncthreads();
            if (threadIdx.x == 0) last_unit = atomicAdd(&dn[t0], 1u) == nch - 1 ? 1u : 0u;
            __syncthreads();
            if (!last_unit) break;   // (uniform over the workgroup)
            __threadfence();         // every other unit of this bucket has arrived: fold their sums
            src = ftmp;
            lo = t0;
            hi = t0 + nch;
            final_round = true;
        }
        __syncthreads();
    }
}

// Bucket reduction by bit planes.  A window's sum  sum_b (b+1) bucket_b  equals  S + sum_j 2^j T_j  with
// S = sum of all buckets and T_j = sum of the buckets whose index has bit j set: only plain sums, no doublings
// and no long dependent chains on the device (the 2^j weights are applied by the host's final Horner pass).
// Level l folds groups of 2^f consecutive entries (f = 2, or 1 for a last odd bit): the group sums feed the next
// level and are what S ends up as; per group and per bit j < f it emits D_j = sum of the entries with bit j set,
// and k_rowsum adds the D rows up.  Four adds per thread and level instead of a 24-add running-sum chain.
struct RowMeta {
    int nlev;
    uint32_t d_off[16], len[16], rows[16], chunks[16], block_base[16];
};
static constexpr uint32_t ROWSUM_CH = 1024;   // entries per k_rowsum block

// one bucket's sum from its items (heavy buckets were folded into their first item)
template <class L, class F>
__device__ __forceinline__ XYZZd<L> bucket_sum(const XYZZ<F>* __restrict__ partial, const uint32_t* __restrict__ item_off,
                                               uint32_t key, uint32_t heavy_items) {
    uint32_t lo = item_off[key], hi = item_off[key + 1];
    if (hi - lo > heavy_items) hi = lo + 1;
    XYZZd<L> r = XYZZd<L>::inf();
    for (uint32_t i = lo; i < hi; i++) r = r.add(XYZZd<L>::load(&partial[i]));
    return r;
}

// One level of the reduction.  `in` holds rows_in rows of (len_out << f) entries: the first kw rows are the windows'
// running group sums ("S rows"; in == nullptr at the leaf level: the buckets themselves), the others the bit-plane rows
// emitted by earlier levels, which only need summing.  Every row is folded 2^f : 1; an S row also emits f new rows
// (entries with index bit b set), appended after the rows_in existing ones.  So all rows of a level have the same
// length, shrink together, and after the last level `out` is [kw] S, [kw * bits] T in (level, window, bit) order --
// no separate row-sum launches (they were 0.25 ms of the 0.87 ms tail of the h multiexp).
// Two lanes per group, two children each: lane A adds c0 + c1 while lane B adds c2 + c3 (= bit 1); they swap c1 and
// that sum through LDS, then A finishes the group sum while B adds c1 + c3 (= bit 0) -- two dependent additions per
// level with every lane busy, where one thread per group ran four (and held four points: 311 registers for G1,
// spills for G2).
static constexpr uint32_t BITS_THREADS = 128;
template <class F, class L, int LP>
// (t: this lane group's index in the level, tid: its index in the workgroup -- k_bits_level derives both from the block and
// thread indices, k_bits_tail walks a level in chunks of one workgroup)
__device__ __forceinline__ void bits_level_body(const XYZZ<F>* __restrict__ in, const XYZZ<F>* __restrict__ partial,
                                                const uint32_t* __restrict__ item_off, uint32_t rows_in, uint32_t kw,
                                                uint32_t len_out, int f, uint32_t heavy_items, uint32_t ones_row,
                                                uint32_t ones_key, XYZZ<F>* __restrict__ out, uint32_t t, uint32_t tid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    XYZZ<F>* lds = reinterpret_cast<XYZZ<F>*>(smem_b);   // one entry per point held (LP lanes each)
    const uint32_t g = t >> 1, q = t & 1u;
    const bool in_range = g < rows_in * len_out;
    const uint32_t row = in_range ? g / len_out : 0u, idx = g - row * len_out;
    const bool full = row < kw;   // an S row: emits bit-plane rows
    XYZZ<F>* outS = out + (size_t)row * len_out + idx;
    XYZZ<F>* outD = out + ((size_t)rows_in + (size_t)row * f) * len_out + idx;   // + b * len_out for bit b
    const uint32_t child0 = (row * len_out + idx) << f;   // rows of len_out << f entries
    // ones_row (leaf level of a small job, else ~0u): one more input row right after the S rows whose entries are the
    // ITEM sums of the ones bucket (padded with the identity) -- a row that only needs summing, so the levels fold it
    // along instead of a workgroup folding it on its own between the accumulation and the levels.
    const bool ones = !in && in_range && row == ones_row;
    const uint32_t ones_lo = ones ? item_off[ones_key] : 0u, ones_cnt = ones ? item_off[ones_key + 1] - ones_lo : 0u;
    auto fetch = [&](uint32_t child) -> XYZZd<L> {
        if (in) return XYZZd<L>::load(&in[child]);
        if (ones) {
            const uint32_t cl = child - ((row * len_out) << f);
            return cl < ones_cnt ? XYZZd<L>::load(&partial[ones_lo + cl]) : XYZZd<L>::inf();
        }
        return bucket_sum<L, F>(partial, item_off, child, heavy_items);
    };
    XYZZd<L> lo = XYZZd<L>::inf(), hi = XYZZd<L>::inf();
    if constexpr (LP == 2) {
        // G2: ONE call site of `fetch`.  Used once, it is inlined (bucket_sum and its addition with it); used four times, as
        // in the G1 form below, the compiler calls it -- closure, arguments and the returned point through a frame in scratch
        // memory: 440 B per lane and ~60 reloads per call, every one waited for in a level that is a chain of two dependent
        // additions.
        // f == 1 (uniform; the last level of an odd bit count): two children, lane A does it all.  Else lane A: children 0, 1;
        // lane B: children 2, 3.
        const bool act = f == 1 ? in_range && q == 0 : in_range;
        const uint32_t ch = f == 1 ? child0 : child0 + 2u * q;
        if (act) {
#pragma clang loop unroll(disable)
            for (int k = 0; k < 2; k++) {
                const XYZZd<L> v = fetch(ch + (uint32_t)k);
                if (k == 0) lo = v;
                else hi = v;
            }
        }
        if (f == 1) {
            if (act) {
                if (full) hi.store(outD);
                lo.add(hi).store(outS);
            }
            return;
        }
    } else {
        if (f == 1) {   // uniform; the last level of an odd bit count: two children, lane A does it all
            if (in_range && q == 0) {
                XYZZd<L> c0 = fetch(child0);
                XYZZd<L> c1 = fetch(child0 + 1);
                if (full) c1.store(outD);
                c0.add(c1).store(outS);
            }
            return;
        }
        // lane A: children 0, 1; lane B: children 2, 3
        if (in_range) {
            const uint32_t ch = child0 + 2u * q;
            lo = fetch(ch);
            hi = fetch(ch + 1);
        }
    }
    XYZZd<L> r = lo.add(hi);                       // A: c0 + c1;  B: c2 + c3 (bit 1)
    (q == 0 ? hi : r).store(&lds[tid]);            // A publishes c1, B its sum
    if (in_range && full && q == 1) r.store(outD + len_out);
    __syncthreads();
    if (in_range && (q == 0 || full)) {
        const XYZZd<L> other = XYZZd<L>::load(&lds[tid ^ 1u]);
        (q == 0 ? r : hi).add(other).store(q == 0 ? outS : outD);   // A: group sum;  B: c3 + c1 (bit 0)
    }
}

template <class F, class L = typename LazyOf<F>::L, int LP = 1>
__global__ void __launch_bounds__(BITS_THREADS * LP) k_bits_level(Batch<F> batch, const uint32_t* __restrict__ item_off, uint32_t rows_in,
                                                             uint32_t kw, uint32_t len_out, int f, uint32_t heavy_items,
                                                             uint32_t ones_row, uint32_t ones_key) {
    bits_level_body<F, L, LP>(batch.q[blockIdx.y].lvl_in, batch.q[blockIdx.y].partial, item_off, rows_in, kw, len_out, f, heavy_items,
                              ones_row, ones_key, batch.q[blockIdx.y].lvl_out, (blockIdx.x * blockDim.x + threadIdx.x) / LP,
                              threadIdx.x / LP);
}
// The LAST levels of a reduction have a few dozen groups each: launches of one or two workgroups whose cost is the launch.
// One workgroup runs them all, level after level (a level in chunks of BITS_THREADS lane groups), with a barrier in between:
// two or three launches fewer at the end of every multiexp's chain.  Level i reads the buffer level i - 1 wrote (the two
// buffers alternate: lvl_in is what the first of these levels reads, lvl_out what it writes).
static constexpr int TAIL_LEVELS_MAX = 4;
static constexpr uint32_t TAIL_LANES_MAX = 2 * BITS_THREADS;   // a level joins the tail if it has at most this many lane groups
struct TailLevels {
    int n;
    uint32_t rows_in[TAIL_LEVELS_MAX], len_out[TAIL_LEVELS_MAX];
    int f[TAIL_LEVELS_MAX];
};
template <class F, class L = typename LazyOf<F>::L, int LP = 1>
__global__ void __launch_bounds__(BITS_THREADS * LP) k_bits_tail(Batch<F> batch, const uint32_t* __restrict__ item_off, TailLevels TL,
                                                            uint32_t kw, uint32_t heavy_items, uint32_t ones_key) {
    XYZZ<F>* buf[2] = {const_cast<XYZZ<F>*>(batch.q[blockIdx.y].lvl_in), batch.q[blockIdx.y].lvl_out};
    for (int i = 0; i < TL.n; i++) {
        const uint32_t lanes = TL.rows_in[i] * TL.len_out[i] * 2u;
        for (uint32_t base = 0; base < lanes; base += BITS_THREADS) {
            bits_level_body<F, L, LP>(buf[i & 1], batch.q[blockIdx.y].partial, item_off, TL.rows_in[i], kw, TL.len_out[i], TL.f[i],
                                      heavy_items, ~0u, ones_key, buf[(i & 1) ^ 1], base + threadIdx.x / LP, threadIdx.x / LP);
            __syncthreads();   // (the level's rows are written: the next level -- or chunk: the LDS slots -- may go on)
        }
    }
}
// One block sums up to ROWSUM_CH consecutive entries of one row; rows are grouped in "levels" of equal length
// (RowMeta).  Output: one entry per block, in block order -- again rows of `chunks` entries, so the kernel is
// applied until every row has length 1.
template <class F, class L = typename LazyOf<F>::L, int LP = 1>
__global__ void __launch_bounds__(256 * LP) k_rowsum(const XYZZ<F>* __restrict__ in, RowMeta M, XYZZ<F>* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
    XYZZ<F>* lds = reinterpret_cast<XYZZ<F>*>(smem_r);
    const uint32_t tid = threadIdx.x / LP, bdim = blockDim.x / LP;   // points, not lanes
    int lv = 0;
    while (lv + 1 < M.nlev && blockIdx.x >= M.block_base[lv + 1]) lv++;
    const uint32_t local = blockIdx.x - M.block_base[lv];
    const uint32_t row = local / M.chunks[lv], chunk = local - row * M.chunks[lv];
    const uint32_t len = M.len[lv];
    const size_t base = (size_t)M.d_off[lv] + (size_t)row * len;
    const uint32_t lo = chunk * ROWSUM_CH, hi = lo + ROWSUM_CH < len ? lo + ROWSUM_CH : len;
    XYZZd<L> acc = XYZZd<L>::inf();
    for (uint32_t i = lo + tid; i < hi; i += bdim) acc = acc.add(XYZZd<L>::load(&in[base + i]));
    uint32_t active = hi - lo < bdim ? hi - lo : bdim;   // points holding a term
    if (active > 1) {
        acc.store(&lds[tid]);
        __syncthreads();
        uint32_t span = 1;
        while (span < active) span <<= 1;
        for (uint32_t st = span >> 1; st > 0; st >>= 1) {
            if (tid < st && tid + st < active) {
                acc = acc.add(XYZZd<L>::load(&lds[tid + st]));
                acc.store(&lds[tid]);
            }
            __syncthreads();
            if (active > st) active = st;
        }
    }
    if (tid == 0) acc.store(&out[blockIdx.x]);
}

// ZA_AMD_TRACE=2: host time stamps inside a job's phase (finds the HIP call a phase blocks in)
static void trace_point(const char* what) {
    if (trace_level() < 2) return;
    static const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "[za_amd trace]     %-28s %.3f ms\n", what,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

// ------------------------------------------------------------------------------------------------ host driver
// An MSM runs in three device phases separated by two tiny host read-backs (class counts, then
// totals).  msm_batch() drives several jobs phase by phase, each on its own HIP stream, so the
// latency-bound tree launches of one MSM overlap the throughput-bound accumulation of the next.
struct MsmJobBase {
    virtual ~MsmJobBase() {}
    virtual int phase_a() = 0;
    virtual int phase_b() = 0;
    virtual int phase_c() = 0;
    virtual int finish() = 0;
    hipStream_t st = nullptr;
    bool done = false;
    bool chained = false;   // phase_b also queued phase_c's work (dense jobs)
    // a job split into parts (the h multiexp): part k sorts after part k-1 has sorted and accumulates after part
    // k-1 has accumulated, so its memory-bound sort overlaps the previous part's ALU-bound accumulation and the
    // previous part's latency-bound reduction overlaps its own accumulation
    MsmJobBase* dep = nullptr;
    hipEvent_t ev_sorted = nullptr, ev_acc = nullptr;
    bool launched = false;   // phase_b has run (its events are recorded)
    bool eager = false;      // phase_a leaves nothing to wait for (dense jobs): phase_b is queued right behind it
    bool witness = false;    // one of the proof's witness multiexps (run_jobs' on_witness fires when the last of them is done)
    // Shared digit sort (see ShQuery): a job with `src` set runs no sort of its own -- it accumulates over the sorted entries,
    // bucket offsets and items of `src`, a sort-only job over the witness positions, through its rank map.  The fields below
    // are what such followers (of either group) read from the sorter once its phases have run on the host.
    MsmJobBase* src = nullptr;
    bool sort_only = false;
    bool c_launched = false;          // phase_c has run on the host: the item arrays are queued, the totals below are set
    hipEvent_t ev_items = nullptr;    // recorded behind the last kernel that writes the item arrays
    MsmScratch* Sb = nullptr;         // this job's scratch (the sorter's: sorted / offsets / item_off / item_key / perm)
    MsmPlan P{};
    size_t n_pos = 0;                 // points (positions) of this job's own sort
    uint32_t ones_key = 0, nkeys = 0;
    uint32_t t_ones = T_ONES, t_pack = 0;   // item sizes (regular | ones << 16) as the kernels take them
    uint32_t heavy_items = HEAVY_ITEMS;     // buckets of more items go through k_heavy_fold
    bool order_items = true;          // items sorted by length before the accumulation (not for small jobs)
    bool eager_small = false;         // (see MsmJob)
    bool use_cw_b = false;            // this proof's entries index the compact window table (MsmJob::use_cw)
    uint32_t total_items_h = 0, ones_lo_h = 0, ones_hi_h = 0;   // what phase_c launched with
};

// Which arithmetic the latency-bound kernels of a job run: G1 one lane per point; G2 a lane PAIR per point (ecd.h Fq2PL).
template <class F> struct ChainCfg { using L = typename LazyOf<F>::L; static constexpr int LP = 1; };
template <> struct ChainCfg<Fq2> { using L = Fq2PL; static constexpr int LP = 2; };

template <class F>
struct MsmJob : MsmJobBase {
    Ctx* ctx;
    MsmScratch& S;
    const Affine<F>* d_bases;
    const uint32_t* d_scalars;
    const uint32_t* d_index;
    const uint8_t* d_cls = nullptr;   // MsmDesc::cls for this job's range
    size_t n;
    XYZZ<F>* out;
    bool dense = false;   // plan from n directly, no k_classify round trip
    // A small job (see msm_batch5) is latency bound: it is planned from n as if every scalar were "real" and its whole
    // chain is queued at once like a dense job's -- no classification or totals read-back -- but it keeps the ones
    // bucket (a bit-heavy witness would otherwise pile its ones onto one bucket of window 0), reduced on the device
    // by k_ones_sum from the device-side item range.
    // (eager_small: in MsmJobBase)
    ShQuery shq{nullptr, 0u, 0u};     // follower of a shared sort: how this query reads the sorter's entries
    // Followers of one shared sort in small-job mode run as ONE chain of launches (see Batch): the leader queues the
    // accumulation, the heavy-bucket fold, the levels and the read-backs of its members too, on its own stream
    MsmJob* leader = nullptr;
    MsmJob* members[BATCH_MAX - 1] = {nullptr, nullptr};
    int n_members = 0;
    bool fbw_virtual = false;         // sort-only job: the window-table plan (fbw_c, fbw_nwin, fbw_stride) without a table
    const Affine<F>* quad_table = nullptr;  // [n bases][15 subset sums per 4 positions] (msm_quad_build)
    uint32_t quad_base = 0;   // entry (relative to quad_table) of group 0's sums; 0 = n (the job covers the whole query)
    int quad_w = 4;           // positions per subset-sum group (4 or 8)
    // Compact window table (prover.hip ensure_tables): behind the subset sums of quad_table sit cw_nwin rows of cw_n
    // points, 2^(cw_c w) P_i for the positions i that hold field elements (a circuit's bit signals are structural: the
    // positions are those of the witness the table was built from); cw_rank[i] = the position's column, or ~0.  The job
    // then folds its windows into ONE bucket set like a window-table job while its ones keep their subset sums.  A
    // proof whose witness has a field element anywhere else (k_classify counts them) takes the per-window path.
    const uint32_t* cw_rank = nullptr;
    uint32_t cw_off = 0, cw_n = 0;
    int cw_c = 0, cw_nwin = 0;
    bool use_cw = false;
    const Affine<F>* fbw_table = nullptr;   // fixed-base window table: [fbw_nwin][fbw_stride] affine points
    int fbw_c = 0, fbw_nwin = 0;
    size_t fbw_stride = 0;
    uint32_t items_bound = 0;      // dense jobs: upper bound used instead of the read-back total
    bool ones_in_levels = false;          // small jobs: the ones bucket's item sums are a row of the reduction levels
    static constexpr uint32_t ITEM_ORDER_MIN = 16384;
    static constexpr size_t QUAD_ACC_MAX_LANES = 262144;    // four-lane mixed additions: launches of up to this many lanes
    static constexpr size_t QUAD_LEVEL_MAX_LANES = 32768;   // four-lane additions in the reduction: levels of up to this many logical lanes
    int acc_prof = -1;
    int grid = 1;
    const XYZZ<F>* ones_ptr = nullptr;
    int red_nlev = 0, red_f[16], red_bitbase[16];   // bit-plane reduction levels (bits folded per level)
    uint32_t red_rowbase[16];
    static constexpr bool IS_G1 = sizeof(F) == sizeof(Fq);
    static constexpr uint64_t ALG_BYTES = sizeof(Affine<F>) + 32;  // base + scalar, each read once

    MsmJob(Ctx* c, MsmScratch& scratch, hipStream_t stream, const Affine<F>* b, const uint32_t* sc, const uint32_t* idx,
           size_t n_, XYZZ<F>* o)
        : ctx(c), S(scratch), d_bases(b), d_scalars(sc), d_index(idx), n(n_), out(o) {
        st = stream;
        Sb = &scratch;
        n_pos = n_;
        *out = XYZZ<F>::inf();
        if (n == 0) done = true;
    }

    int phase_a() override {
        if (done) return ZG_OK;
        if (src) {   // nothing to classify or sort: the sorter's phases carry this job
            eager = src->eager;
            if (!S.host) ZG_HIP(ctx, hipHostMalloc((void**)&S.host, MsmScratch::HOST_BYTES, hipHostMallocDefault));
            return ZG_OK;
        }
        if (n >= (1ull << 31)) return fail(ctx, ZG_E_INVALID_ARG, "msm: n = %zu exceeds 2^31-1", n);
        if (!S.host) ZG_HIP(ctx, hipHostMalloc((void**)&S.host, MsmScratch::HOST_BYTES, hipHostMallocDefault));
        grid = (int)std::min<size_t>((n + 255) / 256, (size_t)ctx->num_cu * 4);
        ZG_HIP(ctx, S.meta.reserve(64));
        if (dense || eager_small) {   // counts are taken as (0 zeros, 0 ones): the plan and the buffer bounds stay valid for any data
            uint32_t* cls = reinterpret_cast<uint32_t*>(S.host);
            cls[0] = cls[1] = cls[2] = 0;
            eager = true;
            return ZG_OK;
        }
        ZG_HIP(ctx, hipMemsetAsync(S.meta.p, 0, 64, st));
        ProfScope ps(ctx, dense ? PROF_MSM_SORT_DENSE : PROF_MSM_SORT, n * 32, st);
        ZG_LAUNCH(ctx, k_classify, dim3(grid), dim3(256), 0, st, d_scalars, d_index, d_cls, n, S.meta.as<uint32_t>() + 8,
                           cw_rank);
        ZG_HIP(ctx, hipMemcpyAsync(S.host, S.meta.as<uint32_t>() + 8, 12, hipMemcpyDeviceToHost, st));
        return ZG_OK;
    }

    int phase_b() override {
        if (done) {
            launched = true;
            return ZG_OK;
        }
        if (src) {   // (run_jobs calls this once the sorter's phase_b has run)
            launched = true;
            if (src->done) {   // every scalar is zero
                done = true;
                return ZG_OK;
            }
            if (src->chained) {   // the sorter queued its item arrays too: this job's whole chain follows now
                chained = true;
                items_bound = src->total_items_h;
                return phase_c();
            }
            return ZG_OK;
        }
        const uint32_t* cls = reinterpret_cast<const uint32_t*>(S.host);
        const size_t n_zero = cls[0], n_ones = cls[1], n_real = n - n_zero - n_ones;
        if (n_real == 0 && n_ones == 0) {  // all scalars zero
            done = true;
            launched = true;
            if (ev_sorted) ZG_HIP(ctx, hipEventRecord(ev_sorted, st));
            if (ev_acc) ZG_HIP(ctx, hipEventRecord(ev_acc, st));
            return ZG_OK;
        }
        if (dep && dep->ev_sorted) ZG_HIP(ctx, hipStreamWaitEvent(st, dep->ev_sorted, 0));
        const bool has_fbw = fbw_table || fbw_virtual;
        use_cw = cw_rank && !has_fbw && !dense && !eager_small && cls[2] == 0 && n_real > 0;
        use_cw_b = use_cw;
        P = has_fbw ? plan_fbw_for(fbw_c, fbw_nwin, n_real) : use_cw ? plan_fbw_for(cw_c, cw_nwin, n_real) : plan_msm(n_real, n_ones);
        {   // Small jobs are latency bound: a thread runs its item's mixed adds one after the other (64 G2 adds are
            // 1.1 ms whatever n is).  Below ~64K items the items are shortened -- down to half the mean bucket load, so
            // that only buckets above twice the mean need the heavy-bucket fold -- and the GPU gets more, shorter threads.
            // (round 5: the target is what fills k_accumulate's wave slots -- three waves of 64 one-lane items per SIMD in G1,
            // two waves of lane pairs in G2 -- and an item may go down to a quarter of the mean bucket load.  With 65 536 items
            // and half the mean, the h job of a 1/8 shard (2^19 points, 2^16 buckets of ~120) ran ONE wave per SIMD: 96 ps per
            // addition against the 71 ps of a full launch.)
            const size_t entries = n_real * (size_t)P.nwin, target = IS_G1 ? 3 * 65536 : 65536;
            heavy_items = HEAVY_ITEMS;
            if (entries / P.T < target) {
                size_t t2 = std::max<size_t>(8, entries / target);
                t2 = std::max<size_t>(t2, (entries / P.nkeys + 3) / 4);
                if (t2 < P.T) P.T = (uint32_t)t2;
                // buckets up to ~2.5 x the mean load stay "light": their item sums are added up by the leaf level's lanes
                heavy_items = (uint32_t)std::max<size_t>(HEAVY_ITEMS, (5 * (entries / P.nkeys + 1) / 2 + P.T - 1) / P.T);
            }
            t_ones = T_ONES;
            if (n_ones / T_ONES < 32768) t_ones = (uint32_t)std::min<size_t>(T_ONES, std::max<size_t>(8, n_ones / 32768));
            if (eager_small || n <= msm_small_n(ctx)) {
                // what counts for a small job is the number of dependent additions: T in a thread of k_accumulate, then
                // up to heavy_items in bucket_sum.  T ~ sqrt(2 * mean load); buckets up to ~2.5 x the mean stay "light".
                const double mean = (double)entries / (double)P.nkeys;
                const uint32_t t = (uint32_t)std::min(64.0, std::max(4.0, std::ceil(std::sqrt(2.0 * mean))));
                P.T = t;
                heavy_items = (uint32_t)std::max(4.0, std::ceil(2.5 * mean / t));
                // (the ones bucket's item sums ride through the reduction levels as ONE row of B entries, see phase_c: items just
                // long enough that n / t_ones + 1 of them fit the row -- 8 points where the row is long enough)
                if (eager_small) t_ones = (uint32_t)std::min<size_t>(T_ONES, std::max<size_t>(8, (n + P.B - 2) / std::max<size_t>(P.B - 1, 1)));
            }
            t_pack = P.T | (t_ones << 16);
        }
        ones_key = P.nkeys;  // one extra bucket after the kw * B regular ones
        nkeys = P.nkeys + 1;
        // bucket bits split into high (level-1 bins) and low (last level, in LDS) parts -- and for a large job a middle
        // part as well: the staged three-level partition (see k_part_staged)
        const int bbits = P.c - 1;
        const size_t max_entries = n_real * (size_t)P.nwin + n_ones;
        const bool small_job = eager_small || n <= msm_small_n(ctx);
        // (for jobs in which most points carry their nwin digits: a bit-heavy witness query -- 3 % field elements among
        // zeros and ones -- leaves the staged kernel's tiles nearly empty and is better off with the plain scatter)
        bool staged = !small_job && max_entries >= ((size_t)1 << 20) && bbits >= 7 && max_entries >= n * (size_t)P.nwin / 2;
        if (ctx->opt_sort_staged >= 0) {   // ZG_OPT_SORT_STAGED: 0 = the two-level scatter (A/B); 2 / 3 = staged for every large /
            const int v = ctx->opt_sort_staged;   // every job with at least 2^7 buckets (tests)
            staged = v == 0 ? false : v == 2 ? (!small_job && bbits >= 7) : v == 3 ? bbits >= 7 : staged;
        }
        int hb, lb, b2 = 0;
        if (staged) {
            const int b3 = 7, r = bbits - b3;
            int lgkw = 0;
            while ((1 << lgkw) < P.kw) lgkw++;
            if (((size_t)P.kw << r) <= 512) {   // the key space is small enough for two levels
                hb = r;
            } else {
                hb = std::min(r, std::max(0, 7 - lgkw));
                b2 = r - hb;
                if (b2 > 7) {
                    hb += b2 - 7;
                    b2 = 7;
                }
            }
            lb = b2 + b3;   // what level 1 leaves in an entry's low word
            if ((((size_t)P.kw << hb) + 1) > 1024) staged = false;   // (c > 23 with unfolded windows: not a plan we make)
        }
        if (!staged) {
            b2 = 0;
            hb = (bbits + 1) / 2;
            if (hb > 8) hb = 8;
            if (P.kw == 1 && bbits > 12) hb = std::min(bbits - 7, 12);   // folded windows: keep about 4096 level-1 bins
            while (hb > 0 && (((size_t)P.kw << hb) + 1) * 4 > 48 * 1024) hb--;   // level-1 LDS histogram budget
            lb = bbits - hb;
        }
        const uint32_t nbins = ((uint32_t)P.kw << hb) + 1u;   // level-1 bins (+ the ones bin)
        // (staged tiles: as many points as fill ONE round of the stage -- every workgroup then runs a single straight-line
        // round, thousands of them in flight, instead of a loop of dependent rounds)
        // (... up to 16 K tiles: beyond that the [bin][tile] table costs more than the rounds' latencies, which the many
        // workgroups in flight hide anyway)
        // (... and no more than fit the kernel's 64 KB of dynamic LDS beside the stage: [tp + 1] prefix words -- a single job
        // of 2^27 points would ask for 8192-point tiles, 67.7 KB)
        const uint32_t tp_lds_cap = (uint32_t)((64u * 1024u - PS_CAP * 8u - (4u * nbins + 17u) * 4u) / 4u) & ~255u;
        const uint32_t tp = staged ? std::min(tp_lds_cap, std::max(std::max(256u, (PS_CAP / (uint32_t)P.nwin) & ~255u),
                                                                   (uint32_t)(((n >> 14) + 255) & ~(size_t)255)))
                                   : n <= msm_small_n(ctx) ? PART_TP_SMALL : PART_TP;
        const uint32_t ntiles = (uint32_t)((n + tp - 1) / tp);
        const size_t tab_n = (size_t)nbins * ntiles;
        // last level: lb3 low bits per segment, nbins3 segments (+ the ones bin)
        const int lb3 = staged ? 7 : lb;
        const uint32_t nb1 = nbins - 1u, nbins3 = staged ? (nb1 << b2) + 1u : nbins;
        // counts is indexed by key = segment << lb3 | low bits; the ones bin only uses low bits == 0
        const size_t counts_n = ((size_t)nbins3 << lb3) + 1;
        ZG_HIP(ctx, S.counts.reserve(counts_n * 4));
        ZG_HIP(ctx, S.offsets.reserve((size_t)(nkeys + 1) * 4));
        ZG_HIP(ctx, S.item_off.reserve((size_t)(nkeys + 1) * 4));
        ZG_HIP(ctx, S.tile_tab.reserve((tab_n + 1) * 4));
        ZG_HIP(ctx, S.abuf.reserve(max_entries * 8));
        ZG_HIP(ctx, S.sorted.reserve(max_entries * 4));
        // level-2 table of the staged partition: [segment][sub-bin][chunk], sized from the bound on the entries
        const size_t maxchunks = max_entries / P2_CHUNK + nb1 + 1, n2 = staged && b2 > 0 ? maxchunks << b2 : 0;
        // (one block-sum buffer for every scan of this phase, reserved before anything is queued: growing it later would
        // free memory that queued kernels still use)
        ZG_HIP(ctx, S.bsum.reserve(std::max<size_t>(std::max<size_t>((tab_n + 1 + SCAN_TILE - 1) / SCAN_TILE, (n2 + 1 + SCAN_TILE - 1) / SCAN_TILE),
                                                    (size_t)2 * ((nkeys + 1 + SCAN_TILE - 1) / SCAN_TILE)) * 4));
        ProfScope ps(ctx, dense ? PROF_MSM_SORT_DENSE : PROF_MSM_SORT, n * 32, st);
        const size_t lds1 = (size_t)nbins * 4;
        const uint32_t vstride = use_cw ? cw_n : P.kw == 1 ? (uint32_t)fbw_stride : 0u;
        const uint32_t* rank = use_cw ? cw_rank : (const uint32_t*)nullptr;
        const uint32_t quad_n = quad_table && !has_fbw && !dense ? (quad_base ? quad_base : (uint32_t)n) : 0u;
        const int use_ones_bin = dense ? 0 : 1;   // a dense job's chained phase_c has no ones bucket (see k_part)
        ZG_LAUNCH(ctx, k_part<0>, dim3(ntiles), dim3(256), lds1, st, d_scalars, d_index, n, P.c, P.nwin, P.kw, vstride,
                           quad_n, hb, lb, ntiles, tp, S.tile_tab.as<uint32_t>(), (uint2*)nullptr, (uint32_t*)nullptr, use_ones_bin,
                           d_cls, quad_w, rank, cw_off);
        // exclusive scan of a u32 table of tn entries in place (+ the total in entry tn)
        auto xscan = [&](uint32_t* tab, size_t tn) -> int {
            const uint32_t nbx = (uint32_t)((tn + 1 + SCAN_TILE - 1) / SCAN_TILE);
            if (tn + 1 <= SCAN_ONE_MAX) {
                ZG_LAUNCH(ctx, k_xscan_one, dim3(1), dim3(SCAN_THREADS), 0, st, tab, tn);
            } else {
                ZG_LAUNCH(ctx, k_xscan_reduce, dim3(nbx), dim3(SCAN_THREADS), 0, st, tab, tn, S.bsum.as<uint32_t>());
                ZG_LAUNCH(ctx, k_xscan_bsums, dim3(1), dim3(256), 0, st, S.bsum.as<uint32_t>(), nbx);
                ZG_LAUNCH(ctx, k_xscan_final, dim3(nbx), dim3(SCAN_THREADS), 0, st, tab, tn, S.bsum.as<uint32_t>());
            }
            return ZG_OK;
        };
        ZG_TRY(xscan(S.tile_tab.as<uint32_t>(), tab_n));   // the [bin][tile] table
        // (long segments of the last level go through an LDS stage, see k_part_local; ZG_OPT_PART_STAGE_MIN: the tests
        // force it for all)
        uint32_t stage_min = 2 * PART_STAGE_TILE;
        if (ctx->opt_part_stage_min >= 0) stage_min = (uint32_t)ctx->opt_part_stage_min;
        const bool may_stage = max_entries / nbins3 * 2 >= stage_min && lb3 <= 10;   // mean segment within 2x of the threshold
        if (!may_stage) stage_min = 0xffffffffu;
        // (segments of up to PART_STAGE_TILE entries go through in one pass, see k_part_local)
        const uint32_t single_max = lb3 <= 10 ? PART_STAGE_TILE : 0u;
        const size_t lds2 = ((size_t)4 << lb3) + 64 +
                            std::max<size_t>(may_stage ? ((size_t)8 << lb3) + PART_STAGE_TILE * 6 : 0,
                                             single_max ? ((size_t)4 << lb3) + (size_t)single_max * 4 : 0);
        if (staged) {
            const size_t lds_s = ((size_t)4 * nbins + 16 + tp + 1) * 4 + (size_t)PS_CAP * 8;
            if (b2 > 0) {
                ZG_HIP(ctx, S.p2m.reserve(((size_t)2 * nb1 + 4) * 4));
                ZG_HIP(ctx, S.tab2.reserve((n2 + 1) * 4));
                ZG_HIP(ctx, S.abuf2.reserve(max_entries * 8));
            }
            ZG_LAUNCH(ctx, k_part_staged, dim3(ntiles), dim3(256), lds_s, st, d_scalars, d_index, n, P.c, P.nwin, P.kw,
                               vstride, quad_n, hb, lb, ntiles, tp, S.tile_tab.as<uint32_t>(), S.abuf.as<uint2>(),
                               S.sorted.as<uint32_t>(), use_ones_bin, d_cls, b2 > 0 ? S.p2m.as<uint32_t>() : (uint32_t*)nullptr, quad_w,
                               rank, cw_off);
            if (b2 > 0) {
                ZG_LAUNCH(ctx, k_p2<0>, dim3((unsigned)maxchunks), dim3(256), 0, st, S.abuf.as<uint2>(), S.p2m.as<uint32_t>(),
                                   nb1, b2, 7, S.tab2.as<uint32_t>(), (uint2*)nullptr);
                ZG_TRY(xscan(S.tab2.as<uint32_t>(), n2));
                ZG_LAUNCH(ctx, k_p2<1>, dim3((unsigned)maxchunks), dim3(256), 0, st, S.abuf.as<uint2>(), S.p2m.as<uint32_t>(),
                                   nb1, b2, 7, S.tab2.as<uint32_t>(), S.abuf2.as<uint2>());
                ZG_LAUNCH(ctx, k_part_local, dim3(nbins3), dim3(1024), lds2, st, S.abuf2.as<uint2>(), (const uint32_t*)nullptr,
                                   0u, nbins3, lb3, stage_min, S.counts.as<uint32_t>(), S.sorted.as<uint32_t>(),
                                   S.p2m.as<uint32_t>(), S.tab2.as<uint32_t>(), nb1, b2, single_max);
            } else {
                ZG_LAUNCH(ctx, k_part_local, dim3(nbins), dim3(1024), lds2, st, S.abuf.as<uint2>(),
                                   S.tile_tab.as<uint32_t>(), ntiles, nbins, lb3, stage_min, S.counts.as<uint32_t>(),
                                   S.sorted.as<uint32_t>(), (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, 0, single_max);
            }
        } else {
            ZG_LAUNCH(ctx, k_part<1>, dim3(ntiles), dim3(256), lds1, st, d_scalars, d_index, n, P.c, P.nwin, P.kw, vstride,
                               quad_n, hb, lb, ntiles, tp, S.tile_tab.as<uint32_t>(), S.abuf.as<uint2>(), S.sorted.as<uint32_t>(),
                               use_ones_bin, d_cls, quad_w, rank, cw_off);
            // k_part<1> advanced only its LDS copy of the offsets, so tile_tab still holds the segment starts
            ZG_LAUNCH(ctx, k_part_local, dim3(nbins), dim3(1024), lds2, st, S.abuf.as<uint2>(),
                               S.tile_tab.as<uint32_t>(), ntiles, nbins, lb, stage_min, S.counts.as<uint32_t>(),
                               S.sorted.as<uint32_t>(), (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, 0, single_max);
        }
        // bucket offsets and item offsets from the counts; nkeys + 1 entries so entry nkeys gets the totals
        const uint32_t nb = (nkeys + 1 + SCAN_TILE - 1) / SCAN_TILE;
        if ((size_t)nkeys + 1 <= SCAN_ONE_MAX) {
            ZG_LAUNCH(ctx, k_scan_one, dim3(1), dim3(SCAN_THREADS), 0, st, S.counts.as<uint32_t>(), nkeys, t_pack,
                               S.offsets.as<uint32_t>(), S.item_off.as<uint32_t>());
        } else {
            ZG_LAUNCH(ctx, k_scan_reduce, dim3(nb), dim3(SCAN_THREADS), 0, st, S.counts.as<uint32_t>(), nkeys, t_pack,
                               S.bsum.as<uint32_t>(), nb);
            ZG_LAUNCH(ctx, k_scan_bsums, dim3(1), dim3(256), 0, st, S.bsum.as<uint32_t>(), nb);
            ZG_LAUNCH(ctx, k_scan_final, dim3(nb), dim3(SCAN_THREADS), 0, st, S.counts.as<uint32_t>(), nkeys, t_pack,
                               S.bsum.as<uint32_t>(), nb, S.offsets.as<uint32_t>(), S.item_off.as<uint32_t>(),
                               (uint32_t*)nullptr);
        }
        // the totals come back to the host for the jobs that plan their last phase from them -- a job queued whole only
        // needs them for the profile's unit count
        if (!(dense || eager_small) || ctx->prof_enabled) {
            ZG_LAUNCH(ctx, k_meta, dim3(1), dim3(64), 0, st, S.offsets.as<uint32_t>(), S.item_off.as<uint32_t>(), nkeys,
                               S.meta.as<uint32_t>());
            ZG_HIP(ctx, hipMemcpyAsync(S.host, S.meta.p, 16, hipMemcpyDeviceToHost, st));
        }
        ZG_HIP(ctx, hipGetLastError());
        if (ev_sorted) ZG_HIP(ctx, hipEventRecord(ev_sorted, st));
        launched = true;
        if (dense || eager_small) {   // a known bound on the item count: queue the rest without the read-back
            // (every key's last item may be short: + nkeys; the ones bucket's items hold t_ones points)
            items_bound = (uint32_t)std::min<size_t>(max_entries / P.T + (eager_small ? n / t_ones : 0) + nkeys, 0xffffffffu);
            chained = true;
            return phase_c();
        }
        return ZG_OK;
    }

    int phase_c() override {
        if (done) return ZG_OK;
        if (chained && !items_bound) return ZG_OK;   // already queued by phase_b
        const uint32_t* meta = reinterpret_cast<const uint32_t*>(S.host);
        uint32_t total_items, ones_lo, ones_hi;
        if (src) {   // the sorter's plan and totals (run_jobs calls this once the sorter's phase_c has run on the host)
            if (src->done) {
                done = true;
                return ZG_OK;
            }
            P = src->P;
            use_cw = src->use_cw_b;
            nkeys = src->nkeys;
            ones_key = src->ones_key;
            t_ones = src->t_ones;
            t_pack = src->t_pack;
            heavy_items = src->heavy_items;
            order_items = src->order_items;
            eager_small = src->eager_small;
            total_items = src->total_items_h;
            ones_lo = src->ones_lo_h;
            ones_hi = src->ones_hi_h;
        } else {
            total_items = items_bound ? items_bound : meta[1];
            ones_lo = items_bound ? 0 : meta[2];
            ones_hi = items_bound ? 0 : meta[3];
        }
        items_bound = 0;
        if (total_items == 0) {
            done = true;
            c_launched = true;
            return ZG_OK;
        }
        MsmScratch& SS = src ? *src->Sb : S;   // whose sorted entries / offsets / items this job accumulates over
        const size_t n_sort = src ? src->n_pos : n;
        trace_point("phase_c enter");
        if (!src) ZG_HIP(ctx, S.item_key.reserve((size_t)total_items * 4));
        if (!sort_only) ZG_HIP(ctx, S.partial.reserve((size_t)total_items * sizeof(XYZZ<F>)));
        trace_point("phase_c reserved items");
        // one zeroed block: [ITEM_CLASSES] histogram, [ITEM_CLASSES] claimed per size (k_item_scatter), [2] the heavy-bucket counters
        ZG_HIP(ctx, S.ihist.reserve((2 * ITEM_CLASSES + 2) * 4));
        if (src) {
            ZG_HIP(ctx, hipMemsetAsync(S.ihist.as<uint32_t>() + 2 * ITEM_CLASSES, 0, 8, st));   // (its own heavy-bucket counters)
            ZG_HIP(ctx, hipStreamWaitEvent(st, src->ev_items, 0));
        } else {
            ProfScope ps(ctx, dense ? PROF_MSM_SORT_DENSE : PROF_MSM_SORT, 0, st);
            uint32_t* hist = S.ihist.as<uint32_t>();
            ZG_LAUNCH(ctx, k_build_items, dim3((nkeys + 255) / 256), dim3(256), 0, st, S.item_off.as<uint32_t>(),
                               nkeys, S.item_key.as<uint32_t>(), eager_small ? 1 : 0, hist, 2 * ITEM_CLASSES + 2);
            if (ones_hi > ones_lo)
                ZG_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)(S.item_key.as<uint32_t>() + ones_lo), (int)ones_key,
                                              ones_hi - ones_lo, st));
            // order the items by length -- not for a small job: three launches and a memset cost more than the lanes of
            // its few hundred waves waiting for their longest item (it is launch and latency bound, not throughput bound)
            // (`total_items` of such a job is the bound it was queued with, several times the real count)
            order_items = !eager_small && total_items >= ITEM_ORDER_MIN;
            if (order_items) {
                ZG_HIP(ctx, S.perm.reserve((size_t)total_items * 4));
                const unsigned ib = (unsigned)std::min<size_t>((total_items + 255) / 256, (size_t)ctx->num_cu * 8);
                ZG_LAUNCH(ctx, k_item_hist, dim3(ib), dim3(256), 0, st, S.item_key.as<uint32_t>(),
                                   S.item_off.as<uint32_t>(), S.offsets.as<uint32_t>(), nkeys, t_pack, ones_key, hist);
                ZG_LAUNCH(ctx, k_item_scatter, dim3(ib), dim3(256), 0, st, S.item_key.as<uint32_t>(),
                                   S.item_off.as<uint32_t>(), S.offsets.as<uint32_t>(), nkeys, t_pack, ones_key, hist,
                                   hist + ITEM_CLASSES, S.perm.as<uint32_t>());
            }
        }
        total_items_h = total_items;
        ones_lo_h = ones_lo;
        ones_hi_h = ones_hi;
        if (ev_items && !src) {   // this job's sort carries other queries too: they follow on their own streams behind ev_items
            ZG_HIP(ctx, hipGetLastError());
            ZG_HIP(ctx, hipEventRecord(ev_items, st));
            c_launched = true;
        }
        if (sort_only) return ZG_OK;
        if (leader) return ZG_OK;   // (queued by the leader of its batch, which also sets what finish() reads)
        // the batch: this job and its members
        MsmJob* bj[BATCH_MAX] = {this, members[0], members[1]};
        const unsigned nb = 1u + (unsigned)n_members;
        Batch<F> B{};
        for (unsigned k = 0; k < nb; k++) {
            MsmJob* j = bj[k];
            if (k) {   // what the member's own phase_c would have set up
                j->P = P; j->nkeys = nkeys; j->ones_key = ones_key; j->t_ones = t_ones; j->t_pack = t_pack;
                j->heavy_items = heavy_items; j->order_items = order_items; j->eager_small = eager_small;
                ZG_HIP(ctx, j->S.partial.reserve((size_t)total_items * sizeof(XYZZ<F>)));
                if (!j->S.host) ZG_HIP(ctx, hipHostMalloc((void**)&j->S.host, MsmScratch::HOST_BYTES, hipHostMallocDefault));
            }
            B.q[k].bases = j->fbw_table ? j->fbw_table : j->quad_table ? j->quad_table : j->d_bases;
            B.q[k].partial = j->S.partial.template as<XYZZ<F>>();
            B.q[k].shq = j->shq;
        }
        {
            if (dep && dep->ev_acc) ZG_HIP(ctx, hipStreamWaitEvent(st, dep->ev_acc, 0));
            ProfScope ps(ctx, IS_G1 ? (dense ? PROF_MSM_ACC_G1_DENSE : PROF_MSM_ACC_G1) : PROF_MSM_ACC_G2, n * ALG_BYTES, st,
                         chained || src ? 0 : meta[0]);  // units: mixed adds (a chained job fills them in at finish())
            acc_prof = ps.idx;
            using CL = typename ChainCfg<F>::L;
            constexpr int CLP = ChainCfg<F>::LP;
            // G2 runs on lane pairs at every size: a job of few items is a chain of dependent additions, which the pairs
            // halve; a big one is bound by throughput, and there 178 registers per lane (two waves per SIMD) beat the 329
            // of one lane per point (one wave) by more than the operand exchange costs -- dense 2^22 witness: G2
            // accumulation 18.4 -> 15.8 ms of kernel time, proof 33.5 -> 32.9 ms.
            const uint32_t* acc_perm = order_items ? SS.perm.as<uint32_t>() : (const uint32_t*)nullptr;
            // A small G1 job's accumulation is a chain of T dependent mixed additions per item on a nearly idle GPU: every
            // addition then runs on four lanes (ecd.h add_mixed_quad: four rounds of one product instead of ten serial ones).
            // Largest such launch: QUAD_ACC_MAX_LANES lanes.
            bool quad_acc = false;
            if constexpr (IS_G1) quad_acc = (eager_small || n <= msm_small_n(ctx)) && (size_t)total_items * 4 <= QUAD_ACC_MAX_LANES;
            // (SH: this job reads the entries of a shared sort through its rank map, see ShQuery)
            auto launch_acc = [&](auto kern, unsigned lanes_per_item) {
                ZG_LAUNCH(ctx, kern, dim3((unsigned)(((size_t)total_items * lanes_per_item + 255) / 256), nb), dim3(256), 0, st,
                                   B, SS.sorted.as<uint32_t>(), SS.offsets.as<uint32_t>(), SS.item_off.as<uint32_t>(),
                                   SS.item_key.as<uint32_t>(), acc_perm, nkeys, t_pack, ones_key);
            };
            // (a follower of a query over the SAME positions -- b_g1 behind b_g2 -- reads the entries as they are: they index
            // its own table, laid out like the sorter's)
            const bool sh = src && shq.nq != 0;
            if (quad_acc) {
                if constexpr (IS_G1) {
                    if (sh) launch_acc(k_accumulate<F, FqQL, 4, true>, 4);
                    else launch_acc(k_accumulate<F, FqQL, 4, false>, 4);
                }
            } else if (CLP > 1) {
                if (sh) launch_acc(k_accumulate<F, CL, CLP, true>, CLP);
                else launch_acc(k_accumulate<F, CL, CLP, false>, CLP);
            } else {
                if (sh) launch_acc(k_accumulate<F, typename LazyOf<F>::L, 1, true>, 1);
                else launch_acc(k_accumulate<F, typename LazyOf<F>::L, 1, false>, 1);
            }
            if (ev_acc) ZG_HIP(ctx, hipEventRecord(ev_acc, st));
        }
        trace_point("phase_c accumulate queued");
        ProfScope ps(ctx, dense ? PROF_MSM_REDUCE_DENSE : PROF_MSM_REDUCE, 0, st);
        // ---- bit-plane reduction over the regular keys (see k_bits_level)
        {   // fold heavy buckets first (list built on the device; the fold grid is fixed and strides over it)
            // units of HEAVY_CH items (k_heavy_collect): at most one per listed bucket plus one per HEAVY_CH items; slots for the
            // sums of the buckets of several units: at most twice the latter
            const size_t max_units = (size_t)nkeys + 1 + total_items / HEAVY_CH + 1;
            const uint32_t tmp_stride = 2 * (uint32_t)(total_items / HEAVY_CH) + 2;
            ZG_HIP(ctx, S.heavy.reserve(max_units * sizeof(uint4) + (size_t)nb * tmp_stride * 4));
            ZG_HIP(ctx, S.heavy_tmp.reserve((size_t)nb * tmp_stride * sizeof(XYZZ<F>)));
            uint32_t* hcnt = S.ihist.as<uint32_t>() + 2 * ITEM_CLASSES;   // zeroed with the histogram above
            uint4* hlist = S.heavy.as<uint4>();
            uint32_t* hdone = reinterpret_cast<uint32_t*>(hlist + max_units);
            // a job queued without read-backs folds its ones bucket here too (listed whenever it holds an item)
            // (a small job's ones bucket has at most n / t_ones + 1 items: if they fit a row of the leaf level they are
            // summed there, see k_bits_level)
            ones_in_levels = eager_small && P.c >= 2 && n_sort / t_ones + 1 <= P.B;
            const bool fold_ones = eager_small && !ones_in_levels;
            if (fold_ones) {
                for (unsigned k = 0; k < nb; k++) {
                    ZG_HIP(ctx, bj[k]->S.ones[0].reserve(sizeof(XYZZ<F>)));
                    B.q[k].ones_out = bj[k]->S.ones[0].template as<XYZZ<F>>();
                    ZG_HIP(ctx, hipMemsetAsync(B.q[k].ones_out, 0, sizeof(XYZZ<F>), st));   // all zero = the identity
                }
            }
            const uint32_t listed = fold_ones ? nkeys : P.nkeys;
            // (the list of heavy buckets depends on the items only: one for the whole batch)
            ZG_LAUNCH(ctx, k_heavy_collect, dim3((listed + 255) / 256), dim3(256), 0, st, SS.item_off.as<uint32_t>(),
                               listed, heavy_items, fold_ones ? ones_key : ~0u, hcnt, hlist, hdone, nb, tmp_stride);
            ZG_LAUNCH(ctx, (k_heavy_fold<F, typename ChainCfg<F>::L, ChainCfg<F>::LP>), dim3(1024, nb), dim3(128 * ChainCfg<F>::LP),
                               128 * sizeof(XYZZ<F>), st, B, SS.item_off.as<uint32_t>(), hcnt, hlist, fold_ones ? ones_key : ~0u,
                               S.heavy_tmp.template as<XYZZ<F>>(), hdone, tmp_stride);
        }
        const int bits = P.c - 1;
        red_nlev = 0;
        for (int left = bits; left > 0;) {
            red_f[red_nlev] = left >= 2 ? 2 : 1;
            left -= red_f[red_nlev++];
        }
        // levels: every row is folded 2^f : 1, S rows append their bit-plane rows (see k_bits_level)
        const uint32_t xrow = ones_in_levels ? 1u : 0u;   // the ones row, right after the S rows
        for (unsigned q = 0; q < nb; q++)
            for (int k = 0; k < 2; k++) ZG_HIP(ctx, bj[q]->S.lvl[k].reserve(((size_t)P.nkeys + (size_t)xrow * P.B) * sizeof(XYZZ<F>)));
        const XYZZ<F>* cur = nullptr;
        {
            uint32_t len = P.B, rows = (uint32_t)P.kw + xrow;
            size_t rb = 0;
            int bb = 0;
            int tail_from = -1;   // first level that k_bits_tail runs (-1: not decided yet, -2: no tail)
            for (int l = 0; l < red_nlev; l++) {
                len >>= red_f[l];
                const size_t groups = (size_t)rows * len;
                XYZZ<F>* outp = S.lvl[l & 1].template as<XYZZ<F>>();
                for (unsigned q = 0; q < nb; q++) {
                    B.q[q].lvl_in = l == 0 ? (const XYZZ<F>*)nullptr : bj[q]->S.lvl[(l - 1) & 1].template as<XYZZ<F>>();
                    B.q[q].lvl_out = bj[q]->S.lvl[l & 1].template as<XYZZ<F>>();
                }
                // a level of few groups is a chain of dependent additions with most of the GPU idle: G1 then runs every
                // addition on four lanes (ecd.h FqQL; QUAD_LEVEL_MAX_LANES: largest such level in logical lanes).
                // (The same on four lane pairs for G2 -- eight lanes, ds_swizzle broadcasts -- was measured: no gain.)
                bool quad = false;
                if constexpr (IS_G1) quad = groups * 2 <= QUAD_LEVEL_MAX_LANES;
                // the tail: from the first level on (never the leaf) behind which every level has at most TAIL_LANES_MAX lane
                // groups, if there are at least two of them -- one workgroup runs them all (k_bits_tail)
                if (tail_from < 0 && l >= 1 && red_nlev - l >= 2 && red_nlev - l <= TAIL_LEVELS_MAX && groups * 2 <= TAIL_LANES_MAX) {
                    tail_from = l;   // (rows x len only shrinks from here on: len falls 4 : 1 while rows grows by 2 kw)
                    uint32_t tl_len = len, tl_rows = rows;
                    TailLevels TL{};
                    for (int k = l; k < red_nlev; k++) {
                        if (k > l) tl_len >>= red_f[k];
                        if ((size_t)tl_rows * tl_len * 2 > TAIL_LANES_MAX) tail_from = -2;   // (does not shrink after all: no tail)
                        TL.rows_in[k - l] = tl_rows;
                        TL.len_out[k - l] = tl_len;
                        TL.f[k - l] = red_f[k];
                        tl_rows += (uint32_t)P.kw * red_f[k];
                    }
                    TL.n = red_nlev - l;
                    if (tail_from == l) {
                        for (unsigned q = 0; q < nb; q++) {
                            B.q[q].lvl_in = bj[q]->S.lvl[(l - 1) & 1].template as<XYZZ<F>>();
                            B.q[q].lvl_out = bj[q]->S.lvl[l & 1].template as<XYZZ<F>>();
                        }
                        if (quad) {
                            if constexpr (IS_G1)
                                ZG_LAUNCH(ctx, (k_bits_tail<F, FqQL, 4>), dim3(1, nb), dim3(BITS_THREADS * 4), BITS_THREADS * sizeof(XYZZ<F>), st, B,
                                          SS.item_off.as<uint32_t>(), TL, (uint32_t)P.kw, heavy_items, ones_key);
                        } else {
                            ZG_LAUNCH(ctx, (k_bits_tail<F, typename ChainCfg<F>::L, ChainCfg<F>::LP>), dim3(1, nb),
                                      dim3(BITS_THREADS * ChainCfg<F>::LP), BITS_THREADS * sizeof(XYZZ<F>), st, B,
                                      SS.item_off.as<uint32_t>(), TL, (uint32_t)P.kw, heavy_items, ones_key);
                        }
                    }
                }
                if (tail_from >= 0 && l >= tail_from) {
                    // (queued with the tail above)
                } else if (quad) {
                    if constexpr (IS_G1)
                        ZG_LAUNCH(ctx, (k_bits_level<F, FqQL, 4>), dim3((unsigned)((groups * 2 + BITS_THREADS - 1) / BITS_THREADS), nb),
                                           dim3(BITS_THREADS * 4), BITS_THREADS * sizeof(XYZZ<F>), st, B,
                                           SS.item_off.as<uint32_t>(), rows, (uint32_t)P.kw, len, red_f[l], heavy_items,
                                           l == 0 && xrow ? (uint32_t)P.kw : ~0u, ones_key);
                } else {
                    ZG_LAUNCH(ctx, (k_bits_level<F, typename ChainCfg<F>::L, ChainCfg<F>::LP>),
                                       dim3((unsigned)((groups * 2 + BITS_THREADS - 1) / BITS_THREADS), nb),
                                       dim3(BITS_THREADS * ChainCfg<F>::LP), BITS_THREADS * sizeof(XYZZ<F>), st, B,
                                       SS.item_off.as<uint32_t>(), rows, (uint32_t)P.kw, len, red_f[l],
                                       heavy_items, l == 0 && xrow ? (uint32_t)P.kw : ~0u, ones_key);
                }
                cur = outp;
                red_rowbase[l] = (uint32_t)rb;   // finish(): rows of level l start here (after the kw S rows)
                red_bitbase[l] = bb;
                rb += (size_t)P.kw * red_f[l];
                bb += red_f[l];
                rows += (uint32_t)P.kw * red_f[l];
            }
        }
        trace_point("phase_c levels queued");
        // cur: [kw] S, then one entry per bit-plane row, rows ordered [level][window][bit within level]
        // ---- ones bucket: its item sums form one row
        ones_ptr = nullptr;
        if (ones_in_levels) {
            ones_ptr = cur + P.kw;                          // the ones row's sum, right after the S entries
        } else if (eager_small) {
            ones_ptr = S.ones[0].template as<XYZZ<F>>();   // written by k_heavy_fold above
        } else if (ones_hi > ones_lo) {
            uint32_t cnt = ones_hi - ones_lo;
            const XYZZ<F>* in = S.partial.as<XYZZ<F>>() + ones_lo;
            int pp = 0;
            while (cnt > 1) {
                RowMeta O{};
                O.nlev = 1;
                O.len[0] = cnt;
                O.rows[0] = 1;
                O.chunks[0] = (cnt + ROWSUM_CH - 1) / ROWSUM_CH;
                ZG_HIP(ctx, S.ones[pp].reserve((size_t)O.chunks[0] * sizeof(XYZZ<F>)));
                ZG_LAUNCH(ctx, (k_rowsum<F, typename ChainCfg<F>::L, ChainCfg<F>::LP>), dim3(O.chunks[0]),
                                   dim3(256 * ChainCfg<F>::LP), 256 * sizeof(XYZZ<F>), st, in, O,
                                   S.ones[pp].as<XYZZ<F>>());
                in = S.ones[pp].template as<XYZZ<F>>();
                cnt = O.chunks[0];
                pp ^= 1;
            }
            ones_ptr = in;
        }
        trace_point("phase_c rowsums queued");
        // read-back layout: [kw] S, [kw * bits] T in (level, window, bit) order, [1] ones
        // (with the ones row among the levels: [kw] S, [1] ones, [kw * bits] T -- one copy)
        const size_t n_t = (size_t)P.kw * bits;
        if (((size_t)P.kw + n_t + 1) * sizeof(XYZZ<F>) + 64 > MsmScratch::HOST_BYTES)
            return fail(ctx, ZG_E_INVALID_ARG, "msm: read-back of %zu window sums exceeds the pinned buffer", n_t);
        XYZZ<F>* hw = reinterpret_cast<XYZZ<F>*>(S.host + 64);   // the first 64 bytes keep the meta words
        ZG_HIP(ctx, hipMemcpyAsync(hw, cur, ((size_t)P.kw + xrow + n_t) * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
        if (ones_ptr && !ones_in_levels)
            ZG_HIP(ctx, hipMemcpyAsync(hw + P.kw + n_t, ones_ptr, sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
        for (unsigned q = 1; q < nb; q++) {   // (members exist in small-job mode only: the ones are in the levels or folded)
            MsmJob* j = bj[q];
            j->red_nlev = red_nlev;
            for (int l = 0; l < red_nlev; l++) {
                j->red_f[l] = red_f[l];
                j->red_bitbase[l] = red_bitbase[l];
                j->red_rowbase[l] = red_rowbase[l];
            }
            j->ones_in_levels = ones_in_levels;
            const XYZZ<F>* jcur = j->S.lvl[(red_nlev - 1) & 1].template as<XYZZ<F>>();
            j->ones_ptr = ones_in_levels ? jcur + P.kw : j->S.ones[0].template as<XYZZ<F>>();
            XYZZ<F>* jhw = reinterpret_cast<XYZZ<F>*>(j->S.host + 64);
            ZG_HIP(ctx, hipMemcpyAsync(jhw, jcur, ((size_t)P.kw + xrow + n_t) * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
            if (!ones_in_levels)
                ZG_HIP(ctx, hipMemcpyAsync(jhw + P.kw + n_t, j->ones_ptr, sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
        }
        ZG_HIP(ctx, hipGetLastError());
        return ZG_OK;
    }

    int finish() override {
        if (done) return ZG_OK;
        if (sort_only) {
            done = true;
            return ZG_OK;
        }
        const XYZZ<F>* hw = reinterpret_cast<const XYZZ<F>*>(S.host + 64);
        if (chained && acc_prof >= 0 && !src) {
            std::lock_guard<std::mutex> lock(ctx->prof_mu);
            if (acc_prof < (int)ctx->prof_pending.size())
                ctx->prof_pending[acc_prof].units = reinterpret_cast<const uint32_t*>(S.host)[0];
        }
        XYZZ<F> acc = XYZZ<F>::inf();
        // the device works in the 2^261 Montgomery domain; window sums come back reduce()d in that form
        // window value = S_w + sum_j 2^j T_{w,j}; one Horner pass over all bit positions
        const int bits = P.c - 1;
        const XYZZ<F>* ht = hw + P.kw + (ones_in_levels ? 1 : 0);
        for (int w = P.kw - 1; w >= 0; w--) {
            for (int j = P.c - 1; j >= 0; j--) {
                acc = acc.dbl();
                if (j < bits) {
                    int l = 0;
                    while (l + 1 < red_nlev && red_bitbase[l + 1] <= j) l++;
                    const XYZZ<F>& t = ht[red_rowbase[l] + (size_t)w * red_f[l] + (j - red_bitbase[l])];
                    if (!t.is_inf()) acc = acc.add(m261_to_host(t));
                }
            }
            if (!hw[w].is_inf()) acc = acc.add(m261_to_host(hw[w]));
        }
        const XYZZ<F>& ho = ones_in_levels ? hw[P.kw] : hw[P.kw + (size_t)P.kw * bits];
        if (ones_ptr && !ho.is_inf()) acc = acc.add(m261_to_host(ho));
        *out = acc;
        done = true;
        return ZG_OK;
    }
};

// Advance every job as soon as ITS stream has drained (hipStreamQuery), so jobs do not wait for each
// other's read-backs: a job whose stream is gated on an event (the h MSM waits for the NTT pipeline)
// simply stays in its first phase while the others run ahead.
// (One host thread per job with blocking waits was measured: 0.6 ms slower per 2^22 proof -- the runtime's launch
// path serialises the threads, the waits wake late, and the witness accumulations then crowd the first NTT passes.)
// on_witness: called once, by the thread that finished the last job marked `witness`, if all of them succeeded
// The host parts of the jobs (finish(): a Horner pass of a few dozen point operations) run beside the polling loop on two
// helper threads that live as long as the context.  They used to be std::async calls: a thread spawned per job and proof,
// ~20 us each, one after the other at the very end of a small proof, where its five chains finish within 0.1 ms of each
// other (0.12 ms of a 0.94 ms proof at 2^14 between the last kernel and the last pass).  While a batch of jobs is in flight
// the helpers poll their queue (a proof lasts a millisecond or ten); otherwise they sleep on a condition variable.
struct FinishPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::packaged_task<int()>> q;
    std::vector<std::thread> workers;
    std::atomic<int> hot{0};   // run_jobs calls in flight: the workers spin instead of sleeping
    bool stop = false;
    FinishPool() {
        for (int k = 0; k < 2; k++) workers.emplace_back([this] { run(); });
    }
    ~FinishPool() {
        {
            std::lock_guard<std::mutex> lock(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : workers) t.join();
    }
    void run() {
        for (;;) {
            std::packaged_task<int()> task;
            {
                std::unique_lock<std::mutex> lock(mu);
                while (q.empty() && !stop) {
                    if (hot.load(std::memory_order_relaxed) > 0) {
                        lock.unlock();
                        std::this_thread::yield();
                        lock.lock();
                    } else {
                        cv.wait(lock);
                    }
                }
                if (q.empty()) return;   // stop
                task = std::move(q.front());
                q.pop_front();
            }
            task();
        }
    }
    std::future<int> submit(std::function<int()> fn) {
        std::packaged_task<int()> task(std::move(fn));
        std::future<int> f = task.get_future();
        {
            std::lock_guard<std::mutex> lock(mu);
            q.push_back(std::move(task));
        }
        cv.notify_one();
        return f;
    }
};
static FinishPool& finish_pool(Ctx* ctx) {
    if (!ctx->finish_pool) ctx->finish_pool = new FinishPool();
    return *ctx->finish_pool;
}

static int run_jobs(Ctx* ctx, MsmJobBase** jobs, int njobs, const std::function<void()>& on_witness = {}) {
    int state[16] = {0};  // 0: launch a, 1: a->b, 2: b->c, 3: c->finish, 4: done
    int remaining = njobs;
    std::atomic<int> w_left{0};
    std::atomic<bool> w_ok{true};
    for (int j = 0; j < njobs; j++) w_left += jobs[j]->witness ? 1 : 0;
    auto finish_job = [&w_left, &w_ok, &on_witness](MsmJobBase* jb) {
        const int rc = jb->finish();
        if (jb->witness) {
            if (rc != ZG_OK) w_ok.store(false);
            if (w_left.fetch_sub(1) == 1 && w_ok.load() && on_witness) on_witness();
        }
        return rc;
    };
    // (every future is waited for before this function returns, whatever way it ends: the tasks refer to its locals)
    struct Finishing {
        std::vector<std::future<int>> f;
        FinishPool* pool = nullptr;
        ~Finishing() {
            for (auto& x : f)
                if (x.valid()) x.wait();
            if (pool) pool->hot.fetch_sub(1);
        }
    } finishing_guard;
    std::vector<std::future<int>>& finishing = finishing_guard.f;
    if (njobs > 1) {
        finishing_guard.pool = &finish_pool(ctx);
        finishing_guard.pool->hot.fetch_add(1);
        finishing_guard.pool->cv.notify_all();
    }
    // ZA_AMD_TRACE=1: host timestamps (ms since entry) of every phase transition of every job, to stderr
    const bool trace = trace_level() != 0;
    const auto t_entry = std::chrono::steady_clock::now();
    auto stamp = [&](int j, const char* what, double t_begin) {
        if (!trace) return;
        const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count();
        fprintf(stderr, "[za_amd trace] job %d %-8s host %.3f -> %.3f ms\n", j, what, t_begin, now);
    };
    auto tnow = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count(); };
    // (trace only) GPU-side begin / end of every job queued whole, relative to the end of what the main stream holds on entry
    hipEvent_t tev[16][2] = {{nullptr}}, tbase = nullptr;
    if (trace) {
        (void)hipEventCreate(&tbase);
        (void)hipEventRecord(tbase, ctx->stream);
        for (int j = 0; j < njobs; j++)
            for (int k = 0; k < 2; k++) (void)hipEventCreate(&tev[j][k]);
    }
    while (remaining > 0) {
        bool progressed = false;
        // launches first: a job's host-side finish() (a few hundred point operations) must not delay another
        // job's next phase, so it only runs once every job has its last device phase queued
        bool launches_pending = false;
        for (int j = 0; j < njobs; j++) launches_pending |= state[j] < 3;
        for (int j = 0; j < njobs; j++) {
            if (state[j] == 4) continue;
            if (state[j] == 0) {
                const double tb = tnow();
                ZG_TRY(jobs[j]->phase_a());
                stamp(j, "phase_a", tb);
                state[j] = 1;
                progressed = true;
                // A dense job's whole chain is known on the host: queue it now, behind the stream's event gate, so the
                // GPU starts it the moment the NTT pipeline is done -- not when the host gets round to it (a 2^14
                // proof waited 0.9 ms here while the host was queueing the witness jobs' phases).
                if (jobs[j]->eager && (!jobs[j]->dep || jobs[j]->dep->launched) && (!jobs[j]->src || jobs[j]->src->launched)) {
                    const double tb2 = tnow();
                    if (trace) (void)hipEventRecord(tev[j][0], jobs[j]->st);
                    ZG_TRY(jobs[j]->phase_b());
                    if (trace) (void)hipEventRecord(tev[j][1], jobs[j]->st);
                    stamp(j, "phase_b+c", tb2);
                    state[j] = jobs[j]->chained ? 3 : 2;
                }
                continue;
            }
            if (state[j] == 3 && launches_pending) continue;
            if (state[j] == 1 && jobs[j]->dep && !jobs[j]->dep->launched) continue;   // parts launch in order
            // a follower of a shared sort moves when the sorter has: phase_b behind its phase_b, phase_c behind its phase_c
            if (state[j] == 1 && jobs[j]->src && !jobs[j]->src->launched) continue;
            if (state[j] == 2 && jobs[j]->src && !(jobs[j]->src->c_launched || jobs[j]->src->done)) continue;
            hipError_t q = hipStreamQuery(jobs[j]->st);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) return fail(ctx, ZG_E_HIP, "stream query failed: %s", hipGetErrorString(q));
            const double tb = tnow();
            if (state[j] == 1) {
                ZG_TRY(jobs[j]->phase_b());
                stamp(j, "phase_b", tb);
                if (jobs[j]->chained) state[j]++;   // nothing left to launch
            } else if (state[j] == 2) {
                ZG_TRY(jobs[j]->phase_c());
                stamp(j, "phase_c", tb);
            } else {
                // The host part of a job (its Horner pass: ~250 doublings and additions) runs beside the polling loop
                // when other jobs are still in flight -- a small proof or a shard otherwise ends with the jobs'
                // host passes one after the other (0.35 ms each at 2^14).
                if (remaining > 1) {
                    MsmJobBase* jb = jobs[j];
                    finishing.push_back(finish_pool(ctx).submit([jb, &finish_job] { return finish_job(jb); }));
                } else {
                    ZG_TRY(finish_job(jobs[j]));
                }
                stamp(j, "finish", tb);
                remaining--;
            }
            state[j]++;
            progressed = true;
            break;   // rescan from the front: the jobs are listed by priority (the h parts first)
        }
        if (!progressed) std::this_thread::yield();
    }
    if (trace) {
        for (int j = 0; j < njobs; j++) {
            float a = -1, b = -1;
            if (hipEventElapsedTime(&a, tbase, tev[j][0]) == hipSuccess && hipEventElapsedTime(&b, tbase, tev[j][1]) == hipSuccess)
                fprintf(stderr, "[za_amd trace] job %d on the GPU %.3f -> %.3f ms\n", j, a, b);
            for (int k = 0; k < 2; k++) (void)hipEventDestroy(tev[j][k]);
        }
        (void)hipGetLastError();
        (void)hipEventDestroy(tbase);
    }
    int rc = ZG_OK;
    for (auto& f : finishing) {
        const int r = f.get();
        if (rc == ZG_OK) rc = r;
    }
    return rc;
}

static MsmScratch& scratch_slot(Ctx* ctx, int slot) {
    if (!ctx->msm_scratch[slot]) ctx->msm_scratch[slot] = new MsmScratch();
    return *ctx->msm_scratch[slot];
}

// Device bytes a slot's (grow-only) scratch holds, and their release.  The l parts behind a chunked witness (msm_batch5) sort on
// the slots further h parts would use: two more sets of sort buffers of a third of the query each.  At 2^26 that is 11 GB, kept
// beside an unsplit l job's 16 GB if proofs of both kinds alternate -- more than a key of that size leaves free.  So the split
// is only taken while the memory is there (msm_scratch_bytes, prover.hip witness_handover) and an unsplit proof drops what
// the parts left behind (msm_scratch_trim).
template <class Fn>
static void scratch_bufs(MsmScratch& S, Fn fn) {
    DevBuf* all[] = {&S.counts, &S.offsets, &S.item_off, &S.tile_tab, &S.abuf, &S.sorted, &S.item_key, &S.perm, &S.ihist,
                     &S.partial, &S.lvl[0], &S.lvl[1], &S.ones[0], &S.ones[1], &S.bsum, &S.meta, &S.heavy, &S.heavy_tmp,
                     &S.abuf2, &S.tab2, &S.p2m};
    for (DevBuf* b : all) fn(*b);
}
size_t msm_scratch_bytes(Ctx* ctx, int slot) {
    if (slot < 0 || slot > Ctx::MSM_SLOTS || !ctx->msm_scratch[slot]) return 0;
    size_t total = 0;
    scratch_bufs(*ctx->msm_scratch[slot], [&](DevBuf& b) { total += b.bytes; });
    return total;
}
static void msm_scratch_trim(Ctx* ctx, int slot) {
    if (!ctx->msm_scratch[slot]) return;
    scratch_bufs(*ctx->msm_scratch[slot], [&](DevBuf& b) { b.release(); });
}

int msm_g1(Ctx* ctx, const G1Affine* d_bases, const uint32_t* d_scalars, const uint32_t* d_index, size_t n,
           G1XYZZ* out) {
    MsmJob<Fq> job(ctx, scratch_slot(ctx, 0), ctx->stream, d_bases, d_scalars, d_index, n, out);
    MsmJobBase* jobs[1] = {&job};
    return run_jobs(ctx, jobs, 1);
}
int msm_g2(Ctx* ctx, const G2Affine* d_bases, const uint32_t* d_scalars, const uint32_t* d_index, size_t n,
           G2XYZZ* out) {
    MsmJob<Fq2> job(ctx, scratch_slot(ctx, 0), ctx->stream, d_bases, d_scalars, d_index, n, out);
    MsmJobBase* jobs[1] = {&job};
    return run_jobs(ctx, jobs, 1);
}

// The five multiexps of create_proof, concurrently on the ctx's side streams.  The witness MSMs
// (l, a, b_g1, b_g2) are gated on `ev_witness`, the h MSM on `ev_h` (both recorded by the caller on
// ctx->stream; NULL = whatever is already queued there).  The witness jobs run at high stream
// priority: their big-register G2 / tree kernels would otherwise starve behind the h accumulation.
// sub-range [lo, hi) of a query as a job of its own (lo a multiple of 4 when the query has a subset-sum table)
template <class F>
static std::unique_ptr<MsmJob<F>> make_range_job(Ctx* ctx, int slot, const MsmDesc& d, size_t lo, size_t hi, XYZZ<F>* out) {
    const uint32_t* sc = d.scalars ? (d.index ? d.scalars : d.scalars + 8 * lo) : nullptr;
    const uint32_t* ix = d.index ? d.index + lo : nullptr;
    std::unique_ptr<MsmJob<F>> j(new MsmJob<F>(ctx, scratch_slot(ctx, slot), ctx->side[slot], (const Affine<F>*)d.bases + lo, sc, ix,
                                               hi - lo, out));
    j->dense = d.dense;
    j->eager_small = !d.dense && (d.small_hint >= 0 ? d.small_hint != 0 : hi - lo <= msm_small_n(ctx));
    j->d_cls = d.cls ? (d.index ? d.cls : d.cls + lo) : nullptr;
    if (d.quad_table) {   // (lo is a multiple of the group width)
        j->quad_table = (const Affine<F>*)d.quad_table + lo;
        j->quad_w = d.quad_w;
        // the sums start at entry n of the table, 2^qw - 1 per group of qw positions
        j->quad_base = (uint32_t)(d.n - lo + (((size_t)1 << d.quad_w) - 1) * (lo / (size_t)d.quad_w));
        if (d.cw_rank && lo == 0 && hi == d.n) {   // (the compact rows are addressed from the table's start: whole-query jobs)
            j->cw_rank = d.cw_rank;
            j->cw_off = d.cw_off;
            j->cw_n = d.cw_n;
            j->cw_c = d.cw_c;
            j->cw_nwin = d.cw_nwin;
        }
    }
    if (d.fbw_table) {
        j->fbw_table = (const Affine<F>*)d.fbw_table + lo;   // row w of the range starts at w * stride + lo
        j->fbw_c = d.fbw_c;
        j->fbw_nwin = d.fbw_nwin;
        j->fbw_stride = d.fbw_stride;
    }
    return j;
}

// The side streams of a context (one per slot) and the events between them.  Creating them takes 70-80 ms (eight streams at
// ~10 ms each): zg_init starts it on a helper thread (msm_side_streams_begin), so that it runs beside whatever the caller
// does first -- reading a key, mostly -- instead of inside its first proof (one prove() of the outer boundary: 0.11 s of
// "proof" of which 0.08 were this).
// All side streams at the default priority.  Until round 4 the witness streams of a mostly-0/1 assignment ran at
// high priority (short jobs beside the transforms).  With the stream assignment of msm_batch5 that is worth nothing for a
// lone context (2^22, four alternated pairs of processes: 11.60 / 11.84 / 11.95 / 11.85 with, 11.08 / 11.53 / 11.81 /
// 10.87 without) and with three contexts on the GPU it produced proofs of 60-470 ms every few runs -- mixed
// priorities over 27 streams on 8 hardware queues; equal priorities: 10.0-10.1 ms per proof, every run
// (profiles/r04_exp_stream_assignment.txt).
// (Confining the witness streams to 16-128 of the 256 CUs with hipExtStreamCreateWithCUMask: measured, no gain --
// they need few CUs; what they cost the NTT pipeline is memory passes and launches, not compute units.)
static bool side_streams_create(Ctx* ctx) {   // (may run on a helper thread: reports nothing, leaves nothing behind on failure)
    if (hipSetDevice(ctx->device) != hipSuccess) return false;
    int lo_prio = 0, hi_prio = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);   // numerically lower = higher priority
    bool ok = true;
    for (int k = 0; k < Ctx::MSM_SLOTS && ok; k++)
        ok = hipStreamCreateWithPriority(&ctx->side[k], hipStreamNonBlocking, lo_prio) == hipSuccess;
    hipEvent_t* evs[3] = {&ctx->side_ev, &ctx->share_ev, &ctx->follow_ev};
    for (hipEvent_t* e : evs) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < 2 * Ctx::H_PARTS_MAX && ok; k++) ok = hipEventCreateWithFlags(&ctx->part_ev[k], hipEventDisableTiming) == hipSuccess;
    if (ok) return true;
    for (int k = 0; k < Ctx::MSM_SLOTS; k++)
        if (ctx->side[k]) {
            (void)hipStreamDestroy(ctx->side[k]);
            ctx->side[k] = nullptr;
        }
    for (hipEvent_t* e : evs)
        if (*e) {
            (void)hipEventDestroy(*e);
            *e = nullptr;
        }
    for (int k = 0; k < 2 * Ctx::H_PARTS_MAX; k++)
        if (ctx->part_ev[k]) {
            (void)hipEventDestroy(ctx->part_ev[k]);
            ctx->part_ev[k] = nullptr;
        }
    return false;
}
void msm_side_streams_begin(Ctx* ctx) {
    if (ctx->side_ready || ctx->side_future.valid()) return;
    ctx->side_future = std::async(std::launch::async, [ctx] { return side_streams_create(ctx); });
}
int msm_side_streams_ready(Ctx* ctx) {
    if (ctx->side_ready) return ZG_OK;
    const auto t_streams = std::chrono::steady_clock::now();
    bool ok = ctx->side_future.valid() && ctx->side_future.get();
    if (!ok) ok = side_streams_create(ctx);   // (not started, or it failed on the helper thread: once more, here, with the error reported)
    if (!ok) return fail(ctx, ZG_E_HIP, "creating the multiexp streams failed: %s", hipGetErrorString(hipGetLastError()));
    ctx->side_ready = true;
    if (trace_level() != 0)
        fprintf(stderr, "[za_amd trace] side streams and events ready after  %.3f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_streams).count());
    return ZG_OK;
}
int msm_batch5(Ctx* ctx, const MsmDesc g1[4], const MsmDesc& g2, G1XYZZ out1[4], G2XYZZ* out2, hipEvent_t ev_witness,
               hipEvent_t ev_h, const MsmShare* share) {
    constexpr int NS = Ctx::MSM_SLOTS, HP = Ctx::H_PARTS_MAX;
    // slots: 0-4 the five queries (job_slot below) | 5-7 further h parts
    // Which of the side streams 0-4 each of h (its first part), l, a, b_g1, b_g2 runs on.  HIP streams are backed by hardware
    // queues in creation order, the command processor serves the queues through a few pipes, and two queues that meet on
    // one pipe take TURNS: a chain of small dependent kernels on one of them runs between the kernels of the other, not
    // beside them (tools/probe.py small-chains: the G2 chain and the h chain of a 7 469-constraint proof, 0.64 and 0.30 ms
    // alone, took 0.95 ms together on side streams 4 and 0 and 0.68 ms on 3 and 0).  Which streams meet is not documented,
    // so the assignment was searched: all 120 of them, one process each, on the EdDSA circuit, at 2^14 and at 2^20
    // (profiles/r04_exp_stream_assignment.txt).  This one is within 1 % of the best at
    // each size: 2^14 0.97 -> 0.92 ms, EdDSA 1.04 -> 0.97, 2^16 1.79 -> 1.53, 2^18 2.53 -> 2.40, 2^20 4.33 -> 4.13, 2^22
    // unchanged; the worst ones (h on stream 1, or the G2 chain on stream 2) cost a small proof 30 %.
    // The streams are interchangeable (same priority, scratch memory of any kind), so the assignment is made per call: the
    // searched one for a whole proof of a lone context; streams 0-4 in order when the h multiexp or the witness multiexps
    // run on their own (a rank of a sharded proof: its four witness shards alone 1.28 ms in order, 1.62 ms the other way)
    // and when several contexts share the GPU (three provers: 9.7 against 10.3 ms per proof).
    static const std::array<int, 5> in_order = {0, 1, 2, 3, 4}, whole_proof = {3, 2, 4, 1, 0};
    const bool whole = g1[0].n != 0 && (g1[1].n != 0 || g1[2].n != 0 || g1[3].n != 0 || g2.n != 0) && live_contexts() == 1;
    const std::array<int, 5>& job_slot = whole ? whole_proof : in_order;
    const int h_slot0 = job_slot[0];
    auto is_h_slot = [h_slot0](int k) { return k == h_slot0 || (k >= 5 && k <= 7); };
    ZG_TRY(msm_side_streams_ready(ctx));
    // The h job is split into parts over consecutive base ranges (see MsmJobBase::dep): part 0 on side[0], the
    // others on side[5..].  Two parts take 0.33 ms off a 2^22 proof (three: +0.1 ms); when several contexts share the
    // GPU their proofs already fill each other's sort and reduction phases and the split costs 0.37 ms, so it is
    // only used by a lone context.  ZG_OPT_H_PARTS overrides the count (1 = one job).
    int parts = 1;
    // (round 5: from 2^22 points -- at 2^21 one part measured 0.1 ms faster, at 2^20 and 2^22 the same)
    if (g1[0].dense && g1[0].n >= ((size_t)1 << 22) - 1 && live_contexts() == 1) parts = 2;
    if (g1[0].parts > 0) parts = g1[0].parts;
    const int gates_n = g1[0].part_gate ? std::max(g1[0].parts, 1) : 0;
    // (not when the caller gates the parts on events of its own -- zg_msm_resident's scalar chunks: its part count IS
    // the chunking, and a different split would read scalars that have not arrived yet)
    if (ctx->opt_h_parts > 0 && !g1[0].part_gate) parts = ctx->opt_h_parts;
    if (parts < 1) parts = 1;
    if (parts > HP) parts = HP;
    if (g1[0].n < (size_t)parts * 4096) parts = 1;
    // witness queries
    const MsmDesc* wq[4] = {&g1[1], &g1[2], &g1[3], &g2};
    const int* w_slot = &job_slot[1];
    // The l multiexp in parts behind the chunks of a witness that is still arriving (MsmDesc::part_gate, prover.hip
    // witness_handover): part 0 on the l query's own side stream, the others on the streams that further h parts would take --
    // so only while those are free, and not for a query that reads a shared sort.  Each part is a job of its own over a
    // range of the query (make_range_job); their sums are added before the witness hook sees them.
    int lparts = 1, lslot[HP] = {w_slot[0], 0, 0, 0};
    if (g1[1].part_gate && g1[1].part_gate_only && g1[1].part_bounds && g1[1].parts > 1 && g1[1].parts <= HP && g1[1].n &&
        !(share && (share->mask & 2u)) && g1[1].parts - 1 <= 3 - (parts - 1)) {
        lparts = g1[1].parts;
        for (int k = 1; k < lparts; k++) lslot[k] = 4 + parts + (k - 1);   // the slots after the h parts': 5 .. 7
    }
    if (lparts == 1)
        for (int k = 4 + parts; k <= 7; k++)   // (nothing of this proof is queued on these slots)
            if (msm_scratch_bytes(ctx, k) > ((size_t)1 << 30)) msm_scratch_trim(ctx, k);
    auto is_l_part_slot = [&](int k) {
        for (int q = 0; q < lparts; q++)
            if (lparts > 1 && lslot[q] == k) return true;
        return false;
    };
    ZG_HIP(ctx, hipEventRecord(ctx->side_ev, ctx->stream));
    for (int k = 0; k < NS; k++) {
        if (is_l_part_slot(k)) continue;   // gated per part below
        if (k >= 5 && k <= 7 && k - 4 >= parts) continue;
        hipEvent_t ev = is_h_slot(k) ? ev_h : ev_witness;
        if (is_h_slot(k) && g1[0].part_gate && g1[0].part_gate_only) continue;   // gated per part below
        ZG_HIP(ctx, hipStreamWaitEvent(ctx->side[k], ev ? ev : ctx->side_ev, 0));
    }
    for (int q = 0; q < 4; q++)
        if (wq[q]->gate && wq[q]->n) ZG_HIP(ctx, hipStreamWaitEvent(ctx->side[w_slot[q]], wq[q]->gate, 0));
    std::unique_ptr<MsmJob<Fq>> wj1[3];
    std::unique_ptr<MsmJob<Fq2>> wj2;
    std::unique_ptr<MsmJob<Fq>> lpj[HP];   // l parts 1 .. (part 0 is wj1[0])
    G1XYZZ lout[HP];
    for (int q = 0; q < 3; q++) {
        if (q == 0 && lparts > 1) {
            for (int k = 0; k < lparts; k++) {
                const size_t lo = std::min(wq[0]->n, wq[0]->part_bounds[k]), hi = std::min(wq[0]->n, wq[0]->part_bounds[k + 1]);
                // (the part's own gate and nothing else: "everything queued so far" would be the whole H pipeline)
                ZG_HIP(ctx, hipStreamWaitEvent(ctx->side[lslot[k]], wq[0]->part_gate[k] ? wq[0]->part_gate[k] : ctx->side_ev, 0));
                auto j = make_range_job<Fq>(ctx, lslot[k], *wq[0], lo, hi, &lout[k]);
                j->witness = true;
                if (k == 0) wj1[0] = std::move(j); else lpj[k] = std::move(j);
            }
            continue;
        }
        wj1[q] = make_range_job<Fq>(ctx, w_slot[q], *wq[q], 0, wq[q]->n, &out1[q + 1]);
    }
    wj2 = make_range_job<Fq2>(ctx, w_slot[3], *wq[3], 0, wq[3]->n, out2);
    for (int q = 0; q < 3; q++) wj1[q]->witness = true;
    wj2->witness = true;
    // One digit sort for the witness queries that share a window-table plan (MsmShare): a sort-only job over the witness
    // positions, on the G2 query's stream (its chain is the longest: no event hop in front of it); the queries follow.
    // b_g1 and b_g2 multiply the same scalars at the same positions: with tables of one layout (MsmDesc::follow_g2) the G2
    // job's sorted entries ARE the G1 job's, which then runs no classification and no sort at all
    if (g1[3].follow_g2 && g1[3].n == g2.n && g2.n != 0 && !(share && (share->mask >> 3 & 3u))) {
        wj1[2]->src = wj2.get();
        wj2->ev_items = ctx->follow_ev;
    }
    std::unique_ptr<MsmJob<Fq>> sorter;
    G1XYZZ sorter_out;
    if (share && share->mask && share->n > 0) {
        sorter.reset(new MsmJob<Fq>(ctx, scratch_slot(ctx, Ctx::MSM_SLOTS), ctx->side[w_slot[3]], nullptr, share->scalars, nullptr,
                                    share->n, &sorter_out));
        sorter->sort_only = true;
        sorter->fbw_virtual = true;
        sorter->fbw_c = share->c;
        sorter->fbw_nwin = share->nwin;
        sorter->fbw_stride = (size_t)1 << SH_POS_BITS;
        sorter->d_cls = share->cls;
        sorter->eager_small = share->small != 0;
        sorter->ev_items = ctx->share_ev;
        MsmJobBase* fj[4] = {wj1[0].get(), wj1[1].get(), wj1[2].get(), wj2.get()};
        for (int q = 0; q < 4; q++) {
            if (!(share->mask >> (q + 1) & 1u) || wq[q]->n == 0) continue;
            fj[q]->src = sorter.get();
            const ShQuery sq{share->q[q + 1].rank, share->q[q + 1].sub, (uint32_t)wq[q]->n};
            if (q < 3) wj1[q]->shq = sq; else wj2->shq = sq;
        }
        if (share->small) {   // the G1 followers as one chain of launches (Batch), on the first one's stream
            MsmJob<Fq>* lead = nullptr;
            for (int q = 0; q < 3; q++) {
                MsmJob<Fq>* j = wj1[q].get();
                if (j->src != sorter.get() || j->done) continue;
                if (!lead) {
                    lead = j;
                } else {
                    j->leader = lead;
                    j->st = lead->st;
                    lead->members[lead->n_members++] = j;
                }
            }
        }
    }
    // h parts
    G1XYZZ hout[HP];
    std::unique_ptr<MsmJob<Fq>> hp[HP];
    {
        const size_t n = g1[0].n, chunk = (n + parts - 1) / parts;
        for (int k = 0; k < parts; k++) {
            size_t lo = std::min(n, (size_t)k * chunk), hi = std::min(n, lo + chunk);
            if (g1[0].part_bounds && parts == g1[0].parts) {   // the caller's own (unequal) split
                lo = std::min(n, g1[0].part_bounds[k]);
                hi = std::min(n, g1[0].part_bounds[k + 1]);
            }
            const int slot = k == 0 ? h_slot0 : 4 + k;
            hp[k] = make_range_job<Fq>(ctx, slot, g1[0], lo, hi, &hout[k]);
            if (g1[0].part_gate) {   // the caller's gates go with ITS part count; fewer parts here: wait for all of them
                if (parts == gates_n) {
                    if (g1[0].part_gate[k]) ZG_HIP(ctx, hipStreamWaitEvent(ctx->side[slot], g1[0].part_gate[k], 0));
                } else {
                    for (int g = 0; g < gates_n; g++)
                        if (g1[0].part_gate[g]) ZG_HIP(ctx, hipStreamWaitEvent(ctx->side[slot], g1[0].part_gate[g], 0));
                }
            }
            if (parts > 1) {
                hp[k]->ev_sorted = ctx->part_ev[2 * k];
                hp[k]->ev_acc = ctx->part_ev[2 * k + 1];
                if (k > 0) hp[k]->dep = hp[k - 1].get();
            }
        }
    }
    // Priority order for the host's polling loop: the h parts (on the critical path once the NTT pipeline is done; a
    // small proof otherwise waits ~1 ms for the host to finish queueing the witness jobs' phases), then the G2 job
    // (the longest chain), then the G1 witness jobs.
    // A small proof's jobs are all queued whole (eager): the G2 job first -- its chain of dependent additions is the
    // longest -- then the h job, then the G1 witness jobs (EdDSA circuit: the h job queued last finished last, 1.11 ms).
    MsmJobBase* jobs[16];
    int nj = 0;
    const bool small = wj2->eager_small && wj1[0]->eager_small && wj1[1]->eager_small && wj1[2]->eager_small;
    if (!small)
        for (int k = 0; k < parts; k++) jobs[nj++] = hp[k].get();
    if (sorter) jobs[nj++] = sorter.get();
    jobs[nj++] = wj2.get();
    // (b_g1 behind the G2 job's sort: its short chain -- no classification, no sort -- is queued right behind that job's, not
    // after the other queries': queued last it started 0.3 ms after the entries it reads were ready)
    const bool b1_follows = wj1[2]->src == wj2.get();
    if (b1_follows) jobs[nj++] = wj1[2].get();
    if (small)   // second: queueing a job takes the host ~80 us, and the NTT pipeline of a small proof is done in 0.3 ms
        for (int k = 0; k < parts; k++) jobs[nj++] = hp[k].get();
    for (int q = 0; q < 3; q++) {
        if (!(q == 2 && b1_follows)) jobs[nj++] = wj1[q].get();
        if (q == 0)
            for (int k = 1; k < lparts; k++) jobs[nj++] = lpj[k].get();
    }
    auto sum_l_parts = [&] {   // (every part has finished: called from the witness hook and again, harmlessly, below)
        if (lparts <= 1) return;
        G1XYZZ acc = lout[0];
        for (int k = 1; k < lparts; k++) acc = acc.add(lout[k]);
        out1[1] = acc;
    };
    std::function<void()> on_witness;
    if (ctx->witness_hook)
        on_witness = [&] {
            sum_l_parts();
            ctx->witness_hook(out1, out2);
        };
    ZG_TRY(run_jobs(ctx, jobs, nj, on_witness));
    sum_l_parts();
    G1XYZZ h = hout[0];
    for (int k = 1; k < parts; k++) h = h.add(hout[k]);
    out1[0] = h;
    return ZG_OK;
}

// table[w * stride + i] = 2^(c*w) * bases[i], affine, w = 0 .. nwin-1 (row 0 is a copy of the bases)
template <class F>
__global__ void __launch_bounds__(256) k_fbw_build(const Affine<F>* __restrict__ bases, size_t n, size_t stride, int c,
                                                   int nwin, Affine<F>* __restrict__ table) {
    using L = typename LazyOf<F>::L;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> cur = bases[i];
    table[i] = cur;
    for (int w = 1; w < nwin; w++) {
        AffD<L> a = AffD<L>::load(&cur);
        XYZZd<L> p = XYZZd<L>::dbl_affine(a);
        for (int k = 1; k < c; k++) p = p.dbl();
        p.to_affine_mem(&cur);
        table[(size_t)w * stride + i] = cur;
    }
}

int msm_fbw_build_g1(Ctx* ctx, const G1Affine* d_bases, size_t n, size_t stride, int c, int nwin, G1Affine* d_table) {
    if (n == 0) return ZG_OK;
    ZG_LAUNCH(ctx, (k_fbw_build<Fq>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_bases, n, stride, c,
                       nwin, d_table);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}
int msm_fbw_build_g2(Ctx* ctx, const G2Affine* d_bases, size_t n, size_t stride, int c, int nwin, G2Affine* d_table) {
    if (n == 0) return ZG_OK;
    ZG_LAUNCH(ctx, (k_fbw_build<Fq2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_bases, n, stride,
                       c, nwin, d_table);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

// Subset-sum table for a query whose assignment is mostly 0 / 1: out[(2^qw - 1) g + m - 1] = sum of bases[qw g + j] over
// the bits j of m (m = 1 .. 2^qw - 1), affine, qw = 4 or 8 positions per group.  The ones of a group then cost ONE mixed
// add instead of one each (two on average per 4 positions, four per 8).  3.75 x (qw = 4) resp. 31.9 x (qw = 8) the
// query's footprint -- 8.8 GB for a 2^22-position G1 query: the kind of trade 288 GB of HBM is there for.
template <class F>
__global__ void __launch_bounds__(256) k_quad_build(const Affine<F>* __restrict__ bases, size_t n, Affine<F>* __restrict__ out,
                                                    int qw) {
    using L = typename LazyOf<F>::L;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nsum = ((size_t)1 << qw) - 1, groups = (n + qw - 1) / qw;
    if (t >= groups * nsum) return;
    const size_t g = t / nsum;
    const uint32_t m = (uint32_t)(t - g * nsum) + 1u;
    XYZZd<L> acc = XYZZd<L>::inf();
    for (int j = 0; j < qw; j++) {
        const size_t i = (size_t)qw * g + j;
        if (!((m >> j) & 1u) || i >= n) continue;
        acc = acc.add_mixed(AffD<L>::load(&bases[i]));
    }
    acc.to_affine_mem(&out[t]);
}
template <class F>
static int quad_build(Ctx* ctx, const Affine<F>* d_bases, size_t n, Affine<F>* d_buf, int qw) {
    if (n == 0) return ZG_OK;
    ZG_HIP(ctx, hipMemcpyAsync(d_buf, d_bases, n * sizeof(Affine<F>), hipMemcpyDeviceToDevice, ctx->stream));
    const size_t threads = (n + qw - 1) / qw * (((size_t)1 << qw) - 1);
    ZG_LAUNCH(ctx, (k_quad_build<F>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, d_bases, n,
                       d_buf + n, qw);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}
int msm_quad_build_g1(Ctx* ctx, const G1Affine* d_bases, size_t n, G1Affine* d_buf, int qw) { return quad_build<Fq>(ctx, d_bases, n, d_buf, qw); }
int msm_quad_build_g2(Ctx* ctx, const G2Affine* d_bases, size_t n, G2Affine* d_buf, int qw) { return quad_build<Fq2>(ctx, d_bases, n, d_buf, qw); }
size_t msm_quad_entries(size_t n, int qw) { return n + (n + qw - 1) / qw * (((size_t)1 << qw) - 1); }

// blocking: how many of the n scalars (through the index indirection) are 0 / 1
int msm_classify_counts(Ctx* ctx, const uint32_t* d_scalars, const uint32_t* d_index, size_t n, uint64_t* zeros,
                        uint64_t* ones) {
    *zeros = *ones = 0;
    if (n == 0) return ZG_OK;
    DevBuf cnt;
    ZG_HIP(ctx, cnt.alloc(16));
    ZG_HIP(ctx, hipMemsetAsync(cnt.p, 0, 16, ctx->stream));
    const int grid = (int)std::min<size_t>((n + 255) / 256, (size_t)ctx->num_cu * 4);
    ZG_LAUNCH(ctx, k_classify, dim3(grid), dim3(256), 0, ctx->stream, d_scalars, d_index, (const uint8_t*)nullptr, n,
                       cnt.as<uint32_t>(), (const uint32_t*)nullptr);
    uint32_t h[2] = {0, 0};
    ZG_HIP(ctx, hipMemcpyAsync(h, cnt.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *zeros = h[0];
    *ones = h[1];
    return ZG_OK;
}

// ---- the positions of a query that hold field elements, compacted (the columns of the compact window table)
__global__ void __launch_bounds__(256) k_real_flags(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ index,
                                                    size_t n, uint32_t* __restrict__ flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Scalar256 s = load_scalar(scalars, index ? (size_t)index[i] : i);
    flag[i] = scalar_is_zero(s) || scalar_is_one(s) ? 0u : 1u;
}
template <class F>
__global__ void __launch_bounds__(256) k_compact_reals(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ excl,
                                                       size_t n, uint32_t* __restrict__ rank, Affine<F>* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = excl[i];
    if (excl[i + 1] != r) {   // (the scanned flags have n + 1 entries)
        rank[i] = r;
        out[r] = bases[i];
    } else {
        rank[i] = 0xffffffffu;
    }
}
template <class F>
static int compact_reals(Ctx* ctx, const Affine<F>* d_bases, const uint32_t* d_scalars, const uint32_t* d_index, size_t n,
                         uint32_t* d_rank, Affine<F>* d_compact, uint64_t* n_real) {
    *n_real = 0;
    if (n == 0) return ZG_OK;
    DevBuf fl, bs;
    const uint32_t nbx = (uint32_t)((n + 1 + SCAN_TILE - 1) / SCAN_TILE);
    ZG_HIP(ctx, fl.alloc((n + 1) * 4));
    ZG_HIP(ctx, bs.alloc((size_t)nbx * 4));
    hipStream_t st = ctx->stream;
    ZG_LAUNCH(ctx, k_real_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, d_index, n, fl.as<uint32_t>());
    ZG_LAUNCH(ctx, k_xscan_reduce, dim3(nbx), dim3(SCAN_THREADS), 0, st, fl.as<uint32_t>(), n, bs.as<uint32_t>());
    ZG_LAUNCH(ctx, k_xscan_bsums, dim3(1), dim3(256), 0, st, bs.as<uint32_t>(), nbx);
    ZG_LAUNCH(ctx, k_xscan_final, dim3(nbx), dim3(SCAN_THREADS), 0, st, fl.as<uint32_t>(), n, bs.as<uint32_t>());
    ZG_LAUNCH(ctx, (k_compact_reals<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_bases, fl.as<uint32_t>(), n,
                       d_rank, d_compact);
    uint32_t total = 0;
    ZG_HIP(ctx, hipMemcpyAsync(&total, fl.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, st));
    ZG_HIP(ctx, hipStreamSynchronize(st));
    ZG_HIP(ctx, hipGetLastError());
    *n_real = total;
    return ZG_OK;
}
int msm_compact_reals_g1(Ctx* ctx, const G1Affine* d_bases, const uint32_t* d_scalars, const uint32_t* d_index, size_t n,
                         uint32_t* d_rank, G1Affine* d_compact, uint64_t* n_real) {
    return compact_reals<Fq>(ctx, d_bases, d_scalars, d_index, n, d_rank, d_compact, n_real);
}
int msm_compact_reals_g2(Ctx* ctx, const G2Affine* d_bases, const uint32_t* d_scalars, const uint32_t* d_index, size_t n,
                         uint32_t* d_rank, G2Affine* d_compact, uint64_t* n_real) {
    return compact_reals<Fq2>(ctx, d_bases, d_scalars, d_index, n, d_rank, d_compact, n_real);
}

void msm_scratch_free(Ctx* ctx) {
    delete ctx->finish_pool;
    ctx->finish_pool = nullptr;
    for (int k = 0; k < Ctx::MSM_SLOTS + 1; k++) {
        if (ctx->msm_scratch[k]) {
            if (ctx->msm_scratch[k]->host) (void)hipHostFree(ctx->msm_scratch[k]->host);
            delete ctx->msm_scratch[k];
            ctx->msm_scratch[k] = nullptr;
        }
    }
    if (ctx->side_future.valid() && ctx->side_future.get()) ctx->side_ready = true;   // (created on the helper thread, never used)
    if (ctx->side_ready) {
        for (int k = 0; k < Ctx::MSM_SLOTS; k++) (void)hipStreamDestroy(ctx->side[k]);
        for (int k = 0; k < 2 * Ctx::H_PARTS_MAX; k++) (void)hipEventDestroy(ctx->part_ev[k]);
        (void)hipEventDestroy(ctx->side_ev);
        (void)hipEventDestroy(ctx->share_ev);
        (void)hipEventDestroy(ctx->follow_ev);
        ctx->side_ready = false;
    }
}

}  // namespace zg
