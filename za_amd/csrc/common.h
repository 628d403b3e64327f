// Internal declarations shared by the HIP translation units of libza_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <thread>
#include <algorithm>
#include <functional>
#include <future>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/za_amd.h"
#include "ecd.h"

namespace zg {

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // stream every kernel is launched on
    hipStream_t own_stream = nullptr;   // created by zg_init
    int num_cu = 256;
    std::string last_error;
    double timings[6] = {0, 0, 0, 0, 0, 0};
    double load_timings[4] = {0, 0, 0, 0};   // zg_last_load_timings
    // optional per-kernel timing with HIP events on `stream` (zg_profile_*)
    int prof_enabled = 0;   // 0 off, 1 = the accumulation kernels only (about 5 scopes per proof), 2 = every class
    struct ProfPending { int id; hipEvent_t e0, e1; uint64_t bytes, units; };
    std::vector<ProfPending> prof_pending;
    std::vector<hipEvent_t> prof_pool;   // recycled timing events (creating two per scope cost 0.3 ms per proof)
    std::mutex prof_mu;                  // the MSM jobs of a proof run on their own host threads
    static constexpr int PROF_SLOTS = 10;   // = PROF_COUNT (ProfId below; checked there)
    struct ProfStat { double ms = 0; uint64_t launches = 0, bytes = 0, units = 0; } prof[PROF_SLOTS];
    // cached NTT tables per log_m (device pointers), see ntt.hip
    struct NttTables* ntt[28] = {nullptr};
    bool one_shot = false;       // zg_set_option(ZG_OPT_ONE_SHOT): no per-key precomputation
    // zg_set_option: the witness hand-over and the per-key precomputation (include/za_amd.h)
    int opt_pack = 1;            // 0 never, 1 always for a bit-heavy witness, 2 with the slow-host fallback
    int opt_pack_threads = 0;    // 0 = min(16, hardware threads / 2)
    int opt_pack_min_log_n = 18;
    bool opt_pack_pin = true;
    int opt_pack_budget_permille = 900;
    int opt_pack_raw_permille = 400;  // ZG_OPT_PACK_RAW_PERMILLE
    bool opt_combine_early = true;    // ZG_OPT_COMBINE_EARLY
    // schedule switches (tests, the fuzzer and A/B tools set them; deployments leave the defaults) -- see include/za_amd.h
    int opt_small_log_n = 16;         // ZG_OPT_SMALL_LOG_N: dense jobs of up to 2^value points run in small-job mode (0 = off)
    int opt_small_w_log_n = -1;       // ZG_OPT_SMALL_W_LOG_N: witness jobs; -1 = 600 000 positions ("2^19 and a bit")
    int opt_sort_staged = -1;         // ZG_OPT_SORT_STAGED: -1 = per job, 0 = two-level scatter, 2 / 3 = staged for every large / every job
    int64_t opt_part_stage_min = -1;  // ZG_OPT_PART_STAGE_MIN: last-level segments from this many entries go through LDS (-1 = default)
    int opt_subset_w = 8;             // ZG_OPT_SUBSET_W: positions per subset-sum group (4 | 8), read when a key's tables are built
    bool opt_compact_table = true;    // ZG_OPT_COMPACT_TABLE: compact window table behind the subset sums, read at table build
    int opt_h_parts = 0;              // ZG_OPT_H_PARTS: parts of the h multiexp (0 = the scheduler's choice)
    int opt_h_pipe = -1;              // ZG_OPT_H_PIPE: last transform kernel launched range by range, h parts gated per range
                                      // (-1 = where the h multiexp runs in two parts: a lone context from 2^22 rows; 1 = from 2^20; 0 = off)
    int opt_ntt_batch_log_m = 21;     // ZG_OPT_NTT_BATCH_LOG_M: up to 2^value rows the H pipeline's passes take a, b, c in one launch
    int opt_share_sort = 1;           // ZG_OPT_SHARE_SORT: 0 a sort per query; 1 b_g1 behind b_g2, and one sort for the
                                      // full-density queries of one window-table plan; 2 one sort for all of them (position maps)
    int opt_c_fold = 1;               // ZG_OPT_C_FOLD: 0 six transforms; 1 four where the C rows evaluate to bits (decided per key, when
                                      // its tables are built); 2 four for every key that has the evaluation form (tests)
    int opt_c_fold_min_log_m = 20;    // ZG_OPT_C_FOLD_MIN_LOG_M: mode 1 only from domains of 2^value rows
    int opt_upload_gates = 23;        // ZG_OPT_UPLOAD_GATES: a witness of 2^value values or more that does not pack goes over in chunks, the
                                      // l multiexp in parts behind them (0 = never)
    // msm_batch5 calls it (on the thread that finished the last witness job) once the sums of l, a, b_g1 (g1[1..3]) and b_g2
    // are final and the h job is still running; set and cleared by zg_prove_dev
    std::function<void(const XYZZ<Fq>* g1, const XYZZ<Fq2>* g2)> witness_hook;
    bool opt_h_eval = true;
    unsigned opt_tables = 31;
    uint64_t opt_tables_max_mb = 0;
    bool ntt_attr_set = false;   // the transform kernels' dynamic-LDS attribute is set on this ctx's device
    // persistent prove workspace (a, b, c evaluation vectors and H scalars), grown on demand
    struct ProveWs* prove_ws = nullptr;
    // reusable MSM scratch (one per concurrent job) and side streams, see msm.hip
    static constexpr int MSM_SLOTS = 8;   // 5 queries + up to 3 extra parts of the h job (main stream + 8 side streams:
                                          // more would share hardware queues)
    static constexpr int H_PARTS_MAX = 4;
    struct FinishPool* finish_pool = nullptr;   // msm.hip: the helper threads of the jobs' host passes
    struct MsmScratch* msm_scratch[MSM_SLOTS + 1] = {nullptr};   // [MSM_SLOTS]: the shared digit sort of the witness queries
    hipEvent_t share_ev = nullptr, follow_ev = nullptr;           // "the shared sort's / the b_g2 job's item arrays are final"
    hipStream_t side[MSM_SLOTS] = {nullptr};
    hipEvent_t side_ev = nullptr;
    hipEvent_t part_ev[2 * H_PARTS_MAX] = {nullptr};   // (sorted, accumulated) per part of the h job
    hipEvent_t ev_witness = nullptr, ev_h = nullptr;   // gates for the overlapped prove (prover.hip): borrowed pointers
    hipEvent_t gate_w = nullptr, gate_h = nullptr;     // the ctx's own gate events (created lazily, destroyed by zg_free)
    hipEvent_t h_split_ev[H_PARTS_MAX] = {nullptr};    // "range k of the evaluation-form H scalars is there" (h_eval_dev)
    hipEvent_t tm_ev[3] = {nullptr, nullptr, nullptr};  // timing of the H pipeline inside zg_prove_dev: start, R1CS evaluation done, H done
    bool witness_gate_armed = false;                   // zg_prove_h_enqueue_dev recorded gate_w for the next partials call
    bool side_ready = false;
    std::future<bool> side_future;   // msm.hip msm_side_streams_begin: the side streams being created on a helper thread
    // zg_last_stats: kernel launches since the counter was last cleared (zg_prove* clear it on entry; the helper threads
    // of a proof only run host arithmetic, every launch comes from the calling thread), queries of the last multiexp batch
    // that shared one digit sort (bit k = query k)
    uint64_t n_launches = 0;
    unsigned last_share_mask = 0;
    int last_l_parts = 1;       // the l multiexp ran in this many parts behind a witness handed over in chunks (ZG_OPT_UPLOAD_GATES)
    bool last_follow_g2 = false;
    bool last_c_fold = false;
    bool fold_guarded = false;   // the pipeline queued last left the fold to k_fold_decide (prover.hip)
    // zg_prove_h_eval_enqueue_fold_dev queued a pipeline without C's transforms for this key: the next zg_prove_partials_ex
    // with the l query of the same key takes the folded query over these scalars / class bytes
    const void* fold_armed_key = nullptr;
    const void* fold_armed_r1cs = nullptr;      // ... for this circuit and this device witness: a partials call that names
    const void* fold_armed_witness = nullptr;   // another is refused (its class bytes would be rewritten underneath)
    const uint32_t* fold_scalars = nullptr;
    const uint8_t* fold_cls = nullptr;
    bool stats_keep = false;   // zg_prove / zg_prove_packed cleared the counter for the zg_prove_dev they are about to call
};

int fail(Ctx* ctx, int code, const char* fmt, ...);
int fail_msg(Ctx* ctx, int code, const char* msg);   // (fail with a finished message: what csrc/internal_api.h hands out)
// Every extern "C" entry point is a function-try-block ending in `return on_exception(ctx)`: no C++ exception
// (std::bad_alloc from a container sized by untrusted input, ...) crosses the C boundary.
int on_exception(Ctx* ctx);
int live_contexts();   // zg_ctx handles currently alive in this process (prover.hip)
// prover.hip's host-side witness packer on plain memory (tests: zg_selftest_pack_mt in libza_amd_synth.so)
int host_pack_for_test(const uint8_t* witness32, size_t nv, int threads, uint8_t* classes, uint32_t* other_index,
                       uint8_t* others32, size_t* n_others);
int host_pack_split_for_test(const uint8_t* witness32, size_t nv, int threads, int cap_permille, int ready_every, uint8_t* raw_out,
                             uint8_t* classes, uint32_t* other_index, uint8_t* others32, size_t* n_others, size_t* first);

// kernel classes timed by the profiling facility
enum ProfId { PROF_MSM_ACC_G1 = 0, PROF_MSM_ACC_G2, PROF_MSM_SORT, PROF_MSM_REDUCE, PROF_NTT_PASS, PROF_NTT_POINTWISE,
              PROF_R1CS_EVAL, PROF_MSM_ACC_G1_DENSE /* the h job: every point reaches the kernel */,
              PROF_MSM_SORT_DENSE, PROF_MSM_REDUCE_DENSE /* the h job's digit sort and bucket reduction */, PROF_COUNT };
static_assert(PROF_COUNT == Ctx::PROF_SLOTS, "Ctx::prof holds one slot per ProfId");
struct ProfScope {
    Ctx* c; int idx = -1;
    hipStream_t st;
    ProfScope(Ctx* ctx, int id, uint64_t alg_bytes, hipStream_t stream = nullptr, uint64_t units = 0)
        : c(ctx), st(stream ? stream : ctx->stream) {
        if (ctx->prof_enabled < (id == PROF_MSM_ACC_G1 || id == PROF_MSM_ACC_G2 || id == PROF_MSM_ACC_G1_DENSE ? 1 : 2)) return;
        std::lock_guard<std::mutex> lock(ctx->prof_mu);
        Ctx::ProfPending p{id, nullptr, nullptr, alg_bytes, units};
        for (hipEvent_t* e : {&p.e0, &p.e1}) {
            if (!ctx->prof_pool.empty()) {
                *e = ctx->prof_pool.back();
                ctx->prof_pool.pop_back();
            } else if (hipEventCreate(e) != hipSuccess) {
                return;
            }
        }
        (void)hipEventRecord(p.e0, st);
        ctx->prof_pending.push_back(p);
        idx = (int)ctx->prof_pending.size() - 1;
    }
    ~ProfScope() {
        if (idx < 0) return;
        hipEvent_t e1;
        {
            std::lock_guard<std::mutex> lock(c->prof_mu);
            e1 = c->prof_pending[idx].e1;
        }
        (void)hipEventRecord(e1, st);
    }
};

#define ZG_HIP(ctx, expr)                                                                      \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::zg::fail((ctx), ZG_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                              __FILE__, __LINE__);                                             \
    } while (0)

// every kernel launch of the product path goes through here: zg_last_stats reports the count per proof
#define ZG_LAUNCH(ctxp, ...)                 \
    do {                                     \
        (ctxp)->n_launches++;                \
        hipLaunchKernelGGL(__VA_ARGS__);     \
    } while (0)

#define ZG_TRY(expr)                \
    do {                            \
        int _rc = (expr);           \
        if (_rc != ZG_OK) return _rc; \
    } while (0)

// RAII device buffer on the ctx's device
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    // grow-only
    hipError_t reserve(size_t n) {
        if (n <= bytes) return hipSuccess;
        return alloc(n);
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// fn(0) .. fn(n_tasks - 1) on up to 16 host threads (the calling one among them); a single task runs in place.  fn must not throw.
template <class Fn>
inline void host_par_for(size_t n_tasks, Fn fn) {
    unsigned hw = std::thread::hardware_concurrency();
    const size_t T = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 16), n_tasks);
    if (T <= 1) {
        for (size_t i = 0; i < n_tasks; i++) fn(i);
        return;
    }
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_tasks) break;
            fn(i);
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < T; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}

// ---- msm.hip
int msm_g1(Ctx* ctx, const G1Affine* d_bases, const uint32_t* d_scalars, const uint32_t* d_index,
           size_t n, G1XYZZ* out);
int msm_g2(Ctx* ctx, const G2Affine* d_bases, const uint32_t* d_scalars, const uint32_t* d_index,
           size_t n, G2XYZZ* out);
struct MsmDesc {
    const void* bases;
    const uint32_t* scalars;
    const uint32_t* index;
    size_t n;
    bool dense = false;   // scalars known to be essentially all "real" (H coefficients): skip the classification pass
    // optional fixed-base window table: [fbw_nwin][fbw_stride] affine points, row w = 2^(fbw_c*w) * bases
    const void* fbw_table = nullptr;
    int fbw_c = 0, fbw_nwin = 0;
    size_t fbw_stride = 0;
    // optional subset-sum table for a mostly-0/1 assignment: [n bases][15 sums per 4 positions]
    const void* quad_table = nullptr;
    int quad_w = 4;   // positions per group of the subset-sum table (4: 15 sums, 8: 255 sums)
    // optional with quad_table: the compact window table behind the subset sums (msm.hip MsmJob::cw_rank)
    const uint32_t* cw_rank = nullptr;
    uint32_t cw_off = 0, cw_n = 0;
    int cw_c = 0, cw_nwin = 0;
    // optional, dense jobs run in parts: part k's stream waits for part_gate[k] (e.g. "scalar chunk k has arrived")
    const hipEvent_t* part_gate = nullptr;
    bool part_gate_only = false;   // the part gates replace the job's common gate (ev_h) instead of adding to it
    int parts = 0;   // 0 = the scheduler's choice
    const size_t* part_bounds = nullptr;   // optional with parts > 0: parts + 1 ascending positions, [0] = 0, [parts] = n
    // optional: one class byte per scalar position (0 = zero, 1 = one, 2 = anything else; addressed like `scalars`,
    // through `index` if there is one).  The classification and the digit sort then read 1 byte instead of 32 for
    // every 0 / 1 scalar -- a witness query of the SHA-like workload is 97 % of those, and each job makes three passes.
    const uint8_t* cls = nullptr;
    // small-job mode (msm.hip): 1 / 0 = decided with the key's table for this query, -1 = by the job's size
    int small_hint = -1;
    // (the b_g1 query only) its table is laid out like the b_g2 query's -- same kind, digits, group width, compact rows --
    // so its job may run over the G2 job's sorted entries instead of sorting the same scalars again
    bool follow_g2 = false;
    // optional (witness queries): the job's stream also waits for this event (the l query with the C rows folded in reads
    // scalars that the R1CS evaluation writes)
    hipEvent_t gate = nullptr;
};
int msm_quad_build_g1(Ctx* ctx, const G1Affine* d_bases, size_t n, G1Affine* d_buf, int qw);
int msm_quad_build_g2(Ctx* ctx, const G2Affine* d_bases, size_t n, G2Affine* d_buf, int qw);
size_t msm_quad_entries(size_t n, int qw);
bool msm_fbw_plan(size_t n_real, size_t n_table, size_t max_bytes, size_t elem_bytes, int* c_out, int* nwin_out, bool force = false,
                  bool always = false);
size_t msm_small_n(const Ctx* ctx);           // dense jobs up to this size are latency bound (planned for depth, always with a window table)
size_t msm_small_witness_n(const Ctx* ctx);   // witness (non-dense) jobs: queued whole, always with a window table
int trace_level();   // ZA_AMD_TRACE (0 = off): host time stamps to stderr -- the one environment variable the library reads
int msm_fbw_build_g1(Ctx* ctx, const G1Affine* d_bases, size_t n, size_t stride, int c, int nwin, G1Affine* d_table);
int msm_fbw_build_g2(Ctx* ctx, const G2Affine* d_bases, size_t n, size_t stride, int c, int nwin, G2Affine* d_table);
int msm_classify_counts(Ctx* ctx, const uint32_t* d_scalars, const uint32_t* d_index, size_t n, uint64_t* zeros,
                        uint64_t* ones);
// d_rank[i] = column of position i among the n scalars that are neither 0 nor 1 (in position order), ~0 for the others;
// d_compact[column] = d_bases[i].  *n_real = their number.  Blocking.
int msm_compact_reals_g1(Ctx* ctx, const G1Affine* d_bases, const uint32_t* d_scalars, const uint32_t* d_index, size_t n,
                         uint32_t* d_rank, G1Affine* d_compact, uint64_t* n_real);
int msm_compact_reals_g2(Ctx* ctx, const G2Affine* d_bases, const uint32_t* d_scalars, const uint32_t* d_index, size_t n,
                         uint32_t* d_rank, G2Affine* d_compact, uint64_t* n_real);
// the five multiexps of create_proof (h, l, a, b_g1 over G1; b_g2 over G2), run concurrently
// One digit sort for several witness queries (msm.hip ShQuery): l, a, b_g1 and b_g2 multiply the same scalars w_j, so the
// (position, digit) pairs are sorted once over the witness positions [pos0, pos0 + n) and every query in `mask` (bit k: query
// k, 1 = l .. 4 = b_g2; each with a window table of THIS plan) accumulates over that list through its own position map.
struct MsmShare {
    unsigned mask = 0;
    const uint32_t* scalars = nullptr;   // the witness from position pos0 on
    const uint8_t* cls = nullptr;        // its class bytes (MsmDesc::cls) from pos0 on, or NULL
    size_t n = 0;                        // positions; at most 2^26
    int c = 0, nwin = 0;                 // the common window-table plan
    int small = 0;                       // the queries run in small-job mode (all of them do, or none)
    struct Q {
        const uint32_t* rank = nullptr;  // [n] position - pos0 -> index in the whole query, ~0 = not in its density; NULL = identity
        uint32_t sub = 0;                // subtracted from that index: the first one this shard holds (or, for NULL, the position of index 0)
    } q[5];
};
int msm_batch5(Ctx* ctx, const MsmDesc g1[4], const MsmDesc& g2, G1XYZZ out1[4], G2XYZZ* out2,
               hipEvent_t ev_witness = nullptr, hipEvent_t ev_h = nullptr, const MsmShare* share = nullptr);
void msm_scratch_free(Ctx* ctx);
void msm_side_streams_begin(Ctx* ctx);   // start creating the context's side streams on a helper thread (zg_init)
int msm_side_streams_ready(Ctx* ctx);    // ... and wait for them (or create them now)
size_t msm_scratch_bytes(Ctx* ctx, int slot);   // device bytes the slot's sort buffers hold

// ---- pairing.hip (host): prod_i e(P_i, Q_i) == 1 ?  slow = the definition; gt_out (optional, 384 bytes): the product's value
bool pairing_product_is_one(const G2Affine* qs, const G1Affine* ps, int n, bool slow, uint8_t* gt_out = nullptr);

// ---- ntt.hip
int ntt_fr_dev(Ctx* ctx, Fr* d_data, uint32_t log_m, bool inverse, bool coset);
// a,b,c: m device-form elements each in HBM, zero padded, row i at index bitrev(i); d_h_out: m plain
// integers in natural order (coefficient m-1 is zero for a satisfied system).  a, b, c are clobbered.
int h_poly_dev(Ctx* ctx, Fr* d_a, Fr* d_b, Fr* d_c, uint32_t log_m, Fr* d_h_out);
// H in evaluation form: d_e_out[p] = ((A B - C) / Z)(g omega^bitrev(p)) as plain integers, m of them -- the scalars of
// a key's h_eval bases (heval.hip).  Six transforms in 4 L - 3 launches.  a, b, c as for h_poly_dev; clobbered.
int h_eval_dev(Ctx* ctx, Fr* d_a, Fr* d_b, Fr* d_c, uint32_t log_m, Fr* d_e_out, int nsplit = 0, const hipEvent_t* ev_split = nullptr,
               const uint32_t* d_skip_c = nullptr);
struct FrLimbs;
int ntt_tables_for_heval(Ctx* ctx, uint32_t log_m, const FrLimbs** tw_inv, const Fr** gi_lo, const Fr** gi_hi_int, int* lo_bits,
                         Fr* cfold_int = nullptr);
// heval.hip: d_out[p] (m affine points, device form) = sum_i omega^(-i bitrev(p)) (g^-i / m) H_i over the first m - 1
// points of the h query -- the bases that go with h_eval_dev's scalars.  One-time work per key.
// c_fold: d_out[j] (NATURAL order) = -(1 / (m Z(g))) sum_i omega^(-i j) H_i -- the bases that go with the C rows' values
// when the C term of the quotient is folded into the l multiexp (h_eval_dev with d_c == NULL) -- and d_out[m + j] = 2 d_out[j]:
// 2 m points
int h_eval_bases_build(Ctx* ctx, const G1Affine* d_h, uint64_t n_h, uint32_t log_m, G1Affine* d_out, bool c_fold = false);
// prover.hip: give a SHARDED key its slice of the evaluation-form h bases from the whole query (device form, on this
// device); no-op (state -1) when the query length is not 2^k - 1 or ZG_OPT_H_EVAL is 0
int params_h_eval_slice(Ctx* ctx, const struct ::zg_params* p, const G1Affine* d_h_full, uint64_t n_h, const G1Affine* d_h_eval_full);
int lagrange_at_tau_dev(Ctx* ctx, uint32_t log_m, const Fr& tau_host, Fr* d_out);
int powers_int_dev(Ctx* ctx, Fr* d_out, size_t n, const Fr& base_host, const Fr& scale_host);
// out[row] = sum_e coeff[e] * x[idx[e]] over CSR/CSC rows; coeff pre-multiplied by 2^522, x plain integers -> M261
int spmv_dev(Ctx* ctx, const uint64_t* d_ptr, const uint32_t* d_idx, const Fr* d_coeff, const Fr* d_x, uint64_t n_rows,
             Fr* d_out);
int bitrev_dev(Ctx* ctx, Fr* d_data, uint32_t log_m);   // in-place bit-reversal permutation
void ntt_tables_free(Ctx* ctx);

// ---- M261 (device memory form, x*2^261) <-> host Montgomery form (x*2^256), host side.
// m261_to_host accepts any representative below 2^256 (device values are reduce()d, < 2.01 p).
template <class Pw>
inline Fp<Pw> m261_to_host(const Fp<Pw>& v) {
    static const Fp<Pw> inv32 = Fp<Pw>::from_u64(32).inv();
    return v * inv32;
}
template <class Pw>
inline Fp<Pw> host_to_m261(const Fp<Pw>& v) {
    static const Fp<Pw> c32 = Fp<Pw>::from_u64(32);
    return v * c32;
}
inline Fq2 m261_to_host(const Fq2& v) { return Fq2{m261_to_host(v.c0), m261_to_host(v.c1)}; }
inline Fq2 host_to_m261(const Fq2& v) { return Fq2{host_to_m261(v.c0), host_to_m261(v.c1)}; }
template <class F>
inline XYZZ<F> m261_to_host(const XYZZ<F>& p) {
    return XYZZ<F>{m261_to_host(p.x), m261_to_host(p.y), m261_to_host(p.zz), m261_to_host(p.zzz)};
}
template <class F>
inline Affine<F> host_to_m261(const Affine<F>& p) { return Affine<F>{host_to_m261(p.x), host_to_m261(p.y)}; }
template <class F>
inline Affine<F> m261_to_host(const Affine<F>& p) { return Affine<F>{m261_to_host(p.x), m261_to_host(p.y)}; }
// in-place conversion of n field elements in HBM: dir 0 = ABI (2^256) -> M261, 1 = M261 -> ABI; which 0 = Fq, 1 = Fr
int convert_mont_dev(Ctx* ctx, void* d_elems, size_t n, int which, int dir, hipStream_t st = nullptr);

// field selector for the lazy device arithmetic
template <class F> struct LazyOf;
template <> struct LazyOf<Fq> { using L = FqL; };
template <> struct LazyOf<Fq2> { using L = Fq2L; };

// ---- host helpers (field element <-> bytes)
template <class F>
inline void store_bytes(uint8_t* dst, const F& v) { memcpy(dst, &v, sizeof(F)); }
template <class F>
inline F load_bytes(const uint8_t* src) {
    F v;
    memcpy(&v, src, sizeof(F));
    return v;
}

}  // namespace zg
