// Groth16 create_proof on one MI355X plus the C ABI (include/za_amd.h).
//
// Mirrors bellman_ce::groth16::create_proof as called by za through create_random_proof
// (prover/src/groth16/prover.rs:173): R1CS evaluation (ProvingAssignment::enforce/eval), the H
// polynomial, the multiexps over h / l / a / b_g1 / b_g2 and the final combination with r, s.
// Everything heavy runs in HIP kernels; the host only sequences launches and does the O(1)
// final group arithmetic.  There is no CPU fallback: zg_init fails without a GPU.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <future>
#include <chrono>
#include <thread>
#include <memory>
#include <cstdarg>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <random>
#include <pthread.h>
#include <sched.h>

#include "common.h"
#include "formats.h"
#include "handles.h"
#include "internal_api.h"

namespace zg {

// ctx-less entry points (zg_verify, the JSON readers, ...) report through a per-thread message (zg_last_error_thread)
static thread_local std::string t_last_error;
int fail(Ctx* ctx, int code, const char* fmt, ...) {
    static std::mutex mu;   // the MSM jobs of a proof report from their own threads
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    t_last_error = buf;
    if (ctx) {
        std::lock_guard<std::mutex> lock(mu);
        ctx->last_error = buf;
    }
    return code;
}

int on_exception(Ctx* ctx) {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        return fail(ctx, ZG_E_IO, "out of host memory");
    } catch (const std::exception& e) {
        return fail(ctx, ZG_E_IO, "unexpected exception: %s", e.what());
    } catch (...) {
        return fail(ctx, ZG_E_IO, "unexpected exception");
    }
}

int fail_msg(Ctx* ctx, int code, const char* msg) { return fail(ctx, code, "%s", msg); }

static std::atomic<int> g_live_ctx{0};
int live_contexts() { return g_live_ctx.load(); }

static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------ kernels
// Device field values are M261 residues (x * 2^261, fd.h).
// coefficient c (plain integer) -> c * 2^522 mod r, so that coeff * w_integer (one Montgomery
// multiplication) is c*w in M261 form with no per-proof conversion of the witness
// Also classifies the coefficient: cls 0 = generic, 1 + s = +2^s, 65 + s = -2^s (s < 64).  Circuits are dominated
// by +-1 / +-2^k coefficients on bit signals; k_r1cs_eval_tiled sums those terms as plain integers.
__device__ __forceinline__ uint32_t pow2_class(const uint32_t v[8]) {
    if (v[2] | v[3] | v[4] | v[5] | v[6] | v[7]) return 0;
    const uint64_t x = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
    if (x == 0 || (x & (x - 1)) != 0) return 0;
    return 1u + (uint32_t)__builtin_ctzll(x);
}
__global__ void __launch_bounds__(256) k_coeff_prepare(Fr* __restrict__ c, size_t n, uint8_t* __restrict__ cls) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr v = c[i];
    uint32_t k = pow2_class(v.l);
    if (!k) {   // r - v
        uint32_t d[8];
        uint64_t borrow = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t t = (uint64_t)FrParams::MOD[j] - v.l[j] - borrow;
            d[j] = (uint32_t)t;
            borrow = (t >> 32) & 1;
        }
        k = pow2_class(d);
        if (k) k += 64;
    }
    cls[i] = (uint8_t)k;
    (FrD::to_mont(FrD::load_canon(&c[i])) * FrD::r2()).canonical().store(&c[i]);
}
// witness classes: 0 = zero, 1 = one, 2 = anything else.  The same pass validates the assignment: bellman's typed
// `Fr` cannot hold a value >= r and its variable 0 is the constant ONE (CS::one(), prover.rs:66), so a raw 32-byte
// witness that breaks either would give a silently wrong proof (next_digit assumes scalars below 2^254).
// bad[0] = lowest index of a non-canonical value, bad[1] = 0 if w[0] != 1; both words start as 0xffffffff.
// (the values [i0, i0 + n) of the witness: a hand-over in chunks classifies each chunk behind its copy)
__global__ void __launch_bounds__(256) k_witness_classes(const Fr* __restrict__ w, size_t i0, size_t n, uint8_t* __restrict__ cls,
                                                         uint32_t* __restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    i += i0;
    const Fr v = w[i];
    const uint32_t hi = v.l[1] | v.l[2] | v.l[3] | v.l[4] | v.l[5] | v.l[6] | v.l[7];
    const bool small = hi == 0 && v.l[0] <= 1;
    cls[i] = small ? (uint8_t)v.l[0] : (uint8_t)2;
    if (i == 0 && !(small && v.l[0] == 1)) bad[1] = 0u;
    if (v.l[7] >= FrParams::MOD[7]) {   // only then can v reach r (top word of r: 0x30644e72)
        bool ge = true;                  // v >= r, decided at the most significant differing word
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            if (v.l[j] != FrParams::MOD[j]) {
                ge = v.l[j] > FrParams::MOD[j];
                break;
            }
        }
        if (ge) atomicMin(&bad[0], (uint32_t)i);
    }
}

// zg_prove_packed: classes -> 32-byte values (class 2 positions are filled by k_unpack_others)
__global__ void __launch_bounds__(256) k_unpack_bits(const uint8_t* __restrict__ packed, size_t n, Fr* __restrict__ w) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = (packed[i >> 2] >> (2 * (i & 3))) & 3u;
    Fr v = Fr::zero();
    v.l[0] = c == 1 ? 1u : 0u;
    w[i] = v;
}
__global__ void __launch_bounds__(256) k_unpack_others(const uint32_t* __restrict__ idx, const Fr* __restrict__ vals, size_t n,
                                                       Fr* __restrict__ w) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    w[idx[j]] = vals[j];
}

// out[row] = sum_e coeff[e] * w[col[e]]  for one CSR matrix; one thread per row
__global__ void __launch_bounds__(256) k_r1cs_eval(const uint64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                   const Fr* __restrict__ coeff, const Fr* __restrict__ w,
                                                   uint64_t n_rows, Fr* __restrict__ out, int rev_bits) {
    uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    FrD acc = FrD::zero();
    uint32_t since = 0;
    for (uint64_t e = row_ptr[row], end = row_ptr[row + 1]; e < end; e++) {
        acc = acc + FrD::load_canon(&coeff[e]) * FrD::load_canon(&w[col[e]]);
        if (++since == 64) {  // each term adds < 1.01 p to the lazy bound
            acc = acc.reduce();
            since = 0;
        }
    }
    // rev_bits > 0: row i goes to index bitrev(i) -- the order the decimation-in-time inverse NTT consumes
    const uint64_t dst = rev_bits ? (uint64_t)(__brev((uint32_t)row) >> (32 - rev_bits)) : row;
    acc.reduce().store(&out[dst]);
}
// the `input_i * 0 = 0` rows: a[n + i] = w_i; b, c stay zero
__global__ void k_input_rows(const Fr* __restrict__ w, uint32_t n_inputs, uint64_t first_row, Fr* __restrict__ a,
                             int rev_bits) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inputs) return;
    const uint64_t row = first_row + i;
    const uint64_t dst = rev_bits ? (uint64_t)(__brev((uint32_t)row) >> (32 - rev_bits)) : row;
    FrD::to_mont(FrD::load_canon(&w[i])).reduce().store(&a[dst]);
}

// sum over v[0..n) of the non-zero 16-bit digits of each value: what a multiexp of these scalars costs in additions (zero
// digits are skipped), whatever their classes are -- a row whose C side evaluates to 2 is "anything else" by class and one
// addition by cost
__global__ void __launch_bounds__(256) k_count_digits16(const Fr* __restrict__ v, size_t n, unsigned long long* __restrict__ out) {
    __shared__ uint32_t part[4];
    uint32_t cnt = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Fr x = v[i];
#pragma unroll
        for (int q = 0; q < 8; q++) cnt += ((x.l[q] & 0xffffu) != 0) + ((x.l[q] >> 16) != 0);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)part[0] + part[1] + part[2] + part[3]);
}

// The per-proof guard of the folded form (ZG_OPT_C_FOLD = 1).  A key folds its C rows because the FIRST witness it saw had
// rows of bits and small integers; a later witness whose rows hold field elements would send m dense scalars into the l
// multiexp (Poseidon's shape forced: 7.5 -> 8.4 ms at 2^20).  The evaluator has summed the non-zero 16-bit digits of this
// witness's rows (R1csMats::c_digits); this kernel takes the key's own rule -- at most two digits per row -- for THIS proof:
// *flag = 1: folded (c's transform passes, already queued, leave at once: ntt.hip SkipC); *flag = 0: six transforms -- the
// rows' values are written where c's transform reads them and struck from the l multiexp (class 0 = a zero scalar, which
// no pass of the multiexp reads).  No host round trip: the launches behind it are queued before it runs.
__device__ __forceinline__ uint32_t rev4(uint32_t x) { return __brev(x) >> 28; }
static constexpr uint32_t DIGIT_SLOTS = 64;   // (a wave's worth: k_fold_decide adds them with shuffles)
__global__ void __launch_bounds__(256) k_fold_decide(const unsigned long long* __restrict__ digits, unsigned long long limit,
                                                     uint32_t* __restrict__ flag, Fr* __restrict__ c_nat, uint8_t* __restrict__ c_cls,
                                                     Fr* __restrict__ c_out, int log_m) {
    // (DIGIT_SLOTS counters: a SHA-256 circuit's ~10^5 long rows each add theirs, and on one word those atomics took 0.15 ms)
    unsigned long long sum = digits[threadIdx.x & (DIGIT_SLOTS - 1)];
#pragma unroll
    for (int d = DIGIT_SLOTS / 2; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    const bool fold = sum <= limit;
    if (blockIdx.x == 0 && threadIdx.x == 0) *flag = fold ? 1u : 0u;
    if (fold) return;
    // Six transforms: the rows go where the evaluator would have put them for c's transform (k_r1cs_eval_tiled's order, in
    // Montgomery form) and leave the l multiexp.  Only here -- a proof that folds pays nothing for the other form.
    const size_t m = (size_t)1 << log_m;
    for (size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x; row < m; row += (size_t)gridDim.x * blockDim.x) {
        Fr v = c_nat[row];
        if (c_nat[m + row].l[0]) v.l[0] = 2;   // (store_folded_c: a row equal to 2 is a one on the doubled bases)
        const uint32_t ta = (uint32_t)(row >> (log_m - 4)), tb = (uint32_t)(row & 15);
        const uint32_t mid = (uint32_t)((row >> 4) & (((uint64_t)1 << (log_m - 8)) - 1));
        const uint32_t rmid = log_m > 8 ? (__brev(mid) >> (32 - (log_m - 8))) : 0u;
        const uint64_t dst = ((uint64_t)rev4(tb) << (log_m - 4)) | ((uint64_t)rmid << 4) | rev4(ta);
        FrD::to_mont(FrD::load_canon(&v)).reduce().store(&c_out[dst]);
        c_nat[row] = Fr::zero();
        c_nat[m + row] = Fr::zero();
        c_cls[row] = 0;
        c_cls[m + row] = 0;
    }
}

// first row with a*b != c (mod r): atomicMin over the row index
__global__ void __launch_bounds__(256) k_r1cs_check(const Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c,
                                                    uint64_t n_rows, unsigned long long* __restrict__ first_bad) {
    uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const FrD ab = FrD::load_red(&a[row]) * FrD::load_red(&b[row]);   // a*b in M261 needs one factor out of Montgomery form:
    // a = x*R, b = y*R  ->  a*b = x*y*R (Montgomery product); c = z*R: compare directly
    const FrD d = FrD::sub<3>(ab, FrD::load_red(&c[row]));
    if (!d.is_zero_mod_p()) atomicMin(first_bad, (unsigned long long)row);
}

// A few persistent host threads for the one piece of per-proof host work that is worth spreading: packing a
// bit-heavy witness before it crosses PCIe (zg_prove).  parallel_for(n, fn) runs fn(0) .. fn(n - 1) on the workers
// and the calling thread and returns when all TASKS are done -- not when every worker has checked in: on a busy host a
// worker may wake a millisecond late, and then it simply finds nothing left to do (the caller and the others have taken
// its share), instead of the whole pass waiting for it.
// cpus (may be empty): the workers are bound to these CPUs -- the GPU's NUMA node, where the pinned witness lives.
class WorkerPool {
public:
    WorkerPool(int n_threads, const std::vector<int>& cpus) {
        for (int i = 0; i < n_threads; i++) th.emplace_back([this] { loop(); });
        if (!cpus.empty()) {
            cpu_set_t set;
            CPU_ZERO(&set);
            for (int c : cpus)
                if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
            for (auto& t : th) (void)pthread_setaffinity_np(t.native_handle(), sizeof(set), &set);   // best effort
            pinned = true;
        }
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lock(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    int threads() const { return (int)th.size() + 1; }
    bool is_pinned() const { return pinned; }
    // lead: run by the calling thread once the workers have been woken, before it takes tasks itself
    void parallel_for(int n_tasks, const std::function<void(int)>& f, const std::function<void()>* lead = nullptr) {
        if (n_tasks <= 0) return;
        uint32_t g;
        {
            std::lock_guard<std::mutex> lock(mu);
            fn = &f;
            total = n_tasks;
            g = ++gen;
            done.store(0, std::memory_order_relaxed);
            next.store((uint64_t)g << 32, std::memory_order_release);
        }
        cv.notify_all();
        if (lead) (*lead)();
        run(g, &f, n_tasks);
        // what is left are the tasks other threads are in the middle of: microseconds
        for (int spin = 0; spin < 200000 && done.load(std::memory_order_acquire) != n_tasks; spin++) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        std::unique_lock<std::mutex> lock(mu);
        cv_done.wait(lock, [&] { return done.load(std::memory_order_acquire) == n_tasks; });
        fn = nullptr;
    }

private:
    // take tasks of generation g only: a worker that wakes after its generation is over must neither run a stale
    // function nor take an index of the next generation
    void run(uint32_t g, const std::function<void(int)>* f, int n_tasks) {
        for (;;) {
            uint64_t v = next.load(std::memory_order_acquire);
            if ((uint32_t)(v >> 32) != g || (int)(uint32_t)v >= n_tasks) return;
            if (!next.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) continue;
            (*f)((int)(uint32_t)v);
            if (done.fetch_add(1, std::memory_order_acq_rel) + 1 == n_tasks) {
                std::lock_guard<std::mutex> lock(mu);   // (the waiter checks `done` under this mutex: no lost wake-up)
                cv_done.notify_one();
            }
        }
    }
    void loop() {
        uint32_t seen = 0;
        for (;;) {
            const std::function<void(int)>* f;
            int n_tasks;
            uint32_t g;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = g = gen;
                f = fn;
                n_tasks = total;
            }
            if (f) run(g, f, n_tasks);
        }
    }
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    const std::function<void(int)>* fn = nullptr;
    int total = 0;
    std::atomic<uint64_t> next{0};   // generation << 32 | next task
    std::atomic<int> done{0};
    uint32_t gen = 0;
    bool stop = false, pinned = false;
};

// CPUs of the NUMA node the GPU hangs off (sysfs; empty when that cannot be told, when the node has no CPUs this process
// may run on, or when there is only one node)
static std::vector<int> gpu_node_cpus(int device) {
    std::vector<int> cpus;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) {
        (void)hipGetLastError();
        return cpus;
    }
    for (char* c = bus; *c; c++) *c = (char)tolower(*c);
    int node = -1;
    {
        const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
        FILE* f = fopen(path.c_str(), "r");
        if (!f) return cpus;
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    if (node < 0) return cpus;
    {
        FILE* f = fopen("/sys/devices/system/node/node1/cpulist", "r");   // a single-node host: nothing to choose
        if (!f) return cpus;
        fclose(f);
    }
    char list[4096] = {0};
    {
        const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
        FILE* f = fopen(path.c_str(), "r");
        if (!f) return cpus;
        if (!fgets(list, sizeof(list), f)) list[0] = 0;
        fclose(f);
    }
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return cpus;
    for (char* tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k < 1) continue;
        if (k == 1) b = a;
        for (int c = a; c <= b && c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
    }
    return cpus;
}

static inline uint32_t value_class(const uint8_t* v) {
    uint64_t q[4];
    memcpy(q, v, 32);
    return (q[1] | q[2] | q[3]) == 0 && q[0] <= 1 ? (uint32_t)q[0] : 2u;
}

// The host part alone (no GPU call): classes -> `classes` ((nv + 3) / 4 bytes), the other values' positions and bytes
// back to back -> idx_out / val_out (room for nv / 8 of them).  *n_others_out = their number, or SIZE_MAX when the
// witness is not the kind that packs (too dense over all, or one 2^14-value stretch more than half dense).
struct HostPacker {
    std::unique_ptr<WorkerPool> pool;
    struct Slice {
        std::vector<uint32_t> idx;
        std::vector<uint8_t> val;
        size_t count = 0, offset = 0;
    };
    std::vector<Slice> slices;
    static constexpr size_t SLICE_VALS = (size_t)1 << 14;   // 512 KiB of witness per task: 256 tasks at 2^22, handed out dynamically

    // pin_cpus: where the worker threads run (empty = wherever the scheduler puts them)
    // What the copy engine does meanwhile (pack_and_upload): ready() = it can take another piece now, submit(lo, n) queues
    // the values [lo, lo + n) as they are.  The array is consumed from both ends: the copier takes pieces from the front,
    // the packing threads slices from the back, until they meet -- whichever is faster on this host at this moment gets
    // the larger share, and both are done at the same time.
    struct RawCopier {
        std::function<bool()> ready;
        std::function<bool(size_t, size_t)> submit;
        size_t max_vals = 0;   // the copier's largest share
    };
    // Packs the values [*first_out, nv): classes[i >> 2] describes value i (absolute positions, also in the index list);
    // *first_out = 0 without a copier, else the (SLICE_VALS-aligned) number of values the copier took.  A witness that is
    // not of the packing kind: *n_others_out = SIZE_MAX (what the copier has taken by then stays taken).
    void run(const uint8_t* witness32, size_t nv, int nthreads, uint8_t* classes, uint32_t* idx_out, uint8_t* val_out,
             size_t* n_others_out, const std::function<void(const char*)>& stamp, const std::vector<int>& pin_cpus = {},
             const RawCopier* copier = nullptr, size_t* first_out = nullptr) {
        *n_others_out = SIZE_MAX;
        if (first_out) *first_out = 0;
        if (!pool || pool->threads() != nthreads || pool->is_pinned() != !pin_cpus.empty())
            pool.reset(new WorkerPool(nthreads - 1, pin_cpus));
        const int n_slices = (int)((nv + SLICE_VALS - 1) / SLICE_VALS);
        const size_t max_others = nv / 8;   // beyond 12.5 % the plain upload is as good
        if ((int)slices.size() < n_slices) slices.resize(n_slices);
        std::atomic<size_t> total{0};
        std::atomic<bool> abort{false};
        std::atomic<uint64_t> range{(uint64_t)(uint32_t)n_slices};   // lo << 32 | hi: the slices [lo, hi) are nobody's yet
        auto take_back = [&]() -> int {
            uint64_t v = range.load(std::memory_order_acquire);
            for (;;) {
                const uint32_t lo = (uint32_t)(v >> 32), hi = (uint32_t)v;
                if (lo >= hi) return -1;
                if (range.compare_exchange_weak(v, v - 1, std::memory_order_acq_rel)) return (int)hi - 1;
            }
        };
        auto take_front = [&](uint32_t want, uint32_t limit, uint32_t* lo_out) -> uint32_t {
            uint64_t v = range.load(std::memory_order_acquire);
            for (;;) {
                const uint32_t lo = (uint32_t)(v >> 32), hi = (uint32_t)v;
                if (lo >= hi || lo >= limit) return 0;
                const uint32_t k = std::min(want, std::min(hi, limit) - lo);
                if (range.compare_exchange_weak(v, v + ((uint64_t)k << 32), std::memory_order_acq_rel)) {
                    *lo_out = lo;
                    return k;
                }
            }
        };
        const std::function<void(int)> pack_one = [&](int) {
            try {   // (a worker thread must not let an exception escape: std::bad_alloc from the slice's lists ends the attempt)
                if (abort.load(std::memory_order_relaxed)) return;
                const int si = take_back();
                if (si < 0) return;
                Slice& sl = slices[si];
                sl.count = 0;
                const size_t lo = (size_t)si * SLICE_VALS, hi = std::min(nv, lo + SLICE_VALS);
                const size_t cap = SLICE_VALS / 2;   // a slice denser than 50 % ends the attempt
                // (the lists grow on demand: sized for 50 % from the start they were 74 MB of zero-filled heap at 2^22, 1.2 GB
                // at 2^26, for a witness whose slices are 3 % dense)
                if (sl.idx.size() < SLICE_VALS / 16) {
                    sl.idx.resize(SLICE_VALS / 16);
                    sl.val.resize(SLICE_VALS / 16 * 32);
                }
                size_t k = 0;
                for (size_t i = lo; i < hi; i += 4) {   // slice bounds are multiples of 4: one class byte per group
                    uint32_t byte = 0;
                    const size_t e = std::min(hi, i + 4);
                    for (size_t j = i; j < e; j++) {
                        const uint8_t* v = witness32 + 32 * j;
                        const uint32_t c = value_class(v);
                        byte |= c << (2 * (j - i));
                        if (c == 2u) {
                            if (k == sl.idx.size()) {
                                if (k >= cap) {
                                    abort.store(true);
                                    return;
                                }
                                sl.idx.resize(std::min(cap, 2 * k));
                                sl.val.resize(sl.idx.size() * 32);
                            }
                            sl.idx[k] = (uint32_t)j;
                            memcpy(&sl.val[32 * k], v, 32);
                            k++;
                        }
                    }
                    classes[i >> 2] = (uint8_t)byte;
                }
                sl.count = k;
                if (total.fetch_add(k) + k > max_others) abort.store(true);
            } catch (...) {
                abort.store(true);
            }
        };
        // the calling thread feeds the copy engine (pieces of ~4 MB, two in flight) until the two ends meet, then packs too
        const uint32_t piece = (uint32_t)std::min(8, std::max(1, n_slices / 32));
        const uint32_t limit = copier ? (uint32_t)(copier->max_vals / SLICE_VALS) : 0u;
        const std::function<void()> feed = [&] {
            while (!abort.load(std::memory_order_relaxed)) {
                if (!copier->ready()) {
                    if ((uint32_t)(range.load(std::memory_order_acquire) >> 32) >= limit) break;
                    const uint64_t v = range.load(std::memory_order_acquire);
                    if ((uint32_t)(v >> 32) >= (uint32_t)v) break;
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                    continue;
                }
                uint32_t lo = 0;
                const uint32_t k = take_front(piece, limit, &lo);
                if (!k) break;
                const size_t v0 = (size_t)lo * SLICE_VALS, v1 = std::min(nv, (size_t)(lo + k) * SLICE_VALS);
                if (!copier->submit(v0, v1 - v0)) {
                    abort.store(true);
                    break;
                }
            }
        };
        pool->parallel_for(n_slices, pack_one, copier && limit ? &feed : nullptr);
        const uint32_t first_slice = (uint32_t)(range.load() >> 32);   // (after an abort: slices [lo, hi) were never taken)
        const size_t first = std::min(nv, (size_t)first_slice * SLICE_VALS);
        if (first_out) *first_out = first;
        stamp("scan");
        if (abort.load()) return;
        size_t n_others = 0;
        for (int si = (int)first_slice; si < n_slices; si++) {
            slices[si].offset = n_others;
            n_others += slices[si].count;
        }
        if (n_others > max_others) return;
        const int n_packed = n_slices - (int)first_slice;
        pool->parallel_for(n_packed, [&](int k) {   // the slices' lists, back to back, into the (pinned) staging buffer
            const Slice& sl = slices[first_slice + k];
            if (!sl.count) return;
            memcpy(idx_out + sl.offset, sl.idx.data(), sl.count * 4);
            memcpy(val_out + 32 * sl.offset, sl.val.data(), sl.count * 32);
        });
        stamp("gather");
        *n_others_out = n_others;
    }
};

struct ProveWs {
    DevBuf abc[3], h, wcls, w;
    DevBuf w_lc;   // "C folded", witness in the caller's memory: [aux part of the witness | the C rows' values] (c_fold_buffers)
    DevBuf cls_lc; // ... and for rank 0's shard of a sharded key: [class bytes of its aux range | the rows' classes]
    // zg_prove's host-side packing (see there): the packer (pool + per-slice lists) and the pinned staging of the packed form
    std::unique_ptr<HostPacker> packer;
    double pack_ema_ms = 0;   // running average of the packing pass's host time
    int pack_slow = 0;        // consecutive passes slower than the plain upload
    int pack_skip = 0;        // hand-overs of pack_skip_nv values left to go unpacked: the pass was slower than the plain upload
                              // (ZG_OPT_PACK = 2), or gave up on a witness of this size (not the kind that packs)
    size_t pack_skip_nv = 0;
    // pinned: [classes | other_index | others32].  Two of them, used in turn, each with the event behind its copies: a
    // hand-over is queued, not waited for (zg_witness_to_device), so the next one must not pack into bytes still in flight
    uint8_t* stage[2] = {nullptr, nullptr};
    size_t stage_bytes[2] = {0, 0};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    hipEvent_t dma_ev[2] = {nullptr, nullptr};   // behind the last two raw pieces of the hand-over in progress
    int stage_next = 0;
    std::vector<int> pin_cpus;   // the GPU's NUMA node (ZG_OPT_PACK_PIN), looked up once
    bool pin_looked_up = false;
    DevBuf pk_cls, pk_idx, pk_val;   // zg_prove_packed: the packed witness as uploaded
    // wcls: one class byte per witness value (k_witness_classes), shared by the evaluator and the witness multiexps.
    // wcls_for: the witness it was computed from; wcls_fresh: computed inside the current API call (the evaluator
    // takes it as is); wcls_armed: a proof is in flight whose multiexps may use it (zg_prove_dev, or
    // zg_prove_h_enqueue_dev -> zg_prove_partials_ex)
    const void* wcls_for = nullptr;
    bool wcls_fresh = false, wcls_armed = false;
    // A hand-over in chunks (witness_handover): gate_ev[k] = "the values below gate_var[k + 1] are in HBM and classified";
    // the l multiexp of the proof that follows runs in gate_n parts, part k over the aux values of chunk k, so its first
    // parts sort and accumulate while the rest of the witness is still on the bus.  gate_for: the device witness they
    // describe (NULL: none); classes_done_for: the class bytes were already made for it, chunk by chunk.
    static constexpr int GATES = 3;
    int gate_n = 0;
    size_t gate_var[GATES + 1] = {0, 0, 0, 0};
    hipEvent_t gate_ev[GATES] = {nullptr, nullptr, nullptr};
    const void* gate_for = nullptr;
    const void* classes_done_for = nullptr;
    // k_witness_classes' validation result: device words, their pinned host copy, the event behind the copy
    DevBuf bad;
    uint32_t* bad_host = nullptr;
    hipEvent_t bad_ev = nullptr;
    bool bad_pending = false;
    ~ProveWs() {
        for (int k = 0; k < 2; k++) {
            if (stage[k]) (void)hipHostFree(stage[k]);
            if (stage_ev[k]) (void)hipEventDestroy(stage_ev[k]);
            if (dma_ev[k]) (void)hipEventDestroy(dma_ev[k]);
        }
        if (bad_host) (void)hipHostFree(bad_host);
        if (bad_ev) (void)hipEventDestroy(bad_ev);
        for (auto& e : gate_ev)
            if (e) (void)hipEventDestroy(e);
    }
};
static ProveWs& prove_ws(Ctx* ctx) {
    if (!ctx->prove_ws) ctx->prove_ws = new ProveWs();
    return *ctx->prove_ws;
}

// The prove path's evaluator: one thread per row of the 2^log_m padded domain (rows >= n_rows are written as
// zero, so no memset), result stored at bitrev(row).  A block owns rows (a, mid, b) with a = top 4 bits, b = low 4
// bits, and exchanges through LDS so that both the CSR reads (b fastest) and the bit-reversed stores (rev(a)
// fastest) touch 512-byte runs instead of one 32-byte element per DRAM page.
// Terms whose coefficient is +-2^s and whose witness value is 0 or 1 are summed in two 128-bit integers (no gather
// of the 32-byte operands, no field multiplication); everything else takes the Montgomery path.
static constexpr uint64_t R1CS_LONG_ROW = 32;   // terms; longer rows go to k_r1cs_eval_long
// (the three matrices A, B, C in ONE launch: blockIdx.y picks the matrix -- two launches fewer on a small proof's critical path)
struct R1csMats {
    const uint64_t* row_ptr[3];
    const uint32_t* col[3];
    const Fr* coeff[3];
    const uint8_t* ccls[3];
    Fr* out[3];
    // "C folded" (ensure_c_fold): matrix 2's rows leave as plain integers in ROW order with their class bytes (0 / 1 /
    // other, as k_witness_classes) -- scalars of a multiexp, not input of a transform -- and out[2] is not written.
    // 2 m values: row j at [j], except that a row evaluating to 2 is written as 0 there and as 1 at [m + j] (the doubled
    // bases, heval.hip)
    Fr* c_nat = nullptr;
    uint8_t* c_cls = nullptr;
    // The per-proof guard of the folded form (k_fold_decide): with c_digits the non-zero 16-bit digits of the rows' values are
    // summed into c_digits[DIGIT_SLOTS] -- what their multiexp costs in additions; the decision "folded or six transforms" is
    // then taken on the device, for this witness, and only a proof that does not fold gets the rows in c's transform order.
    unsigned long long* c_digits = nullptr;
};
__device__ __forceinline__ uint32_t digits16(const Fr& x) {
    uint32_t cnt = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) cnt += ((x.l[q] & 0xffffu) != 0) + ((x.l[q] >> 16) != 0);
    return cnt;
}
__device__ __forceinline__ void store_folded_c(Fr* __restrict__ c_nat, uint8_t* __restrict__ c_cls, uint64_t row, uint64_t m, Fr cv) {
    const bool small = (cv.l[1] | cv.l[2] | cv.l[3] | cv.l[4] | cv.l[5] | cv.l[6] | cv.l[7]) == 0 && cv.l[0] <= 2;
    const bool two = small && cv.l[0] == 2;
    Fr hi = Fr::zero();
    if (two) {
        cv.l[0] = 0;
        hi.l[0] = 1;
    }
    c_nat[row] = cv;
    c_cls[row] = small ? (uint8_t)cv.l[0] : (uint8_t)2;
    c_nat[m + row] = hi;
    c_cls[m + row] = two ? (uint8_t)1 : (uint8_t)0;
}
__global__ void __launch_bounds__(256) k_r1cs_eval_tiled(R1csMats M, const Fr* __restrict__ w, const uint8_t* __restrict__ wcls,
                                                         uint64_t n_rows, int log_m) {
    const uint64_t* __restrict__ row_ptr = M.row_ptr[blockIdx.y];
    const uint32_t* __restrict__ col = M.col[blockIdx.y];
    const Fr* __restrict__ coeff = M.coeff[blockIdx.y];
    const uint8_t* __restrict__ ccls = M.ccls[blockIdx.y];
    Fr* __restrict__ out = M.out[blockIdx.y];
    __shared__ Fr tile[256];
    const uint32_t t = threadIdx.x, ta = t >> 4, tb = t & 15, mid = blockIdx.x;
    const uint64_t row = ((uint64_t)ta << (log_m - 4)) | ((uint64_t)mid << 4) | tb;
    FrD acc = FrD::zero();
    uint64_t e_lo = 0, e_hi = 0;
    if (row < n_rows) {
        e_lo = row_ptr[row];
        e_hi = row_ptr[row + 1];
        if (e_hi - e_lo > R1CS_LONG_ROW) e_hi = e_lo;   // left to k_r1cs_eval_long (queued behind this kernel)
    }
    if (row < n_rows) {
        uint64_t pos_lo = 0, pos_hi = 0, neg_lo = 0, neg_hi = 0;
        uint32_t since = 0;
        for (uint64_t e = e_lo, end = e_hi; e < end; e++) {
            const uint32_t c = col[e];
            const uint32_t wc = wcls[c];
            if (wc == 0) continue;
            const uint32_t k = ccls[e];
            if (wc == 1 && k != 0) {
                const uint64_t bit = 1ull << ((k - 1) & 63);
                if (k <= 64) {
                    pos_lo += bit;
                    pos_hi += pos_lo < bit;
                } else {
                    neg_lo += bit;
                    neg_hi += neg_lo < bit;
                }
                continue;
            }
            acc = acc + FrD::load_canon(&coeff[e]) * FrD::load_canon(&w[c]);
            if (++since == 64) {  // each term adds < 1.01 p to the lazy bound
                acc = acc.reduce();
                since = 0;
            }
        }
        acc = acc.reduce();
        if (pos_lo | pos_hi) {
            Fr v = Fr::zero();
            v.l[0] = (uint32_t)pos_lo; v.l[1] = (uint32_t)(pos_lo >> 32);
            v.l[2] = (uint32_t)pos_hi; v.l[3] = (uint32_t)(pos_hi >> 32);
            acc = acc + FrD::to_mont(FrD::load_canon(&v));
        }
        if (neg_lo | neg_hi) {
            Fr v = Fr::zero();
            v.l[0] = (uint32_t)neg_lo; v.l[1] = (uint32_t)(neg_lo >> 32);
            v.l[2] = (uint32_t)neg_hi; v.l[3] = (uint32_t)(neg_hi >> 32);
            acc = FrD::sub<2>(acc, FrD::to_mont(FrD::load_canon(&v)));
        }
    }
    if (blockIdx.y == 2 && M.c_nat) {   // (uniform over the workgroup)
        Fr cv;
        acc.reduce().from_mont().store(&cv);
        store_folded_c(M.c_nat, M.c_cls, row, (uint64_t)1 << log_m, cv);
        if (!M.c_digits) return;
        // guarded: count the digits (a row equal to 2 is a one on the doubled bases: no digit)
        const bool two = cv.l[0] == 2 && (cv.l[1] | cv.l[2] | cv.l[3] | cv.l[4] | cv.l[5] | cv.l[6] | cv.l[7]) == 0;
        uint32_t cnt = two ? 0u : digits16(cv);
        __shared__ uint32_t part[4];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d);
        if ((t & 63) == 0) part[t >> 6] = cnt;
        __syncthreads();
        if (t == 0 && (part[0] | part[1] | part[2] | part[3]))
            atomicAdd(&M.c_digits[blockIdx.x & (DIGIT_SLOTS - 1)], (unsigned long long)part[0] + part[1] + part[2] + part[3]);
        return;
    }
    acc.reduce().store(&tile[t]);
    __syncthreads();
    const uint32_t src = (rev4(t & 15) << 4) | rev4(t >> 4);
    const uint32_t rmid = log_m > 8 ? (__brev(mid) >> (32 - (log_m - 8))) : 0u;
    const uint64_t dst = ((uint64_t)(t >> 4) << (log_m - 4)) | ((uint64_t)rmid << 4) | (t & 15);
    out[dst] = tile[src];
}

// One wave per long row (a 32-bit adder's linear row: ~200 terms, all of them bit x 2^s): a thread walking it alone
// is 200 dependent col -> class gathers, and a third of the waves of a SHA-256 circuit contain such a row.  The lanes
// take the terms 64 at a time, sum their share as the tiled kernel does, and the 64 partial sums are added in the field.
__global__ void __launch_bounds__(256) k_r1cs_eval_long(const uint32_t* __restrict__ rows, uint32_t n_long,
                                                        const uint64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                        const Fr* __restrict__ coeff, const uint8_t* __restrict__ ccls,
                                                        const Fr* __restrict__ w, const uint8_t* __restrict__ wcls,
                                                        Fr* __restrict__ out, int log_m, Fr* __restrict__ c_nat,
                                                        uint8_t* __restrict__ c_cls, unsigned long long* __restrict__ c_digits) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_long) return;
    const uint64_t row = rows[wave];
    FrD acc = FrD::zero();
    uint64_t pos_lo = 0, pos_hi = 0, neg_lo = 0, neg_hi = 0;
    uint32_t since = 0;
    for (uint64_t e = row_ptr[row] + lane, end = row_ptr[row + 1]; e < end; e += 64) {
        const uint32_t c = col[e];
        const uint32_t wc = wcls[c];
        if (wc == 0) continue;
        const uint32_t k = ccls[e];
        if (wc == 1 && k != 0) {
            const uint64_t bit = 1ull << ((k - 1) & 63);
            if (k <= 64) {
                pos_lo += bit;
                pos_hi += pos_lo < bit;
            } else {
                neg_lo += bit;
                neg_hi += neg_lo < bit;
            }
            continue;
        }
        acc = acc + FrD::load_canon(&coeff[e]) * FrD::load_canon(&w[c]);
        if (++since == 64) {
            acc = acc.reduce();
            since = 0;
        }
    }
    acc = acc.reduce();
    if (pos_lo | pos_hi) {
        Fr v = Fr::zero();
        v.l[0] = (uint32_t)pos_lo; v.l[1] = (uint32_t)(pos_lo >> 32);
        v.l[2] = (uint32_t)pos_hi; v.l[3] = (uint32_t)(pos_hi >> 32);
        acc = acc + FrD::to_mont(FrD::load_canon(&v));
    }
    if (neg_lo | neg_hi) {
        Fr v = Fr::zero();
        v.l[0] = (uint32_t)neg_lo; v.l[1] = (uint32_t)(neg_lo >> 32);
        v.l[2] = (uint32_t)neg_hi; v.l[3] = (uint32_t)(neg_hi >> 32);
        acc = FrD::sub<2>(acc, FrD::to_mont(FrD::load_canon(&v)));
    }
    acc = acc.reduce();
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {   // field sum over the wave
        FrD o;
#pragma unroll
        for (int q = 0; q < 9; q++) o.l[q] = __shfl_down(acc.l[q], d);
#ifdef ZG_FD_DEBUG
        o.ub = acc.ub; o.lm = acc.lm;
#endif
        acc = (acc + o).reduce();
    }
    if (lane == 0 && c_nat) {   // (R1csMats::c_nat)
        Fr cv;
        acc.from_mont().store(&cv);
        store_folded_c(c_nat, c_cls, row, (uint64_t)1 << log_m, cv);
        if (c_digits) {   // (R1csMats::c_digits: the tiled kernel counted this row as zero)
            const bool two = cv.l[0] == 2 && (cv.l[1] | cv.l[2] | cv.l[3] | cv.l[4] | cv.l[5] | cv.l[6] | cv.l[7]) == 0;
            const uint32_t cnt = two ? 0u : digits16(cv);
            if (cnt) atomicAdd(&c_digits[wave & (DIGIT_SLOTS - 1)], (unsigned long long)cnt);
        }
    }
    if (lane == 0 && !c_nat) {
        // the tiled kernel's destination: row = (ta, mid, tb) -> (ta, bitrev(mid), tb) with ta / tb swapped and reversed
        const uint32_t ta = (uint32_t)(row >> (log_m - 4)), tb = (uint32_t)(row & 15);
        const uint32_t mid = (uint32_t)((row >> 4) & (((uint64_t)1 << (log_m - 8)) - 1));
        const uint32_t rmid = log_m > 8 ? (__brev(mid) >> (32 - (log_m - 8))) : 0u;
        const uint64_t dst = ((uint64_t)rev4(tb) << (log_m - 4)) | ((uint64_t)rmid << 4) | rev4(ta);
        acc.store(&out[dst]);
    }
}

int spmv_dev(Ctx* ctx, const uint64_t* d_ptr, const uint32_t* d_idx, const Fr* d_coeff, const Fr* d_x, uint64_t n_rows,
             Fr* d_out) {
    if (n_rows == 0) return ZG_OK;
    ZG_LAUNCH(ctx, k_r1cs_eval, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, ctx->stream, d_ptr, d_idx,
                       d_coeff, d_x, n_rows, d_out, 0);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

}  // namespace zg

using namespace zg;

// ------------------------------------------------------------------------------------------------ ctx
extern "C" int zg_init(int device, zg_ctx** out) try {
    if (!out) return ZG_E_INVALID_ARG;
    *out = nullptr;
    // five MSM streams + the main stream want their own hardware queues (ROCm's default is 4 per
    // process); only effective if this is the first HIP call of the process, never overrides the user
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return ZG_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return ZG_E_NO_DEVICE;
    zg_ctx* c = new zg_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return ZG_E_HIP;
    }
    c->stream = c->own_stream;
    g_live_ctx.fetch_add(1);
    // 70-80 ms of stream creation beside the caller's next steps instead of inside its first proof.  One prove() of the outer
    // boundary at 2^22, six alternated pairs of processes: 0.717 -> 0.674 s (the proof 0.106 -> 0.032; the key read beside it
    // 0.348 -> 0.393: the runtime serialises part of its calls with the creation)
    msm_side_streams_begin(c);
    *out = c;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
extern "C" void zg_free(zg_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ntt_tables_free(ctx);
    msm_scratch_free(ctx);
    for (auto& p : ctx->prof_pending) {
        if (p.e0) (void)hipEventDestroy(p.e0);
        if (p.e1) (void)hipEventDestroy(p.e1);
    }
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    for (auto& e : ctx->tm_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->h_split_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->gate_w) (void)hipEventDestroy(ctx->gate_w);
    if (ctx->gate_h) (void)hipEventDestroy(ctx->gate_h);
    delete ctx->prove_ws;
    ctx->prove_ws = nullptr;
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    g_live_ctx.fetch_sub(1);
    delete ctx;
}
extern "C" const char* zg_strerror(int code) {
    switch (code) {
        case ZG_OK: return "ok";
        case ZG_E_ASSIGNMENT_MISSING: return "AssignmentMissing";
        case ZG_E_DIVISION_BY_ZERO: return "DivisionByZero";
        case ZG_E_UNSATISFIABLE: return "Unsatisfiable";
        case ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE: return "PolynomialDegreeTooLarge";
        case ZG_E_UNEXPECTED_IDENTITY: return "UnexpectedIdentity";
        case ZG_E_IO: return "IoError";
        case ZG_E_MALFORMED_VERIFYING_KEY: return "MalformedVerifyingKey";
        case ZG_E_UNCONSTRAINED_VARIABLE: return "UnconstrainedVariable";
        case ZG_E_BAD_FORMAT: return "BadFormat";
        case ZG_E_QUERY_SIZE: return "query size does not match density";
        case ZG_E_INVALID_ARG: return "invalid argument";
        case ZG_E_BUFFER_TOO_SMALL: return "buffer too small";
        case ZG_E_HIP: return "HIP runtime error";
        case ZG_E_NO_DEVICE: return "no MI355X/HIP device available (no CPU fallback)";
        default: return "unknown error";
    }
}
extern "C" const char* zg_last_error(const zg_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
extern "C" const char* zg_last_error_thread(void) { return t_last_error.c_str(); }
extern "C" int zg_set_stream(zg_ctx* ctx, void* hip_stream) try {
    if (!ctx) return ZG_E_INVALID_ARG;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_set_option(zg_ctx* ctx, int option, int64_t value) try {
    if (!ctx) return ZG_E_INVALID_ARG;
    switch (option) {
        case ZG_OPT_ONE_SHOT: ctx->one_shot = value != 0; return ZG_OK;
        case ZG_OPT_PACK:
            if (value < 0 || value > 2) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_PACK takes 0, 1 or 2");
            ctx->opt_pack = (int)value;
            return ZG_OK;
        case ZG_OPT_PACK_THREADS:
            if (value < 0 || value > 1024) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_PACK_THREADS: 0 .. 1024");
            ctx->opt_pack_threads = (int)value;
            return ZG_OK;
        case ZG_OPT_PACK_MIN_LOG_N:
            if (value < 10 || value > 32) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_PACK_MIN_LOG_N: 10 .. 32");
            ctx->opt_pack_min_log_n = (int)value;
            return ZG_OK;
        case ZG_OPT_PACK_PIN: ctx->opt_pack_pin = value != 0; return ZG_OK;
        case ZG_OPT_PACK_BUDGET_PERMILLE:
            if (value < 0 || value > 100000) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_PACK_BUDGET_PERMILLE: 0 .. 100000");
            ctx->opt_pack_budget_permille = (int)value;
            return ZG_OK;
        case ZG_OPT_PACK_RAW_PERMILLE:
            if (value < 0 || value > 900) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_PACK_RAW_PERMILLE: 0 .. 900");
            ctx->opt_pack_raw_permille = (int)value;
            return ZG_OK;
        case ZG_OPT_COMBINE_EARLY: ctx->opt_combine_early = value != 0; return ZG_OK;
        case ZG_OPT_H_EVAL: ctx->opt_h_eval = value != 0; return ZG_OK;
        case ZG_OPT_TABLES:
            if (value < 0 || value > 31) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_TABLES: a mask over five queries");
            ctx->opt_tables = (unsigned)value;
            return ZG_OK;
        case ZG_OPT_TABLES_MAX_MB:
            if (value < 0) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_TABLES_MAX_MB is negative");
            ctx->opt_tables_max_mb = (uint64_t)value;
            return ZG_OK;
        case ZG_OPT_SMALL_LOG_N:
            if (value < 0 || value > 24) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_SMALL_LOG_N: 0 .. 24");
            ctx->opt_small_log_n = (int)value;
            return ZG_OK;
        case ZG_OPT_SMALL_W_LOG_N:
            if (value < -1 || value > 24) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_SMALL_W_LOG_N: -1 .. 24");
            ctx->opt_small_w_log_n = (int)value;
            return ZG_OK;
        case ZG_OPT_SORT_STAGED:
            if (value != -1 && value != 0 && value != 2 && value != 3) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_SORT_STAGED takes -1, 0, 2 or 3");
            ctx->opt_sort_staged = (int)value;
            return ZG_OK;
        case ZG_OPT_PART_STAGE_MIN:
            if (value < -1 || value > 0xffffffffll) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_PART_STAGE_MIN: -1 .. 2^32 - 1");
            ctx->opt_part_stage_min = value;
            return ZG_OK;
        case ZG_OPT_SUBSET_W:
            if (value != 4 && value != 8) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_SUBSET_W takes 4 or 8");
            ctx->opt_subset_w = (int)value;
            return ZG_OK;
        case ZG_OPT_COMPACT_TABLE: ctx->opt_compact_table = value != 0; return ZG_OK;
        case ZG_OPT_H_PARTS:
            if (value < 0 || value > Ctx::H_PARTS_MAX) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_H_PARTS: 0 .. %d", (int)Ctx::H_PARTS_MAX);
            ctx->opt_h_parts = (int)value;
            return ZG_OK;
        case ZG_OPT_H_PIPE:
            if (value < -1 || value > 1) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_H_PIPE takes -1, 0 or 1");
            ctx->opt_h_pipe = (int)value;
            return ZG_OK;
        case ZG_OPT_NTT_BATCH_LOG_M:
            if (value < 0 || value > 28) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_NTT_BATCH_LOG_M: 0 .. 28");
            ctx->opt_ntt_batch_log_m = (int)value;
            return ZG_OK;
        case ZG_OPT_SHARE_SORT:
            if (value < 0 || value > 2) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_SHARE_SORT takes 0, 1 or 2");
            ctx->opt_share_sort = (int)value;
            return ZG_OK;
        case ZG_OPT_C_FOLD:
            if (value < 0 || value > 2) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_C_FOLD takes 0, 1 or 2");
            ctx->opt_c_fold = (int)value;
            return ZG_OK;
        case ZG_OPT_C_FOLD_MIN_LOG_M:
            if (value < 0 || value > 28) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_C_FOLD_MIN_LOG_M: 0 .. 28");
            ctx->opt_c_fold_min_log_m = (int)value;
            return ZG_OK;
        case ZG_OPT_UPLOAD_GATES:
            if (value < 0 || value > 31) return fail(ctx, ZG_E_INVALID_ARG, "ZG_OPT_UPLOAD_GATES: 0 (off) or the log2 of the smallest witness, up to 31");
            ctx->opt_upload_gates = (int)value;
            return ZG_OK;
        default: return fail(ctx, ZG_E_INVALID_ARG, "unknown option %d", option);
    }
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_sync(zg_ctx* ctx) try {
    if (!ctx) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_dev_alloc(zg_ctx* ctx, size_t bytes, void** d_ptr) try {
    if (!ctx || !d_ptr) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    ZG_HIP(ctx, hipMalloc(d_ptr, bytes ? bytes : 16));
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_dev_free(zg_ctx* ctx, void* d_ptr) try {
    if (!ctx) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ZG_HIP(ctx, hipFree(d_ptr));
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_h2d(zg_ctx* ctx, void* d_dst, const void* src, size_t bytes) try {
    if (!ctx) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_d2h(zg_ctx* ctx, void* dst, const void* d_src, size_t bytes) try {
    if (!ctx) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_last_stats(const zg_ctx* ctx, uint64_t stats[4]) try {
    if (!ctx || !stats) return ZG_E_INVALID_ARG;
    stats[0] = ctx->n_launches;
    stats[1] = ctx->last_share_mask | (ctx->last_l_parts > 1 ? 0x100u : 0u);
    stats[2] = ctx->last_follow_g2 ? 1 : 0;
    stats[3] = ctx->last_c_fold ? 1 : 0;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
extern "C" int zg_last_timings(const zg_ctx* ctx, double ms[6]) try {
    if (!ctx || !ms) return ZG_E_INVALID_ARG;
    for (int i = 0; i < 6; i++) ms[i] = ctx->timings[i];
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
extern "C" int zg_last_load_timings(const zg_ctx* ctx, double ms[4]) try {
    if (!ctx || !ms) return ZG_E_INVALID_ARG;
    for (int i = 0; i < 4; i++) ms[i] = ctx->load_timings[i];
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

// ------------------------------------------------------------------------------------------------ profiling
extern "C" int zg_profile_enable(zg_ctx* ctx, int on) try {
    if (!ctx) return ZG_E_INVALID_ARG;
    ctx->prof_enabled = on < 0 ? 0 : on > 2 ? 2 : on;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
// Waits for the pending scopes' OWN end events, not for the context's stream: zg_prove_dev recycles the previous proof's
// events (long complete) when the next proof's witness is already on its way on that stream -- a stream synchronisation there
// made every profiled proof wait for its upload before anything else was queued (2^24, field-element witness: 119 -> 123 ms
// with the profile on, found in round 6; the scopes on the side streams were never covered by it anyway).
static int prof_drain(zg_ctx* ctx) {
    for (auto& p : ctx->prof_pending) {
        float ms = 0;
        if (p.e0 && p.e1 && hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            ctx->prof[p.id].ms += ms;
            ctx->prof[p.id].launches += 1;
            ctx->prof[p.id].bytes += p.bytes;
            ctx->prof[p.id].units += p.units;
        }
        if (p.e0) ctx->prof_pool.push_back(p.e0);
        if (p.e1) ctx->prof_pool.push_back(p.e1);
    }
    ctx->prof_pending.clear();
    return ZG_OK;
}
extern "C" int zg_profile_reset(zg_ctx* ctx) try {
    if (!ctx) return ZG_E_INVALID_ARG;
    ZG_TRY(prof_drain(ctx));
    for (auto& s : ctx->prof) s = zg::Ctx::ProfStat();
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_profile_get(zg_ctx* ctx, int id, double* ms, uint64_t* launches, uint64_t* alg_bytes) try {
    if (!ctx || id < 0 || id >= PROF_COUNT || !ms || !launches || !alg_bytes) return ZG_E_INVALID_ARG;
    ZG_TRY(prof_drain(ctx));
    *ms = ctx->prof[id].ms;
    *launches = ctx->prof[id].launches;
    *alg_bytes = ctx->prof[id].bytes;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_profile_get_units(zg_ctx* ctx, int id, uint64_t* units) try {
    if (!ctx || id < 0 || id >= PROF_COUNT || !units) return ZG_E_INVALID_ARG;
    ZG_TRY(prof_drain(ctx));
    *units = ctx->prof[id].units;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
static_assert(ZG_PROFILE_KERNELS == PROF_COUNT, "include/za_amd.h lists every profile class");
extern "C" const char* zg_profile_name(int id) {
    static const char* names[] = {"msm_accumulate_g1", "msm_accumulate_g2", "msm_digit_sort", "msm_bucket_reduce",
                                  "ntt_pass", "ntt_pointwise", "r1cs_eval", "msm_accumulate_g1_dense",
                                  "msm_digit_sort_dense", "msm_bucket_reduce_dense"};
    return id >= 0 && id < PROF_COUNT ? names[id] : "";
}

// ------------------------------------------------------------------------------------------------ point <-> bytes
template <class F>
static void jac_bytes(uint8_t* out, const XYZZ<F>& p) {
    constexpr size_t FB = sizeof(F);
    if (p.is_inf()) {
        memset(out, 0, 3 * FB);
        return;
    }
    Affine<F> a = p.to_affine();
    F one = F::one();
    memcpy(out, &a.x, FB);
    memcpy(out + FB, &a.y, FB);
    memcpy(out + 2 * FB, &one, FB);
}
template <class F>
static XYZZ<F> jac_from_bytes(const uint8_t* in) {
    constexpr size_t FB = sizeof(F);
    F x, y, z;
    memcpy(&x, in, FB);
    memcpy(&y, in + FB, FB);
    memcpy(&z, in + 2 * FB, FB);
    if (z.is_zero()) return XYZZ<F>::inf();
    // general Jacobian (X, Y, Z) -> XYZZ (X, Y, Z^2, Z^3)
    F zz = z.sqr();
    return XYZZ<F>{x, y, zz, zz * z};
}

// ------------------------------------------------------------------------------------------------ MSM API
template <class F>
static int msm_host(zg_ctx* ctx, const uint8_t* bases, size_t n_bases, const uint8_t* scalars,
                    const uint64_t* density_bits, size_t n_scalars, uint8_t* out_jac) {
    if (!ctx || !out_jac || (n_bases && !bases) || (n_scalars && !scalars)) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> index;
    if (density_bits) {
        index.reserve(n_bases);
        for (size_t i = 0; i < n_scalars; i++)
            if ((density_bits[i >> 6] >> (i & 63)) & 1) index.push_back((uint32_t)i);
        if (index.size() != n_bases)
            return fail(ctx, ZG_E_QUERY_SIZE, "density total %zu != number of bases %zu", index.size(), n_bases);
    } else if (n_bases != n_scalars) {
        return fail(ctx, ZG_E_QUERY_SIZE, "full density: %zu scalars but %zu bases", n_scalars, n_bases);
    }
    DevBuf db, ds, di;
    ZG_HIP(ctx, db.alloc(n_bases * sizeof(Affine<F>)));
    ZG_HIP(ctx, ds.alloc(n_scalars * 32));
    ZG_HIP(ctx, hipMemcpyAsync(db.p, bases, n_bases * sizeof(Affine<F>), hipMemcpyHostToDevice, ctx->stream));
    ZG_HIP(ctx, hipMemcpyAsync(ds.p, scalars, n_scalars * 32, hipMemcpyHostToDevice, ctx->stream));
    if (density_bits) {
        ZG_HIP(ctx, di.alloc(index.size() * 4));
        ZG_HIP(ctx, hipMemcpyAsync(di.p, index.data(), index.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    ZG_TRY(convert_mont_dev(ctx, db.p, n_bases * sizeof(Affine<F>) / 32, 0, 0));  // ABI form -> device form
    XYZZ<F> r;
    int rc;
    if constexpr (sizeof(F) == sizeof(Fq))
        rc = msm_g1(ctx, db.as<G1Affine>(), ds.as<uint32_t>(), density_bits ? di.as<uint32_t>() : nullptr, n_bases, &r);
    else
        rc = msm_g2(ctx, db.as<G2Affine>(), ds.as<uint32_t>(), density_bits ? di.as<uint32_t>() : nullptr, n_bases, &r);
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (rc) return rc;
    jac_bytes(out_jac, r);
    return ZG_OK;
}
extern "C" int zg_msm_g1(zg_ctx* ctx, const uint8_t* bases64, size_t n_bases, const uint8_t* scalars32,
                         const uint64_t* density_bits, size_t n_scalars, uint8_t out_jac[96]) try {
    return msm_host<Fq>(ctx, bases64, n_bases, scalars32, density_bits, n_scalars, out_jac);
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_msm_g2(zg_ctx* ctx, const uint8_t* bases128, size_t n_bases, const uint8_t* scalars32,
                         const uint64_t* density_bits, size_t n_scalars, uint8_t out_jac[192]) try {
    return msm_host<Fq2>(ctx, bases128, n_bases, scalars32, density_bits, n_scalars, out_jac);
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_msm_g1_dev(zg_ctx* ctx, const void* d_bases, const void* d_scalars, const uint32_t* d_scalar_index,
                             size_t n, uint8_t out_jac[96]) try {
    if (!ctx || !out_jac) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    G1XYZZ r;
    DevBuf tmp;  // the caller's bases are in the ABI's 2^256 form: work on a converted copy
    ZG_HIP(ctx, tmp.alloc(n * sizeof(G1Affine)));
    ZG_HIP(ctx, hipMemcpyAsync(tmp.p, d_bases, n * sizeof(G1Affine), hipMemcpyDeviceToDevice, ctx->stream));
    ZG_TRY(convert_mont_dev(ctx, tmp.p, n * 2, 0, 0));
    ZG_TRY(msm_g1(ctx, tmp.as<G1Affine>(), (const uint32_t*)d_scalars, d_scalar_index, n, &r));
    jac_bytes(out_jac, r);
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_msm_g2_dev(zg_ctx* ctx, const void* d_bases, const void* d_scalars, const uint32_t* d_scalar_index,
                             size_t n, uint8_t out_jac[192]) try {
    if (!ctx || !out_jac) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    G2XYZZ r;
    DevBuf tmp;
    ZG_HIP(ctx, tmp.alloc(n * sizeof(G2Affine)));
    ZG_HIP(ctx, hipMemcpyAsync(tmp.p, d_bases, n * sizeof(G2Affine), hipMemcpyDeviceToDevice, ctx->stream));
    ZG_TRY(convert_mont_dev(ctx, tmp.p, n * 4, 0, 0));
    ZG_TRY(msm_g2(ctx, tmp.as<G2Affine>(), (const uint32_t*)d_scalars, d_scalar_index, n, &r));
    jac_bytes(out_jac, r);
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

// ---- resident bases (the second half of BASELINE.json's metric: G1 MSM points/s with the bases resident)
template <class F>
static int msm_bases_make(zg_ctx* ctx, const void* src, bool src_is_device, size_t n, double table_budget_gb,
                          zg_msm_bases** out) {
    if (!ctx || !out || (n && !src)) return ZG_E_INVALID_ARG;
    *out = nullptr;
    if (n >= (1ull << 31)) return fail(ctx, ZG_E_INVALID_ARG, "msm bases: n = %zu exceeds 2^31-1", n);
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<zg_msm_bases> b(new zg_msm_bases());
    b->device = ctx->device;
    b->g2 = sizeof(F) != sizeof(Fq);
    b->n = n;
    ZG_HIP(ctx, b->bases.alloc(n * sizeof(Affine<F>)));
    if (n) {
        ZG_HIP(ctx, hipMemcpyAsync(b->bases.p, src, n * sizeof(Affine<F>), src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                   ctx->stream));
        ZG_TRY(convert_mont_dev(ctx, b->bases.p, n * sizeof(Affine<F>) / 32, 0, 0));   // ABI form -> device form, once
    }
    int c = 0, nwin = 0;
    // (a negative budget: that much, and the table is built even where the cost model sees less than 8 % in it)
    const double budget_gb = table_budget_gb < 0 ? -table_budget_gb : table_budget_gb;
    if (n >= 64 && budget_gb > 0 &&
        msm_fbw_plan(n, n, (size_t)(budget_gb * 1e9), sizeof(Affine<F>), &c, &nwin, false, table_budget_gb < 0) &&
        b->table.alloc((size_t)nwin * n * sizeof(Affine<F>)) == hipSuccess) {
        if constexpr (sizeof(F) == sizeof(Fq))
            ZG_TRY(msm_fbw_build_g1(ctx, b->bases.as<G1Affine>(), n, n, c, nwin, b->table.as<G1Affine>()));
        else
            ZG_TRY(msm_fbw_build_g2(ctx, b->bases.as<G2Affine>(), n, n, c, nwin, b->table.as<G2Affine>()));
        b->c = c;
        b->nwin = nwin;
    } else {
        (void)hipGetLastError();
    }
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = b.release();
    return ZG_OK;
}
extern "C" int zg_msm_bases_g1(zg_ctx* ctx, const uint8_t* bases64, size_t n, double table_budget_gb, zg_msm_bases** out) try {
    return msm_bases_make<Fq>(ctx, bases64, false, n, table_budget_gb, out);
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_msm_bases_g1_dev(zg_ctx* ctx, const void* d_bases, size_t n, double table_budget_gb, zg_msm_bases** out) try {
    return msm_bases_make<Fq>(ctx, d_bases, true, n, table_budget_gb, out);
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_msm_bases_g2(zg_ctx* ctx, const uint8_t* bases128, size_t n, double table_budget_gb, zg_msm_bases** out) try {
    return msm_bases_make<Fq2>(ctx, bases128, false, n, table_budget_gb, out);
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" void zg_msm_bases_free(zg_msm_bases* b) { delete b; }
extern "C" int zg_msm_bases_info(const zg_msm_bases* b, uint64_t info[4]) {
    if (!b || !info) return ZG_E_INVALID_ARG;
    info[0] = b->n; info[1] = b->table.bytes; info[2] = (uint64_t)b->c; info[3] = (uint64_t)b->nwin;
    return ZG_OK;
}
// one multiexp over resident bases; `gates` (nullable) = per-part events for chunked scalar uploads
static int msm_resident_run(zg_ctx* ctx, const zg_msm_bases* b, const void* d_scalars, size_t n, int dense_hint, int parts,
                            const hipEvent_t* gates, uint8_t* out_jac, const size_t* part_bounds = nullptr) {
    MsmDesc d1[4], d2;
    for (int k = 0; k < 4; k++) d1[k] = MsmDesc{nullptr, nullptr, nullptr, 0};
    d2 = MsmDesc{nullptr, nullptr, nullptr, 0};
    MsmDesc d{b->bases.p, (const uint32_t*)d_scalars, nullptr, n, dense_hint != 0};
    if (b->table.p && n == b->n) {
        d.fbw_table = b->table.p;
        d.fbw_c = b->c;
        d.fbw_nwin = b->nwin;
        d.fbw_stride = (size_t)b->n;
    }
    d.part_gate = gates;
    d.parts = parts;
    d.part_bounds = part_bounds;
    G1XYZZ o1[4];
    G2XYZZ o2;
    if (!b->g2) {
        d1[0] = d;
        ZG_TRY(msm_batch5(ctx, d1, d2, o1, &o2, nullptr, nullptr));
        jac_bytes(out_jac, o1[0]);
    } else {
        d2 = d;
        ZG_TRY(msm_batch5(ctx, d1, d2, o1, &o2, nullptr, nullptr));
        jac_bytes(out_jac, o2);
    }
    return ZG_OK;
}
extern "C" int zg_msm_resident_dev(zg_ctx* ctx, const zg_msm_bases* b, const void* d_scalars, size_t n, int dense_hint,
                                   uint8_t* out_jac) try {
    if (!ctx || !b || !out_jac || (n && !d_scalars) || n > b->n) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    return msm_resident_run(ctx, b, d_scalars, n, dense_hint, 0, nullptr, out_jac);
} catch (...) {
    return zg::on_exception(ctx);
}
// scalars in host memory (pinned, see zg_host_alloc, for full PCIe speed): uploaded in up to four chunks on a copy
// stream; the multiexp runs in as many dependent parts over consecutive base ranges, each gated on its chunk, so the
// upload of chunk k+1 overlaps the digit sort and accumulation of chunk k
extern "C" int zg_msm_resident(zg_ctx* ctx, zg_msm_bases* b, const uint8_t* scalars32, size_t n, int dense_hint,
                               uint8_t* out_jac) try {
    if (!ctx || !b || !out_jac || (n && !scalars32) || n > b->n) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    ZG_HIP(ctx, b->scalars.reserve(n * 32));
    if (!b->copy_stream) {
        ZG_HIP(ctx, hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
        for (auto& e : b->chunk_ev) ZG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    int parts = 1;
    if (!b->g2 && dense_hint && n >= ((size_t)1 << 22)) parts = 4;
    // Chunks of 1 : 2 : 4 : 8.  Nothing can start before the first chunk is there, so it is small (1/15 of the scalars:
    // 2.4 ms of PCIe at 2^26 instead of 9); and since a part computes ~2 x longer than its chunk uploads, a chunk twice the
    // size of the one before still arrives before the part before it is done.
    size_t bounds[5] = {0, n, n, n, n};
    if (parts == 4) {
        const int w[4] = {1, 2, 4, 8};
        const int tot = w[0] + w[1] + w[2] + w[3];
        size_t acc = 0;
        for (int k = 0; k < 3; k++) {
            acc += (size_t)w[k];
            bounds[k + 1] = std::min(n, (n / (size_t)tot * acc + 4095) & ~(size_t)4095);
        }
    }
    // the copy stream must not overtake earlier work of this ctx that still reads the staging buffer
    ZG_HIP(ctx, hipEventRecord(b->chunk_ev[0], ctx->stream));
    ZG_HIP(ctx, hipStreamWaitEvent(b->copy_stream, b->chunk_ev[0], 0));
    for (int k = 0; k < parts; k++) {
        const size_t lo = bounds[k], hi = bounds[k + 1];
        if (hi > lo)
            ZG_HIP(ctx, hipMemcpyAsync((uint8_t*)b->scalars.p + lo * 32, scalars32 + lo * 32, (hi - lo) * 32, hipMemcpyHostToDevice,
                                       b->copy_stream));
        ZG_HIP(ctx, hipEventRecord(b->chunk_ev[k], b->copy_stream));
    }
    if (parts == 1) ZG_HIP(ctx, hipStreamWaitEvent(ctx->stream, b->chunk_ev[0], 0));
    return msm_resident_run(ctx, b, b->scalars.p, n, dense_hint, parts, parts > 1 ? b->chunk_ev : nullptr, out_jac,
                            parts > 1 ? bounds : nullptr);
} catch (...) {
    return zg::on_exception(ctx);
}
// pinned host memory: H2D copies from it run at PCIe speed (a pageable buffer is staged by the runtime)
extern "C" int zg_host_alloc(zg_ctx* ctx, size_t bytes, void** out) try {
    if (!ctx || !out) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    ZG_HIP(ctx, hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_host_free(zg_ctx* ctx, void* p) try {
    if (!ctx) return ZG_E_INVALID_ARG;
    if (p) ZG_HIP(ctx, hipHostFree(p));
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

// ------------------------------------------------------------------------------------------------ NTT / H API
extern "C" int zg_ntt_fr_dev(zg_ctx* ctx, void* d_inout, uint32_t log_m, int inverse, int coset) try {
    if (!ctx || !d_inout) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    return ntt_fr_dev(ctx, (Fr*)d_inout, log_m, inverse != 0, coset != 0);
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_ntt_fr(zg_ctx* ctx, uint8_t* inout, uint32_t log_m, int inverse, int coset) try {
    if (!ctx || !inout) return ZG_E_INVALID_ARG;
    if (log_m >= 28) return fail(ctx, ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE, "domain 2^%u >= 2^28", log_m);
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)32 << log_m;
    DevBuf d;
    ZG_HIP(ctx, d.alloc(bytes));
    ZG_HIP(ctx, hipMemcpyAsync(d.p, inout, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZG_TRY(ntt_fr_dev(ctx, d.as<Fr>(), log_m, inverse != 0, coset != 0));
    ZG_HIP(ctx, hipMemcpyAsync(inout, d.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_h_scalars(zg_ctx* ctx, const uint8_t* a, const uint8_t* b, const uint8_t* c, size_t n_rows,
                            uint32_t log_m, uint8_t* h_out) try {
    if (!ctx || !a || !b || !c || !h_out) return ZG_E_INVALID_ARG;
    if (log_m >= 28) return fail(ctx, ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE, "domain 2^%u >= 2^28", log_m);
    const size_t m = (size_t)1 << log_m;
    if (n_rows > m) return fail(ctx, ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE, "%zu rows do not fit a 2^%u domain", n_rows, log_m);
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf d[3], dh;
    const uint8_t* src[3] = {a, b, c};
    for (int k = 0; k < 3; k++) {
        ZG_HIP(ctx, d[k].alloc(m * 32));
        ZG_HIP(ctx, hipMemsetAsync(d[k].p, 0, m * 32, ctx->stream));
        ZG_HIP(ctx, hipMemcpyAsync(d[k].p, src[k], n_rows * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    ZG_HIP(ctx, dh.alloc(m * 32));
    for (int k = 0; k < 3; k++) {
        ZG_TRY(convert_mont_dev(ctx, d[k].p, n_rows, 1, 0));   // ABI form -> device form
        ZG_TRY(bitrev_dev(ctx, d[k].as<Fr>(), log_m));         // h_poly_dev consumes bit-reversed rows
    }
    ZG_TRY(h_poly_dev(ctx, d[0].as<Fr>(), d[1].as<Fr>(), d[2].as<Fr>(), log_m, dh.as<Fr>()));
    if (m > 1) ZG_HIP(ctx, hipMemcpyAsync(h_out, dh.p, (m - 1) * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

// ------------------------------------------------------------------------------------------------ R1CS
// rank[index[i]] = i (the inverse of a query's index list; the other entries stay ~0)
__global__ void __launch_bounds__(256) k_rank_of_index(const uint32_t* __restrict__ index, uint32_t n, uint32_t* __restrict__ rank) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rank[index[i]] = i;
}
extern "C" int zg_r1cs_load(zg_ctx* ctx, uint32_t n_inputs, uint32_t n_aux, uint64_t n_constraints,
                            const uint64_t* const row_ptr[3], const uint32_t* const col[3],
                            const uint8_t* const coeff[3], zg_r1cs** out) try {
    if (!ctx || !out || !row_ptr || !col || !coeff || n_inputs == 0) return ZG_E_INVALID_ARG;
    *out = nullptr;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t rows = n_constraints + n_inputs;
    uint32_t log_m = 0;
    while (((uint64_t)1 << log_m) < rows) log_m++;
    if (log_m >= 28)
        return fail(ctx, ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE, "%llu constraints + %u inputs need a domain >= 2^28",
                    (unsigned long long)n_constraints, n_inputs);
    const uint64_t nvars = (uint64_t)n_inputs + n_aux;
    // The host passes over the CSR arrays (range check, densities by mention, index lists, long rows) run on a few threads from
    // 2^16 entries: single-threaded they were 60 of the 105 ms this call took for the 2^22 circuit -- inside every prove() of
    // the outer boundary, which loads the key's constraints anew (za2c.cpp).
    for (int k = 0; k < 3; k++) {
        if (!row_ptr[k]) return ZG_E_INVALID_ARG;
        const uint64_t nnz = row_ptr[k][n_constraints];
        if (nnz && (!col[k] || !coeff[k])) return ZG_E_INVALID_ARG;
    }
    constexpr size_t HCH = (size_t)1 << 16;   // entries per task
    for (int k = 0; k < 3; k++) {             // (first offending entry in matrix, then entry order -- as the plain loop found it)
        const uint64_t nnz = row_ptr[k][n_constraints];
        const size_t nt = (size_t)((nnz + HCH - 1) / HCH);
        std::vector<uint64_t> bad(nt, ~(uint64_t)0);
        const uint32_t* ck = col[k];
        host_par_for(nt, [&](size_t t) {
            const uint64_t e1 = std::min<uint64_t>(nnz, (uint64_t)(t + 1) * HCH);
            for (uint64_t e = (uint64_t)t * HCH; e < e1; e++)
                if (ck[e] >= nvars) {
                    bad[t] = e;
                    break;
                }
        });
        for (size_t t = 0; t < nt; t++)
            if (bad[t] != ~(uint64_t)0)
                return fail(ctx, ZG_E_INVALID_ARG, "matrix %d: variable index %u out of range", k, ck[bad[t]]);
    }
    std::unique_ptr<zg_r1cs> r(new zg_r1cs());
    r->n_inputs = n_inputs;
    r->n_aux = n_aux;
    r->n_constraints = n_constraints;
    r->log_m = log_m;
    // densities by MENTION (bellman's DensityTracker::inc in ProvingAssignment::eval)
    // (flag bytes set from several threads: relaxed atomic stores of the same value)
    std::vector<uint8_t> a_aux(n_aux, 0), b_in(n_inputs, 0), b_aux(n_aux, 0);
    for (int k = 0; k < 2; k++) {
        const uint64_t nnz = row_ptr[k][n_constraints];
        const uint32_t* ck = col[k];
        uint8_t* aux = k == 0 ? a_aux.data() : b_aux.data();
        uint8_t* in = k == 0 ? nullptr : b_in.data();
        host_par_for((size_t)((nnz + HCH - 1) / HCH), [&](size_t t) {
            const uint64_t e1 = std::min<uint64_t>(nnz, (uint64_t)(t + 1) * HCH);
            for (uint64_t e = (uint64_t)t * HCH; e < e1; e++) {
                const uint32_t v = ck[e];
                if (v >= n_inputs) __atomic_store_n(&aux[v - n_inputs], (uint8_t)1, __ATOMIC_RELAXED);
                else if (in) __atomic_store_n(&in[v], (uint8_t)1, __ATOMIC_RELAXED);
            }
        });
    }
    // dense position -> variable: the inputs (A: all of them; B: the mentioned ones), then the mentioned aux variables -- counted
    // per slice, then written per slice
    std::vector<uint32_t> a_index, b_index;
    auto build_index = [&](std::vector<uint32_t>& ix, const uint8_t* in_flags, const uint8_t* aux_flags, uint64_t* in_total,
                           uint64_t* aux_total) {
        const size_t nt = (size_t)((nvars + HCH - 1) / HCH);
        std::vector<uint64_t> cnt(nt + 1, 0);
        auto flag = [&](uint64_t v) -> bool { return v < n_inputs ? (in_flags ? in_flags[v] != 0 : true) : aux_flags[v - n_inputs] != 0; };
        host_par_for(nt, [&](size_t t) {
            const uint64_t v1 = std::min<uint64_t>(nvars, (uint64_t)(t + 1) * HCH);
            uint64_t c = 0;
            for (uint64_t v = (uint64_t)t * HCH; v < v1; v++) c += flag(v);
            cnt[t + 1] = c;
        });
        for (size_t t = 0; t < nt; t++) cnt[t + 1] += cnt[t];
        ix.resize(cnt[nt]);
        host_par_for(nt, [&](size_t t) {
            const uint64_t v1 = std::min<uint64_t>(nvars, (uint64_t)(t + 1) * HCH);
            uint64_t o = cnt[t];
            for (uint64_t v = (uint64_t)t * HCH; v < v1; v++)
                if (flag(v)) ix[o++] = (uint32_t)v;
        });
        uint64_t n_in = 0;
        for (uint32_t i = 0; i < n_inputs; i++) n_in += in_flags ? in_flags[i] != 0 : 1;
        if (in_total) *in_total = n_in;
        *aux_total = ix.size() - n_in;
    };
    build_index(a_index, nullptr, a_aux.data(), nullptr, &r->a_aux_total);
    build_index(b_index, b_in.data(), b_aux.data(), &r->b_in_total, &r->b_aux_total);
    r->n_a = a_index.size();
    r->n_b = b_index.size();
    hipStream_t st = ctx->stream;
    ZG_HIP(ctx, r->a_index.alloc(a_index.size() * 4));
    ZG_HIP(ctx, r->b_index.alloc(b_index.size() * 4));
    ZG_HIP(ctx, hipMemcpyAsync(r->a_index.p, a_index.data(), a_index.size() * 4, hipMemcpyHostToDevice, st));
    if (!b_index.empty())
        ZG_HIP(ctx, hipMemcpyAsync(r->b_index.p, b_index.data(), b_index.size() * 4, hipMemcpyHostToDevice, st));
    // the inverse maps (variable -> position in the a / b query, ~0 = not in its density): what a query reads the entries
    // of a shared digit sort through (msm.hip ShQuery); a query of full density needs none.  Scattered on the device from the
    // index lists just uploaded.
    for (int k = 0; k < 2; k++) {
        const std::vector<uint32_t>& ix = k == 0 ? a_index : b_index;
        zg::DevBuf& dst = k == 0 ? r->a_rank : r->b_rank;
        if (ix.size() == nvars) continue;   // the identity
        ZG_HIP(ctx, dst.alloc(nvars * 4));
        ZG_HIP(ctx, hipMemsetAsync(dst.p, 0xff, nvars * 4, st));
        if (!ix.empty())
            ZG_LAUNCH(ctx, k_rank_of_index, dim3((unsigned)((ix.size() + 255) / 256)), dim3(256), 0, st,
                      (k == 0 ? r->a_index : r->b_index).as<uint32_t>(), (uint32_t)ix.size(), dst.as<uint32_t>());
    }
    // the variables this process's shard of a query [lo, hi) spans are needed on the host when a sharded key is used
    r->a_first = a_index.empty() ? 0 : a_index.front();
    r->b_first = b_index.empty() ? 0 : b_index.front();
    r->a_index_host = a_index;   // (4 bytes per dense position, like the device copy)
    r->b_index_host = b_index;
    for (int k = 0; k < 3; k++) {
        uint64_t nnz = row_ptr[k][n_constraints];
        ZG_HIP(ctx, r->row_ptr[k].alloc((n_constraints + 1) * 8));
        ZG_HIP(ctx, r->col[k].alloc(nnz * 4));
        ZG_HIP(ctx, r->coeff[k].alloc(nnz * 32));
        ZG_HIP(ctx, hipMemcpyAsync(r->row_ptr[k].p, row_ptr[k], (n_constraints + 1) * 8, hipMemcpyHostToDevice, st));
        if (nnz) {
            ZG_HIP(ctx, hipMemcpyAsync(r->col[k].p, col[k], nnz * 4, hipMemcpyHostToDevice, st));
            ZG_HIP(ctx, hipMemcpyAsync(r->coeff[k].p, coeff[k], nnz * 32, hipMemcpyHostToDevice, st));
            ZG_HIP(ctx, r->ccls[k].alloc(nnz));
            ZG_LAUNCH(ctx, k_coeff_prepare, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st,
                               r->coeff[k].as<Fr>(), (size_t)nnz, r->ccls[k].as<uint8_t>());
        }
    }
    std::vector<uint32_t> long_rows[3];   // (alive until the synchronize below)
    for (int k = 0; k < 3; k++) {
        const size_t nt = (size_t)((n_constraints + HCH - 1) / HCH);
        std::vector<std::vector<uint32_t>> part(nt);
        const uint64_t* rp = row_ptr[k];
        host_par_for(nt, [&](size_t t) {
            const uint64_t i1 = std::min<uint64_t>(n_constraints, (uint64_t)(t + 1) * HCH);
            for (uint64_t i = (uint64_t)t * HCH; i < i1; i++)
                if (rp[i + 1] - rp[i] > R1CS_LONG_ROW) part[t].push_back((uint32_t)i);
        });
        for (auto& v : part) long_rows[k].insert(long_rows[k].end(), v.begin(), v.end());
        r->n_long[k] = (uint32_t)long_rows[k].size();
        if (r->n_long[k]) {
            ZG_HIP(ctx, r->long_rows[k].alloc(long_rows[k].size() * 4));
            ZG_HIP(ctx, hipMemcpyAsync(r->long_rows[k].p, long_rows[k].data(), long_rows[k].size() * 4, hipMemcpyHostToDevice, st));
        }
    }
    ZG_HIP(ctx, hipStreamSynchronize(st));
    ZG_HIP(ctx, hipGetLastError());
    *out = r.release();
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
static int params_load_impl(zg_ctx* ctx, const uint8_t* body, size_t len, bool checked, int rank, int world, zg_params** out,
                            size_t* consumed, ParsedParams* vk_out);
// za proving key -> (R1CS, Parameters) in one call
extern "C" int zg_pk_load(zg_ctx* ctx, const uint8_t* pk, size_t len, int checked, int rank, int world, zg_r1cs** r1cs_out,
                          zg_params** params_out) try {
    if (!ctx || !pk || !r1cs_out || !params_out) return ZG_E_INVALID_ARG;
    *r1cs_out = nullptr;
    *params_out = nullptr;
    const double t_walk = now_ms();
    PkWalk walk;
    ZG_TRY(pk_walk(ctx, pk, len, &walk));
    const size_t off = walk.params_offset;
    const double walk_ms = now_ms() - t_walk;
    // the key itself tells how many public inputs (|ic| - 1) and aux variables (|l|) the circuit has
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    if (world < 1 || rank < 0 || rank >= world) return ZG_E_INVALID_ARG;
    // The constraint sections become CSR on host threads WHILE the queries are uploaded and checked on the device: the two
    // are 70 ms and 60-100 ms of a 2^22 key's read and need nothing of each other beyond |ic| and |l|, which the body's length
    // prefixes give.  (The helper reports into a context of its own: fail() writes the caller's error text.)
    ParsedParams pp;
    zg_params* p = nullptr;
    PkR1cs pr;
    uint32_t peek_ic = 0, peek_l = 0;
    const bool peeked = off <= len && peek_bellman_counts(pk + off, len - off, &peek_ic, &peek_l) && peek_ic > 0;
    Ctx side_ctx;
    double parse_ms = 0;
    std::future<int> parsing;
    if (peeked)
        parsing = std::async(std::launch::async, [&]() -> int {
            const double t = now_ms();
            int rc = ZG_E_HIP;
            try {
                rc = parse_pk_r1cs(&side_ctx, pk, len, walk, peek_ic - 1, peek_l, &pr);
            } catch (...) {
                side_ctx.last_error = "out of host memory";
                rc = ZG_E_IO;
            }
            parse_ms = now_ms() - t;
            return rc;
        });
    int rc_params = params_load_impl(ctx, pk + off, len - off, checked != 0, rank, world, &p, nullptr, &pp);
    std::unique_ptr<zg_params> pguard(p);
    int rc_parse = parsing.valid() ? parsing.get() : ZG_OK;
    ZG_TRY(rc_params);   // (the key's own errors first, as when the two ran one after the other)
    if (pp.ic.empty()) return fail(ctx, ZG_E_MALFORMED_VERIFYING_KEY, "pk: empty ic");
    if (peeked && (pp.ic.size() != peek_ic || p->q[1].total != peek_l)) return fail(ctx, ZG_E_BAD_FORMAT, "pk: inconsistent length prefixes");
    if (peeked && rc_parse != ZG_OK) {
        ctx->last_error = side_ctx.last_error;
        return rc_parse;
    }
    if (!peeked) {
        const double t = now_ms();
        ZG_TRY(parse_pk_r1cs(ctx, pk, len, walk, (uint32_t)pp.ic.size() - 1, (uint32_t)p->q[1].total, &pr));
        parse_ms = now_ms() - t;
    }
    double t0 = now_ms();
    ctx->load_timings[2] = parse_ms;   // (beside [1] when the two overlapped)
    ctx->load_timings[0] += walk_ms;
    const uint64_t* rp[3] = {pr.row_ptr[0].data(), pr.row_ptr[1].data(), pr.row_ptr[2].data()};
    const uint32_t* cl[3] = {pr.col[0].data(), pr.col[1].data(), pr.col[2].data()};
    const uint8_t* cf[3] = {pr.coeff[0].data(), pr.coeff[1].data(), pr.coeff[2].data()};
    zg_r1cs* r = nullptr;
    t0 = now_ms();
    ZG_TRY(zg_r1cs_load(ctx, pr.n_inputs, pr.n_aux, pr.n_constraints, rp, cl, cf, &r));
    ctx->load_timings[3] = now_ms() - t0;
    r->var_of_signal = std::move(pr.var_of_signal);
    pguard.release();
    *r1cs_out = r;
    *params_out = p;
    if (trace_level() != 0)
        fprintf(stderr, "[za_amd trace] pk_load: done +%.3f ms (walk %.3f, header + queries %.3f, constraints %.3f, R1CS %.3f)\n",
                now_ms() - t_walk, walk_ms, ctx->load_timings[0] - walk_ms + ctx->load_timings[1], ctx->load_timings[2], ctx->load_timings[3]);
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_witness_from_binwitness(zg_ctx* ctx, const zg_r1cs* r, const uint8_t* binwitness, size_t len,
                                          uint8_t* witness32_out, size_t cap) try {
    if (!ctx || !r || !binwitness || !witness32_out) return ZG_E_INVALID_ARG;
    if (r->var_of_signal.empty()) return fail(ctx, ZG_E_INVALID_ARG, "this R1CS was not loaded from a za proving key");
    const size_t nv = (size_t)r->n_inputs + r->n_aux;
    if (cap < nv * 32) return ZG_E_BUFFER_TOO_SMALL;
    return witness_from_binwitness(ctx, r->var_of_signal, (uint32_t)nv, binwitness, len, witness32_out);
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" void zg_r1cs_free(zg_r1cs* r) { delete r; }
extern "C" int zg_r1cs_info(const zg_r1cs* r, uint64_t info[7]) try {
    if (!r || !info) return ZG_E_INVALID_ARG;
    info[0] = r->n_inputs; info[1] = r->n_aux; info[2] = r->n_constraints; info[3] = r->log_m;
    info[4] = r->a_aux_total; info[5] = r->b_in_total; info[6] = r->b_aux_total;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

// (begin: buffers and the cleared verdict words; range: the class bytes of [lo, hi); end: the verdict's way to the host)
static int witness_classes_begin(zg_ctx* ctx, const zg_r1cs* r) {
    ProveWs& ws = prove_ws(ctx);
    const size_t nv = (size_t)r->n_inputs + r->n_aux;
    ZG_HIP(ctx, ws.wcls.reserve(nv + ((size_t)2 << r->log_m)));   // (behind the witness: the classes of the C rows' values, c_fold_buffers)
    if (!ws.bad_host) {
        // [0, 8) the validation words; [16, 20) the fold flag of the proof in flight and [64, 64 + 8 DIGIT_SLOTS) the digit sums of
        // its C rows (k_fold_decide); host: the two validation words, then the flag of the last guarded proof
        ZG_HIP(ctx, ws.bad.reserve(64 + 8 * DIGIT_SLOTS));
        ZG_HIP(ctx, hipHostMalloc((void**)&ws.bad_host, 16, hipHostMallocDefault));
        ws.bad_host[2] = 0;
        ZG_HIP(ctx, hipEventCreateWithFlags(&ws.bad_ev, hipEventDisableTiming));
    }
    ZG_HIP(ctx, hipMemsetAsync(ws.bad.p, 0xff, 8, ctx->stream));   // "nothing wrong" = all ones in both words
    ZG_HIP(ctx, hipMemsetAsync((uint8_t*)ws.bad.p + 8, 0, 56 + 8 * DIGIT_SLOTS, ctx->stream));
    return ZG_OK;
}
static int witness_classes_range(zg_ctx* ctx, const void* d_w, size_t lo, size_t hi) {
    ProveWs& ws = prove_ws(ctx);
    if (hi <= lo) return ZG_OK;
    ZG_LAUNCH(ctx, k_witness_classes, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, ctx->stream, (const Fr*)d_w, lo, hi - lo,
                       ws.wcls.as<uint8_t>(), ws.bad.as<uint32_t>());
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}
static int witness_classes_end(zg_ctx* ctx, const void* d_w) {
    ProveWs& ws = prove_ws(ctx);
    ZG_HIP(ctx, hipMemcpyAsync(ws.bad_host, ws.bad.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(ctx, hipEventRecord(ws.bad_ev, ctx->stream));
    ws.bad_pending = true;
    ws.wcls_for = d_w;
    return ZG_OK;
}
static int witness_classes(zg_ctx* ctx, const zg_r1cs* r, const void* d_w) {   // queued on ctx->stream
    ProveWs& ws = prove_ws(ctx);
    if (ws.classes_done_for == d_w && d_w) {   // made chunk by chunk behind the hand-over's copies (witness_handover)
        ws.classes_done_for = nullptr;
        return ZG_OK;
    }
    ws.classes_done_for = nullptr;
    ws.gate_for = nullptr;   // (whatever gates an earlier hand-over left do not describe this witness's classes)
    ZG_TRY(witness_classes_begin(ctx, r));
    ZG_TRY(witness_classes_range(ctx, d_w, 0, (size_t)r->n_inputs + r->n_aux));
    return witness_classes_end(ctx, d_w);
}
// The verdict of the last witness_classes pass (waits for its 8-byte read-back only, not for the stream): a witness
// with a value >= r or with variable 0 != ONE is refused, as bellman's typed assignment cannot express it.
static int witness_verdict(zg_ctx* ctx) {
    ProveWs& ws = prove_ws(ctx);
    if (!ws.bad_pending) return ZG_OK;
    ws.bad_pending = false;
    ZG_HIP(ctx, hipEventSynchronize(ws.bad_ev));
    if (ws.bad_host[0] != 0xffffffffu)
        return fail(ctx, ZG_E_BAD_FORMAT, "witness value %u is not a canonical field element (>= r)", ws.bad_host[0]);
    if (ws.bad_host[1] != 0xffffffffu)
        return fail(ctx, ZG_E_BAD_FORMAT, "witness value 0 must be the constant ONE");
    return ZG_OK;
}

// d_abc[k]: m device-form elements (zero padded), rows = n_constraints + n_inputs; with `bitrev` row i is
// stored at index bitrev(i) (what h_poly_dev consumes)
// c_nat / c_cls (both or neither; needs the tiled path: bitrev, log_m >= 8, at least one constraint): see R1csMats
// c_digits (with c_nat): the guarded form -- the folded C rows and their digit sum (R1csMats::c_digits)
static int r1cs_eval_dev(zg_ctx* ctx, const zg_r1cs* r, const Fr* d_w, Fr* d_abc[3], bool bitrev, Fr* c_nat = nullptr,
                         uint8_t* c_cls = nullptr, unsigned long long* c_digits = nullptr) {
    const size_t m = (size_t)1 << r->log_m;
    hipStream_t st = ctx->stream;
    const int rev_bits = bitrev ? (int)r->log_m : 0;
    uint64_t nnz_total = 0;
    for (int k = 0; k < 3; k++) nnz_total += r->col[k].bytes / 4;
    ProfScope ps(ctx, PROF_R1CS_EVAL, 36 * nnz_total + 96 * (r->n_constraints + r->n_inputs));
    if (bitrev && r->log_m >= 8 && r->n_constraints) {
        ProveWs& ws = prove_ws(ctx);
        if (!ws.wcls_fresh || ws.wcls_for != (const void*)d_w) ZG_TRY(witness_classes(ctx, r, d_w));
        ws.wcls_fresh = false;
        R1csMats M;
        for (int k = 0; k < 3; k++) {
            M.row_ptr[k] = r->row_ptr[k].as<uint64_t>();
            M.col[k] = r->col[k].as<uint32_t>();
            M.coeff[k] = r->coeff[k].as<Fr>();
            M.ccls[k] = r->ccls[k].as<uint8_t>();
            M.out[k] = d_abc[k];
        }
        M.c_nat = c_nat;
        M.c_cls = c_cls;
        M.c_digits = c_nat ? c_digits : nullptr;
        ZG_LAUNCH(ctx, k_r1cs_eval_tiled, dim3((unsigned)(m >> 8), 3), dim3(256), 0, st, M, d_w, ws.wcls.as<uint8_t>(),
                           r->n_constraints, (int)r->log_m);
        for (int k = 0; k < 3; k++)
            if (r->n_long[k])
                ZG_LAUNCH(ctx, k_r1cs_eval_long, dim3((r->n_long[k] + 3) / 4), dim3(256), 0, st,
                                   r->long_rows[k].as<uint32_t>(), r->n_long[k], r->row_ptr[k].as<uint64_t>(),
                                   r->col[k].as<uint32_t>(), r->coeff[k].as<Fr>(), r->ccls[k].as<uint8_t>(), d_w,
                                   ws.wcls.as<uint8_t>(), d_abc[k], (int)r->log_m, k == 2 ? c_nat : nullptr,
                                   k == 2 ? c_cls : nullptr, k == 2 ? M.c_digits : nullptr);
        ZG_LAUNCH(ctx, k_input_rows, dim3((r->n_inputs + 255) / 256), dim3(256), 0, st, d_w, r->n_inputs,
                           r->n_constraints, d_abc[0], rev_bits);
        ZG_HIP(ctx, hipGetLastError());
        return ZG_OK;
    }
    if (c_nat) return fail(ctx, ZG_E_INVALID_ARG, "internal: folded C rows need the tiled evaluator");
    for (int k = 0; k < 3; k++) {
        if (bitrev || r->n_constraints == 0)
            ZG_HIP(ctx, hipMemsetAsync(d_abc[k], 0, m * sizeof(Fr), st));   // padding is scattered
        else
            ZG_HIP(ctx, hipMemsetAsync(d_abc[k] + r->n_constraints, 0, (m - r->n_constraints) * sizeof(Fr), st));
        if (r->n_constraints)
            ZG_LAUNCH(ctx, k_r1cs_eval, dim3((unsigned)((r->n_constraints + 255) / 256)), dim3(256), 0, st,
                               r->row_ptr[k].as<uint64_t>(), r->col[k].as<uint32_t>(), r->coeff[k].as<Fr>(), d_w,
                               r->n_constraints, d_abc[k], rev_bits);
    }
    ZG_LAUNCH(ctx, k_input_rows, dim3((r->n_inputs + 255) / 256), dim3(256), 0, st, d_w, r->n_inputs,
                       r->n_constraints, d_abc[0], rev_bits);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

extern "C" int zg_r1cs_check(zg_ctx* ctx, const zg_r1cs* r, const uint8_t* witness32, uint64_t* first_bad) try {
    if (!ctx || !r || !first_bad) return ZG_E_INVALID_ARG;
    if (!witness32) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "witness is NULL");
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    *first_bad = UINT64_MAX;
    if (r->n_constraints == 0) return ZG_OK;
    const size_t m = (size_t)1 << r->log_m, nv = (size_t)r->n_inputs + r->n_aux;
    DevBuf dw, d[3], flag;
    ZG_HIP(ctx, dw.alloc(nv * 32));
    ZG_HIP(ctx, flag.alloc(8));
    ZG_HIP(ctx, hipMemcpyAsync(dw.p, witness32, nv * 32, hipMemcpyHostToDevice, ctx->stream));
    ZG_HIP(ctx, hipMemsetAsync(flag.p, 0xff, 8, ctx->stream));
    Fr* abc[3];
    for (int k = 0; k < 3; k++) {
        ZG_HIP(ctx, d[k].alloc(m * 32));
        abc[k] = d[k].as<Fr>();
    }
    ZG_TRY(r1cs_eval_dev(ctx, r, dw.as<Fr>(), abc, /*bitrev=*/false));
    ZG_LAUNCH(ctx, k_r1cs_check, dim3((unsigned)((r->n_constraints + 255) / 256)), dim3(256), 0, ctx->stream, abc[0],
                       abc[1], abc[2], r->n_constraints, flag.as<unsigned long long>());
    ZG_HIP(ctx, hipGetLastError());
    unsigned long long h = 0;
    ZG_HIP(ctx, hipMemcpyAsync(&h, flag.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *first_bad = (uint64_t)h;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

extern "C" int zg_r1cs_eval(zg_ctx* ctx, const zg_r1cs* r, const uint8_t* witness32, uint8_t* a_out, uint8_t* b_out,
                            uint8_t* c_out) try {
    if (!ctx || !r || !a_out || !b_out || !c_out) return ZG_E_INVALID_ARG;
    if (!witness32) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "witness is NULL");
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    const size_t m = (size_t)1 << r->log_m, nv = (size_t)r->n_inputs + r->n_aux, rows = r->n_constraints + r->n_inputs;
    DevBuf dw, d[3];
    ZG_HIP(ctx, dw.alloc(nv * 32));
    ZG_HIP(ctx, hipMemcpyAsync(dw.p, witness32, nv * 32, hipMemcpyHostToDevice, ctx->stream));
    Fr* abc[3];
    for (int k = 0; k < 3; k++) {
        ZG_HIP(ctx, d[k].alloc(m * 32));
        abc[k] = d[k].as<Fr>();
    }
    ZG_TRY(r1cs_eval_dev(ctx, r, dw.as<Fr>(), abc, /*bitrev=*/false));
    for (int k = 0; k < 3; k++) ZG_TRY(convert_mont_dev(ctx, d[k].p, rows, 1, 1));  // device form -> ABI form
    uint8_t* outs[3] = {a_out, b_out, c_out};
    for (int k = 0; k < 3; k++)
        ZG_HIP(ctx, hipMemcpyAsync(outs[k], d[k].p, rows * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

// ------------------------------------------------------------------------------------------------ params
static void shard_range(uint64_t total, int rank, int world, uint64_t* lo, uint64_t* hi) {
    uint64_t chunk = (total + world - 1) / world;
    uint64_t l = std::min<uint64_t>(total, (uint64_t)rank * chunk);
    *lo = l;
    *hi = std::min<uint64_t>(total, l + chunk);
}

static int params_upload(zg_ctx* ctx, zg_params* p, int which, const void* host, uint64_t total, size_t elem) {
    auto& q = p->q[which];
    q.total = total;
    shard_range(total, p->rank, p->world, &q.lo, &q.hi);
    ZG_HIP(ctx, q.buf.alloc((q.hi - q.lo) * elem));
    if (q.hi > q.lo) {
        ZG_HIP(ctx, hipMemcpyAsync(q.buf.p, (const uint8_t*)host + q.lo * elem, (q.hi - q.lo) * elem,
                                   hipMemcpyHostToDevice, ctx->stream));
        // resident bases are kept in the device field's M261 form
        ZG_TRY(convert_mont_dev(ctx, q.buf.p, (q.hi - q.lo) * elem / 32, 0, 0));
    }
    return ZG_OK;
}

extern "C" int zg_params_from_mont(zg_ctx* ctx, const uint8_t vk5[448], const uint8_t* h, size_t n_h, const uint8_t* l,
                                   size_t n_l, const uint8_t* a, size_t n_a, const uint8_t* b_g1, size_t n_b_g1,
                                   const uint8_t* b_g2, size_t n_b_g2, int rank, int world, zg_params** out) try {
    if (!ctx || !vk5 || !out || world < 1 || rank < 0 || rank >= world) return ZG_E_INVALID_ARG;
    *out = nullptr;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<zg_params> p(new zg_params());
    p->rank = rank;
    p->world = world;
    memcpy(&p->alpha_g1, vk5, 64);
    memcpy(&p->beta_g1, vk5 + 64, 64);
    memcpy(&p->beta_g2, vk5 + 128, 128);
    memcpy(&p->delta_g1, vk5 + 256, 64);
    memcpy(&p->delta_g2, vk5 + 320, 128);
    ZG_TRY(params_upload(ctx, p.get(), 0, h, n_h, 64));
    ZG_TRY(params_upload(ctx, p.get(), 1, l, n_l, 64));
    ZG_TRY(params_upload(ctx, p.get(), 2, a, n_a, 64));
    ZG_TRY(params_upload(ctx, p.get(), 3, b_g1, n_b_g1, 64));
    ZG_TRY(params_upload(ctx, p.get(), 4, b_g2, n_b_g2, 128));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = p.release();
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

// Parameters::read(reader, checked): the verifying key on the host, the five queries decoded -- and, checked, validated:
// on the curve, G2 points in the order-r subgroup, no identity -- by the device (keyload.hip) straight into the resident
// arrays.  A sharded key validates the WHOLE key on every rank (as every process of the reference reads all of it) but
// keeps, and when unchecked only uploads, its own chunk of each query.
static int params_load_impl(zg_ctx* ctx, const uint8_t* body, size_t len, bool checked, int rank, int world, zg_params** out,
                            size_t* consumed, ParsedParams* vk_out /* nullable: the parsed vk + ic */) {
    ParsedParams pp;
    KeyQuery kq[5];
    const double t_start = now_ms();
    ZG_TRY(parse_bellman_header(ctx, body, len, checked, &pp, kq, consumed));
    ctx->load_timings[0] = now_ms() - t_start;
    ctx->load_timings[1] = ctx->load_timings[2] = ctx->load_timings[3] = 0;
    std::unique_ptr<zg_params> p(new zg_params());
    p->rank = rank;
    p->world = world;
    p->alpha_g1 = pp.alpha_g1; p->beta_g1 = pp.beta_g1; p->delta_g1 = pp.delta_g1;
    p->beta_g2 = pp.beta_g2; p->delta_g2 = pp.delta_g2;
    p->gamma_g2 = pp.gamma_g2;
    p->ic = pp.ic;
    KeyStage stage;
    DevBuf verdict, whole_h;
    ZG_HIP(ctx, verdict.alloc(5 * sizeof(unsigned long long)));
    ZG_HIP(ctx, hipMemsetAsync(verdict.p, 0xff, 5 * sizeof(unsigned long long), ctx->stream));
    // a sharded key takes its slice of the evaluation-form h bases now, while the whole query is at hand: decode all of h
    bool want_whole_h = world > 1 && kq[0].n > 0;
    if (want_whole_h && whole_h.alloc(kq[0].n * sizeof(G1Affine)) != hipSuccess) {
        (void)hipGetLastError();
        want_whole_h = false;
    }
    for (int which = 0; which < 5; which++) {
        auto& q = p->q[which];
        const size_t elem = which == 4 ? 128 : 64;
        q.total = kq[which].n;
        shard_range(q.total, rank, world, &q.lo, &q.hi);
        ZG_HIP(ctx, q.buf.alloc((q.hi - q.lo) * elem));
        unsigned long long* v = verdict.as<unsigned long long>() + which;
        if (which == 0 && want_whole_h) {
            ZG_TRY(key_decode_query_dev(ctx, 0, body + kq[0].off, q.total, checked, 0, q.total, whole_h.p, v, &stage));
            if (q.hi > q.lo)
                ZG_HIP(ctx, hipMemcpyAsync(q.buf.p, whole_h.as<G1Affine>() + q.lo, (q.hi - q.lo) * elem, hipMemcpyDeviceToDevice,
                                           ctx->stream));
        } else {
            ZG_TRY(key_decode_query_dev(ctx, which, body + kq[which].off, q.total, checked, q.lo, q.hi, q.buf.p, v, &stage));
        }
    }
    unsigned long long vh[5];
    ZG_HIP(ctx, hipMemcpyAsync(vh, verdict.p, sizeof(vh), hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ZG_TRY(key_verdict_to_error(ctx, vh));
    ctx->load_timings[1] = now_ms() - t_start - ctx->load_timings[0];
    if (want_whole_h) ZG_TRY(params_h_eval_slice(ctx, p.get(), whole_h.as<G1Affine>(), kq[0].n, nullptr));
    if (vk_out) *vk_out = std::move(pp);
    *out = p.release();
    return ZG_OK;
}

extern "C" int zg_params_load(zg_ctx* ctx, const uint8_t* bellman_params, size_t len, int checked, int rank, int world,
                              zg_params** out, size_t* consumed) try {
    if (!ctx || !bellman_params || !out || world < 1 || rank < 0 || rank >= world) return ZG_E_INVALID_ARG;
    *out = nullptr;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    return params_load_impl(ctx, bellman_params, len, checked != 0, rank, world, out, consumed, nullptr);
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" void zg_params_free(zg_params* p) { delete p; }
extern "C" int zg_params_shard_range(const zg_params* p, int which, uint64_t* lo, uint64_t* hi) try {
    if (!p || which < 0 || which > 4 || !lo || !hi) return ZG_E_INVALID_ARG;
    *lo = p->q[which].lo;
    *hi = p->q[which].hi;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
extern "C" int zg_params_query_total(const zg_params* p, int which, uint64_t* total) try {
    if (!p || which < 0 || which > 4 || !total) return ZG_E_INVALID_ARG;
    *total = p->q[which].total;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
extern "C" int zg_pk_params_offset(const uint8_t* pk, size_t len, size_t* offset) try {
    if (!pk || !offset) return ZG_E_INVALID_ARG;
    return pk_params_offset(pk, len, offset);
} catch (...) {
    return zg::on_exception(nullptr);
}

// ------------------------------------------------------------------------------------------------ prove
static int check_query_sizes(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r) {
    const uint64_t m = (uint64_t)1 << r->log_m;
    if (p->q[0].total < m - 1)
        return fail(ctx, ZG_E_QUERY_SIZE, "h query has %llu points, need %llu", (unsigned long long)p->q[0].total,
                    (unsigned long long)(m - 1));
    if (p->q[1].total < r->n_aux)
        return fail(ctx, ZG_E_QUERY_SIZE, "l query has %llu points, need %u", (unsigned long long)p->q[1].total, r->n_aux);
    if (p->q[2].total != r->n_a)
        return fail(ctx, ZG_E_QUERY_SIZE, "a query has %llu points, densities give %llu",
                    (unsigned long long)p->q[2].total, (unsigned long long)r->n_a);
    if (p->q[3].total != r->n_b || p->q[4].total != r->n_b)
        return fail(ctx, ZG_E_QUERY_SIZE, "b queries have %llu/%llu points, densities give %llu",
                    (unsigned long long)p->q[3].total, (unsigned long long)p->q[4].total, (unsigned long long)r->n_b);
    return ZG_OK;
}

// The two events that gate the multiexps of an overlapped proof (witness ready / H ready) belong to the ctx: they
// are created on the ctx's device and destroyed with it (zg_free).
static int gate_events(zg_ctx* ctx) {
    if (!ctx->gate_w) ZG_HIP(ctx, hipEventCreateWithFlags(&ctx->gate_w, hipEventDisableTiming));
    if (!ctx->gate_h) ZG_HIP(ctx, hipEventCreateWithFlags(&ctx->gate_h, hipEventDisableTiming));
    for (auto& e : ctx->tm_ev)
        if (!e) ZG_HIP(ctx, hipEventCreate(&e));
    for (auto& e : ctx->h_split_ev)
        if (!e) ZG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return ZG_OK;
}

// queue R1CS evaluation + the H pipeline on ctx->stream without waiting for it
// c_nat / c_cls (evaluation form only): the C rows' values leave as multiexp scalars (R1csMats::c_nat) and the pipeline
// transforms a and b only (h_eval_dev with d_c == NULL)
// guard (with c_nat): whether THIS proof runs folded is decided on the device from its rows' digits (k_fold_decide); without
// it the rows are folded whatever they hold (ZG_OPT_C_FOLD = 2, and the builders of the folded query's table)
static int prove_h_enqueue(zg_ctx* ctx, const zg_r1cs* r, const void* d_witness, void* d_h_out, hipEvent_t ev_eval_done = nullptr,
                           bool eval_form = false, int h_split = 0, Fr* c_nat = nullptr, uint8_t* c_cls = nullptr, bool guard = false) {
    const size_t m = (size_t)1 << r->log_m;
    ProveWs& ws = prove_ws(ctx);
    if (!c_nat) guard = false;
    Fr* abc[3];
    for (int k = 0; k < 3; k++) {
        if (k == 2 && c_nat && !guard) {
            abc[k] = nullptr;
            continue;
        }
        ZG_HIP(ctx, ws.abc[k].reserve(m * sizeof(Fr)));
        abc[k] = ws.abc[k].as<Fr>();
    }
    if (c_nat && !eval_form) return fail(ctx, ZG_E_INVALID_ARG, "internal: folded C rows without the evaluation form");
    unsigned long long* d_digits = guard ? (unsigned long long*)((uint8_t*)ws.bad.p + 64) : nullptr;
    uint32_t* d_flag = guard ? (uint32_t*)((uint8_t*)ws.bad.p + 16) : nullptr;
    ZG_TRY(r1cs_eval_dev(ctx, r, (const Fr*)d_witness, abc, /*bitrev=*/r->log_m > 0, c_nat, c_cls, d_digits));
    ctx->fold_guarded = guard;
    if (guard) {
        // (a proof that folds -- nearly all of them -- only reads the counters: a small grid)
        ZG_LAUNCH(ctx, k_fold_decide, dim3((unsigned)std::min<size_t>((m + 255) / 256, 512)), dim3(256), 0, ctx->stream, d_digits,
                  (unsigned long long)m * 2, d_flag, c_nat, c_cls, abc[2], (int)r->log_m);
        ZG_HIP(ctx, hipGetLastError());
        ZG_HIP(ctx, hipMemcpyAsync(&ws.bad_host[2], d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));   // (zg_last_stats)
    }
    if (ev_eval_done) ZG_HIP(ctx, hipEventRecord(ev_eval_done, ctx->stream));
    if (eval_form)
        return h_eval_dev(ctx, abc[0], abc[1], abc[2], r->log_m, (Fr*)d_h_out, h_split, h_split ? ctx->h_split_ev : nullptr, d_flag);
    return h_poly_dev(ctx, abc[0], abc[1], abc[2], r->log_m, (Fr*)d_h_out);
}
extern "C" int zg_prove_h_dev(zg_ctx* ctx, const zg_r1cs* r, const void* d_witness, void* d_h_out) try {
    if (!ctx || !r || !d_h_out) return ZG_E_INVALID_ARG;
    if (!d_witness) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "witness is NULL");
    ctx->fold_armed_key = nullptr;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    double t0 = now_ms();
    ZG_TRY(prove_h_enqueue(ctx, r, d_witness, d_h_out));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->timings[1] = 0;
    ctx->timings[2] = now_ms() - t0;   // R1CS evaluation + H
    return witness_verdict(ctx);
} catch (...) {
    return zg::on_exception(ctx);
}

extern "C" int zg_prove_h_eval_dev(zg_ctx* ctx, const zg_r1cs* r, const void* d_witness, void* d_e_out) try {
    if (!ctx || !r || !d_e_out) return ZG_E_INVALID_ARG;
    if (!d_witness) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "witness is NULL");
    ctx->fold_armed_key = nullptr;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    double t0 = now_ms();
    ZG_TRY(prove_h_enqueue(ctx, r, d_witness, d_e_out, nullptr, /*eval_form=*/true));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->timings[1] = 0;
    ctx->timings[2] = now_ms() - t0;   // R1CS evaluation + H
    return witness_verdict(ctx);
} catch (...) {
    return zg::on_exception(ctx);
}

extern "C" int zg_prove_h_eval_enqueue_dev(zg_ctx* ctx, const zg_r1cs* r, const void* d_witness, void* d_e_out) try {
    if (!ctx || !r || !d_e_out) return ZG_E_INVALID_ARG;
    if (!d_witness) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "witness is NULL");
    ctx->fold_armed_key = nullptr;   // (whatever zg_prove_h_eval_enqueue_fold_dev queued before is superseded)
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    ZG_TRY(gate_events(ctx));
    ZG_TRY(witness_classes(ctx, r, d_witness));   // as zg_prove_h_enqueue_dev: the witness gate sits before the pipeline
    prove_ws(ctx).wcls_fresh = prove_ws(ctx).wcls_armed = true;
    ZG_HIP(ctx, hipEventRecord(ctx->gate_w, ctx->stream));
    ctx->witness_gate_armed = true;
    return prove_h_enqueue(ctx, r, d_witness, d_e_out, nullptr, /*eval_form=*/true);
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_params_h_eval_info(const zg_params* p, uint64_t info[4]) try {
    if (!p || !info) return ZG_E_INVALID_ARG;
    std::lock_guard<std::mutex> lock(p->fbw_mu);
    info[0] = p->h_eval_state == 1 ? 1 : 0;
    info[1] = p->h_eval_log_m;
    info[2] = p->h_eval_lo;
    info[3] = p->h_eval_hi;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

extern "C" int zg_prove_partials_dev(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const void* d_witness,
                                     const void* d_h_shard, uint8_t partials[ZG_PARTIALS_BYTES]) try {
    return zg_prove_partials_ex(ctx, p, r, d_witness, d_h_shard, 0x1f, partials);
} catch (...) {
    return zg::on_exception(ctx);
}

extern "C" int zg_prove_h_enqueue_dev(zg_ctx* ctx, const zg_r1cs* r, const void* d_witness, void* d_h_out) try {
    if (!ctx || !r || !d_h_out) return ZG_E_INVALID_ARG;
    if (!d_witness) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "witness is NULL");
    ctx->fold_armed_key = nullptr;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    // "witness ready" is recorded BEFORE the H pipeline is queued: a following zg_prove_partials_ex with the witness
    // mask gates its multiexps on this event instead of on everything queued so far, so rank 0's l / a / b jobs run
    // beside its NTT passes (as they do inside zg_prove_dev) rather than after them.
    ZG_TRY(gate_events(ctx));
    ZG_TRY(witness_classes(ctx, r, d_witness));   // before the gate: the witness multiexps read the class bytes too
    prove_ws(ctx).wcls_fresh = prove_ws(ctx).wcls_armed = true;
    ZG_HIP(ctx, hipEventRecord(ctx->gate_w, ctx->stream));
    ctx->witness_gate_armed = true;
    return prove_h_enqueue(ctx, r, d_witness, d_h_out);
} catch (...) {
    return zg::on_exception(ctx);
}

// Fixed-base window tables for this rank's part of each query, built once per key: every proof with the key
// re-uses them.  The digit width is planned from the number of scalars that are neither 0 nor 1 in the witness the
// tables are first built for (a circuit's share of bit signals is structural), and a table is only built where it
// beats the per-window schedule.  Budget: ZG_OPT_TABLES_MAX_MB (default: 60% of the free HBM) in the order b_g2, h, l,
// a, b_g1; ZG_OPT_TABLES = 0 turns the tables off.
struct QueryUse {
    const void* bases;
    const uint32_t* scalars;
    const uint32_t* index;
    uint64_t n;
    const uint8_t* cls = nullptr;   // class bytes addressed like `scalars` (MsmDesc::cls)
};
// The h query in evaluation form (heval.hip), built on first use: unsharded keys only (the group transform needs the
// whole query); ZG_OPT_H_EVAL = 0 keeps bellman's coefficient form (seven transforms per proof).
static bool h_eval_allowed(const Ctx* ctx) { return ctx->opt_h_eval; }   // ZG_OPT_H_EVAL, read per key
// A sharded key's slice: positions [rank * ceil(m / world), ...) of the m-point evaluation-form array.  The group
// transform needs the whole query, so this runs where it is at hand: zg_params_load (every rank parses the whole key),
// zg_params_shard (cut from a full resident key; d_h_eval_full = its transformed array if it has one).
static void c_fold_shard_bases(Ctx* ctx, const zg_params* p, const G1Affine* d_h_full, uint64_t n_h, uint32_t log_m);
int zg::params_h_eval_slice(Ctx* ctx, const zg_params* p, const G1Affine* d_h_full, uint64_t n_h, const G1Affine* d_h_eval_full) {
    std::lock_guard<std::mutex> lock(p->fbw_mu);
    p->h_eval.release();
    p->h_eval_state = -1;
    uint32_t log_m = 0;
    while (((uint64_t)1 << log_m) < n_h + 1) log_m++;
    if (!h_eval_allowed(ctx) || log_m == 0 || log_m >= 28 || ((uint64_t)1 << log_m) != n_h + 1) return ZG_OK;   // bellman: |h| = m - 1
    const uint64_t m = (uint64_t)1 << log_m, chunk = (m + p->world - 1) / p->world;
    const uint64_t lo = std::min<uint64_t>(m, (uint64_t)p->rank * chunk), hi = std::min<uint64_t>(m, lo + chunk);
    DevBuf full;
    if (!d_h_eval_full) {
        if (full.alloc(m * sizeof(G1Affine)) != hipSuccess) {
            (void)hipGetLastError();
            return ZG_OK;
        }
        if (h_eval_bases_build(ctx, d_h_full, n_h, log_m, full.as<G1Affine>()) != ZG_OK) {   // a fallback, not an error
            (void)hipGetLastError();
            ctx->last_error.clear();
            return ZG_OK;
        }
        d_h_eval_full = full.as<G1Affine>();
    }
    if (p->h_eval.alloc(std::max<uint64_t>(hi - lo, 1) * sizeof(G1Affine)) != hipSuccess) {
        (void)hipGetLastError();
        return ZG_OK;
    }
    if ((hi > lo && hipMemcpyAsync(p->h_eval.p, d_h_eval_full + lo, (hi - lo) * sizeof(G1Affine), hipMemcpyDeviceToDevice,
                                   ctx->stream) != hipSuccess) ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        p->h_eval.release();
        return ZG_OK;
    }
    p->h_eval_log_m = log_m;
    p->h_eval_lo = lo;
    p->h_eval_hi = hi;
    p->h_eval_state = 1;
    // rank 0 -- the rank that runs the transforms -- also takes the bases of the folded C rows (ensure_c_fold), while the
    // whole h query is at hand
    if (p->world > 1 && p->rank == 0) c_fold_shard_bases(ctx, p, d_h_full, n_h, log_m);
    return ZG_OK;
}
// States of zg_params::h_eval_state: 0 not tried, 2 being built by some context, 1 built -- and from then on IMMUTABLE for
// the key's lifetime (readers may keep the pointer after dropping the lock), -1 no evaluation form for this key (switched
// off, the query has another shape, no room, the build failed): sticky, so a proof never pays a second attempt -- only
// zg_params_prepare (retry) tries again.  The multi-second build runs OUTSIDE fbw_mu: contexts sharing the key go on in
// the coefficient form meanwhile.  Any failure on the way is a fallback to bellman's seven transforms, never an error.
static bool h_eval_ready(const zg_params* p, uint32_t log_m) {
    std::lock_guard<std::mutex> lock(p->fbw_mu);
    return p->h_eval_state == 1 && p->h_eval_log_m == log_m;
}
static int ensure_h_eval(Ctx* ctx, const zg_params* p, uint32_t log_m, bool retry = false) {
    if (p->world != 1) return ZG_OK;   // a sharded key has its slice from the start, or never (params_h_eval_slice)
    const uint64_t m = (uint64_t)1 << log_m;
    {
        std::lock_guard<std::mutex> lock(p->fbw_mu);
        if (p->h_eval_state == 1 || p->h_eval_state == 2) return ZG_OK;
        if (p->h_eval_state == -1 && !retry) return ZG_OK;
        if (ctx->one_shot) return ZG_OK;   // (the state stays as it is: another context may still build the bases)
        if (!h_eval_allowed(ctx) || log_m == 0 || p->q[0].total < m - 1) {
            p->h_eval_state = -1;
            return ZG_OK;
        }
        p->h_eval_state = 2;
    }
    DevBuf built;
    int rc = ZG_OK;
    try {   // (whatever ends the build -- an exception included -- must not leave the state at "being built" for ever)
        if (built.alloc(m * sizeof(G1Affine)) != hipSuccess) rc = ZG_E_HIP;   // no room: the coefficient form is used
        if (rc == ZG_OK) rc = h_eval_bases_build(ctx, p->q[0].buf.as<G1Affine>(), p->q[0].total, log_m, built.as<G1Affine>());
    } catch (...) {
        rc = ZG_E_HIP;
    }
    std::lock_guard<std::mutex> lock(p->fbw_mu);
    if (rc != ZG_OK) {
        (void)hipGetLastError();
        ctx->last_error.clear();
        p->h_eval_state = -1;
        return ZG_OK;
    }
    std::swap(p->h_eval.p, built.p);
    std::swap(p->h_eval.bytes, built.bytes);
    p->h_eval_log_m = log_m;
    p->h_eval_lo = 0;
    p->h_eval_hi = m;
    p->h_eval_state = 1;
    return ZG_OK;
}

// lc_form: use[1] is the l query with the C rows' bases behind it (ensure_c_fold) and its table is zg_params::fbw_lc
static int ensure_tables(Ctx* ctx, const zg_params* p, const QueryUse use[5], unsigned mask, bool h_eval_form = false,
                         bool lc_form = false) {
    std::lock_guard<std::mutex> lock(p->fbw_mu);
    static const int order[5] = {4, 0, 1, 2, 3};
    // The witness queries that get a WINDOW table get it with one common plan (c, nwin) where they can, decided when the
    // first of them is built: queries with the same plan share one digit sort per proof (msm.hip MsmShare).  The plan is
    // made for the largest of them -- a smaller query then runs with a digit a bit wider than its own optimum.
    uint64_t w_real[5] = {0, 0, 0, 0, 0};
    bool w_counted[5] = {false, false, false, false, false};
    for (int oi = 0; oi < 5; oi++) {
        const int k = order[oi];
        auto& t = k == 0 && h_eval_form ? p->fbw_he : k == 1 && lc_form ? p->fbw_lc : p->fbw[k];
        const uint64_t n = use[k].n;
        if (!(mask & (1u << k)) || n == 0) continue;
        if (k != 0 && !use[k].scalars) continue;   // zg_params_prepare without a witness: only the h table
        if ((t.state == 1 && t.n == n) || t.state == -1) continue;
        if (ctx->one_shot) continue;   // no tables for a key that is used once (their state stays "not tried")
        // While another context builds the evaluation form of this key's h query, proofs go on in the coefficient form --
        // WITHOUT its window table: those gigabytes would never be released once zg_prove* has moved to the evaluation
        // form, and they would shrink the budget of the tables that stay (the state remains "not tried")
        if (k == 0 && !h_eval_form && p->h_eval_state == 2) continue;
        t.buf.release();
        t.state = -1;
        if (!(ctx->opt_tables & (1u << k))) continue;   // ZG_OPT_TABLES
        if (n < 64) continue;
        uint64_t n_real = n;
        if (k != 0) {   // h is dense by construction
            uint64_t zeros = 0, ones = 0;
            ZG_TRY(msm_classify_counts(ctx, use[k].scalars, use[k].index, n, &zeros, &ones));
            n_real = n - zeros - ones;
            w_real[k] = n_real;
            w_counted[k] = true;
        }
        size_t free_b = 0, total_b = 0;
        ZG_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
        // keep room for what the proofs themselves allocate later: the digit-sort buffers of every job (12 bytes per
        // (point, digit) pair, up to 16 digits) and the evaluation / H vectors -- at 2^26 with a dense witness that is
        // ~70 GB, and tables sized from "60 % of what is free now" alone starved them (hipMalloc failed inside the
        // first proof)
        size_t reserve = 0;
        for (int q = 0; q < 5; q++) reserve += (size_t)use[q].n * 16 * 12;
        reserve += (size_t)use[0].n * 32 * 8;
        const size_t avail = free_b > reserve ? free_b - reserve : 0;
        size_t budget = (size_t)((double)avail * 0.6);
        // (round 6: letting b_g1's table -- the last one built, and the one that lets its multiexp ride the G2 job's sort --
        // take 95 % of what is left instead of 60 % made it fit at 2^26 (34 GB) and the first proof ran out of memory in its
        // staged sort buffers: the 60 % stays)
        if (ctx->opt_tables_max_mb) budget = std::min<size_t>(budget, (size_t)ctx->opt_tables_max_mb * 1000000u);
        const size_t elem = k == 4 ? sizeof(G2Affine) : sizeof(G1Affine);
        // latency bound (msm.hip small-job mode: always a window table): a dense query up to 2^16 points; a witness query up
        // to ~2^19 positions IF few of its scalars are "real" (a bit-heavy circuit: shard of a 2^22 proof, 3 % of 2^19) --
        // with 183 k field elements (the b queries of the 2^20 Poseidon chain) the mode's short items and wide digits cost
        // throughput: G2 accumulation 1.8 -> 2.6 ms of kernel time, the proof 8.3 -> 8.8 ms
        const bool small = n <= msm_small_n(ctx) || (k != 0 && n <= msm_small_witness_n(ctx) && n_real <= msm_small_n(ctx));
        t.small = small;
        // The b_g1 query takes the layout of the b_g2 query's table (built just before it: same scalars, same positions) where
        // it can -- its multiexp then runs over the G2 job's sorted entries (MsmDesc::follow_g2).  A G1 table of the same shape
        // always fits where the G2 one did.
        const zg_params::Fbw* twin = k == 3 && ctx->opt_share_sort && (!small || ctx->opt_share_sort >= 2) && p->fbw[4].state == 1 &&
                                                 p->fbw[4].n == n && p->fbw[4].small == small
                                         ? &p->fbw[4] : nullptr;
        // (the folded l query from 60 %: rows that evaluate to small integers are "anything else" by class, and cheap)
        if (!small && k != 0 && (n - n_real) * 10 >= n * (k == 1 && lc_form ? 6 : 9) && n >= 1024) {
            // a mostly-0/1 assignment: subset sums over groups of four positions halve the additions of its 1s
            // groups of eight positions (255 sums: 2 KB per position) where that fits, else of four (240 B per position);
            // (ZG_OPT_SUBSET_W = 4 | 8)
            int qw = ctx->opt_subset_w == 4 ? 4 : 8;
            if (twin && twin->kind == 1) qw = twin->qw;
            size_t entries = msm_quad_entries(n, qw);
            if (qw == 8 && (entries * elem > budget || entries >= (1ull << 31))) {
                qw = 4;
                entries = msm_quad_entries(n, qw);
            }
            // ... and behind the sums a window table over the positions that hold field elements (3 % of a SHA-like circuit,
            // fixed by the circuit): their digits then share ONE bucket set like a window-table job's -- 16 digits of 16
            // bits instead of 22 per-window digits of 12 -- gathered from a table 3 % the size of a full one (the full table was
            // measured slower: gathers over an 8 GB footprint).  ZG_OPT_COMPACT_TABLE = 0 leaves it out.
            int cw_c = 0, cw_nwin = 0;
            const bool cw_on = ctx->opt_compact_table;
            if (cw_on && n_real >= 1024 && budget > entries * elem &&
                !msm_fbw_plan(n_real, n_real, budget - entries * elem, elem, &cw_c, &cw_nwin))
                cw_c = cw_nwin = 0;
            if (!cw_on || n_real < 1024) cw_c = cw_nwin = 0;
            if (twin && twin->kind == 1 && twin->qw == qw && twin->cw_n == n_real && (twin->cw_nwin == 0 || cw_on)) {
                cw_c = twin->cw_c;
                cw_nwin = twin->cw_nwin;
            }
            const size_t cw_entries = (size_t)cw_nwin * n_real;
            if (entries + cw_entries >= (1ull << 31)) cw_c = cw_nwin = 0;
            const size_t all_entries = entries + (size_t)cw_nwin * n_real;
            if (entries * elem > budget || entries >= (1ull << 31) || t.buf.alloc(all_entries * elem) != hipSuccess) {
                (void)hipGetLastError();
                continue;
            }
            if (k == 4)
                ZG_TRY(msm_quad_build_g2(ctx, (const G2Affine*)use[k].bases, n, t.buf.as<G2Affine>(), qw));
            else
                ZG_TRY(msm_quad_build_g1(ctx, (const G1Affine*)use[k].bases, n, t.buf.as<G1Affine>(), qw));
            t.qw = qw;
            t.cw_c = t.cw_nwin = 0;
            t.cw_n = t.cw_off = 0;
            if (cw_nwin && t.rank.alloc(n * 4) == hipSuccess) {
                // row 0 = the compacted bases themselves, written where the table's first row belongs
                uint64_t got = 0;
                int rc;
                if (k == 4) {
                    G2Affine* rows = t.buf.as<G2Affine>() + entries;
                    rc = msm_compact_reals_g2(ctx, (const G2Affine*)use[k].bases, use[k].scalars, use[k].index, n,
                                              t.rank.as<uint32_t>(), rows, &got);
                    if (rc == ZG_OK && got == n_real) rc = msm_fbw_build_g2(ctx, rows, n_real, n_real, cw_c, cw_nwin, rows);
                } else {
                    G1Affine* rows = t.buf.as<G1Affine>() + entries;
                    rc = msm_compact_reals_g1(ctx, (const G1Affine*)use[k].bases, use[k].scalars, use[k].index, n,
                                              t.rank.as<uint32_t>(), rows, &got);
                    if (rc == ZG_OK && got == n_real) rc = msm_fbw_build_g1(ctx, rows, n_real, n_real, cw_c, cw_nwin, rows);
                }
                ZG_TRY(rc);
                if (got == n_real) {
                    t.cw_c = cw_c;
                    t.cw_nwin = cw_nwin;
                    t.cw_n = (uint32_t)n_real;
                    t.cw_off = (uint32_t)entries;
                }
            } else {
                (void)hipGetLastError();
            }
            ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            t.kind = 1;
            t.n = n;
            t.state = 1;
            continue;
        }
        // few real scalars leave few digits to save, and gathering them from an nwin-times larger table costs more
        // than that (measured at 2^22: b_g2 accumulation 1.5 -> 2.1 ms with 3% real scalars)
        if (!small && (n_real < 64 || n_real * 2 < n)) continue;
        int c = 0, nwin = 0;
        // (a latency-bound witness query is planned from the field elements its witness holds -- a circuit's share of bit
        // signals is structural -- not from its positions: 8 K field elements among 2^18 positions planned as 2^18 took 20-bit
        // digits, 2^19 buckets and ten reduction levels where 16 bits and eight levels do; the job's buffers are still sized
        // for any witness, and one with more field elements only runs with a narrower digit than its optimum)
        if (!msm_fbw_plan(small ? std::max<uint64_t>(n_real, 1024) : n_real, n, budget, elem, &c, &nwin, small)) continue;
        // (not for queries in small-job mode unless mode 2 asks for their shared sort: their digits are planned for the depth
        // of each query's own chain, and the largest query's plan for all of them cost a 2^18 proof 0.35 ms)
        if (k != 0 && !(k == 1 && lc_form) && ctx->opt_share_sort && (!small || ctx->opt_share_sort >= 2)) {
            // the common plan of the witness queries' window tables: made with the first one built from the largest query
            // that will take this path (same mode: small or not), then reused while it fits this query's budget
            if (p->w_plan_c == 0) {
                uint64_t big_n = n, big_real = n_real;
                for (int q = 1; q <= 4; q++) {
                    if (q == k || !(mask & (1u << q)) || use[q].n == 0 || !use[q].scalars) continue;
                    uint64_t qr = w_real[q];
                    if (!w_counted[q]) {
                        uint64_t z = 0, o = 0;
                        ZG_TRY(msm_classify_counts(ctx, use[q].scalars, use[q].index, use[q].n, &z, &o));
                        qr = use[q].n - z - o;
                        w_real[q] = qr;
                        w_counted[q] = true;
                    }
                    const bool q_small = use[q].n <= msm_small_n(ctx) || (use[q].n <= msm_small_witness_n(ctx) && qr <= msm_small_n(ctx));
                    const bool q_window = q_small || !((use[q].n - qr) * 10 >= use[q].n * 9 && use[q].n >= 1024 || qr < 64 || qr * 2 < use[q].n);
                    if (q_small != small || !q_window) continue;
                    big_n = std::max<uint64_t>(big_n, use[q].n);
                    big_real = std::max<uint64_t>(big_real, qr);
                }
                int cc = 0, cn = 0;
                if (msm_fbw_plan(std::max<uint64_t>(big_real, small ? 1024 : 0), big_n, budget, sizeof(G2Affine), &cc, &cn, small, /*always=*/true)) {
                    p->w_plan_c = cc;
                    p->w_plan_nwin = cn;
                    p->w_plan_small = small;
                }
            }
            if (p->w_plan_c && p->w_plan_small == small && (double)n * p->w_plan_nwin * (double)elem <= (double)budget &&
                (double)n * p->w_plan_nwin < 2.0e9) {
                c = p->w_plan_c;
                nwin = p->w_plan_nwin;
            }
        }
        if (twin && twin->kind == 0) {
            c = twin->c;
            nwin = twin->nwin;
        }
        if (t.buf.alloc((size_t)nwin * n * elem) != hipSuccess) {
            (void)hipGetLastError();
            continue;   // no room: the plain per-window schedule is used
        }
        if (k == 4)
            ZG_TRY(msm_fbw_build_g2(ctx, (const G2Affine*)use[k].bases, n, n, c, nwin, t.buf.as<G2Affine>()));
        else
            ZG_TRY(msm_fbw_build_g1(ctx, (const G1Affine*)use[k].bases, n, n, c, nwin, t.buf.as<G1Affine>()));
        ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        t.c = c;
        t.nwin = nwin;
        t.kind = 0;
        t.n = n;
        t.state = 1;
    }
    return ZG_OK;
}

// the five multiexps' operands for this rank's shard (n = 0 where the mask leaves a query out)
static int query_uses(Ctx* ctx, const zg_params* p, const zg_r1cs* r, const void* d_witness, const void* d_h_shard,
                      unsigned mask, QueryUse use[5], const uint8_t* wcls = nullptr) {
    const uint64_t m = (uint64_t)1 << r->log_m;
    const uint32_t* w = (const uint32_t*)d_witness;
    {   // h: first m-1 points of the query; this shard holds [lo, hi)
        const auto& q = p->q[0];
        uint64_t hi = std::min<uint64_t>(q.hi, m - 1);
        uint64_t n = hi > q.lo && (mask & 1u) ? hi - q.lo : 0;
        use[0] = QueryUse{q.buf.p, (const uint32_t*)d_h_shard, nullptr, n};
    }
    {   // l: aux assignment, full density
        const auto& q = p->q[1];
        uint64_t hi = std::min<uint64_t>(q.hi, r->n_aux);
        uint64_t n = hi > q.lo && (mask & 2u) ? hi - q.lo : 0;
        use[1] = QueryUse{q.buf.p, w ? w + 8 * ((size_t)r->n_inputs + q.lo) : nullptr, nullptr, n,
                          wcls ? wcls + (size_t)r->n_inputs + q.lo : nullptr};
    }
    {   // a: inputs (full density) then aux by a_aux_density
        const auto& q = p->q[2];
        use[2] = QueryUse{q.buf.p, w, r->a_index.as<uint32_t>() + q.lo, (mask & 4u) ? q.hi - q.lo : 0, wcls};
    }
    {
        const auto& q = p->q[3];
        use[3] = QueryUse{q.buf.p, w, r->b_index.as<uint32_t>() + q.lo, (mask & 8u) ? q.hi - q.lo : 0, wcls};
    }
    {
        const auto& q = p->q[4];
        use[4] = QueryUse{q.buf.p, w, r->b_index.as<uint32_t>() + q.lo, (mask & 16u) ? q.hi - q.lo : 0, wcls};
    }
    (void)ctx;
    return ZG_OK;
}

// ---- "C folded": four transforms per proof instead of six (ZG_OPT_C_FOLD; ntt.hip h_eval_dev, heval.hip c_fold)
// The C term of the quotient enters the proof as  sum_j c_j C_j  over the values c_j of the R1CS rows' C sides -- and C's two
// transforms go.  l and the h term are only ever ADDED in a proof (bellman: g_c = ... + h + l), so the m bases C_j are kept
// behind the l query as ONE query of n_l + 2 m points: no sixth multiexp, no sixth chain of launches.  Its scalars are the
// aux part of the witness followed by the c_j, its class bytes likewise (both contiguous: c_fold_buffers); the second m
// points are the doubled bases, for the rows that evaluate to 2 (R1csMats::c_nat).  Worth it where
// the c_j are bits -- the subset-sum table makes their multiexp one addition per eight rows -- which is decided once per
// key, from the witness its tables are first built for (as the tables' own plans are).  zg_prove* with an unsharded key, and
// the phase API through zg_prove_h_eval_enqueue_fold_dev (below: the rank that runs a sharded proof's transforms); the
// plain phase entries keep six transforms (their h scalars and the l partial are documented quantities).
// mode 1's sizes: from ZG_OPT_C_FOLD_MIN_LOG_M (below 2^20 the folded l job's longer chain costs more than two transforms of
// a short pipeline save) to 2^26 (measured: config-4 shape 163.3 -> 154.0 ms; 2^27, the largest legal domain, is not)
static constexpr int C_FOLD_MAX_LOG_M = 26;
static bool c_fold_size_ok(const Ctx* ctx, uint32_t log_m) {
    return ctx->opt_c_fold >= 2 || ((int)log_m >= ctx->opt_c_fold_min_log_m && (int)log_m <= C_FOLD_MAX_LOG_M);
}
static uint64_t c_fold_n_l(const zg_params* p, const zg_r1cs* r) {
    const auto& q = p->q[1];
    const uint64_t hi = std::min<uint64_t>(q.hi, r->n_aux);
    return hi > q.lo ? hi - q.lo : 0;
}
static bool c_fold_ready(const zg_params* p, const zg_r1cs* r) {
    std::lock_guard<std::mutex> lock(p->fbw_mu);
    return p->lc_state == 1 && p->lc_log_m == r->log_m && p->lc_n_l == c_fold_n_l(p, r) && p->h_eval_state == 1 &&
           p->h_eval_log_m == r->log_m;
}
// The folded query's scalars and class bytes for this witness: [aux values | c rows] and [aux classes | c classes].
// In the context's own witness buffer (zg_prove: it is allocated with room for the rows) nothing is copied; a caller's
// device witness has its aux part copied once (134 MB at 2^22: 0.06 ms on the context's stream).
struct CFoldBufs {
    const uint32_t* scalars = nullptr;   // n_l + 2 m values
    const uint8_t* cls = nullptr;
    Fr* c_nat = nullptr;
    uint8_t* c_cls = nullptr;
};
// (rank 0's shard of a sharded key: its range of the aux values and of their class bytes is copied in front of the rows' --
// the classes behind it in the witness's array belong to the other aux values; call it AFTER witness_classes)
static int c_fold_buffers(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const void* d_witness, uint64_t n_l, CFoldBufs* out) {
    ProveWs& ws = prove_ws(ctx);
    const size_t m = (size_t)1 << r->log_m, nv = (size_t)r->n_inputs + r->n_aux;
    if (p->world != 1) {
        const size_t lo = (size_t)p->q[1].lo;
        if (lo + n_l > r->n_aux) return fail(ctx, ZG_E_INVALID_ARG, "internal: l shard beyond the aux values");
        ZG_HIP(ctx, ws.w_lc.reserve((n_l + 2 * m) * 32));
        ZG_HIP(ctx, ws.cls_lc.reserve(n_l + 2 * m));
        if (n_l) {
            ZG_HIP(ctx, hipMemcpyAsync(ws.w_lc.p, (const Fr*)d_witness + r->n_inputs + lo, n_l * 32, hipMemcpyDeviceToDevice, ctx->stream));
            ZG_HIP(ctx, hipMemcpyAsync(ws.cls_lc.p, ws.wcls.as<uint8_t>() + r->n_inputs + lo, n_l, hipMemcpyDeviceToDevice, ctx->stream));
        }
        out->scalars = ws.w_lc.as<uint32_t>();
        out->c_nat = ws.w_lc.as<Fr>() + n_l;
        out->cls = ws.cls_lc.as<uint8_t>();
        out->c_cls = ws.cls_lc.as<uint8_t>() + n_l;
        return ZG_OK;
    }
    if (n_l != r->n_aux) return fail(ctx, ZG_E_INVALID_ARG, "internal: folded C rows need the whole l query");
    ZG_HIP(ctx, ws.wcls.reserve(nv + 2 * m));
    Fr* base;
    if (d_witness == ws.w.p && ws.w.bytes >= (nv + 2 * m) * 32) {
        base = ws.w.as<Fr>() + r->n_inputs;
    } else {
        ZG_HIP(ctx, ws.w_lc.reserve((n_l + 2 * m) * 32));
        base = ws.w_lc.as<Fr>();
        if (n_l)
            ZG_HIP(ctx, hipMemcpyAsync(base, (const Fr*)d_witness + r->n_inputs, n_l * 32, hipMemcpyDeviceToDevice, ctx->stream));
    }
    out->scalars = (const uint32_t*)base;
    out->c_nat = base + n_l;
    out->cls = ws.wcls.as<uint8_t>() + r->n_inputs;
    out->c_cls = ws.wcls.as<uint8_t>() + nv;
    return ZG_OK;
}
// States of zg_params::lc_state as h_eval_state (0 not tried, 2 being built, 1 built and immutable, -1 not for this key).
// Blocking (a group transform of the h query and a table: ~2 s at 2^22), once per key, outside fbw_mu.
static int ensure_c_fold(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const void* d_witness, bool retry = false) {
    if (p->world != 1 || !ctx->opt_c_fold || !d_witness) return ZG_OK;
    const uint32_t log_m = r->log_m;
    const uint64_t m = (uint64_t)1 << log_m, n_l = c_fold_n_l(p, r);
    {
        std::lock_guard<std::mutex> lock(p->fbw_mu);
        if (p->lc_state == 1 || p->lc_state == 2) return ZG_OK;
        if (p->lc_state == -1 && !retry) return ZG_OK;
        if (ctx->one_shot) return ZG_OK;
        if (p->h_eval_state == 0 || p->h_eval_state == 2) return ZG_OK;   // not yet known: ask again with the next proof
        const bool can = p->h_eval_state == 1 && p->h_eval_log_m == log_m && log_m >= 8 && r->n_constraints != 0 &&
                         n_l == r->n_aux && n_l + 2 * m < ((uint64_t)1 << 31) && p->q[0].total >= m - 1;
        if (!can || !c_fold_size_ok(ctx, log_m)) {
            if (!can) p->lc_state = -1;   // (a size limit of mode 1 is the context's, not the key's: the state stays)
            return ZG_OK;
        }
        p->lc_state = 2;
    }
    DevBuf built;
    bool ok = false;
    int rc = ZG_OK;
    try {
        do {
            // the C rows for THIS witness: is their multiexp a bit job?
            CFoldBufs cb;
            if ((rc = c_fold_buffers(ctx, p, r, d_witness, n_l, &cb)) != ZG_OK) break;
            ProveWs& ws = prove_ws(ctx);
            Fr* abc[3] = {nullptr, nullptr, nullptr};
            for (int k = 0; k < 2 && rc == ZG_OK; k++) {
                if (ws.abc[k].reserve(m * sizeof(Fr)) != hipSuccess) rc = ZG_E_HIP;
                abc[k] = ws.abc[k].as<Fr>();
            }
            if (rc != ZG_OK) break;
            if ((rc = r1cs_eval_dev(ctx, r, (const Fr*)d_witness, abc, /*bitrev=*/true, cb.c_nat, cb.c_cls)) != ZG_OK) break;
            uint64_t zeros = 0, ones = 0;   // (for the trace only)
            if (trace_level() != 0 && (rc = msm_classify_counts(ctx, cb.scalars, nullptr, n_l + 2 * m, &zeros, &ones)) != ZG_OK) break;
            // What the rows cost as a multiexp: their non-zero 16-bit digits (an addition each; rows equal to 2 are ones on
            // the doubled bases and not counted).  Two transforms of m elements are worth ~3.4 additions per row (1.0 ms
            // against 71 ps per addition at 2^22, and the rows' entries are also sorted); folded up to 2 digits per row, the
            // neighbourhood that was measured -- the SHA-256 Merkle circuit has 0.18, the config-4 recipe 0.27 (AND
            // rows: bits; XOR rows as (2a) b = a + b - c: 0 or 2; adder rows: a word), the config-4 recipe at 90 % bits 1.8 (2^22:
            // 11.66 -> 10.88 ms folded), a circuit of field elements (Poseidon) 16 (2^20: 7.5 -> 8.4 ms folded).
            DevBuf cnt;
            unsigned long long nz = 0;
            if (cnt.alloc(8) != hipSuccess || hipMemsetAsync(cnt.p, 0, 8, ctx->stream) != hipSuccess) break;
            ZG_LAUNCH(ctx, k_count_digits16, dim3((unsigned)std::min<uint64_t>((m + 255) / 256, 4096)), dim3(256), 0, ctx->stream, cb.c_nat,
                      (size_t)m, (unsigned long long*)cnt.p);
            if (hipMemcpyAsync(&nz, cnt.p, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                break;
            if (trace_level() != 0)
                fprintf(stderr, "[za_amd trace] c_fold: %llu zeros, %llu ones among %llu aux values + 2 x %llu rows; the rows hold %llu digits\n",
                        (unsigned long long)zeros, (unsigned long long)ones, (unsigned long long)n_l, (unsigned long long)m, nz);
            if (ctx->opt_c_fold == 1 && nz > m * 2) break;   // field elements: six transforms stay
            if (built.alloc((n_l + 2 * m) * sizeof(G1Affine)) != hipSuccess) break;
            if (n_l && hipMemcpyAsync(built.p, p->q[1].buf.p, n_l * sizeof(G1Affine), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                break;
            if ((rc = h_eval_bases_build(ctx, p->q[0].buf.as<G1Affine>(), p->q[0].total, log_m, built.as<G1Affine>() + n_l, /*c_fold=*/true)) != ZG_OK)
                break;
            {   // the folded query's table is built with the other queries' (ensure_tables, in their order of priority)
                std::lock_guard<std::mutex> lock(p->fbw_mu);
                p->fbw_lc.buf.release();
                p->fbw_lc.rank.release();
                p->fbw_lc.state = 0;
            }
            ok = true;
        } while (false);
    } catch (...) {
        rc = ZG_E_HIP;
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (trace_level() != 0) fprintf(stderr, "[za_amd trace] c_fold: rc %d ok %d (%s)\n", rc, (int)ok, ctx->last_error.c_str());
    std::lock_guard<std::mutex> lock(p->fbw_mu);
    if (rc != ZG_OK || !ok) {   // a fallback to six transforms, never an error
        (void)hipGetLastError();
        ctx->last_error.clear();
        p->fbw_lc.buf.release();
        p->fbw_lc.rank.release();
        p->fbw_lc.state = 0;
        p->lc_state = -1;
        return ZG_OK;
    }
    std::swap(p->lc.p, built.p);
    std::swap(p->lc.bytes, built.bytes);
    p->lc_log_m = log_m;
    p->lc_n_l = n_l;
    p->lc_state = 1;
    return ZG_OK;
}

// ---- the folded form in the phase API: the rank that runs the transforms of a SHARDED proof (rank 0, za_amd/dist.py) drops
// C's two as well.  Nothing changes for the other ranks and nothing more crosses the links: the scalars rank 0 scatters
// are a b / Z instead of (a b - c) / Z, and rank 0's l partial carries the C term -- the whole of it: a bit multiexp over
// the m rows' values against the 2 m bases behind ITS shard of the l query, which costs rank 0 a third of what the two
// transforms did (its transforms are the serial part of a sharded proof: 3.7 of ~6.3 ms at 2^22 and N = 8).
// The bases are made when the key is loaded or cut (params_h_eval_slice: the group transform needs the whole h query,
// fbw_mu held); any failure leaves the key as it was (six transforms).
static void c_fold_shard_bases(Ctx* ctx, const zg_params* p, const G1Affine* d_h_full, uint64_t n_h, uint32_t log_m) {
    const uint64_t m = (uint64_t)1 << log_m;
    const auto& q = p->q[1];
    const uint64_t n_l = q.hi > q.lo ? q.hi - q.lo : 0;
    p->lc_state = -1;
    if (!ctx->opt_c_fold || ctx->one_shot || log_m < 8 || !c_fold_size_ok(ctx, log_m) || q.lo != 0 || n_l + 2 * m >= ((uint64_t)1 << 31) || n_h < m - 1)
        return;
    DevBuf built;
    bool ok = false;
    try {
        ok = built.alloc((n_l + 2 * m) * sizeof(G1Affine)) == hipSuccess &&
             (n_l == 0 || hipMemcpyAsync(built.p, q.buf.p, n_l * sizeof(G1Affine), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess) &&
             h_eval_bases_build(ctx, d_h_full, n_h, log_m, built.as<G1Affine>() + n_l, /*c_fold=*/true) == ZG_OK &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
    } catch (...) {
        ok = false;
    }
    if (!ok) {
        (void)hipGetLastError();
        ctx->last_error.clear();
        return;
    }
    std::swap(p->lc.p, built.p);
    std::swap(p->lc.bytes, built.bytes);
    p->lc_log_m = log_m;
    p->lc_n_l = n_l;
    p->lc_rule = 0;
    p->lc_state = 1;
}
// Whether the proofs of a key whose bases were made without a witness (c_fold_shard_bases) use them: ensure_c_fold's rule, on
// the first witness this key sees (blocking, once).  `cb`: this witness's buffers; the rows are evaluated into them.
static int c_fold_rule(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const void* d_witness, const CFoldBufs& cb, bool* fold) {
    {
        std::lock_guard<std::mutex> lock(p->fbw_mu);
        if (p->lc_rule != 0 || ctx->opt_c_fold >= 2) {
            *fold = ctx->opt_c_fold >= 2 || p->lc_rule > 0;
            return ZG_OK;
        }
    }
    const size_t m = (size_t)1 << r->log_m;
    ProveWs& ws = prove_ws(ctx);
    Fr* abc[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < 2; k++) {
        ZG_HIP(ctx, ws.abc[k].reserve(m * sizeof(Fr)));
        abc[k] = ws.abc[k].as<Fr>();
    }
    ZG_TRY(r1cs_eval_dev(ctx, r, (const Fr*)d_witness, abc, /*bitrev=*/true, cb.c_nat, cb.c_cls));
    DevBuf cnt;
    unsigned long long nz = 0;
    ZG_HIP(ctx, cnt.alloc(8));
    ZG_HIP(ctx, hipMemsetAsync(cnt.p, 0, 8, ctx->stream));
    ZG_LAUNCH(ctx, k_count_digits16, dim3((unsigned)std::min<uint64_t>((m + 255) / 256, 4096)), dim3(256), 0, ctx->stream, cb.c_nat, m,
              (unsigned long long*)cnt.p);
    ZG_HIP(ctx, hipMemcpyAsync(&nz, cnt.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lock(p->fbw_mu);
    if (p->lc_rule == 0) p->lc_rule = nz <= (unsigned long long)m * 2 ? 1 : -1;
    *fold = p->lc_rule > 0;
    if (p->lc_rule < 0) {   // six transforms for this key: the folded rows' bases (64 (n_l + 2 m) bytes) and their table go
        p->lc.release();
        p->fbw_lc.buf.release();
        p->fbw_lc.rank.release();
        p->fbw_lc.state = 0;
        p->lc_state = -1;
    }
    return ZG_OK;
}
// zg_prove_h_eval_enqueue_dev for a caller that names the key: where the key has the bases of the folded C rows (an
// unsharded key: made here on first use; rank 0's shard of a sharded one: made when it was loaded) the pipeline runs four
// transforms, d_e_out receives a b / Z, and the NEXT zg_prove_partials_ex on this context with this key and the l query in
// its mask adds the rows' multiexp to the l partial.  Any other key: exactly zg_prove_h_eval_enqueue_dev.
extern "C" int zg_prove_h_eval_enqueue_fold_dev(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const void* d_witness, void* d_e_out,
                                                int* folded) try {
    if (!ctx || !p || !r || !d_e_out) return ZG_E_INVALID_ARG;
    if (!d_witness) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "witness is NULL");
    if (folded) *folded = 0;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    ZG_TRY(gate_events(ctx));
    ctx->fold_armed_key = nullptr;
    if (p->world == 1) {   // (an unsharded key makes its transformed bases on first use, here as in zg_prove)
        ZG_TRY(ensure_h_eval(ctx, p, r->log_m));
        if (h_eval_ready(p, r->log_m)) ZG_TRY(ensure_c_fold(ctx, p, r, d_witness));
    }
    bool fold = ctx->opt_c_fold && (p->world == 1 || p->rank == 0) && c_fold_ready(p, r) && c_fold_size_ok(ctx, r->log_m);
    ZG_TRY(witness_classes(ctx, r, d_witness));   // as zg_prove_h_enqueue_dev: the witness gate sits before the pipeline
    CFoldBufs cfb;
    if (fold) {
        ZG_TRY(c_fold_buffers(ctx, p, r, d_witness, c_fold_n_l(p, r), &cfb));
        if (p->world != 1) ZG_TRY(c_fold_rule(ctx, p, r, d_witness, cfb, &fold));
    }
    prove_ws(ctx).wcls_fresh = prove_ws(ctx).wcls_armed = true;
    ZG_HIP(ctx, hipEventRecord(ctx->gate_w, ctx->stream));
    ctx->witness_gate_armed = true;
    ZG_TRY(prove_h_enqueue(ctx, r, d_witness, d_e_out, ctx->tm_ev[1], /*eval_form=*/true, 0, fold ? cfb.c_nat : nullptr,
                           fold ? cfb.c_cls : nullptr, /*guard=*/ctx->opt_c_fold == 1));
    if (fold) {
        ctx->fold_armed_key = p;
        ctx->fold_armed_r1cs = r;
        ctx->fold_armed_witness = d_witness;
        ctx->fold_scalars = cfb.scalars;
        ctx->fold_cls = cfb.cls;
        if (folded) *folded = 1;
    }
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

extern "C" int zg_params_table_info(const zg_params* p, int which, int* kind, uint64_t* bytes) try {
    if (!p || which < 0 || which > 5 || !kind || !bytes) return ZG_E_INVALID_ARG;
    std::lock_guard<std::mutex> lock(p->fbw_mu);
    const auto& t = which == 5 ? p->fbw_lc : which == 0 && p->fbw_he.state == 1 ? p->fbw_he : p->fbw[which];
    *kind = t.state == 1 ? (t.kind == 1 ? 2 : 1) : 0;
    *bytes = t.state == 1 ? t.buf.bytes : 0;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

extern "C" int zg_params_prepare(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const void* d_witness) try {
    if (!ctx || !p || !r) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    ZG_TRY(check_query_sizes(ctx, p, r));
    QueryUse use[5];
    ZG_TRY(query_uses(ctx, p, r, d_witness, nullptr, 0x1f, use));
    // zg_prove* takes H in evaluation form when the key allows it: then the table that matters is the one over the
    // transformed bases (the coefficient-form table is still built on demand by the phase API)
    ZG_TRY(ensure_h_eval(ctx, p, r->log_m, /*retry=*/true));
    const bool eval_form = h_eval_ready(p, r->log_m);
    if (eval_form) use[0] = QueryUse{p->h_eval.p, nullptr, nullptr, p->h_eval_hi - p->h_eval_lo};   // (a sharded key: its slice)
    // (first: a key whose C rows are folded into the l query needs no table for the plain l query, but one for the folded
    // query -- over this witness's aux values and the values its C rows evaluate to)
    if (eval_form && d_witness) ZG_TRY(ensure_c_fold(ctx, p, r, d_witness, /*retry=*/true));
    bool folded = eval_form && d_witness && ctx->opt_c_fold && (p->world == 1 || p->rank == 0) && c_fold_ready(p, r) &&
                  c_fold_size_ok(ctx, r->log_m);
    CFoldBufs cb;
    if (folded && p->world != 1) {   // rank 0's shard: its class bytes are copied (after they are made), its rule may not be decided yet
        ZG_TRY(witness_classes(ctx, r, d_witness));
        ZG_TRY(c_fold_buffers(ctx, p, r, d_witness, c_fold_n_l(p, r), &cb));
        ZG_TRY(c_fold_rule(ctx, p, r, d_witness, cb, &folded));
    } else if (folded) {
        ZG_TRY(c_fold_buffers(ctx, p, r, d_witness, c_fold_n_l(p, r), &cb));
    }
    if (folded) {
        ProveWs& ws = prove_ws(ctx);
        Fr* abc[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < 2; k++) {
            ZG_HIP(ctx, ws.abc[k].reserve(((size_t)1 << r->log_m) * sizeof(Fr)));
            abc[k] = ws.abc[k].as<Fr>();
        }
        ZG_TRY(r1cs_eval_dev(ctx, r, (const Fr*)d_witness, abc, /*bitrev=*/true, cb.c_nat, cb.c_cls));
        use[1] = QueryUse{p->lc.p, cb.scalars, nullptr, p->lc_n_l + ((uint64_t)2 << r->log_m), cb.cls};
    }
    return ensure_tables(ctx, p, use, 0x1fu, eval_form, folded);
} catch (...) {
    return zg::on_exception(ctx);
}

// h_eval_form: d_h_shard holds h_eval_dev's m scalars and the h multiexp runs over the key's h_eval bases
// h_split > 0: the scalars arrive in that many ranges (ctx->h_split_ev, see h_eval_dev) and the h job runs in as many parts
// c_fold: the l query is the key's folded one (zg_params::lc) over c_fold->scalars, gated on c_fold_ev (the R1CS evaluation
// that writes the C rows' values); the "l" partial then holds l + the C term and the "h" partial the a b / Z term
static int prove_partials_impl(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const void* d_witness, const void* d_h_shard,
                               unsigned mask, uint8_t partials[ZG_PARTIALS_BYTES], bool h_eval_form, int h_split = 0,
                               const CFoldBufs* c_fold = nullptr, hipEvent_t c_fold_ev = nullptr) {
    if (!ctx || !p || !r || !partials) return ZG_E_INVALID_ARG;
    if (!d_witness) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "witness is NULL");
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    ZG_TRY(check_query_sizes(ctx, p, r));
    double t0 = now_ms();
    G1XYZZ g1out[4];
    G2XYZZ b2_acc;
    QueryUse use[5];
    // class bytes of the witness: from this proof's zg_prove_dev / zg_prove_h_enqueue_dev, else computed here (the
    // multiexps of a rank that does not evaluate the R1CS) -- then the jobs wait for everything queued so far
    ProveWs& ws = prove_ws(ctx);
    bool own_classes = false;
    if (mask & 0x1eu) {
        if (!(ws.wcls_armed && ws.wcls_for == d_witness)) {
            ZG_TRY(witness_classes(ctx, r, d_witness));
            own_classes = true;
        }
    }
    ws.wcls_armed = false;
    ZG_TRY(query_uses(ctx, p, r, d_witness, d_h_shard, mask, use, (mask & 0x1eu) ? ws.wcls.as<uint8_t>() : nullptr));
    if (h_eval_form && (mask & 1u)) {
        if (!h_eval_ready(p, r->log_m))
            return fail(ctx, ZG_E_INVALID_ARG, "this key has no evaluation-form h bases for a 2^%u domain", r->log_m);
        use[0] = QueryUse{p->h_eval.p, (const uint32_t*)d_h_shard, nullptr, p->h_eval_hi - p->h_eval_lo};
    }
    if (use[0].n && !d_h_shard) return fail(ctx, ZG_E_INVALID_ARG, "d_h_shard is NULL");
    if (c_fold) use[1] = QueryUse{p->lc.p, c_fold->scalars, nullptr, p->lc_n_l + ((uint64_t)2 << r->log_m), c_fold->cls};
    ZG_TRY(ensure_tables(ctx, p, use, mask, h_eval_form, c_fold != nullptr));
    MsmDesc d1[4], d2;
    for (int k = 0; k < 5; k++) {
        MsmDesc d{use[k].bases, use[k].scalars, use[k].index, (size_t)use[k].n, /*dense=*/k == 0};
        d.cls = k == 0 ? nullptr : use[k].cls;
        if (k != 0) {
            const auto& tk = k == 1 && c_fold ? p->fbw_lc : p->fbw[k];
            if (tk.state == 1 && tk.n == use[k].n) d.small_hint = tk.small ? 1 : 0;
        }
        if (k == 1 && c_fold) d.gate = c_fold_ev;
        if (k == 0 && h_split > 0) {
            d.parts = h_split;
            d.part_gate = ctx->h_split_ev;
            d.part_gate_only = true;
        }
        const auto& t = k == 0 && h_eval_form ? p->fbw_he : k == 1 && c_fold ? p->fbw_lc : p->fbw[k];
        if (use[k].n && t.state == 1 && t.n == use[k].n && t.kind == 1) {
            d.quad_table = t.buf.p;
            d.quad_w = t.qw;
            if (t.cw_nwin) {
                d.cw_rank = t.rank.as<uint32_t>();
                d.cw_off = t.cw_off;
                d.cw_n = t.cw_n;
                d.cw_c = t.cw_c;
                d.cw_nwin = t.cw_nwin;
            }
        } else if (use[k].n && t.state == 1 && t.n == use[k].n) {
            d.fbw_table = t.buf.p;
            d.fbw_c = t.c;
            d.fbw_nwin = t.nwin;
            d.fbw_stride = (size_t)use[k].n;
        }
        if (k < 4) d1[k] = d; else d2 = d;
    }
    // The l multiexp of a witness that was handed over in chunks (witness_handover): part k over the aux values of chunk k,
    // gated on that chunk's event alone.  Not for the folded l query (its job waits for the evaluation anyway) nor a
    // latency-bound one; a query in a shared sort is left whole by msm_batch5.
    size_t l_bounds[ProveWs::GATES + 1];
    {
        ProveWs& ws = prove_ws(ctx);
        const size_t n_l = (size_t)use[1].n, var0 = (size_t)r->n_inputs + (size_t)p->q[1].lo;   // variable of l position 0
        if (ws.gate_n > 1 && ws.gate_for == d_witness && d_witness && !c_fold && n_l >= ((size_t)1 << 19) && d1[1].small_hint != 1) {
            for (int g = 0; g <= ws.gate_n; g++) {
                const size_t v = ws.gate_var[g];
                l_bounds[g] = g == ws.gate_n ? n_l : std::min(n_l, v > var0 ? (v - var0) & ~(size_t)63 : 0);
            }
            d1[1].parts = ws.gate_n;
            d1[1].part_gate = ws.gate_ev;
            d1[1].part_bounds = l_bounds;
            d1[1].part_gate_only = true;
        }
    }
    // (b_g1 may only ride b_g2's sorted entries when both jobs cover the SAME positions: equal lengths, the same start of the
    // shard, the same index list and scalars -- true of every key cut by shard_range from equal totals, asserted here)
    if (ctx->opt_share_sort && use[3].n && use[3].n == use[4].n && p->q[3].lo == p->q[4].lo && use[3].index == use[4].index &&
        use[3].scalars == use[4].scalars) {
        const auto &t1 = p->fbw[3], &t2 = p->fbw[4];
        const bool tab1 = t1.state == 1 && t1.n == use[3].n, tab2 = t2.state == 1 && t2.n == use[4].n;
        // (mode 1: not for jobs in small-job mode -- their chains of short launches are bound by latency, not by the work of a
        // sort, and b_g1 behind b_g2's sort measured 0.25 ms slower at 2^18; mode 2 follows at every size)
        const bool latency_bound = tab1 && t1.small && ctx->opt_share_sort < 2;
        d1[3].follow_g2 = latency_bound ? false : (!tab1 && !tab2) ||
                          (tab1 && tab2 && t1.kind == t2.kind && t1.small == t2.small &&
                           (t1.kind == 0 ? t1.c == t2.c && t1.nwin == t2.nwin
                                         : t1.qw == t2.qw && t1.cw_c == t2.cw_c && t1.cw_nwin == t2.cw_nwin && t1.cw_n == t2.cw_n &&
                                               t1.cw_off == t2.cw_off));
    }
    // One digit sort for the witness queries whose window tables have the key's common plan (ensure_tables): they multiply
    // the same scalars, so their (position, digit) pairs are sorted once, over the variables this shard's queries span, and
    // each query reads the sorted entries through its position map (msm.hip MsmShare / ShQuery).
    MsmShare share;
    int w_plan_c = 0, w_plan_nwin = 0;
    bool w_plan_small = false;
    {   // (the plan is written by ensure_tables under the key's lock: another context may be building tables right now)
        std::lock_guard<std::mutex> lock(p->fbw_mu);
        w_plan_c = p->w_plan_c;
        w_plan_nwin = p->w_plan_nwin;
        w_plan_small = p->w_plan_small;
    }
    if (ctx->opt_share_sort && w_plan_c && w_plan_nwin <= 32) {
        const uint64_t nvars = (uint64_t)r->n_inputs + r->n_aux;
        uint64_t pos0 = nvars, pos1 = 0;
        unsigned members = 0;
        int n_members = 0;
        for (int k = 1; k <= 4; k++) {
            const auto& t = p->fbw[k];
            if (k == 1 && c_fold) continue;   // (the folded l query spans the C rows too: a sort of its own)
            if (!use[k].n || !(mask & (1u << k)) || t.state != 1 || t.kind != 0 || t.n != use[k].n || t.c != w_plan_c ||
                t.nwin != w_plan_nwin || t.small != w_plan_small)
                continue;
            // (a query that reads the entries through a rank map pays a gather per entry and walks the entries of variables
            // outside its density: measured, a sort of its own is cheaper -- mode 2 shares those too, for the tests)
            if (ctx->opt_share_sort < 2 && k != 1 && (k == 2 ? r->a_rank.p : r->b_rank.p)) continue;
            const auto& q = p->q[k];
            uint64_t v0, v1;   // the variables the shard's positions [q.lo, q.lo + n) map to
            if (k == 1) {
                v0 = (uint64_t)r->n_inputs + q.lo;
                v1 = v0 + use[k].n;
            } else {
                const std::vector<uint32_t>& ix = k == 2 ? r->a_index_host : r->b_index_host;
                if (q.lo + use[k].n > ix.size()) continue;
                v0 = ix[q.lo];
                v1 = (uint64_t)ix[q.lo + use[k].n - 1] + 1;
            }
            pos0 = std::min(pos0, v0);
            pos1 = std::max(pos1, v1);
            members |= 1u << k;
            n_members++;
        }
        if (n_members >= 2 && pos1 > pos0 && pos1 - pos0 <= ((uint64_t)1 << 26)) {
            share.mask = members;
            share.scalars = (const uint32_t*)d_witness + 8 * pos0;
            share.cls = ws.wcls.as<uint8_t>() + pos0;
            share.n = (size_t)(pos1 - pos0);
            share.c = w_plan_c;
            share.nwin = w_plan_nwin;
            share.small = w_plan_small ? 1 : 0;
            for (int k = 1; k <= 4; k++) {
                if (!(members & (1u << k))) continue;
                const auto& q = p->q[k];
                if (k == 1) {
                    share.q[k].rank = nullptr;
                    share.q[k].sub = (uint32_t)((uint64_t)r->n_inputs + q.lo - pos0);
                } else {
                    const DevBuf& rk = k == 2 ? r->a_rank : r->b_rank;
                    share.q[k].rank = rk.p ? rk.as<uint32_t>() + pos0 : nullptr;
                    share.q[k].sub = rk.p ? (uint32_t)q.lo : (uint32_t)(q.lo - pos0);   // (full density: variable = position in the query)
                }
            }
        }
    }
    hipEvent_t ev_wit = ctx->ev_witness;
    if (!ev_wit && ctx->witness_gate_armed && !(mask & 1u)) ev_wit = ctx->gate_w;   // see zg_prove_h_enqueue_dev
    if (own_classes) ev_wit = nullptr;   // the class kernel was queued just now: gate on "everything so far"
    ctx->witness_gate_armed = false;
    if (trace_level() != 0)
        fprintf(stderr, "[za_amd trace] partials: jobs described         +%.3f ms (since partials entry)\n", now_ms() - t0);
    ctx->last_share_mask = share.mask;
    ctx->last_l_parts = d1[1].part_gate && !(share.mask & 2u) ? d1[1].parts : 1;
    ctx->last_follow_g2 = d1[3].follow_g2 && !(share.mask >> 3 & 3u);
    // (a guarded proof: what k_fold_decide decided for it -- its 4-byte copy was queued in front of the l job's gate, and the
    // multiexps below have been waited for by the time this is read: set again at the end)
    ctx->last_c_fold = c_fold != nullptr;
    const bool fold_guarded = c_fold != nullptr && ctx->fold_guarded;
    ZG_TRY(msm_batch5(ctx, d1, d2, g1out, &b2_acc, ev_wit, ctx->ev_h, share.mask ? &share : nullptr));
    if (fold_guarded) ctx->last_c_fold = prove_ws(ctx).bad_host[2] != 0u;
    const G1XYZZ &h_acc = g1out[0], &l_acc = g1out[1], &a_acc = g1out[2], &b1_acc = g1out[3];
    jac_bytes(partials, h_acc);
    jac_bytes(partials + 96, l_acc);
    jac_bytes(partials + 192, a_acc);
    jac_bytes(partials + 288, b1_acc);
    jac_bytes(partials + 384, b2_acc);
    ctx->timings[3] = now_ms() - t0;
    return witness_verdict(ctx);
}
extern "C" int zg_prove_partials_ex(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const void* d_witness,
                                    const void* d_h_shard, unsigned mask, uint8_t partials[ZG_PARTIALS_BYTES]) try {
    // (a pipeline queued by zg_prove_h_eval_enqueue_fold_dev left out C's transforms: the l query of that key carries its rows)
    if (ctx && (mask & 2u) && ctx->fold_armed_key && ctx->fold_armed_key == (const void*)p) {
        // the armed scalars and class bytes were derived from the witness and circuit of the enqueue call: another witness
        // here would have its classes rewritten underneath them and a silently wrong l partial
        if (ctx->fold_armed_r1cs != (const void*)r || ctx->fold_armed_witness != d_witness) {
            ctx->fold_armed_key = nullptr;
            return fail(ctx, ZG_E_INVALID_ARG, "zg_prove_partials_ex: the pipeline queued by zg_prove_h_eval_enqueue_fold_dev left C's "
                                               "rows to the l partial of THAT circuit and witness; this call names another");
        }
        CFoldBufs cfb;
        cfb.scalars = ctx->fold_scalars;
        cfb.cls = ctx->fold_cls;
        ctx->fold_armed_key = nullptr;
        return prove_partials_impl(ctx, p, r, d_witness, d_h_shard, mask & 0x1fu, partials, (mask & ZG_MASK_H_EVAL) != 0, 0, &cfb, ctx->tm_ev[1]);
    }
    return prove_partials_impl(ctx, p, r, d_witness, d_h_shard, mask & 0x1fu, partials, (mask & ZG_MASK_H_EVAL) != 0);
} catch (...) {
    return zg::on_exception(ctx);
}

static void scalar_words(uint32_t out[8], const uint8_t le[32]) { memcpy(out, le, 32); }


// k0*P0 + k1*P1 + k2*P2 with one shared doubling chain (Straus): the 7 non-empty subset sums are tabulated,
// then 254 doublings + at most 254 additions instead of three separate double-and-add ladders.
static G1XYZZ straus3(const G1XYZZ pts[3], const uint32_t* const k[3]) {
    G1XYZZ tab[8];
    tab[0] = G1XYZZ::inf();
    for (int m = 1; m < 8; m++) {
        const int low = m & -m, idx = low == 1 ? 0 : low == 2 ? 1 : 2;
        tab[m] = tab[m ^ low].add(pts[idx]);
    }
    G1XYZZ acc = G1XYZZ::inf();
    for (int i = 255; i >= 0; i--) {
        acc = acc.dbl();
        const int m = (int)((k[0][i >> 5] >> (i & 31)) & 1) | (int)(((k[1][i >> 5] >> (i & 31)) & 1) << 1) |
                      (int)(((k[2][i >> 5] >> (i & 31)) & 1) << 2);
        if (m) acc = acc.add(tab[m]);
    }
    return acc;
}

// The part of the combination that needs the witness sums only (everything but "+ sum h"): A and B in affine form and
// C without its h term.  zg_prove_dev runs it while the h multiexp is still on the GPU (Ctx::witness_hook).
struct CombineEarly {
    bool done = false;
    G1Affine A;
    G2Affine B;
    G1XYZZ c_part;
};
static bool combine_scalars(const uint8_t r_le[32], const uint8_t s_le[32], uint32_t r[8], uint32_t s[8], uint32_t rs[8]) {
    scalar_words(r, r_le);
    scalar_words(s, s_le);
    if (!scalar_is_canonical(r) || !scalar_is_canonical(s)) return false;
    Fr fr_r, fr_s;
    memcpy(&fr_r, r, 32);
    memcpy(&fr_s, s, 32);
    Fr prod = (fr_r.to_mont() * fr_s.to_mont()).from_mont();
    memcpy(rs, &prod, 32);
    return true;
}
// What the combination needs of r, s and the key alone: r delta_1, s delta_2 (the one G2 multiplication) and
// s alpha + r beta_1 + rs delta_1.  zg_prove_dev computes it on a helper thread while the GPU works.
struct CombinePre {
    bool ok = false;
    uint32_t r[8], s[8];
    G1XYZZ r_delta1, c0;
    G2XYZZ s_delta2;
};
static void combine_pre(const zg_params* p, const uint8_t r_le[32], const uint8_t s_le[32], CombinePre* out) {
    uint32_t rs[8];
    if (p->delta_g1.is_inf() || p->delta_g2.is_inf() || !combine_scalars(r_le, s_le, out->r, out->s, rs)) return;
    const G1XYZZ delta1 = G1XYZZ::from_affine(p->delta_g1);
    out->r_delta1 = delta1.mul(out->r);
    out->s_delta2 = G2XYZZ::from_affine(p->delta_g2).mul(out->s);
    const G1XYZZ pts[3] = {G1XYZZ::from_affine(p->alpha_g1), G1XYZZ::from_affine(p->beta_g1), delta1};
    const uint32_t* const ks[3] = {out->s, out->r, rs};
    out->c0 = straus3(pts, ks);
    out->ok = true;
}
// ... and what is left once the witness sums are there: two additions for A and B, s * sum a + r * sum b_g1 for C
static void combine_witness_part_pre(const zg_params* p, const CombinePre& pre, const G1XYZZ& l_acc, const G1XYZZ& a_ans,
                                     const G1XYZZ& b1_ans, const G2XYZZ& b2_ans, CombineEarly* out) {
    out->A = pre.r_delta1.add_mixed(p->alpha_g1).add(a_ans).to_affine();
    out->B = pre.s_delta2.add_mixed(p->beta_g2).add(b2_ans).to_affine();
    static const uint32_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const G1XYZZ pts[3] = {a_ans, b1_ans, G1XYZZ::inf()};
    const uint32_t* const ks[3] = {pre.s, pre.r, zero};
    out->c_part = pre.c0.add(straus3(pts, ks)).add(l_acc);
    out->done = true;
}
static void combine_witness_part(const zg_params* p, const uint32_t r[8], const uint32_t s[8], const uint32_t rs[8],
                                 const G1XYZZ& l_acc, const G1XYZZ& a_ans, const G1XYZZ& b1_ans, const G2XYZZ& b2_ans,
                                 CombineEarly* out) {
    const G1XYZZ delta1 = G1XYZZ::from_affine(p->delta_g1), alpha = G1XYZZ::from_affine(p->alpha_g1),
                 beta1 = G1XYZZ::from_affine(p->beta_g1);
    const G2XYZZ delta2 = G2XYZZ::from_affine(p->delta_g2);
    // B = beta2 + s*delta2 + sum b2: the one G2 multiplication (about 3 G1 ones) runs on a helper thread
    std::thread tb([&] { out->B = delta2.mul(s).add_mixed(p->beta_g2).add(b2_ans).to_affine(); });
    // A = alpha + r*delta + sum a
    out->A = delta1.mul(r).add_mixed(p->alpha_g1).add(a_ans).to_affine();
    // C = s*(alpha + sum a) + r*(beta1 + sum b1) + rs*delta + sum l + sum h (the last term is the caller's)
    const G1XYZZ pts[3] = {alpha.add(a_ans), beta1.add(b1_ans), delta1};
    const uint32_t* const ks[3] = {s, r, rs};
    out->c_part = straus3(pts, ks).add(l_acc);
    tb.join();
    out->done = true;
}

static int combine_impl(zg_ctx* ctx, const zg_params* p, const uint8_t* partials_all, int world, const uint8_t r_le[32],
                        const uint8_t s_le[32], uint8_t proof_out[256], const CombineEarly* early) {
    double t0 = now_ms();
    G1XYZZ h_acc = G1XYZZ::inf(), l_acc = G1XYZZ::inf(), a_ans = G1XYZZ::inf(), b1_ans = G1XYZZ::inf();
    G2XYZZ b2_ans = G2XYZZ::inf();
    for (int k = 0; k < world; k++) {
        const uint8_t* q = partials_all + (size_t)k * ZG_PARTIALS_BYTES;
        h_acc = h_acc.add(jac_from_bytes<Fq>(q));
        if (early && early->done) continue;
        l_acc = l_acc.add(jac_from_bytes<Fq>(q + 96));
        a_ans = a_ans.add(jac_from_bytes<Fq>(q + 192));
        b1_ans = b1_ans.add(jac_from_bytes<Fq>(q + 288));
        b2_ans = b2_ans.add(jac_from_bytes<Fq2>(q + 384));
    }
    CombineEarly own;
    if (!(early && early->done)) {
        if (p->delta_g1.is_inf() || p->delta_g2.is_inf()) return fail(ctx, ZG_E_UNEXPECTED_IDENTITY, "vk.delta is the identity");
        uint32_t r[8], s[8], rs[8];
        if (!combine_scalars(r_le, s_le, r, s, rs))
            return fail(ctx, ZG_E_BAD_FORMAT, "r and s must be canonical scalars (< the group order)");
        combine_witness_part(p, r, s, rs, l_acc, a_ans, b1_ans, b2_ans, &own);
        early = &own;
    }
    const G1Affine& A = early->A;
    const G2Affine& B = early->B;
    const G1Affine C = early->c_part.add(h_acc).to_affine();
    if (A.is_inf() || B.is_inf() || C.is_inf())
        return fail(ctx, ZG_E_UNEXPECTED_IDENTITY, "a proof element is the identity (BadFormat(\"invalid point\") in za)");
    fq_to_be(proof_out, A.x); fq_to_be(proof_out + 32, A.y);
    fq_to_be(proof_out + 64, B.x.c0); fq_to_be(proof_out + 96, B.x.c1);
    fq_to_be(proof_out + 128, B.y.c0); fq_to_be(proof_out + 160, B.y.c1);
    fq_to_be(proof_out + 192, C.x); fq_to_be(proof_out + 224, C.y);
    ctx->timings[4] = now_ms() - t0;
    return ZG_OK;
}
extern "C" int zg_prove_combine(zg_ctx* ctx, const zg_params* p, const uint8_t* partials_all, int world,
                                const uint8_t r_le[32], const uint8_t s_le[32], uint8_t proof_out[256]) try {
    if (!ctx || !p || !partials_all || world < 1 || !r_le || !s_le || !proof_out) return ZG_E_INVALID_ARG;
    return combine_impl(ctx, p, partials_all, world, r_le, s_le, proof_out, nullptr);
} catch (...) {
    return zg::on_exception(ctx);
}

extern "C" int zg_prove_dev(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const void* d_witness,
                            const uint8_t r_le[32], const uint8_t s_le[32], uint8_t proof_out[256]) try {
    if (!ctx || !p || !r || !proof_out || !r_le || !s_le) return ZG_E_INVALID_ARG;
    if (!d_witness) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "witness is NULL");
    if (p->world != 1) return fail(ctx, ZG_E_INVALID_ARG, "zg_prove needs an unsharded key (world = 1); use the phase API");
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    double t0 = now_ms();
    if (!ctx->stats_keep) ctx->n_launches = 0;
    ctx->stats_keep = false;
    ctx->fold_armed_key = nullptr;
    if (ctx->prof_enabled && !ctx->prof_pending.empty()) ZG_TRY(prof_drain(ctx));   // recycle the last proof's events
    ZG_TRY(check_query_sizes(ctx, p, r));
    const size_t m = (size_t)1 << r->log_m;
    DevBuf& dh = prove_ws(ctx).h;
    ZG_HIP(ctx, dh.reserve(m * 32));
    // The witness MSMs do not depend on H: gate them on "now", queue the NTT pipeline, gate the h MSM
    // on its completion, and let the job scheduler overlap everything.
    ZG_TRY(gate_events(ctx));
    struct ClassGuard {   // whatever way this call ends, the class bytes are not taken for a later proof's
        ProveWs& ws;
        ~ClassGuard() { ws.wcls_armed = ws.wcls_fresh = ws.bad_pending = false; }
    } class_guard{prove_ws(ctx)};
    // H in evaluation form where the key has (or can get) its transformed h bases: six transforms instead of seven
    ZG_TRY(ensure_h_eval(ctx, p, r->log_m));
    const bool eval_form = h_eval_ready(p, r->log_m);
    // ... and four instead of six where the C rows' values are bits (ensure_c_fold): they leave the evaluator as scalars of
    // the l multiexp
    if (eval_form) ZG_TRY(ensure_c_fold(ctx, p, r, d_witness));
    const bool c_fold = eval_form && ctx->opt_c_fold && c_fold_ready(p, r) && c_fold_size_ok(ctx, r->log_m);
    CFoldBufs cfb;
    if (c_fold) ZG_TRY(c_fold_buffers(ctx, p, r, d_witness, c_fold_n_l(p, r), &cfb));
    ZG_TRY(witness_classes(ctx, r, d_witness));   // 26 us at 2^22; read by the evaluator and by the witness multiexps
    prove_ws(ctx).wcls_fresh = prove_ws(ctx).wcls_armed = true;
    ZG_HIP(ctx, hipEventRecord(ctx->gate_w, ctx->stream));
    ZG_HIP(ctx, hipEventRecord(ctx->tm_ev[0], ctx->stream));
    // The h multiexp of a large proof runs in parts (msm_batch5); in evaluation form the scalars of part k are final as
    // soon as the last transform kernel has passed them, so that kernel can be launched range by range with part k gated
    // on its range alone (ZG_OPT_H_PIPE).  Round 4, 2^22, A/B on one box: no difference (11.93 / 11.76 vs 11.77 / 11.83 ms: the
    // first part's digit sort and the rest of the transform kernel want the same CUs).  Re-measured at the end of round 5 with
    // the shorter pipeline (four transforms): 8.96 -> 8.87, 9.01 -> 8.90, 9.68 -> 9.57 (90 % bits), 9.82 -> 9.68 (six
    // transforms) witness resident, zg_prove -0.09 / -0.10 / -0.18 / -0.01 -- one per cent, every time: on (-1 = auto) where the
    // h multiexp runs in two parts anyway (a lone context, from 2^22 rows); 1 = from 2^20 rows (tests); 0 = off.
    int h_split = 0;
    const bool pipe_auto = ctx->opt_h_pipe < 0 && r->log_m >= 22;
    if ((ctx->opt_h_pipe > 0 || pipe_auto) && eval_form && r->log_m >= 14) {
        h_split = r->log_m >= (pipe_auto ? 22u : 20u) && live_contexts() == 1 ? 2 : 0;
        if (ctx->opt_h_parts > 0) h_split = std::min(ctx->opt_h_parts, (int)Ctx::H_PARTS_MAX);
        if (h_split == 1) h_split = 0;
    }
    const bool trace = trace_level() != 0;
    if (trace) fprintf(stderr, "[za_amd trace] prove_dev: classes queued      +%.3f ms\n", now_ms() - t0);
    ZG_TRY(prove_h_enqueue(ctx, r, d_witness, dh.p, ctx->tm_ev[1], eval_form, h_split, cfb.c_nat, cfb.c_cls, /*guard=*/ctx->opt_c_fold == 1));
    if (trace) fprintf(stderr, "[za_amd trace] prove_dev: R1CS + H queued     +%.3f ms\n", now_ms() - t0);
    ZG_HIP(ctx, hipEventRecord(ctx->tm_ev[2], ctx->stream));
    ZG_HIP(ctx, hipEventRecord(ctx->gate_h, ctx->stream));
    ctx->ev_witness = ctx->gate_w;
    ctx->ev_h = ctx->gate_h;
    uint8_t partials[ZG_PARTIALS_BYTES];
    // The combination's witness part (two scalar multiplications in G1, one in G2, Straus: 0.16 ms of host time) runs on the
    // thread that finishes the last witness multiexp, while the h multiexp is still on the GPU; only "+ sum h" and one
    // inversion are left for the end.  Whatever it cannot do (vk.delta at infinity, r or s out of range) is left to the
    // full combination below, which reports it.
    // What depends on r, s and the key only (three of the four scalar multiplications, the G2 one among them) is computed
    // by a helper thread from the start; the hook then has two additions and one two-point Straus pass left -- a small
    // proof's five multiexps end within 0.1 ms of each other, and there the hook is on the critical path (2^14: its 0.08 ms
    // -> 0.03).
    CombineEarly early;
    CombinePre pre;
    std::future<void> pre_done;   // (joins in its destructor: declared before the guard that clears the hook)
    struct HookGuard {
        Ctx* c;
        ~HookGuard() { c->witness_hook = nullptr; }
    } hook_guard{ctx};
    if (ctx->opt_combine_early) {
        pre_done = std::async(std::launch::async, [&] {
            try {
                combine_pre(p, r_le, s_le, &pre);
            } catch (...) {
                pre.ok = false;
            }
        });
        ctx->witness_hook = [&](const G1XYZZ* g1, const G2XYZZ* g2) {
            try {
                pre_done.wait();
                if (pre.ok) combine_witness_part_pre(p, pre, g1[1], g1[2], g1[3], *g2, &early);
            } catch (...) {
                early.done = false;
            }
        };
    }
    int rc_p = prove_partials_impl(ctx, p, r, d_witness, dh.p, 0x1f, partials, eval_form, h_split, c_fold ? &cfb : nullptr, ctx->tm_ev[1]);
    ctx->witness_hook = nullptr;
    if (trace) fprintf(stderr, "[za_amd trace] prove_dev: multiexps done      +%.3f ms\n", now_ms() - t0);
    ctx->ev_witness = ctx->ev_h = nullptr;
    ctx->witness_gate_armed = false;
    ZG_TRY(rc_p);
    {   // stream time of the R1CS evaluation and of the H pipeline (7 NTTs), which share the GPU with the witness MSMs
        float ms_eval = 0, ms_h = 0;
        (void)hipEventElapsedTime(&ms_eval, ctx->tm_ev[0], ctx->tm_ev[1]);
        (void)hipEventElapsedTime(&ms_h, ctx->tm_ev[1], ctx->tm_ev[2]);
        ctx->timings[1] = ms_eval;
        ctx->timings[2] = ms_h;
    }
    ZG_TRY(combine_impl(ctx, p, partials, 1, r_le, s_le, proof_out, &early));
    ctx->timings[5] = now_ms() - t0;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

static int pack_and_upload(zg_ctx* ctx, const uint8_t* witness32, size_t nv, bool* done, size_t* bytes_sent, size_t* first_raw);
// The device copy of the witness lives in the ctx's workspace (no allocation per proof) and the upload is simply
// queued in front of the proof's work: everything after it is gated by stream order / the witness event.  From
// pinned memory (zg_host_alloc) the copy runs at PCIe speed, 32 (l + v) bytes: 134 MB = 2.4 ms at 2^22 -- unless the
// witness is mostly zeros and ones, which cross PCIe in packed form (pack_and_upload).
// (A two-part upload with the l / a / b multiexps over the first half's positions started early was built and
// measured in round 2: slower, and eleven streams on eight hardware queues can deadlock -- DESIGN.md section 4.2.)
// r (zg_prove): a witness that goes over as it is (it does not pack: field elements) and is large is handed over in
// ProveWs::GATES chunks, each classified behind its copy, with an event per chunk: the l multiexp of the proof then runs
// in parts gated on them (msm_batch5), sorting and accumulating the first chunks' aux values while the rest is still on
// the bus -- the one stretch of such a proof in which the GPU idled (9.4 ms of 124 at 2^24, 37 of 483 at 2^26).  No new
// streams: the l parts take the side streams that further h parts would (a lone context).  From 2^23 values
// (ZG_OPT_UPLOAD_GATES): at 2^22 the 2.3 ms window buys what three jobs' latency-bound tails cost (33.33 vs 33.34 ms), at
// 2^24 124.5 -> 119.1 ms (profiles/r06_exp_upload_gates.txt).
static int witness_handover(zg_ctx* ctx, const uint8_t* witness32, size_t nv, void** d_w, size_t* sent, const zg_r1cs* r = nullptr) {
    ProveWs& ws = prove_ws(ctx);
    DevBuf& dw = ws.w;
    ZG_HIP(ctx, dw.reserve(nv * 32));
    bool packed = false;
    size_t first = 0;   // values already queued as they are by pack_and_upload (see there)
    *sent = nv * 32;
    ws.gate_for = ws.classes_done_for = nullptr;
    ws.gate_n = 0;
    ZG_TRY(pack_and_upload(ctx, witness32, nv, &packed, sent, &first));
    *d_w = dw.p;
    bool chunked = !packed && first == 0 && r && ctx->opt_upload_gates > 0 && nv >= ((size_t)1 << ctx->opt_upload_gates) &&
                   nv == (size_t)r->n_inputs + r->n_aux && live_contexts() == 1;
    if (chunked) {
        // the l parts sort on two more slots (msm.hip msm_scratch_bytes): about 128 bytes per value and part that have to be
        // free, or held by those slots already -- at 2^26 beside a key of that size they may not be
        size_t free_b = 0, total_b = 0;
        ZG_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
        const size_t have = msm_scratch_bytes(ctx, 6) + msm_scratch_bytes(ctx, 7);
        if (have + free_b < (size_t)256 * nv + ((size_t)4 << 30)) chunked = false;
    }
    if (chunked) {
        constexpr int K = ProveWs::GATES;
        ZG_TRY(witness_classes_begin(ctx, r));
        for (int k = 0; k < K; k++) {
            if (!ws.gate_ev[k]) ZG_HIP(ctx, hipEventCreateWithFlags(&ws.gate_ev[k], hipEventDisableTiming));
            const size_t lo = k == 0 ? 0 : ws.gate_var[k];
            const size_t hi = k + 1 == K ? nv : ((nv * (size_t)(k + 1) / K) & ~(size_t)255);   // (multiples of the subset-sum groups)
            ws.gate_var[k] = lo;
            ws.gate_var[k + 1] = hi;
            ZG_HIP(ctx, hipMemcpyAsync((uint8_t*)dw.p + lo * 32, witness32 + lo * 32, (hi - lo) * 32, hipMemcpyHostToDevice, ctx->stream));
            ZG_TRY(witness_classes_range(ctx, dw.p, lo, hi));
            ZG_HIP(ctx, hipEventRecord(ws.gate_ev[k], ctx->stream));
        }
        ZG_TRY(witness_classes_end(ctx, dw.p));
        ws.gate_n = K;
        ws.gate_for = ws.classes_done_for = dw.p;
        return ZG_OK;
    }
    if (!packed && first < nv)
        ZG_HIP(ctx, hipMemcpyAsync((uint8_t*)dw.p + first * 32, witness32 + first * 32, (nv - first) * 32, hipMemcpyHostToDevice,
                                   ctx->stream));
    return ZG_OK;
}
extern "C" int zg_prove(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const uint8_t* witness32,
                        const uint8_t r_le[32], const uint8_t s_le[32], uint8_t proof_out[256]) try {
    if (!ctx || !p || !r) return ZG_E_INVALID_ARG;
    if (!witness32) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "witness is NULL");
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    double t0 = now_ms();
    const size_t nv = (size_t)r->n_inputs + r->n_aux;
    void* d_w = nullptr;
    size_t sent = 0;
    ctx->n_launches = 0;
    ctx->stats_keep = true;   // (zg_prove_dev below counts on)
    // (a key with its C rows folded into the l query: their values are written behind the witness, c_fold_buffers)
    if (ctx->opt_c_fold && c_fold_ready(p, r)) ZG_HIP(ctx, prove_ws(ctx).w.reserve((nv + ((size_t)2 << r->log_m)) * 32));
    ZG_TRY(witness_handover(ctx, witness32, nv, &d_w, &sent, r));
    if (trace_level() != 0)
        fprintf(stderr, "[za_amd trace] zg_prove: witness handed over  +%.3f ms\n", now_ms() - t0);
    int rc = zg_prove_dev(ctx, p, r, d_w, r_le, s_le, proof_out);
    ctx->timings[0] = (double)sent;   // bytes uploaded (the copy is not timed separately: it is in [5])
    ctx->timings[5] = now_ms() - t0;
    return rc;
} catch (...) {
    return zg::on_exception(ctx);
}
// The same hand-over for callers that drive the phases themselves (a rank of a sharded proof, za_amd/dist.py): queued on
// the context's stream, not waited for.
extern "C" int zg_witness_to_device(zg_ctx* ctx, const uint8_t* witness32, size_t n, void** d_witness, size_t* bytes_sent) try {
    if (!ctx || !d_witness) return ZG_E_INVALID_ARG;
    if (!witness32 || n == 0) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "witness is NULL or empty");
    if (n >= ((size_t)1 << 32)) return fail(ctx, ZG_E_INVALID_ARG, "witness of %zu values", n);
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    size_t sent = 0;
    ZG_TRY(witness_handover(ctx, witness32, n, d_witness, &sent));
    if (bytes_sent) *bytes_sent = sent;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

// ---- packed witness (see include/za_amd.h)
// device side: the packed form crosses PCIe, two kernels rebuild the 32-byte array in the ctx's workspace
// (first > 0: the packed form describes the values [first, nv) only -- classes[0] is value `first`, other_index holds
// positions in the whole array; the values below `first` are put there by the caller)
static int upload_packed(zg_ctx* ctx, size_t nv, const uint8_t* classes, const uint32_t* other_index, const uint8_t* others32,
                         size_t n_others, size_t first = 0) {
    ProveWs& ws = prove_ws(ctx);
    const size_t np = nv - first;
    ZG_HIP(ctx, ws.w.reserve(nv * 32));
    if (np == 0) return ZG_OK;
    ZG_HIP(ctx, ws.pk_cls.reserve((np + 3) / 4));
    ZG_HIP(ctx, hipMemcpyAsync(ws.pk_cls.p, classes, (np + 3) / 4, hipMemcpyHostToDevice, ctx->stream));
    ZG_LAUNCH(ctx, k_unpack_bits, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, ws.pk_cls.as<uint8_t>(), np,
                       ws.w.as<Fr>() + first);
    if (n_others) {
        ZG_HIP(ctx, ws.pk_idx.reserve(n_others * 4));
        ZG_HIP(ctx, ws.pk_val.reserve(n_others * 32));
        ZG_HIP(ctx, hipMemcpyAsync(ws.pk_idx.p, other_index, n_others * 4, hipMemcpyHostToDevice, ctx->stream));
        ZG_HIP(ctx, hipMemcpyAsync(ws.pk_val.p, others32, n_others * 32, hipMemcpyHostToDevice, ctx->stream));
        ZG_LAUNCH(ctx, k_unpack_others, dim3((unsigned)((n_others + 255) / 256)), dim3(256), 0, ctx->stream,
                           ws.pk_idx.as<uint32_t>(), ws.pk_val.as<Fr>(), n_others, ws.w.as<Fr>());
    }
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}

// zg_prove's host side for a witness that is mostly zeros and ones (a SHA-256 circuit: > 99 %; config 4: 97 %): the
// 32 (l + v)-byte upload is the one stretch of a proof in which the GPU idles (2.3 ms of 14 at 2^22), and most of it
// carries zeros.  A few host threads turn the array into the packed form -- 2 bits per value plus (index, value) of
// everything else -- at memory speed, and 3 % of the bytes cross PCIe (pack_and_upload: *done = false, nothing or only a
// raw head queued, when the witness is not of that kind: then the rest is uploaded as it is).  ZG_OPT_PACK_THREADS sets the
// thread count.  Host time at 2^22 with 16 threads: 0.55-0.75 ms of scanning (a fifth to a quarter of the array goes over
// as it is meanwhile) + 0.07 ms to put the lists back to back + 0.03 ms of queueing.
static int host_pack_threads(const Ctx* ctx) {
    if (ctx->opt_pack_threads > 0) return ctx->opt_pack_threads;   // ZG_OPT_PACK_THREADS
    // Measured on the GPU box (256 hardware threads, shared with other jobs; 134 MB witness).  Alone, the pass takes
    // 1.37 ms with 4 threads, 0.82 with 8, 0.73 with 16, 0.56 with 32, 1.0 with 64 (the plain upload: 2.34 ms).  While the
    // pass waited for every worker to check in, more threads meant more chances of a late wake-up on a loaded host (round
    // 3: 8).  It now returns when the TASKS are done (WorkerPool), and a zg_prove at 2^22 on a host with a load average of
    // 52 takes 12.93 / 12.07 / 11.84 / 11.91 ms with 4 / 8 / 16 / 32 threads (tools/ab_opt.py, values alternated in one
    // process): 16, or half of a small host's threads; a host with fewer than 8 hardware threads uploads the array as it is.
    static const int n = [] {
        const unsigned hc = std::thread::hardware_concurrency();
        if (hc < 8) return 0;
        return (int)std::min<unsigned>(16, hc / 2);
    }();
    return n;
}
// the host packer on plain memory, no GPU: what the test hook zg_selftest_pack_mt (libza_amd_synth.so) calls
int zg::host_pack_for_test(const uint8_t* witness32, size_t nv, int threads, uint8_t* classes, uint32_t* other_index,
                           uint8_t* others32, size_t* n_others) {
    HostPacker hp;
    hp.run(witness32, nv, threads, classes, other_index, others32, n_others, [](const char*) {});
    return ZG_OK;
}
// ... with a stand-in for the copy engine: submit() copies into raw_out, ready() says yes every `ready_every`-th time it is
// asked (a slow or a fast engine).  *first = the values that went through it.
int zg::host_pack_split_for_test(const uint8_t* witness32, size_t nv, int threads, int cap_permille, int ready_every,
                                 uint8_t* raw_out, uint8_t* classes, uint32_t* other_index, uint8_t* others32, size_t* n_others,
                                 size_t* first) {
    HostPacker hp;
    HostPacker::RawCopier cp;
    size_t asked = 0, next_v0 = 0;
    bool in_order = true;
    cp.max_vals = (size_t)((double)nv * cap_permille / 1000.0);
    cp.ready = [&] { return ++asked % (size_t)std::max(ready_every, 1) == 0; };
    cp.submit = [&](size_t v0, size_t n) {
        in_order &= v0 == next_v0;   // pieces come in order, back to back, from value 0
        next_v0 = v0 + n;
        memcpy(raw_out + 32 * v0, witness32 + 32 * v0, 32 * n);
        return true;
    };
    hp.run(witness32, nv, threads, classes, other_index, others32, n_others, [](const char*) {}, {}, &cp, first);
    return in_order && next_v0 == *first ? ZG_OK : ZG_E_BAD_FORMAT;
}

static int pack_and_upload(zg_ctx* ctx, const uint8_t* witness32, size_t nv, bool* done, size_t* bytes_sent, size_t* first_raw) {
    *done = false;
    *first_raw = 0;
    if (ctx->opt_pack == 0) return ZG_OK;   // ZG_OPT_PACK
    const int nthreads = host_pack_threads(ctx);
    // below 2^18 values (8 MB, 0.15 ms of PCIe) waking the threads costs what the pass saves (ZG_OPT_PACK_MIN_LOG_N; the limit was
    // 2^20 while the pass had to beat the copy on its own: shared with the copy engine it pays earlier -- 2^18: 2.27 -> 1.99 ms,
    // 2^19: 3.25 -> 2.96, 2^17: 1.42 -> 1.40)
    if (nthreads <= 0 || nv < ((size_t)1 << ctx->opt_pack_min_log_n)) return ZG_OK;
    {   // a cheap look first: 1024 evenly spread values; a witness of field elements is left alone (8 %: below what makes
        // the pass itself give up, so a borderline witness does not flip between the two forms)
        size_t others = 0;
        const size_t step = nv / 1024;
        for (size_t k = 0; k < 1024; k++) others += value_class(witness32 + 32 * (k * step)) == 2u;
        if (others > 82) return ZG_OK;
    }
    const bool trace = trace_level() != 0;
    const double t_begin = trace ? now_ms() : 0;
    auto stamp = [&](const char* what) {
        if (trace) fprintf(stderr, "[za_amd trace] pack %-14s +%.3f ms\n", what, now_ms() - t_begin);
    };
    ProveWs& ws = prove_ws(ctx);
    if (ws.pack_skip > 0 && ws.pack_skip_nv == nv) {   // see below
        ws.pack_skip--;
        return ZG_OK;
    }
    ws.pack_skip = 0;
    const double t_pack0 = now_ms();
    const size_t max_others = nv / 8;
    const size_t cls_bytes = (nv + 3) / 4;
    const size_t need = ((cls_bytes + 63) & ~(size_t)63) + max_others * 36 + 64;
    const int sk = ws.stage_next;
    // a pass that first allocates (worker threads, per-slice lists, the pinned stage) says nothing about the host's load
    const bool cold = !ws.packer || ws.stage_bytes[sk] < need;
    if (!ws.packer) ws.packer.reset(new HostPacker());
    if (!ws.stage_ev[sk]) ZG_HIP(ctx, hipEventCreateWithFlags(&ws.stage_ev[sk], hipEventDisableTiming));
    else ZG_HIP(ctx, hipEventSynchronize(ws.stage_ev[sk]));   // the copies of the hand-over before last have left this stage
    if (ws.stage_bytes[sk] < need) {
        if (ws.stage[sk]) (void)hipHostFree(ws.stage[sk]);
        ws.stage[sk] = nullptr;
        ws.stage_bytes[sk] = 0;
        ZG_HIP(ctx, hipHostMalloc((void**)&ws.stage[sk], need, hipHostMallocDefault));
        ws.stage_bytes[sk] = need;
    }
    if (ctx->opt_pack_pin && !ws.pin_looked_up) {
        ws.pin_cpus = gpu_node_cpus(ctx->device);
        ws.pin_looked_up = true;
        if (trace) fprintf(stderr, "[za_amd trace] pack workers bound to %zu CPUs of the GPU's NUMA node\n", ws.pin_cpus.size());
    }
    // While the host threads pack, PCIe idles; while the packed form crosses, the threads idle.  So the copy engine takes
    // the array AS IT IS from the front, piece by piece, while the threads pack it from the back, until the two meet
    // (HostPacker::RawCopier): with the link at ~57 GB/s and the pass at ~190 GB/s the copier ends up with a fifth to a
    // quarter on a quiet host and with more on a loaded one.  ZG_OPT_PACK_RAW_PERMILLE caps its share (0 = everything
    // is packed).  Needs a pinned witness: a copy from pageable memory is staged by the calling thread.
    HostPacker::RawCopier copier;
    struct {
        int inflight = 0, oldest = 0;
        hipError_t err = hipSuccess;
        double due[2] = {0, 0};   // when each piece in flight is expected to have crossed (57 GB/s)
        double polled = 0;        // the last hipEventQuery
    } dma;
    bool use_copier = false;
    // (Only a lone context: with several provers on the GPU the link is the scarce thing and packing everything is better --
    // three provers, 9.8 ms per proof against 10.1; and their copiers' event polling got in each other's way: see ready().)
    if (ctx->opt_pack_raw_permille > 0 && live_contexts() == 1) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, witness32) == hipSuccess && at.type == hipMemoryTypeHost) {
            use_copier = true;
            ZG_HIP(ctx, ws.w.reserve(nv * 32));
            for (int k = 0; k < 2; k++)
                if (!ws.dma_ev[k]) ZG_HIP(ctx, hipEventCreateWithFlags(&ws.dma_ev[k], hipEventDisableTiming));
            copier.max_vals = (size_t)((double)nv * std::min(ctx->opt_pack_raw_permille, 900) / 1000.0);
            copier.ready = [&]() -> bool {
                if (dma.inflight < 2) return true;
                // The event is not asked before the piece can have arrived, and then at most every 10 us: a thread that
                // calls hipEventQuery in a tight loop holds the runtime's locks most of the time, and the other threads
                // of the process (three provers: their launches) waited for them -- proofs of 150-330 ms were seen.
                const double now = now_ms();
                if (now < dma.due[dma.oldest] || now - dma.polled < 0.010) return false;
                dma.polled = now;
                const hipError_t q = hipEventQuery(ws.dma_ev[dma.oldest]);
                if (q == hipErrorNotReady) return false;
                if (q != hipSuccess) dma.err = q;   // (submit reports it)
                dma.inflight--;
                dma.oldest ^= 1;
                return true;
            };
            copier.submit = [&](size_t v0, size_t n) -> bool {
                if (dma.err != hipSuccess) return false;
                dma.err = hipMemcpyAsync((uint8_t*)ws.w.p + v0 * 32, witness32 + v0 * 32, n * 32, hipMemcpyHostToDevice, ctx->stream);
                if (dma.err != hipSuccess) return false;
                const int slot = (dma.oldest + dma.inflight) & 1;
                dma.err = hipEventRecord(ws.dma_ev[slot], ctx->stream);
                // (pieces cross one after the other: this one is due when the one before it is, plus its own bytes)
                const double start = dma.inflight ? std::max(dma.due[slot ^ 1], now_ms()) : now_ms();
                dma.due[slot] = start + (double)n * 32.0 / 57e6;
                dma.inflight++;
                return dma.err == hipSuccess;
            };
        } else {
            (void)hipGetLastError();   // an unregistered pointer is an error to the attribute query, not to us
        }
    }
    static const std::vector<int> nowhere;
    uint8_t* classes = ws.stage[sk];
    uint32_t* idx_out = reinterpret_cast<uint32_t*>(ws.stage[sk] + ((cls_bytes + 63) & ~(size_t)63));
    uint8_t* val_out = reinterpret_cast<uint8_t*>(idx_out + max_others);
    size_t n_others = SIZE_MAX, first = 0;
    ws.packer->run(witness32, nv, nthreads, classes, idx_out, val_out, &n_others, stamp, ctx->opt_pack_pin ? ws.pin_cpus : nowhere,
                   use_copier ? &copier : nullptr, &first);
    (void)hipGetLastError();   // (hipEventQuery's "not ready")
    *first_raw = first;
    if (dma.err != hipSuccess) return fail(ctx, ZG_E_HIP, "witness hand-over: %s", hipGetErrorString(dma.err));
    if (n_others == SIZE_MAX) {   // not the kind of witness that packs: no point in scanning the next ones either
        ws.pack_skip = 32;
        ws.pack_skip_nv = nv;
        return ZG_OK;
    }
    ZG_TRY(upload_packed(ctx, nv, classes + first / 4, idx_out, val_out, n_others, first));
    ZG_HIP(ctx, hipEventRecord(ws.stage_ev[sk], ctx->stream));
    ws.stage_next = sk ^ 1;
    stamp("queued");
    if (ctx->opt_pack == 2) {
        // ZG_OPT_PACK = 2: the pass competes with whatever else runs on the host (on a shared 256-thread host with a load
        // average of 30 it has been seen to take 2.7 ms instead of 0.8): when its running average exceeds what the plain
        // upload costs (32 B per value at ~55 GB/s), the next 32 proofs go over PCIe unpacked, then it is tried again.
        // (ZG_OPT_PACK_BUDGET_PERMILLE: the share of the plain upload's time the pass may take, default 900; the tests
        // set 0 to walk through the fallback.)  The default, ZG_OPT_PACK = 1, never looks at the clock.
        const double budget = ctx->opt_pack_budget_permille / 1000.0;
        const double t = now_ms() - t_pack0, plain_ms = (double)nv * 32.0 / 55e6 * budget / 0.9;
        if (!cold && nv >= ((size_t)1 << 20)) {
            ws.pack_ema_ms = ws.pack_ema_ms > 0 ? 0.7 * ws.pack_ema_ms + 0.3 * t : t;
            ws.pack_slow = t > 0.9 * plain_ms ? ws.pack_slow + 1 : 0;
            if (ws.pack_slow >= 2 && ws.pack_ema_ms > 0.9 * plain_ms) {   // two slow passes in a row, not one late wake-up
                ws.pack_skip = 32;
                ws.pack_skip_nv = nv;
                ws.pack_ema_ms = 0;
                ws.pack_slow = 0;
            }
        }
    }
    *done = true;
    *bytes_sent = first * 32 + (nv - first + 3) / 4 + n_others * 36;
    return ZG_OK;
}

extern "C" int zg_witness_pack(const uint8_t* witness32, size_t n, uint8_t* classes, uint32_t* other_index,
                               uint8_t* others32, size_t cap, size_t* n_others) try {
    if (!witness32 || !classes || !n_others || (cap && (!other_index || !others32))) return ZG_E_INVALID_ARG;
    if (n >= ((size_t)1 << 32)) return ZG_E_INVALID_ARG;
    memset(classes, 0, (n + 3) / 4);
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        const uint8_t* v = witness32 + 32 * i;
        uint64_t q[4];
        memcpy(q, v, 32);
        uint32_t c = 2;
        if ((q[1] | q[2] | q[3]) == 0 && q[0] <= 1) c = (uint32_t)q[0];
        if (c == 2) {
            if (k < cap) {
                other_index[k] = (uint32_t)i;
                memcpy(others32 + 32 * k, v, 32);
            }
            k++;
        }
        classes[i >> 2] |= (uint8_t)(c << (2 * (i & 3)));
    }
    *n_others = k;
    return k > cap ? ZG_E_BUFFER_TOO_SMALL : ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

extern "C" int zg_prove_packed(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const uint8_t* classes,
                               const uint32_t* other_index, const uint8_t* others32, size_t n_others,
                               const uint8_t r_le[32], const uint8_t s_le[32], uint8_t proof_out[256]) try {
    if (!ctx || !p || !r) return ZG_E_INVALID_ARG;
    if (!classes || (n_others && (!other_index || !others32))) return fail(ctx, ZG_E_ASSIGNMENT_MISSING, "packed witness is NULL");
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    double t0 = now_ms();
    const size_t nv = (size_t)r->n_inputs + r->n_aux;
    {   // the packed form must describe exactly one value per variable: class 3 does not exist, the class-2 positions
        // are the listed ones (strictly increasing), and variable 0 is ONE
        size_t twos = 0;
        const size_t nbytes = (nv + 3) / 4;
        auto word = [&](size_t b0, size_t len) {   // up to 8 class bytes as one integer, padding bits cleared
            uint64_t x = 0;
            memcpy(&x, classes + b0, len);
            if (b0 + len == nbytes && (nv & 3)) x &= ~(~(uint64_t)0 << (8 * (len - 1) + 2 * (nv & 3)));
            return x;
        };
        for (size_t b = 0; b < nbytes; b += 8) {
            const uint64_t x = word(b, std::min<size_t>(8, nbytes - b));
            if (x & (x >> 1) & 0x5555555555555555ull)
                return fail(ctx, ZG_E_BAD_FORMAT, "packed witness: class 3 among the values %zu..%zu", 4 * b, 4 * b + 31);
            twos += (size_t)__builtin_popcountll(x & 0xAAAAAAAAAAAAAAAAull);
        }
        if (twos != n_others)
            return fail(ctx, ZG_E_BAD_FORMAT, "packed witness: %zu values of class 2 but %zu listed", twos, n_others);
        for (size_t j = 0; j < n_others; j++) {
            const uint32_t i = other_index[j];
            if (i >= nv || ((classes[i >> 2] >> (2 * (i & 3))) & 3u) != 2u || (j && other_index[j - 1] >= i))
                return fail(ctx, ZG_E_BAD_FORMAT, "packed witness: other_index[%zu] = %u is not the next class-2 position", j, i);
        }
        if ((classes[0] & 3u) != 1u) return fail(ctx, ZG_E_BAD_FORMAT, "packed witness: variable 0 must be ONE");
    }
    ProveWs& ws = prove_ws(ctx);
    ctx->n_launches = 0;
    ctx->stats_keep = true;
    ZG_TRY(upload_packed(ctx, nv, classes, other_index, others32, n_others));
    int rc = zg_prove_dev(ctx, p, r, ws.w.p, r_le, s_le, proof_out);
    ctx->timings[0] = (double)((nv + 3) / 4 + n_others * 36);   // bytes uploaded
    ctx->timings[5] = now_ms() - t0;
    return rc;
} catch (...) {
    return zg::on_exception(ctx);
}

extern "C" int zg_prove_random(zg_ctx* ctx, const zg_params* p, const zg_r1cs* r, const uint8_t* witness32,
                               uint8_t proof_out[256]) try {
    if (!ctx) return ZG_E_INVALID_ARG;
    std::random_device rd;
    uint8_t rs[2][32];
    for (int k = 0; k < 2; k++) {
        // uniform in [0, r) as Fr::rand: 254 random bits, rejected while >= r (r ~ 0.756 * 2^254)
        for (;;) {
            uint32_t v[8];
            for (int i = 0; i < 8; i++) v[i] = rd();
            v[7] &= 0x3fffffffu;
            if (scalar_is_canonical(v)) {
                memcpy(rs[k], v, 32);
                break;
            }
        }
    }
    return zg_prove(ctx, p, r, witness32, rs[0], rs[1], proof_out);
} catch (...) {
    return zg::on_exception(ctx);
}

// sum of `count` Jacobian points (the 96-byte G1 form zg_msm_* return): the "reduce" of a sharded standalone multiexp
// after the ranks' partial sums have been gathered (RCCL has no point-add reduction)
extern "C" int zg_g1_sum(const uint8_t* jac96, size_t count, uint8_t out_jac[96]) try {
    if (!out_jac || (count && !jac96)) return ZG_E_INVALID_ARG;
    G1XYZZ acc = G1XYZZ::inf();
    for (size_t k = 0; k < count; k++) acc = acc.add(jac_from_bytes<Fq>(jac96 + 96 * k));
    jac_bytes(out_jac, acc);
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

extern "C" int zg_proof_json(const uint8_t proof[256], const uint8_t* public_inputs32, size_t n_public, char* out,
                             size_t cap) try {
    if (!proof || !out || (n_public && !public_inputs32)) return ZG_E_INVALID_ARG;
    std::string j;
    proof_json(proof, public_inputs32, n_public, &j);
    if (j.size() >= cap) return ZG_E_BUFFER_TOO_SMALL;
    memcpy(out, j.c_str(), j.size() + 1);
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

// the scaffolding library's way in (csrc/internal_api.h): the one non-function symbol libza_amd.so exports
#if !defined(__HIP_DEVICE_COMPILE__)   // (host data: the device pass must not emit it)
extern "C" const zg_internal_api zg_internal_v1 = {
    zg::fail_msg,           zg::on_exception,           zg::convert_mont_dev,     zg::host_pack_for_test,
    zg::host_pack_split_for_test, zg::params_h_eval_slice, zg::parse_bellman_params, zg::write_bellman_params,
    zg::pairing_product_is_one,
};
#endif
