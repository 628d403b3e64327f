// Opaque handle layouts behind include/za_amd.h (shared by prover.hip and synth.hip).
#pragma once
#include <mutex>
#include <vector>

#include "common.h"

struct zg_ctx : public zg::Ctx {};

struct zg_r1cs {
    uint32_t n_inputs = 0, n_aux = 0;
    uint64_t n_constraints = 0;
    uint32_t log_m = 0;
    zg::DevBuf row_ptr[3], col[3], coeff[3];
    zg::DevBuf ccls[3];   // per-term coefficient class (prover.hip k_coeff_prepare)
    // rows with more than R1CS_LONG_ROW terms (a 32-bit adder's linear row has ~200): the tiled evaluator leaves them
    // to k_r1cs_eval_long, one wave per row
    zg::DevBuf long_rows[3];
    uint32_t n_long[3] = {0, 0, 0};
    // dense position -> flat variable index for the a and b queries
    zg::DevBuf a_index, b_index;
    // the inverse maps: variable -> dense position in the a / b query, ~0 = not in its density (empty = full density: the
    // identity); host copies of the index lists give a sharded key's variable range
    zg::DevBuf a_rank, b_rank;
    std::vector<uint32_t> a_index_host, b_index_host;
    uint32_t a_first = 0, b_first = 0;
    uint64_t n_a = 0, n_b = 0;
    uint64_t a_aux_total = 0, b_in_total = 0, b_aux_total = 0;
    std::vector<uint32_t> var_of_signal;   // only for circuits loaded from a za proving key
};

struct zg_params {
    zg::G1Affine alpha_g1, beta_g1, delta_g1;
    zg::G2Affine beta_g2, delta_g2;
    // the rest of the verifying key (kept when the key came from Parameters::read or from zg_setup; empty for keys built
    // from raw arrays): zg_params_write needs it
    zg::G2Affine gamma_g2 = zg::G2Affine::inf();
    std::vector<zg::G1Affine> ic;
    struct Query {
        zg::DevBuf buf;
        uint64_t total = 0, lo = 0, hi = 0;
    } q[5];  // h, l, a, b_g1, b_g2
    int rank = 0, world = 1;
    // fixed-base window tables (msm.hip), one per query, built on first use / by zg_params_prepare
    struct Fbw {
        zg::DevBuf buf;
        int c = 0, nwin = 0;
        int kind = 0;    // 0 = window table, 1 = subset-sum table
        int qw = 4;      // subset-sum table: positions per group (4 or 8)
        // subset-sum table with a compact window table behind it (msm.hip MsmJob::cw_rank): cw_nwin rows of cw_n points
        // from entry cw_off of buf on; rank: column per position
        zg::DevBuf rank;
        uint32_t cw_off = 0, cw_n = 0;
        int cw_c = 0, cw_nwin = 0;
        uint64_t n = 0;
        int state = 0;   // 0 = not tried, 1 = built, -1 = disabled / does not fit / does not pay
        bool small = false;   // planned for a latency-bound job (msm.hip small-job mode): decided with the table, when the
                              // number of "real" scalars of the query is known
    };
    mutable Fbw fbw[5];
    mutable std::mutex fbw_mu;
    // the common plan of the witness queries' window tables (ensure_tables): queries with it share one digit sort per proof
    mutable int w_plan_c = 0, w_plan_nwin = 0;
    mutable bool w_plan_small = false;
    // The h query in evaluation form (heval.hip): m affine points that go with h_eval_dev's scalars, built once per key
    // (unsharded keys only) together with their own window table.  state as in Fbw.
    // A sharded key (world > 1) holds the positions [h_eval_lo, h_eval_hi) of the m-point array -- chunk `rank` of
    // ceil(m / world) -- made when the key is loaded or cut from a full key, while the whole query is at hand.
    mutable zg::DevBuf h_eval;
    mutable uint32_t h_eval_log_m = 0;
    mutable int h_eval_state = 0;
    mutable uint64_t h_eval_lo = 0, h_eval_hi = 0;
    mutable Fbw fbw_he;
    // "C folded" (prover.hip ensure_c_fold; unsharded keys with the evaluation form): the l query followed by the m bases
    // that go with the C rows' values and their doubles (heval.hip c_fold) as ONE query of lc_n_l + 2 m points, and its table.  States as
    // h_eval_state.
    mutable zg::DevBuf lc;
    mutable int lc_state = 0;
    mutable uint32_t lc_log_m = 0;
    mutable uint64_t lc_n_l = 0;
    mutable Fbw fbw_lc;
    // rank 0's shard of a sharded key gets `lc` when it is loaded, before any witness is seen (the group transform needs the
    // whole h query): whether proofs USE it is decided with the first witness (0 not yet, 1 fold, -1 six transforms)
    mutable int lc_rule = 0;
};


// Resident multiexp bases (bellman's `Arc<Vec<G::Affine>>` source, kept in HBM in device form) with an optional
// fixed-base window table; see zg_msm_bases_* in include/za_amd.h.
struct zg_msm_bases {
    int device = 0;
    bool g2 = false;
    uint64_t n = 0;
    zg::DevBuf bases, table;
    int c = 0, nwin = 0;
    // staging for host scalars: device buffer + copy stream + one event per part
    zg::DevBuf scalars;
    hipStream_t copy_stream = nullptr;
    hipEvent_t chunk_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    ~zg_msm_bases() {
        for (auto& e : chunk_ev)
            if (e) (void)hipEventDestroy(e);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
};
