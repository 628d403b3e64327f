"""The phases one rank of a W-way sharded proof runs, one after the other on this GPU (bench.py's shard_model alone):
python tools/shard_model_run.py [W [log_m [bit_fraction [seed]]]]   -- ZA_AMD_OPTIONS="12=0" etc. (test binding) forces
schedule options.  Defaults: BASELINE config 4 (2^22, 97 % bits, seed 0x5A); config 5 is `8 26 0 0x26`.
With --msm LOG_N as the last arguments: also the standalone G1 multiexp of BASELINE's metric -- 2^LOG_N resident bases on one
GPU against ONE rank's 2^LOG_N / W points (scalars uploaded from pinned host memory in both).  One GPU: a model of one rank's
share, not a multi-GPU measurement."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth
argv = list(sys.argv[1:])
msm_log_n = 0
if "--msm" in argv:
    k = argv.index("--msm")
    msm_log_n = int(argv[k + 1])
    del argv[k:k + 2]
W = int(argv[0]) if len(argv) > 0 else 8
lg = int(argv[1]) if len(argv) > 1 else 22
bitf = float(argv[2]) if len(argv) > 2 else 0.97
seed = int(argv[3], 0) if len(argv) > 3 else 0x5A
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(lg, 2, bitf, seed)
R = sr.load(ctx)
m = 1 << lg
P = synth.synth_params(ctx, m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 7000)
d_w = ctx.upload(sr.witness)
P.prepare(R, d_w)                      # tables and the evaluation-form h bases of the full key
big = lg >= 25                         # the full key's tables and the shard's do not fit side by side: one after the other
d_h = ctx.alloc(32 << lg)
ctx.prove_h_eval_dev(R, d_w, d_h)
full = []
for _ in range(5):
    t1 = time.perf_counter(); ctx.prove_partials_dev(P, R, d_w, d_h, 0x1e)
    t2 = time.perf_counter(); ctx.prove_partials_dev(P, R, d_w, d_h, 0x21)
    full.append((t2 - t1, time.perf_counter() - t2))
full = np.array(full[2:]) * 1e3
both_full = []
for _ in range(5):
    t1 = time.perf_counter(); ctx.prove_partials_dev(P, R, d_w, d_h, 0x3f)
    both_full.append(time.perf_counter() - t1)
both_full = 1e3 * float(np.mean(both_full[2:]))
Pk = synth.params_shard(ctx, P, 0, W)
if big:
    P.close()
Pk.prepare(R, d_w)
ts = []
for _ in range(6):
    t1 = time.perf_counter(); ctx.prove_h_eval_dev(R, d_w, d_h)
    t2 = time.perf_counter(); ctx.prove_partials_dev(Pk, R, d_w, d_h, 0x1e)
    t3 = time.perf_counter(); ctx.prove_partials_dev(Pk, R, d_w, d_h, 0x21)
    t4 = time.perf_counter(); ts.append((t2 - t1, t3 - t2, t4 - t3))
ts = np.array(ts[2:]) * 1e3
both = []
for _ in range(6):
    t1 = time.perf_counter(); ctx.prove_partials_dev(Pk, R, d_w, d_h, 0x3f)
    both.append(time.perf_counter() - t1)
both = 1e3 * float(np.mean(both[2:]))
print("all five jobs queued together: shard %.3f ms, unsharded %.3f ms: %.2fx" % (both, both_full, both_full / both))
print("config: 2^%d rows, bit fraction %.2f, seed 0x%x" % (lg, bitf, seed))
print("W=%d shard: h_ntt %.3f witness %.3f h_msm %.3f | unsharded witness %.3f h_msm %.3f | msm scaling %.2fx; tables %s"
      % (W, ts[:, 0].mean(), ts[:, 1].mean(), ts[:, 2].mean(), full[:, 0].mean(), full[:, 1].mean(),
         full.sum(axis=1).mean() / ts[:, 1:].sum(axis=1).mean(), [Pk.table_info(k) for k in range(5)]))

if msm_log_n:
    ctx.free(d_h); ctx.free(d_w)
    Pk.close(); R.close(); sr.close()
    if not big:
        P.close()
    res = {}
    for name, n in (("one rank's 2^%d / %d points" % (msm_log_n, W), (1 << msm_log_n) // W), ("all 2^%d points" % msm_log_n, 1 << msm_log_n)):
        d_b = synth.synth_bases_g1(ctx, n, 4242)
        d_s = synth.synth_scalars(ctx, n, 99)
        H = ctx.msm_bases_g1_dev(d_b, n, table_budget_gb=120.0)
        ctx.free(d_b)
        s_host = ctx.host_alloc(32 * n)
        s_host[:] = ctx.d2h(d_s, 32 * n)
        ctx.free(d_s)
        ctx.msm_resident(H, s_host, n, True)
        tt = []
        for _ in range(5):
            t1 = time.perf_counter(); ctx.msm_resident(H, s_host, n, True); tt.append(time.perf_counter() - t1)
        res[name] = 1e3 * float(np.median(tt))
        info = ctx.msm_bases_info(H)
        print("g1_msm, %s, scalars uploaded: %.2f ms (table %.1f GB, c = %d, %d digits)" % (name, res[name], info["table_bytes"] / 1e9, info["c"], info["nwin"]))
        ctx.host_free(s_host)
        ctx.msm_bases_free(H)
    v = list(res.values())
    print("g1_msm scaling model at W = %d: %.2fx" % (W, v[1] / v[0]))
