"""The phases one rank of a W-way sharded 2^22 proof runs, one after the other on this GPU (bench.py's shard_model alone):
python tools/shard_model_run.py [W]   -- ZA_AMD_OPTIONS="12=0" etc. (test binding) forces schedule options."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lg = 22
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
R = sr.load(ctx)
m = 1 << lg
P = synth.synth_params(ctx, m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 7000)
d_w = ctx.upload(sr.witness)
P.prepare(R, d_w)                      # tables and the evaluation-form h bases of the full key
Pk = synth.params_shard(ctx, P, 0, W)
Pk.prepare(R, d_w)
d_h = ctx.alloc(32 << lg)
ctx.prove_h_eval_dev(R, d_w, d_h)
full = []
for _ in range(5):
    t1 = time.perf_counter(); ctx.prove_partials_dev(P, R, d_w, d_h, 0x1e)
    t2 = time.perf_counter(); ctx.prove_partials_dev(P, R, d_w, d_h, 0x21)
    full.append((t2 - t1, time.perf_counter() - t2))
full = np.array(full[2:]) * 1e3
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
    t2 = time.perf_counter(); ctx.prove_partials_dev(P, R, d_w, d_h, 0x3f)
    both.append((t2 - t1, time.perf_counter() - t2))
both = np.array(both[2:]) * 1e3
print("all five jobs queued together: shard %.3f ms, unsharded %.3f ms: %.2fx" % (both[:, 0].mean(), both[:, 1].mean(), both[:, 1].mean() / both[:, 0].mean()))
print("W=%d shard: h_ntt %.3f witness %.3f h_msm %.3f | unsharded witness %.3f h_msm %.3f | msm scaling %.2fx; tables %s"
      % (W, ts[:, 0].mean(), ts[:, 1].mean(), ts[:, 2].mean(), full[:, 0].mean(), full[:, 1].mean(),
         full.sum(axis=1).mean() / ts[:, 1:].sum(axis=1).mean(), [Pk.table_info(k) for k in range(5)]))
