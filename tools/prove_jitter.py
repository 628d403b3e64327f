"""Per-proof wall times of N consecutive zg_prove calls (pinned host witness): mean, percentiles, outliers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(22, 2, 0.97, 0x5A)
R = sr.load(ctx)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
for _ in range(5):
    ctx.prove(P, R, pin, 5, 7)
ts = []
for i in range(N):
    t0 = time.perf_counter()
    ctx.prove(P, R, pin, 5, 7)
    ts.append(1e3 * (time.perf_counter() - t0))
ts = np.array(ts)
print("n %d mean %.3f median %.3f p90 %.3f p99 %.3f max %.3f; > median + 1 ms: %d" %
      (N, ts.mean(), np.median(ts), np.percentile(ts, 90), np.percentile(ts, 99), ts.max(), int((ts > np.median(ts) + 1).sum())))
print("outliers", np.round(ts[ts > np.median(ts) + 1], 2).tolist()[:20])
