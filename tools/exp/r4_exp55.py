"""how often does a step take much longer than the median, and does it depend on the hand-over / combination options?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import za_amd
from za_amd import synth
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(22, 2, 0.97, 0x5A)
R = sr.load(ctx)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
d_w = ctx.upload(sr.witness)
for _ in range(5):
    ctx.prove(P, R, pin, 5, 7)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 250
SETTINGS = ((400, 1, 1), (400, 1, 0), (400, 1, 1), (400, 1, 0)) if os.environ.get("R4_PIN_AB") else ((400, 1, 1), (0, 1, 1), (400, 0, 1), (0, 0, 1))
for raw, early, pin_on in SETTINGS:
    ctx.set_option(10, raw); ctx.set_option(11, early); ctx.set_option(5, pin_on)
    ts = []
    for _ in range(N):
        t0 = time.perf_counter(); ctx.prove(P, R, pin, 5, 7); ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.array(ts); med = np.median(ts)
    print("pin %d raw cap %3d early %d: median %.3f p99 %.3f max %.3f, steps > median + 1 ms: %d of %d %s" % (pin_on, raw, early, med, np.percentile(ts, 99), ts.max(), int((ts > med + 1).sum()), N, np.round(ts[ts > med + 1], 1)))
ts = []
for _ in range(N):
    t0 = time.perf_counter(); ctx.prove_dev(P, R, d_w, 5, 7); ts.append(1e3 * (time.perf_counter() - t0))
ts = np.array(ts); med = np.median(ts)
print("zg_prove_dev: median %.3f max %.3f, > median + 1 ms: %d %s" % (med, ts.max(), int((ts > med + 1).sum()), np.round(ts[ts > med + 1], 1)))
