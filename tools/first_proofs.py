"""Wall times of the first proofs of a fresh context (one-time work: table builds, pools, events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import za_amd
from za_amd import synth
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(22, 2, 0.97, 0x5A)
R = sr.load(ctx)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
d_w = ctx.upload(pin)
P.prepare(R, d_w)
if len(sys.argv) > 1:
    ctx.profile_enable(int(sys.argv[1]))
ts = []
for i in range(12):
    t0 = time.perf_counter()
    ctx.prove(P, R, pin, 5, 7)
    ts.append(round(1e3 * (time.perf_counter() - t0), 2))
print(ts)
