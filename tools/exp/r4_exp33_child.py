"""one process = one stream assignment (ZA_AMD_JSLOTS): median / min of zg_prove for config-4-shaped proofs of the sizes given"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import za_amd
from za_amd import synth
ctx = za_amd.Context(0)
out = []
for lg in [int(a) for a in sys.argv[1:]]:
    sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
    R = sr.load(ctx)
    P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
    pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
    for _ in range(4):
        ctx.prove(P, R, pin, 5, 7)
    ts = []
    for _ in range(24):
        t0 = time.perf_counter(); ctx.prove(P, R, pin, 5, 7); ts.append(1e3 * (time.perf_counter() - t0))
    out.append("2^%d median %.3f min %.3f" % (lg, np.median(ts), np.min(ts)))
    P.close(); R.close(); sr.close(); ctx.host_free(pin)
print(os.environ.get("ZA_AMD_JSLOTS", "default"), " | ".join(out))
