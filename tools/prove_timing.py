"""zg_prove (pinned host witness) against zg_prove_dev: wall time and the event-timed R1CS evaluation / H pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
R = sr.load(ctx)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
d_w = ctx.upload(sr.witness)
P.prepare(R, d_w)
for mode in ("dev", "host", "dev", "host"):
    tm = []
    for i in range(13):
        t0 = time.perf_counter()
        if mode == "dev":
            ctx.prove_dev(P, R, d_w, 5, 7)
        else:
            ctx.prove(P, R, pin, 5, 7)
        tm.append([1e3 * (time.perf_counter() - t0)] + list(ctx.last_timings().values()))
    tm = np.array(tm[3:])
    print(mode, "wall %.3f" % tm[:, 0].mean(), "timings", np.round(tm[:, 1:].mean(axis=0), 3))
