"""Wall time of each witness multiexp job of a 2^22 config-4 proof run alone (mask bit per query)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import za_amd
from za_amd import synth
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(22, 2, 0.97, 0x5A)
R = sr.load(ctx)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
d_w = ctx.upload(sr.witness)
P.prepare(R, d_w)
d_h = ctx.alloc(32 << 22)
ctx.prove_h_dev(R, d_w, d_h)
for name, mask in (("l", 2), ("a", 4), ("b_g1", 8), ("b_g2", 16), ("all four", 0x1e), ("h", 1)):
    ts = []
    for i in range(6):
        t0 = time.perf_counter()
        ctx.prove_partials_dev(P, R, d_w, d_h, mask)
        ts.append(1e3 * (time.perf_counter() - t0))
    print("%-8s %.3f ms (min %.3f)" % (name, sum(ts[2:]) / 4, min(ts)))
