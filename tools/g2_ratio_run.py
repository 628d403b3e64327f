"""Standalone plain Pippenger over 2^n uniform scalars, G1 and G2, for the G2 : G1 cost ratio."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import za_amd
from za_amd import synth
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 21)
ctx = za_amd.Context(0)
d_b1 = synth.synth_bases_g1(ctx, n, 4242)
d_b2 = synth.synth_bases_g2(ctx, n, 4243)
d_s = synth.synth_scalars(ctx, n, 99)
ctx.profile_enable(2)
for name, fn, d_b in (("g1", ctx.msm_g1_dev, d_b1), ("g2", ctx.msm_g2_dev, d_b2)):
    for i in range(4):
        ctx.profile_reset()
        t0 = time.perf_counter()
        fn(d_b, d_s, n)
        dt = 1e3 * (time.perf_counter() - t0)
        print("%s %d: %.3f ms" % (name, i, dt), ctx.profile_get() if i == 3 else "")
