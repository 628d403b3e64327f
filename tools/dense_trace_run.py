"""A 2^n proof with a witness of uniform field elements (config 5's shape), a few times: for kernel traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import za_amd
from za_amd import synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = za_amd.Context(0)
sd = synth.SynthR1CS(lg, 2, 0.0, 0x5A)
rd = sd.load(ctx)
Pd = ctx.setup(rd, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
d_w = ctx.upload(sd.witness)
Pd.prepare(rd, d_w)
for i in range(4):
    t0 = time.perf_counter()
    ctx.prove_dev(Pd, rd, d_w, 12345, 67890)
    print("proof %d: %.3f ms" % (i, 1e3 * (time.perf_counter() - t0)))
