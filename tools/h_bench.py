"""R1CS evaluation + H pipeline alone, both forms: python tools/h_bench.py [log_m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import za_amd
from za_amd import synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
R = sr.load(ctx)
d_w = ctx.upload(sr.witness)
d_h = ctx.alloc(32 << lg)
for name, fn in (("coefficient form, 7 transforms", ctx.prove_h_dev), ("evaluation form, 6 transforms", ctx.prove_h_eval_dev)) * 2:
    for _ in range(3):
        fn(R, d_w, d_h)
    t = time.perf_counter()
    for _ in range(20):
        fn(R, d_w, d_h)
    print("R1CS evaluation + H pipeline (%s): %.3f ms" % (name, 1e3 * (time.perf_counter() - t) / 20))
