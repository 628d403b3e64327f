"""R1CS evaluation + H pipeline alone, both forms: python tools/h_bench.py [log_m] [bit_fraction]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import za_amd
from za_amd import synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
bf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.97
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(lg, 2, bf, 0x5A)
R = sr.load(ctx)
d_w = ctx.upload(sr.witness)
d_h = ctx.alloc(32 << lg)
for name, fn in (("coefficient form, 7 transforms", ctx.prove_h_dev), ("evaluation form, 6 transforms", ctx.prove_h_eval_dev)) * 2:
    for _ in range(3):
        fn(R, d_w, d_h)
    t = time.perf_counter()
    for _ in range(10):
        fn(R, d_w, d_h)
    print("R1CS evaluation + H pipeline (%s): %.3f ms" % (name, 1e3 * (time.perf_counter() - t) / 10))
ctx.profile_enable(2)
ctx.profile_reset()
for _ in range(4):
    ctx.prove_h_eval_dev(R, d_w, d_h)
p = ctx.profile_get()
print("per call: r1cs_eval %.3f ms, transforms %.3f ms in %d launches" % (p["r1cs_eval"][0] / 4, p["ntt_pass"][0] / 4, p["ntt_pass"][1] // 4))
