"""One traced proof (ZA_AMD_TRACE=1 host stamps of the job scheduler): python tools/trace_run.py [log_m] [bit_fraction]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth
ctx = za_amd.Context(0)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
bf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.97
sr = synth.SynthR1CS(lg, 2, bf, 0x5A)
R = sr.load(ctx)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
for i in range(3):
    ctx.prove(P, R, pin, 5, 7)
sys.stderr.write("=== traced proof\n"); sys.stderr.flush()
os.environ["ZA_AMD_TRACE_NOW"] = "1"
t0 = time.perf_counter()
ctx.prove(P, R, pin, 5, 7)
sys.stderr.write("=== total %.3f ms\n" % (1e3 * (time.perf_counter() - t0)))
