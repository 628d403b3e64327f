"""What the per-kernel profile (zg_profile_enable) costs a proof whose witness crosses PCIe as it is: a 2^lg proof of uniform
field elements through zg_prove without the profile, with it, right after zg_profile_reset and after idle gaps.
    python tools/profile_cost_probe.py [log_m] [level]
(Round 6: the drain of the previous proof's events synchronised the stream the next witness was already on.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(lg, 2, 0.0, 0x26)
R = sr.load(ctx)
wb = np.array(sr.witness, copy=True)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
pin = ctx.host_alloc(len(wb)); pin[:] = wb
def run(n):
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); ctx.prove(P, R, pin, 5, 7); ts.append(round(1e3 * (time.perf_counter() - t0), 2))
    return ts
print("no profile   ", run(6)[1:])
ctx.profile_enable(level)
print("profile on   ", run(6))
ctx.profile_reset()
print("after reset  ", run(8))
ctx.profile_reset()
print("after reset  ", run(8))
time.sleep(2.0)
print("after 2 s idle", run(6))
ctx.profile_enable(0)
print("profile off  ", run(6))
time.sleep(2.0)
print("off, after 2 s idle", run(6))
