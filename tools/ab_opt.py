"""A/B of a zg_set_option value inside ONE process:  python tools/ab_opt.py OPTION_NUMBER v1,v2,... [log_m] [rounds] [proofs]
(option numbers: include/za_amd.h ZG_OPT_*; 3 = PACK_THREADS, 5 = PACK_PIN).  Prints zg_prove medians per value."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth
opt, values = int(sys.argv[1]), [int(v) for v in sys.argv[2].split(",")]
lg = int(sys.argv[3]) if len(sys.argv) > 3 else 22
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
proofs = int(sys.argv[5]) if len(sys.argv) > 5 else 8
ctx = za_amd.Context(0)
for kv in os.environ.get("ZA_AB_PRESET", "").split(","):   # other options held fixed: ZA_AB_PRESET="3=4,5=0"
    if kv:
        ctx.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
R = sr.load(ctx)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
for _ in range(3):
    ctx.prove(P, R, pin, 5, 7)
host = {v: [] for v in values}
for _ in range(rounds):
    for v in values:
        ctx.set_option(opt, v)
        ctx.prove(P, R, pin, 5, 7); ctx.prove(P, R, pin, 5, 7)
        for _ in range(proofs):
            t0 = time.perf_counter(); ctx.prove(P, R, pin, 5, 7); host[v].append(1e3 * (time.perf_counter() - t0))
for v in values:
    print("option %d = %-4d zg_prove median %.3f min %.3f max %.3f" % (opt, v, np.median(host[v]), np.min(host[v]), np.max(host[v])))
