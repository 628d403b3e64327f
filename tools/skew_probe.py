"""A witness in which a quarter of the aux values hold the SAME small non-bit value: every such scalar is one digit in one bucket."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (tools/ -> the repo root)
import numpy as np
import za_amd
from za_amd import synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
only = sys.argv[2] if len(sys.argv) > 2 else ""      # "q" = the quarter-equal case alone, "all" = the all-equal case alone (profiling)
ctx = za_amd.Context(0)
ctx.set_option(za_amd.OPT_C_FOLD, 0)
sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
R = sr.load(ctx)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
w = np.array(sr.witness, copy=True).reshape(-1, 32)
pin = ctx.host_alloc(w.size)
def run(tag, wit):
    pin[:] = wit.reshape(-1)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); p = ctx.prove(P, R, pin, 5, 7); ts.append(1e3 * (time.perf_counter() - t0))
    print(tag, "median %.3f ms" % np.median(ts[2:]), "first %.1f" % ts[0], flush=True)
if not only:
    run("plain      ", w)
rng = np.random.default_rng(7)
for val in ((2, 3, 65537) if not only else (65537,) if only == "q" else ()):
    w2 = w.copy()
    idx = rng.choice(np.arange(3, w.shape[0]), size=w.shape[0] // 4, replace=False)
    w2[idx, :] = 0
    w2[idx, 0] = val & 0xff; w2[idx, 1] = (val >> 8) & 0xff; w2[idx, 2] = (val >> 16) & 0xff
    run("25%% = %-6d" % val, w2)
if not only:
    run("plain again", w)
# every aux value the same non-bit constant (never run): one bucket per digit position holds every point
w3 = w.copy()
w3[3:, :] = 0
w3[3:, 0] = 0x01; w3[3:, 2] = 0x01          # 65537
if only != "q":
    run("all = 65537 ", w3)
if not only:
    run("plain again", w)
