"""What the per-proof guard of the folded C rows costs and buys (prover.hip k_fold_decide; VERDICT r5 missing #4):

    python tools/fold_guard_probe.py [log_m]

One key of the config-4 recipe, folded by its first (bit) witness.  (a) the bit witness: ZG_OPT_C_FOLD = 1 (guarded: the
evaluator also writes the rows for the transforms and counts their digits, c's passes are queued and leave at once) against
2 (folded whatever the rows hold: round 5's schedule) against 0 (six transforms), values alternated in one process;
(b) a witness of field elements through the SAME key: 1 (the guard takes six transforms for it) against 2 (its m dense rows
go through the l multiexp: round 5) against 0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = za_amd.Context(0)
ctx.set_option(za_amd.OPT_C_FOLD_MIN_LOG_M, 10)
sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
R = sr.load(ctx)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
nv = sr.n_inputs + sr.n_aux
rng = np.random.default_rng(5)
dense = rng.integers(0, 256, size=32 * nv, dtype=np.uint8)
dense[31::32] = 0
dense[:32] = 0
dense[0] = 1
pin = {"bits": ctx.host_alloc(32 * nv), "dense": ctx.host_alloc(32 * nv)}
pin["bits"][:] = sr.witness
pin["dense"][:] = dense
for _ in range(3):
    ctx.prove(P, R, pin["bits"], 5, 7)            # the key folds on its first witness
print("2^%d, key tables %s" % (lg, [P.table_info(k) for k in range(6)]), flush=True)
for name in ("bits", "dense"):
    ref = None
    res = {v: [] for v in (1, 2, 0)}
    folded = {}
    for rnd in range(6):
        for v in (1, 2, 0):
            ctx.set_option(za_amd.OPT_C_FOLD, v)
            p = ctx.prove(P, R, pin[name], 5, 7)
            assert ref is None or p == ref, "proof bytes differ between the modes"
            ref = p
            for _ in range(6):
                t0 = time.perf_counter(); ctx.prove(P, R, pin[name], 5, 7); res[v].append(1e3 * (time.perf_counter() - t0))
            folded[v] = ctx.last_stats()["c_folded"]
    for v in (1, 2, 0):
        print("%-5s witness, ZG_OPT_C_FOLD = %d: zg_prove median %.3f ms, min %.3f (folded: %s)" %
              (name, v, np.median(res[v]), np.min(res[v]), folded[v]), flush=True)
