"""A small proof a few times, the last ones with ZA_AMD_TRACE host stamps (set the variable): python tools/small_trace_run.py
[eddsa | <log_m>] -- the EdDSA-Poseidon verifier circuit (config 2) or the config-4 shape at 2^log_m."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth, workloads
what = sys.argv[1] if len(sys.argv) > 1 else "eddsa"
ctx = za_amd.Context(0)
if what == "eddsa":
    n_in, n_aux, cons, wit = workloads.eddsa_poseidon_verifier()
    rp, cl, cf = workloads.csr_arrays(cons)
    R = ctx.r1cs_load(n_in, n_aux, len(cons), rp, cl, cf)
    wb = np.frombuffer(workloads.witness_bytes(wit), dtype=np.uint8)
else:
    sr = synth.SynthR1CS(int(what), 2, 0.97, 0x5A)
    R = sr.load(ctx)
    wb = np.array(sr.witness, copy=True)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
pin = ctx.host_alloc(len(wb)); pin[:] = wb
print("log_m", R.log_m, "tables", [P.table_info(k) for k in range(5)])
n = 12
ts = []
for i in range(n):
    if i == n - 1:
        sys.stderr.write("=== traced proof\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    ctx.prove(P, R, pin, 5, 7)
    ts.append(1e3 * (time.perf_counter() - t0))
print("ms per proof:", np.round(ts, 3), "median %.3f" % np.median(ts[3:]), ctx.last_timings())
