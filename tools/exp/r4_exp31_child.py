"""one process = one stream assignment (ZA_AMD_JSLOTS): median / min of zg_prove_dev for the EdDSA circuit and a 2^14 proof"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import za_amd
from za_amd import synth, workloads
ctx = za_amd.Context(0)
out = []
for what in ("eddsa", "14"):
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
    d_w = ctx.upload(wb)
    for _ in range(5):
        ctx.prove_dev(P, R, d_w, 5, 7)
    ts = []
    for _ in range(60):
        t0 = time.perf_counter(); ctx.prove_dev(P, R, d_w, 5, 7); ts.append(1e3 * (time.perf_counter() - t0))
    out.append("%s median %.3f min %.3f" % (what, np.median(ts), np.min(ts)))
print(os.environ.get("ZA_AMD_JSLOTS", "default"), " | ".join(out))
