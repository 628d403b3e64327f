"""One workload proved a few times through zg_prove (pinned host witness) -- for wall times, ZA_AMD_TRACE host stamps
(set the variable: the last proof is announced on stderr) and rocprofv3 kernel traces.
    python tools/proof_run.py [eddsa | poseidon:<log_m> | sha:<depth> | dense:<log_m> | <log_m>] [proofs]
eddsa = the EdDSA-Poseidon verifier circuit (config 2), poseidon:<n> = a Poseidon chain (config 3), sha:<d> = the bit-level
SHA-256 Merkle inclusion circuit of d levels (config 4 as named: 32; 90 fills the 2^22 domain), dense:<n> = uniform
field elements (config 5's shape), <n> = the config-4 shape (97 % bits) at 2^n."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth, workloads
what = sys.argv[1] if len(sys.argv) > 1 else "eddsa"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = za_amd.Context(0)
if what.startswith("sha:"):
    n_in, n_aux, n_cons, rp, cl, cf, wit, _ = workloads.sha256_merkle(int(what.split(":")[1]))
    R = ctx.r1cs_load(n_in, n_aux, n_cons, rp, cl, cf)
    wb = np.frombuffer(wit, dtype=np.uint8)
elif what == "eddsa" or what.startswith("poseidon:"):
    if what == "eddsa":
        n_in, n_aux, cons, wit = workloads.eddsa_poseidon_verifier()
    else:
        n_in, n_aux, cons, wit, _ = workloads.poseidon_chain(int(what.split(":")[1]))
    rp, cl, cf = workloads.csr_arrays(cons)
    R = ctx.r1cs_load(n_in, n_aux, len(cons), rp, cl, cf)
    wb = np.frombuffer(workloads.witness_bytes(wit), dtype=np.uint8)
else:
    dense = what.startswith("dense:")
    sr = synth.SynthR1CS(int(what.split(":")[-1]), 2, 0.0 if dense else 0.97, 0x5A)
    R = sr.load(ctx)
    wb = np.array(sr.witness, copy=True)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
pin = ctx.host_alloc(len(wb)); pin[:] = wb
print("log_m", R.log_m, "tables", [P.table_info(k) for k in range(5)])
ts, proofs = [], set()
for i in range(n):
    if i == n - 1:
        sys.stderr.write("=== traced proof\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    proofs.add(bytes(ctx.prove(P, R, pin, 5, 7)))
    ts.append(1e3 * (time.perf_counter() - t0))
try:
    stats = ctx.last_stats()
except AttributeError:      # (ZA_AMD_LIB pointing at an older build of the library)
    stats = {}
print("ms per proof:", np.round(ts, 3), "median %.3f" % np.median(ts[3:]), ctx.last_timings(), stats, "distinct proofs:", len(proofs))
