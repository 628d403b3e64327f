"""Poseidon-chain 2^n proof (config 3) a few times, the last one with ZA_AMD_TRACE stamps if the variable is set."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import za_amd
from za_amd import workloads
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = za_amd.Context(0)
n_in, n_aux, cons, wit, hashes = workloads.poseidon_chain(lg)
rp, cl, cf = workloads.csr_arrays(cons)
R = ctx.r1cs_load(n_in, n_aux, len(cons), rp, cl, cf)
P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
wb = workloads.witness_bytes(wit)
pin = ctx.host_alloc(len(wb)); pin[:] = __import__("numpy").frombuffer(wb, dtype="uint8")
print("tables", [P.table_info(k) for k in range(5)])
for i in range(5):
    if i == 4:
        sys.stderr.write("=== traced\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    ctx.prove(P, R, pin, 5, 7)
    print("proof %d: %.3f ms" % (i, 1e3 * (time.perf_counter() - t0)), ctx.last_timings())
