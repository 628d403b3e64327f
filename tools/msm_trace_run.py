"""One standalone 2^n G1 multiexp over resident bases (window table, dense scalars), a few times: for kernel traces.
    python tools/msm_trace_run.py [log_n] [host]     host: the scalars come from pinned host memory inside every call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import za_amd
from za_amd import synth
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
host = len(sys.argv) > 2 and sys.argv[2] == "host"
ctx = za_amd.Context(0)
d_b = synth.synth_bases_g1(ctx, n, 4242)
d_s = synth.synth_scalars(ctx, n, 99)
H = ctx.msm_bases_g1_dev(d_b, n, table_budget_gb=120.0)
if host:
    pin = ctx.host_alloc(32 * n)
    pin[:] = ctx.d2h(d_s, 32 * n)
for i in range(4):
    t0 = time.perf_counter()
    if host:
        ctx.msm_resident(H, pin, n, True)
    else:
        ctx.msm_resident_dev(H, d_s, n, True)
    print("msm %d: %.3f ms" % (i, 1e3 * (time.perf_counter() - t0)))
