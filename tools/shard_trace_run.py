"""The multiexp phase one rank of a W-way sharded 2^22 proof runs (witness queries, then its h shard), a few times:
for kernel traces and ZA_AMD_TRACE host stamps.   python tools/shard_trace_run.py [W] [log_m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import za_amd
from za_amd import synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 22
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
R = sr.load(ctx)
full = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
d_w = ctx.upload(sr.witness)
Pk = synth.params_shard(ctx, full, 0, W)
Pk.prepare(R, d_w)
m = 1 << lg
d_h = ctx.alloc(32 * (m + W))
ctx.prove_h_dev(R, d_w, d_h)
for i in range(4):
    if i == 3:
        sys.stderr.write("=== traced\n"); sys.stderr.flush()
        os.environ["ZA_AMD_TRACE_NOW"] = "1"
    t0 = time.perf_counter()
    ctx.prove_partials_dev(Pk, R, d_w, d_h, 0x1e)
    t1 = time.perf_counter()
    ctx.prove_partials_dev(Pk, R, d_w, d_h, 0x01)
    t2 = time.perf_counter()
    sys.stderr.write("witness %.3f ms, h shard %.3f ms\n" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
