"""R1CS evaluation + H pipeline alone on the SHA-256 Merkle circuit: python tools/h_bench_sha.py [depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import workloads
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 90
ctx = za_amd.Context(0)
ni, na, nc, rp, col, cf, wit, info = workloads.sha256_merkle(depth)
R = ctx.r1cs_load(ni, na, nc, rp, col, cf)
d_w = ctx.upload(np.frombuffer(wit, dtype=np.uint8))
d_h = ctx.alloc(32 << R.log_m)
lens = [np.diff(rp[k].astype(np.int64)) for k in range(3)]
print("rows", nc, "log_m", R.log_m, "nnz", info["nnz"], "longest rows", [int(l.max()) for l in lens],
      "rows > 32 terms", [int((l > 32).sum()) for l in lens])
ctx.profile_enable(2)
for _ in range(3):
    ctx.prove_h_dev(R, d_w, d_h)
ctx.profile_reset()
t = time.perf_counter()
for _ in range(10):
    ctx.prove_h_dev(R, d_w, d_h)
print("H pipeline: %.3f ms" % (1e3 * (time.perf_counter() - t) / 10))
p = ctx.profile_get()
print({k: (round(v[0] / 10, 3), v[1] // 10) for k, v in p.items() if v[1]})
