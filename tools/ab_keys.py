"""A/B of a switch that is read when a key's tables are BUILT (ZA_AMD_QUAD_W, ZA_AMD_CW, ...): one key per value, built
under that value, then proofs alternated between the keys inside ONE process.
    python tools/ab_keys.py NAME v1,v2,... [log_m] [rounds] [proofs] [bit_fraction]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth
name, values = sys.argv[1], sys.argv[2].split(",")
lg = int(sys.argv[3]) if len(sys.argv) > 3 else 22
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
proofs = int(sys.argv[5]) if len(sys.argv) > 5 else 8
bits = float(sys.argv[6]) if len(sys.argv) > 6 else 0.97
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(lg, 2, bits, 0x5A if bits > 0 else 0x26)
R = sr.load(ctx)
pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
d_w = ctx.upload(sr.witness)
keys, ref = {}, None
for v in values:
    os.environ[name] = v
    keys[v] = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
    for _ in range(3):
        p = ctx.prove(keys[v], R, pin, 5, 7)
    assert ref is None or p == ref
    ref = p
    print(name, v, "tables", [keys[v].table_info(k) for k in range(5)], flush=True)
dev = {v: [] for v in values}
host = {v: [] for v in values}
for _ in range(rounds):
    for v in values:
        ctx.prove_dev(keys[v], R, d_w, 5, 7)
        for _ in range(proofs):
            t0 = time.perf_counter(); ctx.prove_dev(keys[v], R, d_w, 5, 7); dev[v].append(1e3 * (time.perf_counter() - t0))
        for _ in range(proofs):
            t0 = time.perf_counter(); ctx.prove(keys[v], R, pin, 5, 7); host[v].append(1e3 * (time.perf_counter() - t0))
for v in values:
    print("%s=%-6s prove_dev median %.3f min %.3f | zg_prove median %.3f min %.3f" %
          (name, v, np.median(dev[v]), np.min(dev[v]), np.median(host[v]), np.min(host[v])))
