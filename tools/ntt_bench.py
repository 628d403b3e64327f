"""Exclusive timing of the NTT passes: python tools/ntt_bench.py [log_m] (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, za_amd
from za_amd import synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = za_amd.Context(0)
d = synth.synth_scalars(ctx, 1 << lg, 5)
for _ in range(3):
    ctx.ntt_dev(d, lg, False, False)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    ctx.ntt_dev(d, lg, False, False)
torch.cuda.synchronize()
print("ms per zg_ntt_fr_dev (3 passes + bitrev + 2 conversions):", 1e3 * (time.perf_counter() - t) / 20)
ctx.free(d); ctx.close()
