"""Host packing pass vs NUMA placement: pins the process to one node while the witness is allocated and to another
while proving (argv: alloc_node run_node).  ZA_AMD_TRACE=1 prints the pass's stamps."""
import os, sys, time, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
def cpus(n):
    out = []
    for part in open(n + "/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out
print("nodes", [(os.path.basename(n), len(cpus(n))) for n in nodes])
alloc_node, run_node = int(sys.argv[1]), int(sys.argv[2])
os.sched_setaffinity(0, cpus(nodes[alloc_node]))
import numpy as np
import za_amd
from za_amd import synth
ctx = za_amd.Context(0)
sr = synth.SynthR1CS(22, 2, 0.97, 0x5A)
R = sr.load(ctx)
P = synth.synth_params(ctx, (1 << 22) - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 7)
pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
os.sched_setaffinity(0, cpus(nodes[run_node]))
ts = []
for i in range(12):
    t0 = time.perf_counter(); ctx.prove(P, R, pin, 5, 7); t = time.perf_counter() - t0
    ts.append("%.2f/%dMB" % (1e3 * t, ctx.last_timings()["upload_bytes"] / 1e6))
print("proofs", " ".join(ts))
print("alloc node", alloc_node, "run node", run_node, "last proof %.3f ms" % (1e3 * t), ctx.last_timings()["upload_bytes"])
