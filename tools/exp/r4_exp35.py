"""three provers (one context + host thread each) sharing the GPU and one resident key: ms per proof, per option setting
   usage: r4_exp35.py [log_m] ; options varied inside one process; ZA_AMD_JSLOTS from the environment"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import za_amd
from za_amd import synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
NP, PER = 3, 8
ctxs = [za_amd.Context(0) for _ in range(NP)]
sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
R = sr.load(ctxs[0])
P = ctxs[0].setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
pins = []
for c in ctxs:
    p = c.host_alloc(len(sr.witness)); p[:] = sr.witness; pins.append(p)
ref = ctxs[0].prove(P, R, pins[0], 5, 7)
for c, p in zip(ctxs, pins):
    assert c.prove(P, R, p, 5, 7) == ref
def run(label):
    def work(i):
        for _ in range(PER):
            assert ctxs[i].prove(P, R, pins[i], 5, 7) == ref
    th = [threading.Thread(target=work, args=(i,)) for i in range(NP)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    print("%-40s %.3f ms per proof" % (label, 1e3 * el / (NP * PER)), flush=True)
for rep in range(2):
    for raw, early in ((400, 1), (0, 1), (400, 0), (0, 0)):
        for c in ctxs:
            c.set_option(10, raw); c.set_option(11, early)
        run("raw cap %d, early combination %d" % (raw, early))
