"""Who slows the NTT pipeline inside a proof?  Thread A loops the H pipeline (R1CS evaluation + 21 passes) on one
context; thread B (optional) loops the four witness multiexps of the same key on a second context.
usage: python tools/interference.py [none|witness|poll]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import za_amd
from za_amd import synth
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
ctxA, ctxB = za_amd.Context(0), za_amd.Context(0)
sr = synth.SynthR1CS(22, 2, 0.97, 0x5A)
RA, RB = sr.load(ctxA), sr.load(ctxB)
P = ctxB.setup(RB, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
dwA, dwB = ctxA.upload(sr.witness), ctxB.upload(sr.witness)
P.prepare(RB, dwB)
dhA = ctxA.alloc(32 << 22)
dhB = ctxB.alloc(32 << 22)
ctxB.prove_h_dev(RB, dwB, dhB)
stop = False
count = [0]
def worker():
    while not stop:
        ctxB.prove_partials_dev(P, RB, dwB, dhB, 0x1e)
        count[0] += 1
for _ in range(3):
    ctxA.prove_h_dev(RA, dwA, dhA)
th = None
if mode == "witness":
    th = threading.Thread(target=worker); th.start(); time.sleep(0.2)
t = time.perf_counter()
N = 40
for _ in range(N):
    ctxA.prove_h_dev(RA, dwA, dhA)
dt = 1e3 * (time.perf_counter() - t) / N
stop = True
if th: th.join()
print("mode %s: H pipeline %.3f ms per run; witness phases run meanwhile: %d" % (mode, dt, count[0]))
