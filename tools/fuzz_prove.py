"""Randomised parity sweep (GPU box, uses the C oracle): python tools/fuzz_prove.py [seconds]."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import za_amd
from za_amd import synth
from oracle import coracle as co
from util import download_params
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = random.Random(20261002)
ctx = za_amd.Context(0)
t0, n_ok = time.time(), 0
while time.time() - t0 < budget:
    lg = rng.choice([9, 10, 11, 12, 13, 14, 15, 16, 17])
    bf = rng.choice([0.0, 0.3, 0.5, 0.9, 0.97, 1.0])
    seed = rng.randrange(1 << 30)
    n_in = rng.choice([1, 2, 3, 7])
    os.environ["ZA_AMD_FBW"] = rng.choice(["0", "1", "31", "31", "17", "14"])
    os.environ["ZA_AMD_H_PARTS"] = rng.choice(["1", "2", "3"])
    sr = synth.SynthR1CS(lg, n_in, bf, seed)
    R = sr.load(ctx)
    m = 1 << lg
    P = synth.synth_params(ctx, m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, seed ^ 0x55)
    r, s = rng.randrange(1, 1 << 250), rng.randrange(1, 1 << 250)
    d_w = ctx.upload(sr.witness)
    P.prepare(R, d_w)
    got = ctx.prove_dev(P, R, d_w, r, s)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    exp = co.prove(download_params(ctx, P), csr, sr.witness, r, s, threads=32)
    kinds = [P.table_info(k)[0] for k in range(5)]
    ok = bytes(got) == bytes(exp)
    print("lg %2d bits %.2f inputs %d FBW %-2s parts %s tables %s -> %s" % (lg, bf, n_in, os.environ["ZA_AMD_FBW"], os.environ["ZA_AMD_H_PARTS"], kinds, "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        sys.exit(1)
    n_ok += 1
    ctx.free(d_w); P.close(); R.close(); sr.close()
print("fuzz: %d cases ok" % n_ok)
