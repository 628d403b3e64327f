"""Randomised parity sweep (GPU box, uses the C oracle): python tools/fuzz_prove.py [seconds [seed]].
Two families: (a) the synthetic config-4/5 generator with random size, bit share, inputs, table mask and h parts;
(b) circuits whose H polynomial is PRESCRIBED coefficient by coefficient (oracle/circuits.py::prescribed_h_r1cs) with
runs of 0, 1, r - 1 and random values -- the dense h multiexp never sees such scalars from family (a).  Family (b)
proves through zg_prove (host witness) with a key from the GPU set-up."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import za_amd
from za_amd import synth
from oracle import coracle as co
from util import download_params
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20261002)   # (a second argument: another sequence of cases)
ctx = za_amd.Context(0)
t0, n_ok = time.time(), 0
from oracle import bn254 as bn
from oracle.circuits import prescribed_h_r1cs
ctx.set_option(za_amd.OPT_C_FOLD_MIN_LOG_M, 8)
while time.time() - t0 < budget:
    c_fold = rng.choice([0, 1, 2, 2])      # four transforms with the C rows folded into the l multiexp: never / by the rows' values / always
    ctx.set_option(za_amd.OPT_C_FOLD, c_fold)
    # the hand-over of a witness that does not pack: in chunks with the class bytes made chunk by chunk (from 2^9 values here;
    # the l multiexp itself is only split for large jobs: tests/test_gpu_prove.py), or in one copy
    ctx.set_option(za_amd.OPT_UPLOAD_GATES, rng.choice([0, 9, 23]))
    if rng.random() < 0.3:
        lg = rng.choice([1, 2, 3, 5, 8, 10, 12, 13])
        m = 1 << lg
        fbw = rng.choice([0, 1, 31])
        ctx.set_option(za_amd.OPT_TABLES, fbw)
        h_parts = rng.choice([1, 2, 4])
        ctx.set_option(za_amd.OPT_H_PARTS, h_parts)
        p_special = rng.choice([0.2, 0.7, 1.0])
        pool = [0, 1, 1, 2, bn.R - 1, bn.R - 2, 1 << 253, (1 << 20) - 1, 1 << 20]
        h = [rng.choice(pool) if rng.random() < p_special else rng.randrange(bn.R) for _ in range(m - 1)]
        r1, w = prescribed_h_r1cs(lg, h)
        csr = co.CsrR1CS.from_r1cs(r1)
        wb = co.witness_bytes(w)
        R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
        P, _ = ctx.setup(R, *[rng.randrange(1, bn.R) for _ in range(5)], want_bytes=False)
        r, s = rng.randrange(bn.R), rng.randrange(bn.R)
        got = ctx.prove(P, R, wb, r, s)
        exp = co.prove(download_params(ctx, P), csr, wb, r, s, threads=16)
        ok = bytes(got) == bytes(exp)
        print("prescribed H lg %2d special %.1f FBW %-2d parts %s c_fold %d%s -> %s" % (lg, p_special, fbw, h_parts, c_fold, "*" if ctx.last_stats()["c_folded"] else " ", "ok" if ok else "MISMATCH"), flush=True)
        if not ok:
            sys.exit(1)
        n_ok += 1
        P.close(); R.close()
        continue
    lg = rng.choice([9, 10, 11, 12, 13, 14, 15, 16, 17])
    bf = rng.choice([0.0, 0.3, 0.5, 0.9, 0.97, 1.0])
    seed = rng.randrange(1 << 30)
    n_in = rng.choice([1, 2, 3, 7])
    fbw = rng.choice([0, 1, 31, 31, 17, 14])
    ctx.set_option(za_amd.OPT_TABLES, fbw)
    h_parts = rng.choice([1, 2, 3])
    ctx.set_option(za_amd.OPT_H_PARTS, h_parts)
    sr = synth.SynthR1CS(lg, n_in, bf, seed)
    R = sr.load(ctx)
    m = 1 << lg
    P = synth.synth_params(ctx, m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, seed ^ 0x55)
    r, s = rng.randrange(1, 1 << 250), rng.randrange(1, 1 << 250)
    d_w = ctx.upload(sr.witness)
    P.prepare(R, d_w)
    entry = rng.choice(["dev", "dev", "host", "packed"])
    if entry == "dev":
        got = ctx.prove_dev(P, R, d_w, r, s)
    elif entry == "host":
        got = ctx.prove(P, R, sr.witness, r, s)
    else:
        got = ctx.prove_packed(P, R, za_amd.pack_witness(sr.witness), r, s)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    exp = co.prove(download_params(ctx, P), csr, sr.witness, r, s, threads=32)
    kinds = [P.table_info(k)[0] for k in range(5)]
    ok = bytes(got) == bytes(exp)
    print("lg %2d bits %.2f inputs %d FBW %-2d parts %s c_fold %d%s tables %s entry %-6s -> %s" % (lg, bf, n_in, fbw, h_parts, c_fold, "*" if ctx.last_stats()["c_folded"] else " ", kinds, entry, "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        sys.exit(1)
    n_ok += 1
    if rng.random() < 0.35:
        # a second witness through the SAME key (its tables were built from the first one): some bits become field
        # elements and some field elements bits / zeros -- the compact window tables' fallback, subset-sum groups with mixed
        # contents, the packing pass's give-up path.  The system is no longer satisfied; the proof is still a well-defined
        # group element and the oracle computes the same one.
        w2 = np.array(sr.witness, copy=True).reshape(-1, 32)
        nv = w2.shape[0]
        if rng.random() < 0.25:      # skew: one constant in a large share of the positions (one bucket per digit takes them all)
            const = rng.choice([2, 3, 65537, (65537 << 40) + 3, bn.R - 1])
            share = rng.choice([0.25, 0.6, 1.0])
            for v in range(1, nv):
                if rng.random() < share:
                    w2[v, :] = np.frombuffer(const.to_bytes(32, "little"), dtype=np.uint8)
        for _ in range(rng.choice([1, 5, 40, 400])):
            v = rng.randrange(1, nv)
            kind = rng.random()
            if kind < 0.5:
                w2[v, :31] = np.frombuffer(rng.randbytes(31), dtype=np.uint8); w2[v, 31] = 0
            elif kind < 0.8:
                w2[v, :] = 0; w2[v, 0] = rng.choice([0, 1])
            else:
                w2[v, :] = 0
        w2 = w2.reshape(-1)
        entry2 = rng.choice(["dev", "host"])
        if entry2 == "dev":
            d_w2 = ctx.upload(w2)
            got2 = ctx.prove_dev(P, R, d_w2, r, s)
            ctx.free(d_w2)
        else:
            got2 = ctx.prove(P, R, w2, r, s)
        exp2 = co.prove(download_params(ctx, P), csr, w2, r, s, threads=32)
        ok2 = bytes(got2) == bytes(exp2)
        print("   second witness through the same key, entry %-4s -> %s" % (entry2, "ok" if ok2 else "MISMATCH"), flush=True)
        if not ok2:
            sys.exit(1)
        n_ok += 1
    ctx.free(d_w); P.close(); R.close(); sr.close()
print("fuzz: %d cases ok" % n_ok)
