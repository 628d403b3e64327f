#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the pure-Python big-int oracle.

Run from the repo root:  python tests/golden/gen_golden.py
The reference (adria0/za) holds no golden vectors for the prover path (its tests use thread_rng and
only self-verify, prover/src/groth16/prover.rs:225-373) and cannot be built here (Rust, un-vendored
git dependencies), so these vectors come from oracle/*.py, whose proofs are checked with the Groth16
pairing equation -- the reference's own acceptance criterion -- before being written out.
Public anchor inside the vectors: 2*G1 equals the EIP-196 / alt_bn128 test value.
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bn254 as bn  # noqa: E402
from oracle import groth16 as g  # noqa: E402
from oracle.circuits import chain_r1cs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
rng = random.Random(20261001)


def hx(b):
    return bytes(b).hex()


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(name, os.path.getsize(os.path.join(OUT, name)), "bytes")


# ---- field vectors (canonical integers as hex)
def field_vectors():
    out = {}
    for name, mod in (("fq", bn.Q), ("fr", bn.R)):
        vec = []
        edge = [0, 1, 2, mod - 1, mod - 2, (1 << 253), (1 << 32) - 1, (1 << 64) - 1, (mod + 1) // 2]
        pairs = [(a, b) for a in edge for b in edge[:5]] + [(rng.randrange(mod), rng.randrange(mod)) for _ in range(64)]
        for a, b in pairs:
            vec.append(["%x" % a, "%x" % b, "%x" % (a * b % mod), "%x" % ((a + b) % mod), "%x" % ((a - b) % mod),
                        "%x" % (bn.inv(a, mod) if a else 0)])
        out[name] = vec
    out["fr_root_of_unity"] = "%x" % bn.FR_ROOT_OF_UNITY
    dump("field_vectors.json", out)


def g1_hex(P):
    return hx(bn.g1_to_mont_bytes(P))


def g2_hex(P):
    return hx(bn.g2_to_mont_bytes(P))


def curve_vectors():
    out = {"g1_mul": [], "g2_mul": [], "g1_add": [], "g2_add": []}
    ks = [1, 2, 3, 7, bn.R - 1, bn.R - 2, (1 << 128) + 5] + [rng.randrange(bn.R) for _ in range(6)]
    for k in ks:
        out["g1_mul"].append(["%x" % k, g1_hex(bn.pt_mul(bn.G1_GEN, k))])
        out["g2_mul"].append(["%x" % k, g2_hex(bn.pt_mul(bn.G2_GEN, k))])
    assert bn.pt_mul(bn.G1_GEN, 2)[0].v == 1368015179489954701390400359078579693043519447331113978918064868415326638035
    for _ in range(6):
        a, b = rng.randrange(1, bn.R), rng.randrange(1, bn.R)
        P, S = bn.pt_mul(bn.G1_GEN, a), bn.pt_mul(bn.G1_GEN, b)
        out["g1_add"].append([g1_hex(P), g1_hex(S), g1_hex(bn.pt_add(P, S))])
        P2, S2 = bn.pt_mul(bn.G2_GEN, a), bn.pt_mul(bn.G2_GEN, b)
        out["g2_add"].append([g2_hex(P2), g2_hex(S2), g2_hex(bn.pt_add(P2, S2))])
    P = bn.pt_mul(bn.G1_GEN, 5)
    out["g1_add"].append([g1_hex(P), g1_hex(P), g1_hex(bn.pt_double(P))])          # doubling
    out["g1_add"].append([g1_hex(P), g1_hex(bn.pt_neg(P)), g1_hex(None)])           # inverse -> identity
    out["g1_add"].append([g1_hex(None), g1_hex(P), g1_hex(P)])                      # identity
    dump("curve_vectors.json", out)


def msm_kats():
    out = {"g1": [], "g2": []}
    special = [0, 1, bn.R - 1, 2, (1 << 16), (1 << 253) + 12345]
    for n in (1, 2, 31, 32, 33, 64):
        pts = [bn.pt_mul(bn.G1_GEN, rng.randrange(1, bn.R)) for _ in range(n)]
        if n >= 31:
            pts[5] = pts[4]                 # duplicate base (forces the doubling branch in a bucket)
            pts[7] = bn.pt_neg(pts[6])      # opposite bases
        sc = [special[i % len(special)] if i % 3 == 0 else rng.randrange(bn.R) for i in range(n)]
        if n >= 31:
            sc[5] = sc[4]
            sc[7] = sc[6]
        dens = [1] * n if n < 31 else [rng.randrange(4) != 0 for _ in range(n)]
        dens = [int(d) for d in dens]
        dense_pts = [p for p, d in zip(pts, dens) if d]
        exp = g.msm_naive_density(dense_pts, dens, sc)
        assert exp == g.multiexp_bellman(dense_pts, [bool(d) for d in dens], sc)
        out["g1"].append({"n": n, "bases": hx(b"".join(bn.g1_to_mont_bytes(p) for p in dense_pts)),
                          "scalars": hx(b"".join(bn.le32(s) for s in sc)), "density": dens, "expect": g1_hex(exp)})
    for n in (1, 3, 33):
        pts = [bn.pt_mul(bn.G2_GEN, rng.randrange(1, bn.R)) for _ in range(n)]
        sc = [special[i % len(special)] if i % 3 == 0 else rng.randrange(bn.R) for i in range(n)]
        dens = [1] * n if n < 33 else [int(rng.randrange(4) != 0) for _ in range(n)]
        dense_pts = [p for p, d in zip(pts, dens) if d]
        exp = g.msm_naive_density(dense_pts, dens, sc)
        out["g2"].append({"n": n, "bases": hx(b"".join(bn.g2_to_mont_bytes(p) for p in dense_pts)),
                          "scalars": hx(b"".join(bn.le32(s) for s in sc)), "density": dens, "expect": g2_hex(exp)})
    # all-zero scalars -> identity
    pts = [bn.pt_mul(bn.G1_GEN, 9), bn.pt_mul(bn.G1_GEN, 10)]
    out["g1"].append({"n": 2, "bases": hx(b"".join(bn.g1_to_mont_bytes(p) for p in pts)),
                      "scalars": hx(bn.le32(0) * 2), "density": [1, 1], "expect": g1_hex(None)})
    dump("msm_kat.json", out)


def ntt_kats():
    out = []
    for lg in (0, 1, 2, 4, 6):
        m = 1 << lg
        d = g.Domain(lg)
        a = [rng.randrange(bn.R) for _ in range(m)]
        rec = {"log_m": lg, "input": ["%x" % x for x in a]}
        for name, fn in (("fft", d.fft), ("ifft", d.ifft), ("coset_fft", d.coset_fft), ("icoset_fft", d.icoset_fft)):
            e = list(a)
            fn(e)
            rec[name] = ["%x" % x for x in e]
        if lg <= 4:
            assert [int(x, 16) for x in rec["fft"]] == g.dft_naive(a, d.omega)
        out.append(rec)
    dump("ntt_kat.json", out)


def w32(x):
    """FS::write_256_w32 (compiler/src/algebra/fs.rs:116-126): 32 big-endian bytes, 4-byte groups last to first."""
    b = int(x).to_bytes(32, "big")
    return b"".join(b[n * 4:(n + 1) * 4] for n in reversed(range(8)))


def binwitness(values):
    """compiler/src/tester/embeeded.rs:47-57: count, then the value of every signal (id 0 = ONE)."""
    return w32(len(values)) + b"".join(w32(v) for v in values)


def proof_fixture(name, r1, w, trapdoor, r, s, qeqs=None, ignore_at=None):
    alpha, beta, gamma, delta, tau = trapdoor
    params, trap = g.generate_parameters(r1, alpha, beta, gamma, delta, tau)
    proof = g.create_proof(r1, params, w, r, s)
    assert proof == g.create_proof(r1, params, w, r, s, msm=g.msm_naive_density)
    eA, eB, eC = g.trapdoor_expected_proof(r1, trap, w, r, s)
    assert proof == (bn.pt_mul(bn.G1_GEN, eA), bn.pt_mul(bn.G2_GEN, eB), bn.pt_mul(bn.G1_GEN, eC))
    pub = w[1: r1.num_inputs]
    assert g.verify_proof(params.vk, proof, pub), "oracle proof does not verify"
    bad = list(pub)
    if bad:
        bad[0] = (bad[0] + 1) % bn.R
        assert not g.verify_proof(params.vk, proof, bad)
    hco = g.h_coefficients(r1, w)
    a, b, c, ad, bi, bd = g.eval_constraints(r1, w)
    pbytes = g.params_write(params)
    rec = {
        "n_inputs": r1.num_inputs, "n_aux": r1.num_aux,
        "constraints": [[[[v, "%x" % co] for v, co in lc] for lc in cons] for cons in r1.constraints],
        "witness": ["%x" % x for x in w], "trapdoor": ["%x" % x for x in trapdoor], "r": "%x" % r, "s": "%x" % s,
        "params": hx(pbytes), "proof": hx(g.proof_bytes(proof)), "proof_json": g.proof_json(proof, pub),
        "h": ["%x" % x for x in hco], "a": ["%x" % x for x in a], "b": ["%x" % x for x in b], "c": ["%x" % x for x in c],
        "a_aux_density": [int(x) for x in ad], "b_input_density": [int(x) for x in bi],
        "b_aux_density": [int(x) for x in bd],
    }
    if qeqs is not None:
        rec["pk_container"] = hx(g.pk_container_write(qeqs, [], pbytes))
        rec["binwitness"] = hx(binwitness(w))
    if ignore_at is not None:
        # the same circuit as za would store it when the optimizer removed signal `ignore_at`:
        # signal id = variable index, shifted by one past the ignored id; QEQ c = -C
        def sig(v):
            return v if v < ignore_at else v + 1
        qs = [([(sig(v), c) for v, c in A], [(sig(v), c) for v, c in B], [(sig(v), (bn.R - c) % bn.R) for v, c in C])
              for A, B, C in r1.constraints]
        rec["pk_container"] = hx(g.pk_container_write(qs, [ignore_at], pbytes))
        vals = list(w[:ignore_at]) + [0xDEAD] + list(w[ignore_at:])
        rec["binwitness"] = hx(binwitness(vals))
        rec["ignore_signals"] = [ignore_at]
    dump(name, rec)


def main():
    field_vectors()
    curve_vectors()
    msm_kats()
    ntt_kats()
    # config 1: example/circuit.za with example/input.json {p:2,q:3,r:6}; za QEQ p*q + (-r) = 0 over
    # signal ids one=0, r=1, p=2, q=3
    r1 = g.example_factor_r1cs()
    qeq = [([(2, 1)], [(3, 1)], [(1, bn.R - 1)])]
    proof_fixture("proof_factor.json", r1, [1, 6, 2, 3], (11, 12, 13, 14, 15), 5, 7, qeqs=qeq)
    # a 2^7-domain circuit with mixed bit / field rows, 3 public inputs
    r2, w2 = chain_r1cs(7, n_inputs=3, seed=77, bit_fraction=0.6)
    td = tuple(rng.randrange(1, bn.R) for _ in range(5))
    proof_fixture("proof_chain7.json", r2, w2, td, rng.randrange(bn.R), rng.randrange(bn.R), ignore_at=r2.num_inputs + 5)


if __name__ == "__main__":
    main()
