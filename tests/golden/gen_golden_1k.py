#!/usr/bin/env python3
"""Golden vectors at SURVEY 8(c)'s stated sizes (a 2^10 domain, n = 2^10 multiexps), from the pure-Python big-int
oracle.  Run from the repo root:  python tests/golden/gen_golden_1k.py   (about ten minutes of pure-Python group
arithmetic).  Companion of gen_golden.py, whose fixtures stop at a 2^7 domain and n = 64.

  proof_chain10.json         (circuit recipe, witness, trapdoor, r, s) -> proof bytes, proof.json, digests of the
                             constraints, of a / b / c / h and of the key; the circuit itself is oracle.circuits.chain_r1cs
                             (10, n_inputs=3, seed=1010, bit_fraction=0.6), pinned here by the digest of its rows
  proof_chain10.params.bin   the bellman Parameters::write bytes of the key (binary, about 400 KB)
  msm_kat_1k.json            G1 / G2 multiexps over 2^10 bases k_i G (k_i = SplitMix64(seed, i), the same sequence
                             zg_synth_dlog uses); scalar vectors by recipe (SplitMix64-derived: uniform, bits, 16-bit,
                             and a mix with 0, 1, 2, r - 1, r - 2, 2^16, 2^128 - 1, 2^253 + 12345 -- every case bellman
                             special-cases), pinned by their digests, with and without density masks;
                             expected sums from bellman's windowed algorithm as restated in oracle.groth16
                             (checked against the naive sum and against (sum s_i k_i) G)
  ntt_kat_1k.json            the four transform flavours of a 2^10 vector (input by recipe, outputs as digests + end values)
Every proof is checked with the Groth16 pairing equation and the trapdoor identities before it is written.
"""
import hashlib
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
MASK64 = (1 << 64) - 1


def splitmix64_at(seed, i):
    z = (seed + (i + 1) * 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def constraints_digest(r1):
    h = hashlib.sha256()
    for cons in r1.constraints:
        for lc in cons:
            h.update(len(lc).to_bytes(4, "little"))
            for v, c in lc:
                h.update(int(v).to_bytes(4, "little") + int(c).to_bytes(32, "little"))
    return h.hexdigest()


def vec_digest(vals):
    return hashlib.sha256(b"".join(int(x).to_bytes(32, "little") for x in vals)).hexdigest()


def proof_1k():
    rng = random.Random(20261002)
    args = {"log_m": 10, "n_inputs": 3, "seed": 1010, "bit_fraction": 0.6}
    r1, w = chain_r1cs(args["log_m"], n_inputs=args["n_inputs"], seed=args["seed"], bit_fraction=args["bit_fraction"])
    trapdoor = tuple(rng.randrange(1, bn.R) for _ in range(5))
    r, s = rng.randrange(bn.R), rng.randrange(bn.R)
    params, trap = g.generate_parameters(r1, *trapdoor)
    proof = g.create_proof(r1, params, w, r, s)
    eA, eB, eC = g.trapdoor_expected_proof(r1, trap, w, r, s)
    assert proof == (bn.pt_mul(bn.G1_GEN, eA), bn.pt_mul(bn.G2_GEN, eB), bn.pt_mul(bn.G1_GEN, eC))
    pub = w[1:r1.num_inputs]
    assert g.verify_proof(params.vk, proof, pub)
    assert not g.verify_proof(params.vk, proof, [(pub[0] + 1) % bn.R] + pub[1:])
    a, b, c, ad, bi, bd = g.eval_constraints(r1, w)
    hco = g.h_coefficients(r1, w)
    pbytes = g.params_write(params)
    with open(os.path.join(OUT, "proof_chain10.params.bin"), "wb") as f:
        f.write(pbytes)
    rec = {"circuit": args, "n_inputs": r1.num_inputs, "n_aux": r1.num_aux, "n_constraints": len(r1.constraints),
           "constraints_sha256": constraints_digest(r1), "witness": ["%x" % x for x in w],
           "trapdoor": ["%x" % x for x in trapdoor], "r": "%x" % r, "s": "%x" % s,
           "params_sha256": hashlib.sha256(pbytes).hexdigest(), "params_len": len(pbytes),
           "proof": g.proof_bytes(proof).hex(), "proof_json": g.proof_json(proof, pub),
           "a_sha256": vec_digest(a), "b_sha256": vec_digest(b), "c_sha256": vec_digest(c), "h_sha256": vec_digest(hco),
           "h_first": ["%x" % x for x in hco[:4]], "h_last": ["%x" % x for x in hco[-4:]],
           "density_totals": [int(sum(ad)), int(sum(bi)), int(sum(bd))]}
    with open(os.path.join(OUT, "proof_chain10.json"), "w") as f:
        json.dump(rec, f, separators=(",", ":"))
    print("proof_chain10.json", os.path.getsize(os.path.join(OUT, "proof_chain10.json")), "bytes; params", len(pbytes))


SPECIAL = [0, 1, bn.R - 1, 2, 1 << 16, (1 << 253) + 12345, bn.R - 2, (1 << 128) - 1]


def kat_scalars(pattern, seed, n):
    """The scalar vectors of msm_kat_1k.json by recipe (tests/test_golden_1k.py rebuilds them the same way; the
    fixture pins them with a digest)."""
    def uniform(i):   # below 2^253 < r
        w = [splitmix64_at(seed, 4 * i + j) for j in range(4)]
        return w[0] | (w[1] << 64) | (w[2] << 128) | ((w[3] & ((1 << 61) - 1)) << 192)
    if pattern == "uniform":
        return [uniform(i) for i in range(n)]
    if pattern == "bits":
        return [splitmix64_at(seed, i) & 1 for i in range(n)]
    if pattern == "small_16bit":
        return [splitmix64_at(seed, i) & 0xffff for i in range(n)]
    if pattern == "mixed_special":
        return [SPECIAL[(i // 3) % len(SPECIAL)] if i % 3 == 0 else uniform(i) for i in range(n)]
    raise ValueError(pattern)


def kat_density(seed, n):
    return [int(splitmix64_at(seed, i) % 4 != 0) for i in range(n)]


def msm_1k():
    n = 1 << 10
    out = {"n": n, "special": ["%x" % x for x in SPECIAL], "cases": []}
    case_seed = 0x5000
    for group, gen, seed in (("g1", bn.G1_GEN, 0x1001), ("g2", bn.G2_GEN, 0x2002)):
        ks = [splitmix64_at(seed, i) for i in range(n)]
        pts = [bn.pt_mul(gen, k) for k in ks]
        names = ("uniform", "bits", "mixed_special", "small_16bit") if group == "g1" else ("uniform", "mixed_special")
        for name in names:
            for masked in (False, True):
                case_seed += 1
                sc = kat_scalars(name, case_seed, n)
                dens = kat_density(case_seed + 0x100, n) if masked else [1] * n
                dense_pts = [p for p, d in zip(pts, dens) if d]
                dense_ks = [k for k, d in zip(ks, dens) if d]
                exp = g.multiexp_bellman(dense_pts, [bool(d) for d in dens], sc)
                used = [s_ for s_, d in zip(sc, dens) if d]
                e = sum(s_ * k for s_, k in zip(used, dense_ks)) % bn.R
                assert exp == bn.pt_mul(gen, e), (group, name, masked)
                enc = bn.g1_to_mont_bytes if group == "g1" else bn.g2_to_mont_bytes
                out["cases"].append({"group": group, "bases_seed": seed, "pattern": name, "scalars_seed": case_seed,
                                     "density_seed": case_seed + 0x100 if masked else None,
                                     "scalars_sha256": hashlib.sha256(b"".join(bn.le32(x) for x in sc)).hexdigest(),
                                     "n_dense": len(dense_pts), "expect": enc(exp).hex()})
                print(group, name, "masked" if masked else "full", flush=True)
    with open(os.path.join(OUT, "msm_kat_1k.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("msm_kat_1k.json", os.path.getsize(os.path.join(OUT, "msm_kat_1k.json")), "bytes")


def ntt_1k():
    lg, n = 10, 1 << 10
    d = g.Domain(lg)
    a = kat_scalars("uniform", 0x7001, n)
    rec = {"log_m": lg, "input_seed": 0x7001, "input_sha256": vec_digest(a)}
    for name, fn in (("fft", d.fft), ("ifft", d.ifft), ("coset_fft", d.coset_fft), ("icoset_fft", d.icoset_fft)):
        e = list(a)
        fn(e)
        rec[name] = {"sha256": vec_digest(e), "first": ["%x" % x for x in e[:2]], "last": ["%x" % x for x in e[-2:]]}
    # anchor independent of the butterfly code: entries 0, 1 and 777 of the forward transform as naive dot products
    e = list(a)
    d.fft(e)
    for j in (0, 1, 777):
        assert e[j] == sum(a[i] * pow(d.omega, i * j, bn.R) for i in range(n)) % bn.R
    with open(os.path.join(OUT, "ntt_kat_1k.json"), "w") as f:
        json.dump(rec, f, separators=(",", ":"))
    print("ntt_kat_1k.json", os.path.getsize(os.path.join(OUT, "ntt_kat_1k.json")), "bytes")


if __name__ == "__main__":
    which = sys.argv[1:] or ["msm", "ntt", "proof"]
    if "ntt" in which:
        ntt_1k()
    if "msm" in which:
        msm_1k()
    if "proof" in which:
        proof_1k()
