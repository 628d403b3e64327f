"""Recipes behind tests/golden/{msm_kat_1k, ntt_kat_1k, proof_chain10}.json -- the twins of the functions in
tests/golden/gen_golden_1k.py (the fixtures pin what they produce with digests)."""
import hashlib
import os

from oracle import bn254 as bn
from oracle import coracle as co

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MASK64 = (1 << 64) - 1
SPECIAL = [0, 1, bn.R - 1, 2, 1 << 16, (1 << 253) + 12345, bn.R - 2, (1 << 128) - 1]


def splitmix64_at(seed, i):
    z = (seed + (i + 1) * 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def kat_scalars(pattern, seed, n):
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


def kat_bases(group, seed, n):
    """n bases k_i * generator, k_i = SplitMix64(seed, i), as ABI bytes (affine, Montgomery LE)."""
    if group == "g1":
        gen, mul = bn.g1_to_mont_bytes(bn.G1_GEN), co.g1_mul
    else:
        gen, mul = bn.g2_to_mont_bytes(bn.G2_GEN), co.g2_mul
    return b"".join(bytes(mul(gen, splitmix64_at(seed, i))) for i in range(n))


def le_bytes(vals):
    return b"".join(int(x).to_bytes(32, "little") for x in vals)


def vec_digest(vals):
    return hashlib.sha256(le_bytes(vals)).hexdigest()


def constraints_digest(r1):
    h = hashlib.sha256()
    for cons in r1.constraints:
        for lc in cons:
            h.update(len(lc).to_bytes(4, "little"))
            for v, c in lc:
                h.update(int(v).to_bytes(4, "little") + int(c).to_bytes(32, "little"))
    return h.hexdigest()


def chain10_circuit(fx):
    """(R1CS, witness) of proof_chain10.json: the recipe's circuit, checked against the fixture's digest."""
    from oracle.circuits import chain_r1cs
    a = fx["circuit"]
    r1, w = chain_r1cs(a["log_m"], n_inputs=a["n_inputs"], seed=a["seed"], bit_fraction=a["bit_fraction"])
    assert constraints_digest(r1) == fx["constraints_sha256"], "oracle.circuits.chain_r1cs no longer gives the fixture's circuit"
    assert (r1.num_inputs, r1.num_aux, len(r1.constraints)) == (fx["n_inputs"], fx["n_aux"], fx["n_constraints"])
    return r1, w


def chain10_params():
    with open(os.path.join(GOLDEN, "proof_chain10.params.bin"), "rb") as f:
        return f.read()
