"""za_amd/workloads.py generators that bench.py proves (CPU only): the SHA-256 Merkle circuit (BASELINE config 4 on a
real circuit) is pinned by hashlib -- its root is recomputed independently -- and its witness must satisfy every row."""
import hashlib

import numpy as np

from za_amd import workloads as W


def _rows_satisfied(n_inputs, n_aux, n_rows, rp, col, cf, wit):
    vals = [int.from_bytes(wit[32 * i:32 * i + 32], "little") for i in range(n_inputs + n_aux)]
    cfs = [[int.from_bytes(bytes(cf[k][32 * e:32 * e + 32]), "little") for e in range(len(col[k]))] for k in range(3)]

    def dot(k, r):
        return sum(cfs[k][e] * vals[col[k][e]] for e in range(int(rp[k][r]), int(rp[k][r + 1]))) % W.R
    return all(dot(0, r) * dot(1, r) % W.R == dot(2, r) for r in range(n_rows)), vals


def test_sha256_merkle_circuit_against_hashlib():
    ni, na, nc, rp, col, cf, wit, info = W.sha256_merkle(2, seed=5)
    assert ni == 3 and len(wit) == 32 * (ni + na)
    assert all(len(rp[k]) == nc + 1 and int(rp[k][-1]) == len(col[k]) and len(cf[k]) == 32 * len(col[k]) for k in range(3))
    # independent recomputation of the path
    cur = info["leaf"]
    for sib, d in zip(info["siblings"], info["directions"]):
        cur = hashlib.sha256((sib + cur) if d else (cur + sib)).digest()
    assert cur == info["root"]
    ok, vals = _rows_satisfied(ni, na, nc, rp, col, cf, wit)
    assert ok
    assert vals[0] == 1 and vals[1] == int.from_bytes(cur[:16], "big") and vals[2] == int.from_bytes(cur[16:], "big")
    # every variable is used, every aux value below 2^128 and 99 % of them bits (the statistic config 4 is about)
    used = set()
    for k in range(3):
        used |= set(np.asarray(col[k]).tolist())
    assert len(used) == ni + na
    bits = sum(1 for v in vals[ni:] if v <= 1)
    assert bits > 0.99 * na
    # two compressions per level: ~46 k rows each level
    assert 45000 < nc / 2 < 47500


def test_sha256_merkle_wrong_witness_breaks_a_row():
    ni, na, nc, rp, col, cf, wit, info = W.sha256_merkle(1, seed=9)
    bad = bytearray(wit)
    bad[32 * (ni + 5)] ^= 1          # flip one private leaf bit
    ok, _ = _rows_satisfied(ni, na, nc, rp, col, cf, bytes(bad))
    assert not ok


def test_eddsa_poseidon_verifier_circuit_takes_the_reference_signature():
    """BASELINE config 2's family: the statement of circomlib's EdDSAPoseidonVerifier with this repo's gadgets.  The
    witness is the signature the reference's own test uses (interop/circuits/circomlib/za_test/eddsaposeidon.za:7-12):
    it must satisfy every row, a changed message must not, and a plain Baby Jubjub computation agrees."""
    kat = W.EDDSA_KAT
    A, R8 = (kat["Ax"], kat["Ay"]), (kat["R8x"], kat["R8y"])
    h = W.poseidon_hash([R8[0], R8[1], A[0], A[1], kat["M"]])
    assert W.babyjub_mul(kat["S"], W._BJ_B8) == W.babyjub_add(R8, W.babyjub_mul(h, W.babyjub_mul(8, A)))
    ni, na, cons, w = W.eddsa_poseidon_verifier()
    assert ni == 4 and w[:4] == [1, kat["Ax"], kat["Ay"], kat["M"]] and 7000 < len(cons) < 8192

    def val(lc, ww):
        return sum(c * ww[v] for v, c in lc) % W.R
    assert all(val(a, w) * val(b, w) % W.R == val(c, w) for a, b, c in cons)
    used = {v for con in cons for lc in con for v, _ in lc}
    assert used == set(range(ni + na))
    for field, delta in (("M", 1), ("S", 1), ("R8x", 0)):
        sig = dict(kat)
        if delta:
            sig[field] += delta
        else:
            sig["R8x"], sig["R8y"] = W.babyjub_add(R8, W._BJ_B8)      # another point on the curve
        _, _, cons2, w2 = W.eddsa_poseidon_verifier(sig)
        assert not all(val(a, w2) * val(b, w2) % W.R == val(c, w2) for a, b, c in cons2), field
