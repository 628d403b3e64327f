"""The C oracle (bellman-algorithm restatement) against the committed golden vectors."""
import numpy as np
import pytest

from oracle import bn254 as bn
from oracle import coracle as co
from tests.util import canon_bytes, from_canon_bytes, from_mont_bytes, golden, mont_bytes, r1cs_from_fixture


@pytest.fixture(scope="module", autouse=True)
def _built():
    co.build()


def test_field_vectors():
    fv = golden("field_vectors.json")
    for which, (name, mod) in enumerate((("fq", bn.Q), ("fr", bn.R))):
        a = [int(v[0], 16) for v in fv[name]]
        b = [int(v[1], 16) for v in fv[name]]
        got = from_mont_bytes(co.field_mul(which, mont_bytes(a, mod), mont_bytes(b, mod)), mod)
        assert got == [int(v[2], 16) for v in fv[name]]
        assert bytes(co.to_mont(which, canon_bytes(a))) == mont_bytes(a, mod)
        assert from_canon_bytes(co.from_mont(which, mont_bytes(a, mod))) == a


def test_curve_vectors():
    cv = golden("curve_vectors.json")
    g1, g2 = bn.g1_to_mont_bytes(bn.G1_GEN), bn.g2_to_mont_bytes(bn.G2_GEN)
    for k, exp in cv["g1_mul"]:
        assert co.g1_mul(g1, int(k, 16)).hex() == exp
    for k, exp in cv["g2_mul"]:
        assert co.g2_mul(g2, int(k, 16)).hex() == exp
    for p, q, exp in cv["g1_add"]:
        assert co.g1_add(bytes.fromhex(p), bytes.fromhex(q)).hex() == exp
    for p, q, exp in cv["g2_add"]:
        assert co.g2_add(bytes.fromhex(p), bytes.fromhex(q)).hex() == exp


def test_msm_kats():
    kat = golden("msm_kat.json")
    for rec in kat["g1"]:
        for th in (1, 4):
            got = co.msm_g1(bytes.fromhex(rec["bases"]), bytes.fromhex(rec["scalars"]),
                            np.array(rec["density"], dtype=np.uint8), threads=th)
            assert got.hex() == rec["expect"], rec["n"]
    for rec in kat["g2"]:
        got = co.msm_g2(bytes.fromhex(rec["bases"]), bytes.fromhex(rec["scalars"]),
                        np.array(rec["density"], dtype=np.uint8), threads=2)
        assert got.hex() == rec["expect"], rec["n"]


def test_msm_density_mismatch_is_rejected():
    rec = golden("msm_kat.json")["g1"][2]
    with pytest.raises(RuntimeError):
        co.msm_g1(bytes.fromhex(rec["bases"])[:-64], bytes.fromhex(rec["scalars"]),
                  np.array(rec["density"], dtype=np.uint8))


def test_ntt_kats():
    for rec in golden("ntt_kat.json"):
        lg = rec["log_m"]
        a = mont_bytes([int(x, 16) for x in rec["input"]], bn.R)
        for mode, name in enumerate(("fft", "ifft", "coset_fft", "icoset_fft")):
            for th in (1, 8):
                got = from_mont_bytes(co.fft(a, lg, mode, threads=th), bn.R)
                assert got == [int(x, 16) for x in rec[name]], (lg, name, th)


@pytest.mark.parametrize("name", ["proof_factor.json", "proof_chain7.json"])
def test_prove_matches_golden(name):
    fx = golden(name)
    from oracle import groth16 as g
    r1 = r1cs_from_fixture(fx)
    csr = co.CsrR1CS.from_r1cs(r1)
    w = [int(x, 16) for x in fx["witness"]]
    wb = co.witness_bytes(w)
    a, b, c, ad, bi, bd = co.r1cs_eval(csr, wb, threads=2)
    assert from_mont_bytes(a, bn.R) == [int(x, 16) for x in fx["a"]]
    assert from_mont_bytes(b, bn.R) == [int(x, 16) for x in fx["b"]]
    assert from_mont_bytes(c, bn.R) == [int(x, 16) for x in fx["c"]]
    assert list(ad) == fx["a_aux_density"] and list(bi) == fx["b_input_density"] and list(bd) == fx["b_aux_density"]
    rows = len(fx["a"])
    h = co.h_coeffs(a, b, c, rows, r1.domain_log(), threads=4)
    assert from_canon_bytes(h) == [int(x, 16) for x in fx["h"]]
    params, _ = g.params_read(bytes.fromhex(fx["params"]))
    mp = co.MontParams.from_params(params)
    for th in (1, 8):
        proof = co.prove(mp, csr, wb, int(fx["r"], 16), int(fx["s"], 16), threads=th)
        assert proof.hex() == fx["proof"]
