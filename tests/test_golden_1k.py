"""SURVEY 8(c)'s sizes: the C oracle against the Python-oracle fixtures at a 2^10 domain and n = 2^10 multiexps
(tests/golden/gen_golden_1k.py).  CPU only; the GPU path meets the same fixtures in tests/test_gpu_golden_1k.py."""
import hashlib

import numpy as np
import pytest

from oracle import bn254 as bn
from oracle import coracle as co
from oracle import groth16 as g
from tests import golden_1k as k1
from tests.util import from_canon_bytes, from_mont_bytes, golden, mont_bytes


@pytest.fixture(scope="module", autouse=True)
def _built():
    co.build()


def test_recipes_reproduce_the_pinned_inputs():
    kat = golden("msm_kat_1k.json")
    assert [int(x, 16) for x in kat["special"]] == k1.SPECIAL
    for case in kat["cases"]:
        sc = k1.kat_scalars(case["pattern"], case["scalars_seed"], kat["n"])
        assert hashlib.sha256(k1.le_bytes(sc)).hexdigest() == case["scalars_sha256"]
    fx = golden("proof_chain10.json")
    r1, w = k1.chain10_circuit(fx)
    assert ["%x" % x for x in w] == fx["witness"] and r1.is_satisfied(w)


def test_msm_kats_at_1k():
    kat = golden("msm_kat_1k.json")
    n = kat["n"]
    bases = {}
    for case in kat["cases"]:
        grp = case["group"]
        if grp not in bases:
            bases[grp] = k1.kat_bases(grp, case["bases_seed"], n)
        sc = k1.le_bytes(k1.kat_scalars(case["pattern"], case["scalars_seed"], n))
        dens = np.array(k1.kat_density(case["density_seed"], n) if case["density_seed"] else [1] * n, dtype=np.uint8)
        elem = 64 if grp == "g1" else 128
        dense = b"".join(bases[grp][elem * i:elem * (i + 1)] for i in range(n) if dens[i])
        assert len(dense) == elem * case["n_dense"]
        fn = co.msm_g1 if grp == "g1" else co.msm_g2
        for th in (1, 8):
            assert fn(dense, sc, dens, threads=th).hex() == case["expect"], (grp, case["pattern"], th)


def test_ntt_kats_at_1k():
    rec = golden("ntt_kat_1k.json")
    lg = rec["log_m"]
    a = k1.kat_scalars("uniform", rec["input_seed"], 1 << lg)
    assert k1.vec_digest(a) == rec["input_sha256"]
    am = mont_bytes(a, bn.R)
    for mode, name in enumerate(("fft", "ifft", "coset_fft", "icoset_fft")):
        for th in (1, 8):
            got = from_mont_bytes(co.fft(am, lg, mode, threads=th), bn.R)
            assert k1.vec_digest(got) == rec[name]["sha256"], (name, th)
            assert ["%x" % x for x in got[:2]] == rec[name]["first"] and ["%x" % x for x in got[-2:]] == rec[name]["last"]


def test_proof_at_a_2p10_domain():
    fx = golden("proof_chain10.json")
    r1, w = k1.chain10_circuit(fx)
    assert r1.domain_log() == 10
    csr = co.CsrR1CS.from_r1cs(r1)
    wb = co.witness_bytes(w)
    a, b, c, ad, bi, bd = co.r1cs_eval(csr, wb, threads=4)
    assert k1.vec_digest(from_mont_bytes(a, bn.R)) == fx["a_sha256"]
    assert k1.vec_digest(from_mont_bytes(b, bn.R)) == fx["b_sha256"]
    assert k1.vec_digest(from_mont_bytes(c, bn.R)) == fx["c_sha256"]
    assert [int(np.count_nonzero(x)) for x in (ad, bi, bd)] == fx["density_totals"]
    h = from_canon_bytes(co.h_coeffs(a, b, c, fx["n_constraints"] + fx["n_inputs"], 10, threads=4))
    assert k1.vec_digest(h) == fx["h_sha256"]
    assert ["%x" % x for x in h[:4]] == fx["h_first"] and ["%x" % x for x in h[-4:]] == fx["h_last"]
    pbytes = k1.chain10_params()
    assert hashlib.sha256(pbytes).hexdigest() == fx["params_sha256"] and len(pbytes) == fx["params_len"]
    params, used = g.params_read(pbytes)
    assert used == len(pbytes)
    mp = co.MontParams.from_params(params)
    for th in (1, 8):
        assert co.prove(mp, csr, wb, int(fx["r"], 16), int(fx["s"], 16), threads=th).hex() == fx["proof"]
