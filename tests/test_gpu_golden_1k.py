"""The HIP path against the Python-oracle fixtures at SURVEY 8(c)'s sizes (tests/golden/gen_golden_1k.py): multiexps
over 2^10 bases (per-call and resident-bases entry points, window table on and off), the four NTT flavours at 2^10, and a
full (key bytes, R1CS, witness, r, s) -> proof.json at a 2^10 domain, including the GPU set-up reproducing the key."""
import hashlib

import numpy as np
import pytest

from oracle import bn254 as bn
from oracle import coracle as co
from tests import golden_1k as k1
from tests.util import from_canon_bytes, from_mont_bytes, golden, jac_to_affine_bytes, mont_bytes

pytestmark = pytest.mark.gpu


def test_msm_kats_at_1k(ctx):
    kat = golden("msm_kat_1k.json")
    n = kat["n"]
    bases = {}
    for case in kat["cases"]:
        grp = case["group"]
        g2 = grp == "g2"
        if grp not in bases:
            bases[grp] = k1.kat_bases(grp, case["bases_seed"], n)
        sc = k1.le_bytes(k1.kat_scalars(case["pattern"], case["scalars_seed"], n))
        dens = k1.kat_density(case["density_seed"], n) if case["density_seed"] else None
        elem, fb = (128, 64) if g2 else (64, 32)
        dense = bases[grp] if dens is None else b"".join(bases[grp][elem * i:elem * (i + 1)] for i in range(n) if dens[i])
        got = (ctx.msm_g2 if g2 else ctx.msm_g1)(dense, sc, dens)
        assert jac_to_affine_bytes(got, fb).hex() == case["expect"], (grp, case["pattern"], dens is not None)
        if dens is None:   # resident bases, with and without the fixed-base window table, dense hint on and off
            for budget in (0.0, 1.0):
                H = ctx.msm_bases(bases[grp], g2=g2, table_budget_gb=budget)
                for hint in (False, True):
                    got = ctx.msm_resident(H, sc, n, dense_hint=hint, g2=g2)
                    assert jac_to_affine_bytes(got, fb).hex() == case["expect"], (grp, case["pattern"], budget, hint)
                ctx.msm_bases_free(H)


def test_ntt_kats_at_1k(ctx):
    rec = golden("ntt_kat_1k.json")
    lg = rec["log_m"]
    a = k1.kat_scalars("uniform", rec["input_seed"], 1 << lg)
    am = mont_bytes(a, bn.R)
    for name, inv, cos in (("fft", False, False), ("ifft", True, False), ("coset_fft", False, True), ("icoset_fft", True, True)):
        got = from_mont_bytes(ctx.ntt(am, lg, inverse=inv, coset=cos), bn.R)
        assert k1.vec_digest(got) == rec[name]["sha256"], name
        assert ["%x" % x for x in got[:2]] == rec[name]["first"] and ["%x" % x for x in got[-2:]] == rec[name]["last"]


def test_proof_at_a_2p10_domain(ctx):
    import za_amd
    fx = golden("proof_chain10.json")
    r1, w = k1.chain10_circuit(fx)
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    assert R.log_m == 10 and [R.a_aux_total, R.b_input_total, R.b_aux_total] == fx["density_totals"]
    wb = co.witness_bytes(w)
    a, b, c = R.eval(wb)
    assert k1.vec_digest(from_mont_bytes(a, bn.R)) == fx["a_sha256"]
    assert k1.vec_digest(from_mont_bytes(b, bn.R)) == fx["b_sha256"]
    assert k1.vec_digest(from_mont_bytes(c, bn.R)) == fx["c_sha256"]
    h = from_canon_bytes(ctx.h_scalars(a, b, c, fx["n_constraints"] + fx["n_inputs"], 10))
    assert k1.vec_digest(h) == fx["h_sha256"]
    pbytes = k1.chain10_params()
    P, used = ctx.params_load(pbytes, checked=True)
    assert used == len(pbytes)
    r, s = int(fx["r"], 16), int(fx["s"], 16)
    proof = ctx.prove(P, R, wb, r, s)
    assert proof.hex() == fx["proof"]
    assert za_amd.proof_json(proof, w[1:fx["n_inputs"]]) == fx["proof_json"]
    assert ctx.prove_packed(P, R, za_amd.pack_witness(wb), r, s) == proof
    d_w = ctx.upload(wb)
    assert ctx.prove_dev(P, R, d_w, r, s) == proof
    ctx.free(d_w)
    assert za_amd.verify(pbytes, proof, w[1:fx["n_inputs"]]) is True
    # the trapdoor set-up on the GPU writes the oracle's key byte for byte, and proves to the same bytes with it
    P2, p2 = ctx.setup(R, *[int(x, 16) for x in fx["trapdoor"]])
    assert hashlib.sha256(p2).hexdigest() == fx["params_sha256"]
    assert ctx.prove(P2, R, wb, r, s) == proof
    P.close(); P2.close(); R.close()
