"""The pure-Python oracle against public anchors, its own identities and the committed golden vectors."""
import random

from oracle import bn254 as bn
from oracle import groth16 as g
from tests.util import golden, r1cs_from_fixture


def test_curve_constants_and_public_anchor():
    assert bn.is_on_curve(bn.G1_GEN, bn.G1_B) and bn.is_on_curve(bn.G2_GEN, bn.G2_B)
    p2 = bn.pt_double(bn.G1_GEN)   # EIP-196 / alt_bn128 test value of 2*G1
    assert p2[0].v == 1368015179489954701390400359078579693043519447331113978918064868415326638035
    assert p2[1].v == 9918110051302171585080402603319702774565515993150576347155970296011118125764
    assert bn.pt_mul(bn.G1_GEN, bn.R) is None and bn.pt_mul(bn.G2_GEN, bn.R) is None
    assert pow(bn.FR_ROOT_OF_UNITY, 1 << 28, bn.R) == 1 and pow(bn.FR_ROOT_OF_UNITY, 1 << 27, bn.R) != 1


def test_pairing_bilinear():
    a, b = 1234567, 7654321
    e1 = bn.pairing(bn.pt_mul(bn.G2_GEN, b), bn.pt_mul(bn.G1_GEN, a))
    assert e1 == bn.pairing(bn.G2_GEN, bn.pt_mul(bn.G1_GEN, a * b))
    assert not (e1 == bn.Fq12.one())


def test_fft_matches_naive_dft_and_inverts():
    rng = random.Random(1)
    d = g.Domain(4)
    a = [rng.randrange(bn.R) for _ in range(16)]
    b = list(a)
    d.fft(b)
    assert b == g.dft_naive(a, d.omega)
    d.ifft(b)
    assert b == a
    c = list(a)
    d.coset_fft(c)
    d.icoset_fft(c)
    assert c == a


def test_golden_proofs_verify_under_the_pairing_equation():
    """The reference's own acceptance test (prover/src/groth16/prover.rs:291-305): verify(prove(x)) is
    true, and false for a wrong public input."""
    for name in ("proof_factor.json", "proof_chain7.json"):
        fx = golden(name)
        params, used = g.params_read(bytes.fromhex(fx["params"]))
        assert used == len(fx["params"]) // 2
        pb = bytes.fromhex(fx["proof"])
        v = [int.from_bytes(pb[32 * i:32 * i + 32], "big") for i in range(8)]
        proof = ((bn.Fq(v[0]), bn.Fq(v[1])), (bn.Fq2(v[2], v[3]), bn.Fq2(v[4], v[5])), (bn.Fq(v[6]), bn.Fq(v[7])))
        w = [int(x, 16) for x in fx["witness"]]
        pub = w[1:fx["n_inputs"]]
        assert g.verify_proof(params.vk, proof, pub)
        bad = list(pub)
        bad[0] = (bad[0] + 1) % bn.R
        assert not g.verify_proof(params.vk, proof, bad)


def test_golden_proof_regenerates_and_matches_trapdoor():
    fx = golden("proof_factor.json")
    r1 = r1cs_from_fixture(fx)
    w = [int(x, 16) for x in fx["witness"]]
    td = [int(x, 16) for x in fx["trapdoor"]]
    params, trap = g.generate_parameters(r1, *td)
    assert g.params_write(params).hex() == fx["params"]
    r, s = int(fx["r"], 16), int(fx["s"], 16)
    proof = g.create_proof(r1, params, w, r, s)
    assert g.proof_bytes(proof).hex() == fx["proof"]
    assert g.proof_json(proof, w[1:2]) == fx["proof_json"]
    eA, eB, eC = g.trapdoor_expected_proof(r1, trap, w, r, s)
    assert proof == (bn.pt_mul(bn.G1_GEN, eA), bn.pt_mul(bn.G2_GEN, eB), bn.pt_mul(bn.G1_GEN, eC))
    # public inputs of example/input.json: r = 6  (format.rs:94-97 prints decimal)
    assert '"public_inputs":["6"]' in fx["proof_json"]


def test_h_identity_at_random_point():
    """a(t) b(t) - c(t) = h(t) Z(t) for the H the oracle computes."""
    fx = golden("proof_chain7.json")
    r1 = r1cs_from_fixture(fx)
    w = [int(x, 16) for x in fx["witness"]]
    h = [int(x, 16) for x in fx["h"]]
    a, b, c, *_ = g.eval_constraints(r1, w)
    d = g.Domain(r1.domain_log())
    t = 0x1234567890abcdef
    # Lagrange interpolation through ifft
    co = []
    for v in (a, b, c):
        v = d.pad(v)
        d.ifft(v)
        co.append(sum(x * pow(t, i, bn.R) for i, x in enumerate(v)) % bn.R)
    ht = sum(x * pow(t, i, bn.R) for i, x in enumerate(h)) % bn.R
    assert (co[0] * co[1] - co[2]) % bn.R == ht * d.z(t) % bn.R


def test_params_and_pk_container_round_trip():
    fx = golden("proof_factor.json")
    pk = bytes.fromhex(fx["pk_container"])
    qeqs, ignore, off = g.pk_container_read(pk)
    assert ignore == [] and len(qeqs) == 1
    # za's QEQ for p*q===r: a = p, b = q, c = -r (constraint.rs:51: a*b + c = 0)
    assert qeqs[0] == ([(2, 1)], [(3, 1)], [(1, bn.R - 1)])
    assert pk[off:].hex() == fx["params"]
    for bad_first_byte, msg in ((0x80, "compression"), (0x40, "infinity")):
        buf = bytearray(bytes.fromhex(fx["params"]))
        buf[0] |= bad_first_byte
        try:
            g.params_read(bytes(buf))
            assert False
        except ValueError as e:
            assert msg in str(e)


def test_bellman_multiexp_equals_definition():
    rng = random.Random(5)
    for n in (1, 31, 32, 40):
        pts = [bn.pt_mul(bn.G1_GEN, rng.randrange(1, bn.R)) for _ in range(n)]
        sc = [rng.choice([0, 1, bn.R - 1, rng.randrange(bn.R)]) for _ in range(n)]
        assert g.multiexp_bellman(pts, None, sc) == bn.msm_naive(pts, sc)


def test_poseidon_chain_generator_is_pinned_by_the_reference_known_answers():
    """BASELINE config 3's circuit family.  The hash values are the reference's own
    (interop/circuits/circomlib/test/poseidoncircuit.js:41,54)."""
    from oracle import circuits as ci
    assert ci.poseidon_hash([1, 2]) == 12242166908188651009877250812424843524687801523336557272219921456462821518061
    assert ci.poseidon_hash([3, 4]) == 17185195740979599334254027721507328033796809509313949281114643312710535000993
    r1, w, hs = ci.poseidon_chain_r1cs(12)
    assert len(r1.constraints) + r1.num_inputs == 1 << 12
    assert hs[0] == ci.poseidon_hash([2, 1]) and w[1] == hs[-1]
    assert r1.is_satisfied(w)
    bad = list(w); bad[1] = (bad[1] + 1) % ci.R
    assert not r1.is_satisfied(bad)


def test_workload_csr_arrays_match_the_oracle_layout():
    """za_amd.workloads (bench.py's circuit generator) hands zg_r1cs_load the same CSR arrays and witness bytes as the
    oracle's own conversion of the same constraint list."""
    import numpy as np
    from oracle import circuits as ci, coracle as co
    from za_amd import workloads as wl
    n_inputs, n_aux, cons, w, hs = wl.poseidon_chain(11, n_perm=1)
    r1, w2, hs2 = ci.poseidon_chain_r1cs(11, n_perm=1)
    assert (w, hs) == (w2, hs2) and r1.is_satisfied(w)
    csr = co.CsrR1CS.from_r1cs(r1)
    rp, cl, cf = wl.csr_arrays(cons)
    for k in range(3):
        assert np.array_equal(rp[k], csr.row_ptr[k]) and np.array_equal(cl[k], csr.col[k]) and np.array_equal(cf[k], csr.coeff[k])
    assert wl.witness_bytes(w) == co.witness_bytes(w)
    assert wl.R == ci.R


def test_h_in_evaluation_form_is_the_same_group_element():
    """The identity behind za_amd/csrc/heval.hip, by definitions only (no butterfly code on either side):
    sum_i h_i H_i (bellman: coefficients x h query) = sum_p e[p] H'[p] (coset evaluations x transformed bases)."""
    from oracle.circuits import chain_r1cs
    r1, w = chain_r1cs(5, n_inputs=2, seed=55, bit_fraction=0.5)
    params, _ = g.generate_parameters(r1, 11, 12, 13, 14, 15)
    h = g.h_coefficients(r1, w)
    lhs = None
    for hi, Hi in zip(h, params.h):
        lhs = bn.pt_add(lhs, bn.pt_mul(Hi, hi))
    e = g.h_eval_scalars(r1, w)
    Hp = g.h_eval_bases(params.h, 5)
    assert g.h_eval_bases_fft(params.h, 5) == Hp      # the transform schedule of heval.hip gives the defined array
    rhs = None
    for ep, P in zip(e, Hp):
        rhs = bn.pt_add(rhs, bn.pt_mul(P, ep))
    assert lhs == rhs and lhs is not None
    # and range by range, as the ranks of a sharded proof hold them (za_amd/dist.py): three ranges of ceil(32 / 3)
    parts = None
    for lo in range(0, 32, 11):
        acc = None
        for ep, P in list(zip(e, Hp))[lo:lo + 11]:
            acc = bn.pt_add(acc, bn.pt_mul(P, ep))
        parts = bn.pt_add(parts, acc)
    assert parts == lhs


def test_c_rows_folded_into_the_l_multiexp_is_the_same_group_element():
    """The identity behind ZG_OPT_C_FOLD (za_amd/csrc/prover.hip ensure_c_fold, heval.hip c_fold), by definitions only:
    sum_i h_i H_i = sum_p e'[p] H'[p] + sum_j c_j C_j  with e' the quotient without its C term (four transforms) and c_j the
    values of the R1CS rows' C sides, zero padded -- for a witness that satisfies the circuit and for one that does not
    (bellman interpolates (a b - c) / Z on the coset whatever it is and drops the coefficient of x^(m-1); every sum here runs
    over the m - 1 points the h query has; nothing in the identity uses a b = c on the domain).
    A row that evaluates to 2 enters as a ONE on the doubled base."""
    from oracle.circuits import chain_r1cs
    r1, w = chain_r1cs(4, n_inputs=2, seed=77, bit_fraction=0.5)
    params, _ = g.generate_parameters(r1, 21, 22, 23, 24, 25)
    Hp = g.h_eval_bases(params.h, 4)
    Cj = g.c_fold_bases(params.h, 4)
    assert g.c_fold_bases_fft(params.h, 4) == Cj      # the transform schedule of heval.hip gives the defined array
    C2 = [bn.pt_add(P, P) for P in Cj]
    bad = list(w)
    bad[len(bad) // 2] = (bad[len(bad) // 2] + 5) % bn.R
    assert r1.is_satisfied(w) and not r1.is_satisfied(bad)
    def h_truncated(wit):      # bellman's sequence (oracle/groth16.py h_coefficients) without the satisfaction check
        a, b, c, *_ = g.eval_constraints(r1, wit)
        d = g.Domain(r1.domain_log())
        a, b, c = d.pad(a), d.pad(b), d.pad(c)
        for v in (a, b, c):
            d.ifft(v)
            d.coset_fft(v)
        a = [(x * y - z) % bn.R for x, y, z in zip(a, b, c)]
        d.divide_by_z_on_coset(a)
        d.icoset_fft(a)
        return a[: d.m - 1]
    assert h_truncated(w) == g.h_coefficients(r1, w)
    for wit in (w, bad):
        lhs = None
        for hi, Hi in zip(h_truncated(wit), params.h):
            lhs = bn.pt_add(lhs, bn.pt_mul(Hi, hi))
        rhs = None
        for ep, P in zip(g.h_eval_scalars_without_c(r1, wit), Hp):
            rhs = bn.pt_add(rhs, bn.pt_mul(P, ep))
        c = g.eval_constraints(r1, wit)[2]
        for j, cj in enumerate(c):                     # (rows beyond the constraints -- inputs, padding -- evaluate to 0)
            if cj == 2:
                rhs = bn.pt_add(rhs, C2[j])
            else:
                rhs = bn.pt_add(rhs, bn.pt_mul(Cj[j], cj))
        assert lhs == rhs and lhs is not None


def test_psi_membership_test_is_exact():
    """The device's G2 membership test (za_amd/csrc/keyload.hip) is [u+1]P + psi([u]P) + psi^2([u]P) = psi^3([2u]P) instead
    of pairing_ce's [r]P = O.  Both directions of "accepts the same set" are integer facts (oracle/bn254.py), checked
    here, and the two tests are compared on points in G2, outside it, of small cofactor order, and on sums of both kinds."""
    import math
    q, r, u = bn.Q, bn.R, bn.BN_U
    t = 6 * u * u + 1                      # trace of Frobenius: #E(Fq) = q + 1 - t = r
    assert q + 1 - t == r
    cof = 2 * q - r                        # #E'(Fq2) = r (2q - r)
    # "=>": psi = [q] on G2 and f(q) = 0 (mod r)
    lam = q % r
    assert ((u + 1) + u * lam + u * lam * lam - 2 * u * pow(lam, 3, r)) % r == 0
    assert bn.g2_psi(bn.G2_GEN) == bn.pt_mul(bn.G2_GEN, lam) and bn.is_on_curve(bn.g2_psi(bn.G2_GEN), bn.G2_B)
    # "<=": f(X) mod (X^2 - t X + q) = a + b X; its norm a^2 + a b t + b^2 q annihilates the kernel of f(psi) and is
    # coprime to the cofactor (and a multiple of r, as it must be)
    c = [u + 1, u, u, -2 * u]
    while len(c) > 2:
        top = c.pop()
        c[-1] += top * t
        c[-2] -= top * q
    a, b = c
    norm = a * a + a * b * t + b * b * q
    assert math.gcd(norm, cof) == 1 and norm % r == 0 and cof % r != 0
    # the two tests side by side
    for k in (1, 2, 0xdeadbeef, r - 1):
        P = bn.pt_mul(bn.G2_GEN, k)
        assert bn.g2_in_subgroup_by_psi(P) and bn.g2_in_subgroup_by_order(P)
    assert cof % 10069 == 0
    for seed in (1, 2, 3):
        T = bn.g2_twist_point(seed)
        assert bn.is_on_curve(T, bn.G2_B)
        assert not bn.g2_in_subgroup_by_psi(T) and not bn.g2_in_subgroup_by_order(T)
        in_g2 = bn.pt_mul(T, cof, order=1 << 600)                  # cofactor cleared
        assert bn.g2_in_subgroup_by_psi(in_g2) and bn.g2_in_subgroup_by_order(in_g2)
        small = bn.pt_mul(T, r * (cof // 10069), order=1 << 600)    # order 1 or 10069
        if small is not None:
            assert bn.pt_mul(small, 10069, order=1 << 20) is None
            assert not bn.g2_in_subgroup_by_psi(small) and not bn.g2_in_subgroup_by_order(small)
            mixed = bn.pt_add(in_g2, small)
            assert not bn.g2_in_subgroup_by_psi(mixed) and not bn.g2_in_subgroup_by_order(mixed)
