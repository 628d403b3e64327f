"""Device point arithmetic (csrc/ecd.h on gfx950) operation by operation against the Python oracle -- bit-exact.

SURVEY section 7 step 4: "Fq mul + G1 mixed add parity-checked on random vectors".  Every form the multiexp kernels use
is run on its own through zg_selftest_curve_dev: XYZZ + XYZZ, XYZZ + affine, XYZZ + affine with the base's y negated
limb-wise (what k_accumulate does for a negative digit), doubling, and chains of two additions (the second one's
operands are the first one's outputs, so the lazy-reduction bounds are those of a real chain) -- one lane per point
(FqL, Fq2L), G2 on lane pairs (Fq2PL) and G1 on four lanes (FqQL: add_quad, add_mixed_quad).  The operands arrive in
random XYZZ representations (x z^2, y z^3, z^2, z^3).  Expected values: pairing_ce's group law as restated in
oracle/bn254.py (affine chord-and-tangent), i.e. what bellman's add_assign / add_assign_mixed compute up to the
representation (reference call site prover/src/groth16/prover.rs:173)."""
import ctypes as C

import numpy as np
import pytest

from oracle import bn254 as bn
from tests.util import golden

pytestmark = pytest.mark.gpu

G = {1: (bn.G1_GEN, bn.g1_to_mont_bytes, 64, 32), 2: (bn.G2_GEN, bn.g2_to_mont_bytes, 128, 64)}
LANES = {1: (1, 4), 2: (1, 2)}


def _expected(op, p, q):
    if op in (0, 1):
        return bn.pt_add(p, q)
    if op == 2:
        return bn.pt_add(p, bn.pt_neg(q))
    if op == 3:
        return bn.pt_double(p)
    if op == 4:
        return bn.pt_add(bn.pt_add(p, q), q)
    return bn.pt_add(bn.pt_add(p, q), p)


def _z_bytes(rng, group, n):
    """n non-zero field elements (Fq, or Fq2 as c0 || c1), Montgomery LE"""
    out = []
    for _ in range(n):
        comps = [int.from_bytes(rng.bytes(32), "little") % (bn.Q - 1) + 1 for _ in range(group)]
        out.append(b"".join(bn.le32(bn.mont(c, bn.Q)) for c in comps))
    return b"".join(out)


def _run(ctx, group, lanes, op, ps, qs, zp, zq):
    _, enc, nb, _ = G[group]
    n = len(ps)
    P = np.frombuffer(b"".join(enc(p) for p in ps), dtype=np.uint8)
    Q = np.frombuffer(b"".join(enc(q) for q in qs), dtype=np.uint8)
    ZP = np.frombuffer(zp, dtype=np.uint8)
    ZQ = np.frombuffer(zq, dtype=np.uint8)
    out = np.zeros(n * nb, dtype=np.uint8)
    ctx._check(ctx.lib.zg_selftest_curve_dev(ctx.h, C.c_int(group), C.c_int(lanes), C.c_int(op),
                                             P.ctypes.data_as(C.c_void_p), ZP.ctypes.data_as(C.c_void_p),
                                             Q.ctypes.data_as(C.c_void_p), ZQ.ctypes.data_as(C.c_void_p),
                                             out.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
    return [out[i * nb:(i + 1) * nb].tobytes() for i in range(n)]


def _walk(group, n, seed):
    """n pseudo-random group elements: a walk k_(i+1) = k_i + d_j with eight random step points (affine additions)"""
    gen = G[group][0]
    rng = np.random.default_rng(seed)
    steps = [bn.pt_mul(gen, int.from_bytes(rng.bytes(32), "little") % bn.R) for _ in range(8)]
    p = bn.pt_mul(gen, int.from_bytes(rng.bytes(32), "little") % bn.R)
    out = []
    for i in range(n):
        out.append(p)
        p = bn.pt_add(p, steps[int(rng.integers(0, 8))])
    return out


@pytest.mark.parametrize("group", [1, 2])
def test_device_additions_match_the_golden_vectors(ctx, group):
    """tests/golden/curve_vectors.json: the pairs the host curve code and the C oracle are pinned by, now on the device"""
    cv = golden("curve_vectors.json")
    _, enc, nb, zb = G[group]
    dec = bn.g1_from_mont_bytes if group == 1 else bn.g2_from_mont_bytes
    rows = cv["g%d_add" % group]
    ps = [dec(bytes.fromhex(r[0])) for r in rows]
    qs = [dec(bytes.fromhex(r[1])) for r in rows]
    rng = np.random.default_rng(11 + group)
    for lanes in LANES[group]:
        for op in (0, 1):
            got = _run(ctx, group, lanes, op, ps, qs, _z_bytes(rng, group, len(ps)), _z_bytes(rng, group, len(ps)))
            assert [g.hex() for g in got] == [r[2] for r in rows], (group, lanes, op)


@pytest.mark.parametrize("group,n", [(1, 4096), (2, 2048)])
def test_device_point_operations_on_random_pairs(ctx, group, n):
    _, enc, nb, zb = G[group]
    ps = _walk(group, n, 100 + group)
    qs = _walk(group, n, 200 + group)
    rng = np.random.default_rng(300 + group)
    for lanes in LANES[group]:
        for op in range(6):
            zp, zq = _z_bytes(rng, group, n), _z_bytes(rng, group, n)
            got = _run(ctx, group, lanes, op, ps, qs, zp, zq)
            exp = [enc(_expected(op, p, q)) for p, q in zip(ps, qs)]
            bad = [i for i in range(n) if got[i] != exp[i]]
            assert not bad, (group, lanes, op, bad[:8])


@pytest.mark.parametrize("group", [1, 2])
def test_device_point_operations_edge_cases(ctx, group):
    """identity on either side, P = Q (the doubling branch of every addition), P = -Q (cancellation), and the
    chained forms landing on them: (P + (-P)) + (-P), (P + P) + P, (P + inf) + P"""
    gen, enc, nb, zb = G[group]
    a = bn.pt_mul(gen, 0x1234567)
    b = bn.pt_mul(gen, 0x7654321)
    pairs = [(None, None), (None, a), (a, None), (a, a), (a, bn.pt_neg(a)), (b, a), (bn.pt_neg(a), a),
             (bn.pt_double(a), a), (bn.pt_double(a), bn.pt_neg(a)), (bn.pt_neg(bn.pt_double(a)), a),
             (a, bn.pt_double(a)), (gen, gen), (gen, bn.pt_neg(gen))]
    # every pair four times over so that quads / pairs of lanes next to each other take different branches
    ps = [p for p, _ in pairs for _ in range(4)] + [p for p, _ in pairs]
    qs = [q for _, q in pairs for _ in range(4)] + [q for _, q in reversed(pairs)]
    rng = np.random.default_rng(400 + group)
    for lanes in LANES[group]:
        for op in range(6):
            zp, zq = _z_bytes(rng, group, len(ps)), _z_bytes(rng, group, len(ps))
            got = _run(ctx, group, lanes, op, ps, qs, zp, zq)
            exp = [enc(_expected(op, p, q)) for p, q in zip(ps, qs)]
            bad = [i for i in range(len(ps)) if got[i] != exp[i]]
            assert not bad, (group, lanes, op, bad[:8])


def test_selftest_refuses_unknown_forms(ctx):
    z = np.zeros(128, dtype=np.uint8)
    for group, lanes, op in ((1, 2, 0), (2, 4, 0), (3, 1, 0), (1, 1, 6), (1, 3, 0)):
        rc = ctx.lib.zg_selftest_curve_dev(ctx.h, C.c_int(group), C.c_int(lanes), C.c_int(op), z.ctypes.data_as(C.c_void_p),
                                           z.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p),
                                           z.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), C.c_size_t(1))
        assert rc != 0, (group, lanes, op)
