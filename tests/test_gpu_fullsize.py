"""BASELINE.json's large configurations at full size on one MI355X (the 8-GPU forms shard these same jobs):
config 4 (2^22, 97 %-bit witness) byte for byte against the C oracle, config 5's uniform-Fr shape at 2^24 against the
discrete-log expectation, and the 2^26-point standalone G1 MSM."""
import os

import numpy as np
import pytest

from oracle import bn254 as bn
from oracle import coracle as co
from oracle import groth16 as g
from tests.util import download_params, table_kinds
from za_amd import synth

pytestmark = pytest.mark.gpu
THREADS = min(os.cpu_count() or 8, 64)


def test_config4_at_2p22_matches_the_c_oracle(ctx):
    """The bench workload itself (SplitMix64 seed 0x5A, 97 % bits, key from the GPU trapdoor set-up): zg_prove from a
    host witness equals the oracle's create_proof byte for byte; the public input verifies under the pairing."""
    import za_amd
    from za_amd import synth
    lg = 22
    sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
    R = sr.load(ctx)
    P, _ = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)
    r, s = 0x1234567, 0x7654321
    got = ctx.prove(P, R, sr.witness, r, s)
    kinds = table_kinds(P)
    assert kinds[0] == 1 and kinds[1] == 2, kinds        # h: window table; l: subset-sum table (bit-heavy witness)
    assert ctx.last_stats()["c_folded"] == (dict(za_amd._options_from_env()).get(za_amd.OPT_C_FOLD, 1) != 0)   # the default at 2^22
    mp = download_params(ctx, P)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    assert co.prove(mp, csr, sr.witness, r, s, threads=THREADS) == got
    # the same through the witness-resident entry point and a second call (cached tables / workspaces)
    d_w = ctx.upload(sr.witness)
    assert ctx.prove_dev(P, R, d_w, r, s) == got
    assert ctx.prove(P, R, sr.witness, r, s) == got
    ctx.free(d_w)
    w1 = int.from_bytes(bytes(sr.witness[32:64]), "little")
    assert za_amd.verify(P.write_vk(), got, [w1]) is True
    assert za_amd.verify(P.write_vk(), got, [w1 ^ 1]) is False
    P.close(); R.close(); sr.close()


def test_config5_shape_uniform_witness_at_2p24_matches_discrete_log_expectation(ctx):
    """Uniform-Fr coefficients and witness (config 5's shape, SplitMix64 seed 0x26) at 2^24: every query takes a
    fixed-base window table (13 x the key), no 0/1 shortcuts anywhere.  With every base's discrete log known the
    proof's three elements are k*G for scalars computed independently (R1CS evaluation and H from the C oracle)."""
    from za_amd import synth
    lg = 24
    sr = synth.SynthR1CS(lg, 2, 0.0, 0x26)
    R = sr.load(ctx)
    m = 1 << lg
    ni = sr.n_inputs
    n_a, n_b = ni + R.a_aux_total, R.b_input_total + R.b_aux_total
    seed = 26000
    P = synth.synth_params(ctx, m - 1, sr.n_aux, n_a, n_b, seed)
    r, s = 0xABCDEF01, 0x10FEDCBA
    got = ctx.prove(P, R, sr.witness, r, s)
    assert table_kinds(P) == [1, 1, 1, 1, 1]
    assert got == _dlog_expectation(ctx, sr, R, lg, seed, r, s)
    P.close(); R.close(); sr.close()


def _dlog_expectation(ctx, sr, R, lg, seed, r, s):
    """The proof a synth_params key (every base k_i G, alpha = 3G, beta = 5G, delta = 7G) must give: A, B, C as scalar
    multiples of the generators, with R1CS evaluation and H from the C oracle."""
    m, ni = 1 << lg, sr.n_inputs
    n_a, n_b = ni + R.a_aux_total, R.b_input_total + R.b_aux_total
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    a, b, c, ad, bi, bd = co.r1cs_eval(csr, sr.witness, threads=THREADS)
    h = co.h_coeffs(a, b, c, sr.n_constraints + ni, lg, threads=THREADS)
    del a, b, c
    w = np.frombuffer(sr.witness, dtype=np.uint8).reshape(-1, 32)
    a_idx = np.concatenate([np.arange(ni), ni + np.nonzero(ad)[0]])
    b_idx = np.concatenate([np.nonzero(bi)[0], ni + np.nonzero(bd)[0]])
    assert len(a_idx) == n_a and len(b_idx) == n_b
    sh = co.dot_u64(h, synth.splitmix64_array(seed + 1, m - 1)) % bn.R
    del h
    sl = co.dot_u64(w[ni:], synth.splitmix64_array(seed + 2, sr.n_aux)) % bn.R
    sa = co.dot_u64(np.ascontiguousarray(w[a_idx]), synth.splitmix64_array(seed + 3, n_a)) % bn.R
    sb = co.dot_u64(np.ascontiguousarray(w[b_idx]), synth.splitmix64_array(seed + 4, n_b)) % bn.R
    alpha, beta, delta = 3, 5, 7
    eA = (alpha + sa + r * delta) % bn.R
    eB = (beta + sb + s * delta) % bn.R
    eC = (r * s * delta + s * alpha + r * beta + s * sa + r * sb + sh + sl) % bn.R
    A = bn.g1_from_mont_bytes(co.g1_mul(bn.g1_to_mont_bytes(bn.G1_GEN), eA))
    B = bn.g2_from_mont_bytes(co.g2_mul(bn.g2_to_mont_bytes(bn.G2_GEN), eB))
    Cc = bn.g1_from_mont_bytes(co.g1_mul(bn.g1_to_mont_bytes(bn.G1_GEN), eC))
    return g.proof_bytes((A, B, Cc))


def test_config5_2p26_proof_matches_discrete_log_expectation(ctx):
    """BASELINE config 5 itself on one GPU: the 2^26 random R1CS (uniform-Fr coefficients and witness, SplitMix64 seed
    0x26, n + l = 2^26), a key of 2^26-point queries with known discrete logs, zg_prove from host memory.  What differs
    from 2^24 at this size: the table budget logic (h, a, b_g2 get window tables, the rest of the 288 GB goes to the
    sort buffers), 32-bit index headroom (12 digits x 2^26 entries), four-part h job."""
    lg = 26
    sr = synth.SynthR1CS(lg, 2, 0.0, 0x26)
    R = sr.load(ctx)
    m, ni = 1 << lg, sr.n_inputs
    n_a, n_b = ni + R.a_aux_total, R.b_input_total + R.b_aux_total
    seed = 2600026
    P = synth.synth_params(ctx, m - 1, sr.n_aux, n_a, n_b, seed)
    r, s = 0xABCDEF0123, 0x10FEDCBA98
    got = ctx.prove(P, R, sr.witness, r, s)
    assert P.table_info(0)[0] == 1       # the h query has its window table
    assert ctx.prove(P, R, sr.witness, r, s) == got     # second call: cached tables / workspaces
    want = _dlog_expectation(ctx, sr, R, lg, seed, r, s)
    assert got == want
    P.close(); R.close(); sr.close()


def test_g1_msm_2p26_points_discrete_log_identity(ctx):
    """Config 5's standalone multiexp: 2^26 G1 bases k_i G, uniform scalars.  sum s_i P_i = (sum s_i k_i) G through the
    window table (resident scalars and host scalars uploaded in chunks) and through the plain per-window schedule."""
    from za_amd import synth
    n = 1 << 26
    seed_b, seed_s = 2626, 99
    d_b = synth.synth_bases_g1(ctx, n, seed_b)
    d_s = synth.synth_scalars(ctx, n, seed_s)
    sc = ctx.d2h(d_s, 32 * n)
    e = co.dot_u64(sc, synth.splitmix64_array(seed_b, n)) % bn.R
    want = co.g1_mul(bn.g1_to_mont_bytes(bn.G1_GEN), e)            # affine Montgomery bytes
    H = ctx.msm_bases_g1_dev(d_b, n, table_budget_gb=120.0)
    info = ctx.msm_bases_info(H)
    assert info["n"] == n and info["table_bytes"] >= 10 * 64 * n, info
    got = ctx.msm_resident_dev(H, d_s, n, dense_hint=True)
    assert got[:64] == want and got[64:] == bn.le32(bn.mont(1, bn.Q))
    # host scalars from a pinned buffer: four dependent parts gated on four upload chunks
    pin = ctx.host_alloc(32 * n)
    pin[:] = sc
    assert ctx.msm_resident(H, pin, n, dense_hint=True)[:64] == want
    # a prefix of the bases (n' < n: no table) with unhinted scalars
    n2 = (1 << 20) + 12345
    e2 = co.dot_u64(sc[:32 * n2], synth.splitmix64_array(seed_b, n2)) % bn.R
    assert ctx.msm_resident(H, pin, n2, dense_hint=False)[:64] == co.g1_mul(bn.g1_to_mont_bytes(bn.G1_GEN), e2)
    ctx.host_free(pin)
    ctx.msm_bases_free(H)
    # the per-call entry point (ABI-form bases converted on every call, per-window schedule)
    assert ctx.msm_g1_dev(d_b, d_s, n)[:64] == want
    ctx.free(d_b); ctx.free(d_s)


def test_largest_legal_domain_2p27_multiexp_and_transform(ctx):
    """bellman's limit: Fr has 2-adicity 28 and create_proof refuses a domain of 2^28
    (prover/src/groth16/prover.rs:173 -> PolynomialDegreeTooLarge), so 2^27 is the largest size the path can meet --
    "2^26 constraints" read literally.  Here the data reaches the guards that inspection had to vouch for: a 2^27-point G1
    multiexp through its window table (12 rows x 2^27 entries: table indices up to 1.6e9 of the 2^31 the sort's value word
    holds; four parts of 4.0e8 (point, digit) entries each) against the discrete-log expectation, and the Fr transform at
    2^27 (4 GiB per vector, 3.2 GiB per twiddle table: byte offsets beyond 32 bits) against the definition at sampled
    positions and as a round trip."""
    from za_amd import synth
    lg = 27
    n = 1 << lg
    seed_b, seed_s = 2727, 127
    d_b = synth.synth_bases_g1(ctx, n, seed_b)
    d_s = synth.synth_scalars(ctx, n, seed_s)
    sc = ctx.d2h(d_s, 32 * n)
    e = co.dot_u64(sc, synth.splitmix64_array(seed_b, n)) % bn.R
    want = co.g1_mul(bn.g1_to_mont_bytes(bn.G1_GEN), e)
    # (i) the schedule the library picks by itself: at this size the per-window plan (c = 22, twelve windows of 2^21
    # buckets, 4.0e8 entries per part) is within 8 % of the table's cost, so no table is built
    H = ctx.msm_bases_g1_dev(d_b, n, table_budget_gb=150.0)
    assert ctx.msm_bases_info(H)["n"] == n
    got = ctx.msm_resident_dev(H, d_s, n, dense_hint=True)
    assert got[:64] == want and got[64:] == bn.le32(bn.mont(1, bn.Q))
    ctx.msm_bases_free(H)
    # (i') ONE job over all 2^27 points (the per-call entry point: no parts, no table) with the staged three-level partition
    # forced -- its level-1 tile is capped by the kernel's 64 KB of dynamic LDS (8192-point tiles would ask for 67.7 KB)
    import za_amd
    ctx.set_option(za_amd.OPT_SORT_STAGED, 2)
    try:
        assert ctx.msm_g1_dev(d_b, d_s, n)[:64] == want
    finally:
        ctx.set_option(za_amd.OPT_SORT_STAGED, -1)
    # (ii) the window table built regardless (negative budget): 12 rows x 2^27 entries = 103 GB, table indices to 1.6e9
    H = ctx.msm_bases_g1_dev(d_b, n, table_budget_gb=-150.0)
    ctx.free(d_b)
    info = ctx.msm_bases_info(H)
    assert info["n"] == n and info["table_bytes"] >= 10 * 64 * n, info
    assert ctx.msm_resident_dev(H, d_s, n, dense_hint=True)[:64] == want
    ctx.msm_bases_free(H)
    # ---- the transform: x = the uploaded scalars read as Montgomery residues (all below 2^253 < r)
    sample = [0, 1, 2, n // 2 - 1, n // 2, n - 1, 0x5a5a5a5 % n]
    # (i) forward then inverse is the identity, for the plain and the coset pair
    ctx.ntt_dev(d_s, lg, False, False)
    fwd = {k: bn.unmont(bn.from_le32(bytes(ctx.d2h(d_s + 32 * k, 32))), bn.R) for k in sample}
    ctx.ntt_dev(d_s, lg, True, False)
    ctx.ntt_dev(d_s, lg, False, True)
    ctx.ntt_dev(d_s, lg, True, True)
    assert bytes(ctx.d2h(d_s, 32 * n)) == bytes(sc)
    # (ii) sum_k X[k] = m x[0], and X[0] = sum_i x[i], X[m/2] = sum_i (-1)^i x[i] (omega^(m/2) = -1): the definition, exactly
    xs = np.frombuffer(sc, dtype="<u8").reshape(n, 4)
    def total(rows):   # sum of 256-bit little-endian integers, limb by limb in Python integers
        t = 0
        for limb in range(4):
            lo = rows[:, limb] & np.uint64(0xFFFFFFFF)
            hi = rows[:, limb] >> np.uint64(32)
            t += (int(lo.sum(dtype=np.uint64)) + (int(hi.sum(dtype=np.uint64)) << 32)) << (64 * limb)
        return t
    rinv = bn.inv(bn.MONT_R, bn.R)
    s_all, s_even = total(xs), total(xs[0::2])
    assert fwd[0] == s_all * rinv % bn.R
    assert fwd[n // 2] == (2 * s_even - s_all) * rinv % bn.R
    # (iii) a sparse vector: X[k] = x_a omega^(a k) + x_b omega^(b k) at the sampled k
    ctx.free(d_s)
    omega = pow(bn.FR_ROOT_OF_UNITY, 1 << (bn.FR_S - lg), bn.R)
    pa, pb, va, vb = 3, n - 5, 0x1234567, 0x7654321
    d = ctx.alloc(32 * n)
    z = np.zeros(1 << 26, dtype=np.uint8)
    for off in range(0, 32 * n, z.size):
        ctx.h2d(d + off, z)
    ctx.h2d(d + 32 * pa, np.frombuffer(bn.le32(bn.mont(va, bn.R)), dtype=np.uint8))
    ctx.h2d(d + 32 * pb, np.frombuffer(bn.le32(bn.mont(vb, bn.R)), dtype=np.uint8))
    ctx.ntt_dev(d, lg, False, False)
    for k in sample:
        got_k = bn.unmont(bn.from_le32(bytes(ctx.d2h(d + 32 * k, 32))), bn.R)
        assert got_k == (va * pow(omega, pa * k, bn.R) + vb * pow(omega, pb * k, bn.R)) % bn.R, k
    ctx.free(d)


def test_config4_on_its_own_circuit_sha256_merkle_depth_32(ctx):
    """BASELINE config 4 as named: a depth-32 SHA-256 Merkle inclusion proof on the bit-level circuit of
    za_amd/workloads.py (1.48 M constraints, 2^21 domain; root and witness asserted against hashlib by the generator).
    zg_prove, zg_prove_packed and the C oracle give the same bytes; the pairing check takes the hashlib root."""
    import za_amd
    from za_amd import workloads
    ni, na, nc, rp, col, cf, wit, info = workloads.sha256_merkle(32)
    R = ctx.r1cs_load(ni, na, nc, rp, col, cf)
    assert R.log_m == 21 and R.check(wit) is None
    P, pbytes = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172)
    r, s = 0x1234567, 0x7654321
    got = ctx.prove(P, R, wit, r, s)
    packed = za_amd.pack_witness(wit)
    assert packed[1].size < 200 and ctx.prove_packed(P, R, packed, r, s) == got     # everything but a few words is a bit
    csr = co.CsrR1CS(ni, na, nc, rp, col, cf)
    assert co.prove(download_params(ctx, P), csr, wit, r, s, threads=THREADS) == got
    root = info["root"]
    assert za_amd.verify(pbytes, got, [int.from_bytes(root[:16], "big"), int.from_bytes(root[16:], "big")]) is True
    P.close(); R.close()
