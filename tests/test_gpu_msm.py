"""GPU multi-scalar multiplication (zg_msm_*) against the golden KATs, the C oracle and, at full
size, against the discrete-log identity  sum s_i (k_i G) = (sum s_i k_i) G.  Bit-exact."""
import numpy as np
import pytest

import za_amd

from oracle import bn254 as bn
from oracle import coracle as co
from tests.util import golden, jac_to_affine_bytes

pytestmark = pytest.mark.gpu


def test_golden_kats_g1_g2_with_density(ctx):
    kat = golden("msm_kat.json")
    for rec in kat["g1"]:
        got = ctx.msm_g1(bytes.fromhex(rec["bases"]), bytes.fromhex(rec["scalars"]), rec["density"])
        assert jac_to_affine_bytes(got, 32).hex() == rec["expect"], rec["n"]
    for rec in kat["g2"]:
        got = ctx.msm_g2(bytes.fromhex(rec["bases"]), bytes.fromhex(rec["scalars"]), rec["density"])
        assert jac_to_affine_bytes(got, 64).hex() == rec["expect"], rec["n"]


def test_query_size_mismatch_is_an_error(ctx):
    import za_amd
    rec = golden("msm_kat.json")["g1"][2]
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.msm_g1(bytes.fromhex(rec["bases"])[:-64], bytes.fromhex(rec["scalars"]), rec["density"])
    assert e.value.code == za_amd.ZG_E_QUERY_SIZE


def test_empty_msm_is_identity(ctx):
    out = ctx.msm_g1(b"", b"")
    assert out == b"\0" * 96


def _bases_from_gpu(ctx, n, seed, g2=False):
    from za_amd import synth
    d = synth.synth_bases_g2(ctx, n, seed) if g2 else synth.synth_bases_g1(ctx, n, seed)
    host = ctx.d2h(d, n * (128 if g2 else 64))
    return d, host


def test_synthetic_bases_have_the_advertised_discrete_logs(ctx):
    from za_amd import synth
    n = 64
    d, host = _bases_from_gpu(ctx, n, 77)
    d2, host2 = _bases_from_gpu(ctx, 8, 78, g2=True)
    g1 = bn.g1_to_mont_bytes(bn.G1_GEN)
    g2 = bn.g2_to_mont_bytes(bn.G2_GEN)
    for i in (0, 1, 17, 63):
        assert bytes(host[64 * i:64 * i + 64]) == co.g1_mul(g1, synth.splitmix64_at(77, i))
    for i in (0, 7):
        assert bytes(host2[128 * i:128 * i + 128]) == co.g2_mul(g2, synth.splitmix64_at(78, i))
    ctx.free(d); ctx.free(d2)


SCALAR_PATTERNS = ["uniform", "bits97", "all_ones", "all_equal", "minus_one", "small16", "one_nonzero"]


@pytest.mark.parametrize("pattern", SCALAR_PATTERNS)
@pytest.mark.parametrize("n", [1000, 1 << 14])
def test_g1_against_c_oracle(ctx, n, pattern):
    rng = np.random.default_rng(hash((n, pattern)) % (1 << 32))
    d, host = _bases_from_gpu(ctx, n, 1234 + n)
    def rnd():
        return int.from_bytes(rng.bytes(32), "little") % bn.R
    if pattern == "uniform":
        sc = [rnd() for _ in range(n)]
    elif pattern == "bits97":
        sc = [int(rng.integers(0, 2)) if rng.random() < 0.97 else rnd() for _ in range(n)]
    elif pattern == "all_ones":
        sc = [1] * n
    elif pattern == "all_equal":
        v = rnd()
        sc = [v] * n
    elif pattern == "minus_one":
        sc = [bn.R - 1 if i % 2 else bn.R - 2 for i in range(n)]
    elif pattern == "small16":
        sc = [int(rng.integers(0, 1 << 16)) for _ in range(n)]
    else:
        sc = [0] * n
        sc[n // 2] = rnd()
    sb = co.witness_bytes(sc)
    exp = co.msm_g1(host, sb, threads=8)
    got = ctx.msm_g1(host, sb)
    assert jac_to_affine_bytes(got, 32) == exp
    ctx.free(d)


@pytest.mark.parametrize("pattern", ["uniform", "bits97", "all_equal"])
def test_g1_with_every_sort_segment_through_the_lds_stage(ctx, pattern, opts):
    """k_part_local's staged scatter (normally for segments of 16 K entries and more: 2^24 points and up, where it is
    covered by the full-size tests) forced onto every segment: plain windows, window table and resident bases."""
    opts(za_amd.OPT_PART_STAGE_MIN, 0)
    n = (1 << 15) + 77
    rng = np.random.default_rng(77)
    d, host = _bases_from_gpu(ctx, n, 4242)
    v = int.from_bytes(rng.bytes(32), "little") % bn.R
    if pattern == "uniform":
        sc = [int.from_bytes(rng.bytes(32), "little") % bn.R for _ in range(n)]
    elif pattern == "bits97":
        sc = [int(rng.integers(0, 2)) if rng.random() < 0.97 else int.from_bytes(rng.bytes(32), "little") % bn.R for _ in range(n)]
    else:
        sc = [v] * n          # one bucket per window takes everything: segments of very different lengths
    sb = co.witness_bytes(sc)
    exp = co.msm_g1(host, sb, threads=8)
    assert jac_to_affine_bytes(ctx.msm_g1(host, sb), 32) == exp
    H = ctx.msm_bases_g1_dev(d, n, table_budget_gb=1.0)
    assert jac_to_affine_bytes(ctx.msm_resident(H, np.frombuffer(sb, dtype=np.uint8), n, dense_hint=(pattern == "uniform")), 32) == exp
    ctx.msm_bases_free(H)
    ctx.free(d)


@pytest.mark.parametrize("n", [1000, 1 << 13])
def test_g2_against_c_oracle_with_density(ctx, n):
    rng = np.random.default_rng(n)
    dens = (rng.random(n + n // 3) < 0.75).astype(np.uint8)
    nb = int(dens.sum())
    d, host = _bases_from_gpu(ctx, nb, 4321, g2=True)
    sc = [int(rng.integers(0, 2)) if rng.random() < 0.5 else int.from_bytes(rng.bytes(32), "little") % bn.R
          for _ in range(len(dens))]
    sb = co.witness_bytes(sc)
    exp = co.msm_g2(host, sb, dens, threads=8)
    got = ctx.msm_g2(host, sb, dens)
    assert jac_to_affine_bytes(got, 64) == exp
    ctx.free(d)


def _dlog_expected(seed, n, scal_seed):
    """(sum_i s_i k_i) mod r for the synthetic bases / scalars, computed with Python integers."""
    from za_amd import synth
    k = synth.splitmix64_array(seed, n)
    w = synth.splitmix64_array(scal_seed, 4 * n).reshape(n, 4)
    w[:, 3] &= np.uint64((0x1FFFFFFF << 32) | 0xFFFFFFFF)
    tot = 0
    kk = k.tolist()
    ww = w.tolist()
    for i in range(n):
        s = ww[i][0] | (ww[i][1] << 64) | (ww[i][2] << 128) | (ww[i][3] << 192)
        tot += s * kk[i]
    return tot % bn.R


@pytest.mark.parametrize("log_n", [16, 22])
def test_full_size_g1_msm_discrete_log_identity(ctx, log_n):
    """BASELINE config size (2^22 points): the result must equal (sum s_i k_i) * G."""
    from za_amd import synth
    n = 1 << log_n
    d_b = synth.synth_bases_g1(ctx, n, 9001)
    d_s = synth.synth_scalars(ctx, n, 9002)
    got = ctx.msm_g1_dev(d_b, d_s, n)
    e = _dlog_expected(9001, n, 9002)
    exp = co.g1_mul(bn.g1_to_mont_bytes(bn.G1_GEN), e)
    assert jac_to_affine_bytes(got, 32) == exp
    ctx.free(d_b); ctx.free(d_s)


def test_large_g2_msm_discrete_log_identity(ctx):
    from za_amd import synth
    n = 1 << 18
    d_b = synth.synth_bases_g2(ctx, n, 9101)
    d_s = synth.synth_scalars(ctx, n, 9102)
    got = ctx.msm_g2_dev(d_b, d_s, n)
    e = _dlog_expected(9101, n, 9102)
    exp = co.g2_mul(bn.g2_to_mont_bytes(bn.G2_GEN), e)
    assert jac_to_affine_bytes(got, 64) == exp
    ctx.free(d_b); ctx.free(d_s)


def test_msm_linearity_with_index_indirection(ctx):
    """MSM(bases, s[idx]) through d_scalar_index equals the MSM over the gathered scalars."""
    from za_amd import synth
    n, ns = 5000, 8000
    rng = np.random.default_rng(5)
    d_b = synth.synth_bases_g1(ctx, n, 31)
    d_s = synth.synth_scalars(ctx, ns, 32)
    idx = np.sort(rng.choice(ns, size=n, replace=False)).astype(np.uint32)
    d_i = ctx.upload(idx.view(np.uint8))
    got = ctx.msm_g1_dev(d_b, d_s, n, d_i)
    scal = ctx.d2h(d_s, ns * 32).reshape(ns, 32)
    bases = ctx.d2h(d_b, n * 64)
    exp = co.msm_g1(bases, scal[idx].reshape(-1), threads=8)
    assert jac_to_affine_bytes(got, 32) == exp
    for p in (d_b, d_s, d_i):
        ctx.free(p)


@pytest.mark.parametrize("group,log_n,share", [("g1", 21, 1.0), ("g1", 21, 0.25), ("g1", 20, 0.6), ("g2", 18, 1.0), ("g2", 18, 0.3)])
def test_skewed_scalars_one_value_in_most_positions(ctx, group, log_n, share):
    """A witness it did not choose: `share` of the scalars hold ONE non-bit constant (65537 * 2^40 + 3: every digit position
    of those scalars lands in one bucket per window -- up to two million entries), the rest are uniform.  The sort's last
    level (wave-aggregated counters for a segment that is mostly one bucket), the item map (a big bucket's items written by
    its wave) and the heavy-bucket fold (units of 1024 items, the last workgroup to arrive folds the units' sums) must give
    (sum s_i k_i) G -- per call and through resident bases with a window table."""
    from za_amd import synth
    n = 1 << log_n
    g2 = group == "g2"
    d_b = (synth.synth_bases_g2 if g2 else synth.synth_bases_g1)(ctx, n, 9301)
    d_s = synth.synth_scalars(ctx, n, 9302)
    sc = np.frombuffer(ctx.d2h(d_s, 32 * n), dtype=np.uint8).reshape(n, 32).copy()
    rng = np.random.default_rng(log_n * 100 + int(share * 10))
    idx = np.arange(n) if share >= 1.0 else rng.choice(n, size=int(n * share), replace=False)
    v = (65537 << 40) + 3
    sc[idx, :] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
    k = synth.splitmix64_array(9301, n).tolist()
    words = sc.view("<u8").reshape(n, 4).tolist()
    tot = 0
    for i in range(n):
        wi = words[i]
        tot += (wi[0] | (wi[1] << 64) | (wi[2] << 128) | (wi[3] << 192)) * k[i]
    gen = bn.g2_to_mont_bytes(bn.G2_GEN) if g2 else bn.g1_to_mont_bytes(bn.G1_GEN)
    exp = (co.g2_mul if g2 else co.g1_mul)(gen, tot % bn.R)
    fb = 64 if g2 else 32
    ctx.h2d(d_s, sc.reshape(-1))
    got = (ctx.msm_g2_dev if g2 else ctx.msm_g1_dev)(d_b, d_s, n)
    assert jac_to_affine_bytes(got, fb) == exp
    if not g2:
        H = ctx.msm_bases_g1_dev(d_b, n, table_budget_gb=4.0)
        assert jac_to_affine_bytes(ctx.msm_resident_dev(H, d_s, n, False), fb) == exp
        assert jac_to_affine_bytes(ctx.msm_resident(H, sc.reshape(-1), n, True), fb) == exp
        ctx.msm_bases_free(H)
    ctx.free(d_b); ctx.free(d_s)
