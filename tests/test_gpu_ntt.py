"""GPU Fr NTT and H pipeline against golden KATs, the C oracle, and size-independent properties."""
import numpy as np
import pytest

from oracle import bn254 as bn
from oracle import coracle as co
from tests.util import from_mont_bytes, golden, mont_bytes

pytestmark = pytest.mark.gpu

MODES = (("fft", False, False), ("ifft", True, False), ("coset_fft", False, True), ("icoset_fft", True, True))


def test_golden_ntt_kats(ctx):
    for rec in golden("ntt_kat.json"):
        lg = rec["log_m"]
        a = mont_bytes([int(x, 16) for x in rec["input"]], bn.R)
        for name, inv, cos in MODES:
            got = from_mont_bytes(ctx.ntt(a, lg, inverse=inv, coset=cos), bn.R)
            assert got == [int(x, 16) for x in rec[name]], (lg, name)


@pytest.mark.parametrize("lg", [3, 8, 9, 11, 12, 13, 17, 20])
def test_ntt_against_c_oracle(ctx, lg):
    rng = np.random.default_rng(lg)
    m = 1 << lg
    raw = np.frombuffer(rng.bytes(32 * m), dtype=np.uint8).copy().reshape(m, 32)
    raw[:, 31] &= 0x1F   # < 2^253 < r, a valid residue (read as Montgomery form of something)
    a = raw.reshape(-1)
    for mode, (name, inv, cos) in enumerate(MODES):
        exp = co.fft(a, lg, mode, threads=16)
        got = ctx.ntt(a, lg, inverse=inv, coset=cos)
        assert bytes(exp) == bytes(got), (lg, name)


def test_degree_too_large_is_an_error(ctx):
    import za_amd
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.ntt_dev(ctx.alloc(64), 28)
    assert e.value.code == za_amd.ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE


def test_full_size_round_trips_and_linearity(ctx):
    """2^22 (BASELINE config size): ifft(fft(x)) = x, icoset(coset(x)) = x, fft(x + y) = fft(x) + fft(y)."""
    import ctypes as C
    lg = 22
    m = 1 << lg
    rng = np.random.default_rng(22)
    def vec():
        raw = np.frombuffer(rng.bytes(32 * m), dtype=np.uint8).copy().reshape(m, 32)
        raw[:, 31] &= 0x0F
        return raw.reshape(-1)
    x, y = vec(), vec()
    d = ctx.upload(x)
    ctx.ntt_dev(d, lg, False, False)
    fx = ctx.d2h(d, 32 * m)
    ctx.ntt_dev(d, lg, True, False)
    assert bytes(ctx.d2h(d, 32 * m)) == bytes(x)
    ctx.ntt_dev(d, lg, False, True)
    ctx.ntt_dev(d, lg, True, True)
    assert bytes(ctx.d2h(d, 32 * m)) == bytes(x)
    fy = ctx.ntt(y, lg)
    # x + y (both < 2^252, so the plain sum is still < r) -- add with the library's own field code
    s = np.zeros_like(x)
    ctx.lib.zg_selftest_field_host(C.c_int(1), C.c_int(1), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                                   s.ctypes.data_as(C.c_void_p), C.c_size_t(m))
    fs = ctx.ntt(s, lg)
    fsum = np.zeros_like(x)
    ctx.lib.zg_selftest_field_host(C.c_int(1), C.c_int(1), fx.ctypes.data_as(C.c_void_p), fy.ctypes.data_as(C.c_void_p),
                                   fsum.ctypes.data_as(C.c_void_p), C.c_size_t(m))
    assert bytes(fs) == bytes(fsum)
    ctx.free(d)


@pytest.mark.parametrize("name", ["proof_factor.json", "proof_chain7.json"])
def test_h_scalars_match_golden(ctx, name):
    fx = golden(name)
    a, b, c = (mont_bytes([int(x, 16) for x in fx[k]], bn.R) for k in ("a", "b", "c"))
    rows = len(fx["a"])
    lg = 0
    while (1 << lg) < rows:
        lg += 1
    h = ctx.h_scalars(a, b, c, rows, lg)
    got = [bn.from_le32(bytes(h[32 * i:32 * i + 32])) for i in range((1 << lg) - 1)]
    assert got == [int(x, 16) for x in fx["h"]]


def test_h_scalars_against_c_oracle_2p16(ctx):
    from za_amd import synth
    sr = synth.SynthR1CS(16, 2, 0.5, 99)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    a, b, c, *_ = co.r1cs_eval(csr, sr.witness, threads=8)
    rows = sr.n_constraints + sr.n_inputs
    exp = co.h_coeffs(a, b, c, rows, 16, threads=16)
    got = ctx.h_scalars(a, b, c, rows, 16)
    assert bytes(exp) == bytes(got)
    sr.close()
