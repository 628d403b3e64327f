"""Parameters::read(reader, checked) on the device (za_amd/csrc/keyload.hip; reference call site
prover/src/groth16/format.rs:285, paid by every prove() of the reference: helper.rs:91-98).

The host parser (formats.hip, reached without a GPU through zg_params_inspect) and the device decoder must accept and
refuse the same keys with the same messages, and a key decoded on the device must be the key the oracle reads."""
import ctypes as C
import struct
import time

import numpy as np
import pytest

from oracle import bn254 as bn
from oracle import groth16 as g
from tests import golden_1k as k1
from tests.util import download_params

pytestmark = pytest.mark.gpu


def _inspect(lib, buf, checked=1):
    b = np.frombuffer(bytes(buf), dtype=np.uint8)
    counts = (C.c_uint64 * 7)()
    err = C.create_string_buffer(256)
    rc = lib.zg_params_inspect(b.ctypes.data_as(C.c_void_p), C.c_size_t(b.size), C.c_int(checked), counts, err, C.c_size_t(256))
    return rc, err.value.decode()


def _device(ctx, buf, checked=True, rank=0, world=1):
    import za_amd
    try:
        P, used = ctx.params_load(bytes(buf), checked=checked, rank=rank, world=world)
    except za_amd.ZaAmdError as e:
        return e.code, ctx.lib.zg_last_error(ctx.h).decode()
    assert used == len(buf)
    P.close()
    return 0, ""


def _layout(buf):
    """offset of the first point and count for h, l, a, b_g1, b_g2 in a Parameters body"""
    pos = 64 + 64 + 128 + 128 + 64 + 128
    n_ic = struct.unpack_from(">I", buf, pos)[0]
    pos += 4 + 64 * n_ic
    out = []
    for which in range(5):
        n = struct.unpack_from(">I", buf, pos)[0]
        pos += 4
        out.append((pos, n))
        pos += n * (128 if which == 4 else 64)
    assert pos == len(buf)
    return out


def _small_order_twist_point():
    cof = 2 * bn.Q - bn.R
    seed = 1
    while True:
        T = bn.g2_twist_point(seed)
        S = bn.pt_mul(T, bn.R * (cof // 10069), order=1 << 600)
        if S is not None:
            return S
        seed += 1


def test_device_decoder_reads_the_key_the_oracle_reads(ctx):
    pbytes = k1.chain10_params()
    params, used = g.params_read(pbytes)
    assert used == len(pbytes)
    for checked in (True, False):
        P, used = ctx.params_load(pbytes, checked=checked)
        assert used == len(pbytes)
        mp = download_params(ctx, P)
        for got, want, to_bytes in ((mp.h, params.h, bn.g1_to_mont_bytes), (mp.l, params.l, bn.g1_to_mont_bytes),
                                    (mp.a, params.a, bn.g1_to_mont_bytes), (mp.b_g1, params.b_g1, bn.g1_to_mont_bytes),
                                    (mp.b_g2, params.b_g2, bn.g2_to_mont_bytes)):
            assert bytes(got) == b"".join(to_bytes(p) for p in want)
        assert P.write() == pbytes                    # Parameters::write(Parameters::read(x)) == x
        P.close()


def test_host_and_device_validators_agree_on_mutated_keys(ctx, lib_built):
    import za_amd
    pbytes = k1.chain10_params()
    lay = _layout(pbytes)
    twist = g.g2_uncompressed(bn.g2_twist_point(5))
    small = g.g2_uncompressed(_small_order_twist_point())
    cases = []     # (name, mutation(bytearray), message expected when checked, message expected when unchecked)

    def case(name, fn, checked_msg, unchecked_msg=None):
        cases.append((name, fn, checked_msg, unchecked_msg))

    names = ["h", "l", "a", "b_g1", "b_g2"]
    for which, (off, n) in enumerate(lay):
        elem = 128 if which == 4 else 64
        at = off + elem * (n // 2)                       # a point in the middle of the query
        last = off + elem * (n - 1)
        q = names[which]

        def flip_y(b, at=at, elem=elem):
            b[at + elem - 1] ^= 1
        case(q + " off the curve", flip_y, q + ": point is not on the curve")

        def ident(b, at=last, elem=elem):
            b[at:at + elem] = b"\x40" + b"\0" * (elem - 1)
        case(q + " identity inside the query", ident, q + ": point at infinity", q + ": point at infinity")

        def compressed(b, at=at):
            b[at] |= 0x80
        case(q + " compressed flag", compressed, q + ": unexpected compression mode", q + ": unexpected compression mode")

        def bad_inf(b, at=at, elem=elem):
            b[at] |= 0x40
        case(q + " identity flag over coordinates", bad_inf, q + ": unexpected information in infinity encoding",
             q + ": unexpected information in infinity encoding")
        for coord in range(elem // 32):
            def big(b, at=at + 32 * coord):
                b[at:at + 32] = (bn.Q + (0 if coord else 5)).to_bytes(32, "big")
            case("%s coordinate %d >= q" % (q, coord), big, q + ": coordinate is not in the field", q + ": coordinate is not in the field")
    off, n = lay[4]

    def off_subgroup(b, at=off + 128 * 3):
        b[at:at + 128] = twist
    case("b_g2 twist point outside the r-torsion", off_subgroup, "b_g2: point is not in the correct subgroup")

    def small_order(b, at=off + 128 * (n - 2)):
        b[at:at + 128] = small
    case("b_g2 point of order 10069", small_order, "b_g2: point is not in the correct subgroup")

    def two_defects(b):
        o_l, n_l = lay[1]
        b[o_l + 64 * 7 + 63] ^= 1                      # l[7] off the curve
        b[o_l + 64 * 3] |= 0x80                        # l[3] compressed: the lower index is the one reported
        b[lay[0][0] + 64 * (lay[0][1] - 1) + 63] ^= 1  # and h before l
    case("first defect in file order", two_defects, "h: point is not on the curve", "l: unexpected compression mode")

    def vk_off_curve(b):
        b[40] ^= 1
    case("vk.alpha_g1 off the curve", vk_off_curve, "vk.alpha_g1: point is not on the curve")
    assert _device(ctx, pbytes) == (0, "") and _inspect(lib_built, pbytes)[0] == 0
    for name, fn, checked_msg, unchecked_msg in cases:
        bad = bytearray(pbytes)
        fn(bad)
        for checked, want in ((1, checked_msg), (0, unchecked_msg)):
            rc_h, msg_h = _inspect(lib_built, bad, checked)
            rc_d, msg_d = _device(ctx, bad, bool(checked))
            assert (rc_h, msg_h) == (rc_d, msg_d), (name, checked, msg_h, msg_d)
            if want is None:
                assert rc_d == 0, (name, checked, msg_d)
            else:
                assert rc_d == za_amd.ZG_E_IO and msg_d == want, (name, checked, msg_d)
    # a sharded load validates the whole key: a defect outside the rank's own chunk is still refused
    bad = bytearray(pbytes)
    bad[lay[4][0] + 128 * (lay[4][1] - 1) + 127] ^= 1   # the last b_g2 point, rank 0 of 3 does not hold it
    assert _device(ctx, bad, True, rank=0, world=3) == (za_amd.ZG_E_IO, "b_g2: point is not on the curve")
    assert _device(ctx, bad, False, rank=0, world=3) == (0, "")
    assert _device(ctx, pbytes, True, rank=2, world=3) == (0, "")
    # truncation is the host walk's business and reads the same
    for cut in (5, 64 * 3 + 1, len(pbytes) - lay[2][0] + 9):
        assert _inspect(lib_built, pbytes[:-cut]) == _device(ctx, pbytes[:-cut])


def test_checked_read_of_a_2p20_key_on_the_device(ctx, lib_built):
    """A real 2^20 key (GPU trapdoor set-up): the checked read accepts it, proves the same bytes as the key it was written
    from, and refuses one off-subgroup / one off-curve point deep inside; the time of the checked read is bounded."""
    import za_amd
    from za_amd import synth
    sr = synth.SynthR1CS(20, 2, 0.97, 0x5A)
    R = sr.load(ctx)
    P0, pbytes = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=True)
    lay = _layout(pbytes)
    assert lay[4][1] > 1 << 18
    ctx.set_option(za_amd.OPT_ONE_SHOT, 1)
    try:
        want = ctx.prove(P0, R, sr.witness, 5, 7)
        P0.close()
        t0 = time.time()
        Pu, _ = ctx.params_load(pbytes, checked=False)
        t_unchecked = time.time() - t0
        Pu.close()
        t0 = time.time()
        P, _ = ctx.params_load(pbytes, checked=True)
        t_checked = time.time() - t0
        print("2^20 key, %d bytes, %d b_g2 points: unchecked %.3f s, checked %.3f s" % (len(pbytes), lay[4][1], t_unchecked, t_checked))
        assert ctx.prove(P, R, sr.witness, 5, 7) == want
        P.close()
        assert t_checked - t_unchecked < 1.0      # (the host validator: ~0.57 ms per G2 point = minutes)
    finally:
        ctx.set_option(za_amd.OPT_ONE_SHOT, 0)
    bad = bytearray(pbytes)
    at = lay[4][0] + 128 * (lay[4][1] - 12345)
    bad[at:at + 128] = g.g2_uncompressed(bn.g2_twist_point(9))
    assert _device(ctx, bad) == (za_amd.ZG_E_IO, "b_g2: point is not in the correct subgroup")
    assert _device(ctx, bad, checked=False) == (0, "")
    bad = bytearray(pbytes)
    bad[lay[0][0] + 64 * 777777 + 63] ^= 1
    assert _device(ctx, bad) == (za_amd.ZG_E_IO, "h: point is not on the curve")
    R.close(); sr.close()
