"""The host pairing zg_verify runs (csrc/pairing.hip, round 6: Miller loops on the twist in lockstep, sparse lines, Frobenius
by constants, the final exponentiation by powers of u) against the DEFINITION the library keeps (points untwisted into
Fq12, plain powers q and (q^12 - 1) / r), and against bilinearity.  CPU only."""
import ctypes as C
import time

import numpy as np

from oracle import bn254 as bn


def _prod(lib, pairs, slow):
    g1 = np.frombuffer(b"".join(bn.g1_to_mont_bytes(p) for p, _ in pairs), dtype=np.uint8)
    g2 = np.frombuffer(b"".join(bn.g2_to_mont_bytes(q) for _, q in pairs), dtype=np.uint8)
    gt = np.zeros(384, dtype=np.uint8)
    one = C.c_int(-1)
    rc = lib.zg_selftest_pairing_product(g1.ctypes.data_as(C.c_void_p), g2.ctypes.data_as(C.c_void_p), C.c_size_t(len(pairs)),
                                         C.c_int(slow), gt.ctypes.data_as(C.c_void_p), C.byref(one))
    assert rc == 0
    return gt.tobytes(), bool(one.value)


def test_fast_pairing_equals_the_definition_and_is_bilinear(lib_built):
    G1, G2 = bn.G1_GEN, bn.G2_GEN
    a, b, c = 0x1234567890abcdef1234567, 0xfedcba9876543210fedcba98765, 77
    P, Q = bn.pt_mul(G1, a), bn.pt_mul(G2, b)
    t0 = time.time()
    fast, one_f = _prod(lib_built, [(P, Q)], 0)
    t_fast = time.time() - t0
    t0 = time.time()
    slow, one_s = _prod(lib_built, [(P, Q)], 1)
    t_slow = time.time() - t0
    print("one pairing: fast %.1f ms, definition %.1f ms" % (1e3 * t_fast, 1e3 * t_slow))
    assert fast == slow and not one_f and not one_s          # the same element of GT, bit for bit
    # e(aG, bH) e(-abG, H) = 1;  e(aG, bH) e(-(ab + 1) G, H) != 1;  four pairs as a verification has them
    for slow_flag in (0, 1):
        assert _prod(lib_built, [(P, Q), (bn.pt_neg(bn.pt_mul(G1, a * b)), G2)], slow_flag)[1]
        assert not _prod(lib_built, [(P, Q), (bn.pt_neg(bn.pt_mul(G1, a * b + 1)), G2)], slow_flag)[1]
    four = [(P, Q), (bn.pt_mul(G1, c), bn.pt_mul(G2, 5)), (bn.pt_neg(bn.pt_mul(G1, a * b + 5 * c - 9)), G2), (bn.pt_neg(bn.pt_mul(G1, 3)), bn.pt_mul(G2, 3))]
    f4, one4 = _prod(lib_built, four, 0)
    assert one4 and f4 == _prod(lib_built, four, 1)[0]
    # the identity on either side contributes 1
    assert _prod(lib_built, [(None, Q)], 0)[1] and _prod(lib_built, [(P, None)], 0)[1]
    assert _prod(lib_built, [(P, Q), (None, G2)], 0)[0] == fast
