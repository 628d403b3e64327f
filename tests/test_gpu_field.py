"""Device field arithmetic (csrc/fp.h on gfx950) against the golden vectors -- bit-exact."""
import ctypes as C

import numpy as np
import pytest

from oracle import bn254 as bn
from tests.util import from_mont_bytes, golden, mont_bytes

pytestmark = pytest.mark.gpu


def test_device_field_ops_match_golden(ctx):
    fv = golden("field_vectors.json")
    for which, (name, mod) in enumerate((("fq", bn.Q), ("fr", bn.R))):
        a = [int(v[0], 16) for v in fv[name]]
        b = [int(v[1], 16) for v in fv[name]]
        A = np.frombuffer(mont_bytes(a, mod), dtype=np.uint8)
        B = np.frombuffer(mont_bytes(b, mod), dtype=np.uint8)
        for op, col in ((0, 2), (1, 3), (2, 4), (3, 5)):
            out = np.zeros(A.size, dtype=np.uint8)
            ctx._check(ctx.lib.zg_selftest_field_dev(ctx.h, C.c_int(which), C.c_int(op), A.ctypes.data_as(C.c_void_p),
                                                     B.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                                     C.c_size_t(len(a))))
            assert from_mont_bytes(out, mod) == [int(v[col], 16) for v in fv[name]], (name, op)


def test_device_equals_host_on_random_batch(ctx):
    rng = np.random.default_rng(3)
    n = 4096
    for which, mod in ((0, bn.Q), (1, bn.R)):
        vals = [int.from_bytes(rng.bytes(32), "little") % mod for _ in range(2 * n)]
        A = np.frombuffer(b"".join(bn.le32(v) for v in vals[:n]), dtype=np.uint8)
        B = np.frombuffer(b"".join(bn.le32(v) for v in vals[n:]), dtype=np.uint8)
        for op in (0, 1, 2, 7):
            h = np.zeros(A.size, dtype=np.uint8)
            d = np.zeros(A.size, dtype=np.uint8)
            ctx.lib.zg_selftest_field_host(C.c_int(which), C.c_int(op), A.ctypes.data_as(C.c_void_p),
                                           B.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), C.c_size_t(n))
            ctx._check(ctx.lib.zg_selftest_field_dev(ctx.h, C.c_int(which), C.c_int(op), A.ctypes.data_as(C.c_void_p),
                                                     B.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                                     C.c_size_t(n)))
            assert bytes(h) == bytes(d)
            if op == 0:
                got = [bn.from_le32(bytes(d[32 * i:32 * i + 32])) for i in range(16)]
                exp = [vals[i] * vals[n + i] * bn.inv(bn.MONT_R, mod) % mod for i in range(16)]
                assert got == exp
