"""Helpers shared by the tests (conversions between golden-vector text and the ABI's byte forms)."""
import ctypes as C
import json
import os

import numpy as np

from oracle import bn254 as bn
from oracle import coracle as co
from oracle import groth16 as g

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def mont_bytes(values, mod):
    return b"".join(bn.le32(bn.mont(v, mod)) for v in values)


def canon_bytes(values):
    return b"".join(bn.le32(v) for v in values)


def from_mont_bytes(buf, mod):
    buf = bytes(buf)
    return [bn.unmont(bn.from_le32(buf[i:i + 32]), mod) for i in range(0, len(buf), 32)]


def from_canon_bytes(buf):
    buf = bytes(buf)
    return [bn.from_le32(buf[i:i + 32]) for i in range(0, len(buf), 32)]


def jac_to_affine_bytes(j, fb):
    """ABI Jacobian (X||Y||Z, Z in {0,1}) -> affine bytes with the all-zero identity convention."""
    j = bytes(j)
    if j[2 * fb:3 * fb] == b"\0" * fb:
        return b"\0" * (2 * fb)
    return j[:2 * fb]


def r1cs_from_fixture(fx):
    cons = [tuple([(v, int(co_, 16)) for v, co_ in lc] for lc in c) for c in fx["constraints"]]
    return g.R1CS(fx["n_inputs"], fx["n_aux"], cons)


def download_params(ctx, P):
    """Pull a (synthetic) proving key back from the GPU as the oracle's MontParams."""
    vk5 = np.zeros(448, dtype=np.uint8)
    ctx._check(ctx.lib.zg_params_download(P.h, C.c_int(-1), vk5.ctypes.data_as(C.c_void_p), C.c_size_t(448)))
    arrs = []
    for which, elem in ((0, 64), (1, 64), (2, 64), (3, 64), (4, 128)):
        tot = P.query_total(which)
        a = np.zeros(max(tot, 1) * elem, dtype=np.uint8)
        ctx._check(ctx.lib.zg_params_download(P.h, C.c_int(which), a.ctypes.data_as(C.c_void_p), C.c_size_t(tot * elem)))
        arrs.append(a[: tot * elem])
    return co.MontParams(vk5[0:64].tobytes(), vk5[64:128].tobytes(), vk5[128:256].tobytes(), vk5[256:320].tobytes(),
                         vk5[320:448].tobytes(), *arrs)


def table_infos(P):
    """(kind, bytes) per query (h, l, a, b_g1, b_g2).  A key whose C rows are folded into the l multiexp (ZG_OPT_C_FOLD) has
    no table for the plain l query: the folded query's table (which = 5) stands in for it."""
    k = [P.table_info(i) for i in range(5)]
    if k[1][0] == 0 and P.table_info(5)[0]:
        k[1] = P.table_info(5)
    return k


def table_kinds(P):
    return [k[0] for k in table_infos(P)]
