"""ctypes front-end of oracle/_build/liboracle.so -- TEST ORACLE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import bn254 as bn

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    src = [os.path.join(_HERE, "c", f) for f in ("oracle.c", "field.h", "ec_tmpl.h")]
    if (not force and os.path.exists(_SO)
            and all(not os.path.exists(s) or os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "_build/liboracle.so"])
    return _SO


class _R1CS(C.Structure):
    _fields_ = [("n_inputs", C.c_uint32), ("n_aux", C.c_uint32), ("n_constraints", C.c_uint64),
                ("row_ptr", C.c_void_p * 3), ("col", C.c_void_p * 3), ("coeff", C.c_void_p * 3)]


class _Params(C.Structure):
    _fields_ = [("alpha_g1", C.c_void_p), ("beta_g1", C.c_void_p), ("beta_g2", C.c_void_p),
                ("delta_g1", C.c_void_p), ("delta_g2", C.c_void_p),
                ("h", C.c_void_p), ("n_h", C.c_uint64), ("l", C.c_void_p), ("n_l", C.c_uint64),
                ("a", C.c_void_p), ("n_a", C.c_uint64), ("b_g1", C.c_void_p), ("n_b_g1", C.c_uint64),
                ("b_g2", C.c_void_p), ("n_b_g2", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_init()
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def _buf(b):
    """bytes/ndarray -> (keepalive ndarray, void*)."""
    if isinstance(b, (bytes, bytearray)):
        b = np.frombuffer(bytes(b), dtype=np.uint8)
    return _u8(b)


def msm_g1(bases, scalars, density=None, threads=1):
    """bases: n_bases x 64 B (Montgomery LE affine), scalars: n x 32 B canonical LE, density: n bytes or None."""
    kb, pb = _buf(bases)
    ks, ps = _buf(scalars)
    n = ks.size // 32
    pd = None
    if density is not None:
        kd, pd = _buf(density)
    out = np.zeros(64, dtype=np.uint8)
    rc = lib().orc_msm_g1(pb, C.c_size_t(kb.size // 64), ps, pd, C.c_size_t(n), C.c_int(threads),
                          out.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("orc_msm_g1 rc=%d" % rc)
    return out.tobytes()


def msm_g2(bases, scalars, density=None, threads=1):
    kb, pb = _buf(bases)
    ks, ps = _buf(scalars)
    n = ks.size // 32
    pd = None
    if density is not None:
        kd, pd = _buf(density)
    out = np.zeros(128, dtype=np.uint8)
    rc = lib().orc_msm_g2(pb, C.c_size_t(kb.size // 128), ps, pd, C.c_size_t(n), C.c_int(threads),
                          out.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("orc_msm_g2 rc=%d" % rc)
    return out.tobytes()


def dot_u64(scalars, k64):
    """sum_i scalars[i] * k64[i] as a Python int (scalars: n x 32 B canonical LE; k64: uint64 array)."""
    sa, sp = _buf(scalars)
    ka = np.ascontiguousarray(k64, dtype=np.uint64)
    n = ka.size
    assert sa.size >= 32 * n
    out = np.zeros(48, dtype=np.uint8)
    rc = lib().orc_dot_u64(sp, ka.ctypes.data_as(C.c_void_p), C.c_size_t(n), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return int.from_bytes(out.tobytes(), "little")


def g1_mul(p, k):
    out = np.zeros(64, dtype=np.uint8)
    kp, pp = _buf(p)
    kk, pk = _buf(bn.le32(k % bn.R))
    lib().orc_g1_mul(pp, pk, out.ctypes.data_as(C.c_void_p))
    return out.tobytes()


def g2_mul(p, k):
    out = np.zeros(128, dtype=np.uint8)
    kp, pp = _buf(p)
    kk, pk = _buf(bn.le32(k % bn.R))
    lib().orc_g2_mul(pp, pk, out.ctypes.data_as(C.c_void_p))
    return out.tobytes()


def g1_add(p, q):
    out = np.zeros(64, dtype=np.uint8)
    kp, pp = _buf(p)
    kq, pq = _buf(q)
    lib().orc_g1_add(pp, pq, out.ctypes.data_as(C.c_void_p))
    return out.tobytes()


def g2_add(p, q):
    out = np.zeros(128, dtype=np.uint8)
    kp, pp = _buf(p)
    kq, pq = _buf(q)
    lib().orc_g2_add(pp, pq, out.ctypes.data_as(C.c_void_p))
    return out.tobytes()


def field_mul(which, a, b):
    ka, pa = _buf(a)
    kb, pb = _buf(b)
    out = np.zeros(ka.size, dtype=np.uint8)
    lib().orc_field_mul(C.c_int(which), pa, pb, out.ctypes.data_as(C.c_void_p), C.c_size_t(ka.size // 32))
    return out


def to_mont(which, a):
    ka, pa = _buf(a)
    out = np.zeros(ka.size, dtype=np.uint8)
    lib().orc_to_mont(C.c_int(which), pa, out.ctypes.data_as(C.c_void_p), C.c_size_t(ka.size // 32))
    return out


def from_mont(which, a):
    ka, pa = _buf(a)
    out = np.zeros(ka.size, dtype=np.uint8)
    lib().orc_from_mont(C.c_int(which), pa, out.ctypes.data_as(C.c_void_p), C.c_size_t(ka.size // 32))
    return out


FFT, IFFT, COSET_FFT, ICOSET_FFT = 0, 1, 2, 3


def fft(data, log_m, mode, threads=1):
    """data: m x 32 B Montgomery LE; returns a transformed copy."""
    a = np.array(np.frombuffer(bytes(data), dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data,
                 dtype=np.uint8, copy=True)
    assert a.size == 32 << log_m
    rc = lib().orc_fft(a.ctypes.data_as(C.c_void_p), C.c_uint(log_m), C.c_int(mode), C.c_int(threads))
    if rc:
        raise RuntimeError("orc_fft rc=%d" % rc)
    return a


def h_coeffs(a, b, c, rows, log_m, threads=1):
    ka, pa = _buf(a)
    kb, pb = _buf(b)
    kc, pc = _buf(c)
    out = np.zeros(32 * ((1 << log_m) - 1), dtype=np.uint8)
    rc = lib().orc_h(pa, pb, pc, C.c_size_t(rows), C.c_uint(log_m), C.c_int(threads),
                     out.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("orc_h rc=%d" % rc)
    return out


class CsrR1CS:
    """CSR arrays shared by the oracle and (via the C ABI) the product.
    row_ptr[k]: uint64[n+1], col[k]: uint32[nnz], coeff[k]: uint8[nnz*32] canonical LE; k = A, B, C."""

    def __init__(self, n_inputs, n_aux, n_constraints, row_ptr, col, coeff):
        self.n_inputs, self.n_aux, self.n_constraints = n_inputs, n_aux, n_constraints
        self.row_ptr = [np.ascontiguousarray(x, dtype=np.uint64) for x in row_ptr]
        self.col = [np.ascontiguousarray(x, dtype=np.uint32) for x in col]
        self.coeff = [np.ascontiguousarray(x, dtype=np.uint8) for x in coeff]

    @staticmethod
    def from_r1cs(r1cs):
        row_ptr, col, coeff = [], [], []
        for k in range(3):
            rp, cl, cf = [0], [], []
            for cons in r1cs.constraints:
                for v, co in cons[k]:
                    cl.append(v)
                    cf.append(bn.le32(co % bn.R))
                rp.append(len(cl))
            row_ptr.append(np.array(rp, dtype=np.uint64))
            col.append(np.array(cl, dtype=np.uint32))
            coeff.append(np.frombuffer(b"".join(cf), dtype=np.uint8) if cf else np.zeros(0, dtype=np.uint8))
        return CsrR1CS(r1cs.num_inputs, r1cs.num_aux, len(r1cs.constraints), row_ptr, col, coeff)

    def c_struct(self):
        s = _R1CS()
        s.n_inputs, s.n_aux, s.n_constraints = self.n_inputs, self.n_aux, self.n_constraints
        for k in range(3):
            s.row_ptr[k] = self.row_ptr[k].ctypes.data
            s.col[k] = self.col[k].ctypes.data
            s.coeff[k] = self.coeff[k].ctypes.data
        return s


def r1cs_eval(csr, witness, threads=1):
    kw, pw = _buf(witness)
    rows = csr.n_constraints + csr.n_inputs
    a, b, c = (np.zeros(rows * 32, dtype=np.uint8) for _ in range(3))
    ad = np.zeros(max(csr.n_aux, 1), dtype=np.uint8)
    bi = np.zeros(max(csr.n_inputs, 1), dtype=np.uint8)
    bd = np.zeros(max(csr.n_aux, 1), dtype=np.uint8)
    s = csr.c_struct()
    rc = lib().orc_r1cs_eval(C.byref(s), pw, C.c_int(threads), a.ctypes.data_as(C.c_void_p),
                             b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p),
                             ad.ctypes.data_as(C.c_void_p), bi.ctypes.data_as(C.c_void_p),
                             bd.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("orc_r1cs_eval rc=%d" % rc)
    return a, b, c, ad[:csr.n_aux], bi[:csr.n_inputs], bd[:csr.n_aux]


class MontParams:
    """Proving key as flat Montgomery-LE byte arrays (the in-memory layout both the oracle and the
    product's device upload use)."""

    def __init__(self, alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, h, l, a, b_g1, b_g2):
        self.alpha_g1, self.beta_g1, self.beta_g2 = alpha_g1, beta_g1, beta_g2
        self.delta_g1, self.delta_g2 = delta_g1, delta_g2
        self.h, self.l, self.a, self.b_g1, self.b_g2 = (np.ascontiguousarray(x, dtype=np.uint8)
                                                        for x in (h, l, a, b_g1, b_g2))

    @staticmethod
    def from_params(p):
        def cat1(v):
            return np.frombuffer(b"".join(bn.g1_to_mont_bytes(x) for x in v), dtype=np.uint8)

        def cat2(v):
            return np.frombuffer(b"".join(bn.g2_to_mont_bytes(x) for x in v), dtype=np.uint8)
        vk = p.vk
        return MontParams(bn.g1_to_mont_bytes(vk.alpha_g1), bn.g1_to_mont_bytes(vk.beta_g1),
                          bn.g2_to_mont_bytes(vk.beta_g2), bn.g1_to_mont_bytes(vk.delta_g1),
                          bn.g2_to_mont_bytes(vk.delta_g2), cat1(p.h), cat1(p.l), cat1(p.a), cat1(p.b_g1),
                          cat2(p.b_g2))

    def c_struct(self):
        self._keep = [np.frombuffer(x, dtype=np.uint8) for x in
                      (self.alpha_g1, self.beta_g1, self.beta_g2, self.delta_g1, self.delta_g2)]
        s = _Params()
        s.alpha_g1, s.beta_g1, s.beta_g2, s.delta_g1, s.delta_g2 = (k.ctypes.data for k in self._keep)
        s.h, s.n_h = self.h.ctypes.data, self.h.size // 64
        s.l, s.n_l = self.l.ctypes.data, self.l.size // 64
        s.a, s.n_a = self.a.ctypes.data, self.a.size // 64
        s.b_g1, s.n_b_g1 = self.b_g1.ctypes.data, self.b_g1.size // 64
        s.b_g2, s.n_b_g2 = self.b_g2.ctypes.data, self.b_g2.size // 128
        return s


def prove(mparams, csr, witness, r, s, threads=1, timings=False):
    """witness: bytes/ndarray of (n_inputs+n_aux) x 32 B canonical LE.  Returns the 256-byte proof
    (A.x, A.y, B.x.c0, B.x.c1, B.y.c0, B.y.c1, C.x, C.y canonical big-endian)."""
    kw, pw = _buf(witness)
    ps = mparams.c_struct()
    rs = csr.c_struct()
    out = np.zeros(256, dtype=np.uint8)
    kr, pr = _buf(bn.le32(r % bn.R))
    ks, pss = _buf(bn.le32(s % bn.R))
    tm = (C.c_double * 4)()
    rc = lib().orc_prove(C.byref(ps), C.byref(rs), pw, pr, pss, C.c_int(threads),
                         out.ctypes.data_as(C.c_void_p), tm)
    if rc:
        raise RuntimeError("orc_prove rc=%d" % rc)
    if timings:
        return out.tobytes(), list(tm)
    return out.tobytes()


def witness_bytes(w):
    return b"".join(bn.le32(x % bn.R) for x in w)
