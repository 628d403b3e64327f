"""ctypes access to the synthetic-workload generators (include/za_amd_synth.h).

These stand in for the za compiler (R1CS + witness) and bellman's parameter generator in bench.py
and the full-size tests; they are not part of the drop-in boundary.
"""
import ctypes as C

import numpy as np

from . import Context, Params, ZaAmdError, load_library

SYNTH_SYMBOLS = ["zg_synth_r1cs_new", "zg_synth_r1cs_info", "zg_synth_ptr", "zg_synth_free",
                 "zg_synth_bases_g1_dev", "zg_synth_bases_g2_dev", "zg_synth_params_dev", "zg_synth_dlog",
                 "zg_synth_scalars_dev", "zg_params_download", "zg_params_shard",
                 "zg_selftest_field_host", "zg_selftest_field_dev", "zg_selftest_curve_host", "zg_selftest_curve_dev", "zg_params_inspect",
                 "zg_selftest_pack_mt", "zg_selftest_pack_split", "zg_synth_pk_container_write", "zg_synth_binwitness_write",
                 "zg_selftest_pairing_product"]

MASK64 = (1 << 64) - 1


def splitmix64_at(seed, i):
    """Python twin of zg_synth_dlog (used by tests to recompute discrete logs)."""
    z = (seed + (i + 1) * 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def splitmix64_array(seed, n, start=0):
    """Vectorised splitmix64_at(seed, start + i) for i < n as uint64."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed & MASK64) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class SynthR1CS:
    """Host-resident synthetic R1CS + witness (arrays are views into C++-owned memory)."""

    def __init__(self, log_m, n_inputs=2, bit_fraction=0.97, seed=0x5A):
        lib = load_library()
        lib.zg_synth_ptr.restype = C.c_void_p
        lib.zg_synth_free.restype = None
        h = C.c_void_p()
        rc = lib.zg_synth_r1cs_new(C.c_uint32(log_m), C.c_uint32(n_inputs), C.c_double(bit_fraction),
                                   C.c_uint64(seed), C.byref(h))
        if rc:
            raise ZaAmdError(rc, lib.zg_strerror(rc).decode())
        self.lib, self.h, self.log_m = lib, h, log_m
        info = (C.c_uint64 * 6)()
        lib.zg_synth_r1cs_info(h, info)
        self.n_inputs, self.n_aux, self.n_constraints = int(info[0]), int(info[1]), int(info[2])
        self.nnz = [int(info[3 + k]) for k in range(3)]

        def view(which, dtype, count):
            p = lib.zg_synth_ptr(h, C.c_int(which))
            if count == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(count,))
        self.row_ptr = [view(k, np.uint64, self.n_constraints + 1) for k in range(3)]
        self.col = [view(3 + k, np.uint32, self.nnz[k]) for k in range(3)]
        self.coeff = [view(6 + k, np.uint8, self.nnz[k] * 32) for k in range(3)]
        self.witness = view(9, np.uint8, (self.n_inputs + self.n_aux) * 32)

    def load(self, ctx: Context):
        return ctx.r1cs_load(self.n_inputs, self.n_aux, self.n_constraints, self.row_ptr, self.col, self.coeff)

    def write_proving_key(self, path, params_bytes):
        """The za proving.key container `za setup` would write for this circuit around the Parameters body `params_bytes`."""
        pb = np.frombuffer(params_bytes, dtype=np.uint8)
        rp = (C.c_void_p * 3)(*[a.ctypes.data for a in self.row_ptr])
        cl = (C.c_void_p * 3)(*[a.ctypes.data if a.size else None for a in self.col])
        cf = (C.c_void_p * 3)(*[a.ctypes.data if a.size else None for a in self.coeff])
        rc = self.lib.zg_synth_pk_container_write(str(path).encode(), C.c_uint32(self.n_inputs), C.c_uint32(self.n_aux),
                                                  C.c_uint64(self.n_constraints), rp, cl, cf, pb.ctypes.data_as(C.c_void_p),
                                                  C.c_size_t(pb.size))
        if rc:
            raise ZaAmdError(rc, "writing %s" % path)

    def write_binwitness(self, path):
        """The evaluator's .binwitness dump of this circuit's witness."""
        rc = self.lib.zg_synth_binwitness_write(str(path).encode(), self.witness.ctypes.data_as(C.c_void_p),
                                                C.c_size_t(self.n_inputs + self.n_aux))
        if rc:
            raise ZaAmdError(rc, "writing %s" % path)

    def close(self):
        if self.h:
            self.row_ptr = self.col = self.coeff = self.witness = None
            self.lib.zg_synth_free(self.h)
            self.h = None


def synth_bases_g1(ctx: Context, n, seed):
    d = ctx.alloc(max(n, 1) * 64)
    ctx._check(ctx.lib.zg_synth_bases_g1_dev(ctx.h, C.c_void_p(d), C.c_size_t(n), C.c_uint64(seed)))
    ctx.sync()
    return d


def synth_bases_g2(ctx: Context, n, seed):
    d = ctx.alloc(max(n, 1) * 128)
    ctx._check(ctx.lib.zg_synth_bases_g2_dev(ctx.h, C.c_void_p(d), C.c_size_t(n), C.c_uint64(seed)))
    ctx.sync()
    return d


def synth_scalars(ctx: Context, n, seed):
    d = ctx.alloc(max(n, 1) * 32)
    ctx._check(ctx.lib.zg_synth_scalars_dev(ctx.h, C.c_void_p(d), C.c_size_t(n), C.c_uint64(seed)))
    ctx.sync()
    return d


def params_shard(ctx: Context, full: Params, rank, world):
    h = C.c_void_p()
    ctx._check(ctx.lib.zg_params_shard(ctx.h, full.h, C.c_int(rank), C.c_int(world), C.byref(h)))
    return Params(ctx, h)


def synth_params(ctx: Context, n_h, n_l, n_a, n_b, seed, rank=0, world=1):
    h = C.c_void_p()
    ctx._check(ctx.lib.zg_synth_params_dev(ctx.h, C.c_uint64(n_h), C.c_uint64(n_l), C.c_uint64(n_a), C.c_uint64(n_b),
                                           C.c_uint64(seed), C.c_int(rank), C.c_int(world), C.byref(h)))
    return Params(ctx, h)
