"""za_amd -- MI355X-native Groth16/BN254 prover hot path for adria0/za.

The product is `libza_amd.so` (HIP kernels + the C ABI in include/za_amd.h).  This package is the
thin ctypes binding that tests, bench.py and a Python host use; it mirrors the shape of the
reference's Python binding (binding/python3/native/src/lib.rs: setup / prove / verify taking
strings and returning strings) one level lower, at the create_proof boundary.

There is no CPU fallback: every compute entry point needs a GPU and raises ZaAmdError otherwise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZA_AMD_LIB: another build of the same library (tools/asan_host.sh: host code under ASan / UBSan)
LIB_PATH = os.environ.get("ZA_AMD_LIB") or os.path.join(_HERE, "libza_amd.so")
SYNTH_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libza_amd_synth.so")

ZG_OK = 0
ZG_E_ASSIGNMENT_MISSING = 1
ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE = 4
ZG_E_UNEXPECTED_IDENTITY = 5
ZG_E_IO = 6
ZG_E_MALFORMED_VERIFYING_KEY = 7
ZG_E_UNCONSTRAINED_VARIABLE = 8
ZG_E_BAD_FORMAT = 20
ZG_E_QUERY_SIZE = 21
ZG_E_INVALID_ARG = 22
ZG_E_BUFFER_TOO_SMALL = 23
ZG_E_HIP = 30
ZG_E_NO_DEVICE = 31
ZG_PARTIALS_BYTES = 576
ZG_MASK_H_EVAL = 0x20   # zg_prove_partials_ex: d_h_shard holds evaluation-form scalars

# every symbol include/za_amd.h declares (tests/test_abi.py checks the library exports them all)
ABI_SYMBOLS = [
    "zg_init", "zg_free", "zg_strerror", "zg_last_error", "zg_set_stream", "zg_set_option", "zg_sync",
    "zg_dev_alloc", "zg_dev_free", "zg_h2d", "zg_d2h",
    "zg_msm_g1", "zg_msm_g2", "zg_msm_g1_dev", "zg_msm_g2_dev",
    "zg_msm_bases_g1", "zg_msm_bases_g1_dev", "zg_msm_bases_g2", "zg_msm_bases_free", "zg_msm_bases_info",
    "zg_msm_resident", "zg_msm_resident_dev", "zg_host_alloc", "zg_host_free", "zg_g1_sum",
    "zg_ntt_fr", "zg_ntt_fr_dev", "zg_h_scalars",
    "zg_r1cs_load", "zg_r1cs_free", "zg_r1cs_info", "zg_r1cs_eval",
    "zg_params_load", "zg_params_from_mont", "zg_params_free", "zg_params_write", "zg_params_write_vk", "zg_pk_params_offset",
    "zg_pk_load", "zg_witness_from_binwitness", "zg_setup", "zg_setup_random",
    "zg_prove", "zg_prove_dev", "zg_prove_random", "zg_prove_packed", "zg_witness_pack",
    "zg_prove_h_dev", "zg_prove_h_eval_dev", "zg_prove_h_eval_enqueue_dev", "zg_prove_h_eval_enqueue_fold_dev", "zg_params_h_eval_info", "zg_witness_to_device", "zg_prove_h_enqueue_dev", "zg_prove_partials_dev", "zg_prove_partials_ex", "zg_prove_combine", "zg_params_shard_range", "zg_params_query_total",
    "zg_params_prepare", "zg_params_table_info",
    "zg_proof_json", "zg_verify", "zg_vk_json", "zg_last_timings", "zg_last_load_timings", "zg_last_stats",
    "zg_vk_from_json", "zg_proof_from_json", "zg_verifier_solidity", "zg_last_error_thread", "zg_r1cs_check",
    "zg_profile_enable", "zg_profile_reset", "zg_profile_get", "zg_profile_get_units", "zg_profile_name",
]


class ZaAmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("za_amd error %d: %s" % (code, msg))
        self.code = code


_lib = None


class _Libs:
    """The product library, plus -- looked up only when a name is not the product's -- the scaffolding library
    libza_amd_synth.so (synthetic workloads and self-test hooks of include/za_amd_synth.h: tests and bench.py only)."""

    def __init__(self, main, synth_path):
        self.__dict__["main"] = main
        self.__dict__["_synth_path"] = synth_path
        self.__dict__["_synth"] = None

    @property
    def synth(self):
        if self.__dict__["_synth"] is None:
            if not os.path.exists(self._synth_path):
                raise ZaAmdError(-1, "libza_amd_synth.so is missing: build it with `python za_amd/build.py`")
            self.__dict__["_synth"] = C.CDLL(self._synth_path)
        return self.__dict__["_synth"]

    def __getattr__(self, name):
        try:
            return getattr(self.__dict__["main"], name)
        except AttributeError:
            return getattr(self.synth, name)


def load_library():
    """Load libza_amd.so; raises if it has not been built (python -m za_amd.build / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZaAmdError(-1, "libza_amd.so is missing: build it with `python za_amd/build.py` "
                                 "(there is no Python/CPU fallback)")
        # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HSA runtimes cannot
        # share a process, and whichever loads first wins the SONAME: load torch's first so that
        # libza_amd.so binds to the same runtime torch.distributed (RCCL) uses.  Not needed for C callers.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        main = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        lib = _Libs(main, SYNTH_LIB_PATH)
        lib.zg_strerror.restype = C.c_char_p
        lib.zg_last_error.restype = C.c_char_p
        lib.zg_last_error_thread.restype = C.c_char_p
        lib.zg_free.restype = None
        lib.zg_r1cs_free.restype = None
        lib.zg_params_free.restype = None
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(x):
    if isinstance(x, (bytes, bytearray, memoryview)):
        x = np.frombuffer(bytes(x), dtype=np.uint8)
    return np.ascontiguousarray(x, dtype=np.uint8)


def density_bitmap(flags):
    """bool/0-1 sequence -> LSB-first u64 bitmap (the C ABI's density_bits form)."""
    f = np.asarray(flags, dtype=np.uint8) != 0
    pad = (-len(f)) % 64
    if pad or len(f) == 0:
        f = np.concatenate([f, np.zeros(pad if len(f) else 64, dtype=bool)])
    return np.packbits(f, bitorder="little").view("<u8").copy()


# zg_set_option (include/za_amd.h)
OPT_ONE_SHOT, OPT_PACK, OPT_PACK_THREADS, OPT_PACK_MIN_LOG_N, OPT_PACK_PIN, OPT_H_EVAL, OPT_TABLES, OPT_TABLES_MAX_MB, \
    OPT_PACK_BUDGET_PERMILLE, OPT_PACK_RAW_PERMILLE, OPT_COMBINE_EARLY = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
# schedule switches (tests, fuzzer, A/B tools)
OPT_SMALL_LOG_N, OPT_SMALL_W_LOG_N, OPT_SORT_STAGED, OPT_PART_STAGE_MIN, OPT_SUBSET_W, OPT_COMPACT_TABLE, OPT_H_PARTS, \
    OPT_H_PIPE, OPT_SHARE_SORT, OPT_NTT_BATCH_LOG_M, OPT_C_FOLD, OPT_C_FOLD_MIN_LOG_M, OPT_UPLOAD_GATES = 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24


def _options_from_env():
    """ZA_AMD_OPTIONS="14=3,15=0": options this BINDING applies to every context it creates (whole test-suite runs with a
    schedule forced, e.g. the staged digit sort for every job).  The library itself reads no such variable."""
    out = []
    for item in os.environ.get("ZA_AMD_OPTIONS", "").split(","):
        if item.strip():
            k, v = item.split("=")
            out.append((int(k), int(v)))
    return out


class Context:
    """One per process and GPU (zg_ctx)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.zg_init(C.c_int(device), C.byref(h))
        if rc:
            raise ZaAmdError(rc, self.lib.zg_strerror(rc).decode())
        self.h = h
        self.device = device
        for opt, val in _options_from_env():
            self.set_option(opt, val)

    def close(self):
        if getattr(self, "h", None):
            self.lib.zg_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            detail = self.lib.zg_last_error(self.h).decode()
            raise ZaAmdError(rc, "%s: %s" % (self.lib.zg_strerror(rc).decode(), detail))

    # ---- plumbing
    def set_stream(self, stream_handle):
        self._check(self.lib.zg_set_stream(self.h, C.c_void_p(stream_handle)))

    def set_one_shot(self, on=True):
        """ZG_OPT_ONE_SHOT: no per-key precomputation (a key used for one proof)."""
        self._check(self.lib.zg_set_option(self.h, C.c_int(1), C.c_int64(1 if on else 0)))

    def set_option(self, option, value):
        """zg_set_option: `option` one of the OPT_* numbers below (include/za_amd.h ZG_OPT_*)."""
        self._check(self.lib.zg_set_option(self.h, C.c_int(option), C.c_int64(int(value))))

    def sync(self):
        self._check(self.lib.zg_sync(self.h))

    def alloc(self, nbytes):
        p = C.c_void_p()
        self._check(self.lib.zg_dev_alloc(self.h, C.c_size_t(nbytes), C.byref(p)))
        return p.value

    def free(self, dptr):
        self._check(self.lib.zg_dev_free(self.h, C.c_void_p(dptr)))

    def h2d(self, dptr, data):
        a = _u8(data)
        self._check(self.lib.zg_h2d(self.h, C.c_void_p(dptr), _ptr(a), C.c_size_t(a.size)))

    def d2h(self, dptr, nbytes):
        out = np.empty(nbytes, dtype=np.uint8)
        self._check(self.lib.zg_d2h(self.h, _ptr(out), C.c_void_p(dptr), C.c_size_t(nbytes)))
        return out

    def upload(self, data):
        a = _u8(data)
        d = self.alloc(a.size)
        self.h2d(d, a)
        return d

    # ---- MSM
    def msm_g1(self, bases, scalars, density=None):
        b, s = _u8(bases), _u8(scalars)
        out = np.zeros(96, dtype=np.uint8)
        dm = None if density is None else density_bitmap(density)
        self._check(self.lib.zg_msm_g1(self.h, _ptr(b), C.c_size_t(b.size // 64), _ptr(s),
                                       None if dm is None else _ptr(dm), C.c_size_t(s.size // 32), _ptr(out)))
        return out.tobytes()

    def msm_g2(self, bases, scalars, density=None):
        b, s = _u8(bases), _u8(scalars)
        out = np.zeros(192, dtype=np.uint8)
        dm = None if density is None else density_bitmap(density)
        self._check(self.lib.zg_msm_g2(self.h, _ptr(b), C.c_size_t(b.size // 128), _ptr(s),
                                       None if dm is None else _ptr(dm), C.c_size_t(s.size // 32), _ptr(out)))
        return out.tobytes()

    def msm_g1_dev(self, d_bases, d_scalars, n, d_index=None):
        out = np.zeros(96, dtype=np.uint8)
        self._check(self.lib.zg_msm_g1_dev(self.h, C.c_void_p(d_bases), C.c_void_p(d_scalars),
                                           C.c_void_p(d_index), C.c_size_t(n), _ptr(out)))
        return out.tobytes()

    def msm_g2_dev(self, d_bases, d_scalars, n, d_index=None):
        out = np.zeros(192, dtype=np.uint8)
        self._check(self.lib.zg_msm_g2_dev(self.h, C.c_void_p(d_bases), C.c_void_p(d_scalars),
                                           C.c_void_p(d_index), C.c_size_t(n), _ptr(out)))
        return out.tobytes()

    # ---- resident bases
    def msm_bases_g1_dev(self, d_bases, n, table_budget_gb=0.0):
        h = C.c_void_p()
        self._check(self.lib.zg_msm_bases_g1_dev(self.h, C.c_void_p(d_bases), C.c_size_t(n), C.c_double(table_budget_gb),
                                                 C.byref(h)))
        return h

    def msm_bases(self, bases, g2=False, table_budget_gb=0.0):
        b = _u8(bases)
        h = C.c_void_p()
        fn = self.lib.zg_msm_bases_g2 if g2 else self.lib.zg_msm_bases_g1
        self._check(fn(self.h, _ptr(b), C.c_size_t(b.size // (128 if g2 else 64)), C.c_double(table_budget_gb), C.byref(h)))
        return h

    def msm_bases_info(self, handle):
        info = (C.c_uint64 * 4)()
        self._check(self.lib.zg_msm_bases_info(handle, info))
        return dict(zip(("n", "table_bytes", "c", "nwin"), [int(x) for x in info]))

    def msm_bases_free(self, handle):
        self.lib.zg_msm_bases_free.restype = None
        self.lib.zg_msm_bases_free(handle)

    def msm_resident(self, handle, scalars, n, dense_hint=False, g2=False):
        """scalars: host bytes / uint8 array, or an int host address (e.g. of a pinned buffer)."""
        out = np.zeros(192 if g2 else 96, dtype=np.uint8)
        ptr = C.c_void_p(scalars) if isinstance(scalars, int) else _ptr(_u8(scalars))
        self._check(self.lib.zg_msm_resident(self.h, handle, ptr, C.c_size_t(n), C.c_int(int(dense_hint)), _ptr(out)))
        return out.tobytes()

    def msm_resident_dev(self, handle, d_scalars, n, dense_hint=False, g2=False):
        out = np.zeros(192 if g2 else 96, dtype=np.uint8)
        self._check(self.lib.zg_msm_resident_dev(self.h, handle, C.c_void_p(d_scalars), C.c_size_t(n),
                                                 C.c_int(int(dense_hint)), _ptr(out)))
        return out.tobytes()

    def host_alloc(self, nbytes):
        """Pinned host buffer as a numpy uint8 view (free with host_free(arr))."""
        p = C.c_void_p()
        self._check(self.lib.zg_host_alloc(self.h, C.c_size_t(nbytes), C.byref(p)))
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(nbytes, 1),))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._check(self.lib.zg_host_free(self.h, C.c_void_p(p)))

    # ---- NTT / H
    def ntt(self, data, log_m, inverse=False, coset=False):
        a = np.array(_u8(data), copy=True)
        assert a.size == 32 << log_m
        self._check(self.lib.zg_ntt_fr(self.h, _ptr(a), C.c_uint32(log_m), C.c_int(int(inverse)), C.c_int(int(coset))))
        return a

    def ntt_dev(self, dptr, log_m, inverse=False, coset=False):
        self._check(self.lib.zg_ntt_fr_dev(self.h, C.c_void_p(dptr), C.c_uint32(log_m), C.c_int(int(inverse)),
                                           C.c_int(int(coset))))

    def h_scalars(self, a, b, c, n_rows, log_m):
        a, b, c = _u8(a), _u8(b), _u8(c)
        out = np.zeros(32 * ((1 << log_m) - 1), dtype=np.uint8)
        self._check(self.lib.zg_h_scalars(self.h, _ptr(a), _ptr(b), _ptr(c), C.c_size_t(n_rows), C.c_uint32(log_m),
                                          _ptr(out)))
        return out

    # ---- R1CS / params / prove
    def r1cs_load(self, n_inputs, n_aux, n_constraints, row_ptr, col, coeff):
        rp = [np.ascontiguousarray(x, dtype=np.uint64) for x in row_ptr]
        cl = [np.ascontiguousarray(x, dtype=np.uint32) for x in col]
        cf = [_u8(x) for x in coeff]
        arr3 = C.c_void_p * 3
        h = C.c_void_p()
        self._check(self.lib.zg_r1cs_load(self.h, C.c_uint32(n_inputs), C.c_uint32(n_aux), C.c_uint64(n_constraints),
                                          arr3(*[x.ctypes.data for x in rp]), arr3(*[x.ctypes.data for x in cl]),
                                          arr3(*[x.ctypes.data for x in cf]), C.byref(h)))
        return R1CS(self, h)

    def params_from_mont(self, vk5, h, l, a, b_g1, b_g2, rank=0, world=1):
        vk5, h, l, a, b_g1, b_g2 = (_u8(x) for x in (vk5, h, l, a, b_g1, b_g2))
        assert vk5.size == 448
        hd = C.c_void_p()
        self._check(self.lib.zg_params_from_mont(
            self.h, _ptr(vk5), _ptr(h), C.c_size_t(h.size // 64), _ptr(l), C.c_size_t(l.size // 64),
            _ptr(a), C.c_size_t(a.size // 64), _ptr(b_g1), C.c_size_t(b_g1.size // 64),
            _ptr(b_g2), C.c_size_t(b_g2.size // 128), C.c_int(rank), C.c_int(world), C.byref(hd)))
        return Params(self, hd)

    def params_load(self, bellman_bytes, checked=True, rank=0, world=1):
        b = _u8(bellman_bytes)
        hd = C.c_void_p()
        used = C.c_size_t()
        self._check(self.lib.zg_params_load(self.h, _ptr(b), C.c_size_t(b.size), C.c_int(int(checked)), C.c_int(rank),
                                            C.c_int(world), C.byref(hd), C.byref(used)))
        return Params(self, hd), used.value

    def setup(self, r1cs, alpha, beta, gamma, delta, tau, want_bytes=True):
        """generate_parameters with explicit toxic waste -> (Params, bellman Parameters bytes or None)."""
        tox = _u8(b"".join(int(x).to_bytes(32, "little") for x in (alpha, beta, gamma, delta, tau)))
        ph = C.c_void_p()
        need = C.c_size_t()
        if not want_bytes:
            self._check(self.lib.zg_setup(self.h, r1cs.h, _ptr(tox), C.byref(ph), None, C.c_size_t(0), None))
            return Params(self, ph), None
        m = 1 << r1cs.log_m
        cap = 1024 + 64 * (r1cs.n_inputs + m + 3 * (r1cs.n_inputs + r1cs.n_aux)) + 128 * (r1cs.n_inputs + r1cs.n_aux)
        buf = np.zeros(cap, dtype=np.uint8)
        self._check(self.lib.zg_setup(self.h, r1cs.h, _ptr(tox), C.byref(ph), _ptr(buf), C.c_size_t(cap), C.byref(need)))
        return Params(self, ph), buf[:need.value].tobytes()

    def pk_load(self, pk_bytes, checked=True, rank=0, world=1):
        """A whole za `proving.key` -> (R1CS, Params)."""
        b = _u8(pk_bytes)
        rh, ph = C.c_void_p(), C.c_void_p()
        self._check(self.lib.zg_pk_load(self.h, _ptr(b), C.c_size_t(b.size), C.c_int(int(checked)), C.c_int(rank),
                                        C.c_int(world), C.byref(rh), C.byref(ph)))
        return R1CS(self, rh), Params(self, ph)

    def witness_from_binwitness(self, r1cs, binwitness):
        b = _u8(binwitness)
        out = np.zeros(32 * (r1cs.n_inputs + r1cs.n_aux), dtype=np.uint8)
        self._check(self.lib.zg_witness_from_binwitness(self.h, r1cs.h, _ptr(b), C.c_size_t(b.size), _ptr(out),
                                                        C.c_size_t(out.size)))
        return out

    def prove(self, params, r1cs, witness, r, s):
        """witness: (n_inputs+n_aux) x 32 B canonical LE; r, s: ints.  Returns the 256-byte proof."""
        w = _u8(witness)
        out = np.zeros(256, dtype=np.uint8)
        rb, sb = _u8(int(r).to_bytes(32, "little")), _u8(int(s).to_bytes(32, "little"))
        self._check(self.lib.zg_prove(self.h, params.h, r1cs.h, _ptr(w), _ptr(rb), _ptr(sb), _ptr(out)))
        return out.tobytes()

    def prove_packed(self, params, r1cs, packed, r, s):
        """zg_prove_packed; `packed` = pack_witness(...) (its arrays may live in pinned memory: host_alloc)."""
        classes, idx, vals = packed
        proof = np.zeros(256, dtype=np.uint8)
        n_others = int(idx.size)
        self._check(self.lib.zg_prove_packed(self.h, params.h, r1cs.h, _ptr(classes), _ptr(idx) if n_others else None,
                                             _ptr(vals) if n_others else None, C.c_size_t(n_others),
                                             _ptr(_u8(int(r).to_bytes(32, "little"))), _ptr(_u8(int(s).to_bytes(32, "little"))),
                                             _ptr(proof)))
        return proof.tobytes()

    def prove_dev(self, params, r1cs, d_witness, r, s):
        out = np.zeros(256, dtype=np.uint8)
        rb, sb = _u8(int(r).to_bytes(32, "little")), _u8(int(s).to_bytes(32, "little"))
        self._check(self.lib.zg_prove_dev(self.h, params.h, r1cs.h, C.c_void_p(d_witness), _ptr(rb), _ptr(sb),
                                          _ptr(out)))
        return out.tobytes()

    def prove_h_dev(self, r1cs, d_witness, d_h_out):
        self._check(self.lib.zg_prove_h_dev(self.h, r1cs.h, C.c_void_p(d_witness), C.c_void_p(d_h_out)))

    def prove_h_eval_dev(self, r1cs, d_witness, d_e_out):
        """H in evaluation form (the scalars of the key's transformed h bases); blocking."""
        self._check(self.lib.zg_prove_h_eval_dev(self.h, r1cs.h, C.c_void_p(d_witness), C.c_void_p(d_e_out)))

    def witness_to_device(self, witness):
        """zg_prove's hand-over of a host witness on its own (packed by host threads when it is mostly 0 / 1): returns
        (device address inside the context's buffer, bytes that crossed PCIe).  Queued, not waited for."""
        w = _u8(witness)
        if w.size % 32:
            raise ValueError("witness must be 32 bytes per value")
        d, sent = C.c_void_p(), C.c_size_t()
        self._check(self.lib.zg_witness_to_device(self.h, _ptr(w), C.c_size_t(w.size // 32), C.byref(d), C.byref(sent)))
        return d.value, sent.value

    def prove_h_eval_enqueue_dev(self, r1cs, d_witness, d_e_out):
        """Non-blocking zg_prove_h_eval_dev; call sync() before using d_e_out."""
        self._check(self.lib.zg_prove_h_eval_enqueue_dev(self.h, r1cs.h, C.c_void_p(d_witness), C.c_void_p(d_e_out)))

    def prove_h_eval_enqueue_fold_dev(self, params, r1cs, d_witness, d_e_out):
        """zg_prove_h_eval_enqueue_dev naming the key: four transforms where the key has the bases of the folded C rows (then
        the next prove_partials_dev with this key and the l query adds the rows' multiexp to its l partial).  Returns True
        if the pipeline was queued folded."""
        folded = C.c_int(0)
        self._check(self.lib.zg_prove_h_eval_enqueue_fold_dev(self.h, params.h, r1cs.h, C.c_void_p(d_witness), C.c_void_p(d_e_out),
                                                              C.byref(folded)))
        return bool(folded.value)

    def prove_h_enqueue_dev(self, r1cs, d_witness, d_h_out):
        """Non-blocking: queue R1CS evaluation + H; call sync() before using d_h_out."""
        self._check(self.lib.zg_prove_h_enqueue_dev(self.h, r1cs.h, C.c_void_p(d_witness), C.c_void_p(d_h_out)))

    def prove_partials_dev(self, params, r1cs, d_witness, d_h_shard, mask=0x1f):
        """mask bit 0 h, 1 l, 2 a, 3 b_g1, 4 b_g2; skipped sums come back as the identity."""
        out = np.zeros(ZG_PARTIALS_BYTES, dtype=np.uint8)
        self._check(self.lib.zg_prove_partials_ex(self.h, params.h, r1cs.h, C.c_void_p(d_witness),
                                                  C.c_void_p(d_h_shard), C.c_uint(mask), _ptr(out)))
        return out

    def prove_combine(self, params, partials_all, world, r, s):
        pa = _u8(partials_all)
        assert pa.size == world * ZG_PARTIALS_BYTES
        out = np.zeros(256, dtype=np.uint8)
        rb, sb = _u8(int(r).to_bytes(32, "little")), _u8(int(s).to_bytes(32, "little"))
        self._check(self.lib.zg_prove_combine(self.h, params.h, _ptr(pa), C.c_int(world), _ptr(rb), _ptr(sb),
                                              _ptr(out)))
        return out.tobytes()

    def profile_enable(self, level=2):
        """0 off, 1 = accumulation kernels only, 2 (or True) = every kernel class."""
        level = 2 if level is True else int(level)
        self._check(self.lib.zg_profile_enable(self.h, C.c_int(level)))

    def profile_reset(self):
        self._check(self.lib.zg_profile_reset(self.h))

    def profile_get(self):
        """{kernel class: (total ms, launches, algorithmic bytes, work units)} since the last reset."""
        self.lib.zg_profile_name.restype = C.c_char_p
        out = {}
        for i in range(10):
            if not self.lib.zg_profile_name(C.c_int(i)):   # (an older build of the library: ZA_AMD_LIB)
                break
            ms, n, b = C.c_double(), C.c_uint64(), C.c_uint64()
            self._check(self.lib.zg_profile_get(self.h, C.c_int(i), C.byref(ms), C.byref(n), C.byref(b)))
            u = C.c_uint64()
            self._check(self.lib.zg_profile_get_units(self.h, C.c_int(i), C.byref(u)))
            out[self.lib.zg_profile_name(C.c_int(i)).decode()] = (ms.value, n.value, b.value, u.value)
        return out

    def last_timings(self):
        t = (C.c_double * 6)()
        self._check(self.lib.zg_last_timings(self.h, t))
        return dict(zip(("upload_bytes", "r1cs_eval_ms", "h_ms", "msm_ms", "combine_ms", "total_ms"), list(t)))

    def last_load_timings(self):
        t = (C.c_double * 4)()
        self._check(self.lib.zg_last_load_timings(self.h, t))
        return dict(zip(("vk_and_walk_ms", "queries_device_ms", "constraints_ms", "r1cs_load_ms"), list(t)))

    def last_stats(self):
        """zg_last_stats: kernel launches of the last zg_prove* call, bit mask of the witness queries that shared a digit sort."""
        st = (C.c_uint64 * 4)()
        self._check(self.lib.zg_last_stats(self.h, st))
        return {"launches": int(st[0]), "shared_sort_mask": int(st[1]) & 0xff, "b_g1_behind_b_g2": bool(st[2]), "c_folded": bool(st[3]),
                "l_in_parts_behind_the_upload": bool(int(st[1]) & 0x100)}


class R1CS:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h
        info = (C.c_uint64 * 7)()
        ctx._check(ctx.lib.zg_r1cs_info(h, info))
        (self.n_inputs, self.n_aux, self.n_constraints, self.log_m, self.a_aux_total, self.b_input_total,
         self.b_aux_total) = list(info)

    def eval(self, witness):
        w = _u8(witness)
        rows = self.n_constraints + self.n_inputs
        a, b, c = (np.zeros(rows * 32, dtype=np.uint8) for _ in range(3))
        self.ctx._check(self.ctx.lib.zg_r1cs_eval(self.ctx.h, self.h, _ptr(w), _ptr(a), _ptr(b), _ptr(c)))
        return a, b, c

    def check(self, witness):
        """Index of the first unsatisfied constraint, or None (Constraints::satisfies_with_signals, prover.rs:155)."""
        w = _u8(witness)
        bad = C.c_uint64()
        self.ctx._check(self.ctx.lib.zg_r1cs_check(self.ctx.h, self.h, _ptr(w), C.byref(bad)))
        return None if bad.value == 0xFFFFFFFFFFFFFFFF else bad.value

    def close(self):
        if self.h:
            self.ctx.lib.zg_r1cs_free(self.h)
            self.h = None


class Params:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def shard_range(self, which):
        lo, hi = C.c_uint64(), C.c_uint64()
        self.ctx._check(self.ctx.lib.zg_params_shard_range(self.h, C.c_int(which), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def query_total(self, which):
        t = C.c_uint64()
        self.ctx._check(self.ctx.lib.zg_params_query_total(self.h, C.c_int(which), C.byref(t)))
        return t.value

    @property
    def total_h(self):
        return self.query_total(0)

    def h_eval_info(self):
        """(ready, log_m, lo, hi): this key's evaluation-form h bases (zg_params_h_eval_info)."""
        info = (C.c_uint64 * 4)()
        self.ctx._check(self.ctx.lib.zg_params_h_eval_info(self.h, info))
        return bool(info[0]), int(info[1]), int(info[2]), int(info[3])

    def write(self):
        """Parameters::write bytes of this resident key (needs a key that carries its verifying key)."""
        need = C.c_size_t()
        rc = self.ctx.lib.zg_params_write(self.ctx.h, self.h, None, C.c_size_t(0), C.byref(need))
        if rc != ZG_E_BUFFER_TOO_SMALL:
            self.ctx._check(rc)
        out = np.zeros(need.value, dtype=np.uint8)
        self.ctx._check(self.ctx.lib.zg_params_write(self.ctx.h, self.h, _ptr(out), C.c_size_t(out.size), C.byref(need)))
        return out.tobytes()

    def write_vk(self):
        """VerifyingKey::write bytes (the prefix of write())."""
        need = C.c_size_t()
        rc = self.ctx.lib.zg_params_write_vk(self.h, None, C.c_size_t(0), C.byref(need))
        if rc != ZG_E_BUFFER_TOO_SMALL:
            self.ctx._check(rc)
        out = np.zeros(need.value, dtype=np.uint8)
        self.ctx._check(self.ctx.lib.zg_params_write_vk(self.h, _ptr(out), C.c_size_t(out.size), C.byref(need)))
        return out.tobytes()

    def prepare(self, r1cs, d_witness=None):
        """Build the fixed-base window tables now (otherwise the first prove with this key does it); without a
        device witness only the h query's table."""
        self.ctx._check(self.ctx.lib.zg_params_prepare(self.ctx.h, self.h, r1cs.h, C.c_void_p(d_witness)))

    def table_info(self, which):
        """(kind, bytes): kind 0 none, 1 window table, 2 subset-sum table."""
        k, b = C.c_int(), C.c_uint64()
        self.ctx._check(self.ctx.lib.zg_params_table_info(self.h, C.c_int(which), C.byref(k), C.byref(b)))
        return k.value, b.value

    def close(self):
        if self.h:
            self.ctx.lib.zg_params_free(self.h)
            self.h = None


def pack_witness(witness):
    """zg_witness_pack: (classes u8[(n+3)//4], other_index u32[k], others u8[32 k]) of a 32-byte-per-value witness."""
    lib = load_library()
    w = _u8(witness)
    n = w.size // 32
    classes = np.zeros((n + 3) // 4, dtype=np.uint8)
    k = C.c_size_t()
    rc = lib.zg_witness_pack(_ptr(w), C.c_size_t(n), _ptr(classes), None, None, C.c_size_t(0), C.byref(k))
    if rc not in (0, 23):
        raise ZaAmdError(rc, lib.zg_strerror(rc).decode())
    idx = np.zeros(max(k.value, 1), dtype=np.uint32)
    vals = np.zeros(max(k.value, 1) * 32, dtype=np.uint8)
    if k.value:
        rc = lib.zg_witness_pack(_ptr(w), C.c_size_t(n), _ptr(classes), _ptr(idx), _ptr(vals), C.c_size_t(k.value), C.byref(k))
        if rc:
            raise ZaAmdError(rc, lib.zg_strerror(rc).decode())
    return classes, idx[:k.value], vals[:32 * k.value]


def pk_params_offset(pk_bytes):
    lib = load_library()
    b = _u8(pk_bytes)
    off = C.c_size_t()
    rc = lib.zg_pk_params_offset(_ptr(b), C.c_size_t(b.size), C.byref(off))
    if rc:
        raise ZaAmdError(rc, lib.zg_strerror(rc).decode())
    return off.value


def verify(vk_bytes, proof, public_inputs):
    """Groth16 verification on the host CPU (no GPU needed).  vk_bytes: bellman vk or whole Parameters bytes."""
    lib = load_library()
    v, p = _u8(vk_bytes), _u8(proof)
    pi = _u8(b"".join(int(x).to_bytes(32, "little") for x in public_inputs)) if public_inputs else np.zeros(1, np.uint8)
    ok = C.c_int()
    rc = lib.zg_verify(_ptr(v), C.c_size_t(v.size), _ptr(p), _ptr(pi), C.c_size_t(len(public_inputs)), C.byref(ok))
    if rc:
        raise ZaAmdError(rc, lib.zg_strerror(rc).decode())
    return bool(ok.value)


def vk_json(vk_bytes, input_names=()):
    lib = load_library()
    v = _u8(vk_bytes)
    names = b"".join(n.encode() + b"\0" for n in input_names)
    cap = 4096 + 200 * (len(v) // 64)
    buf = C.create_string_buffer(cap)
    rc = lib.zg_vk_json(_ptr(v), C.c_size_t(v.size), names if input_names else None, C.c_size_t(len(input_names)), buf,
                        C.c_size_t(cap))
    if rc:
        raise ZaAmdError(rc, lib.zg_strerror(rc).decode())
    return buf.value.decode()


def g1_sum(jac_points):
    """Sum of 96-byte Jacobian G1 results (bytes / uint8 array holding k of them)."""
    lib = load_library()
    a = _u8(jac_points)
    out = np.zeros(96, dtype=np.uint8)
    rc = lib.zg_g1_sum(_ptr(a), C.c_size_t(a.size // 96), _ptr(out))
    if rc:
        raise ZaAmdError(rc, lib.zg_strerror(rc).decode())
    return out.tobytes()


def _thread_error(lib, rc):
    return ZaAmdError(rc, "%s: %s" % (lib.zg_strerror(rc).decode(), lib.zg_last_error_thread().decode()))


def vk_from_json(text):
    """JsonVerifyingKey::from_json + to_bellman (format.rs:169-194) -> bellman VerifyingKey::write bytes."""
    lib = load_library()
    t = text.encode() if isinstance(text, str) else bytes(text)
    need = C.c_size_t()
    rc = lib.zg_vk_from_json(t, None, C.c_size_t(0), C.byref(need))
    if rc != ZG_E_BUFFER_TOO_SMALL:
        raise _thread_error(lib, rc)
    out = np.zeros(need.value, dtype=np.uint8)
    rc = lib.zg_vk_from_json(t, _ptr(out), C.c_size_t(out.size), C.byref(need))
    if rc:
        raise _thread_error(lib, rc)
    return out.tobytes()


def proof_from_json(text):
    """JsonProofAndInput::json_to_bellman (format.rs:101-121) -> (256 proof bytes, [public inputs as ints])."""
    lib = load_library()
    t = text.encode() if isinstance(text, str) else bytes(text)
    proof = np.zeros(256, dtype=np.uint8)
    n = C.c_size_t()
    cap = 32 * 64
    while True:
        pub = np.zeros(cap, dtype=np.uint8)
        rc = lib.zg_proof_from_json(t, _ptr(proof), _ptr(pub), C.c_size_t(cap), C.byref(n))
        if rc == ZG_E_BUFFER_TOO_SMALL and 32 * n.value > cap:
            cap = 32 * n.value
            continue
        if rc:
            raise _thread_error(lib, rc)
        break
    vals = [int.from_bytes(pub[32 * i:32 * i + 32].tobytes(), "little") for i in range(n.value)]
    return proof.tobytes(), vals


def verifier_solidity(vk_bytes, input_names=()):
    """generate_solidity (ethereum.rs:216-261): the contract `za setup` writes for verifier type "solidity"."""
    lib = load_library()
    v = _u8(vk_bytes)
    names = b"".join(n.encode() + b"\0" for n in input_names)
    need = C.c_size_t()
    rc = lib.zg_verifier_solidity(_ptr(v), C.c_size_t(v.size), names if input_names else None, C.c_size_t(len(input_names)),
                                  None, C.c_size_t(0), C.byref(need))
    if rc != ZG_E_BUFFER_TOO_SMALL:
        raise _thread_error(lib, rc)
    buf = C.create_string_buffer(need.value)
    rc = lib.zg_verifier_solidity(_ptr(v), C.c_size_t(v.size), names if input_names else None, C.c_size_t(len(input_names)),
                                  buf, C.c_size_t(need.value), C.byref(need))
    if rc:
        raise _thread_error(lib, rc)
    return buf.value.decode()


def proof_json(proof, public_inputs):
    """proof: 256 bytes; public_inputs: list of ints (ONE excluded).  Returns the proof.json text."""
    lib = load_library()
    p = _u8(proof)
    pi = _u8(b"".join(int(x).to_bytes(32, "little") for x in public_inputs)) if public_inputs else np.zeros(1, np.uint8)
    cap = 1024 + 80 * len(public_inputs)
    buf = C.create_string_buffer(cap)
    rc = lib.zg_proof_json(_ptr(p), _ptr(pi), C.c_size_t(len(public_inputs)), buf, C.c_size_t(cap))
    if rc:
        raise ZaAmdError(rc, lib.zg_strerror(rc).decode())
    return buf.value.decode()
