"""The outer side of the boundary (SURVEY 8b outer row, 8f rows 2 and 4): JSON readers for the verifying key and for
proof + public inputs, the Solidity / JSON verifier export, and za2c's four C symbols.  CPU only: none of this needs
a GPU (the GPU halves of setup / prove are in tests/test_gpu_outer.py)."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import pytest

from oracle import bn254 as bn
from oracle import groth16 as g
from tests.util import GOLDEN, golden

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def za(lib_built):
    import za_amd
    return za_amd


@pytest.mark.parametrize("stem", ["factor", "chain7"])
def test_vk_json_round_trip_and_solidity_against_the_fixtures(za, stem):
    fx = golden("vk_json.json")[stem]
    pf = golden("proof_%s.json" % stem)
    vkb = za.vk_from_json(fx["vk_json"])
    assert vkb.hex() == fx["vk_bytes"] == pf["params"][:len(fx["vk_bytes"])]
    assert za.vk_json(vkb, fx["input_names"]) == fx["vk_json"]          # JsonVerifyingKey::to_json, format.rs:191-193
    sol = za.verifier_solidity(vkb, fx["input_names"])
    assert hashlib.sha256(sol.encode()).hexdigest() == fx["solidity_sha256"]
    if stem == "chain7":
        with open(os.path.join(GOLDEN, "verifier_chain7.sol")) as f:
            assert sol == f.read()                                       # generate_solidity, ethereum.rs:216-261
    # G2 is printed c1 first in the contract (ethereum.rs:231-236), c0 first in the JSON (format.rs:69-78)
    params, _ = g.params_read(bytes.fromhex(pf["params"]))
    b = params.vk.beta_g2
    assert "vk.b = Pairing.G2Point([0x%064x,0x%064x],[0x%064x,0x%064x]);" % (b[0].c1, b[0].c0, b[1].c1, b[1].c0) in sol


@pytest.mark.parametrize("stem", ["factor", "chain7"])
def test_proof_json_reader_inverts_the_writer(za, stem):
    pf = golden("proof_%s.json" % stem)
    proof, pub = za.proof_from_json(pf["proof_json"])
    assert proof.hex() == pf["proof"]
    assert pub == [int(x, 16) for x in pf["witness"][1:pf["n_inputs"]]]
    assert za.proof_json(proof, pub) == pf["proof_json"]
    vkb = za.vk_from_json(golden("vk_json.json")[stem]["vk_json"])
    assert za.verify(vkb, proof, pub) is True


def test_readers_accept_decimal_and_hex_like_fs_parse(za):
    """str_to_fq -> FS::parse (format.rs:33-36, fs.rs:43-55): decimal or 0x hex; whitespace and unknown fields are
    serde's business."""
    pf = golden("proof_factor.json")
    j = json.loads(pf["proof_json"])

    def dec(x):
        return str(int(x, 16))
    j2 = {"c": [dec(j["c"][0]), j["c"][1].upper().replace("0X", "0x")], "extra": {"ignored": [1, 2.5e3, None, True]},
          "a": [dec(v) for v in j["a"]], "b": [[dec(v) for v in row] for row in j["b"]], "public_inputs": ["6"]}
    proof, pub = za.proof_from_json(json.dumps(j2, indent=2))
    assert proof.hex() == pf["proof"] and pub == [6]
    assert g.proof_bytes(g.proof_from_json(json.dumps(j2))[0]).hex() == pf["proof"]


@pytest.mark.parametrize("mutate,needle", [
    (lambda j: j.pop("public_inputs"), "missing field `public_inputs`"),
    (lambda j: j.__setitem__("public_inputs", ["06"]), 'BadFormat("bad format")'),          # Fr::from_str: leading zero
    (lambda j: j.__setitem__("public_inputs", ["0x6"]), 'BadFormat("bad format")'),         # decimal only
    (lambda j: j.__setitem__("public_inputs", [""]), 'BadFormat("bad format")'),
    (lambda j: j.__setitem__("public_inputs", [str(bn.R)]), 'BadFormat("bad format")'),     # x + r aliases x
    (lambda j: j.__setitem__("public_inputs", [6]), "expected a string"),
    (lambda j: j["a"].__setitem__(0, "12x"), "is not decimal"),
    (lambda j: j["a"].__setitem__(0, "0xzz"), "is not hexadecimal"),
    (lambda j: j["a"].__setitem__(0, str(bn.Q)), "is not a field element"),
    (lambda j: j["a"].__setitem__(0, "1" + "0" * 80), "is not a field element"),
    (lambda j: j["a"].__setitem__(0, "5"), "bad coordinates"),                                # not on the curve
    (lambda j: j["b"][0].__setitem__(0, "5"), "bad coordinates"),
    (lambda j: j.__setitem__("a", j["a"] + ["1"]), "invalid length 3"),
])
def test_proof_json_reader_rejections(za, mutate, needle):
    j = json.loads(golden("proof_factor.json")["proof_json"])
    mutate(j)
    with pytest.raises(za.ZaAmdError) as e:
        za.proof_from_json(json.dumps(j))
    assert e.value.code == za.ZG_E_BAD_FORMAT and needle in str(e.value)
    with pytest.raises(Exception):
        g.proof_from_json(json.dumps(j))          # the oracle refuses the same inputs


def test_g2_point_outside_the_subgroup_is_rejected(za):
    """A point on the twist but outside the order-r subgroup is not a G2 element: the readers, Parameters::read
    (checked) and zg_verify refuse it (pairing_ce's into_affine checks the subgroup for G2)."""
    q = bn.Q

    def sqrt_fq(a):
        s = pow(a, (q + 1) // 4, q)
        return s if s * s % q == a % q else None

    def sqrt_fq2(a):
        n = sqrt_fq((a.c0 * a.c0 + a.c1 * a.c1) % q)
        if n is None:
            return None
        for s in (n, q - n):
            x0 = sqrt_fq((a.c0 + s) * pow(2, q - 2, q) % q)
            if x0:
                x1 = a.c1 * pow(2 * x0, q - 2, q) % q
                r = bn.Fq2(x0, x1)
                if r * r == a:
                    return r
        return None
    k = 1
    while True:
        x = bn.Fq2(k, 1)
        y = sqrt_fq2(x * x * x + bn.G2_B)
        if y is not None and bn.pt_mul((x, y), bn.R, order=1 << 300) is not None:
            break
        k += 1
    P = (x, y)
    assert bn.is_on_curve(P, bn.G2_B)
    pf = golden("proof_factor.json")
    j = json.loads(pf["proof_json"])
    j["b"] = [["0x%064x" % P[0].c0, "0x%064x" % P[0].c1], ["0x%064x" % P[1].c0, "0x%064x" % P[1].c1]]
    with pytest.raises(za.ZaAmdError) as e:
        za.proof_from_json(json.dumps(j))
    assert "bad coordinates" in str(e.value)
    # zg_verify with raw bytes
    pb = bytearray(bytes.fromhex(pf["proof"]))
    pb[64:192] = b"".join(int(v).to_bytes(32, "big") for v in (P[0].c0, P[0].c1, P[1].c0, P[1].c1))
    with pytest.raises(za.ZaAmdError) as e:
        za.verify(bytes.fromhex(pf["params"]), bytes(pb), [6])
    assert e.value.code == za.ZG_E_BAD_FORMAT
    # a verifying key whose gamma_g2 is such a point
    vkb = bytearray(bytes.fromhex(golden("vk_json.json")["factor"]["vk_bytes"]))
    vkb[256:384] = g.g2_uncompressed(P)
    with pytest.raises(za.ZaAmdError) as e:
        za.verify(bytes(vkb), bytes.fromhex(pf["proof"]), [6])
    assert e.value.code == za.ZG_E_IO


def test_verify_rejects_non_canonical_public_input_bytes(za):
    """x and x + r select the same ic multiple; the byte-level zg_verify must refuse x + r."""
    pf = golden("proof_factor.json")
    lib = za.load_library()
    import numpy as np
    vk = np.frombuffer(bytes.fromhex(pf["params"]), dtype=np.uint8)
    proof = np.frombuffer(bytes.fromhex(pf["proof"]), dtype=np.uint8)
    ok = C.c_int(7)
    for val, want_rc in ((6, 0), (6 + bn.R, za.ZG_E_BAD_FORMAT)):
        pub = np.frombuffer(int(val).to_bytes(32, "little"), dtype=np.uint8)
        rc = lib.zg_verify(vk.ctypes.data_as(C.c_void_p), C.c_size_t(vk.size), proof.ctypes.data_as(C.c_void_p),
                           pub.ctypes.data_as(C.c_void_p), C.c_size_t(1), C.byref(ok))
        assert rc == want_rc
    assert ok.value == 0


def test_vk_json_reader_rejections(za):
    fx = golden("vk_json.json")["factor"]
    j = json.loads(fx["vk_json"])
    for k in ("alpha_g1", "ic", "input_names"):
        j2 = dict(j)
        j2.pop(k)
        with pytest.raises(za.ZaAmdError) as e:
            za.vk_from_json(json.dumps(j2))
        assert "missing field `%s`" % k in str(e.value)
    j2 = json.loads(fx["vk_json"])
    j2["ic"][1][1] = "7"
    with pytest.raises(za.ZaAmdError) as e:
        za.vk_from_json(json.dumps(j2))
    assert "bad coordinates" in str(e.value)
    with pytest.raises(za.ZaAmdError) as e:
        za.vk_from_json(fx["vk_json"] + " x")
    assert "trailing characters" in str(e.value)
    # names with escapes survive the writer / reader pair
    names = ['main.a"b', "main.é\\x"]
    vkb = za.vk_from_json(fx["vk_json"])
    txt = za.vk_json(vkb + b"", names[:1])
    assert json.loads(txt)["input_names"] == names[:1]
    assert za.vk_from_json(txt) == vkb


def test_za2c_symbols_from_c(lib_built, tmp_path):
    """A C host linked against libza2c.so: verbose / setup / prove / verify with caller buffers, the buffer-too-small
    path, verify accepting the golden proof.json with the library's own vk JSON and rejecting a wrong input with 2."""
    from za_amd import build as zb
    pf = golden("proof_factor.json")
    vk_json = golden("vk_json.json")["factor"]["vk_json"]
    j = json.loads(pf["proof_json"])
    j["public_inputs"] = ["7"]
    (tmp_path / "vk.json").write_text(vk_json)
    (tmp_path / "proof.json").write_text(pf["proof_json"])
    (tmp_path / "bad.json").write_text(json.dumps(j))
    exe = str(tmp_path / "t_za2c")
    libdir = os.path.dirname(zb.LIB_ZA2C)
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-o", exe, os.path.join(HERE, "c", "test_za2c.c"), "-L" + libdir,
                    "-lza2c", "-Wl,-rpath," + libdir, "-Wl,-rpath-link," + libdir], check=True)
    r = subprocess.run([exe, str(tmp_path / "vk.json"), str(tmp_path / "proof.json"), str(tmp_path / "bad.json")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "za2c ok" in r.stdout


def test_za2c_prove_reads_its_witness_path_like_serde_json(lib_built, tmp_path):
    """The inputs argument of za2c's prove names the witness dump as a JSON string; escapes follow serde_json: a
    surrogate PAIR is one code point (4 bytes of UTF-8), a lone surrogate and \\u0000 are Json errors.  CPU only: a path
    that parses reaches the file system (Io(..)), one that does not never gets that far."""
    from za_amd import build as zb
    lib = C.CDLL(zb.LIB_ZA2C)

    def prove(inputs):
        out, err = C.create_string_buffer(4096), C.create_string_buffer(4096)
        rc = lib.prove(str(tmp_path / "missing.key").encode(), inputs.encode(), out, C.c_size_t(4096), err, C.c_size_t(4096))
        return rc, err.value.decode()

    rc, err = prove('{"@binwitness": "w\\ud83d\\ude00.bin"}')
    assert rc == 100 and err.startswith("Io("), err            # parsed: the missing key file is what fails
    (tmp_path / "missing.key").write_bytes(b"x")
    rc, err = prove('{"@binwitness": "%s/w\\ud83d\\ude00.bin"}' % tmp_path)
    assert rc == 100 and "w\U0001F600.bin" in err, err           # ... and the path is the 4-byte UTF-8 form
    for bad in ('"\\ud83d"', '"\\ud83dx"', '"\\ude00"', '"a\\u0000b"', '"\\ud83d\\u0041"'):
        rc, err = prove('{"@binwitness": %s}' % bad)
        assert rc == 100 and err.startswith("Json("), (bad, err)


def test_za_gpu_driver_verify_arm(lib_built, tmp_path):
    from za_amd import build as zb
    pf = golden("proof_chain7.json")
    (tmp_path / "verifier.json").write_text(golden("vk_json.json")["chain7"]["vk_json"])
    (tmp_path / "proof.json").write_text(pf["proof_json"])
    r = subprocess.run([zb.DRIVER, "verify"], cwd=tmp_path, capture_output=True, text=True)     # za's default file names
    assert r.returncode == 0 and "proof is valid" in r.stdout, r.stdout + r.stderr
    j = json.loads(pf["proof_json"])
    j["public_inputs"][1] = "1"
    (tmp_path / "p2.json").write_text(json.dumps(j))
    r = subprocess.run([zb.DRIVER, "verify", "--vk", "verifier.json", "--proof", "p2.json"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 1 and "NOT valid" in r.stdout


def test_untrusted_input_fuzz_smoke(lib_built):
    """A few seconds of tools/fuzz_host_parsers.py (mutated vk / proof JSON, verifying-key bytes, pk containers) against
    the regular build: every call returns a code.  tools/asan_host.sh runs the same fuzzer on the ASan / UBSan build."""
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "fuzz_host_parsers.py"), "4", "7"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "no crash" in r.stdout
