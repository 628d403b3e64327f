"""za2c's setup / prove / verify and the za-gpu driver end to end on the GPU (SURVEY 8b outer row): a proving.key
container in, fresh key + verifier out, proof.json out, verification through the JSON readers."""
import ctypes as C
import json
import os
import subprocess

import pytest

from oracle import groth16 as g
from tests.util import golden

pytestmark = pytest.mark.gpu


def _za2c():
    from za_amd import build as zb
    import za_amd
    za_amd.load_library()          # libamdhip64 from torch first, as everywhere else in the tests
    lib = C.CDLL(zb.LIB_ZA2C)
    lib.verbose.restype = None
    return lib


def _call_setup(lib, circuit, pk, vtype, cap=1 << 20):
    out, err = C.create_string_buffer(cap), C.create_string_buffer(4096)
    rc = lib.setup(circuit.encode(), pk.encode(), vtype.encode(), out, C.c_size_t(cap), err, C.c_size_t(4096))
    return rc, out.value.decode(), err.value.decode()


def _call_prove(lib, pk, inputs, cap=1 << 16):
    out, err = C.create_string_buffer(cap), C.create_string_buffer(4096)
    rc = lib.prove(pk.encode(), inputs.encode(), out, C.c_size_t(cap), err, C.c_size_t(4096))
    return rc, out.value.decode(), err.value.decode()


def _call_verify(lib, vk, proof):
    err = C.create_string_buffer(4096)
    return lib.verify(vk.encode(), proof.encode(), err, C.c_size_t(4096)), err.value.decode()


@pytest.mark.parametrize("name", ["proof_factor.json", "proof_chain7.json"])
def test_za2c_setup_prove_verify_cycle(ctx, tmp_path, name):
    """binding/go/test/test.go's cycle (setup -> prove -> verify true, wrong input false) through the C symbols."""
    import za_amd
    fx = golden(name)
    lib = _za2c()
    lib.verbose(0)
    circuit, pk, bw = str(tmp_path / "circuit.pk"), str(tmp_path / "proving.key"), str(tmp_path / "w.binwitness")
    with open(circuit, "wb") as f:
        f.write(bytes.fromhex(fx["pk_container"]))
    with open(bw, "wb") as f:
        f.write(bytes.fromhex(fx["binwitness"]))
    rc, vk_json, err = _call_setup(lib, circuit, pk, "json")
    assert rc == 0, err
    vk = g.vk_from_json(vk_json)                             # the oracle reads what the library wrote
    assert len(vk.ic) == fx["n_inputs"]
    assert json.loads(vk_json)["input_names"] == ["main.public[%d]" % i for i in range(fx["n_inputs"] - 1)]
    # the new container keeps the constraint sections and carries a fresh key (random toxic waste)
    new_pk = open(pk, "rb").read()
    old_pk = bytes.fromhex(fx["pk_container"])
    off = za_amd.pk_params_offset(old_pk)
    assert new_pk[:off] == old_pk[:off] and new_pk[off:] != old_pk[off:] and len(new_pk) == len(old_pk)
    assert za_amd.vk_from_json(vk_json) == new_pk[off:off + len(za_amd.vk_from_json(vk_json))]
    # a too small verifier buffer is ERR_BUFFER_TOO_SMALL = 1
    assert _call_setup(lib, circuit, str(tmp_path / "pk2"), "json", cap=64)[0] == 1
    # prove from the new key and the evaluator's witness dump
    rc, proof_json, err = _call_prove(lib, pk, json.dumps({"@binwitness": bw}))
    assert rc == 0, err
    proof, pub = g.proof_from_json(proof_json)
    assert pub == [int(x, 16) for x in fx["witness"][1:fx["n_inputs"]]]
    assert g.verify_proof(vk, proof, pub)                    # independent pairing check
    assert _call_prove(lib, pk, bw, cap=16)[0] == 1          # bare path form; proof buffer too small
    # verify: 0 for the proof, 2 for a wrong public input, 100 for garbage
    assert _call_verify(lib, vk_json, proof_json) == (0, "")
    j = json.loads(proof_json)
    j["public_inputs"][0] = str(int(j["public_inputs"][0]) + 1)
    assert _call_verify(lib, vk_json, json.dumps(j))[0] == 2
    rc, err = _call_verify(lib, vk_json, "[]")
    assert rc == 100 and err.startswith("Json(")
    # a witness that violates a constraint: the reference panics in check_constrains_eval_zero; here code 100
    bad = bytearray(bytes.fromhex(fx["binwitness"]))
    bad[-1] ^= 1
    with open(bw, "wb") as f:
        f.write(bytes(bad))
    rc, _, err = _call_prove(lib, pk, bw)
    assert rc == 100 and "check_constrains_eval_zero" in err
    # solidity verifier for the same circuit
    rc, sol, err = _call_setup(lib, circuit, str(tmp_path / "pk3"), "solidity")
    assert rc == 0 and sol.startswith("\npragma solidity ^0.5.0;") and "uint[%d] memory input" % (fx["n_inputs"] - 1) in sol


def test_za_gpu_driver_with_za_default_file_names(ctx, tmp_path):
    """`za setup` / `za prove` as commands (cli/src/main.rs:196-226, 249-269): default file names, verifier.sol by default."""
    from za_amd import build as zb
    fx = golden("proof_chain7.json")
    (tmp_path / "circuit.za").write_bytes(bytes.fromhex(fx["pk_container"]))       # the compiled container (deviation)
    (tmp_path / "circuit.za.names").write_text("main.out\nmain.in\n")
    (tmp_path / "w.binwitness").write_bytes(bytes.fromhex(fx["binwitness"]))
    r = subprocess.run([zb.DRIVER, "setup"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "proving.key").exists() and (tmp_path / "verifier.sol").exists()
    assert '// input = ["main.out", "main.in"]' in (tmp_path / "verifier.sol").read_text()
    assert "Setup time" in r.stderr
    r = subprocess.run([zb.DRIVER, "setup", "--verifiertype", "json", "--pk", "k2"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and (tmp_path / "verifier.json").exists()
    (tmp_path / "input.json").write_text(json.dumps({"@binwitness": "w.binwitness"}))
    r = subprocess.run([zb.DRIVER, "prove", "--pk", "k2"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Proof generation time" in r.stderr
    r = subprocess.run([zb.DRIVER, "verify"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "proof is valid" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([zb.DRIVER, "prove", "--pk", "k2", "--witness", "w.binwitness", "--proof", "p2.json"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0 and json.loads((tmp_path / "p2.json").read_text())["public_inputs"] == \
        json.loads((tmp_path / "proof.json").read_text())["public_inputs"]


def test_r1cs_check_finds_the_first_violated_constraint(ctx):
    from oracle import coracle as co
    from oracle.circuits import chain_r1cs
    r1, w = chain_r1cs(10, n_inputs=3, seed=9, bit_fraction=0.5)
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    assert R.check(co.witness_bytes(w)) is None
    # corrupt the variable defined by constraint 700 (aux variable = 3 inputs + 4 seeds + 700)
    w2 = list(w)
    w2[3 + 4 + 700] = (w2[3 + 4 + 700] + 1) % (1 << 200)
    bad = R.check(co.witness_bytes(w2))
    assert bad is not None and bad <= 700 and not g.R1CS(r1.num_inputs, r1.num_aux, [r1.constraints[bad]]).is_satisfied(w2)
    assert all(g.R1CS(r1.num_inputs, r1.num_aux, [c]).is_satisfied(w2) for c in r1.constraints[:bad])


def test_za2c_cycle_at_a_2p20_domain(ctx, tmp_path):
    """The outer boundary at size (it had only ever run at 2^0 / 2^7 domains): setup -> prove -> verify through
    the four C symbols on the 2^20 config-4 recipe written as the files za leaves on disk; prove() reads the key
    checked = true (device validation), checks the constraints, proves one-shot, self-verifies."""
    import za_amd
    from oracle import bn254 as bn
    from za_amd import synth
    sr = synth.SynthR1CS(20, 2, 0.97, 0x5A)
    R = sr.load(ctx)
    P, pbytes = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=True)
    want = ctx.prove(P, R, sr.witness, 5, 7)
    P.close(); R.close()
    circuit, pk, bw = str(tmp_path / "circuit.pk"), str(tmp_path / "proving.key"), str(tmp_path / "w.binwitness")
    sr.write_proving_key(circuit, pbytes)
    sr.write_binwitness(bw)
    w1 = int.from_bytes(bytes(sr.witness[32:64]), "little")
    lib = _za2c()
    lib.verbose(1)
    # prove from the container as it stands (the key of the in-process set-up): the public input and an accepted proof
    rc, proof_json, err = _call_prove(lib, circuit, bw)
    assert rc == 0, err
    proof, pub = za_amd.proof_from_json(proof_json)
    assert pub == [w1] and proof != want                      # fresh r, s
    assert za_amd.verify(pbytes, proof, [w1]) is True
    # setup re-keys the container (fresh toxic waste) and returns the verifier; prove + verify with the new key
    rc, vk_json, err = _call_setup(lib, circuit, pk, "json", cap=1 << 20)
    assert rc == 0, err
    rc, proof_json, err = _call_prove(lib, pk, json.dumps({"@binwitness": bw}))
    assert rc == 0, err
    assert _call_verify(lib, vk_json, proof_json) == (0, "")
    assert g.verify_proof(g.vk_from_json(vk_json), *g.proof_from_json(proof_json))     # independent pairing check
    j = json.loads(proof_json)
    j["public_inputs"][0] = str(int(j["public_inputs"][0]) ^ 1)
    assert _call_verify(lib, vk_json, json.dumps(j))[0] == 2
    # Parameters::read(pk, checked = true) inside prove(): a twist point outside the subgroup deep inside b_g2 -> code 100
    raw = bytearray(open(pk, "rb").read())
    at = len(raw) - 128 * 4321
    raw[at:at + 128] = g.g2_uncompressed(bn.g2_twist_point(11))
    bad_pk = str(tmp_path / "bad.key")
    with open(bad_pk, "wb") as f:
        f.write(raw)
    rc, _, err = _call_prove(lib, bad_pk, bw)
    assert rc == 100 and err == 'Io("b_g2: point is not in the correct subgroup")', err
    lib.verbose(0)
    sr.close()
