"""Fixtures for the outer formats (vk JSON, Solidity verifier), derived from the keys of the proof fixtures.

Run in the build container only: the Solidity contract text is read from the reference checkout
(prover/src/groth16/ethereum.rs:8-214, the r#"..."# literal) and the fields are filled by the Python oracle's
restatement of generate_solidity, so the committed .sol files are the expected OUTPUT for the golden keys.
    python tests/golden/gen_outer_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import groth16 as g  # noqa: E402

REF = "/root/reference/prover/src/groth16/ethereum.rs"


def template():
    src = open(REF).read()
    a = src.index('r#"') + 3
    return src[a:src.index('"#;')]


def main():
    tpl = template()
    out = {}
    for name, names in (("proof_factor.json", ["main.r"]), ("proof_chain7.json", ["main.out", 'main.in["0"]'])):
        fx = json.load(open(os.path.join(HERE, name)))
        params, _ = g.params_read(bytes.fromhex(fx["params"]))
        assert len(names) == fx["n_inputs"] - 1
        stem = name[len("proof_"):-len(".json")]
        sol = g.solidity_verifier(tpl, params.vk, names)
        if stem == "chain7":   # one contract in full; the other key is pinned by the digest of its contract
            with open(os.path.join(HERE, "verifier_%s.sol" % stem), "w") as f:
                f.write(sol)
        out[stem] = {"input_names": names, "vk_json": g.vk_json(params.vk, names), "vk_bytes": g.vk_bytes(params.vk).hex(),
                     "solidity_sha256": hashlib.sha256(sol.encode()).hexdigest()}
        # the readers invert the writers
        vk2 = g.vk_from_json(out[stem]["vk_json"])
        assert g.vk_bytes(vk2) == g.vk_bytes(params.vk)
        proof, pub = g.proof_from_json(fx["proof_json"])
        assert g.proof_bytes(proof).hex() == fx["proof"]
        assert g.verify_proof(vk2, proof, pub)
    with open(os.path.join(HERE, "vk_json.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote verifier_*.sol, vk_json.json")


if __name__ == "__main__":
    main()
