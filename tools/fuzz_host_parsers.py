"""Mutation fuzzing of the host-only entry points that take untrusted bytes (no GPU needed):
zg_vk_from_json, zg_proof_from_json, zg_verify / zg_vk_json / zg_verifier_solidity over verifying-key bytes,
zg_pk_params_offset over a proving-key container.  Meant for the sanitizer build (tools/asan_host.sh); every call must
either succeed or return an error code -- never crash, hang or trip ASan / UBSan.
usage: python tools/fuzz_host_parsers.py [seconds] [seed]"""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import za_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20261002)
G = os.path.join(ROOT, "tests", "golden")
vkfx = json.load(open(os.path.join(G, "vk_json.json")))
seeds_json, seeds_proof, seeds_vk, seeds_pk = [], [], [], []
for stem in ("factor", "chain7"):
    pf = json.load(open(os.path.join(G, "proof_%s.json" % stem)))
    seeds_json.append(vkfx[stem]["vk_json"].encode())
    seeds_proof.append(pf["proof_json"].encode())
    seeds_vk.append(bytes.fromhex(vkfx[stem]["vk_bytes"]))
    if "pk" in pf:
        seeds_pk.append(bytes.fromhex(pf["pk"]))
    seeds_pk.append(bytes.fromhex(pf["params"]))
TOKENS = [b"0x", b"\"", b"{", b"}", b"[", b"]", b",", b":", b"-", b"1e9", b"null", b"\\u0000", b"\\", b"99999999999999999999999999999999999999999999999999999999999999999999999999999999",
          b"0x" + b"f" * 64, b"0x" + b"f" * 65, b" ", b"\n", b"\x00", b"\xff", b"true", b"\"protocol\"", b"\"inputs\""]


def mutate(b):
    b = bytearray(b)
    for _ in range(rng.choice([1, 1, 2, 3, 8])):
        k = rng.randrange(7)
        pos = rng.randrange(len(b) + 1) if b else 0
        if k == 0 and b:
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif k == 1 and b:
            del b[pos:pos + rng.choice([1, 1, 2, 8, 64, 1000])]
        elif k == 2:
            b[pos:pos] = rng.choice(TOKENS)
        elif k == 3 and b:
            b = b[:pos]
        elif k == 4 and b:
            lo = rng.randrange(len(b)); hi = min(len(b), lo + rng.choice([4, 32, 64]))
            b[pos:pos] = b[lo:hi]
        elif k == 5 and b:
            b[rng.randrange(len(b))] = rng.randrange(256)
        elif k == 6 and len(b) >= 4:
            p4 = rng.randrange(len(b) - 3)
            b[p4:p4 + 4] = rng.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x00", b"\xff\xff\xff\x7f", b"\x00\x00\x00\x80"])
    return bytes(b)


def guarded(fn, *a):
    try:
        return fn(*a)
    except za_amd.ZaAmdError:
        return None
    except (UnicodeDecodeError, ValueError):
        return None


t0, n, ok = time.time(), 0, 0
while time.time() - t0 < budget:
    which = rng.randrange(4)
    if which == 0:
        r = guarded(za_amd.vk_from_json, mutate(rng.choice(seeds_json)).decode("utf-8", "replace"))
    elif which == 1:
        r = guarded(za_amd.proof_from_json, mutate(rng.choice(seeds_proof)).decode("utf-8", "replace"))
    elif which == 2:
        vk = mutate(rng.choice(seeds_vk))
        fn = rng.choice([lambda: za_amd.vk_json(vk, ["main.a"]), lambda: za_amd.verifier_solidity(vk, ["main.a"]),
                         lambda: za_amd.verify(vk, bytes(256), [1])])
        r = guarded(fn)
    else:
        r = guarded(za_amd.pk_params_offset, mutate(rng.choice(seeds_pk)))
    n += 1
    ok += r is not None
print("fuzz_host_parsers: %d calls, %d accepted, %d refused, no crash" % (n, ok, n - ok))
