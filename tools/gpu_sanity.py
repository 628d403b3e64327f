import sys, time, random
sys.path.insert(0, '.')
import numpy as np
from oracle import bn254 as bn, groth16 as g, coracle as co
import za_amd
random.seed(7)
ctx = za_amd.Context(0)
def norm_g1(j):  # 96 byte jac (Z=1 or 0)
    return None if j[64:96] == b"\0"*32 else bn.g1_from_mont_bytes(j[:64])
def norm_g2(j):
    return None if j[128:192] == b"\0"*64 else bn.g2_from_mont_bytes(j[:128])
# MSM small
for n in (1, 2, 31, 32, 33, 100, 1000):
    ks = [random.randrange(1, bn.R) for _ in range(n)]
    bases = b"".join(co.g1_mul(bn.g1_to_mont_bytes(bn.G1_GEN), k) for k in ks)
    sc = [random.choice([0, 1, bn.R-1, random.randrange(bn.R), random.randrange(1<<20)]) for _ in range(n)]
    exp = co.msm_g1(bases, co.witness_bytes(sc), threads=4)
    got = ctx.msm_g1(bases, co.witness_bytes(sc))
    ok = bn.g1_from_mont_bytes(exp) == norm_g1(got)
    print("msm g1", n, ok)
    assert ok
    if n <= 100:
        bases2 = b"".join(co.g2_mul(bn.g2_to_mont_bytes(bn.G2_GEN), k) for k in ks)
        exp2 = co.msm_g2(bases2, co.witness_bytes(sc), threads=4)
        got2 = ctx.msm_g2(bases2, co.witness_bytes(sc))
        ok = bn.g2_from_mont_bytes(exp2) == norm_g2(got2)
        print("msm g2", n, ok); assert ok
# NTT
for lg in (0, 1, 2, 5, 10, 12, 16):
    m = 1 << lg
    a = np.frombuffer(b"".join(bn.le32(random.randrange(bn.R)) for _ in range(m)), dtype=np.uint8)
    a = co.to_mont(1, a)
    for mode, (inv, cos) in enumerate(((0,0),(1,0),(0,1),(1,1))):
        e = co.fft(a, lg, mode, threads=4)
        got = ctx.ntt(a, lg, inverse=bool(inv), coset=bool(cos))
        ok = bytes(e) == bytes(got)
        print("ntt", lg, mode, ok); assert ok
# prove example
r1 = g.example_factor_r1cs(); w = [1, 6, 2, 3]
params, trap = g.generate_parameters(r1, 11, 12, 13, 14, 15)
proof = g.create_proof(r1, params, w, 5, 7)
csr = co.CsrR1CS.from_r1cs(r1)
P, used = ctx.params_load(g.params_write(params))
R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
pb = ctx.prove(P, R, co.witness_bytes(w), 5, 7)
print("prove", pb == g.proof_bytes(proof)); assert pb == g.proof_bytes(proof)
print(za_amd.proof_json(pb, [6]) == g.proof_json(proof, [6]))
print(ctx.last_timings())
print("ALL OK")
