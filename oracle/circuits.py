"""Small satisfiable R1CS instances for the parity tests -- TEST ORACLE side, not product code."""
import random

from .bn254 import R
from .groth16 import R1CS


def chain_r1cs(log_m, n_inputs=3, seed=1, bit_fraction=0.5):
    """n_constraints + n_inputs = 2^log_m.  Each row defines one fresh aux variable, either as an XOR
    of two earlier bit variables ((2a)*b = a + b - c) or as a product of two random linear forms."""
    rng = random.Random(seed)
    m = 1 << log_m
    n = m - n_inputs
    n_seed = 4
    w = [1] + [rng.randrange(R) for _ in range(n_inputs - 1)]
    bits = []
    for i in range(n_seed):
        if i % 2 == 0:
            w.append(rng.randrange(2))
            bits.append(len(w) - 1)
        else:
            w.append(rng.randrange(R))
    cons = []
    for _ in range(n):
        c = len(w)
        if rng.random() < bit_fraction:
            a, b = rng.choice(bits), rng.choice(bits)
            cons.append(([(a, 2)], [(b, 1)], [(a, 1), (b, 1), (c, R - 1)]))
            w.append((w[a] + w[b] - 2 * w[a] * w[b]) % R)
            bits.append(c)
        else:
            x, y, z, t = (rng.randrange(c) for _ in range(4))
            k = [rng.randrange(1, R) for _ in range(4)]
            cons.append(([(x, k[0]), (y, k[1])], [(z, k[2]), (t, k[3])], [(c, 1)]))
            w.append((k[0] * w[x] + k[1] * w[y]) * (k[2] * w[z] + k[3] * w[t]) % R)
    r1 = R1CS(n_inputs, len(w) - n_inputs, cons)
    assert r1.is_satisfied(w)
    return r1, w


# ----------------------------------------------------------------------------- Poseidon chain (BASELINE config 3 shape)
# The generator is workload code shared with bench.py (za_amd/workloads.py); the hash is pinned by the reference's own
# known answers (interop/circuits/circomlib/test/poseidoncircuit.js:41,54) in tests/test_oracle_python.py, and the
# circuit by R1CS.is_satisfied below.
from za_amd.workloads import poseidon_chain, poseidon_constants, poseidon_hash, poseidon_matrix  # noqa: E402,F401


def poseidon_chain_r1cs(log_m, n_perm=None):
    """(R1CS, witness, [hash values]) of za_amd.workloads.poseidon_chain."""
    n_inputs, n_aux, cons, w, hashes = poseidon_chain(log_m, n_perm)
    return R1CS(n_inputs, n_aux, cons), w, hashes
