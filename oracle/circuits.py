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


# ----------------------------------------------------------------------------- R1CS with a prescribed H polynomial
def prescribed_h_r1cs(log_m, h):
    """(R1CS, witness) whose quotient polynomial H has exactly the coefficients `h` (m - 1 values, m = 2^log_m).

    One public input (ONE) and m - 1 rows u_i * v_i = t_i over fresh aux variables, so bellman's domain is exactly m
    (it appends the row a = ONE, b = c = 0).  With A(x) = x^(m-1)/omega (a_i = omega^(-i-1), and a_(m-1) = 1 as the
    appended row needs) and B(x) = sum_k beta_k x^k one gets A*B = (1/omega) sum_k beta_k x^(m-1+k), i.e.
    H = (1/omega) sum_(k>=1) beta_k x^(k-1): beta_(j+1) = omega h_j, and beta_0 makes B vanish at omega^(m-1).
    Lets a test put 0, 1, r - 1 ... into the dense h multiexp, which random circuits never do."""
    from .groth16 import Domain
    from .bn254 import inv
    m = 1 << log_m
    assert len(h) == m - 1 and log_m >= 1
    d = Domain(log_m)
    w_ = d.omega
    beta = [0] + [hj * w_ % R for hj in h]
    winv = d.omegainv
    acc, p = 0, winv
    for k in range(1, m):
        acc = (acc + beta[k] * p) % R
        p = p * winv % R
    beta[0] = (-acc) % R
    b = list(beta)
    d.fft(b)                      # b[i] = B(omega^i)
    assert b[m - 1] == 0
    a = []
    p = winv                      # a_i = omega^(-i-1)
    for _ in range(m):
        a.append(p)
        p = p * winv % R
    assert a[m - 1] == 1
    w = [1]
    cons = []
    for i in range(m - 1):
        u = len(w)
        w += [a[i], b[i], a[i] * b[i] % R]
        cons.append(([(u, 1)], [(u + 1, 1)], [(u + 2, 1)]))
    r1 = R1CS(1, len(w) - 1, cons)
    return r1, w
