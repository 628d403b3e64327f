"""Synthetic workloads for bench.py that are real circuits rather than random matrices (pure Python, no GPU).

Poseidon chain = BASELINE.json config 3's family: circomlib's Poseidon(t=6, RF=8, RP=57) exactly as the reference
vendors it (interop/circuits/circomlib/src/poseidon.js:12-110, circuits/poseidon.circom): every round adds ONE constant
to all state words (ark), applies x^5 to all words (full rounds) or to word 0 (partial rounds), then the Cauchy-matrix
mix.  The hash is pinned by the reference's own known answers (test/poseidoncircuit.js:41,54) in tests/.
"""
import hashlib

import numpy as np

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # BN254 scalar field

_POS_T, _POS_RF, _POS_RP = 6, 8, 57


def _pseudo_random(seed, n):
    out, h = [], hashlib.blake2b(seed.encode(), digest_size=32).digest()
    while len(out) < n:
        out.append(int.from_bytes(h, "little") % R)
        h = hashlib.blake2b(h, digest_size=32).digest()
    return out


def poseidon_constants():
    return _pseudo_random("poseidon_constants", _POS_RF + _POS_RP)


def poseidon_matrix():
    t, nonce = _POS_T, 0
    while True:
        cm = _pseudo_random("poseidon_matrix_%04d" % nonce, 2 * t)
        if all(v != 0 for v in cm) and len(set(cm)) == len(cm):
            break
        nonce += 1
    return [[pow((cm[i] - cm[t + j]) % R, -1, R) for j in range(t)] for i in range(t)]


def poseidon_hash(inputs):
    C, M, t = poseidon_constants(), poseidon_matrix(), _POS_T
    st = [x % R for x in inputs] + [0] * (t - len(inputs))
    for i in range(_POS_RF + _POS_RP):
        st = [(x + C[i]) % R for x in st]
        if i < _POS_RF // 2 or i >= _POS_RF // 2 + _POS_RP:
            st = [pow(x, 5, R) for x in st]
        else:
            st[0] = pow(st[0], 5, R)
        st = [sum(M[a][b] * st[b] for b in range(t)) % R for a in range(t)]
    return st[0]


def poseidon_chain(log_m, n_perm=None):
    """x_0 = 1, x_1 = 2, x_{i+2} = Poseidon([x_{i+1}, x_i]) ... with the permutation's R1CS laid out as circom/za emit
    it: one row per ark output, three per S-box (x2 = x*x, x4 = x2*x2, y = x4*x), one per mix output; padded with
    `v * 1 = v` rows to n_constraints + n_inputs = 2^log_m.  Public input: the last hash.
    Returns (n_inputs, n_aux, constraints, witness, hashes); a constraint is (A, B, C), each a list of (var, coeff)."""
    C, M, t = poseidon_constants(), poseidon_matrix(), _POS_T
    n_inputs = 2
    w = [1, 0]            # ONE, placeholder for the public output
    cons = []

    def new(v):
        w.append(v % R)
        return len(w) - 1
    x_prev, x_cur = new(1), new(2)
    rows_per_perm = 1200
    if n_perm is None:
        n_perm = max(1, ((1 << log_m) - n_inputs) // rows_per_perm)
    hashes = []
    for _ in range(n_perm):
        st = [x_cur, x_prev] + [None] * (t - 2)     # None = the constant 0
        for i in range(_POS_RF + _POS_RP):
            nxt = []
            for v in st:                            # ark: out = in + C[i]
                val = (w[v] if v is not None else 0) + C[i]
                o = new(val)
                a = [(0, C[i])] + ([(v, 1)] if v is not None else [])
                cons.append((a, [(0, 1)], [(o, 1)]))
                nxt.append(o)
            st = nxt
            full = i < _POS_RF // 2 or i >= _POS_RF // 2 + _POS_RP
            for k in range(t if full else 1):
                x = st[k]
                x2 = new(w[x] * w[x]); cons.append(([(x, 1)], [(x, 1)], [(x2, 1)]))
                x4 = new(w[x2] * w[x2]); cons.append(([(x2, 1)], [(x2, 1)], [(x4, 1)]))
                y = new(w[x4] * w[x]); cons.append(([(x4, 1)], [(x, 1)], [(y, 1)]))
                st[k] = y
            nxt = []
            for a_ in range(t):
                val = sum(M[a_][b] * w[st[b]] for b in range(t))
                o = new(val)
                cons.append(([(st[b], M[a_][b]) for b in range(t)], [(0, 1)], [(o, 1)]))
                nxt.append(o)
            st = nxt
        hashes.append(w[st[0]])
        x_prev, x_cur = x_cur, st[0]
    w[1] = w[x_cur]                                 # the public output equals the last hash
    cons.append(([(x_cur, 1)], [(0, 1)], [(1, 1)]))
    target = (1 << log_m) - n_inputs
    assert len(cons) <= target, "2^%d rows cannot hold %d permutations" % (log_m, n_perm)
    k = 2
    while len(cons) < target:
        cons.append(([(k, 1)], [(0, 1)], [(k, 1)]))
        k = 2 + (k - 1) % (len(w) - 2)
    return n_inputs, len(w) - n_inputs, cons, w, hashes


def csr_arrays(constraints):
    """(row_ptr[3], col[3], coeff[3]) in the layout zg_r1cs_load takes: u64 row pointers, u32 flat variable indices,
    32-byte little-endian canonical coefficients."""
    row_ptr, col, coeff = [], [], []
    for k in range(3):
        rp, cl, cf = [0], [], bytearray()
        for cons in constraints:
            for v, co in cons[k]:
                cl.append(v)
                cf += (co % R).to_bytes(32, "little")
            rp.append(len(cl))
        row_ptr.append(np.array(rp, dtype=np.uint64))
        col.append(np.array(cl, dtype=np.uint32))
        coeff.append(np.frombuffer(bytes(cf), dtype=np.uint8) if cf else np.zeros(0, dtype=np.uint8))
    return row_ptr, col, coeff


def witness_bytes(w):
    return b"".join((x % R).to_bytes(32, "little") for x in w)
