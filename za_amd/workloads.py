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


# ------------------------------------------------------------------------------------------------------------------
# SHA-256 Merkle inclusion = BASELINE.json config 4's family ("SHA256 Merkle-inclusion depth-32"): a bit-level R1CS of
# `depth` two-to-one SHA-256 hashes (64-byte message = left || right, so two compressions per level, the second over the
# constant padding block), a direction bit per level choosing the order, the root as two 128-bit public inputs.
# Every wire is a bit, as in circomlib's sha256 (interop/circuits/circomlib/circuits/sha256/*.circom); the gadgets are
# this file's own (xor: 1 constraint, ch: 1, maj: 2, k-operand addition mod 2^32: one linear row + booleanity of the
# 32 + carry result bits), so the count per compression (~23 k) differs from circomlib's (~29 k).  The witness comes out
# of the same pass and the root is pinned by hashlib in tests/.  Arrays are built with numpy, one gadget = 32 rows.
_SHA_K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
_SHA_H0 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


class _Word:
    """32 wires, least significant bit first: `var` = variable index per bit (-1: the bit is the constant in `val`),
    `val` = the bit values (uint8)."""
    __slots__ = ("var", "val")

    def __init__(self, var, val):
        self.var, self.val = var, val

    @staticmethod
    def const(x):
        return _Word(np.full(32, -1, dtype=np.int64), np.array([(x >> i) & 1 for i in range(32)], dtype=np.uint8))

    def rotr(self, r):
        return _Word(np.roll(self.var, -r), np.roll(self.val, -r))

    def shr(self, r):
        var = np.concatenate([self.var[r:], np.full(r, -1, dtype=np.int64)])
        val = np.concatenate([self.val[r:], np.zeros(r, dtype=np.uint8)])
        return _Word(var, val)

    def value(self):
        return int(sum(int(b) << i for i, b in enumerate(self.val)))


class _ShaBuilder:
    def __init__(self, n_inputs):
        self.n_inputs = n_inputs          # variable 0 = ONE, 1 .. n_inputs-1 public
        self.values = [np.zeros(n_inputs, dtype=object)]
        self.values[0][0] = 1
        self.nvars = n_inputs
        self.m = [dict(lens=[], cols=[], coefs=[]) for _ in range(3)]
        self.n_rows = 0

    def new_bits(self, vals):
        v = np.arange(self.nvars, self.nvars + len(vals), dtype=np.int64)
        self.nvars += len(vals)
        self.values.append(np.asarray(vals, dtype=object))
        return v

    def _push(self, k, cols, coefs):
        """cols / coefs: (rows, terms) arrays; terms with column < 0 are dropped (constant-zero operands)."""
        cols = np.asarray(cols, dtype=np.int64)
        coefs = np.asarray(coefs, dtype=np.int64)
        if cols.ndim == 1:
            cols, coefs = cols[None, :], coefs[None, :]
        keep = (cols >= 0) & (coefs != 0)
        self.m[k]["lens"].append(keep.sum(axis=1).astype(np.int64))
        self.m[k]["cols"].append(cols[keep])
        self.m[k]["coefs"].append(coefs[keep])

    def rows(self, A, B, C):
        """One batch of constraints <A, w> * <B, w> = <C, w>; each of A, B, C = (cols, coefs) with equal row counts."""
        n = None
        for k, (cols, coefs) in enumerate((A, B, C)):
            cols = np.asarray(cols, dtype=np.int64)
            n = cols.shape[0] if cols.ndim == 2 else 1
            self._push(k, cols, coefs)
        self.n_rows += n

    # ---- gadgets over 32-bit words (bits that are constants never get a row) ----
    def boolean(self, var):
        n = len(var)
        one = np.zeros(n, dtype=np.int64)
        self.rows((var[:, None], np.ones((n, 1), dtype=np.int64)),
                  (np.stack([var, one], axis=1), np.stack([np.ones(n, dtype=np.int64), -np.ones(n, dtype=np.int64)], axis=1)),
                  (np.full((n, 1), -1, dtype=np.int64), np.zeros((n, 1), dtype=np.int64)))

    def xor2(self, x, y):
        val = x.val ^ y.val
        var = np.where(x.var < 0, y.var, x.var).copy()        # one side constant 0: the other side as is
        cx, cy = x.var < 0, y.var < 0
        flip = (cx & (x.val == 1)) | (cy & (y.val == 1))      # xor with the constant 1: handled as 1 - bit below
        both = ~cx & ~cy
        if both.any():
            nv = self.new_bits(val[both])
            a, b = x.var[both], y.var[both]
            n = len(nv)
            o = np.ones(n, dtype=np.int64)
            self.rows((a[:, None], 2 * o[:, None]), (b[:, None], o[:, None]),
                      (np.stack([a, b, nv], axis=1), np.stack([o, o, -o], axis=1)))
            var[both] = nv
        if flip.any():
            idx = np.where(flip & ~(cx & cy))[0]
            if len(idx):                                       # 1 - bit as a new wire: (1 - b) * 1 = n
                src = var[idx]
                nv = self.new_bits(val[idx])
                n = len(nv)
                o = np.ones(n, dtype=np.int64)
                z = np.zeros(n, dtype=np.int64)
                self.rows((np.stack([z, src], axis=1), np.stack([o, -o], axis=1)), (z[:, None], o[:, None]), (nv[:, None], o[:, None]))
                var[idx] = nv
        var[cx & cy] = -1
        return _Word(var, val)

    def xor3(self, x, y, z):
        return self.xor2(self.xor2(x, y), z)

    @staticmethod
    def _t(w, coef=1):
        """(cols, coefs) of the 32 bits of a word with a common coefficient; a constant bit becomes coef * bit on ONE."""
        c = np.full(32, coef, dtype=np.int64)
        isc = w.var < 0
        return np.where(isc, 0, w.var), np.where(isc, c * w.val.astype(np.int64), c)

    @staticmethod
    def _lc(*terms):
        cols, coefs = zip(*terms)
        return np.stack(cols, axis=1), np.stack(coefs, axis=1)

    def ch(self, e, f, g):
        val = (e.val & f.val) | ((1 - e.val) & g.val)
        if all((w.var < 0).all() for w in (e, f, g)):
            return _Word(np.full(32, -1, dtype=np.int64), val)
        nv = self.new_bits(val)
        o = np.ones(32, dtype=np.int64)
        self.rows(self._lc(self._t(e)), self._lc(self._t(f), self._t(g, -1)), self._lc((nv, o), self._t(g, -1)))
        return _Word(nv, val)

    def maj(self, a, b, c):
        val = (a.val & b.val) | (a.val & c.val) | (b.val & c.val)
        if all((w.var < 0).all() for w in (a, b, c)):
            return _Word(np.full(32, -1, dtype=np.int64), val)
        tval = a.val & b.val
        t = self.new_bits(tval)
        o = np.ones(32, dtype=np.int64)
        self.rows(self._lc(self._t(a)), self._lc(self._t(b)), self._lc((t, o)))
        nv = self.new_bits(val)
        self.rows(self._lc(self._t(c)), self._lc(self._t(a), self._t(b), (t, -2 * o)), self._lc((nv, o), (t, -o)))
        return _Word(nv, val)

    def add(self, words, const=0):
        """Sum of the words (+ a constant) mod 2^32: one linear row  sum = result + 2^32 * carry  and booleanity of the
        32 result bits and the carry bits."""
        total = sum(w.value() for w in words) + const
        kmax = (len(words) * 0xFFFFFFFF + const) >> 32
        nc = max(kmax.bit_length(), 0)
        bits = np.array([(total >> i) & 1 for i in range(32 + nc)], dtype=np.uint8)
        nv = self.new_bits(bits)
        self.boolean(nv)
        pw = np.array([1 << i for i in range(32)], dtype=np.int64)
        cols = [np.zeros(1, dtype=np.int64)] + [w.var for w in words]
        cconst = const + sum(int((w.val[w.var < 0].astype(np.int64) * pw[w.var < 0]).sum()) for w in words)
        coefs = [np.array([cconst], dtype=np.int64)] + [pw for _ in words]
        self.rows((np.concatenate(cols), np.concatenate(coefs)), (np.zeros(1, dtype=np.int64), np.ones(1, dtype=np.int64)),
                  (nv, np.array([1 << i for i in range(32 + nc)], dtype=np.int64)))
        return _Word(nv[:32], bits[:32])

    def big_s0(self, a):
        return self.xor3(a.rotr(2), a.rotr(13), a.rotr(22))

    def big_s1(self, e):
        return self.xor3(e.rotr(6), e.rotr(11), e.rotr(25))

    def small_s0(self, x):
        return self.xor3(x.rotr(7), x.rotr(18), x.shr(3))

    def small_s1(self, x):
        return self.xor3(x.rotr(17), x.rotr(19), x.shr(10))

    def compress(self, state, w16):
        """One SHA-256 compression; `w16` words may be constants (then the schedule costs nothing)."""
        W = list(w16)
        all_const = all((w.var < 0).all() for w in W)
        for t in range(16, 64):
            if all_const:
                s0 = (lambda x: ((x >> 7 | x << 25) ^ (x >> 18 | x << 14) ^ (x >> 3)) & 0xFFFFFFFF)(W[t - 15].value())
                s1 = (lambda x: ((x >> 17 | x << 15) ^ (x >> 19 | x << 13) ^ (x >> 10)) & 0xFFFFFFFF)(W[t - 2].value())
                W.append(_Word.const((W[t - 16].value() + s0 + W[t - 7].value() + s1) & 0xFFFFFFFF))
            else:
                W.append(self.add([W[t - 16], self.small_s0(W[t - 15]), W[t - 7], self.small_s1(W[t - 2])]))
        a, b, c, d, e, f, g, h = state
        for t in range(64):
            s1, chv, s0, mj = self.big_s1(e), self.ch(e, f, g), self.big_s0(a), self.maj(a, b, c)
            new_e = self.add([d, h, s1, chv, W[t]], _SHA_K[t])
            new_a = self.add([h, s1, chv, W[t], s0, mj], _SHA_K[t])
            a, b, c, d, e, f, g, h = new_a, a, b, c, new_e, e, f, g
        return [self.add([x, y]) for x, y in zip(state, (a, b, c, d, e, f, g, h))]

    def finish(self):
        values = np.concatenate(self.values)
        out = []
        for k in range(3):
            lens = np.concatenate(self.m[k]["lens"])
            rp = np.zeros(len(lens) + 1, dtype=np.uint64)
            np.cumsum(lens, out=rp[1:])
            out.append((rp, np.concatenate(self.m[k]["cols"]).astype(np.uint32), np.concatenate(self.m[k]["coefs"])))
        return values, out


def _coeff_bytes(coefs):
    """int64 coefficients -> 32-byte little-endian canonical field elements (negative values as R - |c|)."""
    uniq, inv = np.unique(coefs, return_inverse=True)
    table = np.frombuffer(b"".join((int(c) % R).to_bytes(32, "little") for c in uniq), dtype=np.uint8).reshape(len(uniq), 32)
    return table[inv].reshape(-1)


def _bytes_to_words(b):
    """32 bytes -> 8 big-endian 32-bit integers (SHA-256's word order)."""
    return [int.from_bytes(b[4 * i:4 * i + 4], "big") for i in range(8)]


def sha256_merkle(depth, seed=1):
    """R1CS + witness of a depth-`depth` SHA-256 Merkle inclusion proof.
    Returns (n_inputs, n_aux, n_constraints, row_ptr[3], col[3], coeff_bytes[3], witness_bytes, info) in the layout
    zg_r1cs_load / zg_prove take; info = {"root", "leaf", "siblings", "directions"} for the hashlib cross-check."""
    rnd = hashlib.blake2b(b"sha256_merkle:%d:%d" % (depth, seed), digest_size=64).digest
    leaf = hashlib.sha256(b"leaf" + seed.to_bytes(4, "little")).digest()
    sibs = [hashlib.sha256(b"sib%d:%d" % (i, seed)).digest() for i in range(depth)]
    dirs = [(rnd()[i % 64] >> (i // 64 % 8)) & 1 for i in range(depth)]
    B = _ShaBuilder(n_inputs=3)

    def private_words(b):
        words = []
        for x in _bytes_to_words(b):
            bits = np.array([(x >> i) & 1 for i in range(32)], dtype=np.uint8)
            v = B.new_bits(bits)
            B.boolean(v)
            words.append(_Word(v, bits))
        return words

    cur = private_words(leaf)
    cur_bytes = leaf
    pad = [_Word.const(0x80000000)] + [_Word.const(0)] * 14 + [_Word.const(512)]
    o32 = np.ones(32, dtype=np.int64)
    for lvl in range(depth):
        sib = private_words(sibs[lvl])
        dbit = B.new_bits(np.array([dirs[lvl]], dtype=np.uint8))
        B.boolean(dbit)
        left, right = [], []
        for cw, sw in zip(cur, sib):              # left = cur + d (sib - cur), right = cur + sib - left
            lval = np.where(dirs[lvl], sw.val, cw.val).astype(np.uint8)
            lv = B.new_bits(lval)
            B.rows((np.full((32, 1), dbit[0], dtype=np.int64), o32[:, None]),
                   (np.stack([sw.var, cw.var], axis=1), np.stack([o32, -o32], axis=1)),
                   (np.stack([lv, cw.var], axis=1), np.stack([o32, -o32], axis=1)))
            rval = np.where(dirs[lvl], cw.val, sw.val).astype(np.uint8)
            rv = B.new_bits(rval)
            B.rows((np.stack([sw.var, cw.var, lv], axis=1), np.stack([o32, o32, -o32], axis=1)),
                   (np.zeros((32, 1), dtype=np.int64), o32[:, None]), (rv[:, None], o32[:, None]))
            left.append(_Word(lv, lval))
            right.append(_Word(rv, rval))
        st = B.compress([_Word.const(x) for x in _SHA_H0], left + right)
        cur = B.compress(st, pad)
        lb, rb = (sibs[lvl], cur_bytes) if dirs[lvl] else (cur_bytes, sibs[lvl])
        cur_bytes = hashlib.sha256(lb + rb).digest()
        assert b"".join(w.value().to_bytes(4, "big") for w in cur) == cur_bytes, "gadget witness != hashlib at level %d" % lvl
    # public inputs: the root as two 128-bit integers (big-endian halves of the digest)
    root_hi, root_lo = int.from_bytes(cur_bytes[:16], "big"), int.from_bytes(cur_bytes[16:], "big")
    one_c, one_v = np.zeros(1, dtype=np.int64), np.ones(1, dtype=np.int64)
    pw32 = np.array([1 << i for i in range(32)], dtype=np.int64)
    for inp, words in ((1, cur[:4]), (2, cur[4:])):
        # inp = w0 * 2^96 + w1 * 2^64 + w2 * 2^32 + w3 (big-endian words), built from 32-bit word values so that every
        # coefficient fits 64 bits:  wv_j = sum 2^i bit_i;  t = wv0 * 2^32 + wv1;  u = t * 2^32 + wv2;  inp = u * 2^32 + wv3
        wv = []
        for w in words:
            v = B.new_bits(np.array([w.value()], dtype=object))
            B.rows((w.var, pw32), (one_c, one_v), (v, one_v))
            wv.append((v[0], w.value()))
        acc_var, acc_val = wv[0]
        for j in (1, 2, 3):
            nxt_val = (acc_val << 32) + wv[j][1]
            nxt_var = inp if j == 3 else B.new_bits(np.array([nxt_val], dtype=object))[0]
            B.rows((np.array([acc_var]), one_v), (one_c, np.array([1 << 32])), (np.array([nxt_var, wv[j][0]]), np.array([1, -1])))
            acc_var, acc_val = nxt_var, nxt_val
    values, mats = B.finish()
    values[1], values[2] = root_hi, root_lo
    n_aux = B.nvars - 3
    row_ptr = [m[0] for m in mats]
    col = [m[1] for m in mats]
    coeff = [_coeff_bytes(m[2]) for m in mats]
    wit = b"".join(int(v).to_bytes(32, "little") for v in values)
    info = {"root": cur_bytes, "leaf": leaf, "siblings": sibs, "directions": dirs, "nnz": [int(len(c)) for c in col]}
    return 3, n_aux, B.n_rows, row_ptr, col, coeff, wit, info


# ------------------------------------------------------------------------------------------------------------------
# EdDSA-Poseidon signature verification on Baby Jubjub = BASELINE.json config 2's family (circomlib's
# EdDSAPoseidonVerifier, interop/circuits/circomlib/circuits/eddsaposeidon.circom): checks  S * B8 == R8 + h * (8 A)  with
# h = Poseidon(R8x, R8y, Ax, Ay, M), S < the subgroup order.  The statement and the curve / hash parameters are the
# reference's (babyjub.circom: a = 168700, d = 168696; the base point B8 of eddsaposeidon.circom:104-107; Poseidon(5, 6,
# 8, 57)); the gadgets are this file's own (a bit-by-bit double-and-add instead of circomlib's windowed Montgomery
# ladders), so the constraint count (~7.5 k) is of the reference circuit's order, not equal to it.  The witness is the
# reference's own signature (interop/circuits/circomlib/za_test/eddsaposeidon.za:7-12).
_BJ_A, _BJ_D = 168700, 168696
_BJ_B8 = (5299619240641551281634865583518297030282874472190772894086521144482721001553,
          16950150798460657717958625567821834550301663161624707787222815936182638968203)
_BJ_SUBORDER = 2736030358979909402780800718157159386076813972158567259200215660948447373041
EDDSA_KAT = {   # za_test/eddsaposeidon.za
    "Ax": 13277427435165878497778222415993513565335242147425444199013288855685581939618,
    "Ay": 13622229784656158136036771217484571176836296686641868549125388198837476602820,
    "R8x": 11220723668893468001994760120794694848178115379170651044669708829805665054484,
    "R8y": 2367470421002446880004241260470975644531657398480773647535134774673409612366,
    "S": 1307100909096544936550139783786226891472336052773077686618468233418583414320,
    "M": 1234,
}


def babyjub_add(P, Q):
    x1, y1 = P
    x2, y2 = Q
    t = _BJ_D * x1 * x2 * y1 * y2 % R
    return ((x1 * y2 + y1 * x2) * pow(1 + t, -1, R) % R, (y1 * y2 - _BJ_A * x1 * x2) * pow(1 - t, -1, R) % R)


def babyjub_mul(k, P):
    acc = (0, 1)
    while k:
        if k & 1:
            acc = babyjub_add(acc, P)
        P = babyjub_add(P, P)
        k >>= 1
    return acc


def eddsa_poseidon_verifier(sig=None):
    """R1CS + witness of one EdDSA-Poseidon verification.  Public inputs: Ax, Ay, M; private: R8x, R8y, S.
    Returns (n_inputs, n_aux, constraints, witness) in poseidon_chain's form (a constraint = (A, B, C), each a list of
    (variable, coefficient))."""
    sig = dict(EDDSA_KAT if sig is None else sig)
    C, Mx, t = poseidon_constants(), poseidon_matrix(), _POS_T
    w = [1, sig["Ax"] % R, sig["Ay"] % R, sig["M"] % R]
    n_inputs = 4
    ONE, AX, AY, MSG = 0, 1, 2, 3
    cons = []

    def new(v):
        w.append(v % R)
        return len(w) - 1

    def val(lc):
        return sum(c * w[v] for v, c in lc) % R

    def mul(a, b):                     # new wire = <a, w> * <b, w>
        o = new(val(a) * val(b))
        cons.append((a, b, [(o, 1)]))
        return o

    def lin(lc):                       # new wire equal to a linear combination
        o = new(val(lc))
        cons.append((lc, [(ONE, 1)], [(o, 1)]))
        return o

    def bits_of(var, n, value):
        bs = []
        for i in range(n):
            b = new((value >> i) & 1)
            cons.append(([(b, 1)], [(b, 1), (ONE, -1)], []))          # b (b - 1) = 0
            bs.append(b)
        cons.append(([(b, 1 << i) for i, b in enumerate(bs)], [(ONE, 1)], [(var, 1)]))
        return bs

    def less_than_const(bs, bound):
        """value(bs) < bound for a constant bound: walk from the top bit keeping `eq` = "all higher bits equal the
        bound's"; where the bound has a 0 an equal prefix forbids a 1; at the end the value must not equal the bound
        (an equal prefix over all bits is forbidden)."""
        eq = None                                                       # None = the constant 1
        for i in reversed(range(len(bs))):
            cb = (bound >> i) & 1
            b = bs[i]
            if cb == 0:
                if eq is None:
                    cons.append(([(b, 1)], [(ONE, 1)], []))             # b = 0
                else:
                    cons.append(([(eq, 1)], [(b, 1)], []))              # eq * b = 0
                    # eq stays eq * (1 - b) = eq (since eq * b = 0)
            else:
                eq = b if eq is None else mul([(eq, 1)], [(b, 1)])      # eq' = eq * b
        if eq is None:
            cons.append(([(ONE, 1)], [(ONE, 1)], []))                   # bound = 0: unsatisfiable
        else:
            cons.append(([(eq, 1)], [(ONE, 1)], []))                    # not all bits equal: value != bound

    def point_add(x1, y1, x2, y2):
        """Twisted Edwards addition (babyjub.circom's BabyAdd: 6 rows)."""
        beta = mul([(x1, 1)], [(y2, 1)])
        gamma = mul([(y1, 1)], [(x2, 1)])
        delta = mul([(x1, -_BJ_A), (y1, 1)], [(x2, 1), (y2, 1)])
        tau = mul([(beta, 1)], [(gamma, 1)])
        x3v = (w[beta] + w[gamma]) * pow(1 + _BJ_D * w[tau], -1, R)
        x3 = new(x3v)
        cons.append(([(ONE, 1), (tau, _BJ_D)], [(x3, 1)], [(beta, 1), (gamma, 1)]))
        y3v = (w[delta] + _BJ_A * w[beta] - w[gamma]) * pow(1 - _BJ_D * w[tau], -1, R)
        y3 = new(y3v)
        cons.append(([(ONE, 1), (tau, -_BJ_D)], [(y3, 1)], [(delta, 1), (beta, _BJ_A), (gamma, -1)]))
        return x3, y3

    def select(bit, x_if1, y_if1, x_if0, y_if0):
        xs = new(w[x_if1] if w[bit] else w[x_if0])
        cons.append(([(bit, 1)], [(x_if1, 1), (x_if0, -1)], [(xs, 1), (x_if0, -1)]))
        ys = new(w[y_if1] if w[bit] else w[y_if0])
        cons.append(([(bit, 1)], [(y_if1, 1), (y_if0, -1)], [(ys, 1), (y_if0, -1)]))
        return xs, ys

    r8x, r8y, s_var = new(sig["R8x"]), new(sig["R8y"]), new(sig["S"])
    # S < subgroup order
    s_bits = bits_of(s_var, 253, sig["S"])
    less_than_const(s_bits, _BJ_SUBORDER)
    # h = Poseidon(R8x, R8y, Ax, Ay, M)
    st = [r8x, r8y, AX, AY, MSG, None]
    for i in range(_POS_RF + _POS_RP):
        st = [lin([(ONE, C[i])] + ([(v, 1)] if v is not None else [])) for v in st]
        full = i < _POS_RF // 2 or i >= _POS_RF // 2 + _POS_RP
        for k in range(t if full else 1):
            x = st[k]
            x2 = mul([(x, 1)], [(x, 1)])
            x4 = mul([(x2, 1)], [(x2, 1)])
            st[k] = mul([(x4, 1)], [(x, 1)])
        st = [lin([(st[b], Mx[a_][b]) for b in range(t)]) for a_ in range(t)]
    h_var = st[0]
    h_val = w[h_var]
    h_bits = bits_of(h_var, 254, h_val)
    less_than_const(h_bits, R)                                          # the strict decomposition: h's bits < r
    # 8 A (three doublings), nonzero x
    ax8, ay8 = AX, AY
    for _ in range(3):
        ax8, ay8 = point_add(ax8, ay8, ax8, ay8)
    inv = new(pow(w[ax8], -1, R) if w[ax8] else 0)
    cons.append(([(ax8, 1)], [(inv, 1)], [(ONE, 1)]))                   # x(8A) != 0
    # right2 = h * 8A: double-and-add from the top bit, the accumulator starts at the neutral element (0, 1)
    accx, accy = lin([]), lin([(ONE, 1)])
    for i in reversed(range(254)):
        accx, accy = point_add(accx, accy, accx, accy)
        sx, sy = point_add(accx, accy, ax8, ay8)
        accx, accy = select(h_bits[i], sx, sy, accx, accy)
    rx, ry = point_add(r8x, r8y, accx, accy)
    # left = S * B8 with the constant multiples 2^i B8: a conditional addition of a constant point per bit
    lx, ly = lin([]), lin([(ONE, 1)])
    Pi = _BJ_B8
    for i in range(253):
        bx, by = Pi
        beta = lin([(lx, by)])
        gamma = lin([(ly, bx)])
        tau = mul([(beta, 1)], [(gamma, 1)])
        delta_v = (-_BJ_A * w[lx] + w[ly]) * (bx + by) % R
        x3 = new((w[beta] + w[gamma]) * pow(1 + _BJ_D * w[tau], -1, R))
        cons.append(([(ONE, 1), (tau, _BJ_D)], [(x3, 1)], [(beta, 1), (gamma, 1)]))
        y3 = new((delta_v + _BJ_A * w[beta] - w[gamma]) * pow(1 - _BJ_D * w[tau], -1, R))
        cons.append(([(ONE, 1), (tau, -_BJ_D)], [(y3, 1)],
                     [(lx, -_BJ_A * (bx + by)), (ly, bx + by), (beta, _BJ_A), (gamma, -1)]))
        lx, ly = select(s_bits[i], x3, y3, lx, ly)
        Pi = babyjub_add(Pi, Pi)
    cons.append(([(lx, 1), (rx, -1)], [(ONE, 1)], []))                  # left == right
    cons.append(([(ly, 1), (ry, -1)], [(ONE, 1)], []))
    # normalise coefficients into [0, r)
    cons = [tuple([(v, c % R) for v, c in lc if c % R] for lc in con) for con in cons]
    return n_inputs, len(w) - n_inputs, cons, w
