"""BN254 (a.k.a. BN128 / bn256) big-integer arithmetic -- TEST ORACLE, NOT PRODUCT CODE.

This file is test infrastructure.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it; the product path (za_amd/) never does.

It restates, with plain Python integers, the arithmetic that the reference
(adria0/za) reaches through its un-vendored git dependencies
  pairing_ce  (github.com/adria0/pairing  branch feature/affinecoords)  -- prover/Cargo.toml:20
  ff_ce 0.7.1                                                           -- prover/Cargo.toml:23-25
Those sources are NOT under /root/reference, so the curve constants are pinned
to the ones the reference does carry in-tree:
  r  = compiler/src/algebra/fs.rs:15-16   (also prover/src/groth16/ethereum.rs:173)
  q  = prover/src/groth16/ethereum.rs:37
  G1 generator (1, 2)                      prover/src/groth16/ethereum.rs:22
  G2 generator                             prover/src/groth16/ethereum.rs:28-31
PARITY STATUS: byte-level parity with bellman is UNPINNED (the reference holds no
golden proof bytes: prover/src/groth16/prover.rs:225-373 only self-verifies with
thread_rng).  What *is* pinned is the reference's own test property -- a proof
verifies under the Groth16 pairing equation, and fails for a wrong public input
(prover.rs:291-305) -- which `pairing()` below lets the tests check independently.
"""

# ----------------------------------------------------------------------------- constants
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr modulus
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fq modulus
BN_U = 4965661367192848881
assert R == 36 * BN_U**4 + 36 * BN_U**3 + 18 * BN_U**2 + 6 * BN_U + 1
assert Q == 36 * BN_U**4 + 36 * BN_U**3 + 24 * BN_U**2 + 6 * BN_U + 1

FR_S = 28                                   # 2-adicity of r-1
FR_T = (R - 1) >> FR_S
FR_GENERATOR = 7                            # bellman/ff_ce: Fr::multiplicative_generator()
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, FR_T, R)   # order exactly 2^28
assert pow(FR_ROOT_OF_UNITY, 1 << 27, R) == R - 1

MONT_R = 1 << 256                           # Montgomery radix used by the 4x64 / 8x32 limb code


def inv(a, m):
    return pow(a % m, -1, m)


# ----------------------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1)
class Fq2:
    __slots__ = ("c0", "c1")

    def __init__(self, c0, c1=0):
        self.c0 = c0 % Q
        self.c1 = c1 % Q

    def __add__(s, o):
        return Fq2(s.c0 + o.c0, s.c1 + o.c1)

    def __sub__(s, o):
        return Fq2(s.c0 - o.c0, s.c1 - o.c1)

    def __neg__(s):
        return Fq2(-s.c0, -s.c1)

    def __mul__(s, o):
        if isinstance(o, int):
            return Fq2(s.c0 * o, s.c1 * o)
        return Fq2(s.c0 * o.c0 - s.c1 * o.c1, s.c0 * o.c1 + s.c1 * o.c0)

    __rmul__ = __mul__

    def __eq__(s, o):
        return s.c0 == o.c0 and s.c1 == o.c1

    def __hash__(s):
        return hash((s.c0, s.c1))

    def is_zero(s):
        return s.c0 == 0 and s.c1 == 0

    def inv(s):
        d = inv(s.c0 * s.c0 + s.c1 * s.c1, Q)
        return Fq2(s.c0 * d, -s.c1 * d)

    def __repr__(s):
        return "Fq2(%d,%d)" % (s.c0, s.c1)


class Fq:
    """Thin wrapper so that G1 and G2 share one generic curve implementation."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v % Q

    def __add__(s, o):
        return Fq(s.v + o.v)

    def __sub__(s, o):
        return Fq(s.v - o.v)

    def __neg__(s):
        return Fq(-s.v)

    def __mul__(s, o):
        if isinstance(o, int):
            return Fq(s.v * o)
        return Fq(s.v * o.v)

    __rmul__ = __mul__

    def __eq__(s, o):
        return s.v == o.v

    def __hash__(s):
        return hash(s.v)

    def is_zero(s):
        return s.v == 0

    def inv(s):
        return Fq(inv(s.v, Q))

    def __repr__(s):
        return "Fq(%d)" % s.v


# ----------------------------------------------------------------------------- short Weierstrass y^2 = x^3 + b
G1_B = Fq(3)
G2_B = Fq2(3, 0) * Fq2(9, 1).inv()          # twist: y^2 = x^3 + 3/(9+u)
G1_GEN = (Fq(1), Fq(2))
G2_GEN = (
    Fq2(10857046999023057135944570762232829481370756359578518086990519993285655852781,
        11559732032986387107991004021392285783925812861821192530917403151452391805634),
    Fq2(8495653923123431417604973247489272438418190587263600148770280649306958101930,
        4082367875863433681332203403145435568316851327593401208105741076214120093531),
)
# affine points are (x, y) tuples; the identity is None.


def is_on_curve(P, b):
    if P is None:
        return True
    x, y = P
    return y * y == x * x * x + b


def pt_neg(P):
    if P is None:
        return None
    return (P[0], -P[1])


def pt_double(P):
    if P is None:
        return None
    x, y = P
    if y.is_zero():
        return None
    m = (x * x * 3) * (y * 2).inv()
    nx = m * m - x - x
    return (nx, m * (x - nx) - y)


def pt_add(P, S):
    if P is None:
        return S
    if S is None:
        return P
    x1, y1 = P
    x2, y2 = S
    if x1 == x2:
        if y1 == y2:
            return pt_double(P)
        return None
    m = (y2 - y1) * (x2 - x1).inv()
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


# Jacobian versions (no inversion per step) used for scalar multiplication / naive MSM.
def _one_like(x):
    return Fq(1) if isinstance(x, Fq) else Fq2(1, 0)


def jac_from_affine(P):
    if P is None:
        return None
    return (P[0], P[1], _one_like(P[0]))


def jac_to_affine(J):
    if J is None:
        return None
    X, Y, Z = J
    if Z.is_zero():
        return None
    zi = Z.inv()
    zi2 = zi * zi
    return (X * zi2, Y * zi2 * zi)


def jac_double(J):
    if J is None:
        return None
    X, Y, Z = J
    if Y.is_zero() or Z.is_zero():
        return None
    A = X * X
    B = Y * Y
    C = B * B
    t = X + B
    D = (t * t - A - C) * 2
    E = A * 3
    F = E * E
    X3 = F - D * 2
    Y3 = E * (D - X3) - C * 8
    Z3 = Y * Z * 2
    return (X3, Y3, Z3)


def jac_add(J1, J2):
    if J1 is None:
        return J2
    if J2 is None:
        return J1
    X1, Y1, Z1 = J1
    X2, Y2, Z2 = J2
    if Z1.is_zero():
        return J2
    if Z2.is_zero():
        return J1
    Z1Z1 = Z1 * Z1
    Z2Z2 = Z2 * Z2
    U1 = X1 * Z2Z2
    U2 = X2 * Z1Z1
    S1 = Y1 * Z2 * Z2Z2
    S2 = Y2 * Z1 * Z1Z1
    if U1 == U2:
        if S1 == S2:
            return jac_double(J1)
        return None
    H = U2 - U1
    Rr = S2 - S1
    HH = H * H
    HHH = H * HH
    V = U1 * HH
    X3 = Rr * Rr - HHH - V * 2
    Y3 = Rr * (V - X3) - S1 * HHH
    Z3 = Z1 * Z2 * H
    return (X3, Y3, Z3)


def pt_mul(P, k, order=R):
    """k*P for an affine point P (double-and-add, Jacobian inside)."""
    k %= order
    if P is None or k == 0:
        return None
    acc = None
    base = jac_from_affine(P)
    for bit in bin(k)[2:]:
        acc = jac_double(acc)
        if bit == "1":
            acc = jac_add(acc, base)
    return jac_to_affine(acc)


def msm_naive(points, scalars):
    """sum_i scalars[i]*points[i]; the definition every MSM implementation must equal."""
    acc = None
    for P, k in zip(points, scalars):
        k %= R
        if P is None or k == 0:
            continue
        t = None
        b = jac_from_affine(P)
        for bit in bin(k)[2:]:
            t = jac_double(t)
            if bit == "1":
                t = jac_add(t, b)
        acc = jac_add(acc, t)
    return jac_to_affine(acc)


assert is_on_curve(G1_GEN, G1_B)
assert is_on_curve(G2_GEN, G2_B)


# ----------------------------------------------------------------------------- G2 membership (order-r subgroup of the twist)
# pairing_ce's G2Uncompressed::into_affine -- what Parameters::read(checked = true) pays per b_g2 point
# (prover/src/groth16/format.rs:285) -- tests [r]P = O.  The twist E'(Fq2) has order r * (2q - r), so that IS a test:
# g2_in_subgroup_by_order below is the definition.  g2_in_subgroup_by_psi is the endomorphism form the device kernel
# uses (za_amd/csrc/keyload.hip): with psi = twist o Frobenius o untwist,
#     P in G2  <=>  [u + 1]P + psi([u]P) + psi^2([u]P) = psi^3([2u]P)             (u = BN_U)
# "=>": psi acts on G2 as multiplication by q, and f(q) = (u+1) + u q + u q^2 - 2u q^3 = 0 (mod r).
# "<=": psi^2 - t psi + q = 0 on E'(Fq2) (t = 6u^2 + 1), so f(psi) has a norm N = a^2 + a b t + b^2 q in Z (f = a + b psi
# after reduction) with N P = O whenever f(psi) P = O; gcd(N, 2q - r) = 1, hence the order of such a P divides r.
# Both facts are integers and are asserted in tests/test_oracle_python.py::test_psi_membership_test_is_exact.
def fq2_pow(a, e):
    r = Fq2(1, 0)
    while e:
        if e & 1:
            r = r * a
        a = a * a
        e >>= 1
    return r


def fq_sqrt(a):
    a %= Q
    s = pow(a, (Q + 1) // 4, Q)          # q = 3 (mod 4)
    return s if s * s % Q == a else None


def fq2_sqrt(a):
    """A square root of a in Fq2 = Fq[u]/(u^2 + 1), or None."""
    if a.is_zero():
        return Fq2(0, 0)
    n = fq_sqrt(a.c0 * a.c0 + a.c1 * a.c1)
    if n is None:
        return None
    for s in (n, Q - n):
        x0 = fq_sqrt((a.c0 + s) * inv(2, Q))
        if x0:
            r = Fq2(x0, a.c1 * inv(2 * x0, Q))
            if r * r == a:
                return r
    if a.c1 == 0:                        # a = -(x1^2): a pure-imaginary root
        x1 = fq_sqrt(-a.c0)
        if x1 is not None:
            return Fq2(0, x1)
    return None


def fq2_conj(a):
    return Fq2(a.c0, -a.c1)


_XI = Fq2(9, 1)
PSI_X = fq2_pow(_XI, (Q - 1) // 3)       # xi^((q-1)/3)
PSI_Y = fq2_pow(_XI, (Q - 1) // 2)       # xi^((q-1)/2)


def g2_psi(P):
    """The untwist-Frobenius-twist endomorphism of E'(Fq2): (x, y) -> (conj(x) xi^((q-1)/3), conj(y) xi^((q-1)/2))."""
    if P is None:
        return None
    return (fq2_conj(P[0]) * PSI_X, fq2_conj(P[1]) * PSI_Y)


def g2_in_subgroup_by_order(P):
    return P is None or pt_mul(P, R, order=1 << 300) is None


def g2_in_subgroup_by_psi(P):
    if P is None:
        return True
    uP = pt_mul(P, BN_U, order=1 << 300)
    lhs = pt_add(pt_add(pt_add(uP, P), g2_psi(uP)), g2_psi(g2_psi(uP)))
    rhs = pt_double(g2_psi(g2_psi(g2_psi(uP))))
    return lhs == rhs if (lhs is None) == (rhs is None) else False


def g2_twist_point(seed):
    """A point of E'(Fq2) with x derived from `seed` -- almost never in the order-r subgroup (cofactor 2q - r)."""
    k = seed
    while True:
        x = Fq2(k, seed * 7 + 1)
        y = fq2_sqrt(x * x * x + G2_B)
        if y is not None:
            return (x, y)
        k += 1


# ----------------------------------------------------------------------------- Fq12 and the optimal ate pairing
# Degree-12 extension as Fq[w]/(w^12 - 18 w^6 + 82) (w^6 = 9 + u).  Plain schoolbook
# polynomial arithmetic; slow (a verification takes a few seconds) and only used by tests.
_FQ12_MOD = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]


class Fq12:
    __slots__ = ("c",)

    def __init__(self, c):
        assert len(c) == 12
        self.c = [x % Q for x in c]

    @staticmethod
    def one():
        return Fq12([1] + [0] * 11)

    def __add__(s, o):
        return Fq12([a + b for a, b in zip(s.c, o.c)])

    def __sub__(s, o):
        return Fq12([a - b for a, b in zip(s.c, o.c)])

    def __neg__(s):
        return Fq12([-a for a in s.c])

    def __mul__(s, o):
        if isinstance(o, int):
            return Fq12([a * o for a in s.c])
        b = [0] * 23
        for i, x in enumerate(s.c):
            if x:
                for j, y in enumerate(o.c):
                    b[i + j] += x * y
        for top in range(22, 11, -1):
            t = b[top]
            if t:
                b[top] = 0
                b[top - 6] += 18 * t
                b[top - 12] -= 82 * t
        return Fq12(b[:12])

    __rmul__ = __mul__

    def __eq__(s, o):
        return s.c == o.c

    def is_zero(s):
        return all(x == 0 for x in s.c)

    def __pow__(s, e):
        res = Fq12.one()
        base = s
        while e:
            if e & 1:
                res = res * base
            base = base * base
            e >>= 1
        return res

    def inv(s):
        # extended Euclid on polynomials over Fq
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = s.c + [0], [x % Q for x in _FQ12_MOD] + [1]

        def deg(p):
            d = len(p) - 1
            while d and p[d] == 0:
                d -= 1
            return d

        def polydiv(a, b):
            da, db = deg(a), deg(b)
            t = list(a)
            o = [0] * len(a)
            bi = inv(b[db], Q)
            for i in range(da - db, -1, -1):
                o[i] = (o[i] + t[db + i] * bi) % Q
                for c in range(db + 1):
                    t[c + i] = (t[c + i] - o[i] * b[c]) % Q
            return o[: deg(o) + 1]

        while deg(low):
            r = polydiv(high, low)
            r += [0] * (13 - len(r))
            nm, new = list(hm), list(high)
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] -= lm[i] * r[j]
                    new[i + j] -= low[i] * r[j]
            nm = [x % Q for x in nm]
            new = [x % Q for x in new]
            lm, low, hm, high = nm, new, lm, low
        li = inv(low[0], Q)
        return Fq12([x * li for x in lm[:12]])


_W = Fq12([0, 1] + [0] * 10)
_W2 = _W * _W
_W3 = _W2 * _W
ATE_LOOP_COUNT = 6 * BN_U + 2
assert ATE_LOOP_COUNT == 29793968203157093288
_LOG_ATE = 63


def _twist(P):
    """Map a G2 point (over Fq2) onto the curve y^2 = x^3 + 3 over Fq12."""
    if P is None:
        return None
    x, y = P
    xc = [(x.c0 - 9 * x.c1) % Q, x.c1]
    yc = [(y.c0 - 9 * y.c1) % Q, y.c1]
    nx = Fq12([xc[0]] + [0] * 5 + [xc[1]] + [0] * 5)
    ny = Fq12([yc[0]] + [0] * 5 + [yc[1]] + [0] * 5)
    return (nx * _W2, ny * _W3)


def _cast_g1(P):
    x, y = P
    return (Fq12([x.v] + [0] * 11), Fq12([y.v] + [0] * 11))


def _f12_double(P):
    x, y = P
    m = (x * x * 3) * (y * 2).inv()
    nx = m * m - x - x
    return (nx, m * (x - nx) - y)


def _f12_add(P, S):
    x1, y1 = P
    x2, y2 = S
    if x1 == x2:
        if y1 == y2:
            return _f12_double(P)
        return None
    m = (y2 - y1) * (x2 - x1).inv()
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


def _linefunc(P1, P2, T):
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if not (x1 == x2):
        m = (y2 - y1) * (x2 - x1).inv()
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1 * 3) * (y1 * 2).inv()
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def _frob(x):
    return x ** Q


def miller_loop(Qpt, Ppt):
    """Miller loop of the optimal ate pairing (no final exponentiation).  Qpt in G2, Ppt in G1."""
    if Qpt is None or Ppt is None:
        return Fq12.one()
    Qt = _twist(Qpt)
    Pt = _cast_g1(Ppt)
    Rp = Qt
    f = Fq12.one()
    for i in range(_LOG_ATE, -1, -1):
        f = f * f * _linefunc(Rp, Rp, Pt)
        Rp = _f12_double(Rp)
        if ATE_LOOP_COUNT & (1 << i):
            f = f * _linefunc(Rp, Qt, Pt)
            Rp = _f12_add(Rp, Qt)
    Q1 = (_frob(Qt[0]), _frob(Qt[1]))
    nQ2 = (_frob(Q1[0]), -_frob(Q1[1]))
    f = f * _linefunc(Rp, Q1, Pt)
    Rp = _f12_add(Rp, Q1)
    f = f * _linefunc(Rp, nQ2, Pt)
    return f


def final_exponentiation(f):
    return f ** ((Q ** 12 - 1) // R)


def pairing(Qpt, Ppt):
    return final_exponentiation(miller_loop(Qpt, Ppt))


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1, with one shared final exponentiation.  pairs = [(G1 pt, G2 pt)]."""
    f = Fq12.one()
    for P, Qp in pairs:
        f = f * miller_loop(Qp, P)
    return final_exponentiation(f) == Fq12.one()


# ----------------------------------------------------------------------------- limb / byte helpers shared by tests
def to_limbs64(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def le32(x):
    """32-byte little-endian encoding (== 4 x u64 LE limbs == 8 x u32 LE limbs)."""
    return int(x).to_bytes(32, "little")


def from_le32(b):
    return int.from_bytes(bytes(b), "little")


def mont(x, m):
    """x -> x*2^256 mod m (Montgomery form used by the limb implementations)."""
    return (x * MONT_R) % m


def unmont(x, m):
    return (x * inv(MONT_R, m)) % m


def g1_to_mont_bytes(P):
    """64-byte device layout of an affine G1 base: x||y, Montgomery form, LE limbs; identity = all zero."""
    if P is None:
        return b"\0" * 64
    return le32(mont(P[0].v, Q)) + le32(mont(P[1].v, Q))


def g2_to_mont_bytes(P):
    """128-byte device layout of an affine G2 base: x.c0||x.c1||y.c0||y.c1 (Montgomery, LE)."""
    if P is None:
        return b"\0" * 128
    x, y = P
    return b"".join(le32(mont(v, Q)) for v in (x.c0, x.c1, y.c0, y.c1))


def g1_from_mont_bytes(b):
    b = bytes(b)
    if b == b"\0" * 64:
        return None
    return (Fq(unmont(from_le32(b[:32]), Q)), Fq(unmont(from_le32(b[32:64]), Q)))


def g2_from_mont_bytes(b):
    b = bytes(b)
    if b == b"\0" * 128:
        return None
    v = [unmont(from_le32(b[32 * i:32 * i + 32]), Q) for i in range(4)]
    return (Fq2(v[0], v[1]), Fq2(v[2], v[3]))
