"""Groth16 over BN254 with Python integers -- TEST ORACLE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restates the algorithm za reaches through `bellman_ce` (git dependency
github.com/adria0/bellman, branch test/affinecoords -- prover/Cargo.toml:19; source NOT
under /root/reference, no pinned commit: .gitignore:5-7 ignores Cargo.lock).  The
restatement follows the published bellman 0.1/bellman_ce 0.3 algorithm:

  generate_parameters   <- call site prover/src/groth16/prover.rs:122
  create_proof(r, s)    <- call site prover/src/groth16/prover.rs:173 (via create_random_proof)
  verify_proof          <- call site prover/src/groth16/prover.rs:191-200, helper.rs:149-159
  EvaluationDomain      (radix-2 FFT over Fr, generator 7, coset = generator)
  multiexp              (Pippenger, c = 3 if n < 32 else ceil(ln n), 0/1 special cases, density maps)
  Parameters::write/read, za's proving-key container  prover/src/groth16/format.rs:223-293
  proof.json            prover/src/groth16/format.rs:80-128

PARITY STATUS: "parity unpinned" at the byte level -- the reference has no golden proof
vectors (its tests draw r, s and the toxic waste from thread_rng, prover.rs:112,146,248).
A Groth16 proof is three canonical affine points determined uniquely by
(R1CS, witness, proving key, r, s), so any correct implementation gives identical bytes;
this oracle is checked (tests/test_oracle_groth16.py) by (i) the pairing verification
equation the reference's own tests use as their only criterion (prover.rs:291-305),
(ii) trapdoor identities, (iii) agreement between the windowed and naive MSMs.
"""
import json
import math
import struct

from .bn254 import (R, Q, Fq, Fq2, G1_GEN, G2_GEN, G1_B, G2_B, FR_S, FR_ROOT_OF_UNITY,
                    FR_GENERATOR, inv, pt_add, pt_mul, pt_neg, pt_double, jac_add, jac_double,
                    jac_from_affine, jac_to_affine, msm_naive, is_on_curve,
                    pairing_product_is_one)


# ----------------------------------------------------------------------------- R1CS
class R1CS:
    """num_inputs counts ONE (variable 0).  Variables are flat: inputs first, then aux.
    constraints: list of (A, B, C); each LC is a list of (var, coeff) meaning <A,w>*<B,w> = <C,w>.
    (za's QEQ a*b + c = 0 is handed to bellman as a*b = -c, prover/src/groth16/prover.rs:91-100.)"""

    def __init__(self, num_inputs, num_aux, constraints):
        self.num_inputs = num_inputs
        self.num_aux = num_aux
        self.constraints = constraints

    @property
    def num_vars(self):
        return self.num_inputs + self.num_aux

    def domain_log(self):
        n = len(self.constraints) + self.num_inputs
        e = 0
        while (1 << e) < n:
            e += 1
        return e

    def is_satisfied(self, w):
        for A, B, C in self.constraints:
            a = sum(c * w[v] for v, c in A) % R
            b = sum(c * w[v] for v, c in B) % R
            cc = sum(c * w[v] for v, c in C) % R
            if (a * b - cc) % R:
                return False
        return True


def example_factor_r1cs():
    """example/circuit.za: `p * q === r` with private p, q and public r.
    Signal order (compiler/src/evaluator/test.rs:739-768): one, outputs, public inputs, private inputs:
    one=0, r=1 (input), p=2, q=3 (aux).  QEQ: p*q + (-r) = 0  ->  bellman a=p, b=q, c=r."""
    return R1CS(2, 2, [([(2, 1)], [(3, 1)], [(1, 1)])])


# ----------------------------------------------------------------------------- EvaluationDomain
def bitreverse(n, l):
    r = 0
    for _ in range(l):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def serial_fft(a, omega, log_n):
    """In-place radix-2 DIT FFT exactly as bellman's serial_fft (bit-reversal then log_n stages)."""
    n = len(a)
    assert n == 1 << log_n
    for k in range(n):
        rk = bitreverse(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), R)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = a[k + j + m] * w % R
                a[k + j + m] = (a[k + j] - t) % R
                a[k + j] = (a[k + j] + t) % R
                w = w * w_m % R
        m *= 2


class Domain:
    def __init__(self, exp):
        if exp >= FR_S:
            raise ValueError("PolynomialDegreeTooLarge")
        self.exp = exp
        self.m = 1 << exp
        omega = FR_ROOT_OF_UNITY
        for _ in range(exp, FR_S):
            omega = omega * omega % R
        self.omega = omega
        self.omegainv = inv(omega, R)
        self.geninv = inv(FR_GENERATOR, R)
        self.minv = inv(self.m, R)

    def pad(self, coeffs):
        return list(coeffs) + [0] * (self.m - len(coeffs))

    def fft(self, a):
        serial_fft(a, self.omega, self.exp)

    def ifft(self, a):
        serial_fft(a, self.omegainv, self.exp)
        for i in range(self.m):
            a[i] = a[i] * self.minv % R

    def distribute_powers(self, a, g):
        u = 1
        for i in range(self.m):
            a[i] = a[i] * u % R
            u = u * g % R

    def coset_fft(self, a):
        self.distribute_powers(a, FR_GENERATOR)
        self.fft(a)

    def icoset_fft(self, a):
        self.ifft(a)
        self.distribute_powers(a, self.geninv)

    def z(self, tau):
        return (pow(tau, self.m, R) - 1) % R

    def divide_by_z_on_coset(self, a):
        i = inv(self.z(FR_GENERATOR), R)
        for k in range(self.m):
            a[k] = a[k] * i % R


def dft_naive(a, omega):
    n = len(a)
    return [sum(a[j] * pow(omega, i * j, R) for j in range(n)) % R for i in range(n)]


# ----------------------------------------------------------------------------- witness side (ProvingAssignment)
def eval_constraints(r1cs, w):
    """a[i]=<A_i,w>, b[i]=<B_i,w>, c[i]=<C_i,w> for every constraint plus one `input_i * 0 = 0`
    row per input; density bits are set for every variable MENTIONED in A (aux only) and in B
    (inputs and aux), exactly as bellman's ProvingAssignment::enforce/eval do."""
    ni = r1cs.num_inputs
    a, b, c = [], [], []
    a_aux_density = [False] * r1cs.num_aux
    b_input_density = [False] * ni
    b_aux_density = [False] * r1cs.num_aux
    for A, B, C in r1cs.constraints:
        s = 0
        for v, co in A:
            s += co * w[v]
            if v >= ni:
                a_aux_density[v - ni] = True
        a.append(s % R)
        s = 0
        for v, co in B:
            s += co * w[v]
            if v >= ni:
                b_aux_density[v - ni] = True
            else:
                b_input_density[v] = True
        b.append(s % R)
        c.append(sum(co * w[v] for v, co in C) % R)
    for i in range(ni):
        a.append(w[i] % R)
        b.append(0)
        c.append(0)
    return a, b, c, a_aux_density, b_input_density, b_aux_density


def h_coefficients(r1cs, w):
    """The m-1 coefficients of H(x) = (A(x)B(x) - C(x)) / Z(x), following bellman's sequence
    ifft, coset_fft (x3), mul, sub, divide_by_z_on_coset, icoset_fft, truncate."""
    a, b, c, *_ = eval_constraints(r1cs, w)
    d = Domain(r1cs.domain_log())
    a, b, c = d.pad(a), d.pad(b), d.pad(c)
    for v in (a, b, c):
        d.ifft(v)
        d.coset_fft(v)
    for i in range(d.m):
        a[i] = (a[i] * b[i] - c[i]) % R
    d.divide_by_z_on_coset(a)
    d.icoset_fft(a)
    assert a[d.m - 1] == 0, "witness does not satisfy the R1CS"
    return a[: d.m - 1]


# ----------------------------------------------------------------------------- H in evaluation form (checker for za_amd/csrc/heval.hip)
# bellman ends create_proof with h = icoset_fft(e), answer += multiexp(params.h, h).  icoset_fft is linear,
# h_i = (g^-i / m) sum_j omega^(-i j) e_j, so  sum_i h_i H_i = sum_j e_j H'_j  with  H'_j = sum_i omega^(-i j) (g^-i / m) H_i :
# the product moves the last transform from the scalars (every proof) to the bases (once per key).  The two functions
# below restate both sides by their definitions -- no butterflies -- in the order the GPU keeps them (position p holds
# index bitrev(p): what a decimation-in-frequency transform leaves).
def h_eval_scalars(r1cs, w):
    """e[p] = ((A B - C) / Z)(g omega^bitrev(p)): bellman's values just before its icoset_fft, permuted."""
    a, b, c, *_ = eval_constraints(r1cs, w)
    d = Domain(r1cs.domain_log())
    a, b, c = d.pad(a), d.pad(b), d.pad(c)
    for v in (a, b, c):
        d.ifft(v)
        d.coset_fft(v)
    for i in range(d.m):
        a[i] = (a[i] * b[i] - c[i]) % R
    d.divide_by_z_on_coset(a)
    return [a[bitreverse(p, d.exp)] for p in range(d.m)]


def h_eval_bases(h_points, log_m):
    """H'[p] = sum_(i < m-1) omega^(-i bitrev(p)) (g^-i / m) H_i, by the definition (m^2 / 2 scalar multiplications:
    small domains only)."""
    from .bn254 import pt_add, pt_mul
    d = Domain(log_m)
    m = d.m
    assert len(h_points) >= m - 1
    scaled, u = [], d.minv
    for i in range(m - 1):
        scaled.append(pt_mul(h_points[i], u))
        u = u * d.geninv % R
    out = []
    for p in range(m):
        j = bitreverse(p, log_m)
        acc = None
        wj = pow(d.omegainv, j, R)
        t = 1
        for i in range(m - 1):
            acc = pt_add(acc, pt_mul(scaled[i], t))
            t = t * wj % R
        out.append(acc)
    return out


def h_eval_scalars_without_c(r1cs, w):
    """e'[p] = (A B / Z)(g omega^bitrev(p)): the quotient's values WITHOUT its C term -- what za_amd's h_eval_dev leaves when
    the C rows are folded into the l multiexp (ZG_OPT_C_FOLD; four transforms: a and b only).  The quotient is linear in
    c:  e = e' - c(g omega^bitrev(p)) / Z(g)."""
    a, b, _c, *_ = eval_constraints(r1cs, w)
    d = Domain(r1cs.domain_log())
    a, b = d.pad(a), d.pad(b)
    for v in (a, b):
        d.ifft(v)
        d.coset_fft(v)
    for i in range(d.m):
        a[i] = a[i] * b[i] % R
    d.divide_by_z_on_coset(a)
    return [a[bitreverse(p, d.exp)] for p in range(d.m)]


def c_fold_bases(h_points, log_m):
    """C_j = -(1 / (m Z(g))) sum_(i < m-1) omega^(-i j) H_i in ROW order, by the definition: the bases that go with the
    values c_j of the R1CS rows' C sides (za_amd/csrc/heval.hip, c_fold).  c's values on the coset are the transform of its
    values on the domain, so  sum_p (c(g omega^bitrev(p)) / Z(g)) H'[p] = -sum_j c_j C_j  (i < m - 1 in both: the points the
    h query has)."""
    from .bn254 import pt_add, pt_mul
    d = Domain(log_m)
    m = d.m
    assert len(h_points) >= m - 1
    k = (R - inv(d.z(FR_GENERATOR), R) * d.minv % R) % R
    scaled = [pt_mul(h_points[i], k) for i in range(m - 1)]
    out = []
    for j in range(m):
        acc = None
        wj = pow(d.omegainv, j, R)
        t = 1
        for i in range(m - 1):
            acc = pt_add(acc, pt_mul(scaled[i], t))
            t = t * wj % R
        out.append(acc)
    return out


def h_eval_bases_fft(h_points, log_m):
    """The same array by a decimation-in-frequency transform over the group (natural order in, bit-reversed out) -- the
    schedule of heval.hip; tests/test_oracle_python.py checks it against h_eval_bases' definition."""
    from .bn254 import pt_add, pt_mul, pt_neg
    d = Domain(log_m)
    m = d.m
    a, u = [], d.minv
    for i in range(m - 1):
        a.append(pt_mul(h_points[i], u))
        u = u * d.geninv % R
    a.append(None)
    for s in range(log_m - 1, -1, -1):
        span = 1 << s
        for t in range(m // 2):
            low = t & (span - 1)
            i0 = ((t >> s) << (s + 1)) | low
            i1 = i0 | span
            x, y = a[i0], a[i1]
            a[i0] = pt_add(x, y)
            dd = pt_add(x, pt_neg(y))
            a[i1] = pt_mul(dd, pow(d.omegainv, low << (log_m - 1 - s), R)) if low else dd
    return a


def c_fold_bases_fft(h_points, log_m):
    """c_fold_bases by the same decimation-in-frequency transform over the group (heval.hip with c_fold: every point scaled
    by the constant -1 / (m Z(g)), no coset factor, bit-reversed out and put back in row order);
    tests/test_oracle_python.py checks it against the definition."""
    from .bn254 import pt_add, pt_mul, pt_neg
    d = Domain(log_m)
    m = d.m
    k = (R - inv(d.z(FR_GENERATOR), R) * d.minv % R) % R
    a = [pt_mul(h_points[i], k) for i in range(m - 1)] + [None]
    for s in range(log_m - 1, -1, -1):
        span = 1 << s
        for t in range(m // 2):
            low = t & (span - 1)
            i0 = ((t >> s) << (s + 1)) | low
            i1 = i0 | span
            x, y = a[i0], a[i1]
            a[i0] = pt_add(x, y)
            dd = pt_add(x, pt_neg(y))
            a[i1] = pt_mul(dd, pow(d.omegainv, low << (log_m - 1 - s), R)) if low else dd
    return [a[bitreverse(j, log_m)] for j in range(m)]


# ----------------------------------------------------------------------------- multiexp (bellman's Pippenger)
def multiexp_window(n):
    return 3 if n < 32 else int(math.ceil(math.log(n)))


def multiexp_bellman(bases, density, exponents):
    """bellman's multiexp: bases are consumed sequentially, one per SET density bit; per window
    (skip = 0, c, 2c, ...) exponent 0 is skipped, exponent 1 is added directly in the lowest
    window only, everything else goes into bucket[(exp >> skip) % 2^c - 1]; buckets are folded
    with a running sum; windows are combined high to low with c doublings each."""
    if density is None:
        density = [True] * len(exponents)
    assert sum(1 for d in density if d) == len(bases), "query size != density total"
    c = multiexp_window(len(exponents))
    windows = []
    skip = 0
    while skip < 254 + 2:          # Fr::NUM_BITS = 254; bellman stops once skip >= NUM_BITS
        if skip >= 254:
            break
        acc = None
        buckets = [None] * ((1 << c) - 1)
        bi = 0
        for e, d in zip(exponents, density):
            if not d:
                continue
            base = bases[bi]
            bi += 1
            if e == 0:
                continue
            if e == 1:
                if skip == 0:
                    acc = jac_add(acc, jac_from_affine(base))
                continue
            dg = (e >> skip) % (1 << c)
            if dg:
                buckets[dg - 1] = jac_add(buckets[dg - 1], jac_from_affine(base))
        running = None
        for bk in reversed(buckets):
            running = jac_add(running, bk)
            acc = jac_add(acc, running)
        windows.append(acc)
        skip += c
    total = None
    for wv in reversed(windows):
        for _ in range(c):
            total = jac_double(total)
        total = jac_add(total, wv)
    return jac_to_affine(total)


# ----------------------------------------------------------------------------- parameters
class VerifyingKey:
    def __init__(self, alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2, ic):
        self.alpha_g1, self.beta_g1, self.beta_g2 = alpha_g1, beta_g1, beta_g2
        self.gamma_g2, self.delta_g1, self.delta_g2, self.ic = gamma_g2, delta_g1, delta_g2, ic


class Parameters:
    def __init__(self, vk, h, l, a, b_g1, b_g2):
        self.vk, self.h, self.l, self.a, self.b_g1, self.b_g2 = vk, h, l, a, b_g1, b_g2


def generate_parameters(r1cs, alpha, beta, gamma, delta, tau, g1=G1_GEN, g2=G2_GEN):
    """bellman's generate_parameters with explicit toxic waste (generate_random_parameters draws
    g1, g2, alpha, beta, gamma, delta, tau from the rng).  Returns (Parameters, trapdoor-evals)."""
    ni, na = r1cs.num_inputs, r1cs.num_aux
    rows = [(A, B, C) for A, B, C in r1cs.constraints]
    for i in range(ni):
        rows.append(([(i, 1)], [], []))
    d = Domain(r1cs.domain_log())
    m = d.m
    assert len(rows) <= m
    # Lagrange basis at tau = ifft of the powers of tau
    lag = [pow(tau, i, R) for i in range(m)]
    d.ifft(lag)
    at = [0] * r1cs.num_vars
    bt = [0] * r1cs.num_vars
    ct = [0] * r1cs.num_vars
    for j, (A, B, C) in enumerate(rows):
        for v, co in A:
            at[v] = (at[v] + co * lag[j]) % R
        for v, co in B:
            bt[v] = (bt[v] + co * lag[j]) % R
        for v, co in C:
            ct[v] = (ct[v] + co * lag[j]) % R
    gi, di = inv(gamma, R), inv(delta, R)
    zt = d.z(tau)
    coeff = zt * di % R
    h = [pt_mul(g1, pow(tau, i, R) * coeff % R) for i in range(m - 1)]
    ext = []
    for v in range(r1cs.num_vars):
        e = (beta * at[v] + alpha * bt[v] + ct[v]) % R
        e = e * (gi if v < ni else di) % R
        ext.append(e)
    ic = [pt_mul(g1, ext[v]) for v in range(ni)]
    l = [pt_mul(g1, ext[v]) for v in range(ni, ni + na)]
    a_all = [pt_mul(g1, at[v]) for v in range(r1cs.num_vars)]
    b1_all = [pt_mul(g1, bt[v]) for v in range(r1cs.num_vars)]
    b2_all = [pt_mul(g2, bt[v]) for v in range(r1cs.num_vars)]
    vk = VerifyingKey(pt_mul(g1, alpha), pt_mul(g1, beta), pt_mul(g2, beta), pt_mul(g2, gamma),
                      pt_mul(g1, delta), pt_mul(g2, delta), ic)
    params = Parameters(vk, h, l,
                        [p for p in a_all if p is not None],
                        [p for p in b1_all if p is not None],
                        [p for p in b2_all if p is not None])
    trap = dict(at=at, bt=bt, ct=ct, zt=zt, alpha=alpha, beta=beta, gamma=gamma, delta=delta,
                tau=tau, m=m)
    return params, trap


def create_proof(r1cs, params, w, r, s, msm=None):
    """bellman's create_proof with fixed blinding factors.  w = full assignment (inputs then aux),
    w[0] must be 1.  `msm(bases, density_or_None, exponents)` defaults to bellman's Pippenger."""
    if msm is None:
        msm = multiexp_bellman
    ni, na = r1cs.num_inputs, r1cs.num_aux
    assert w[0] == 1 and len(w) == ni + na
    w = [x % R for x in w]
    _, _, _, a_aux_d, b_in_d, b_aux_d = eval_constraints(r1cs, w)
    hco = h_coefficients(r1cs, w)
    inputs, aux = w[:ni], w[ni:]
    vk = params.vk
    h_acc = msm(params.h[: len(hco)], None, hco)
    l_acc = msm(params.l[:na], None, aux)
    a_aux_total = sum(a_aux_d)
    a_inputs = msm(params.a[:ni], None, inputs)
    a_aux = msm(params.a[ni: ni + a_aux_total], a_aux_d, aux)
    b_in_total, b_aux_total = sum(b_in_d), sum(b_aux_d)
    b1_inputs = msm(params.b_g1[:b_in_total], b_in_d, inputs)
    b1_aux = msm(params.b_g1[b_in_total: b_in_total + b_aux_total], b_aux_d, aux)
    b2_inputs = msm(params.b_g2[:b_in_total], b_in_d, inputs)
    b2_aux = msm(params.b_g2[b_in_total: b_in_total + b_aux_total], b_aux_d, aux)
    assert len(params.a) == ni + a_aux_total, "a query does not match density"
    assert len(params.b_g1) == b_in_total + b_aux_total, "b query does not match density"
    if vk.delta_g1 is None or vk.delta_g2 is None:
        raise ValueError("UnexpectedIdentity")
    g_a = pt_add(pt_mul(vk.delta_g1, r), vk.alpha_g1)
    g_b = pt_add(pt_mul(vk.delta_g2, s), vk.beta_g2)
    g_c = pt_add(pt_add(pt_mul(vk.delta_g1, r * s % R), pt_mul(vk.alpha_g1, s)), pt_mul(vk.beta_g1, r))
    a_ans = pt_add(a_inputs, a_aux)
    g_a = pt_add(g_a, a_ans)
    g_c = pt_add(g_c, pt_mul(a_ans, s))
    b1_ans = pt_add(b1_inputs, b1_aux)
    b2_ans = pt_add(b2_inputs, b2_aux)
    g_b = pt_add(g_b, b2_ans)
    g_c = pt_add(g_c, pt_mul(b1_ans, r))
    g_c = pt_add(g_c, h_acc)
    g_c = pt_add(g_c, l_acc)
    return g_a, g_b, g_c


def msm_naive_density(bases, density, exponents):
    if density is None:
        return msm_naive(bases, exponents)
    sel = [e for e, d in zip(exponents, density) if d]
    return msm_naive(bases, sel)


def verify_proof(vk, proof, public_inputs):
    """bellman verify_proof: e(A,B) = e(alpha,beta) * e(sum ic_i x_i, gamma) * e(C, delta);
    public_inputs excludes ONE."""
    A, B, C = proof
    if len(public_inputs) + 1 != len(vk.ic):
        raise ValueError("MalformedVerifyingKey")
    acc = vk.ic[0]
    for x, p in zip(public_inputs, vk.ic[1:]):
        acc = pt_add(acc, pt_mul(p, x))
    return pairing_product_is_one([(A, B), (pt_neg(vk.alpha_g1), vk.beta_g2),
                                   (pt_neg(acc), vk.gamma_g2), (pt_neg(C), vk.delta_g2)])


def trapdoor_expected_proof(r1cs, trap, w, r, s):
    """With the toxic waste known, the proof's discrete logs are plain Fr arithmetic:
       A = alpha + a(tau) + r*delta, B = beta + b(tau) + s*delta,
       C = (h(tau) z(tau) + sum_aux w_i (beta a_i + alpha b_i + c_i))/delta + s*A + r*B - r*s*delta."""
    at, bt, ct = trap["at"], trap["bt"], trap["ct"]
    al, be, de, tau = trap["alpha"], trap["beta"], trap["delta"], trap["tau"]
    ni = r1cs.num_inputs
    a_tau = sum(x * y for x, y in zip(w, at)) % R
    b_tau = sum(x * y for x, y in zip(w, bt)) % R
    hco = h_coefficients(r1cs, w)
    h_tau = sum(c * pow(tau, i, R) for i, c in enumerate(hco)) % R
    A = (al + a_tau + r * de) % R
    B = (be + b_tau + s * de) % R
    aux_part = sum(w[v] * (be * at[v] + al * bt[v] + ct[v]) for v in range(ni, len(w))) % R
    C = ((h_tau * trap["zt"] + aux_part) * inv(de, R) + s * A + r * B - r * s * de) % R
    return A, B, C


# ----------------------------------------------------------------------------- encodings
def fq_be(x):
    return int(x).to_bytes(32, "big")


def g1_uncompressed(P):
    """pairing_ce bn256 G1Uncompressed: x||y big-endian; identity = 0x40 then zeros."""
    if P is None:
        return b"\x40" + b"\0" * 63
    return fq_be(P[0].v) + fq_be(P[1].v)


def g2_uncompressed(P):
    """pairing_ce bn256 G2Uncompressed: x.c1||x.c0||y.c1||y.c0 big-endian."""
    if P is None:
        return b"\x40" + b"\0" * 127
    x, y = P
    return fq_be(x.c1) + fq_be(x.c0) + fq_be(y.c1) + fq_be(y.c0)


def g1_from_uncompressed(b, checked=True):
    b = bytes(b)
    if b[0] & 0x80:
        raise ValueError("unexpected compression mode")
    if b[0] & 0x40:
        if any(b[1:]) or (b[0] & 0x3F):
            raise ValueError("unexpected information in infinity encoding")
        return None
    x, y = int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big")
    if x >= Q or y >= Q:
        raise ValueError("not in field")
    P = (Fq(x), Fq(y))
    if checked and not is_on_curve(P, G1_B):
        raise ValueError("not on curve")
    return P


def g2_from_uncompressed(b, checked=True):
    b = bytes(b)
    if b[0] & 0x80:
        raise ValueError("unexpected compression mode")
    if b[0] & 0x40:
        if any(b[1:]) or (b[0] & 0x3F):
            raise ValueError("unexpected information in infinity encoding")
        return None
    v = [int.from_bytes(b[32 * i: 32 * i + 32], "big") for i in range(4)]
    if any(x >= Q for x in v):
        raise ValueError("not in field")
    P = (Fq2(v[1], v[0]), Fq2(v[3], v[2]))
    if checked and not is_on_curve(P, G2_B):
        raise ValueError("not on curve")
    return P


def params_write(p):
    """bellman Parameters::write (format.rs:250 call site): vk, then h, l, a, b_g1, b_g2 each as
    u32be length + uncompressed points."""
    vk = p.vk
    out = [g1_uncompressed(vk.alpha_g1), g1_uncompressed(vk.beta_g1), g2_uncompressed(vk.beta_g2),
           g2_uncompressed(vk.gamma_g2), g1_uncompressed(vk.delta_g1), g2_uncompressed(vk.delta_g2),
           struct.pack(">I", len(vk.ic))]
    out += [g1_uncompressed(x) for x in vk.ic]
    for arr, enc in ((p.h, g1_uncompressed), (p.l, g1_uncompressed), (p.a, g1_uncompressed),
                     (p.b_g1, g1_uncompressed), (p.b_g2, g2_uncompressed)):
        out.append(struct.pack(">I", len(arr)))
        out += [enc(x) for x in arr]
    return b"".join(out)


def params_read(buf, checked=True):
    """bellman Parameters::read(reader, checked) (format.rs:285 call site); identity is rejected in
    h, l, a, b_g1, b_g2."""
    pos = 0

    def take(n):
        nonlocal pos
        if pos + n > len(buf):
            raise ValueError("IoError: unexpected EOF")
        b = buf[pos: pos + n]
        pos += n
        return b

    def g1():
        return g1_from_uncompressed(take(64), checked)

    def g2():
        return g2_from_uncompressed(take(128), checked)

    alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2 = g1(), g1(), g2(), g2(), g1(), g2()
    n = struct.unpack(">I", take(4))[0]
    ic = [g1() for _ in range(n)]

    def vec(rd):
        k = struct.unpack(">I", take(4))[0]
        out = []
        for _ in range(k):
            pt = rd()
            if pt is None:
                raise ValueError("point at infinity")
            out.append(pt)
        return out

    h, l, a, b1, b2 = vec(g1), vec(g1), vec(g1), vec(g1), vec(g2)
    return Parameters(VerifyingKey(alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2, ic),
                      h, l, a, b1, b2), pos


def _bincode_lc(lc):
    """bincode of za's LC(Vec<(SignalId, FS)>) (compiler/src/algebra/lc.rs:12, fs.rs:34):
    u64le count, then per term u64le signal id, then num-bigint 0.2 BigUint serde = Vec<u32>
    (u64le word count + u32le little-endian base-2^32 digits)."""
    out = [struct.pack("<Q", len(lc))]
    for sig, co in lc:
        co %= R
        words = []
        while co:
            words.append(co & 0xFFFFFFFF)
            co >>= 32
        out.append(struct.pack("<QQ", sig, len(words)))
        out.append(struct.pack("<%dI" % len(words), *words))
    return b"".join(out)


def pk_container_write(qeqs, ignore_signals, params_bytes, asts=b""):
    """za's proving-key container, prover/src/groth16/format.rs:223-252.  `qeqs` are za QEQs
    (a, b, c) with a*b + c = 0 over SIGNAL ids; `asts` is the opaque bincode AST blob (for an
    empty Vec<BodyElementP> that is u64le 0)."""
    if not asts:
        asts = struct.pack("<Q", 0)
    out = [struct.pack(">I", len(asts)), asts, struct.pack(">I", len(qeqs))]
    for a, b, c in qeqs:
        q = _bincode_lc(a) + _bincode_lc(b) + _bincode_lc(c)
        out += [struct.pack(">I", len(q)), q]
    out.append(struct.pack(">I", len(ignore_signals)))
    out += [struct.pack(">I", s) for s in ignore_signals]
    out.append(params_bytes)
    return b"".join(out)


def pk_container_read(buf):
    """Inverse of pk_container_write (format.rs:254-293).  Returns (qeqs, ignore_signals, params_offset)."""
    pos = 0

    def u32be():
        nonlocal pos
        v = struct.unpack_from(">I", buf, pos)[0]
        pos += 4
        return v

    def u64le():
        nonlocal pos
        v = struct.unpack_from("<Q", buf, pos)[0]
        pos += 8
        return v

    n = u32be()
    pos += n
    nq = u32be()
    qeqs = []
    for _ in range(nq):
        ln = u32be()
        end = pos + ln
        lcs = []
        for _ in range(3):
            cnt = u64le()
            lc = []
            for _ in range(cnt):
                sig = u64le()
                nw = u64le()
                co = 0
                for k in range(nw):
                    co |= struct.unpack_from("<I", buf, pos)[0] << (32 * k)
                    pos += 4
                lc.append((sig, co))
            lcs.append(lc)
        assert pos == end
        qeqs.append(tuple(lcs))
    ni = u32be()
    ignore = [u32be() for _ in range(ni)]
    return qeqs, ignore, pos


def fq_repr_str(x):
    """ff_ce FqRepr Display: 0x + 64 lowercase hex digits (format.rs:51,75-76 call into_repr().to_string())."""
    return "0x%064x" % x


def proof_json(proof, public_inputs):
    """JsonProofAndInput (format.rs:80-128): fields a, b, c, public_inputs; compact serde_json."""
    A, B, C = proof
    if A is None or B is None or C is None:
        raise ValueError('BadFormat("invalid point")')
    obj = {
        "a": [fq_repr_str(A[0].v), fq_repr_str(A[1].v)],
        "b": [[fq_repr_str(B[0].c0), fq_repr_str(B[0].c1)], [fq_repr_str(B[1].c0), fq_repr_str(B[1].c1)]],
        "c": [fq_repr_str(C[0].v), fq_repr_str(C[1].v)],
        "public_inputs": [str(int(x)) for x in public_inputs],
    }
    return json.dumps(obj, separators=(",", ":"))


def proof_bytes(proof):
    """256-byte wire form used by the C ABI: A.x||A.y||B.x.c0||B.x.c1||B.y.c0||B.y.c1||C.x||C.y, canonical BE."""
    A, B, C = proof
    return (fq_be(A[0].v) + fq_be(A[1].v) + fq_be(B[0].c0) + fq_be(B[0].c1) + fq_be(B[1].c0)
            + fq_be(B[1].c1) + fq_be(C[0].v) + fq_be(C[1].v))


# ----------------------------------------------------------------------------- vk JSON, JSON readers, Solidity export
def fs_parse(s):
    """FS::parse (compiler/src/algebra/fs.rs:43-55): `0x` + hex digits, else decimal digits; not reduced."""
    if s.startswith("0x"):
        body, base, what = s[2:], 16, "hexadecimal"
    else:
        body, base, what = s, 10, "decimal"
    digits = "0123456789abcdefABCDEF" if base == 16 else "0123456789"
    if not body or any(ch not in digits for ch in body):
        raise ValueError('Algebra(InvalidFormat("%s is not %s"))' % (s, what))
    return int(body, base)


def _str_to_fq(s):
    """str_to_fq (format.rs:33-36); Fq::from_str only exists below q (the reference unwrap()s)."""
    v = fs_parse(s)
    if v >= Q:
        raise ValueError('BadFormat("%s is not a field element")' % s)
    return v


def _g1_from_j(j):
    if not isinstance(j, list) or len(j) != 2 or not all(isinstance(x, str) for x in j):
        raise ValueError("Json")
    P = (Fq(_str_to_fq(j[0])), Fq(_str_to_fq(j[1])))
    if not is_on_curve(P, G1_B):
        raise ValueError('BadFormat("bad coordinates")')
    return P


def _g2_from_j(j):
    if not (isinstance(j, list) and len(j) == 2 and all(isinstance(x, list) and len(x) == 2 for x in j)):
        raise ValueError("Json")
    P = (Fq2(_str_to_fq(j[0][0]), _str_to_fq(j[0][1])), Fq2(_str_to_fq(j[1][0]), _str_to_fq(j[1][1])))
    if not is_on_curve(P, G2_B):
        raise ValueError('BadFormat("bad coordinates")')
    return P


def _g1_j(P):
    if P is None:
        raise ValueError('BadFormat("invalid point")')
    return [fq_repr_str(P[0].v), fq_repr_str(P[1].v)]


def _g2_j(P):
    if P is None:
        raise ValueError('BadFormat("invalid point")')
    return [[fq_repr_str(P[0].c0), fq_repr_str(P[0].c1)], [fq_repr_str(P[1].c0), fq_repr_str(P[1].c1)]]


def vk_json(vk, input_names=()):
    """JsonVerifyingKey::from_bellman(..).with_input_names(..).to_json() (format.rs:130-167,191-193): serde emits the
    fields in declaration order alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, gamma_g2, ic, input_names."""
    obj = {"alpha_g1": _g1_j(vk.alpha_g1), "beta_g1": _g1_j(vk.beta_g1), "beta_g2": _g2_j(vk.beta_g2),
           "delta_g1": _g1_j(vk.delta_g1), "delta_g2": _g2_j(vk.delta_g2), "gamma_g2": _g2_j(vk.gamma_g2),
           "ic": [_g1_j(p) for p in vk.ic], "input_names": list(input_names)}
    return json.dumps(obj, separators=(",", ":"), ensure_ascii=False)


def vk_from_json(text):
    """JsonVerifyingKey::from_json + to_bellman (format.rs:169-194)."""
    j = json.loads(text)
    for k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2", "gamma_g2", "ic", "input_names"):
        if k not in j:
            raise ValueError("Json(missing field `%s`)" % k)
    return VerifyingKey(_g1_from_j(j["alpha_g1"]), _g1_from_j(j["beta_g1"]), _g2_from_j(j["beta_g2"]),
                        _g2_from_j(j["gamma_g2"]), _g1_from_j(j["delta_g1"]), _g2_from_j(j["delta_g2"]),
                        [_g1_from_j(p) for p in j["ic"]])


def vk_bytes(vk):
    """bellman VerifyingKey::write = the prefix of Parameters::write."""
    return b"".join([g1_uncompressed(vk.alpha_g1), g1_uncompressed(vk.beta_g1), g2_uncompressed(vk.beta_g2),
                     g2_uncompressed(vk.gamma_g2), g1_uncompressed(vk.delta_g1), g2_uncompressed(vk.delta_g2),
                     struct.pack(">I", len(vk.ic))] + [g1_uncompressed(x) for x in vk.ic])


def proof_from_json(text):
    """JsonProofAndInput::json_to_bellman (format.rs:101-121): coordinates through FS::parse (decimal or 0x hex),
    public inputs through Fr::from_str (canonical decimal below r only) -> BadFormat("bad format")."""
    j = json.loads(text)
    for k in ("a", "b", "c", "public_inputs"):
        if k not in j:
            raise ValueError("Json(missing field `%s`)" % k)
    proof = (_g1_from_j(j["a"]), _g2_from_j(j["b"]), _g1_from_j(j["c"]))
    pub = []
    for s in j["public_inputs"]:
        ok = isinstance(s, str) and s != "" and all(ch in "0123456789" for ch in s) and (s == "0" or s[0] != "0")
        if not ok or int(s) >= R:
            raise ValueError('BadFormat("bad format")')
        pub.append(int(s))
    return proof, pub


def rust_debug_vec_str(names):
    """format!("{:?}", Vec<String>) for the printable-ASCII names circuits have."""
    def one(s):
        return '"' + s.replace("\\", "\\\\").replace('"', '\\"') + '"'
    return "[" + ", ".join(one(s) for s in names) + "]"


def solidity_verifier(template, vk, input_names):
    """generate_solidity (prover/src/groth16/ethereum.rs:216-261) applied to `template` (the CONTRACT_TEMPLATE text,
    ethereum.rs:8-214 -- the caller supplies it; this module does not carry it).  G2 coordinates are printed c1
    first (:231-236)."""
    def s1(P):
        return "%s,%s" % (fq_repr_str(P[0].v), fq_repr_str(P[1].v))

    def s2(P):
        return "[%s,%s],[%s,%s]" % (fq_repr_str(P[0].c1), fq_repr_str(P[0].c0), fq_repr_str(P[1].c1), fq_repr_str(P[1].c0))
    c = template
    c = c.replace("<%vk_a%>", s1(vk.alpha_g1))
    c = c.replace("<%vk_b%>", s2(vk.beta_g2))
    c = c.replace("<%vk_gamma%>", s2(vk.gamma_g2))
    c = c.replace("<%vk_delta%>", s2(vk.delta_g2))
    c = c.replace("<%vk_inputs_length%>", "%d" % len(input_names))
    c = c.replace("<%vk_inputs%>", rust_debug_vec_str(input_names))
    c = c.replace("<%vk_gammaABC_length%>", "%d" % len(vk.ic))
    c = c.replace("<%vk_gammaABC_pts%>", "\n".join("vk.gammaABC[%d] = Pairing.G1Point(%s);" % (i, s1(x))
                                                     for i, x in enumerate(vk.ic)))
    return c
