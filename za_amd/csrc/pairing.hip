// Groth16 verification on the host CPU.
//
// Replaces bellman_ce::groth16::{prepare_verifying_key, verify_proof} as za uses them to self-verify every proof
// (prover/src/groth16/prover.rs:191-200) and in helper::verify (helper.rs:149-159).  Not on the GPU hot path -- but inside
// every prove() of the outer boundary (the reference self-verifies each proof), where the first version's 60-200 ms were a
// fifth of the call at 2^22.  Tower Fq2[v]/(v^3 - (9+u)), Fq6[w]/(w^2 - v).  Two implementations:
//   * the DEFINITION (round 1): points untwisted into Fq12, the Miller loop in affine coordinates over Fq12, Frobenius and the
//     final exponentiation as plain powers (q, (q^12 - 1)/r).  Kept as the cross-check (pairing_product_slow, tests).
//   * the one zg_verify runs (round 6): the Miller loops of all pairs in lockstep on the twist (affine over Fq2, ONE batched
//     inversion per step), sparse line values  -y_P + (lambda x_P) w + (y_T - lambda x_T) w^3,  Frobenius by constants
//     gamma^k = xi^(k (q-1)/6), and the final exponentiation as (q^6 - 1)(q^2 + 1) x the hard part by three powers of u.
//   e(A, B) = e(alpha, beta) * e(sum ic_i x_i, gamma) * e(C, delta)
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "formats.h"

namespace zg {
namespace {

// ---------------------------------------------------------------- Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + u
static Fq2 mul_xi(const Fq2& a) {   // (9 + u)(a0 + a1 u) = (9 a0 - a1) + (9 a1 + a0) u
    Fq n0 = a.c0.dbl().dbl().dbl() + a.c0, n1 = a.c1.dbl().dbl().dbl() + a.c1;
    return Fq2{n0 - a.c1, n1 + a.c0};
}
struct Fq6 {
    Fq2 c0, c1, c2;
    static Fq6 zero() { return Fq6{Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    static Fq6 one() { return Fq6{Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
    bool operator==(const Fq6& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
    Fq6 operator+(const Fq6& o) const { return Fq6{c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
    Fq6 operator-(const Fq6& o) const { return Fq6{c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
    Fq6 neg() const { return Fq6{c0.neg(), c1.neg(), c2.neg()}; }
    Fq6 operator*(const Fq6& o) const {   // schoolbook with v^3 = xi
        Fq2 t0 = c0 * o.c0, t1 = c1 * o.c1, t2 = c2 * o.c2;
        Fq2 r0 = t0 + mul_xi(c1 * o.c2 + c2 * o.c1);
        Fq2 r1 = c0 * o.c1 + c1 * o.c0 + mul_xi(t2);
        Fq2 r2 = c0 * o.c2 + c2 * o.c0 + t1;
        return Fq6{r0, r1, r2};
    }
    Fq6 mul_v() const { return Fq6{mul_xi(c2), c0, c1}; }   // times v
    Fq6 inv() const {
        Fq2 a = c0.sqr() - mul_xi(c1 * c2);
        Fq2 b = mul_xi(c2.sqr()) - c0 * c1;
        Fq2 c = c1.sqr() - c0 * c2;
        Fq2 d = (c0 * a + mul_xi(c2 * b + c1 * c)).inv();
        return Fq6{a * d, b * d, c * d};
    }
};
// ---------------------------------------------------------------- Fq12 = Fq6[w]/(w^2 - v)
struct Fq12 {
    Fq6 c0, c1;
    static Fq12 zero() { return Fq12{Fq6::zero(), Fq6::zero()}; }
    static Fq12 one() { return Fq12{Fq6::one(), Fq6::zero()}; }
    bool operator==(const Fq12& o) const { return c0 == o.c0 && c1 == o.c1; }
    Fq12 operator+(const Fq12& o) const { return Fq12{c0 + o.c0, c1 + o.c1}; }
    Fq12 operator-(const Fq12& o) const { return Fq12{c0 - o.c0, c1 - o.c1}; }
    Fq12 neg() const { return Fq12{c0.neg(), c1.neg()}; }
    Fq12 operator*(const Fq12& o) const {
        Fq6 aa = c0 * o.c0, bb = c1 * o.c1;
        return Fq12{aa + bb.mul_v(), (c0 + c1) * (o.c0 + o.c1) - aa - bb};
    }
    Fq12 sqr() const { return *this * *this; }
    Fq12 inv() const {
        Fq6 d = (c0 * c0 - (c1 * c1).mul_v()).inv();
        return Fq12{c0 * d, (c1 * d).neg()};
    }
    Fq12 pow(const uint64_t* e, int words) const {
        Fq12 acc = one();
        bool started = false;
        for (int i = words * 64 - 1; i >= 0; i--) {
            if (started) acc = acc.sqr();
            if ((e[i >> 6] >> (i & 63)) & 1) {
                acc = started ? acc * *this : *this;
                started = true;
            }
        }
        return acc;
    }
};
static Fq12 from_fq(const Fq& a) {
    Fq12 r = Fq12::zero();
    r.c0.c0.c0 = a;
    return r;
}
struct Pt12 {
    Fq12 x, y;
};
// D-type twist: (x', y') on y^2 = x^3 + 3/xi  ->  (x' w^2, y' w^3) = (x' v, y' v w) on y^2 = x^3 + 3 over Fq12
static Pt12 untwist(const G2Affine& q) {
    Pt12 r{Fq12::zero(), Fq12::zero()};
    r.x.c0.c1 = q.x;   // x' * v
    r.y.c1.c1 = q.y;   // y' * v * w
    return r;
}
static Pt12 embed(const G1Affine& p) { return Pt12{from_fq(p.x), from_fq(p.y)}; }

static Fq12 times3(const Fq12& a) { return a + a + a; }
static Pt12 pt_double(const Pt12& p) {
    Fq12 m = times3(p.x.sqr()) * (p.y + p.y).inv();
    Fq12 nx = m.sqr() - p.x - p.x;
    return Pt12{nx, m * (p.x - nx) - p.y};
}
static Pt12 pt_add(const Pt12& p, const Pt12& s) {   // distinct x (always the case in the Miller loop)
    Fq12 m = (s.y - p.y) * (s.x - p.x).inv();
    Fq12 nx = m.sqr() - p.x - s.x;
    return Pt12{nx, m * (p.x - nx) - p.y};
}
// line through p1, p2 evaluated at t
static Fq12 line(const Pt12& p1, const Pt12& p2, const Pt12& t) {
    if (!(p1.x == p2.x)) {
        Fq12 m = (p2.y - p1.y) * (p2.x - p1.x).inv();
        return m * (t.x - p1.x) - (t.y - p1.y);
    }
    if (p1.y == p2.y) {
        Fq12 m = times3(p1.x.sqr()) * (p1.y + p1.y).inv();
        return m * (t.x - p1.x) - (t.y - p1.y);
    }
    return t.x - p1.x;
}
static const uint64_t Q_WORDS[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull,
                                    0x30644e72e131a029ull};
static Fq12 frobenius(const Fq12& a) { return a.pow(Q_WORDS, 4); }

static Fq12 miller_loop(const G2Affine& qa, const G1Affine& pa) {
    if (qa.is_inf() || pa.is_inf()) return Fq12::one();
    const Pt12 q = untwist(qa), p = embed(pa);
    const unsigned __int128 ate = ((unsigned __int128)1 << 64) | 0x9d797039be763ba8ull;   // 6u + 2 = 29793968203157093288
    Pt12 r = q;
    Fq12 f = Fq12::one();
    for (int i = 63; i >= 0; i--) {
        f = f.sqr() * line(r, r, p);
        r = pt_double(r);
        if ((ate >> i) & 1) {
            f = f * line(r, q, p);
            r = pt_add(r, q);
        }
    }
    Pt12 q1{frobenius(q.x), frobenius(q.y)};
    Pt12 nq2{frobenius(q1.x), frobenius(q1.y).neg()};
    f = f * line(r, q1, p);
    r = pt_add(r, q1);
    f = f * line(r, nq2, p);
    return f;
}
static const uint64_t FINAL_EXP[44] = {
    0x86964b64ca86f120ull, 0x40a4efb7e54523a4ull, 0x837fa97896e84abbull,
    0x361102b6b9b2b918ull, 0xc0de81def35692daull, 0xbe04c7e8a6c3c760ull,
    0xd766f9c9d570bb7full, 0xc230974d83561841ull, 0x5bba1668c3be69a3ull,
    0x7f3811c410526294ull, 0x29baee7ddadda71cull, 0xbf813b8d145da900ull,
    0x641bbadf423f9a2cull, 0xa80bb4ea44eacc5eull, 0xcd65664814fde37cull,
    0x4a0364b9580291d2ull, 0xee93dfb10826f0ddull, 0x6b42db8dc5514724ull,
    0xbb10cf430b0f3785ull, 0x40494e406f804216ull, 0x55cfe107acf3aafbull,
    0x2088ec80e0ebae87ull, 0x846a3ed011a337a0ull, 0x48a45a4a1e3a5195ull,
    0xe5664568dfc50e16ull, 0xab6a41294c0cc4ebull, 0x82d0d602d268c7daull,
    0x6668449aed3cc48aull, 0x5062cd0fb2015dfcull, 0x7f2940a8b1ddb3d1ull,
    0x77f5b63a2a226448ull, 0xfef0781361e443aeull, 0xf977870e88d5c6c8ull,
    0x790364a61f676baaull, 0x5887e72eceaddea3ull, 0x1377e563a09a1b70ull,
    0x0c54efee1bd8c3b2ull, 0x3ec3d15ad524d8f7ull, 0xdaf15466b2383a5dull,
    0xe1e30a73bb94fec0ull, 0x6a1c71015f3f7be2ull, 0x842d43bf6369b1ffull,
    0x20fddadf107d20bcull, 0x0000002f4b6dc970ull};   // (q^12 - 1) / r

// ---------------------------------------------------------------- the fast path
static Fq2 fq2_pow_words(Fq2 a, const uint32_t e[8]) {
    Fq2 r = Fq2::one();
    for (int i = 255; i >= 0; i--) {
        r = r.sqr();
        if ((e[i >> 5] >> (i & 31)) & 1) r = r * a;
    }
    return r;
}
static Fq2 conj2(const Fq2& a) { return Fq2{a.c0, a.c1.neg()}; }
// gamma[k] = xi^(k (q - 1) / 6), k = 0 .. 5: what w^k picks up under Frobenius (w^6 = xi)
struct FrobConsts {
    Fq2 g[6];
    FrobConsts() {
        uint32_t w[8], e[8];
        for (int i = 0; i < 8; i++) w[i] = FqParams::MOD[i];
        w[0] -= 1;
        uint64_t rem = 0;
        for (int i = 7; i >= 0; i--) {   // (q - 1) / 6
            const uint64_t cur = (rem << 32) | w[i];
            e[i] = (uint32_t)(cur / 6);
            rem = cur % 6;
        }
        const Fq2 xi{Fq::from_u64(9), Fq::from_u64(1)};
        g[0] = Fq2::one();
        g[1] = fq2_pow_words(xi, e);
        for (int k = 2; k < 6; k++) g[k] = g[k - 1] * g[1];
    }
};
static const FrobConsts& frob_consts() {
    static const FrobConsts c;
    return c;
}
// a^q: coefficient of v^i w^j is conjugated and multiplied by gamma[2 i + j]
static Fq12 frob1(const Fq12& a) {
    const FrobConsts& F = frob_consts();
    Fq12 r;
    r.c0.c0 = conj2(a.c0.c0);
    r.c0.c1 = conj2(a.c0.c1) * F.g[2];
    r.c0.c2 = conj2(a.c0.c2) * F.g[4];
    r.c1.c0 = conj2(a.c1.c0) * F.g[1];
    r.c1.c1 = conj2(a.c1.c1) * F.g[3];
    r.c1.c2 = conj2(a.c1.c2) * F.g[5];
    return r;
}
static Fq12 frob2(const Fq12& a) { return frob1(frob1(a)); }
static Fq12 conj12(const Fq12& a) { return Fq12{a.c0, a.c1.neg()}; }   // a^(q^6)
static Fq12 pow_u(const Fq12& a) {                                      // a^u, u = 4965661367192848881
    const uint64_t u = 4965661367192848881ull;
    Fq12 acc = a;
    for (int i = 61; i >= 0; i--) {   // (bit 62 is the top bit of u)
        acc = acc.sqr();
        if ((u >> i) & 1) acc = acc * a;
    }
    return acc;
}
// f^((q^12 - 1) / r) as (q^6 - 1)(q^2 + 1) and the hard part (q^4 - q^2 + 1) / r through f^u, f^(u^2), f^(u^3) and
// Frobenius maps (the addition chain of Scott et al., "On the final exponentiation for calculating pairings on ordinary
// elliptic curves", as used for this curve by the common bn256 implementations)
static Fq12 final_exp_fast(const Fq12& in) {
    Fq12 t1 = conj12(in) * in.inv();
    t1 = frob2(t1) * t1;
    const Fq12 fp = frob1(t1), fp2 = frob2(t1), fp3 = frob1(fp2);
    const Fq12 fu = pow_u(t1), fu2 = pow_u(fu), fu3 = pow_u(fu2);
    const Fq12 y3 = conj12(frob1(fu)), fu2p = frob1(fu2), fu3p = frob1(fu3), y2 = frob2(fu2);
    const Fq12 y0 = fp * fp2 * fp3, y1 = conj12(t1), y5 = conj12(fu2), y4 = conj12(fu * fu2p), y6 = conj12(fu3 * fu3p);
    Fq12 t0 = y6.sqr() * y4 * y5;
    Fq12 s1 = y3 * y5 * t0;
    t0 = t0 * y2;
    s1 = (s1.sqr() * t0).sqr();
    t0 = s1 * y1;
    s1 = s1 * y0;
    return t0.sqr() * s1;
}
// psi on the twist: (x, y) -> (conj(x) gamma[2], conj(y) gamma[3])  (= Frobenius on the untwisted point)
static G2Affine psi_twist(const G2Affine& q) {
    const FrobConsts& F = frob_consts();
    return G2Affine{conj2(q.x) * F.g[2], conj2(q.y) * F.g[3]};
}
// the value at P of the line through T1, T2 (both on the twist, slope lambda already known) : -y_P + lambda x_P w + (y1 - lambda x1) w^3
static Fq12 line_value(const Fq2& lambda, const G2Affine& t1, const G1Affine& p) {
    Fq12 l = Fq12::zero();
    l.c0.c0 = Fq2{p.y.neg(), Fq::zero()};
    l.c1.c0 = Fq2{lambda.c0 * p.x, lambda.c1 * p.x};
    l.c1.c1 = t1.y - lambda * t1.x;
    return l;
}
// prod_i miller(Q_i, P_i) for up to MAXP pairs in lockstep: one shared squaring and ONE inversion (Montgomery's trick over
// the pairs) per step.  Pairs with an identity contribute 1.  The points met in the loop are multiples of Q below 6u + 2 < r,
// so for Q in G2 no step degenerates (T = +-Q, 2T = O); a degenerate step is reported (false) and the caller falls back on the
// definition.
static constexpr int MAXP = 4;
static bool batch_inverse(Fq2* d, int n) {   // d[i] <- 1 / d[i]; false if one of them is zero
    Fq2 pre[MAXP];
    Fq2 acc = Fq2::one();
    for (int i = 0; i < n; i++) {
        if (d[i].is_zero()) return false;
        pre[i] = acc;
        acc = acc * d[i];
    }
    Fq2 inv = acc.inv();
    for (int i = n - 1; i >= 0; i--) {
        const Fq2 di = d[i];
        d[i] = inv * pre[i];
        inv = inv * di;
    }
    return true;
}
static bool multi_miller_fast(const G2Affine* qs, const G1Affine* ps, int n_in, Fq12* out) {
    G2Affine q[MAXP], t[MAXP];
    G1Affine p[MAXP];
    int n = 0;
    for (int i = 0; i < n_in; i++)
        if (!qs[i].is_inf() && !ps[i].is_inf()) {
            q[n] = t[n] = qs[i];
            p[n] = ps[i];
            n++;
        }
    Fq12 f = Fq12::one();
    if (n == 0) {
        *out = f;
        return true;
    }
    const unsigned __int128 ate = ((unsigned __int128)1 << 64) | 0x9d797039be763ba8ull;   // 6u + 2
    Fq2 d[MAXP], lam[MAXP];
    auto dbl_step = [&]() {
        for (int i = 0; i < n; i++) d[i] = t[i].y.dbl();
        if (!batch_inverse(d, n)) return false;
        for (int i = 0; i < n; i++) {
            const Fq2 xx = t[i].x.sqr();
            lam[i] = (xx.dbl() + xx) * d[i];
            f = f * line_value(lam[i], t[i], p[i]);
            const Fq2 nx = lam[i].sqr() - t[i].x.dbl();
            t[i] = G2Affine{nx, lam[i] * (t[i].x - nx) - t[i].y};
        }
        return true;
    };
    auto add_step = [&](const G2Affine* s) {
        for (int i = 0; i < n; i++) d[i] = s[i].x - t[i].x;
        if (!batch_inverse(d, n)) return false;
        for (int i = 0; i < n; i++) {
            lam[i] = (s[i].y - t[i].y) * d[i];
            f = f * line_value(lam[i], t[i], p[i]);
            const Fq2 nx = lam[i].sqr() - t[i].x - s[i].x;
            t[i] = G2Affine{nx, lam[i] * (t[i].x - nx) - t[i].y};
        }
        return true;
    };
    for (int b = 63; b >= 0; b--) {
        f = f.sqr();
        if (!dbl_step()) return false;
        if ((ate >> b) & 1)
            if (!add_step(q)) return false;
    }
    G2Affine q1[MAXP], nq2[MAXP];
    for (int i = 0; i < n; i++) {
        q1[i] = psi_twist(q[i]);
        nq2[i] = psi_twist(q1[i]);
        nq2[i].y = nq2[i].y.neg();
    }
    if (!add_step(q1)) return false;
    // the last line only (the point is not needed any more)
    for (int i = 0; i < n; i++) d[i] = nq2[i].x - t[i].x;
    if (!batch_inverse(d, n)) return false;
    for (int i = 0; i < n; i++) f = f * line_value((nq2[i].y - t[i].y) * d[i], t[i], p[i]);
    *out = f;
    return true;
}

struct Vk {
    G1Affine alpha_g1, beta_g1, delta_g1;
    G2Affine beta_g2, gamma_g2, delta_g2;
    std::vector<G1Affine> ic;
};

}  // namespace

// prod_i e(P_i, Q_i) == 1 ?  slow = the definition (Fq12 points, plain powers); otherwise the fast path, which falls back
// on the definition if a step of its loops degenerates (it cannot for Q in G2).  gt_out (optional): the product's value.
bool pairing_product_is_one(const G2Affine* qs, const G1Affine* ps, int n, bool slow, uint8_t* gt_out) {
    Fq12 e;
    Fq12 f = Fq12::one();
    if (!slow && n <= MAXP && multi_miller_fast(qs, ps, n, &f)) {
        e = final_exp_fast(f);
    } else {
        f = Fq12::one();
        for (int i = 0; i < n; i++) f = f * miller_loop(qs[i], ps[i]);
        e = f.pow(FINAL_EXP, 44);
    }
    if (gt_out) memcpy(gt_out, &e, sizeof(Fq12));
    return e == Fq12::one();
}

static int parse_vk(Ctx* ctx, const uint8_t* buf, size_t len, Vk* vk) {
    ParsedParams pp;
    ZG_TRY(parse_bellman_vk(ctx, buf, len, &pp));
    vk->alpha_g1 = pp.alpha_g1; vk->beta_g1 = pp.beta_g1; vk->delta_g1 = pp.delta_g1;
    vk->beta_g2 = pp.beta_g2; vk->gamma_g2 = pp.gamma_g2; vk->delta_g2 = pp.delta_g2;
    vk->ic = pp.ic;
    return ZG_OK;
}

static void fq_hex(std::string* s, const Fq& m) {
    uint8_t be[32];
    fq_to_be(be, m);
    static const char* d = "0123456789abcdef";
    *s += "\"0x";
    for (int i = 0; i < 32; i++) { s->push_back(d[be[i] >> 4]); s->push_back(d[be[i] & 15]); }
    *s += "\"";
}
static int g1_json(std::string* s, const G1Affine& p) {
    if (p.is_inf()) return ZG_E_BAD_FORMAT;   // BadFormat("invalid point"), format.rs:49-50
    *s += "["; fq_hex(s, p.x); *s += ","; fq_hex(s, p.y); *s += "]";
    return ZG_OK;
}
static int g2_json(std::string* s, const G2Affine& p) {
    if (p.is_inf()) return ZG_E_BAD_FORMAT;
    *s += "[["; fq_hex(s, p.x.c0); *s += ","; fq_hex(s, p.x.c1); *s += "],["; fq_hex(s, p.y.c0); *s += ",";
    fq_hex(s, p.y.c1); *s += "]]";
    return ZG_OK;
}

}  // namespace zg

using namespace zg;

// proof: 256 bytes (A.x A.y B.x.c0 B.x.c1 B.y.c0 B.y.c1 C.x C.y canonical big-endian)
extern "C" int zg_verify(const uint8_t* vk_bytes, size_t vk_len, const uint8_t proof[256], const uint8_t* public_inputs32,
                         size_t n_public, int* ok) try {
    if (!vk_bytes || !proof || !ok || (n_public && !public_inputs32)) return ZG_E_INVALID_ARG;
    *ok = 0;
    Ctx tmp;
    Vk vk;
    ZG_TRY(parse_vk(&tmp, vk_bytes, vk_len, &vk));
    if (n_public + 1 != vk.ic.size()) return ZG_E_MALFORMED_VERIFYING_KEY;
    G1Affine A, C;
    G2Affine B;
    if (!fq_from_be(proof, &A.x) || !fq_from_be(proof + 32, &A.y) || !fq_from_be(proof + 64, &B.x.c0) ||
        !fq_from_be(proof + 96, &B.x.c1) || !fq_from_be(proof + 128, &B.y.c0) || !fq_from_be(proof + 160, &B.y.c1) ||
        !fq_from_be(proof + 192, &C.x) || !fq_from_be(proof + 224, &C.y))
        return ZG_E_BAD_FORMAT;
    if (!on_curve(A, Fq::from_u64(3)) || !on_curve(C, Fq::from_u64(3)) || !on_curve(B, g2_coeff_b()))
        return ZG_E_BAD_FORMAT;   // try_from_coordinates fails: "bad coordinates" (format.rs:41-44)
    // G1 has cofactor 1; a B outside the order-r subgroup of the twist is not a group element bellman can hold
    if (!g2_in_subgroup(B)) return ZG_E_BAD_FORMAT;
    G1XYZZ acc = G1XYZZ::from_affine(vk.ic[0]);
    for (size_t i = 0; i < n_public; i++) {
        uint32_t k[8];
        memcpy(k, public_inputs32 + 32 * i, 32);
        // Fr::from_str rejects values >= r ("bad format", format.rs:112-116): x and x + r would verify alike
        if (!scalar_is_canonical(k)) return ZG_E_BAD_FORMAT;
        acc = acc.add(G1XYZZ::from_affine(vk.ic[i + 1]).mul(k));
    }
    const G2Affine qs[4] = {B, vk.beta_g2, vk.gamma_g2, vk.delta_g2};
    const G1Affine ps[4] = {A, vk.alpha_g1.neg(), acc.to_affine().neg(), C.neg()};
    *ok = pairing_product_is_one(qs, ps, 4, /*slow=*/false) ? 1 : 0;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

// vk JSON as JsonVerifyingKey serialises it (prover/src/groth16/format.rs:130-167): alpha_g1, beta_g1, beta_g2,
// delta_g1, delta_g2, gamma_g2, ic, input_names.  names: n_names NUL-terminated strings back to back (may be NULL).
extern "C" int zg_vk_json(const uint8_t* vk_bytes, size_t vk_len, const char* names, size_t n_names, char* out, size_t cap) try {
    if (!vk_bytes || !out) return ZG_E_INVALID_ARG;
    Ctx tmp;
    Vk vk;
    ZG_TRY(parse_vk(&tmp, vk_bytes, vk_len, &vk));
    std::string j = "{\"alpha_g1\":";
    ZG_TRY(g1_json(&j, vk.alpha_g1));
    j += ",\"beta_g1\":"; ZG_TRY(g1_json(&j, vk.beta_g1));
    j += ",\"beta_g2\":"; ZG_TRY(g2_json(&j, vk.beta_g2));
    j += ",\"delta_g1\":"; ZG_TRY(g1_json(&j, vk.delta_g1));
    j += ",\"delta_g2\":"; ZG_TRY(g2_json(&j, vk.delta_g2));
    j += ",\"gamma_g2\":"; ZG_TRY(g2_json(&j, vk.gamma_g2));
    j += ",\"ic\":[";
    for (size_t i = 0; i < vk.ic.size(); i++) {
        if (i) j += ",";
        ZG_TRY(g1_json(&j, vk.ic[i]));
    }
    j += "],\"input_names\":[";
    const char* p = names;
    for (size_t i = 0; i < n_names && p; i++) {
        if (i) j += ",";
        j += "\"";
        for (; *p; p++) {
            if (*p == '"' || *p == '\\') j.push_back('\\');
            j.push_back(*p);
        }
        p++;
        j += "\"";
    }
    j += "]}";
    if (j.size() >= cap) return ZG_E_BUFFER_TOO_SMALL;
    memcpy(out, j.c_str(), j.size() + 1);
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
