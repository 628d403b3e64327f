/* Short-Weierstrass (a = 0) Jacobian arithmetic, instantiated for G1 (over Fq) and G2 (over Fq2).
 * TEST ORACLE, NOT PRODUCT CODE (see field.h).
 *
 * Restates pairing_ce::bn256's CurveProjective::{double, add_assign, add_assign_mixed,
 * mul_assign, into_affine} as used by bellman's multiexp and create_proof (call site
 * prover/src/groth16/prover.rs:173).  Include with:  F (field elem type), FN(x) (field fn
 * prefix macro), G(x) (group prefix macro).
 */
typedef struct { F x, y; int inf; } G(aff);
typedef struct { F x, y, z; } G(jac);     /* identity: z == 0 */

static inline void G(jac_set_inf)(G(jac)* p) { memset(p, 0, sizeof(*p)); }
static inline int G(jac_is_inf)(const G(jac)* p) { return FN(is_zero)(&p->z); }

static inline void G(jac_from_aff)(G(jac)* r, const G(aff)* a) {
    if (a->inf) { G(jac_set_inf)(r); return; }
    r->x = a->x; r->y = a->y; r->z = FN(ONE);
}

static void G(jac_double)(G(jac)* r, const G(jac)* p) {
    if (G(jac_is_inf)(p)) { *r = *p; return; }
    F A, B, C, D, E, Fv, t;
    FN(sqr)(&A, &p->x);
    FN(sqr)(&B, &p->y);
    FN(sqr)(&C, &B);
    FN(add)(&t, &p->x, &B);
    FN(sqr)(&t, &t);
    FN(sub)(&t, &t, &A);
    FN(sub)(&t, &t, &C);
    FN(dbl)(&D, &t);
    FN(dbl)(&E, &A);
    FN(add)(&E, &E, &A);
    FN(sqr)(&Fv, &E);
    F z3;
    FN(mul)(&z3, &p->y, &p->z);
    FN(dbl)(&z3, &z3);
    F x3;
    FN(sub)(&x3, &Fv, &D);
    FN(sub)(&x3, &x3, &D);
    F y3;
    FN(sub)(&y3, &D, &x3);
    FN(mul)(&y3, &y3, &E);
    FN(dbl)(&C, &C);
    FN(dbl)(&C, &C);
    FN(dbl)(&C, &C);
    FN(sub)(&y3, &y3, &C);
    r->x = x3; r->y = y3; r->z = z3;
}

static void G(jac_add)(G(jac)* r, const G(jac)* p, const G(jac)* q) {
    if (G(jac_is_inf)(p)) { *r = *q; return; }
    if (G(jac_is_inf)(q)) { *r = *p; return; }
    F z1z1, z2z2, u1, u2, s1, s2;
    FN(sqr)(&z1z1, &p->z);
    FN(sqr)(&z2z2, &q->z);
    FN(mul)(&u1, &p->x, &z2z2);
    FN(mul)(&u2, &q->x, &z1z1);
    FN(mul)(&s1, &p->y, &q->z);
    FN(mul)(&s1, &s1, &z2z2);
    FN(mul)(&s2, &q->y, &p->z);
    FN(mul)(&s2, &s2, &z1z1);
    if (FN(eq)(&u1, &u2)) {
        if (FN(eq)(&s1, &s2)) { G(jac_double)(r, p); return; }
        G(jac_set_inf)(r);
        return;
    }
    F h, rr, hh, hhh, v, x3, y3, z3, t;
    FN(sub)(&h, &u2, &u1);
    FN(sub)(&rr, &s2, &s1);
    FN(sqr)(&hh, &h);
    FN(mul)(&hhh, &hh, &h);
    FN(mul)(&v, &u1, &hh);
    FN(sqr)(&x3, &rr);
    FN(sub)(&x3, &x3, &hhh);
    FN(sub)(&x3, &x3, &v);
    FN(sub)(&x3, &x3, &v);
    FN(sub)(&t, &v, &x3);
    FN(mul)(&y3, &rr, &t);
    FN(mul)(&t, &s1, &hhh);
    FN(sub)(&y3, &y3, &t);
    FN(mul)(&z3, &p->z, &q->z);
    FN(mul)(&z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}

static void G(jac_add_mixed)(G(jac)* r, const G(jac)* p, const G(aff)* q) {
    if (q->inf) { *r = *p; return; }
    if (G(jac_is_inf)(p)) { G(jac_from_aff)(r, q); return; }
    F z1z1, u2, s2;
    FN(sqr)(&z1z1, &p->z);
    FN(mul)(&u2, &q->x, &z1z1);
    FN(mul)(&s2, &q->y, &p->z);
    FN(mul)(&s2, &s2, &z1z1);
    if (FN(eq)(&p->x, &u2)) {
        if (FN(eq)(&p->y, &s2)) { G(jac_double)(r, p); return; }
        G(jac_set_inf)(r);
        return;
    }
    F h, rr, hh, hhh, v, x3, y3, z3, t;
    FN(sub)(&h, &u2, &p->x);
    FN(sub)(&rr, &s2, &p->y);
    FN(sqr)(&hh, &h);
    FN(mul)(&hhh, &hh, &h);
    FN(mul)(&v, &p->x, &hh);
    FN(sqr)(&x3, &rr);
    FN(sub)(&x3, &x3, &hhh);
    FN(sub)(&x3, &x3, &v);
    FN(sub)(&x3, &x3, &v);
    FN(sub)(&t, &v, &x3);
    FN(mul)(&y3, &rr, &t);
    FN(mul)(&t, &p->y, &hhh);
    FN(sub)(&y3, &y3, &t);
    FN(mul)(&z3, &p->z, &h);
    r->x = x3; r->y = y3; r->z = z3;
}

static void G(jac_to_aff)(G(aff)* r, const G(jac)* p) {
    if (G(jac_is_inf)(p)) { memset(r, 0, sizeof(*r)); r->inf = 1; return; }
    F zi, zi2, zi3;
    FN(inv)(&zi, &p->z);
    FN(sqr)(&zi2, &zi);
    FN(mul)(&zi3, &zi2, &zi);
    FN(mul)(&r->x, &p->x, &zi2);
    FN(mul)(&r->y, &p->y, &zi3);
    r->inf = 0;
}

/* k is a canonical (non-Montgomery) 256-bit scalar, little-endian limbs */
static void G(jac_mul)(G(jac)* r, const G(jac)* p, const uint64_t k[4]) {
    G(jac) acc;
    G(jac_set_inf)(&acc);
    for (int i = 255; i >= 0; i--) {
        G(jac_double)(&acc, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) G(jac_add)(&acc, &acc, p);
    }
    *r = acc;
}

/* One window of bellman's multiexp_inner: exponent 0 skipped, exponent 1 added directly (lowest
 * window only), others into bucket[digit-1]; running-sum fold.  Bases are consumed sequentially
 * for SET density bits only. */
static void G(multiexp_window)(G(jac)* out, const G(aff)* bases, const uint8_t* density,
                               const uint64_t* exps /* n x 4 */, size_t n, unsigned skip, unsigned c) {
    size_t nb = ((size_t)1 << c) - 1;
    G(jac)* buckets = (G(jac)*)calloc(nb, sizeof(G(jac)));
    G(jac) acc;
    G(jac_set_inf)(&acc);
    size_t bi = 0;
    for (size_t i = 0; i < n; i++) {
        if (density && !density[i]) continue;
        const G(aff)* base = &bases[bi++];
        const uint64_t* e = exps + 4 * i;
        if ((e[0] | e[1] | e[2] | e[3]) == 0) continue;
        if (e[0] == 1 && (e[1] | e[2] | e[3]) == 0) {
            if (skip == 0) G(jac_add_mixed)(&acc, &acc, base);
            continue;
        }
        /* (exp >> skip) % 2^c */
        unsigned w = skip >> 6, b = skip & 63;
        uint64_t d = e[w] >> b;
        if (b + c > 64 && w + 1 < 4) d |= e[w + 1] << (64 - b);
        d &= ((uint64_t)1 << c) - 1;
        if (d) G(jac_add_mixed)(&buckets[d - 1], &buckets[d - 1], base);
    }
    G(jac) running;
    G(jac_set_inf)(&running);
    for (size_t k = nb; k-- > 0;) {
        G(jac_add)(&running, &running, &buckets[k]);
        G(jac_add)(&acc, &acc, &running);
    }
    free(buckets);
    *out = acc;
}
