/* BN254 field arithmetic, 4 x u64 Montgomery -- TEST ORACLE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library
 * built from this directory; za_amd/ never links or calls it.
 *
 * Restates the limb-level arithmetic of ff_ce 0.7.1's derived PrimeField (4 x u64, Montgomery,
 * R = 2^256) that the reference uses through pairing_ce::bn256::{Fr, Fq}
 * (prover/Cargo.toml:19-25; imports at prover/src/groth16/prover.rs:10-17).  The sources are
 * not vendored under /root/reference; constants are pinned to compiler/src/algebra/fs.rs:15-16
 * (r) and prover/src/groth16/ethereum.rs:37 (q).  Parity: unpinned at byte level (see
 * oracle/groth16.py header); this C code is itself checked against the Python big-int oracle.
 */
#ifndef ORACLE_FIELD_H
#define ORACLE_FIELD_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;

#define DEFINE_FIELD(P, M0, M1, M2, M3, NINV)                                                   \
    static const fe P##_MOD = {{M0, M1, M2, M3}};                                               \
    static fe P##_ONE, P##_R2;                                                                  \
    static inline int P##_is_zero(const fe* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; } \
    static inline int P##_eq(const fe* a, const fe* b) { return memcmp(a, b, 32) == 0; }        \
    static inline int P##_geq_mod(const fe* a) {                                                \
        for (int i = 3; i >= 0; i--) {                                                          \
            if (a->l[i] > P##_MOD.l[i]) return 1;                                               \
            if (a->l[i] < P##_MOD.l[i]) return 0;                                               \
        }                                                                                       \
        return 1;                                                                               \
    }                                                                                           \
    static inline void P##_sub_mod(fe* a) {                                                     \
        u128 br = 0;                                                                            \
        for (int i = 0; i < 4; i++) {                                                           \
            u128 t = (u128)a->l[i] - P##_MOD.l[i] - (uint64_t)br;                               \
            a->l[i] = (uint64_t)t;                                                              \
            br = (t >> 64) & 1;                                                                 \
        }                                                                                       \
    }                                                                                           \
    static inline void P##_add(fe* r, const fe* a, const fe* b) {                               \
        u128 c = 0;                                                                             \
        for (int i = 0; i < 4; i++) {                                                           \
            c += (u128)a->l[i] + b->l[i];                                                       \
            r->l[i] = (uint64_t)c;                                                              \
            c >>= 64;                                                                           \
        }                                                                                       \
        if (P##_geq_mod(r)) P##_sub_mod(r);                                                     \
    }                                                                                           \
    static inline void P##_sub(fe* r, const fe* a, const fe* b) {                               \
        u128 br = 0;                                                                            \
        uint64_t t[4];                                                                          \
        for (int i = 0; i < 4; i++) {                                                           \
            u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br;                                    \
            t[i] = (uint64_t)d;                                                                 \
            br = (d >> 64) & 1;                                                                 \
        }                                                                                       \
        if (br) {                                                                               \
            u128 c = 0;                                                                         \
            for (int i = 0; i < 4; i++) {                                                       \
                c += (u128)t[i] + P##_MOD.l[i];                                                 \
                t[i] = (uint64_t)c;                                                             \
                c >>= 64;                                                                       \
            }                                                                                   \
        }                                                                                       \
        memcpy(r->l, t, 32);                                                                    \
    }                                                                                           \
    static inline void P##_neg(fe* r, const fe* a) {                                            \
        if (P##_is_zero(a)) { *r = *a; return; }                                                \
        fe z = {{0, 0, 0, 0}};                                                                  \
        P##_sub(r, &z, a);                                                                      \
    }                                                                                           \
    static inline void P##_dbl(fe* r, const fe* a) { P##_add(r, a, a); }                        \
    /* CIOS Montgomery product a*b*2^-256 mod p */                                              \
    static inline void P##_mul(fe* r, const fe* a, const fe* b) {                               \
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};                                                     \
        for (int i = 0; i < 4; i++) {                                                           \
            u128 c = 0;                                                                         \
            for (int j = 0; j < 4; j++) {                                                       \
                c += (u128)a->l[j] * b->l[i] + t[j];                                            \
                t[j] = (uint64_t)c;                                                             \
                c >>= 64;                                                                       \
            }                                                                                   \
            c += t[4];                                                                          \
            t[4] = (uint64_t)c;                                                                 \
            t[5] = (uint64_t)(c >> 64);                                                         \
            uint64_t m = t[0] * (uint64_t)NINV;                                                 \
            c = (u128)m * P##_MOD.l[0] + t[0];                                                  \
            c >>= 64;                                                                           \
            for (int j = 1; j < 4; j++) {                                                       \
                c += (u128)m * P##_MOD.l[j] + t[j];                                             \
                t[j - 1] = (uint64_t)c;                                                         \
                c >>= 64;                                                                       \
            }                                                                                   \
            c += t[4];                                                                          \
            t[3] = (uint64_t)c;                                                                 \
            t[4] = t[5] + (uint64_t)(c >> 64);                                                  \
        }                                                                                       \
        memcpy(r->l, t, 32);                                                                    \
        if (t[4] || P##_geq_mod(r)) P##_sub_mod(r);                                             \
    }                                                                                           \
    static inline void P##_sqr(fe* r, const fe* a) { P##_mul(r, a, a); }                        \
    static inline void P##_to_mont(fe* r, const fe* a) { P##_mul(r, a, &P##_R2); }              \
    static inline void P##_from_mont(fe* r, const fe* a) {                                      \
        fe one = {{1, 0, 0, 0}};                                                                \
        P##_mul(r, a, &one);                                                                    \
    }                                                                                           \
    static void P##_pow(fe* r, const fe* a, const uint64_t e[4]) {                              \
        fe acc = P##_ONE, base = *a;                                                            \
        for (int i = 0; i < 256; i++) {                                                         \
            if ((e[i >> 6] >> (i & 63)) & 1) P##_mul(&acc, &acc, &base);                        \
            P##_sqr(&base, &base);                                                              \
        }                                                                                       \
        *r = acc;                                                                               \
    }                                                                                           \
    static void P##_inv(fe* r, const fe* a) {                                                   \
        uint64_t e[4] = {P##_MOD.l[0] - 2, P##_MOD.l[1], P##_MOD.l[2], P##_MOD.l[3]};           \
        P##_pow(r, a, e);                                                                       \
    }                                                                                           \
    static void P##_init(void) {                                                                \
        /* ONE = 2^256 mod p by 256 modular doublings of 1; R2 = 2^512 mod p by 256 more */     \
        fe x = {{1, 0, 0, 0}};                                                                  \
        for (int i = 0; i < 256; i++) P##_add(&x, &x, &x);                                      \
        P##_ONE = x;                                                                            \
        for (int i = 0; i < 256; i++) P##_add(&x, &x, &x);                                      \
        P##_R2 = x;                                                                             \
    }

DEFINE_FIELD(fq, 0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL,
             0x30644e72e131a029ULL, 0x87d20782e4866389ULL)
DEFINE_FIELD(fr, 0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
             0x30644e72e131a029ULL, 0xc2e1f593efffffffULL)

/* ---- Fq2 = Fq[u]/(u^2+1) ---- */
typedef struct { fe c0, c1; } fe2;
static fe2 fq2_ONE;
static inline int fq2_is_zero(const fe2* a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
static inline int fq2_eq(const fe2* a, const fe2* b) { return fq_eq(&a->c0, &b->c0) && fq_eq(&a->c1, &b->c1); }
static inline void fq2_add(fe2* r, const fe2* a, const fe2* b) { fq_add(&r->c0, &a->c0, &b->c0); fq_add(&r->c1, &a->c1, &b->c1); }
static inline void fq2_sub(fe2* r, const fe2* a, const fe2* b) { fq_sub(&r->c0, &a->c0, &b->c0); fq_sub(&r->c1, &a->c1, &b->c1); }
static inline void fq2_neg(fe2* r, const fe2* a) { fq_neg(&r->c0, &a->c0); fq_neg(&r->c1, &a->c1); }
static inline void fq2_dbl(fe2* r, const fe2* a) { fq2_add(r, a, a); }
static inline void fq2_mul(fe2* r, const fe2* a, const fe2* b) {
    fe aa, bb, s, t, o;
    fq_mul(&aa, &a->c0, &b->c0);
    fq_mul(&bb, &a->c1, &b->c1);
    fq_add(&s, &a->c0, &a->c1);
    fq_add(&t, &b->c0, &b->c1);
    fq_mul(&o, &s, &t);
    fq_sub(&o, &o, &aa);
    fq_sub(&o, &o, &bb);
    fq_sub(&r->c0, &aa, &bb);
    r->c1 = o;
}
static inline void fq2_sqr(fe2* r, const fe2* a) {
    fe s, d, p;
    fq_add(&s, &a->c0, &a->c1);
    fq_sub(&d, &a->c0, &a->c1);
    fq_mul(&p, &a->c0, &a->c1);
    fq_mul(&r->c0, &s, &d);
    fq_add(&r->c1, &p, &p);
}
static void fq2_inv(fe2* r, const fe2* a) {
    fe t0, t1;
    fq_sqr(&t0, &a->c0);
    fq_sqr(&t1, &a->c1);
    fq_add(&t0, &t0, &t1);
    fq_inv(&t0, &t0);
    fq_mul(&r->c0, &a->c0, &t0);
    fq_mul(&t1, &a->c1, &t0);
    fq_neg(&r->c1, &t1);
}
static void fields_init(void) {
    fq_init();
    fr_init();
    fq2_ONE.c0 = fq_ONE;
    memset(&fq2_ONE.c1, 0, 32);
}
#endif
