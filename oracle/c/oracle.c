/* CPU restatement of the Groth16/BN254 prover hot path -- TEST ORACLE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library
 * (oracle/_build/liboracle.so); za_amd/ never links or calls it.
 *
 * Follows the algorithm of bellman_ce (git dep github.com/adria0/bellman, branch
 * test/affinecoords, prover/Cargo.toml:19 -- not vendored, no pinned commit) behind za's call
 * `create_random_proof(circuit, params, rng)` at prover/src/groth16/prover.rs:173:
 *   - multiexp: window c = 3 if n < 32 else ceil(ln n), one task per window, 0/1 exponent special
 *     cases, density maps, windows folded high->low with c doublings;
 *   - EvaluationDomain: radix-2 in-place FFT (bit reversal + log n stages), parallel_fft split
 *     into 2^log_cpus sub-FFTs, coset generator 7;
 *   - ProvingAssignment::enforce/eval: a,b,c rows and the three density trackers, plus one
 *     `input_i * 0 = 0` row per input;
 *   - create_proof's final combination with explicit r, s.
 * Parity status: unpinned at byte level (no golden vectors exist in the reference); pinned to
 * oracle/groth16.py, whose proofs satisfy the pairing equation (tests/test_oracle_*.py).
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>

#include "field.h"

/* ---- G1 ---- */
#define F fe
#define FN(x) fq_##x
#define G(x) g1_##x
#include "ec_tmpl.h"
#undef F
#undef FN
#undef G
/* ---- G2 ---- */
#define F fe2
#define FN(x) fq2_##x
#define G(x) g2_##x
#include "ec_tmpl.h"
#undef F
#undef FN
#undef G

#define ORC_OK 0
#define ORC_E_UNEXPECTED_IDENTITY 5
#define ORC_E_DEGREE_TOO_LARGE 4
#define ORC_E_QUERY_SIZE 20

static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static fe FR_ROOT; /* 2^28-th root of unity, Montgomery */
static fe FR_GEN;  /* 7, Montgomery */

static void do_init(void) {
    fields_init();
    /* root_of_unity = 7^((r-1)/2^28) */
    fe seven = {{7, 0, 0, 0}};
    fr_to_mont(&FR_GEN, &seven);
    /* t = (r-1) >> 28 */
    uint64_t t[4];
    fe rm1 = fr_MOD;
    rm1.l[0] -= 1;
    for (int i = 0; i < 4; i++) t[i] = (rm1.l[i] >> 28) | (i < 3 ? rm1.l[i + 1] << 36 : 0);
    fr_pow(&FR_ROOT, &FR_GEN, t);
}
void orc_init(void) { pthread_once(&g_once, do_init); }

/* ------------------------------------------------------------------ tiny thread pool */
typedef void (*task_fn)(void* ctx, size_t idx);
typedef struct { task_fn fn; void* ctx; size_t n; size_t next; pthread_mutex_t mu; } pool_job;
static void* pool_worker(void* p) {
    pool_job* j = (pool_job*)p;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        size_t i = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (i >= j->n) break;
        j->fn(j->ctx, i);
    }
    return NULL;
}
static void parallel_for(size_t n, int threads, task_fn fn, void* ctx) {
    if (threads <= 1 || n <= 1) {
        for (size_t i = 0; i < n; i++) fn(ctx, i);
        return;
    }
    pool_job j = {fn, ctx, n, 0, PTHREAD_MUTEX_INITIALIZER};
    if ((size_t)threads > n) threads = (int)n;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, pool_worker, &j);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
}

/* ------------------------------------------------------------------ byte <-> struct */
static void g1_aff_load(g1_aff* p, const uint8_t* b) {
    memcpy(&p->x, b, 32);
    memcpy(&p->y, b + 32, 32);
    p->inf = fq_is_zero(&p->x) && fq_is_zero(&p->y);
}
static void g1_aff_store(uint8_t* b, const g1_aff* p) {
    if (p->inf) { memset(b, 0, 64); return; }
    memcpy(b, &p->x, 32);
    memcpy(b + 32, &p->y, 32);
}
static void g2_aff_load(g2_aff* p, const uint8_t* b) {
    memcpy(&p->x, b, 64);
    memcpy(&p->y, b + 64, 64);
    p->inf = fq2_is_zero(&p->x) && fq2_is_zero(&p->y);
}
static void g2_aff_store(uint8_t* b, const g2_aff* p) {
    if (p->inf) { memset(b, 0, 128); return; }
    memcpy(b, &p->x, 64);
    memcpy(b + 64, &p->y, 64);
}

/* ------------------------------------------------------------------ multiexp */
static unsigned multiexp_c(size_t n) { return n < 32 ? 3u : (unsigned)ceil(log((double)n)); }

/* A multiexp is prepared (bases loaded, window sums allocated), its windows run as independent tasks, and the
 * window sums are folded high to low.  create_proof submits ALL its multiexps to one Worker pool before waiting for
 * any of them (bellman's futures), so mx_run_all() puts the windows of every prepared multiexp into one task list, in
 * submission order. */
#define DEFINE_MULTIEXP(GP, STRIDE)                                                              \
    typedef struct { GP##_aff* bases; const uint8_t* density; const uint64_t* exps; size_t n;    \
                     unsigned c; size_t nw; GP##_jac* win; } GP##_mx_ctx;                        \
    static void GP##_mx_task(void* p, size_t w) {                                                \
        GP##_mx_ctx* x = (GP##_mx_ctx*)p;                                                        \
        GP##_multiexp_window(&x->win[w], x->bases, x->density, x->exps, x->n, (unsigned)w * x->c, x->c); \
    }                                                                                            \
    static int GP##_mx_prepare(GP##_mx_ctx* x, const uint8_t* bases_b, size_t n_bases,           \
                               const uint8_t* scalars, const uint8_t* density, size_t n) {       \
        size_t tot = 0;                                                                          \
        memset(x, 0, sizeof(*x));                                                                \
        if (density) { for (size_t i = 0; i < n; i++) tot += density[i] != 0; } else tot = n;    \
        if (tot != n_bases) return ORC_E_QUERY_SIZE;                                             \
        x->bases = (GP##_aff*)malloc(sizeof(GP##_aff) * (n_bases ? n_bases : 1));                \
        for (size_t i = 0; i < n_bases; i++) GP##_aff_load(&x->bases[i], bases_b + (size_t)STRIDE * i); \
        x->density = density; x->exps = (const uint64_t*)scalars; x->n = n;                      \
        x->c = multiexp_c(n);                                                                    \
        x->nw = (254 + x->c - 1) / x->c;                                                         \
        x->win = (GP##_jac*)calloc(x->nw, sizeof(GP##_jac));                                     \
        return ORC_OK;                                                                           \
    }                                                                                            \
    static void GP##_mx_finish(GP##_jac* out, GP##_mx_ctx* x) {                                  \
        GP##_jac total;                                                                          \
        GP##_jac_set_inf(&total);                                                                \
        for (size_t w = x->nw; w-- > 0;) {                                                       \
            for (unsigned k = 0; k < x->c; k++) GP##_jac_double(&total, &total);                 \
            GP##_jac_add(&total, &total, &x->win[w]);                                            \
        }                                                                                        \
        free(x->win);                                                                            \
        free(x->bases);                                                                          \
        x->win = NULL; x->bases = NULL;                                                          \
        *out = total;                                                                            \
    }                                                                                            \
    static int GP##_multiexp(GP##_jac* out, const uint8_t* bases_b, size_t n_bases,              \
                             const uint8_t* scalars, const uint8_t* density, size_t n, int threads) { \
        GP##_mx_ctx x;                                                                           \
        int rc = GP##_mx_prepare(&x, bases_b, n_bases, scalars, density, n);                     \
        if (rc) return rc;                                                                       \
        parallel_for(x.nw, threads, GP##_mx_task, &x);                                           \
        GP##_mx_finish(out, &x);                                                                 \
        return ORC_OK;                                                                           \
    }
DEFINE_MULTIEXP(g1, 64)
DEFINE_MULTIEXP(g2, 128)

typedef struct { task_fn fn; void* ctx; size_t w; } mx_any_task;
static void mx_any_run(void* p, size_t i) {
    mx_any_task* t = &((mx_any_task*)p)[i];
    t->fn(t->ctx, t->w);
}
/* the windows of n1 G1 and n2 G2 prepared multiexps as ONE task list, in submission order */
static void mx_run_all(g1_mx_ctx* j1, int n1, g2_mx_ctx* j2, int n2, int threads) {
    size_t total = 0;
    for (int k = 0; k < n1; k++) total += j1[k].nw;
    for (int k = 0; k < n2; k++) total += j2[k].nw;
    mx_any_task* t = (mx_any_task*)malloc(sizeof(mx_any_task) * (total ? total : 1));
    size_t at = 0;
    for (int k = 0; k < n1; k++)
        for (size_t w = 0; w < j1[k].nw; w++) t[at++] = (mx_any_task){g1_mx_task, &j1[k], w};
    for (int k = 0; k < n2; k++)
        for (size_t w = 0; w < j2[k].nw; w++) t[at++] = (mx_any_task){g2_mx_task, &j2[k], w};
    parallel_for(total, threads, mx_any_run, t);
    free(t);
}

int orc_msm_g1(const uint8_t* bases, size_t n_bases, const uint8_t* scalars, const uint8_t* density,
               size_t n, int threads, uint8_t out[64]) {
    orc_init();
    g1_jac r;
    int rc = g1_multiexp(&r, bases, n_bases, scalars, density, n, threads);
    if (rc) return rc;
    g1_aff a;
    g1_jac_to_aff(&a, &r);
    g1_aff_store(out, &a);
    return ORC_OK;
}
int orc_msm_g2(const uint8_t* bases, size_t n_bases, const uint8_t* scalars, const uint8_t* density,
               size_t n, int threads, uint8_t out[128]) {
    orc_init();
    g2_jac r;
    int rc = g2_multiexp(&r, bases, n_bases, scalars, density, n, threads);
    if (rc) return rc;
    g2_aff a;
    g2_jac_to_aff(&a, &r);
    g2_aff_store(out, &a);
    return ORC_OK;
}

/* sum_i scalars[i] * k[i] as a 384-bit little-endian integer (not reduced): the discrete-log side of the
 * "bases with known discrete logs" identities the full-size tests use (scalars: n x 32 B canonical LE, k: n x u64) */
int orc_dot_u64(const uint8_t* scalars, const uint64_t* k, size_t n, uint8_t out48[48]) {
    uint64_t acc[6] = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        const uint64_t* s = (const uint64_t*)(scalars + 32 * i);
        unsigned __int128 carry = 0;
        for (int j = 0; j < 4; j++) {
            unsigned __int128 t = (unsigned __int128)s[j] * k[i] + acc[j] + carry;
            acc[j] = (uint64_t)t;
            carry = t >> 64;
        }
        for (int j = 4; j < 6 && carry; j++) {
            unsigned __int128 t = (unsigned __int128)acc[j] + carry;
            acc[j] = (uint64_t)t;
            carry = t >> 64;
        }
    }
    memcpy(out48, acc, 48);
    return ORC_OK;
}

/* ------------------------------------------------------------------ small group helpers for tests */
int orc_g1_mul(const uint8_t p[64], const uint8_t k[32], uint8_t out[64]) {
    orc_init();
    g1_aff a; g1_jac j, r; uint64_t e[4];
    g1_aff_load(&a, p); g1_jac_from_aff(&j, &a); memcpy(e, k, 32);
    g1_jac_mul(&r, &j, e); g1_jac_to_aff(&a, &r); g1_aff_store(out, &a);
    return ORC_OK;
}
int orc_g2_mul(const uint8_t p[128], const uint8_t k[32], uint8_t out[128]) {
    orc_init();
    g2_aff a; g2_jac j, r; uint64_t e[4];
    g2_aff_load(&a, p); g2_jac_from_aff(&j, &a); memcpy(e, k, 32);
    g2_jac_mul(&r, &j, e); g2_jac_to_aff(&a, &r); g2_aff_store(out, &a);
    return ORC_OK;
}
int orc_g1_add(const uint8_t p[64], const uint8_t q[64], uint8_t out[64]) {
    orc_init();
    g1_aff a, b; g1_jac j, r;
    g1_aff_load(&a, p); g1_aff_load(&b, q); g1_jac_from_aff(&j, &a);
    g1_jac_add_mixed(&r, &j, &b); g1_jac_to_aff(&a, &r); g1_aff_store(out, &a);
    return ORC_OK;
}
int orc_g2_add(const uint8_t p[128], const uint8_t q[128], uint8_t out[128]) {
    orc_init();
    g2_aff a, b; g2_jac j, r;
    g2_aff_load(&a, p); g2_aff_load(&b, q); g2_jac_from_aff(&j, &a);
    g2_jac_add_mixed(&r, &j, &b); g2_jac_to_aff(&a, &r); g2_aff_store(out, &a);
    return ORC_OK;
}
/* field helpers: which = 0 Fq, 1 Fr; all operands Montgomery LE */
int orc_field_mul(int which, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) {
        fe x, y, z;
        memcpy(&x, a + 32 * i, 32); memcpy(&y, b + 32 * i, 32);
        if (which) fr_mul(&z, &x, &y); else fq_mul(&z, &x, &y);
        memcpy(out + 32 * i, &z, 32);
    }
    return ORC_OK;
}
int orc_to_mont(int which, const uint8_t* a, uint8_t* out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) {
        fe x, z;
        memcpy(&x, a + 32 * i, 32);
        if (which) fr_to_mont(&z, &x); else fq_to_mont(&z, &x);
        memcpy(out + 32 * i, &z, 32);
    }
    return ORC_OK;
}
int orc_from_mont(int which, const uint8_t* a, uint8_t* out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) {
        fe x, z;
        memcpy(&x, a + 32 * i, 32);
        if (which) fr_from_mont(&z, &x); else fq_from_mont(&z, &x);
        memcpy(out + 32 * i, &z, 32);
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ EvaluationDomain */
static uint32_t bitrev32(uint32_t n, unsigned l) {
    uint32_t r = 0;
    for (unsigned i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}
static void fr_pow_u64(fe* r, const fe* a, uint64_t e) {
    uint64_t ee[4] = {e, 0, 0, 0};
    fr_pow(r, a, ee);
}
static void serial_fft(fe* a, const fe* omega, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t k = 0; k < n; k++) {
        size_t rk = bitrev32((uint32_t)k, log_n);
        if (k < rk) { fe t = a[k]; a[k] = a[rk]; a[rk] = t; }
    }
    size_t m = 1;
    for (unsigned s = 0; s < log_n; s++) {
        fe w_m;
        fr_pow_u64(&w_m, omega, n / (2 * m));
        for (size_t k = 0; k < n; k += 2 * m) {
            fe w = fr_ONE;
            for (size_t j = 0; j < m; j++) {
                fe t;
                fr_mul(&t, &a[k + j + m], &w);
                fr_sub(&a[k + j + m], &a[k + j], &t);
                fr_add(&a[k + j], &a[k + j], &t);
                fr_mul(&w, &w, &w_m);
            }
        }
        m *= 2;
    }
}
typedef struct { fe* a; fe** tmp; const fe* omega; fe new_omega; unsigned log_n, log_cpus, log_new_n; } pfft_ctx;
static void pfft_task(void* p, size_t j) {
    pfft_ctx* x = (pfft_ctx*)p;
    size_t num_cpus = (size_t)1 << x->log_cpus, nn = (size_t)1 << x->log_new_n, n = (size_t)1 << x->log_n;
    fe* tmp = x->tmp[j];
    fe omega_j, omega_step, elt = fr_ONE;
    fr_pow_u64(&omega_j, x->omega, j);
    fr_pow_u64(&omega_step, x->omega, (uint64_t)j << x->log_new_n);
    for (size_t i = 0; i < nn; i++) {
        fe acc = {{0, 0, 0, 0}};
        for (size_t s = 0; s < num_cpus; s++) {
            size_t idx = (i + (s << x->log_new_n)) % n;
            fe t;
            fr_mul(&t, &x->a[idx], &elt);
            fr_add(&acc, &acc, &t);
            fr_mul(&elt, &elt, &omega_step);
        }
        tmp[i] = acc;
        fr_mul(&elt, &elt, &omega_j);
    }
    serial_fft(tmp, &x->new_omega, x->log_new_n);
}
static void best_fft(fe* a, const fe* omega, unsigned log_n, int threads) {
    unsigned log_cpus = 0;
    while ((2 << log_cpus) <= threads) log_cpus++;
    if (log_n <= log_cpus || threads <= 1) { serial_fft(a, omega, log_n); return; }
    size_t num_cpus = (size_t)1 << log_cpus, n = (size_t)1 << log_n;
    pfft_ctx x;
    x.a = a; x.omega = omega; x.log_n = log_n; x.log_cpus = log_cpus; x.log_new_n = log_n - log_cpus;
    fr_pow_u64(&x.new_omega, omega, num_cpus);
    x.tmp = (fe**)malloc(sizeof(fe*) * num_cpus);
    for (size_t j = 0; j < num_cpus; j++) x.tmp[j] = (fe*)malloc(sizeof(fe) << x.log_new_n);
    parallel_for(num_cpus, threads, pfft_task, &x);
    size_t mask = num_cpus - 1;
    for (size_t idx = 0; idx < n; idx++) a[idx] = x.tmp[idx & mask][idx >> log_cpus];
    for (size_t j = 0; j < num_cpus; j++) free(x.tmp[j]);
    free(x.tmp);
}
typedef struct { fe* a; size_t n, chunk; fe g; fe scale; int use_pow; } scale_ctx;
static void scale_task(void* p, size_t ci) {
    scale_ctx* x = (scale_ctx*)p;
    size_t lo = ci * x->chunk, hi = lo + x->chunk > x->n ? x->n : lo + x->chunk;
    fe u = x->scale;
    if (x->use_pow) {
        fe gp;
        fr_pow_u64(&gp, &x->g, lo);
        fr_mul(&u, &u, &gp);
    }
    for (size_t i = lo; i < hi; i++) {
        fr_mul(&x->a[i], &x->a[i], &u);
        if (x->use_pow) fr_mul(&u, &u, &x->g);
    }
}
/* a[i] *= scale * g^i (use_pow) or a[i] *= scale */
static void scale_vec(fe* a, size_t n, const fe* scale, const fe* g, int threads) {
    scale_ctx x;
    x.a = a; x.n = n; x.scale = *scale; x.use_pow = g != NULL;
    if (g) x.g = *g;
    size_t chunks = threads > 1 ? (size_t)threads : 1;
    x.chunk = (n + chunks - 1) / chunks;
    if (x.chunk == 0) x.chunk = 1;
    parallel_for((n + x.chunk - 1) / x.chunk, threads, scale_task, &x);
}
typedef struct { fe omega, omegainv, geninv, minv; unsigned exp; size_t m; } domain;
static int domain_init(domain* d, unsigned exp) {
    if (exp >= 28) return ORC_E_DEGREE_TOO_LARGE;
    d->exp = exp; d->m = (size_t)1 << exp;
    d->omega = FR_ROOT;
    for (unsigned i = exp; i < 28; i++) fr_sqr(&d->omega, &d->omega);
    fr_inv(&d->omegainv, &d->omega);
    fr_inv(&d->geninv, &FR_GEN);
    fe mm = {{(uint64_t)d->m, 0, 0, 0}};
    fr_to_mont(&mm, &mm);
    fr_inv(&d->minv, &mm);
    return ORC_OK;
}
static void dom_fft(const domain* d, fe* a, int th) { best_fft(a, &d->omega, d->exp, th); }
static void dom_ifft(const domain* d, fe* a, int th) {
    best_fft(a, &d->omegainv, d->exp, th);
    scale_vec(a, d->m, &d->minv, NULL, th);
}
static void dom_coset_fft(const domain* d, fe* a, int th) {
    scale_vec(a, d->m, &fr_ONE, &FR_GEN, th);
    dom_fft(d, a, th);
}
static void dom_icoset_fft(const domain* d, fe* a, int th) {
    dom_ifft(d, a, th);
    scale_vec(a, d->m, &fr_ONE, &d->geninv, th);
}

/* mode: 0 fft, 1 ifft, 2 coset_fft, 3 icoset_fft; data = m x 32 B Montgomery LE, in place */
int orc_fft(uint8_t* data, unsigned log_m, int mode, int threads) {
    orc_init();
    domain d;
    int rc = domain_init(&d, log_m);
    if (rc) return rc;
    fe* a = (fe*)data;
    switch (mode) {
        case 0: dom_fft(&d, a, threads); break;
        case 1: dom_ifft(&d, a, threads); break;
        case 2: dom_coset_fft(&d, a, threads); break;
        case 3: dom_icoset_fft(&d, a, threads); break;
        default: return -1;
    }
    return ORC_OK;
}

typedef struct { fe *a, *b, *c; size_t n, chunk; } mulsub_ctx;
static void mulsub_task(void* p, size_t ci) {
    mulsub_ctx* x = (mulsub_ctx*)p;
    size_t lo = ci * x->chunk, hi = lo + x->chunk > x->n ? x->n : lo + x->chunk;
    for (size_t i = lo; i < hi; i++) {
        fr_mul(&x->a[i], &x->a[i], &x->b[i]);
        fr_sub(&x->a[i], &x->a[i], &x->c[i]);
    }
}
typedef struct { fe* a; uint8_t* out; size_t n, chunk; } repr_ctx;
static void repr_task(void* p, size_t ci) {
    repr_ctx* x = (repr_ctx*)p;
    size_t lo = ci * x->chunk, hi = lo + x->chunk > x->n ? x->n : lo + x->chunk;
    for (size_t i = lo; i < hi; i++) {
        fe t;
        fr_from_mont(&t, &x->a[i]);
        memcpy(x->out + 32 * i, &t, 32);
    }
}

/* H coefficients: a, b, c are `rows` Montgomery Fr values (rows <= m); h_out gets (m-1) canonical
 * LE scalars.  Sequence exactly as bellman's create_proof. */
int orc_h(const uint8_t* a_in, const uint8_t* b_in, const uint8_t* c_in, size_t rows, unsigned log_m,
          int threads, uint8_t* h_out) {
    orc_init();
    domain d;
    int rc = domain_init(&d, log_m);
    if (rc) return rc;
    if (rows > d.m) return ORC_E_DEGREE_TOO_LARGE;
    fe* v[3];
    const uint8_t* src[3] = {a_in, b_in, c_in};
    for (int k = 0; k < 3; k++) {
        v[k] = (fe*)calloc(d.m, sizeof(fe));
        memcpy(v[k], src[k], rows * 32);
        dom_ifft(&d, v[k], threads);
        dom_coset_fft(&d, v[k], threads);
    }
    size_t chunks = threads > 1 ? (size_t)threads : 1;
    mulsub_ctx ms = {v[0], v[1], v[2], d.m, (d.m + chunks - 1) / chunks};
    parallel_for((d.m + ms.chunk - 1) / ms.chunk, threads, mulsub_task, &ms);
    /* divide by Z on the coset: Z(g) = g^m - 1 */
    fe z, zi;
    fr_pow_u64(&z, &FR_GEN, d.m);
    fr_sub(&z, &z, &fr_ONE);
    fr_inv(&zi, &z);
    scale_vec(v[0], d.m, &zi, NULL, threads);
    dom_icoset_fft(&d, v[0], threads);
    repr_ctx rp = {v[0], h_out, d.m - 1, (d.m + chunks - 1) / chunks};
    parallel_for((d.m - 1 + rp.chunk - 1) / rp.chunk, threads, repr_task, &rp);
    for (int k = 0; k < 3; k++) free(v[k]);
    return ORC_OK;
}

/* ------------------------------------------------------------------ R1CS evaluation */
typedef struct {
    uint32_t n_inputs, n_aux;
    uint64_t n_constraints;
    const uint64_t* row_ptr[3]; /* n_constraints + 1 each */
    const uint32_t* col[3];     /* flat variable index: inputs first, then aux */
    const uint8_t* coeff[3];    /* 32 B canonical LE per nonzero */
} orc_r1cs;

typedef struct {
    const orc_r1cs* r; const fe* w; fe* out[3]; uint8_t *a_aux_d, *b_in_d, *b_aux_d; size_t chunk;
} eval_ctx;
static void eval_task(void* p, size_t ci) {
    eval_ctx* x = (eval_ctx*)p;
    const orc_r1cs* r = x->r;
    size_t lo = ci * x->chunk, hi = lo + x->chunk > r->n_constraints ? r->n_constraints : lo + x->chunk;
    for (int k = 0; k < 3; k++) {
        for (size_t i = lo; i < hi; i++) {
            fe acc = {{0, 0, 0, 0}};
            for (uint64_t e = r->row_ptr[k][i]; e < r->row_ptr[k][i + 1]; e++) {
                uint32_t v = r->col[k][e];
                fe co, t;
                memcpy(&co, r->coeff[k] + 32 * e, 32);
                if (co.l[0] == 1 && (co.l[1] | co.l[2] | co.l[3]) == 0) {
                    fr_add(&acc, &acc, &x->w[v]);
                } else {
                    fr_to_mont(&co, &co);
                    fr_mul(&t, &x->w[v], &co);
                    fr_add(&acc, &acc, &t);
                }
                /* density by MENTION (racy byte stores of the same value 1 are benign) */
                if (k == 0 && v >= r->n_inputs) x->a_aux_d[v - r->n_inputs] = 1;
                if (k == 1) {
                    if (v >= r->n_inputs) x->b_aux_d[v - r->n_inputs] = 1;
                    else x->b_in_d[v] = 1;
                }
            }
            x->out[k][i] = acc;
        }
    }
}
/* witness: (n_inputs + n_aux) canonical LE scalars.  a/b/c out: (n_constraints + n_inputs) Montgomery. */
int orc_r1cs_eval(const orc_r1cs* r, const uint8_t* witness, int threads, uint8_t* a_out, uint8_t* b_out,
                  uint8_t* c_out, uint8_t* a_aux_density, uint8_t* b_input_density, uint8_t* b_aux_density) {
    orc_init();
    size_t nv = (size_t)r->n_inputs + r->n_aux;
    fe* w = (fe*)malloc(sizeof(fe) * nv);
    for (size_t i = 0; i < nv; i++) {
        memcpy(&w[i], witness + 32 * i, 32);
        fr_to_mont(&w[i], &w[i]);
    }
    memset(a_aux_density, 0, r->n_aux);
    memset(b_aux_density, 0, r->n_aux);
    memset(b_input_density, 0, r->n_inputs);
    eval_ctx x;
    x.r = r; x.w = w; x.out[0] = (fe*)a_out; x.out[1] = (fe*)b_out; x.out[2] = (fe*)c_out;
    x.a_aux_d = a_aux_density; x.b_in_d = b_input_density; x.b_aux_d = b_aux_density;
    size_t chunks = threads > 1 ? (size_t)threads * 4 : 1;
    x.chunk = (r->n_constraints + chunks - 1) / chunks;
    if (x.chunk == 0) x.chunk = 1;
    parallel_for((r->n_constraints + x.chunk - 1) / x.chunk, threads, eval_task, &x);
    for (uint32_t i = 0; i < r->n_inputs; i++) {
        x.out[0][r->n_constraints + i] = w[i];
        memset(&x.out[1][r->n_constraints + i], 0, 32);
        memset(&x.out[2][r->n_constraints + i], 0, 32);
    }
    free(w);
    return ORC_OK;
}

/* ------------------------------------------------------------------ create_proof */
typedef struct {
    const uint8_t *alpha_g1, *beta_g1, *beta_g2, *delta_g1, *delta_g2; /* affine Montgomery LE */
    const uint8_t* h; uint64_t n_h;
    const uint8_t* l; uint64_t n_l;
    const uint8_t* a; uint64_t n_a;
    const uint8_t* b_g1; uint64_t n_b_g1;
    const uint8_t* b_g2; uint64_t n_b_g2;
} orc_params;

static void fq_store_be(uint8_t* out, const fe* mont) {
    fe c;
    fq_from_mont(&c, mont);
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(c.l[(31 - i) >> 3] >> (8 * ((31 - i) & 7)));
}

/* timings (seconds): [0] r1cs eval, [1] H (7 FFTs), [2] MSMs, [3] total.  May be NULL. */
int orc_prove(const orc_params* pp, const orc_r1cs* r1, const uint8_t* witness, const uint8_t r_le[32],
              const uint8_t s_le[32], int threads, uint8_t proof[256], double* timings) {
    orc_init();
    struct timespec t0, t1, t2, t3;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    size_t ni = r1->n_inputs, na = r1->n_aux, rows = r1->n_constraints + ni;
    unsigned log_m = 0;
    while (((size_t)1 << log_m) < rows) log_m++;
    if (log_m >= 28) return ORC_E_DEGREE_TOO_LARGE;
    size_t m = (size_t)1 << log_m;
    uint8_t* abc[3];
    for (int k = 0; k < 3; k++) abc[k] = (uint8_t*)malloc(rows * 32);
    uint8_t* a_aux_d = (uint8_t*)malloc(na + 1);
    uint8_t* b_aux_d = (uint8_t*)malloc(na + 1);
    uint8_t* b_in_d = (uint8_t*)malloc(ni + 1);
    orc_r1cs_eval(r1, witness, threads, abc[0], abc[1], abc[2], a_aux_d, b_in_d, b_aux_d);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    uint8_t* h = (uint8_t*)malloc(m * 32);
    int rc = orc_h(abc[0], abc[1], abc[2], rows, log_m, threads, h);
    for (int k = 0; k < 3; k++) free(abc[k]);
    if (rc) return rc;
    clock_gettime(CLOCK_MONOTONIC, &t2);
    size_t a_aux_tot = 0, b_aux_tot = 0, b_in_tot = 0;
    for (size_t i = 0; i < na; i++) { a_aux_tot += a_aux_d[i]; b_aux_tot += b_aux_d[i]; }
    for (size_t i = 0; i < ni; i++) b_in_tot += b_in_d[i];
    if (pp->n_h < m - 1 || pp->n_l < na || pp->n_a != ni + a_aux_tot || pp->n_b_g1 != b_in_tot + b_aux_tot ||
        pp->n_b_g2 != b_in_tot + b_aux_tot)
        return ORC_E_QUERY_SIZE;
    const uint8_t* w_in = witness;
    const uint8_t* w_aux = witness + 32 * ni;
    g1_jac h_acc, l_acc, a_in, a_aux, b1_in, b1_aux;
    g2_jac b2_in, b2_aux;
    {   /* the eight multiexp calls of create_proof share one pool: h, l, a (inputs, aux), b_g1 (inputs, aux), b_g2 */
        g1_mx_ctx j1[6];
        g2_mx_ctx j2[2];
        memset(j1, 0, sizeof j1); memset(j2, 0, sizeof j2);
        rc |= g1_mx_prepare(&j1[0], pp->h, m - 1, h, NULL, m - 1);
        rc |= g1_mx_prepare(&j1[1], pp->l, na, w_aux, NULL, na);
        rc |= g1_mx_prepare(&j1[2], pp->a, ni, w_in, NULL, ni);
        rc |= g1_mx_prepare(&j1[3], pp->a + 64 * ni, a_aux_tot, w_aux, a_aux_d, na);
        rc |= g1_mx_prepare(&j1[4], pp->b_g1, b_in_tot, w_in, b_in_d, ni);
        rc |= g1_mx_prepare(&j1[5], pp->b_g1 + 64 * b_in_tot, b_aux_tot, w_aux, b_aux_d, na);
        rc |= g2_mx_prepare(&j2[0], pp->b_g2, b_in_tot, w_in, b_in_d, ni);
        rc |= g2_mx_prepare(&j2[1], pp->b_g2 + 128 * b_in_tot, b_aux_tot, w_aux, b_aux_d, na);
        if (!rc) mx_run_all(j1, 6, j2, 2, threads);
        g1_jac* o1[6] = {&h_acc, &l_acc, &a_in, &a_aux, &b1_in, &b1_aux};
        g2_jac* o2[2] = {&b2_in, &b2_aux};
        for (int k = 0; k < 6; k++) {
            if (!rc) g1_mx_finish(o1[k], &j1[k]);
            else { free(j1[k].win); free(j1[k].bases); }
        }
        for (int k = 0; k < 2; k++) {
            if (!rc) g2_mx_finish(o2[k], &j2[k]);
            else { free(j2[k].win); free(j2[k].bases); }
        }
    }
    free(h); free(a_aux_d); free(b_aux_d); free(b_in_d);
    if (rc) return rc;
    clock_gettime(CLOCK_MONOTONIC, &t3);

    g1_aff alpha, beta1, delta1;
    g2_aff beta2, delta2;
    g1_aff_load(&alpha, pp->alpha_g1); g1_aff_load(&beta1, pp->beta_g1); g1_aff_load(&delta1, pp->delta_g1);
    g2_aff_load(&beta2, pp->beta_g2); g2_aff_load(&delta2, pp->delta_g2);
    if (delta1.inf || delta2.inf) return ORC_E_UNEXPECTED_IDENTITY;
    uint64_t rr[4], ss[4], rs[4];
    memcpy(rr, r_le, 32); memcpy(ss, s_le, 32);
    {
        fe a, b, c;
        memcpy(&a, rr, 32); memcpy(&b, ss, 32);
        fr_to_mont(&a, &a); fr_to_mont(&b, &b);
        fr_mul(&c, &a, &b);
        fr_from_mont(&c, &c);
        memcpy(rs, &c, 32);
    }
    g1_jac d1j, al_j, be_j, g_a, g_c, t;
    g2_jac d2j, g_b;
    g1_jac_from_aff(&d1j, &delta1); g1_jac_from_aff(&al_j, &alpha); g1_jac_from_aff(&be_j, &beta1);
    g2_jac_from_aff(&d2j, &delta2);
    g1_jac_mul(&g_a, &d1j, rr);
    g1_jac_add_mixed(&g_a, &g_a, &alpha);
    g2_jac_mul(&g_b, &d2j, ss);
    g2_jac_add_mixed(&g_b, &g_b, &beta2);
    g1_jac_mul(&g_c, &d1j, rs);
    g1_jac_mul(&t, &al_j, ss); g1_jac_add(&g_c, &g_c, &t);
    g1_jac_mul(&t, &be_j, rr); g1_jac_add(&g_c, &g_c, &t);
    g1_jac a_ans, b1_ans;
    g2_jac b2_ans;
    g1_jac_add(&a_ans, &a_in, &a_aux);
    g1_jac_add(&g_a, &g_a, &a_ans);
    g1_jac_mul(&t, &a_ans, ss); g1_jac_add(&g_c, &g_c, &t);
    g1_jac_add(&b1_ans, &b1_in, &b1_aux);
    g2_jac_add(&b2_ans, &b2_in, &b2_aux);
    g2_jac_add(&g_b, &g_b, &b2_ans);
    g1_jac_mul(&t, &b1_ans, rr); g1_jac_add(&g_c, &g_c, &t);
    g1_jac_add(&g_c, &g_c, &h_acc);
    g1_jac_add(&g_c, &g_c, &l_acc);
    g1_aff A, C;
    g2_aff B;
    g1_jac_to_aff(&A, &g_a); g2_jac_to_aff(&B, &g_b); g1_jac_to_aff(&C, &g_c);
    if (A.inf || B.inf || C.inf) return ORC_E_UNEXPECTED_IDENTITY;
    fq_store_be(proof, &A.x); fq_store_be(proof + 32, &A.y);
    fq_store_be(proof + 64, &B.x.c0); fq_store_be(proof + 96, &B.x.c1);
    fq_store_be(proof + 128, &B.y.c0); fq_store_be(proof + 160, &B.y.c1);
    fq_store_be(proof + 192, &C.x); fq_store_be(proof + 224, &C.y);
    if (timings) {
#define DT(a, b) ((b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec))
        struct timespec t4;
        clock_gettime(CLOCK_MONOTONIC, &t4);
        timings[0] = DT(t0, t1); timings[1] = DT(t1, t2); timings[2] = DT(t2, t3); timings[3] = DT(t0, t4);
    }
    return ORC_OK;
}
