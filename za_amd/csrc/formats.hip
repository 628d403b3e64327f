// Wire formats on the za side of the boundary: bellman Parameters body, za's proving-key container
// header and proof.json.  Host code only (compiled by hipcc with the rest of the library).
#include "formats.h"

#include <atomic>
#include <cstring>
#include <thread>

namespace zg {

Fq2 g2_coeff_b() {
    Fq2 three{Fq::from_u64(3), Fq::zero()};
    Fq2 xi{Fq::from_u64(9), Fq::from_u64(1)};
    return three * xi.inv();
}

bool g2_in_subgroup(const G2Affine& p) {
    if (p.is_inf()) return true;
    return G2XYZZ::from_affine(p).mul(FrParams::MOD).is_inf();
}
bool scalar_is_canonical(const uint32_t v[8]) {
    for (int i = 7; i >= 0; i--) {
        if (v[i] < FrParams::MOD[i]) return true;
        if (v[i] > FrParams::MOD[i]) return false;
    }
    return false;
}

void fq_to_be(uint8_t out[32], const Fq& mont) {
    Fq c = mont.from_mont();
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(c.l[(31 - i) >> 2] >> (8 * ((31 - i) & 3)));
}

bool fq_from_be(const uint8_t in[32], Fq* mont) {
    Fq c = Fq::zero();
    for (int i = 0; i < 32; i++) c.l[(31 - i) >> 2] |= (uint32_t)in[i] << (8 * ((31 - i) & 3));
    // canonical check: c < q
    bool lt = false;
    for (int i = 7; i >= 0; i--) {
        if (c.l[i] < FqParams::MOD[i]) { lt = true; break; }
        if (c.l[i] > FqParams::MOD[i]) { lt = false; break; }
    }
    if (!lt) return false;
    *mont = c.to_mont();
    return true;
}

namespace {

// (the same classes, in the same order, as the device decoder's: formats.h KeyPointStatus)
using PointStatus = KeyPointStatus;
constexpr PointStatus PT_OK = KEY_PT_OK, PT_COMPRESSED = KEY_PT_COMPRESSED, PT_BAD_INFINITY = KEY_PT_BAD_INFINITY,
                      PT_NOT_IN_FIELD = KEY_PT_NOT_IN_FIELD, PT_NOT_ON_CURVE = KEY_PT_NOT_ON_CURVE,
                      PT_NOT_IN_SUBGROUP = KEY_PT_NOT_IN_SUBGROUP;

// pairing_ce bn256 G1Uncompressed: x || y big-endian; byte0 bit7 = compression flag (must be clear),
// bit6 = identity (then everything else must be zero).
PointStatus read_g1(const uint8_t* b, bool checked, G1Affine* out) {
    if (b[0] & 0x80) return PT_COMPRESSED;
    if (b[0] & 0x40) {
        if (b[0] & 0x3f) return PT_BAD_INFINITY;
        for (int i = 1; i < 64; i++)
            if (b[i]) return PT_BAD_INFINITY;
        *out = G1Affine::inf();
        return PT_OK;
    }
    G1Affine p;
    if (!fq_from_be(b, &p.x) || !fq_from_be(b + 32, &p.y)) return PT_NOT_IN_FIELD;
    if (checked && !on_curve(p, Fq::from_u64(3))) return PT_NOT_ON_CURVE;
    *out = p;
    return PT_OK;
}
// G2Uncompressed: x.c1 || x.c0 || y.c1 || y.c0 big-endian
PointStatus read_g2(const uint8_t* b, bool checked, const Fq2& coeff_b, G2Affine* out) {
    if (b[0] & 0x80) return PT_COMPRESSED;
    if (b[0] & 0x40) {
        if (b[0] & 0x3f) return PT_BAD_INFINITY;
        for (int i = 1; i < 128; i++)
            if (b[i]) return PT_BAD_INFINITY;
        *out = G2Affine::inf();
        return PT_OK;
    }
    G2Affine p;
    if (!fq_from_be(b, &p.x.c1) || !fq_from_be(b + 32, &p.x.c0) || !fq_from_be(b + 64, &p.y.c1) ||
        !fq_from_be(b + 96, &p.y.c0))
        return PT_NOT_IN_FIELD;
    if (checked && !on_curve(p, coeff_b)) return PT_NOT_ON_CURVE;
    // into_affine() of a G2Uncompressed also checks the subgroup (one multiplication by r per point, as bellman's
    // checked Parameters::read pays); G1 needs none (cofactor 1)
    if (checked && !g2_in_subgroup(p)) return PT_NOT_IN_SUBGROUP;
    *out = p;
    return PT_OK;
}

const char* status_str(int s) {
    switch (s) {
        case PT_COMPRESSED: return "unexpected compression mode";
        case PT_BAD_INFINITY: return "unexpected information in infinity encoding";
        case PT_NOT_IN_FIELD: return "coordinate is not in the field";
        case PT_NOT_ON_CURVE: return "point is not on the curve";
        case PT_NOT_IN_SUBGROUP: return "point is not in the correct subgroup";
        default: return "ok";
    }
}

template <class Fn>
void parallel_chunks(size_t n, Fn fn) {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = hw ? hw : 4;
    if (nt > 32) nt = 32;
    if (n < 4096 || nt <= 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> th;
    size_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        size_t lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
        if (lo >= hi) break;
        th.emplace_back([=] { fn(lo, hi); });
    }
    for (auto& t : th) t.join();
}

struct Reader {
    const uint8_t* p;
    size_t len, pos = 0;
    bool take(size_t n, const uint8_t** out) {
        if (pos + n > len || pos + n < pos) return false;
        *out = p + pos;
        pos += n;
        return true;
    }
    bool u32be(uint32_t* v) {
        const uint8_t* b;
        if (!take(4, &b)) return false;
        *v = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
        return true;
    }
};

// lowest (index << 3 | class) over the points that fail: what the device decoder reports too
void note_bad(std::atomic<uint64_t>* first, uint64_t index, int status) {
    const uint64_t v = (index << 3) | (uint64_t)status;
    uint64_t cur = first->load();
    while (v < cur && !first->compare_exchange_weak(cur, v)) {
    }
}

}  // namespace

int key_point_error(Ctx* ctx, const char* what, int status) {
    if (status == KEY_PT_INFINITY) return fail(ctx, ZG_E_IO, "%s: point at infinity", what);
    return fail(ctx, ZG_E_IO, "%s: %s", what, status_str(status));
}

// vk (alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2, ic) on the host, then the bounds of the five queries
int parse_bellman_header(Ctx* ctx, const uint8_t* buf, size_t len, bool checked, ParsedParams* out, KeyQuery q[5], size_t* consumed) {
    Reader rd{buf, len};
    const Fq2 b2 = g2_coeff_b();
    const uint8_t* b;
    PointStatus st;
#define RD_G1(dst, what)                                                                              \
    do {                                                                                              \
        if (!rd.take(64, &b)) return fail(ctx, ZG_E_IO, "unexpected end of parameters reading " what); \
        if ((st = read_g1(b, checked, &(dst))) != PT_OK) return key_point_error(ctx, what, st);      \
    } while (0)
#define RD_G2(dst, what)                                                                              \
    do {                                                                                              \
        if (!rd.take(128, &b)) return fail(ctx, ZG_E_IO, "unexpected end of parameters reading " what); \
        if ((st = read_g2(b, checked, b2, &(dst))) != PT_OK) return key_point_error(ctx, what, st);  \
    } while (0)
    RD_G1(out->alpha_g1, "vk.alpha_g1");
    RD_G1(out->beta_g1, "vk.beta_g1");
    RD_G2(out->beta_g2, "vk.beta_g2");
    RD_G2(out->gamma_g2, "vk.gamma_g2");
    RD_G1(out->delta_g1, "vk.delta_g1");
    RD_G2(out->delta_g2, "vk.delta_g2");
    uint32_t n;
    if (!rd.u32be(&n)) return fail(ctx, ZG_E_IO, "unexpected end of parameters reading |ic|");
    if ((size_t)n * 64 > len - rd.pos) return fail(ctx, ZG_E_IO, "ic length %u exceeds the buffer", n);
    out->ic.resize(n);
    for (uint32_t i = 0; i < n; i++) RD_G1(out->ic[i], "vk.ic");
#undef RD_G1
#undef RD_G2
    for (int which = 0; which < 5; which++) {
        const char* what = key_query_name(which);
        uint32_t cnt;
        if (!rd.u32be(&cnt)) return fail(ctx, ZG_E_IO, "unexpected end of parameters reading |%s|", what);
        const uint8_t* base;
        if (!rd.take((size_t)cnt * (which == 4 ? 128 : 64), &base))
            return fail(ctx, ZG_E_IO, "unexpected end of parameters inside %s", what);
        q[which].off = (size_t)(base - buf);
        q[which].n = cnt;
    }
    if (consumed) *consumed = rd.pos;
    return ZG_OK;
}

// |ic| and |l| of a Parameters body from its length prefixes alone (nothing validated; false: the body is too short for them)
bool peek_bellman_counts(const uint8_t* buf, size_t len, uint32_t* n_ic, uint32_t* n_l) {
    Reader rd{buf, len};
    const uint8_t* b;
    if (!rd.take(64 + 64 + 128 + 128 + 64 + 128, &b)) return false;
    uint32_t n;
    if (!rd.u32be(&n) || !rd.take((size_t)n * 64, &b)) return false;
    *n_ic = n;
    if (!rd.u32be(&n) || !rd.take((size_t)n * 64, &b)) return false;   // h
    if (!rd.u32be(&n)) return false;
    *n_l = n;
    return true;
}

int parse_bellman_params(Ctx* ctx, const uint8_t* buf, size_t len, bool checked, ParsedParams* out, size_t* consumed) {
    KeyQuery q[5];
    ZG_TRY(parse_bellman_header(ctx, buf, len, checked, out, q, consumed));
    const Fq2 b2 = g2_coeff_b();
    std::vector<G1Affine>* g1v[4] = {&out->h, &out->l, &out->a, &out->b_g1};
    for (int which = 0; which < 5; which++) {
        const uint8_t* base = buf + q[which].off;
        const uint64_t cnt = q[which].n;
        if (which < 4) g1v[which]->resize(cnt);
        else out->b_g2.resize(cnt);
        std::atomic<uint64_t> first{~0ull};
        parallel_chunks(cnt, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                PointStatus s;
                bool inf;
                if (which < 4) {
                    G1Affine& p = (*g1v[which])[i];
                    s = read_g1(base + 64 * i, checked, &p);
                    inf = s == PT_OK && p.is_inf();
                } else {
                    G2Affine& p = out->b_g2[i];
                    s = read_g2(base + 128 * i, checked, b2, &p);
                    inf = s == PT_OK && p.is_inf();
                }
                if (s != PT_OK) note_bad(&first, i, (int)s);
                else if (inf) note_bad(&first, i, KEY_PT_INFINITY);
            }
        });
        const uint64_t v = first.load();
        if (v != ~0ull) return key_point_error(ctx, key_query_name(which), (int)(v & 7u));
    }
    return ZG_OK;
}

static void put_g1(std::vector<uint8_t>* o, const G1Affine& p) {
    size_t at = o->size();
    o->resize(at + 64, 0);
    if (p.is_inf()) { (*o)[at] = 0x40; return; }
    fq_to_be(o->data() + at, p.x);
    fq_to_be(o->data() + at + 32, p.y);
}
static void put_g2(std::vector<uint8_t>* o, const G2Affine& p) {
    size_t at = o->size();
    o->resize(at + 128, 0);
    if (p.is_inf()) { (*o)[at] = 0x40; return; }
    fq_to_be(o->data() + at, p.x.c1);
    fq_to_be(o->data() + at + 32, p.x.c0);
    fq_to_be(o->data() + at + 64, p.y.c1);
    fq_to_be(o->data() + at + 96, p.y.c0);
}
static void put_u32be(std::vector<uint8_t>* o, uint32_t v) {
    o->push_back((uint8_t)(v >> 24)); o->push_back((uint8_t)(v >> 16));
    o->push_back((uint8_t)(v >> 8)); o->push_back((uint8_t)v);
}

void write_bellman_vk(const ParsedParams& p, std::vector<uint8_t>* out) {
    put_g1(out, p.alpha_g1); put_g1(out, p.beta_g1); put_g2(out, p.beta_g2); put_g2(out, p.gamma_g2);
    put_g1(out, p.delta_g1); put_g2(out, p.delta_g2);
    put_u32be(out, (uint32_t)p.ic.size());
    for (auto& x : p.ic) put_g1(out, x);
}
int parse_bellman_vk(Ctx* ctx, const uint8_t* buf, size_t len, ParsedParams* out) {
    // reuse the Parameters parser on a buffer that ends after ic by appending five empty queries
    const size_t need = 64 + 64 + 128 + 128 + 64 + 128 + 4;
    if (len < need) return fail(ctx, ZG_E_IO, "verifying key: unexpected end");
    const uint32_t n = ((uint32_t)buf[need - 4] << 24) | ((uint32_t)buf[need - 3] << 16) | ((uint32_t)buf[need - 2] << 8) | buf[need - 1];
    if ((size_t)n > (len - need) / 64) return fail(ctx, ZG_E_IO, "verifying key: ic exceeds the buffer");
    std::vector<uint8_t> tmp(buf, buf + need + (size_t)n * 64);
    tmp.resize(tmp.size() + 20, 0);   // h, l, a, b_g1, b_g2 with zero entries
    return parse_bellman_params(ctx, tmp.data(), tmp.size(), true, out, nullptr);
}
void write_bellman_params(const ParsedParams& p, std::vector<uint8_t>* out) {
    write_bellman_vk(p, out);
    for (const std::vector<G1Affine>* v : {&p.h, &p.l, &p.a, &p.b_g1}) {
        put_u32be(out, (uint32_t)v->size());
        for (auto& x : *v) put_g1(out, x);
    }
    put_u32be(out, (uint32_t)p.b_g2.size());
    for (auto& x : p.b_g2) put_g2(out, x);
}

// za container: u32be len || bincode(asts); u32be n_constraints, each u32be len || bincode(QEQ);
// u32be n_ignore, each u32be id; then the bellman Parameters body (format.rs:223-252).
int pk_params_offset(const uint8_t* pk, size_t len, size_t* offset) {
    Reader rd{pk, len};
    uint32_t n;
    const uint8_t* b;
    if (!rd.u32be(&n) || !rd.take(n, &b)) return ZG_E_IO;
    uint32_t nc;
    if (!rd.u32be(&nc)) return ZG_E_IO;
    for (uint32_t i = 0; i < nc; i++) {
        if (!rd.u32be(&n) || !rd.take(n, &b)) return ZG_E_IO;
        if (rd.pos + 4096 < len) {   // (as pk_walk)
            __builtin_prefetch(pk + rd.pos + 1024);
            __builtin_prefetch(pk + rd.pos + 2048);
            __builtin_prefetch(pk + rd.pos + 3072);
            __builtin_prefetch(pk + rd.pos + 4096);
        }
    }
    uint32_t ni;
    if (!rd.u32be(&ni) || !rd.take((size_t)ni * 4, &b)) return ZG_E_IO;
    *offset = rd.pos;
    return ZG_OK;
}

// decimal string of a 256-bit little-endian integer
static std::string dec_of_le32(const uint8_t* le) {
    uint32_t w[8];
    for (int i = 0; i < 8; i++)
        w[i] = (uint32_t)le[4 * i] | ((uint32_t)le[4 * i + 1] << 8) | ((uint32_t)le[4 * i + 2] << 16) |
               ((uint32_t)le[4 * i + 3] << 24);
    std::string rev;
    for (;;) {
        bool nz = false;
        uint64_t rem = 0;
        for (int i = 7; i >= 0; i--) {
            uint64_t cur = (rem << 32) | w[i];
            w[i] = (uint32_t)(cur / 1000000000ull);
            rem = cur % 1000000000ull;
            if (w[i]) nz = true;
        }
        if (!nz) {
            // last group without zero padding
            std::string g = std::to_string(rem);
            for (auto it = g.rbegin(); it != g.rend(); ++it) rev.push_back(*it);
            break;
        }
        for (int k = 0; k < 9; k++) {
            rev.push_back((char)('0' + rem % 10));
            rem /= 10;
        }
    }
    return std::string(rev.rbegin(), rev.rend());
}

int proof_json(const uint8_t proof[256], const uint8_t* public_inputs32, size_t n_public, std::string* out) {
    auto hex = [&](int k) {
        static const char* d = "0123456789abcdef";
        std::string s = "\"0x";
        for (int i = 0; i < 32; i++) {
            s.push_back(d[proof[32 * k + i] >> 4]);
            s.push_back(d[proof[32 * k + i] & 15]);
        }
        s.push_back('"');
        return s;
    };
    std::string j = "{\"a\":[" + hex(0) + "," + hex(1) + "],\"b\":[[" + hex(2) + "," + hex(3) + "],[" + hex(4) + "," +
                    hex(5) + "]],\"c\":[" + hex(6) + "," + hex(7) + "],\"public_inputs\":[";
    for (size_t i = 0; i < n_public; i++) {
        if (i) j.push_back(',');
        j += "\"" + dec_of_le32(public_inputs32 + 32 * i) + "\"";
    }
    j += "]}";
    *out = j;
    return ZG_OK;
}

}  // namespace zg

namespace zg {

// ---- za proving-key container -> R1CS ------------------------------------------
// Layout: prover/src/groth16/format.rs:223-252 (write_pk), read back at :254-293.  bincode(QEQ) = three LCs
// (a, b, c with a*b + c = 0, compiler/src/types/constraint.rs:51); an LC is u64le count, then per term
// u64le signal id, u64le n_words, n_words x u32le little-endian base-2^32 digits (num-bigint 0.2 serde of
// compiler/src/algebra/fs.rs:34).
namespace {
struct LeReader {
    const uint8_t* p;
    size_t len, pos;
    bool u64(uint64_t* v) {
        if (pos + 8 > len) return false;
        uint64_t x = 0;
        for (int i = 0; i < 8; i++) x |= (uint64_t)p[pos + i] << (8 * i);
        pos += 8;
        *v = x;
        return true;
    }
    bool u32(uint32_t* v) {
        if (pos + 4 > len) return false;
        *v = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24);
        pos += 4;
        return true;
    }
};
}  // namespace

// One pass over the container's length prefixes (a chain of dependent loads, ~190 bytes apart: 0.15 s at 2^22
// constraints -- done once, for the offset of the Parameters body AND for the constraint parser)
int pk_walk(Ctx* ctx, const uint8_t* pk, size_t len, PkWalk* w) {
    Reader rd{pk, len};
    uint32_t n;
    const uint8_t* b;
    if (!rd.u32be(&n) || !rd.take(n, &b)) return fail(ctx, ZG_E_IO, "pk: truncated AST blob");
    uint32_t nc;
    if (!rd.u32be(&nc)) return fail(ctx, ZG_E_IO, "pk: truncated constraint count");
    // a constraint takes at least its u32 length prefix + three u64 term counts: bound the allocation by the buffer
    if ((size_t)nc > (len - rd.pos) / 28) return fail(ctx, ZG_E_IO, "pk: %u constraints do not fit the container", nc);
    w->at.resize((size_t)nc + 1);
    for (uint32_t i = 0; i < nc; i++) {
        if (!rd.u32be(&n) || !rd.take(n, &b)) return fail(ctx, ZG_E_IO, "pk: truncated constraint %u", i);
        w->at[i] = (size_t)(b - pk);
        // (the next prefixes lie a few hundred bytes ahead: ask for those lines now -- the chain is otherwise one cache miss
        // per constraint)
        if (rd.pos + 4096 < len) {
            __builtin_prefetch(pk + rd.pos + 1024);
            __builtin_prefetch(pk + rd.pos + 2048);
            __builtin_prefetch(pk + rd.pos + 3072);
            __builtin_prefetch(pk + rd.pos + 4096);
        }
    }
    w->at[nc] = rd.pos + 4;   // (constraint i spans [at[i], at[i + 1] - 4): the next one's length prefix lies in between)
    uint32_t ni;
    if (!rd.u32be(&ni)) return fail(ctx, ZG_E_IO, "pk: truncated ignore list");
    if ((size_t)ni > (len - rd.pos) / 4) return fail(ctx, ZG_E_IO, "pk: %u ignored signals do not fit the container", ni);
    w->ignore.resize(ni);
    for (uint32_t i = 0; i < ni; i++)
        if (!rd.u32be(&w->ignore[i])) return fail(ctx, ZG_E_IO, "pk: truncated ignore list");
    w->params_offset = rd.pos;
    return ZG_OK;
}

int parse_pk_r1cs(Ctx* ctx, const uint8_t* pk, size_t len, const PkWalk& walk, uint32_t n_public /* |ic| - 1 */,
                  uint32_t n_aux /* |l| */, PkR1cs* out) {
    (void)len;
    const std::vector<size_t>& at = walk.at;
    const std::vector<uint32_t>& ignore = walk.ignore;
    const uint32_t nc = (uint32_t)(at.size() - 1), ni = (uint32_t)ignore.size();
    out->params_offset = walk.params_offset;
    // variable numbering of CircomCircuit::synthesize (prover.rs:55-88): ONE, then signal ids in order;
    // ignored ids get no variable; main public inputs (ids 1..n_public: the evaluator numbers outputs and public
    // inputs first, compiler/src/evaluator/test.rs:739-768) are inputs; the rest are aux.
    const uint64_t n_signals = 1ull + n_public + n_aux + ni;
    out->n_inputs = n_public + 1;
    out->n_aux = n_aux;
    out->var_of_signal.assign(n_signals, UINT32_MAX);
    out->var_of_signal[0] = 0;
    size_t ig = 0;
    uint32_t next_in = 1, next_aux = 0;
    for (uint64_t id = 1; id < n_signals; id++) {
        if (ig < ignore.size() && ignore[ig] == id) {
            ig++;
            if (id <= n_public) return fail(ctx, ZG_E_BAD_FORMAT, "pk: public signal %llu is in ignore_signals", (unsigned long long)id);
            continue;
        }
        if (id <= n_public) out->var_of_signal[id] = next_in++;
        else out->var_of_signal[id] = out->n_inputs + next_aux++;
    }
    if (ig != ignore.size()) return fail(ctx, ZG_E_BAD_FORMAT, "pk: ignore_signals is not sorted / in range");
    if (next_aux != n_aux) return fail(ctx, ZG_E_BAD_FORMAT, "pk: aux count mismatch");
    // pass 2, in ranges on host threads: every constraint is walked (format checks, signal ids) and its terms counted;
    // pass 3, same ranges: the terms are written where the counts say -- straight into the final arrays, which are
    // first touched by the threads that fill them (no zero fill, no joining copy: at 2^22 constraints that is 0.5 GB)
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt ? std::min(nt, 32u) : 4u;
    if (nc < 1u << 14) nt = 1;
    struct Range {
        uint64_t lo = 0, hi = 0, terms[3] = {0, 0, 0};
        uint64_t bad_row = UINT64_MAX;   // first constraint of the range that does not parse
        int bad_code = 0;
        std::string bad_msg;
    };
    std::vector<Range> ranges(nt);
    for (unsigned t = 0; t < nt; t++) {
        ranges[t].lo = (uint64_t)nc * t / nt;
        ranges[t].hi = (uint64_t)nc * (t + 1) / nt;
    }
    for (int k = 0; k < 3; k++) out->row_ptr[k].alloc((size_t)nc + 1);
    auto count = [&](unsigned t) {
        Range& rg = ranges[t];
        char msg[160];
        auto bad = [&](uint64_t i, int code, const char* fmt, unsigned long long a1 = 0, unsigned long long a2 = 0) {
            snprintf(msg, sizeof(msg), fmt, a1, a2);
            rg.bad_row = i;
            rg.bad_code = code;
            rg.bad_msg = msg;
        };
        for (uint64_t i = rg.lo; i < rg.hi; i++) {
            const size_t nbytes = at[i + 1] - 4 - at[i];
            LeReader lr{pk + at[i], nbytes, 0};
            for (int k = 0; k < 3; k++) {
                uint64_t cnt;
                if (!lr.u64(&cnt) || cnt > nbytes) return bad(i, ZG_E_BAD_FORMAT, "pk: bad LC in constraint %llu", i);
                for (uint64_t tm = 0; tm < cnt; tm++) {
                    uint64_t sig, nw;
                    if (!lr.u64(&sig) || !lr.u64(&nw) || nw > 8) return bad(i, ZG_E_BAD_FORMAT, "pk: bad term in constraint %llu", i);
                    if (lr.pos + 4 * nw > nbytes) return bad(i, ZG_E_BAD_FORMAT, "pk: truncated coefficient in constraint %llu", i);
                    lr.pos += 4 * nw;
                    if (sig >= n_signals)
                        return bad(i, ZG_E_BAD_FORMAT, "pk: signal id %llu but the key describes %llu signals", sig, n_signals);
                    if (out->var_of_signal[sig] == UINT32_MAX)
                        return bad(i, ZG_E_BAD_FORMAT, "pk: constraint %llu uses ignored signal %llu (signal not defined, format.rs:215)", i, sig);
                }
                rg.terms[k] += cnt;
            }
            if (lr.pos != nbytes) return bad(i, ZG_E_BAD_FORMAT, "pk: trailing bytes in constraint %llu", i);
        }
    };
    uint64_t base[3] = {0, 0, 0};
    std::vector<uint64_t> first[3];
    auto fill = [&](unsigned t) {
        const Range& rg = ranges[t];
        uint64_t e[3] = {first[0][t], first[1][t], first[2][t]};
        for (uint64_t i = rg.lo; i < rg.hi; i++) {
            LeReader lr{pk + at[i], at[i + 1] - 4 - at[i], 0};
            for (int k = 0; k < 3; k++) {
                uint64_t cnt = 0;
                lr.u64(&cnt);
                for (uint64_t tm = 0; tm < cnt; tm++, e[k]++) {
                    uint64_t sig = 0, nw = 0;
                    lr.u64(&sig);
                    lr.u64(&nw);
                    Fr c = Fr::zero();
                    for (uint64_t w = 0; w < nw; w++) lr.u32(&c.l[w]);
                    if (k == 2) c = c.to_mont().neg().from_mont();   // bellman gets a*b = -c (prover.rs:98)
                    out->col[k].data()[e[k]] = out->var_of_signal[sig];
                    memcpy(out->coeff[k].data() + 32 * e[k], &c, 32);
                }
                out->row_ptr[k].data()[i + 1] = e[k];
            }
        }
    };
    auto run = [&](auto&& fn) {
        if (nt == 1) return fn(0u);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(fn, t);
        for (auto& t : th) t.join();
    };
    run(count);
    for (const Range& rg : ranges)   // (ranges are in constraint order: the first one with an error holds the lowest row)
        if (rg.bad_row != UINT64_MAX) return fail(ctx, rg.bad_code, "%s", rg.bad_msg.c_str());
    for (int k = 0; k < 3; k++) {
        first[k].resize(nt);
        for (unsigned t = 0; t < nt; t++) {
            first[k][t] = base[k];
            base[k] += ranges[t].terms[k];
        }
        out->col[k].alloc(base[k]);
        out->coeff[k].alloc(base[k] * 32);
        out->row_ptr[k].data()[0] = 0;
    }
    run(fill);
    out->n_constraints = nc;
    return ZG_OK;
}

// .binwitness (compiler/src/tester/embeeded.rs:47-57): 256-bit values written by FS::write_256_w32
// (fs.rs:116-126: 32 big-endian bytes, 4-byte groups emitted last-to-first): first the signal count, then the
// value of every signal id 0.. (id 0 = ONE).
static void read_w32(const uint8_t* p, uint8_t out_le[32]) {
    // group g (0..7) holds big-endian bytes of 32-bit word g counted from the least significant
    for (int g = 0; g < 8; g++)
        for (int k = 0; k < 4; k++) out_le[4 * g + k] = p[4 * g + (3 - k)];
}
int witness_from_binwitness(Ctx* ctx, const std::vector<uint32_t>& var_of_signal, uint32_t n_vars, const uint8_t* bw,
                            size_t len, uint8_t* witness_out) {
    if (len < 32 || len % 32) return fail(ctx, ZG_E_BAD_FORMAT, "binwitness: length %zu is not a multiple of 32", len);
    uint8_t tmp[32];
    read_w32(bw, tmp);
    uint64_t count = 0;
    for (int i = 0; i < 8; i++) count |= (uint64_t)tmp[i] << (8 * i);
    if (count != var_of_signal.size() || len != 32 * (count + 1))
        return fail(ctx, ZG_E_BAD_FORMAT, "binwitness: %llu signals, the key describes %zu", (unsigned long long)count,
                    var_of_signal.size());
    // (268 MB touched for a 2^22 circuit: on a few threads -- the signals of a slice go to variables of their own)
    constexpr size_t CH = (size_t)1 << 16;
    host_par_for(((size_t)n_vars + CH - 1) / CH, [&](size_t t) {
        const size_t lo = t * CH, hi = std::min<size_t>(n_vars, lo + CH);
        memset(witness_out + 32 * lo, 0, 32 * (hi - lo));
    });
    host_par_for((size_t)((count + CH - 1) / CH), [&](size_t t) {
        const uint64_t hi = std::min<uint64_t>(count, (uint64_t)(t + 1) * CH);
        for (uint64_t id = (uint64_t)t * CH; id < hi; id++) {
            const uint32_t v = var_of_signal[id];
            if (v == UINT32_MAX) continue;
            read_w32(bw + 32 * (id + 1), witness_out + 32 * (size_t)v);
        }
    });
    return ZG_OK;
}

}  // namespace zg
