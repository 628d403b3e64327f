// Synthetic workloads for bench.py and the full-size parity tests, and the self-test hooks (include/za_amd_synth.h).
// Not part of the drop-in boundary and not in the product library: this file is libza_amd_synth.so, which links against
// libza_amd.so and reaches its internals through the C++ symbols of common.h.  Stands in for the za compiler's R1CS +
// witness output and for bellman's generate_random_parameters.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/za_amd_synth.h"
#include "common.h"
#include "formats.h"
#include "handles.h"
#include "internal_api.h"

// libza_amd.so exports no C++ symbol: the internals this library uses are reached through its private table.  The
// definitions below give the declarations of common.h / formats.h a body HERE, forwarding to it.
namespace zg {
int fail(Ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return zg_internal_v1.fail_msg(ctx, code, buf);
}
int on_exception(Ctx* ctx) { return zg_internal_v1.on_exception(ctx); }   // (called inside a handler: the callee rethrows)
int convert_mont_dev(Ctx* ctx, void* d_elems, size_t n, int which, int dir, hipStream_t st) {
    return zg_internal_v1.convert_mont_dev(ctx, d_elems, n, which, dir, st);
}
int host_pack_for_test(const uint8_t* witness32, size_t nv, int threads, uint8_t* classes, uint32_t* other_index,
                       uint8_t* others32, size_t* n_others) {
    return zg_internal_v1.host_pack_for_test(witness32, nv, threads, classes, other_index, others32, n_others);
}
int host_pack_split_for_test(const uint8_t* witness32, size_t nv, int threads, int cap_permille, int ready_every, uint8_t* raw_out,
                             uint8_t* classes, uint32_t* other_index, uint8_t* others32, size_t* n_others, size_t* first) {
    return zg_internal_v1.host_pack_split_for_test(witness32, nv, threads, cap_permille, ready_every, raw_out, classes, other_index,
                                                   others32, n_others, first);
}
int params_h_eval_slice(Ctx* ctx, const ::zg_params* p, const G1Affine* d_h_full, uint64_t n_h, const G1Affine* d_h_eval_full) {
    return zg_internal_v1.params_h_eval_slice(ctx, p, d_h_full, n_h, d_h_eval_full);
}
int parse_bellman_params(Ctx* ctx, const uint8_t* buf, size_t len, bool checked, ParsedParams* out, size_t* consumed) {
    return zg_internal_v1.parse_bellman_params(ctx, buf, len, checked, out, consumed);
}
void write_bellman_params(const ParsedParams& p, std::vector<uint8_t>* out) { zg_internal_v1.write_bellman_params(p, out); }
}  // namespace zg

using namespace zg;

static inline uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ static inline uint64_t splitmix64_hd(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct zg_synth {
    uint32_t n_inputs = 0, n_aux = 0;
    uint64_t n_constraints = 0;
    std::vector<uint64_t> row_ptr[3];
    std::vector<uint32_t> col[3];
    std::vector<uint8_t> coeff[3];
    std::vector<uint8_t> witness;
};

namespace {
struct Rng {
    uint64_t seed, ctr = 0;
    uint64_t next() { return splitmix64_at(seed, ctr++); }
    Fr fr() {  // uniform below 2^253, Montgomery form
        Fr c;
        for (int i = 0; i < 4; i++) {
            uint64_t v = next();
            c.l[2 * i] = (uint32_t)v;
            c.l[2 * i + 1] = (uint32_t)(v >> 32);
        }
        c.l[7] &= 0x1fffffffu;
        return c.to_mont();
    }
};
void push_term(zg_synth* s, int k, uint32_t var, const Fr& coeff_mont) {
    s->col[k].push_back(var);
    Fr c = coeff_mont.from_mont();
    const uint8_t* b = reinterpret_cast<const uint8_t*>(&c);
    s->coeff[k].insert(s->coeff[k].end(), b, b + 32);
}
}  // namespace

extern "C" int zg_synth_r1cs_new(uint32_t log_m, uint32_t n_inputs, double bit_fraction, uint64_t seed, zg_synth** out) try {
    if (!out || log_m >= 28 || n_inputs == 0 || ((uint64_t)1 << log_m) <= n_inputs) return ZG_E_INVALID_ARG;
    const uint64_t m = (uint64_t)1 << log_m;
    const uint64_t n = m - n_inputs;
    const uint32_t n_seed = 64;
    std::unique_ptr<zg_synth> s(new zg_synth());
    s->n_inputs = n_inputs;
    s->n_constraints = n;
    s->n_aux = (uint32_t)(n + n_seed);
    const size_t nv = (size_t)n_inputs + s->n_aux;
    std::vector<Fr> w(nv);  // Montgomery
    Rng rng{seed};
    const Fr one = Fr::one(), two = Fr::from_u64(2), minus_one = one.neg();
    std::vector<uint32_t> bits;
    bits.reserve(nv);
    w[0] = one;
    for (uint32_t i = 1; i < n_inputs; i++) w[i] = rng.fr();
    for (uint32_t i = 0; i < n_seed; i++) {
        uint32_t v = n_inputs + i;
        if (i & 1) {
            w[v] = rng.fr();
        } else {
            w[v] = (rng.next() & 1) ? one : Fr::zero();
            bits.push_back(v);
        }
    }
    const uint64_t thresh = bit_fraction >= 1.0 ? ~0ull : (uint64_t)(bit_fraction * 18446744073709551615.0);
    for (int k = 0; k < 3; k++) {
        s->row_ptr[k].reserve(n + 1);
        s->row_ptr[k].push_back(0);
    }
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t c = n_inputs + n_seed + (uint32_t)i;
        if (i < n_seed / 4) {
            // the first rows tie the seed variables in (a valid set-up rejects unconstrained aux variables)
            const uint32_t s0 = n_inputs + 4 * (uint32_t)i;
            Fr k1 = rng.fr(), k2 = rng.fr(), k3 = rng.fr(), k4 = rng.fr();
            push_term(s.get(), 0, s0, k1);
            push_term(s.get(), 0, s0 + 1, k2);
            push_term(s.get(), 1, s0 + 2, k3);
            push_term(s.get(), 1, s0 + 3, k4);
            push_term(s.get(), 2, c, one);
            w[c] = (k1 * w[s0] + k2 * w[s0 + 1]) * (k3 * w[s0 + 2] + k4 * w[s0 + 3]);
            for (int k = 0; k < 3; k++) s->row_ptr[k].push_back(s->col[k].size());
            continue;
        }
        const bool bit_row = bit_fraction > 0.0 && rng.next() <= thresh;
        if (bit_row) {
            uint32_t a = bits[rng.next() % bits.size()];
            uint32_t b = bits[(rng.next() % bits.size()) & ~(size_t)1];
            // (2a) * b = a + b - c   <=>   c = a xor b
            push_term(s.get(), 0, a, two);
            push_term(s.get(), 1, b, one);
            push_term(s.get(), 2, a, one);
            push_term(s.get(), 2, b, one);
            push_term(s.get(), 2, c, minus_one);
            w[c] = w[a] + w[b] - two * w[a] * w[b];
            bits.push_back(c);
        } else {
            uint32_t x = (uint32_t)(rng.next() % c), y = (uint32_t)(rng.next() % c);
            uint32_t z = (uint32_t)(rng.next() % c), t = (uint32_t)(rng.next() % c);
            Fr k1 = rng.fr(), k2 = rng.fr(), k3 = rng.fr(), k4 = rng.fr();
            push_term(s.get(), 0, x, k1);
            push_term(s.get(), 0, y, k2);
            push_term(s.get(), 1, z, k3);
            push_term(s.get(), 1, t, k4);
            push_term(s.get(), 2, c, one);
            w[c] = (k1 * w[x] + k2 * w[y]) * (k3 * w[z] + k4 * w[t]);
        }
        for (int k = 0; k < 3; k++) s->row_ptr[k].push_back(s->col[k].size());
    }
    s->witness.resize(nv * 32);
    for (size_t i = 0; i < nv; i++) {
        Fr c = w[i].from_mont();
        memcpy(s->witness.data() + 32 * i, &c, 32);
    }
    *out = s.release();
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

extern "C" int zg_synth_r1cs_info(const zg_synth* s, uint64_t info[6]) try {
    if (!s || !info) return ZG_E_INVALID_ARG;
    info[0] = s->n_inputs; info[1] = s->n_aux; info[2] = s->n_constraints;
    for (int k = 0; k < 3; k++) info[3 + k] = s->col[k].size();
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
extern "C" const void* zg_synth_ptr(const zg_synth* s, int which) {
    if (!s || which < 0 || which > 9) return nullptr;
    if (which < 3) return s->row_ptr[which].data();
    if (which < 6) return s->col[which - 3].data();
    if (which < 9) return s->coeff[which - 6].data();
    return s->witness.data();
}
extern "C" void zg_synth_free(zg_synth* s) { delete s; }

// ---- what `za setup` and the evaluator would leave on disk for a synthetic circuit: the proving.key container
// (prover/src/groth16/format.rs:223-252) around a Parameters body, and the .binwitness dump
// (compiler/src/tester/embeeded.rs:47-57, fs.rs:116-126).  Signal id = variable index (ONE, public inputs, aux), no
// ignored signals, an empty AST vector.  The container stores a*b + c = 0 (constraint.rs:51): C's coefficients are negated.
namespace {
struct FileOut {
    FILE* f;
    std::vector<uint8_t> buf;
    explicit FileOut(const char* path) : f(fopen(path, "wb")) { buf.reserve(1 << 22); }
    ~FileOut() { if (f) fclose(f); }
    bool flush() {
        const bool ok = buf.empty() || fwrite(buf.data(), 1, buf.size(), f) == buf.size();
        buf.clear();
        return ok;
    }
    bool put(const void* p, size_t n) {
        if (buf.size() + n > buf.capacity() && !flush()) return false;
        if (n > buf.capacity()) return fwrite(p, 1, n, f) == n;
        buf.insert(buf.end(), (const uint8_t*)p, (const uint8_t*)p + n);
        return true;
    }
    bool u32be(uint32_t v) {
        const uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
        return put(b, 4);
    }
    bool u64le(uint64_t v) { return put(&v, 8); }
};
}  // namespace
extern "C" int zg_synth_pk_container_write(const char* path, uint32_t n_inputs, uint32_t n_aux, uint64_t n_constraints,
                                           const uint64_t* const row_ptr[3], const uint32_t* const col[3],
                                           const uint8_t* const coeff[3], const uint8_t* params, size_t params_len) try {
    if (!path || !row_ptr || !col || !coeff || !params || n_constraints > 0xffffffffull) return ZG_E_INVALID_ARG;
    FileOut o(path);
    if (!o.f) return ZG_E_IO;
    bool ok = o.u32be(8) && o.u64le(0) && o.u32be((uint32_t)n_constraints);
    std::vector<uint8_t> q;
    for (uint64_t i = 0; ok && i < n_constraints; i++) {
        q.clear();
        for (int k = 0; k < 3; k++) {
            const uint64_t e0 = row_ptr[k][i], e1 = row_ptr[k][i + 1], cnt = e1 - e0;
            q.insert(q.end(), (const uint8_t*)&cnt, (const uint8_t*)&cnt + 8);
            for (uint64_t e = e0; e < e1; e++) {
                uint32_t w[8];
                memcpy(w, coeff[k] + 32 * e, 32);
                if (k == 2) {   // r - c (c != 0)
                    uint64_t br = 0;
                    bool nz = false;
                    for (int j = 0; j < 8; j++) nz |= w[j] != 0;
                    if (nz)
                        for (int j = 0; j < 8; j++) {
                            const uint64_t d = (uint64_t)FrParams::MOD[j] - w[j] - br;
                            w[j] = (uint32_t)d;
                            br = (d >> 32) & 1;
                        }
                }
                uint64_t nw = 8;
                while (nw && !w[nw - 1]) nw--;
                const uint64_t sig = col[k][e];
                q.insert(q.end(), (const uint8_t*)&sig, (const uint8_t*)&sig + 8);
                q.insert(q.end(), (const uint8_t*)&nw, (const uint8_t*)&nw + 8);
                q.insert(q.end(), (const uint8_t*)w, (const uint8_t*)w + 4 * nw);
            }
        }
        ok = o.u32be((uint32_t)q.size()) && o.put(q.data(), q.size());
    }
    ok = ok && o.u32be(0) && o.flush() && fwrite(params, 1, params_len, o.f) == params_len;
    (void)n_inputs; (void)n_aux;
    return ok ? ZG_OK : ZG_E_IO;
} catch (...) {
    return zg::on_exception(nullptr);
}
// prod_i e(P_i, Q_i) through the library's host pairing: slow = 1 the definition (Fq12 points, plain powers), 0 = what zg_verify
// runs.  g1: n x 64 bytes, g2: n x 128 bytes (Montgomery LE affine, all zero = identity); gt_out: the value (384 bytes).
extern "C" int zg_selftest_pairing_product(const uint8_t* g1, const uint8_t* g2, size_t n, int slow, uint8_t* gt_out, int* is_one) try {
    if (!g1 || !g2 || !is_one || n == 0 || n > 4) return ZG_E_INVALID_ARG;
    G1Affine ps[4];
    G2Affine qs[4];
    for (size_t i = 0; i < n; i++) {
        memcpy(&ps[i], g1 + 64 * i, 64);
        memcpy(&qs[i], g2 + 128 * i, 128);
    }
    *is_one = zg_internal_v1.pairing_product_is_one(qs, ps, (int)n, slow != 0, gt_out) ? 1 : 0;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
extern "C" int zg_synth_binwitness_write(const char* path, const uint8_t* witness32, size_t n) try {
    if (!path || !witness32) return ZG_E_INVALID_ARG;
    FileOut o(path);
    if (!o.f) return ZG_E_IO;
    auto w32 = [&](const uint8_t* le) {   // FS::write_256_w32: 4-byte groups from the least significant, each big-endian
        uint8_t b[32];
        for (int g = 0; g < 8; g++)
            for (int k = 0; k < 4; k++) b[4 * g + (3 - k)] = le[4 * g + k];
        return o.put(b, 32);
    };
    uint8_t cnt[32] = {0};
    const uint64_t n64 = n;
    memcpy(cnt, &n64, 8);
    bool ok = w32(cnt);
    for (size_t i = 0; ok && i < n; i++) ok = w32(witness32 + 32 * i);
    return ok && o.flush() ? ZG_OK : ZG_E_IO;
} catch (...) {
    return zg::on_exception(nullptr);
}
extern "C" uint64_t zg_synth_dlog(uint64_t seed, uint64_t i) { return splitmix64_at(seed, i); }

// ------------------------------------------------------------------------------------------------ device generators
template <class F>
__global__ void __launch_bounds__(128) k_synth_bases(Affine<F>* __restrict__ out, size_t n, uint64_t seed,
                                                     Affine<F> gen_m261) {
    using L = typename LazyOf<F>::L;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = splitmix64_hd(seed, i);
    const AffD<L> gen = AffD<L>::load(&gen_m261);
    XYZZd<L> acc = XYZZd<L>::inf();
    for (int b = 63; b >= 0; b--) {
        acc = acc.dbl();
        if ((k >> b) & 1) acc = acc.add_mixed(gen);
    }
    acc.to_affine_mem(&out[i]);   // canonical, device (M261) form
}
__global__ void __launch_bounds__(256) k_synth_scalars(uint32_t* __restrict__ out, size_t n, uint64_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 a, b;
    uint64_t v0 = splitmix64_hd(seed, 4 * i), v1 = splitmix64_hd(seed, 4 * i + 1), v2 = splitmix64_hd(seed, 4 * i + 2),
             v3 = splitmix64_hd(seed, 4 * i + 3);
    a.x = (uint32_t)v0; a.y = (uint32_t)(v0 >> 32); a.z = (uint32_t)v1; a.w = (uint32_t)(v1 >> 32);
    b.x = (uint32_t)v2; b.y = (uint32_t)(v2 >> 32); b.z = (uint32_t)v3; b.w = (uint32_t)(v3 >> 32) & 0x1fffffffu;
    reinterpret_cast<uint4*>(out)[2 * i] = a;
    reinterpret_cast<uint4*>(out)[2 * i + 1] = b;
}

static G1Affine g1_generator() { return G1Affine{Fq::from_u64(1), Fq::from_u64(2)}; }
static G2Affine g2_generator() {
    // prover/src/groth16/ethereum.rs:28-31 (printed there as [c1, c0])
    auto fq = [](uint32_t l0, uint32_t l1, uint32_t l2, uint32_t l3, uint32_t l4, uint32_t l5, uint32_t l6, uint32_t l7) {
        Fq c;
        c.l[0] = l0; c.l[1] = l1; c.l[2] = l2; c.l[3] = l3; c.l[4] = l4; c.l[5] = l5; c.l[6] = l6; c.l[7] = l7;
        return c.to_mont();
    };
    G2Affine g;
    // x.c0 = 0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed
    g.x.c0 = fq(0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu);
    // x.c1 = 0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2
    g.x.c1 = fq(0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u);
    // y.c0 = 0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa
    g.y.c0 = fq(0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u);
    // y.c1 = 0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b
    g.y.c1 = fq(0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u);
    return g;
}

// internal: bases in the device's M261 form (what zg_params keeps resident)
static int synth_bases_g1_m261(zg_ctx* ctx, void* d_out, size_t n, uint64_t seed) {
    if (n == 0) return ZG_OK;
    hipLaunchKernelGGL((k_synth_bases<Fq>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream,
                       (G1Affine*)d_out, n, seed, host_to_m261(g1_generator()));
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}
static int synth_bases_g2_m261(zg_ctx* ctx, void* d_out, size_t n, uint64_t seed) {
    if (n == 0) return ZG_OK;
    G2Affine g = g2_generator();
    if (!on_curve(g, Fq2{Fq::from_u64(3), Fq::zero()} * Fq2{Fq::from_u64(9), Fq::from_u64(1)}.inv()))
        return fail(ctx, ZG_E_INVALID_ARG, "internal: G2 generator constant is not on the twist");
    hipLaunchKernelGGL((k_synth_bases<Fq2>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream,
                       (G2Affine*)d_out, n, seed, host_to_m261(g));
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
}
// public: bases in the ABI's 2^256 Montgomery form, like any caller-supplied device bases
extern "C" int zg_synth_bases_g1_dev(zg_ctx* ctx, void* d_out, size_t n, uint64_t seed) try {
    if (!ctx || (!d_out && n)) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    ZG_TRY(synth_bases_g1_m261(ctx, d_out, n, seed));
    return convert_mont_dev(ctx, d_out, n * 2, 0, 1);
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_synth_bases_g2_dev(zg_ctx* ctx, void* d_out, size_t n, uint64_t seed) try {
    if (!ctx || (!d_out && n)) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    ZG_TRY(synth_bases_g2_m261(ctx, d_out, n, seed));
    return convert_mont_dev(ctx, d_out, n * 4, 0, 1);
} catch (...) {
    return zg::on_exception(ctx);
}
extern "C" int zg_synth_scalars_dev(zg_ctx* ctx, void* d_out, size_t n, uint64_t seed) try {
    if (!ctx || (!d_out && n)) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) return ZG_OK;
    hipLaunchKernelGGL(k_synth_scalars, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)d_out, n,
                       seed);
    ZG_HIP(ctx, hipGetLastError());
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

// A sharded view of a resident key: chunk `rank` of `world` of every query, copied into a new handle.
extern "C" int zg_params_shard(zg_ctx* ctx, const zg_params* full, int rank, int world, zg_params** out) try {
    if (!ctx || !full || !out || world < 1 || rank < 0 || rank >= world || full->world != 1) return ZG_E_INVALID_ARG;
    *out = nullptr;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<zg_params> p(new zg_params());
    p->alpha_g1 = full->alpha_g1; p->beta_g1 = full->beta_g1; p->delta_g1 = full->delta_g1;
    p->beta_g2 = full->beta_g2; p->delta_g2 = full->delta_g2;
    p->rank = rank;
    p->world = world;
    for (int k = 0; k < 5; k++) {
        auto& q = p->q[k];
        q.total = full->q[k].total;
        uint64_t chunk = (q.total + world - 1) / world;
        q.lo = std::min<uint64_t>(q.total, (uint64_t)rank * chunk);
        q.hi = std::min<uint64_t>(q.total, q.lo + chunk);
        const size_t elem = k == 4 ? 128 : 64;
        ZG_HIP(ctx, q.buf.alloc((q.hi - q.lo) * elem));
        if (q.hi > q.lo)
            ZG_HIP(ctx, hipMemcpyAsync(q.buf.p, (const uint8_t*)full->q[k].buf.p + q.lo * elem, (q.hi - q.lo) * elem,
                                       hipMemcpyDeviceToDevice, ctx->stream));
    }
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    {   // the shard's slice of the evaluation-form h bases: from the full key's transformed array if it has one, else made here
        // (sampled under the full key's lock; a built array is immutable for the key's lifetime -- prover.hip ensure_h_eval)
        const G1Affine* have = nullptr;
        {
            std::lock_guard<std::mutex> lock(full->fbw_mu);
            if (full->h_eval_state == 1 && ((uint64_t)1 << full->h_eval_log_m) == full->q[0].total + 1)
                have = full->h_eval.as<G1Affine>();
        }
        ZG_TRY(params_h_eval_slice(ctx, p.get(), full->q[0].buf.as<G1Affine>(), full->q[0].total, have));
    }
    *out = p.release();
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

extern "C" int zg_synth_params_dev(zg_ctx* ctx, uint64_t n_h, uint64_t n_l, uint64_t n_a, uint64_t n_b, uint64_t seed,
                                   int rank, int world, zg_params** out) try {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world) return ZG_E_INVALID_ARG;
    *out = nullptr;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<zg_params> p(new zg_params());
    p->rank = rank;
    p->world = world;
    const G1XYZZ g1 = G1XYZZ::from_affine(g1_generator());
    const G2XYZZ g2 = G2XYZZ::from_affine(g2_generator());
    p->alpha_g1 = g1.mul_small(3).to_affine();
    p->beta_g1 = g1.mul_small(5).to_affine();
    p->beta_g2 = g2.mul_small(5).to_affine();
    p->delta_g1 = g1.mul_small(7).to_affine();
    p->delta_g2 = g2.mul_small(7).to_affine();
    const uint64_t totals[5] = {n_h, n_l, n_a, n_b, n_b};
    const uint64_t seeds[5] = {seed + 1, seed + 2, seed + 3, seed + 4, seed + 4};
    for (int k = 0; k < 5; k++) {
        auto& q = p->q[k];
        q.total = totals[k];
        uint64_t chunk = (q.total + world - 1) / world;
        q.lo = std::min<uint64_t>(q.total, (uint64_t)rank * chunk);
        q.hi = std::min<uint64_t>(q.total, q.lo + chunk);
        const size_t elem = k == 4 ? 128 : 64;
        ZG_HIP(ctx, q.buf.alloc((q.hi - q.lo) * elem));
        // generator index i maps to global position lo + i: fold lo into the seed stream position
        if (q.hi > q.lo) {
            const size_t cnt = q.hi - q.lo;
            // splitmix64_hd(seed, lo + i) == splitmix64_hd(seed + lo * GOLDEN, i)
            const uint64_t s2 = seeds[k] + q.lo * 0x9E3779B97F4A7C15ull;
            if (k == 4) ZG_TRY(synth_bases_g2_m261(ctx, q.buf.p, cnt, s2));
            else ZG_TRY(synth_bases_g1_m261(ctx, q.buf.p, cnt, s2));
        }
    }
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = p.release();
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

// Copy a resident query shard (which = 0..4) or the five vk points (which = -1, 448 bytes in the
// zg_params_from_mont order) back to the host -- lets tests and the CPU baseline see a synthetic key.
extern "C" int zg_params_download(const zg_params* p, int which, void* out, size_t bytes) try {
    if (!p || !out || which < -1 || which > 4) return ZG_E_INVALID_ARG;
    if (which == -1) {
        if (bytes < 448) return ZG_E_BUFFER_TOO_SMALL;
        uint8_t* o = (uint8_t*)out;
        memcpy(o, &p->alpha_g1, 64);
        memcpy(o + 64, &p->beta_g1, 64);
        memcpy(o + 128, &p->beta_g2, 128);
        memcpy(o + 256, &p->delta_g1, 64);
        memcpy(o + 320, &p->delta_g2, 128);
        return ZG_OK;
    }
    const auto& q = p->q[which];
    const size_t elem = which == 4 ? 128 : 64;
    const size_t have = (q.hi - q.lo) * elem;
    if (bytes < have) return ZG_E_BUFFER_TOO_SMALL;
    if (have && hipMemcpy(out, q.buf.p, have, hipMemcpyDeviceToHost) != hipSuccess) return ZG_E_HIP;
    // resident bases are in the device's M261 form: hand back the ABI's 2^256 form
    Fq* f = reinterpret_cast<Fq*>(out);
    for (size_t i = 0; i < have / 32; i++) {
        Fq v;
        memcpy(&v, f + i, 32);
        v = m261_to_host(v);
        memcpy(f + i, &v, 32);
    }
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

// ------------------------------------------------------------------------------------------------ self-test hooks
// Host-side (no GPU) and device-side entry points that expose the product's own field / curve
// arithmetic and the key parser to the tests.
#include "formats.h"

template <class F>
static void st_field_host(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        F x, y, z;
        memcpy(&x, a + 32 * i, 32);
        memcpy(&y, b + 32 * i, 32);
        switch (op) {
            case 0: z = x * y; break;
            case 1: z = x + y; break;
            case 2: z = x - y; break;
            case 3: z = x.inv(); break;
            case 4: z = x.to_mont(); break;
            case 5: z = x.from_mont(); break;
            case 6: z = x.neg(); break;
            default: z = x.sqr(); break;
        }
        memcpy(out + 32 * i, &z, 32);
    }
}
// which: 0 Fq, 1 Fr.  op: 0 mul 1 add 2 sub 3 inv 4 to_mont 5 from_mont 6 neg 7 sqr.  Montgomery LE operands.
extern "C" int zg_selftest_field_host(int which, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) try {
    if (!a || !b || !out) return ZG_E_INVALID_ARG;
    if (which) st_field_host<Fr>(op, a, b, out, n); else st_field_host<Fq>(op, a, b, out, n);
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
// operands arrive in the ABI's 2^256 form; the op runs in the device field (fd.h) and converts back
template <class P29>
__global__ void k_st_field(int op, const Fp<typename P29::Wide>* a, const Fp<typename P29::Wide>* b,
                           Fp<typename P29::Wide>* out, size_t n) {
    using D = Fd<P29>;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    D x = D::load_canon(&a[i]).m256_to_m261(), y = D::load_canon(&b[i]).m256_to_m261(), z;
    switch (op) {
        case 0: z = x * y; break;
        case 1: z = x + y; break;
        case 2: z = D::template sub<2>(x, y); break;
        case 3: z = x.inv(); break;
        case 4: z = D::load_canon(&a[i]) * D::r2(); break;                 // integer a -> a * 2^256 once converted back
        case 5: x.from_mont().store(&out[i]); return;                       // a * 2^256 -> integer a
        case 6: z = x.template neg<2>(); break;
        default: z = x.sqr(); break;
    }
    z.m261_to_m256().canonical().store(&out[i]);
}
extern "C" int zg_selftest_field_dev(zg_ctx* ctx, int which, int op, const uint8_t* a, const uint8_t* b, uint8_t* out,
                                     size_t n) try {
    if (!ctx || !a || !b || !out) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf da, db, dout;
    ZG_HIP(ctx, da.alloc(n * 32)); ZG_HIP(ctx, db.alloc(n * 32)); ZG_HIP(ctx, dout.alloc(n * 32));
    ZG_HIP(ctx, hipMemcpyAsync(da.p, a, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZG_HIP(ctx, hipMemcpyAsync(db.p, b, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if (n) {
        if (which)
            hipLaunchKernelGGL((k_st_field<Fr29>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream, op,
                               da.as<Fr>(), db.as<Fr>(), dout.as<Fr>(), n);
        else
            hipLaunchKernelGGL((k_st_field<Fq29>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream, op,
                               da.as<Fq>(), db.as<Fq>(), dout.as<Fq>(), n);
    }
    ZG_HIP(ctx, hipMemcpyAsync(out, dout.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}
// out = k * p (+ q if q != NULL) with the product's HOST curve code; group 1 = G1 (64 B points), 2 = G2 (128 B)
extern "C" int zg_selftest_curve_host(int group, const uint8_t* p, const uint8_t* k32, const uint8_t* q, uint8_t* out) try {
    if (!p || !k32 || !out) return ZG_E_INVALID_ARG;
    uint32_t k[8];
    memcpy(k, k32, 32);
    if (group == 1) {
        G1Affine a, b;
        memcpy(&a, p, 64);
        G1XYZZ r = G1XYZZ::from_affine(a).mul(k);
        if (q) { memcpy(&b, q, 64); r = r.add_mixed(b); }
        G1Affine o = r.to_affine();
        memcpy(out, &o, 64);
    } else {
        G2Affine a, b;
        memcpy(&a, p, 128);
        G2XYZZ r = G2XYZZ::from_affine(a).mul(k);
        if (q) { memcpy(&b, q, 128); r = r.add_mixed(b); }
        G2Affine o = r.to_affine();
        memcpy(out, &o, 128);
    }
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
// ---- device point arithmetic, operation by operation (tests/test_gpu_curve.py).  The multiexp kernels only ever see
// ecd.h's additions end to end; here every form runs on the device over caller-chosen operands: one lane per point
// (FqL / Fq2L), G2 on lane pairs (Fq2PL), G1 on four lanes (FqQL).
// P, Q: affine, device form.  ZP, ZQ: field elements z != 0 that pick the XYZZ representation (x z^2, y z^3, z^2, z^3).
template <class F>
__global__ void __launch_bounds__(128) k_st_mkxyzz(const Affine<F>* __restrict__ a, const F* __restrict__ z,
                                                   XYZZ<F>* __restrict__ out, size_t n) {
    using L = typename LazyOf<F>::L;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AffD<L> p = AffD<L>::load(&a[i]);
    if (p.inf) {
        XYZZd<L>::inf().store(&out[i]);
        return;
    }
    const typename L::T zv = L::load_canon(&z[i]);
    const typename L::T zz = L::sqr(zv), zzz = L::mul(zz, zv);
    XYZZd<L>{L::mul(p.x, zz), L::mul(p.y, zzz), zz, zzz}.store(&out[i]);
}
// op: 0 A + B (both XYZZ), 1 A + Q (mixed), 2 A - Q (mixed, y negated lazily as k_accumulate does), 3 2A,
//     4 (A + Q) + Q and 5 (A + B) + A: the second addition's operands are the first one's outputs (bounds of a chain)
template <class F, class L, int LP>
__global__ void __launch_bounds__(128) k_st_curve(int op, const XYZZ<F>* __restrict__ A, const XYZZ<F>* __restrict__ B,
                                                  const Affine<F>* __restrict__ Q, XYZZ<F>* __restrict__ out, size_t n) {
    const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LP;
    if (t >= n) return;   // (the LP lanes of a point leave together)
    const XYZZd<L> a = XYZZd<L>::load(&A[t]);
    XYZZd<L> r = XYZZd<L>::inf();
    switch (op) {
        case 0: r = a.add(XYZZd<L>::load(&B[t])); break;
        case 1: r = a.add_mixed(AffD<L>::load(&Q[t])); break;
        case 2: r = a.add_mixed(AffD<L>::load(&Q[t]).negated_lazy()); break;
        case 3: r = a.dbl(); break;
        case 4: { const AffD<L> q = AffD<L>::load(&Q[t]); r = a.add_mixed(q).add_mixed(q); break; }
        default: r = a.add(XYZZd<L>::load(&B[t])).add(a); break;
    }
    r.store(&out[t]);
}
template <class F>
__global__ void __launch_bounds__(128) k_st_toaffine(const XYZZ<F>* __restrict__ in, Affine<F>* __restrict__ out, size_t n) {
    using L = typename LazyOf<F>::L;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZd<L>::load(&in[i]).to_affine_mem(&out[i]);
}
template <class F>
static int st_curve_dev(zg_ctx* ctx, int lanes, int op, const uint8_t* p, const uint8_t* zp, const uint8_t* q,
                        const uint8_t* zq, uint8_t* out, size_t n) {
    constexpr bool G1 = sizeof(F) == sizeof(Fq);
    if ((lanes != 1 && lanes != (G1 ? 4 : 2)) || op < 0 || op > 5) return fail(ctx, ZG_E_INVALID_ARG, "selftest: lanes / op");
    if (n == 0) return ZG_OK;
    const size_t ab = sizeof(Affine<F>), fb = sizeof(F);
    DevBuf dp, dq, dzp, dzq, xa, xb, xo, ao;
    ZG_HIP(ctx, dp.alloc(n * ab)); ZG_HIP(ctx, dq.alloc(n * ab)); ZG_HIP(ctx, dzp.alloc(n * fb)); ZG_HIP(ctx, dzq.alloc(n * fb));
    ZG_HIP(ctx, xa.alloc(n * sizeof(XYZZ<F>))); ZG_HIP(ctx, xb.alloc(n * sizeof(XYZZ<F>))); ZG_HIP(ctx, xo.alloc(n * sizeof(XYZZ<F>)));
    ZG_HIP(ctx, ao.alloc(n * ab));
    hipStream_t st = ctx->stream;
    ZG_HIP(ctx, hipMemcpyAsync(dp.p, p, n * ab, hipMemcpyHostToDevice, st));
    ZG_HIP(ctx, hipMemcpyAsync(dq.p, q, n * ab, hipMemcpyHostToDevice, st));
    ZG_HIP(ctx, hipMemcpyAsync(dzp.p, zp, n * fb, hipMemcpyHostToDevice, st));
    ZG_HIP(ctx, hipMemcpyAsync(dzq.p, zq, n * fb, hipMemcpyHostToDevice, st));
    ZG_TRY(convert_mont_dev(ctx, dp.p, n * ab / 32, 0, 0));   // ABI form -> device form, canonical
    ZG_TRY(convert_mont_dev(ctx, dq.p, n * ab / 32, 0, 0));
    ZG_TRY(convert_mont_dev(ctx, dzp.p, n * fb / 32, 0, 0));
    ZG_TRY(convert_mont_dev(ctx, dzq.p, n * fb / 32, 0, 0));
    const unsigned g1 = (unsigned)((n + 127) / 128);
    hipLaunchKernelGGL((k_st_mkxyzz<F>), dim3(g1), dim3(128), 0, st, dp.as<Affine<F>>(), dzp.as<F>(), xa.as<XYZZ<F>>(), n);
    hipLaunchKernelGGL((k_st_mkxyzz<F>), dim3(g1), dim3(128), 0, st, dq.as<Affine<F>>(), dzq.as<F>(), xb.as<XYZZ<F>>(), n);
    const unsigned gl = (unsigned)((n * (size_t)lanes + 127) / 128);
    if constexpr (G1) {
        if (lanes == 4)
            hipLaunchKernelGGL((k_st_curve<F, FqQL, 4>), dim3(gl), dim3(128), 0, st, op, xa.as<XYZZ<F>>(), xb.as<XYZZ<F>>(),
                               dq.as<Affine<F>>(), xo.as<XYZZ<F>>(), n);
        else
            hipLaunchKernelGGL((k_st_curve<F, FqL, 1>), dim3(gl), dim3(128), 0, st, op, xa.as<XYZZ<F>>(), xb.as<XYZZ<F>>(),
                               dq.as<Affine<F>>(), xo.as<XYZZ<F>>(), n);
    } else {
        if (lanes == 2)
            hipLaunchKernelGGL((k_st_curve<F, Fq2PL, 2>), dim3(gl), dim3(128), 0, st, op, xa.as<XYZZ<F>>(), xb.as<XYZZ<F>>(),
                               dq.as<Affine<F>>(), xo.as<XYZZ<F>>(), n);
        else
            hipLaunchKernelGGL((k_st_curve<F, Fq2L, 1>), dim3(gl), dim3(128), 0, st, op, xa.as<XYZZ<F>>(), xb.as<XYZZ<F>>(),
                               dq.as<Affine<F>>(), xo.as<XYZZ<F>>(), n);
    }
    hipLaunchKernelGGL((k_st_toaffine<F>), dim3(g1), dim3(128), 0, st, xo.as<XYZZ<F>>(), ao.as<Affine<F>>(), n);
    ZG_HIP(ctx, hipGetLastError());
    ZG_TRY(convert_mont_dev(ctx, ao.p, n * ab / 32, 0, 1));
    ZG_HIP(ctx, hipMemcpyAsync(out, ao.p, n * ab, hipMemcpyDeviceToHost, st));
    ZG_HIP(ctx, hipStreamSynchronize(st));
    return ZG_OK;
}
extern "C" int zg_selftest_curve_dev(zg_ctx* ctx, int group, int lanes, int op, const uint8_t* p, const uint8_t* zp,
                                     const uint8_t* q, const uint8_t* zq, uint8_t* out, size_t n) try {
    if (!ctx || !p || !zp || !q || !zq || !out) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    if (group == 1) return st_curve_dev<Fq>(ctx, lanes, op, p, zp, q, zq, out, n);
    if (group == 2) return st_curve_dev<Fq2>(ctx, lanes, op, p, zp, q, zq, out, n);
    return fail(ctx, ZG_E_INVALID_ARG, "selftest: group %d", group);
} catch (...) {
    return zg::on_exception(ctx);
}
// zg_prove's host-side packer (prover.hip HostPacker) on plain memory, no GPU
extern "C" int zg_selftest_pack_mt(const uint8_t* witness32, size_t nv, int threads, uint8_t* classes, uint32_t* other_index,
                                   uint8_t* others32, size_t* n_others) try {
    if (!witness32 || !classes || !other_index || !others32 || !n_others || threads < 1) return ZG_E_INVALID_ARG;
    return zg::host_pack_for_test(witness32, nv, threads, classes, other_index, others32, n_others);
} catch (...) {
    return zg::on_exception(nullptr);
}
// ... shared with a stand-in copy engine (HostPacker::RawCopier): what went through it lands in raw_out, *first = how many values
extern "C" int zg_selftest_pack_split(const uint8_t* witness32, size_t nv, int threads, int cap_permille, int ready_every,
                                      uint8_t* raw_out, uint8_t* classes, uint32_t* other_index, uint8_t* others32,
                                      size_t* n_others, size_t* first) try {
    if (!witness32 || !raw_out || !classes || !other_index || !others32 || !n_others || !first || threads < 1) return ZG_E_INVALID_ARG;
    return zg::host_pack_split_for_test(witness32, nv, threads, cap_permille, ready_every, raw_out, classes, other_index, others32,
                                        n_others, first);
} catch (...) {
    return zg::on_exception(nullptr);
}
// Parse and validate a bellman Parameters body without a GPU.  counts: |ic|, |h|, |l|, |a|, |b_g1|, |b_g2|, consumed.
// err (nullable) receives the parser's message.
extern "C" int zg_params_inspect(const uint8_t* buf, size_t len, int checked, uint64_t counts[7], char* err, size_t err_cap) try {
    if (!buf || !counts) return ZG_E_INVALID_ARG;
    zg_ctx tmp;
    ParsedParams pp;
    size_t used = 0;
    int rc = parse_bellman_params(&tmp, buf, len, checked != 0, &pp, &used);
    if (err && err_cap) {
        strncpy(err, tmp.last_error.c_str(), err_cap - 1);
        err[err_cap - 1] = 0;
    }
    if (rc) return rc;
    counts[0] = pp.ic.size(); counts[1] = pp.h.size(); counts[2] = pp.l.size(); counts[3] = pp.a.size();
    counts[4] = pp.b_g1.size(); counts[5] = pp.b_g2.size(); counts[6] = used;
    // round trip: Parameters::write(Parameters::read(x)) == x
    std::vector<uint8_t> again;
    write_bellman_params(pp, &again);
    if (again.size() != used || memcmp(again.data(), buf, used) != 0) return ZG_E_BAD_FORMAT;
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
