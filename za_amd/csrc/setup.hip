// Trapdoor set-up on the GPU.
//
// Replaces bellman_ce::groth16::generate_parameters(circuit, g1, g2, alpha, beta, gamma, delta, tau), the
// deterministic core of generate_random_parameters that za calls at prover/src/groth16/prover.rs:122, followed by
// Parameters::write (format.rs:250).  With the 2^log_m domain's Lagrange basis evaluated at tau (inverse NTT of the
// powers of tau, as bellman does), every query point is a fixed-base multiple of the generator:
//   h_i   = tau^i * Z(tau)/delta * G1                     i < m-1
//   a_v   = A_v(tau) * G1,  b_v = B_v(tau) * G1 / G2       (identity entries dropped -> the density maps)
//   ic/l  = (beta A_v + alpha B_v + C_v)(tau) / gamma|delta * G1
// A_v(tau) = sum_j coeff * L_j(tau) is the transposed sparse product, done with the R1CS kernel on CSC arrays.
// The generators are the standard ones ((1,2) and the EIP-197 G2 point); bellman draws random ones, which only
// re-randomises the key.
#include <algorithm>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "common.h"
#include "formats.h"
#include "handles.h"

using namespace zg;

namespace {

constexpr int FB_WBITS = 8, FB_WINDOWS = 32, FB_ENTRIES = 255;

// table[w * 255 + d - 1] = d * 2^(8 w) * G   (affine, device form)
template <class F>
__global__ void __launch_bounds__(64) k_fb_table(Affine<F>* __restrict__ table, Affine<F> gen_m261) {
    using L = typename LazyOf<F>::L;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FB_WINDOWS * FB_ENTRIES) return;
    const int w = t / FB_ENTRIES, d = t % FB_ENTRIES + 1;
    const AffD<L> gen = AffD<L>::load(&gen_m261);
    XYZZd<L> acc = XYZZd<L>::inf();
    for (int b = 7; b >= 0; b--) {   // d * G
        acc = acc.dbl();
        if ((d >> b) & 1) acc = acc.add_mixed(gen);
    }
    for (int k = 0; k < FB_WBITS * w; k++) acc = acc.dbl();
    acc.to_affine_mem(&table[t]);
}
// out[i] = scalar[i] * G  (scalars: plain integers)
template <class F>
__global__ void __launch_bounds__(128) k_fixed_base(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ scalars,
                                                    size_t n, Affine<F>* __restrict__ out) {
    using L = typename LazyOf<F>::L;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + 8 * i);
    const uint4 s0 = sp[0], s1 = sp[1];
    const uint32_t sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    XYZZd<L> acc = XYZZd<L>::inf();
    for (int w = 0; w < FB_WINDOWS; w++) {
        const uint32_t d = (sw[w >> 2] >> (8 * (w & 3))) & 0xffu;
        if (d) acc = acc.add_mixed(AffD<L>::load(&table[w * FB_ENTRIES + d - 1]));
    }
    acc.to_affine_mem(&out[i]);
}
// flags[v] bit0: at != 0, bit1: bt != 0, bit2: ext != 0 ; integer forms of at, bt and ext written for the scalar muls
__global__ void __launch_bounds__(256) k_setup_scalars(const Fr* __restrict__ at, const Fr* __restrict__ bt,
                                                       const Fr* __restrict__ ct, uint32_t n_inputs, size_t nv, Fr alpha_m,
                                                       Fr beta_m, Fr ginv_m, Fr dinv_m, Fr* __restrict__ at_int,
                                                       Fr* __restrict__ bt_int, Fr* __restrict__ ext_int,
                                                       uint8_t* __restrict__ flags) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const FrD a = FrD::load_red(&at[v]), b = FrD::load_red(&bt[v]), c = FrD::load_red(&ct[v]);
    FrD e = FrD::load_canon(&beta_m) * a + FrD::load_canon(&alpha_m) * b + c;
    e = e * FrD::load_canon(v < n_inputs ? &ginv_m : &dinv_m);
    const FrD ai = a.from_mont(), bi = b.from_mont(), ei = e.from_mont();
    ai.store(&at_int[v]);
    bi.store(&bt_int[v]);
    ei.store(&ext_int[v]);
    flags[v] = (uint8_t)((ai.is_exact_zero() ? 0 : 1) | (bi.is_exact_zero() ? 0 : 2) | (ei.is_exact_zero() ? 0 : 4));
}
// at[i] += L_{n + i}(tau) for the `input_i * 0 = 0` rows bellman appends (integers in, M261 accumulate)
__global__ void k_add_input_rows(Fr* __restrict__ at, const Fr* __restrict__ lag_int, uint64_t n_constraints,
                                 uint32_t n_inputs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inputs) return;
    (FrD::load_red(&at[i]) + FrD::to_mont(FrD::load_canon(&lag_int[n_constraints + i]))).reduce().store(&at[i]);
}
template <class T>
__global__ void __launch_bounds__(256) k_gather(const T* __restrict__ in, const uint32_t* __restrict__ idx, size_t n,
                                                T* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}

G1Affine g1_gen() { return G1Affine{Fq::from_u64(1), Fq::from_u64(2)}; }
G2Affine g2_gen() {
    auto fq = [](uint32_t l0, uint32_t l1, uint32_t l2, uint32_t l3, uint32_t l4, uint32_t l5, uint32_t l6, uint32_t l7) {
        Fq c;
        c.l[0] = l0; c.l[1] = l1; c.l[2] = l2; c.l[3] = l3; c.l[4] = l4; c.l[5] = l5; c.l[6] = l6; c.l[7] = l7;
        return c.to_mont();
    };
    G2Affine g;   // prover/src/groth16/ethereum.rs:28-31
    g.x.c0 = fq(0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu);
    g.x.c1 = fq(0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u);
    g.y.c0 = fq(0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u);
    g.y.c1 = fq(0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u);
    return g;
}

template <class F>
int fixed_base(zg_ctx* ctx, const Affine<F>& gen_host, const Fr* d_scalars_int, size_t n, Affine<F>* d_out) {
    DevBuf table;
    ZG_HIP(ctx, table.alloc((size_t)FB_WINDOWS * FB_ENTRIES * sizeof(Affine<F>)));
    hipLaunchKernelGGL((k_fb_table<F>), dim3((FB_WINDOWS * FB_ENTRIES + 63) / 64), dim3(64), 0, ctx->stream,
                       table.as<Affine<F>>(), host_to_m261(gen_host));
    if (n)
        hipLaunchKernelGGL((k_fixed_base<F>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream,
                           table.as<Affine<F>>(), reinterpret_cast<const uint32_t*>(d_scalars_int), n, d_out);
    ZG_HIP(ctx, hipGetLastError());
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));   // table goes out of scope
    return ZG_OK;
}

struct Csc {
    std::vector<uint64_t> ptr;
    std::vector<uint32_t> row;
    std::vector<Fr> coeff;
};

}  // namespace

// Parameters::write (format.rs:250) of a resident, unsharded key: vk (alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1,
// delta_g2, ic) from the host copy, then h, l, a, b_g1, b_g2: every query is ENCODED ON THE DEVICE (keyload.hip
// k_key_encode: out of Montgomery form, big-endian, identity flag) and copied straight into the caller's buffer -- at 2^22
// constraints 0.96 GB in ~0.1 s where the host loop (two Montgomery products and 32 byte stores per coordinate on one thread)
// took 4 s, twice per `setup` (size query + write).
static size_t params_write_size(const zg_params* p, size_t vk_bytes) {
    size_t n = vk_bytes;
    for (int k = 0; k < 5; k++) n += 4 + (size_t)p->q[k].total * (k == 4 ? 128 : 64);
    return n;
}
extern "C" int zg_params_write(zg_ctx* ctx, const zg_params* p, uint8_t* out, size_t cap, size_t* written) try {
    if (!ctx || !p || !written) return ZG_E_INVALID_ARG;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    if (p->world != 1) return fail(ctx, ZG_E_INVALID_ARG, "Parameters::write needs an unsharded key (world = 1)");
    if (p->ic.empty()) return fail(ctx, ZG_E_MALFORMED_VERIFYING_KEY, "this key was built without gamma_g2 / ic");
    ParsedParams pp;
    pp.alpha_g1 = p->alpha_g1; pp.beta_g1 = p->beta_g1; pp.beta_g2 = p->beta_g2;
    pp.gamma_g2 = p->gamma_g2;
    pp.delta_g1 = p->delta_g1; pp.delta_g2 = p->delta_g2;
    pp.ic = p->ic;
    std::vector<uint8_t> vk;
    write_bellman_vk(pp, &vk);
    const size_t need = params_write_size(p, vk.size());
    *written = need;
    if (!out || cap < need) return ZG_E_BUFFER_TOO_SMALL;
    memcpy(out, vk.data(), vk.size());
    size_t at = vk.size();
    KeyStage stage;
    ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 5; k++) {
        const uint32_t cnt = (uint32_t)p->q[k].total;
        out[at] = (uint8_t)(cnt >> 24); out[at + 1] = (uint8_t)(cnt >> 16); out[at + 2] = (uint8_t)(cnt >> 8); out[at + 3] = (uint8_t)cnt;
        at += 4;
        ZG_TRY(key_encode_query_dev(ctx, k, p->q[k].buf.p, p->q[k].total, out + at, &stage));
        at += (size_t)p->q[k].total * (k == 4 ? 128 : 64);
    }
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

extern "C" int zg_params_write_vk(const zg_params* p, uint8_t* out, size_t cap, size_t* written) try {
    if (!p || !written) return ZG_E_INVALID_ARG;
    if (p->ic.empty()) return fail(nullptr, ZG_E_MALFORMED_VERIFYING_KEY, "this key was built without gamma_g2 / ic");
    ParsedParams pp;
    pp.alpha_g1 = p->alpha_g1; pp.beta_g1 = p->beta_g1; pp.beta_g2 = p->beta_g2;
    pp.gamma_g2 = p->gamma_g2;
    pp.delta_g1 = p->delta_g1; pp.delta_g2 = p->delta_g2;
    pp.ic = p->ic;
    std::vector<uint8_t> bytes;
    write_bellman_vk(pp, &bytes);
    *written = bytes.size();
    if (!out || cap < bytes.size()) return ZG_E_BUFFER_TOO_SMALL;
    memcpy(out, bytes.data(), bytes.size());
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

extern "C" int zg_setup(zg_ctx* ctx, const zg_r1cs* r, const uint8_t toxic[160], zg_params** out, uint8_t* params_out,
                        size_t cap, size_t* written) try {
    if (!ctx || !r || !toxic || !out) return ZG_E_INVALID_ARG;
    *out = nullptr;
    ZG_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    Fr tw[5];   // alpha, beta, gamma, delta, tau in host Montgomery form
    for (int k = 0; k < 5; k++) {
        Fr c;
        memcpy(&c, toxic + 32 * k, 32);
        tw[k] = c.to_mont();
    }
    const Fr &alpha = tw[0], &beta = tw[1], &gamma = tw[2], &delta = tw[3], &tau = tw[4];
    if (gamma.is_zero() || delta.is_zero()) return fail(ctx, ZG_E_DIVISION_BY_ZERO, "gamma or delta is zero");
    const uint32_t ni = r->n_inputs, na = r->n_aux;
    const size_t nv = (size_t)ni + na, m = (size_t)1 << r->log_m;
    const uint64_t nc = r->n_constraints;

    // ---- Lagrange basis at tau (plain integers)
    DevBuf lag;
    ZG_HIP(ctx, lag.alloc(m * 32));
    ZG_TRY(lagrange_at_tau_dev(ctx, r->log_m, tau, lag.as<Fr>()));

    // ---- transpose the three matrices on the host (one-off), evaluate A_v, B_v, C_v at tau on the GPU
    DevBuf abct[3];
    std::vector<uint8_t> mention_a(nv, 0), mention_b(nv, 0);
    for (int k = 0; k < 3; k++) {
        std::vector<uint64_t> rp(nc + 1);
        ZG_HIP(ctx, hipMemcpyAsync(rp.data(), r->row_ptr[k].p, (nc + 1) * 8, hipMemcpyDeviceToHost, st));
        ZG_HIP(ctx, hipStreamSynchronize(st));
        const size_t nz = rp[nc];
        std::vector<uint32_t> col(nz);
        std::vector<Fr> cf(nz);
        if (nz) {
            ZG_HIP(ctx, hipMemcpyAsync(col.data(), r->col[k].p, nz * 4, hipMemcpyDeviceToHost, st));
            ZG_HIP(ctx, hipMemcpyAsync(cf.data(), r->coeff[k].p, nz * 32, hipMemcpyDeviceToHost, st));
            ZG_HIP(ctx, hipStreamSynchronize(st));
        }
        Csc t;
        t.ptr.assign(nv + 1, 0);
        for (size_t e = 0; e < nz; e++) t.ptr[col[e] + 1]++;
        for (size_t v = 0; v < nv; v++) t.ptr[v + 1] += t.ptr[v];
        t.row.resize(nz);
        t.coeff.resize(nz);
        std::vector<uint64_t> cur(t.ptr.begin(), t.ptr.end() - 1);
        for (uint64_t row = 0; row < nc; row++)
            for (uint64_t e = rp[row]; e < rp[row + 1]; e++) {
                const uint64_t pos = cur[col[e]]++;
                t.row[pos] = (uint32_t)row;
                t.coeff[pos] = cf[e];
                if (k == 0) mention_a[col[e]] = 1;
                if (k == 1) mention_b[col[e]] = 1;
            }
        DevBuf dptr, drow, dcf;
        ZG_HIP(ctx, dptr.alloc((nv + 1) * 8));
        ZG_HIP(ctx, drow.alloc(nz * 4));
        ZG_HIP(ctx, dcf.alloc(nz * 32));
        ZG_HIP(ctx, abct[k].alloc(nv * 32));
        ZG_HIP(ctx, hipMemcpyAsync(dptr.p, t.ptr.data(), (nv + 1) * 8, hipMemcpyHostToDevice, st));
        if (nz) {
            ZG_HIP(ctx, hipMemcpyAsync(drow.p, t.row.data(), nz * 4, hipMemcpyHostToDevice, st));
            ZG_HIP(ctx, hipMemcpyAsync(dcf.p, t.coeff.data(), nz * 32, hipMemcpyHostToDevice, st));
        }
        ZG_TRY(spmv_dev(ctx, dptr.as<uint64_t>(), drow.as<uint32_t>(), dcf.as<Fr>(), lag.as<Fr>(), nv, abct[k].as<Fr>()));
        ZG_HIP(ctx, hipStreamSynchronize(st));
    }
    hipLaunchKernelGGL(k_add_input_rows, dim3((ni + 255) / 256), dim3(256), 0, st, abct[0].as<Fr>(), lag.as<Fr>(), nc, ni);

    // ---- scalars of the query points
    DevBuf at_int, bt_int, ext_int, flags_d, h_int;
    ZG_HIP(ctx, at_int.alloc(nv * 32));
    ZG_HIP(ctx, bt_int.alloc(nv * 32));
    ZG_HIP(ctx, ext_int.alloc(nv * 32));
    ZG_HIP(ctx, flags_d.alloc(nv));
    const Fr ginv = gamma.inv(), dinv = delta.inv();
    hipLaunchKernelGGL(k_setup_scalars, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, abct[0].as<Fr>(),
                       abct[1].as<Fr>(), abct[2].as<Fr>(), ni, nv, host_to_m261(alpha), host_to_m261(beta),
                       host_to_m261(ginv), host_to_m261(dinv), at_int.as<Fr>(), bt_int.as<Fr>(), ext_int.as<Fr>(),
                       flags_d.as<uint8_t>());
    ZG_HIP(ctx, hipGetLastError());
    const Fr zt = tau.pow_u64(m) - Fr::one();
    const size_t n_h = m - 1;
    ZG_HIP(ctx, h_int.alloc(std::max<size_t>(n_h, 1) * 32));
    ZG_TRY(powers_int_dev(ctx, h_int.as<Fr>(), n_h, tau, zt * dinv));
    std::vector<uint8_t> flags(nv);
    ZG_HIP(ctx, hipMemcpyAsync(flags.data(), flags_d.p, nv, hipMemcpyDeviceToHost, st));
    ZG_HIP(ctx, hipStreamSynchronize(st));

    // ---- densities: the generator keeps non-identity points, the prover counts mentions; they must agree
    std::vector<uint32_t> keep_a, keep_b;
    for (size_t v = 0; v < nv; v++) {
        const bool a_nz = flags[v] & 1, b_nz = flags[v] & 2;
        if (v < ni) {
            if (!a_nz) return fail(ctx, ZG_E_BAD_FORMAT, "setup: A query of input %zu is the identity", v);
        } else if (a_nz != (mention_a[v] != 0)) {
            return fail(ctx, ZG_E_BAD_FORMAT, "setup: A(tau) of variable %zu vanishes although it is mentioned (or vice versa)", v);
        }
        if (b_nz != (mention_b[v] != 0))
            return fail(ctx, ZG_E_BAD_FORMAT, "setup: B(tau) of variable %zu vanishes although it is mentioned (or vice versa)", v);
        if (a_nz) keep_a.push_back((uint32_t)v);
        if (b_nz) keep_b.push_back((uint32_t)v);
        if (v >= ni && !(flags[v] & 4))
            return fail(ctx, ZG_E_UNCONSTRAINED_VARIABLE, "setup: aux variable %zu is unconstrained", v - ni);
    }

    // ---- fixed-base multiplications
    std::unique_ptr<zg_params> p(new zg_params());
    p->rank = 0;
    p->world = 1;
    const G1Affine g1 = g1_gen();
    const G2Affine g2 = g2_gen();
    auto set_q = [&](int which, size_t total, size_t elem) -> hipError_t {
        auto& q = p->q[which];
        q.total = total; q.lo = 0; q.hi = total;
        return q.buf.alloc(total * elem);
    };
    ZG_HIP(ctx, set_q(0, n_h, 64));
    ZG_TRY(fixed_base<Fq>(ctx, g1, h_int.as<Fr>(), n_h, p->q[0].buf.as<G1Affine>()));
    DevBuf ext_pts, a_all, b1_all, b2_all, idx_d;
    ZG_HIP(ctx, ext_pts.alloc(nv * 64));
    ZG_TRY(fixed_base<Fq>(ctx, g1, ext_int.as<Fr>(), nv, ext_pts.as<G1Affine>()));
    ZG_HIP(ctx, set_q(1, na, 64));
    if (na) ZG_HIP(ctx, hipMemcpyAsync(p->q[1].buf.p, ext_pts.as<G1Affine>() + ni, (size_t)na * 64, hipMemcpyDeviceToDevice, st));
    ZG_HIP(ctx, a_all.alloc(nv * 64));
    ZG_TRY(fixed_base<Fq>(ctx, g1, at_int.as<Fr>(), nv, a_all.as<G1Affine>()));
    ZG_HIP(ctx, b1_all.alloc(nv * 64));
    ZG_TRY(fixed_base<Fq>(ctx, g1, bt_int.as<Fr>(), nv, b1_all.as<G1Affine>()));
    ZG_HIP(ctx, b2_all.alloc(nv * 128));
    ZG_TRY(fixed_base<Fq2>(ctx, g2, bt_int.as<Fr>(), nv, b2_all.as<G2Affine>()));
    ZG_HIP(ctx, idx_d.alloc(std::max(keep_a.size(), keep_b.size()) * 4 + 4));
    ZG_HIP(ctx, set_q(2, keep_a.size(), 64));
    ZG_HIP(ctx, hipMemcpyAsync(idx_d.p, keep_a.data(), keep_a.size() * 4, hipMemcpyHostToDevice, st));
    if (!keep_a.empty())
        hipLaunchKernelGGL((k_gather<G1Affine>), dim3((unsigned)((keep_a.size() + 255) / 256)), dim3(256), 0, st,
                           a_all.as<G1Affine>(), idx_d.as<uint32_t>(), keep_a.size(), p->q[2].buf.as<G1Affine>());
    ZG_HIP(ctx, hipStreamSynchronize(st));
    ZG_HIP(ctx, set_q(3, keep_b.size(), 64));
    ZG_HIP(ctx, set_q(4, keep_b.size(), 128));
    ZG_HIP(ctx, hipMemcpyAsync(idx_d.p, keep_b.data(), keep_b.size() * 4, hipMemcpyHostToDevice, st));
    if (!keep_b.empty()) {
        hipLaunchKernelGGL((k_gather<G1Affine>), dim3((unsigned)((keep_b.size() + 255) / 256)), dim3(256), 0, st,
                           b1_all.as<G1Affine>(), idx_d.as<uint32_t>(), keep_b.size(), p->q[3].buf.as<G1Affine>());
        hipLaunchKernelGGL((k_gather<G2Affine>), dim3((unsigned)((keep_b.size() + 255) / 256)), dim3(256), 0, st,
                           b2_all.as<G2Affine>(), idx_d.as<uint32_t>(), keep_b.size(), p->q[4].buf.as<G2Affine>());
    }
    ZG_HIP(ctx, hipGetLastError());
    ZG_HIP(ctx, hipStreamSynchronize(st));

    // ---- verifying-key elements on the host
    auto words = [](const Fr& v_mont, uint32_t out8[8]) {
        Fr c = v_mont.from_mont();
        memcpy(out8, &c, 32);
    };
    uint32_t wa[8], wb[8], wg[8], wd[8];
    words(alpha, wa); words(beta, wb); words(gamma, wg); words(delta, wd);
    const G1XYZZ G1x = G1XYZZ::from_affine(g1);
    const G2XYZZ G2x = G2XYZZ::from_affine(g2);
    p->alpha_g1 = G1x.mul(wa).to_affine();
    p->beta_g1 = G1x.mul(wb).to_affine();
    p->beta_g2 = G2x.mul(wb).to_affine();
    p->delta_g1 = G1x.mul(wd).to_affine();
    p->delta_g2 = G2x.mul(wd).to_affine();

    p->gamma_g2 = G2x.mul(wg).to_affine();
    p->ic.resize(ni);
    if (ni) ZG_HIP(ctx, hipMemcpy(p->ic.data(), ext_pts.p, (size_t)ni * 64, hipMemcpyDeviceToHost));
    for (auto& x : p->ic) x = m261_to_host(x);
    if (params_out || written) {
        size_t need = 0;
        const int rc = zg_params_write(ctx, p.get(), params_out, params_out ? cap : 0, &need);   // (device-side encoding)
        if (written) *written = need;
        if (rc != ZG_OK && !(rc == ZG_E_BUFFER_TOO_SMALL && !params_out)) return rc;
    }
    *out = p.release();
    return ZG_OK;
} catch (...) {
    return zg::on_exception(ctx);
}

// generate_random_parameters: the toxic waste from the OS entropy source
extern "C" int zg_setup_random(zg_ctx* ctx, const zg_r1cs* r, zg_params** out, uint8_t* params_out, size_t cap,
                               size_t* written) try {
    std::random_device rd;
    uint8_t toxic[160];
    for (int i = 0; i < 160; i += 4) {
        uint32_t v = rd();
        memcpy(toxic + i, &v, 4);
    }
    for (int k = 0; k < 5; k++) {
        toxic[32 * k + 31] &= 0x1f;   // below 2^253 < r
        toxic[32 * k] |= 1;           // never zero
    }
    return zg_setup(ctx, r, toxic, out, params_out, cap, written);
} catch (...) {
    return zg::on_exception(ctx);
}
