// Host unit test of the device field / curve code (za_amd/csrc/fd.h, ecd.h) against the plain
// 8x32 Montgomery implementation (fp.h, ec.h), which is itself pinned to the golden vectors.
// Built with -DZG_FD_DEBUG: every Fd carries an upper bound and each sub<K> / pack asserts it,
// so the lazy-reduction constants in ecd.h are machine-checked here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../za_amd/csrc/nttd.h"

using namespace zg;

static std::mt19937_64 rng(12345);
static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            printf("CHECK failed at line %d: %s\n", __LINE__, #cond);        \
            if (++failures > 10) exit(1);                                    \
        }                                                                    \
    } while (0)

template <class Pw>
static Fp<Pw> rand_canon() {  // uniform-ish canonical integer < p
    Fp<Pw> r;
    for (;;) {
        for (int i = 0; i < 8; i++) r.l[i] = (uint32_t)rng();
        r.l[7] &= 0x3fffffffu;
        bool lt = false;
        for (int i = 7; i >= 0; i--) {
            if (r.l[i] < Pw::MOD[i]) { lt = true; break; }
            if (r.l[i] > Pw::MOD[i]) break;
        }
        if (lt) return r;
    }
}
// Fd (M261) from a canonical integer; and back to a canonical integer
template <class P29>
static Fd<P29> fd_of(const Fp<typename P29::Wide>& canon) { return Fd<P29>::to_mont(Fd<P29>::load_canon(&canon)); }
template <class P29>
static Fp<typename P29::Wide> canon_of(const Fd<P29>& v) {
    Fp<typename P29::Wide> out;
    v.from_mont().store(&out);
    return out;
}

template <class P29>
static void test_field(const char* name) {
    using W = typename P29::Wide;
    using F = Fp<W>;
    using D = Fd<P29>;
    // constants
    {
        F one_c = F::zero(); one_c.l[0] = 1;
        CHECK(canon_of<P29>(D::one()) == one_c);
        F x = rand_canon<W>();
        // m256 <-> m261: x*2^256 (host Montgomery form of x) -> x*2^261
        F xm = x.to_mont();
        D viaconv = D::load_canon(&xm).m256_to_m261();
        CHECK(canon_of<P29>(viaconv) == x);
        F back; viaconv.m261_to_m256().canonical().store(&back);
        CHECK(back == xm);
    }
    for (int it = 0; it < 20000; it++) {
        F a = rand_canon<W>(), b = rand_canon<W>();
        if (it == 0) { a = F::zero(); }
        if (it == 1) { for (int i = 0; i < 8; i++) a.l[i] = W::MOD[i]; a.l[0] -= 1; b = a; }
        F am = a.to_mont(), bm = b.to_mont();
        D ad = fd_of<P29>(a), bd = fd_of<P29>(b);
        CHECK(canon_of<P29>(ad * bd) == (am * bm).from_mont());
        CHECK(canon_of<P29>(ad.sqr()) == am.sqr().from_mont());
        CHECK(canon_of<P29>(ad + bd) == (am + bm).from_mont());
        CHECK(canon_of<P29>(D::template sub<2>(ad, bd)) == (am - bm).from_mont());
        // lazy chains: ((a+b)+(a+b)) - b*7 ... against the strict implementation
        D s = ad + bd; s = s + s; s = s + ad;                  // 2a+2b+a, bound ~ 5
        F sm = am + bm; sm = sm + sm; sm = sm + am;
        D t = D::template sub<6>(ad, s);                       // a - s
        CHECK(canon_of<P29>(t) == (am - sm).from_mont());
        D big = t;
        for (int k = 0; k < 4; k++) big = big + big;           // bound ~ 7*16 = 112 < cap
        F bigm = am - sm;
        for (int k = 0; k < 4; k++) bigm = bigm + bigm;
        CHECK(canon_of<P29>(big.reduce()) == bigm.from_mont());
        CHECK(canon_of<P29>(big * bd) == (bigm * bm).from_mont());
        CHECK(big.is_zero_mod_p() == bigm.is_zero());
        // lazy limbs: sums / differences without carry propagation feeding multiplications, one-reduction a0 b0 + a1 b1
        {
            D l2 = D::add_lazy(ad, bd);                                   // limbs < 2^30
            CHECK(canon_of<P29>(l2 * bd) == ((am + bm) * bm).from_mont());
            CHECK(canon_of<P29>(l2.norm()) == (am + bm).from_mont());
            D l3 = D::add_lazy(l2, ad);                                   // limbs < 3 * 2^29
            D dl = D::template sub_lazy<4, 31>(bd, l3);                   // b - (2a + b) + 4p
            CHECK(canon_of<P29>(dl.norm()) == (bm - (am + am + bm)).from_mont());
            D d29 = D::template sub_lazy<2, 29>(ad, bd);                  // limbs < 3 * 2^29: still a valid factor
            CHECK(canon_of<P29>(d29 * ad) == ((am - bm) * am).from_mont());
            D nb = bd.template neg_lazy<2, 29>();
            CHECK(canon_of<P29>(nb * ad) == (bm.neg() * am).from_mont());
            CHECK(canon_of<P29>(D::mac2(ad, bd, d29.norm(), nb)) == (am * bm + (am - bm) * bm.neg()).from_mont());
            CHECK(canon_of<P29>(D::mac2(ad, ad, bd, bd)) == (am * am + bm * bm).from_mont());
            // the one-multiply zero filter never misses a multiple of p
            D z = D::template sub<2>(ad, ad);
            CHECK(z.template maybe_zero_mod_p<16>() && (z + z + z).template maybe_zero_mod_p<16>());
            CHECK(D::add_lazy(z, z).template maybe_zero_mod_p<16>());
            if (!(am - bm).is_zero()) CHECK(!(d29.template maybe_zero_mod_p<16>() && d29.norm().is_zero_mod_p()));
        }
        // pack / unpack of a reduced value
        F mem; big.reduce().store(&mem);
        CHECK(canon_of<P29>(D::load_red(&mem)) == bigm.from_mont());
        if (it < 50) {
            D inv = ad.inv();
            if (!a.is_zero()) CHECK(canon_of<P29>(inv * ad) == F::one().from_mont());
        }
    }
    // values that are multiples of p must test as zero
    {
        F a = rand_canon<W>();
        D ad = fd_of<P29>(a);
        D z = D::template sub<2>(ad, ad);   // == 2p as an integer
        CHECK(z.is_zero_mod_p());
        D z2 = z + z + z;                   // 6p
        CHECK(z2.is_zero_mod_p());
        CHECK(!(z2 + D::one()).is_zero_mod_p());
        CHECK(D::zero().is_zero_mod_p());
    }
    printf("%s field ok\n", name);
}

// ---- curve: lazy XYZZ vs strict XYZZ
template <class L, class F>
static Affine<F> aff_to_m261(const Affine<F>& p);  // host M256 affine -> memory in M261
template <>
Affine<Fq> aff_to_m261<FqL, Fq>(const Affine<Fq>& p) {
    Affine<Fq> o;
    FqD::load_canon(&p.x).m256_to_m261().canonical().store(&o.x);
    FqD::load_canon(&p.y).m256_to_m261().canonical().store(&o.y);
    return o;
}
template <>
Affine<Fq2> aff_to_m261<Fq2L, Fq2>(const Affine<Fq2>& p) {
    Affine<Fq2> o;
    FqD::load_canon(&p.x.c0).m256_to_m261().canonical().store(&o.x.c0);
    FqD::load_canon(&p.x.c1).m256_to_m261().canonical().store(&o.x.c1);
    FqD::load_canon(&p.y.c0).m256_to_m261().canonical().store(&o.y.c0);
    FqD::load_canon(&p.y.c1).m256_to_m261().canonical().store(&o.y.c1);
    return o;
}
static Fq m261_to_host(const Fq& m) {
    Fq o;
    FqD::load_canon(&m).m261_to_m256().canonical().store(&o);
    return o;
}
static Fq2 m261_to_host(const Fq2& m) { return Fq2{m261_to_host(m.c0), m261_to_host(m.c1)}; }

template <class L, class F>
static bool same_point(const XYZZd<L>& d, const XYZZ<F>& h) {
    Affine<F> dm;
    d.to_affine_mem(&dm);
    Affine<F> ha = h.to_affine();
    Affine<F> dh{m261_to_host(dm.x), m261_to_host(dm.y)};
    return dh.x == ha.x && dh.y == ha.y;
}

template <class L, class F>
static void test_curve(const char* name, const Affine<F>& gen) {
    using XD = XYZZd<L>;
    using XH = XYZZ<F>;
    auto rand_pt = [&]() {
        uint32_t k[8];
        for (int i = 0; i < 8; i++) k[i] = (uint32_t)rng();
        k[7] &= 0x0fffffffu;
        return XH::from_affine(gen).mul(k).to_affine();
    };
    auto dev_aff = [&](const Affine<F>& p) {
        Affine<F> m = aff_to_m261<L, F>(p);
        return AffD<L>::load(&m);
    };
    for (int it = 0; it < 300; it++) {
        Affine<F> p = rand_pt(), q = rand_pt(), r = rand_pt();
        AffD<L> pd = dev_aff(p), qd = dev_aff(q), rd = dev_aff(r);
        XH hacc = XH::from_affine(p);
        XD dacc = XD::from_affine(pd);
        // a run of mixed adds with signs, as in bucket accumulation
        for (int k = 0; k < 6; k++) {
            bool neg = (rng() & 1) != 0;
            const Affine<F>& b = (k & 1) ? q : r;
            const AffD<L>& bd = (k & 1) ? qd : rd;
            hacc = hacc.add_mixed(neg ? b.neg() : b);
            dacc = dacc.add_mixed(neg ? ((k & 2) ? bd.negated_lazy() : bd.negated()) : bd);
            CHECK((same_point<L, F>(dacc, hacc)));
        }
        // store / load round trip
        XYZZ<F> mem;
        dacc.store(&mem);
        XD re = XD::load(&mem);
        CHECK((same_point<L, F>(re, hacc)));
        // full add, doubling, negation
        XH h2 = XH::from_affine(q).add_mixed(r);
        XD d2 = XD::from_affine(qd).add_mixed(rd);
        CHECK((same_point<L, F>(re.add(d2), hacc.add(h2))));
        CHECK((same_point<L, F>(re.dbl(), hacc.dbl())));
        CHECK((same_point<L, F>(re.add(re), hacc.dbl())));          // add() must detect doubling
        CHECK(re.add(re.neg()).is_inf());                          // P + (-P)
        CHECK((same_point<L, F>(re.dbl().dbl().add(re.neg()), hacc.dbl().dbl().add(hacc.neg()))));
        // mixed add special cases
        XD one = XD::from_affine(pd);
        CHECK((same_point<L, F>(one.add_mixed(pd), XH::from_affine(p).dbl())));
        CHECK(one.add_mixed(pd.negated()).is_inf());
        CHECK(one.add_mixed(pd.negated_lazy()).is_inf());
        CHECK((same_point<L, F>(XD::inf().add_mixed(pd.negated_lazy()).add_mixed(qd), XH::from_affine(p.neg()).add_mixed(q))));
        CHECK((same_point<L, F>(XD::from_affine(pd.negated()).add_mixed(pd.negated_lazy()), XH::from_affine(p.neg()).dbl())));
        CHECK((same_point<L, F>(XD::inf().add_mixed(pd), XH::from_affine(p))));
        Affine<F> infm = Affine<F>::inf();
        CHECK((same_point<L, F>(one.add_mixed(AffD<L>::load(&infm)), XH::from_affine(p))));
        // tree-style chain: running sums with repeated add / dbl keeps every bound in range
        XD run = XD::inf(), acc = XD::inf();
        XH hrun = XH::inf(), hac = XH::inf();
        for (int t = 0; t < 8; t++) {
            run = run.add(t & 1 ? re : d2);
            hrun = hrun.add(t & 1 ? hacc : h2);
            acc = acc.add(run);
            hac = hac.add(hrun);
        }
        XD tr = acc.add(run.neg());
        XH htr = hac.add(hrun.neg());
        for (int d = 0; d < 5; d++) { tr = tr.dbl(); htr = htr.dbl(); }
        CHECK((same_point<L, F>(tr, htr)));
    }
    printf("%s curve ok\n", name);
}

// ---- NTT butterflies: a whole pass of k stages on a small tile, exactly as k_ntt_pass schedules it (an odd stage
// count starts with one radix-2 stage, then radix-4 pairs; DIF reduces every 2nd stage and after the last one), with
// the bound tracking of ZG_FD_DEBUG checking every lazy sum / difference, against a plain radix-2 transform in Fr.
static void test_ntt_passes() {
    for (int k = 1; k <= 9; k++) {
        for (int dit = 0; dit < 2; dit++) {
            const int n = 1 << k;
            // omega of order n
            Fr g = Fr::from_u64(7);
            uint32_t e[8];
            for (int i = 0; i < 8; i++) e[i] = FrParams::MOD[i];
            e[0] -= 1;
            // (r - 1) / n
            uint32_t q[8];
            for (int i = 0; i < 8; i++) q[i] = (e[i] >> k) | (i < 7 ? (e[i + 1] << (32 - k)) : 0);
            Fr omega = g.pow(q);
            std::vector<Fr> in(n), ref(n);
            for (int i = 0; i < n; i++) in[i] = rand_canon<FrParams>().to_mont();
            // reference: out[j] = sum_i in[i] omega^(i j)
            std::vector<Fr> wpow(n);
            wpow[0] = Fr::one();
            for (int i = 1; i < n; i++) wpow[i] = wpow[i - 1] * omega;
            for (int j = 0; j < n; j++) {
                Fr acc = Fr::zero();
                for (int i = 0; i < n; i++) acc = acc + in[i] * wpow[(size_t)i * j % n];
                ref[j] = acc;
            }
            auto brev = [&](int x) { int r = 0; for (int b = 0; b < k; b++) r |= ((x >> b) & 1) << (k - 1 - b); return r; };
            // tile of lazy values (device form): worst-case entry bound 2.01 p, as after load_red
            std::vector<FrD> t(n);
            for (int i = 0; i < n; i++) {
                Fr c = in[dit ? brev(i) : i].from_mont();
                t[i] = fd_of<Fr29>(c);
                t[i].ub = 2.01;
            }
            auto tw = [&](size_t exp_) {   // omega^exp as a canonical device value
                Fr c = wpow[exp_ % n].from_mont();
                FrD w = fd_of<Fr29>(c).canonical();
                return w;
            };
            auto reduces = [&](int it) { return (it & 1) != 0 || it == k - 1; };
            int it = 0;
            if (k & 1) {
                const int qb = dit ? 0 : k - 1;
                for (int b = 0; b < n / 2; b++) {
                    int i0 = ((b >> qb) << (qb + 1)) | (b & ((1 << qb) - 1)), i1 = i0 | (1 << qb);
                    size_t texp = (size_t)(i0 & ((1 << qb) - 1)) << (k - 1 - qb);
                    if (dit) ntt_dit2(t[i0], t[i1], tw(texp));
                    else ntt_dif2(t[i0], t[i1], tw(texp), reduces(0));
                }
                it = 1;
            }
            for (; it < k; it += 2) {
                const int qlo = dit ? it : k - 2 - it;
                for (int gq = 0; gq < n / 4; gq++) {
                    int i00 = ((gq >> qlo) << (qlo + 2)) | (gq & ((1 << qlo) - 1));
                    int i01 = i00 | (1 << qlo), i10 = i00 | (2 << qlo), i11 = i01 | i10;
                    size_t low = (size_t)(i00 & ((1 << qlo) - 1));
                    size_t t1 = low << (k - 1 - qlo), t2 = low << (k - 2 - qlo), t3 = t2 + ((size_t)1 << (k - 2));
                    if (dit) ntt_dit4(t[i00], t[i01], t[i10], t[i11], tw(t1), tw(t2), tw(t3));
                    else ntt_dif4(t[i00], t[i01], t[i10], t[i11], tw(t1), tw(t2), tw(t3), reduces(it), reduces(it + 1));
                }
            }
            for (int i = 0; i < n; i++) {
                Fr got;
                t[i].reduce().store(&got);   // what the pass writes back (asserts the 2^256 packing bound)
                Fr want = ref[dit ? i : brev(i)];
                Fr got_m = canon_of<Fr29>(FrD::load_red(&got)).to_mont();
                CHECK(got_m == want);
            }
        }
    }
    // twiddle tables in limb form
    {
        Fr c = rand_canon<FrParams>();
        FrD w = fd_of<Fr29>(c).canonical();
        FrLimbs mem;
        store_limbs(&mem, w);
        CHECK(canon_of<Fr29>(load_limbs(&mem)) == c);
    }
    printf("NTT passes ok\n");
}

// The fused kernels of the evaluation-form H pipeline (ntt.hip: k_ntt_mid, k_ntt_tail3) through the very function
// they run, ntt_tile_stages, with bound tracking: k DIT stages on bit-reversed input -> coset scaling -> k DIF stages ->
// quotient (a b - c) z as a plain integer.  One "thread", whole-vector tiles of 2^1 .. 2^10 elements.
struct HostTile {
    std::vector<FrD>& v;
    FrD get(uint32_t li) const { return v[li]; }
    void put(uint32_t li, const FrD& x) const { v[li] = x; }
};
static void test_ntt_tile_fused() {
    for (int k = 1; k <= 10; k++) {
        const int n = 1 << k;
        Fr g = Fr::from_u64(7);
        uint32_t e[8];
        for (int i = 0; i < 8; i++) e[i] = FrParams::MOD[i];
        e[0] -= 1;
        uint32_t q[8];
        for (int i = 0; i < 8; i++) q[i] = (e[i] >> k) | (i < 7 ? (e[i + 1] << (32 - k)) : 0);
        const Fr omega = g.pow(q), omega_inv = omega.inv();
        std::vector<Fr> wf(n), wi(n);
        wf[0] = wi[0] = Fr::one();
        for (int i = 1; i < n; i++) { wf[i] = wf[i - 1] * omega; wi[i] = wi[i - 1] * omega_inv; }
        std::vector<FrLimbs> twf(n / 2 ? n / 2 : 1), twi(n / 2 ? n / 2 : 1);
        for (int i = 0; i < n / 2; i++) {
            store_limbs(&twf[i], fd_of<Fr29>(wf[i].from_mont()).canonical());
            store_limbs(&twi[i], fd_of<Fr29>(wi[i].from_mont()).canonical());
        }
        auto brev = [&](int x) { int r = 0; for (int b = 0; b < k; b++) r |= ((x >> b) & 1) << (k - 1 - b); return r; };
        const Fr minv = Fr::from_u64((uint64_t)n).inv();
        const Fr zinv = (g.pow_u64((uint64_t)n) - Fr::one()).inv();
        std::vector<Fr> ev[3], coset[3];
        std::vector<FrD> res[3];
        for (int vsel = 0; vsel < 3; vsel++) {
            ev[vsel].resize(n);
            for (int i = 0; i < n; i++) ev[vsel][i] = rand_canon<FrParams>().to_mont();
            // reference: coefficients c_i = (1/n) sum_j ev_j omega^(-i j); coset[j] = sum_i c_i g^i omega^(i j)
            std::vector<Fr> coef(n);
            for (int i = 0; i < n; i++) {
                Fr acc = Fr::zero();
                for (int j = 0; j < n; j++) acc = acc + ev[vsel][j] * wi[(size_t)i * j % n];
                coef[i] = acc * minv;
            }
            coset[vsel].resize(n);
            Fr gp = Fr::one();
            std::vector<Fr> cg(n);
            for (int i = 0; i < n; i++) { cg[i] = coef[i] * gp; gp = gp * g; }
            for (int j = 0; j < n; j++) {
                Fr acc = Fr::zero();
                for (int i = 0; i < n; i++) acc = acc + cg[i] * wf[(size_t)i * j % n];
                coset[vsel][j] = acc;
            }
            // the kernel's sequence on a tile in device form (entry bound 2.01 p, as after load_red)
            std::vector<FrD> t(n);
            for (int i = 0; i < n; i++) {
                t[i] = fd_of<Fr29>(ev[vsel][brev(i)].from_mont());
                t[i].ub = 2.01;
            }
            HostTile T{t};
            const TileGeom G = TileGeom::make(k, 0, k, 0, 1, 0);
            ntt_tile_stages<true>(T, G, twi.data(), 0, 1, [] {});
            gp = Fr::one();
            for (int i = 0; i < n; i++) {   // scaling by g^i / n: a product of two canonical table entries
                const FrD sc = fd_of<Fr29>((gp * minv).from_mont()).canonical();
                t[i] = t[i] * sc;
                gp = gp * g;
            }
            ntt_tile_stages<false>(T, G, twf.data(), 0, 1, [] {});
            res[vsel] = t;
            for (int i = 0; i < n; i++) {
                Fr mem;
                t[i].reduce().store(&mem);
                CHECK(canon_of<Fr29>(FrD::load_red(&mem)).to_mont() == coset[vsel][brev(i)]);
            }
        }
        // quotient: (a b - c) z with z a plain integer -> plain integer
        for (int i = 0; i < n; i++) {
            Fr ma, mb, mc;
            res[0][i].reduce().store(&ma);
            res[1][i].reduce().store(&mb);
            res[2][i].reduce().store(&mc);
            const FrD ab = FrD::load_red(&ma) * FrD::load_red(&mb);
            Fr zint = zinv.from_mont();   // canonical integer
            const FrD z = FrD::load_canon(&zint);
            Fr out;
            (FrD::sub<3>(ab, FrD::load_red(&mc)) * z).canonical().store(&out);
            const int j = brev(i);
            const Fr want = ((coset[0][j] * coset[1][j] - coset[2][j]) * zinv).from_mont();
            CHECK(out == want);
        }
    }
    printf("fused NTT tiles ok\n");
}

int main() {
    test_ntt_passes();
    test_ntt_tile_fused();
    test_field<Fq29>("Fq");
    test_field<Fr29>("Fr");
    G1Affine g1{Fq::from_u64(1), Fq::from_u64(2)};
    test_curve<FqL, Fq>("G1", g1);
    // G2 generator (prover/src/groth16/ethereum.rs:28-31)
    auto fq = [](uint32_t l0, uint32_t l1, uint32_t l2, uint32_t l3, uint32_t l4, uint32_t l5, uint32_t l6, uint32_t l7) {
        Fq c; c.l[0]=l0; c.l[1]=l1; c.l[2]=l2; c.l[3]=l3; c.l[4]=l4; c.l[5]=l5; c.l[6]=l6; c.l[7]=l7; return c.to_mont();
    };
    G2Affine g2;
    g2.x.c0 = fq(0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu);
    g2.x.c1 = fq(0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u);
    g2.y.c0 = fq(0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u);
    g2.y.c1 = fq(0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u);
    test_curve<Fq2L, Fq2>("G2", g2);
    if (failures) { printf("FAILED: %d\n", failures); return 1; }
    printf("ALL OK\n");
    return 0;
}
