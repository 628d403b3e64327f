// Device-side curve arithmetic over the lazy 9x29 field (fd.h): Fq2 tower and XYZZ points.
//
// Same formulas as ec.h (madd-2008-s, add-2008-s, dbl-2008-s-1); what differs is that every
// subtraction names the multiple of p it adds (sub<K>) and values are only reduced where the bound
// analysis needs it.  With MB = bound of a multiplication result (units of p):
//     point invariants   X <= 4 MB,  Y <= 2 MB,  ZZ, ZZZ <= MB        (MB = 2 for Fq, 3 for Fq2 components)
// The ZG_FD_DEBUG host build (tests/cpp/test_fd.cpp) carries bounds through every operation and
// asserts each K, so these constants are machine-checked.
#pragma once
#include "ec.h"
#include "fd.h"

namespace zg {

using FqD = Fd<Fq29>;
using FrD = Fd<Fr29>;

// ---- traits glue so XYZZd can be written once for Fq and Fq2
template <class Pm>
struct LazyTraits;

struct FqL {
    using T = FqD;
    using Mem = Fq;
    static constexpr int MB = 2;
    static constexpr bool QUAD = false;   // (FqQL below: one addition spread over four lanes)
    static ZG_HD T zero() { return T::zero(); }
    static ZG_HD T one() { return T::one(); }
    static ZG_HD T load_canon(const Mem* p) { return T::load_canon(p); }
    static ZG_HD T load_red(const Mem* p) { return T::load_red(p); }
    static ZG_HD void store_red(Mem* p, const T& v) { v.reduce().store(p); }
    static ZG_HD void store_canon(Mem* p, const T& v) { v.canonical().store(p); }
    static ZG_HD bool is_exact_zero(const T& a) { return a.is_exact_zero(); }
    static ZG_HD bool is_zero_mod_p(const T& a) { return a.is_zero_mod_p(); }
    static ZG_HD T add(const T& a, const T& b) { return a + b; }
    template <int K>
    static ZG_HD T sub(const T& a, const T& b) { return T::template sub<K>(a, b); }
    static ZG_HD T mul(const T& a, const T& b) { return a * b; }
    static ZG_HD T sqr(const T& a) { return a.sqr(); }
    static ZG_HD T inv(const T& a) { return a.inv(); }
    // hot-path variants (add_mixed): over Fq2 they skip the reduction of the result; here they are the plain operations
    template <int KB>
    static ZG_HD T mulx(const T& a, const T& b) { return a * b; }
    template <int K>
    static ZG_HD T sqrx(const T& a) { return a.sqr(); }
    // lazy-limb forms (fd.h): results may only feed a multiplication (or norm())
    static ZG_HD T add_lazy(const T& a, const T& b) { return T::add_lazy(a, b); }
    template <int K, int S>
    static ZG_HD T sub_lazy(const T& a, const T& b) { return T::template sub_lazy<K, S>(a, b); }
    static ZG_HD T norm(const T& a) { return a.norm(); }
    // -y for a canonical y (as 2p - y): only ever multiplied
    static ZG_HD T neg_canon_lazy(const T& y) { return y.template neg_lazy<2, 29>(); }   // 2p - y: the spread form of 1p has no room in its top limb
    template <int KMAX>
    static ZG_HD bool maybe_zero_mod_p(const T& a) { return a.template maybe_zero_mod_p<KMAX>(); }
    // r*t - y*w for normalised r, y, w (w < 2p) and t = qq - x3 formed inside: ONE reduction for both products.
    // qq, x3 normalised, x3 < KX p.
    template <int KX>
    static ZG_HD T mul_sub_of_diff(const T& r, const T& qq, const T& x3, const T& y, const T& w) {
        const T t = T::template sub_lazy<KX, 29>(qq, x3);
        const T nw = w.template neg_lazy<2, 29>();
        return T::mac2(r, t, y, nw);
    }
};

struct Fq2D {
    FqD c0, c1;
};

struct Fq2L {
    using T = Fq2D;
    using Mem = Fq2;
    static constexpr bool QUAD = false;
    static constexpr int MB = 3;  // components of a product are reduce()d: < 2.01 p
    static ZG_HD T zero() { return T{FqD::zero(), FqD::zero()}; }
    static ZG_HD T one() { return T{FqD::one(), FqD::zero()}; }
    static ZG_HD T load_canon(const Mem* p) { return T{FqD::load_canon(&p->c0), FqD::load_canon(&p->c1)}; }
    static ZG_HD T load_red(const Mem* p) { return T{FqD::load_red(&p->c0), FqD::load_red(&p->c1)}; }
    static ZG_HD void store_red(Mem* p, const T& v) {
        v.c0.reduce().store(&p->c0);
        v.c1.reduce().store(&p->c1);
    }
    static ZG_HD void store_canon(Mem* p, const T& v) {
        v.c0.canonical().store(&p->c0);
        v.c1.canonical().store(&p->c1);
    }
    static ZG_HD bool is_exact_zero(const T& a) { return a.c0.is_exact_zero() && a.c1.is_exact_zero(); }
    static ZG_HD bool is_zero_mod_p(const T& a) { return a.c0.is_zero_mod_p() && a.c1.is_zero_mod_p(); }
    static ZG_HD T add(const T& a, const T& b) { return T{a.c0 + b.c0, a.c1 + b.c1}; }
    template <int K>
    static ZG_HD T sub(const T& a, const T& b) {
        return T{FqD::sub<K>(a.c0, b.c0), FqD::sub<K>(a.c1, b.c1)};
    }
    // c0 = a0 b0 - a1 b1 and c1 = a0 b1 + a1 b0, each as ONE fused multiply-accumulate with a single Montgomery
    // reduction (FqD::mac2): 2 x 243 multiply-adds and no additions, against Karatsuba's 3 x 162 + five additions + two
    // reductions.  b must be normalised with components below KB p (its c1 is negated limb-wise); a may be a lazy
    // sum of two normalised values.  Results are normalised, below (|a0||b0| + |a1| KB) / 168.8 + 1.
    template <int KB>
    static ZG_HD T mulx(const T& a, const T& b) {
        const FqD nb1 = b.c1.template neg_lazy<KB, 29>();
        return T{FqD::mac2(a.c0, b.c0, a.c1, nb1), FqD::mac2(a.c0, b.c1, a.c1, b.c0)};
    }
    // general form: components up to 18 p in, reduce()d components out (< 2.01 p, so MB = 3 holds everywhere)
    static ZG_HD T mul(const T& a, const T& b) {
        const T r = mulx<18>(a, b);
        return T{r.c0.reduce(), r.c1.reduce()};
    }
    // (c0 + c1)(c0 - c1), c0 * 2 c1 with the sums formed limb-wise; components normalised and below K p
    template <int K>
    static ZG_HD T sqrx(const T& a) {
        const FqD s = FqD::add_lazy(a.c0, a.c1);
        const FqD d = FqD::sub<K>(a.c0, a.c1);
        return T{s * d, a.c0 * FqD::add_lazy(a.c1, a.c1)};
    }
    static ZG_HD T sqr(const T& a) {
        const T r = sqrx<16>(a);
        return T{r.c0.reduce(), r.c1.reduce()};
    }
    static ZG_HD T add_lazy(const T& a, const T& b) { return T{FqD::add_lazy(a.c0, b.c0), FqD::add_lazy(a.c1, b.c1)}; }
    template <int K, int S>
    static ZG_HD T sub_lazy(const T& a, const T& b) {
        return T{FqD::sub_lazy<K, S>(a.c0, b.c0), FqD::sub_lazy<K, S>(a.c1, b.c1)};
    }
    static ZG_HD T norm(const T& a) { return T{a.c0.norm(), a.c1.norm()}; }
    static ZG_HD T neg_canon_lazy(const T& y) { return T{y.c0.template neg_lazy<2, 29>(), y.c1.template neg_lazy<2, 29>()}; }
    template <int KMAX>
    static ZG_HD bool maybe_zero_mod_p(const T& a) {
        return a.c0.template maybe_zero_mod_p<KMAX>() && a.c1.template maybe_zero_mod_p<KMAX>();
    }
    // r*(qq - x3) - y*w: over Fq2 the difference is normalised (a lazy 3-limb-unit operand would overflow a mac2 column)
    template <int KX>
    static ZG_HD T mul_sub_of_diff(const T& r, const T& qq, const T& x3, const T& y, const T& w) {
        const T t = sub<KX>(qq, x3);
        return sub<MB>(mulx<MB + 2 * MB>(t, r), mulx<MB>(y, w));
    }
    static ZG_HD T inv(const T& a) {
        FqD n = (a.c0.sqr() + a.c1.sqr()).inv();
        FqD c1 = (a.c1 * n);
        return T{a.c0 * n, c1.template neg<2>()};
    }
};

#if defined(__HIPCC__)
// ---- Fq2 on LANE PAIRS: lane 2i holds the c0 component of every Fq2 value of a point, lane 2i + 1 the c1 component.
// What it is for: the latency-bound G2 chains (bucket reduction levels, heavy-bucket folds, the items of a small job):
// a G2 addition is 19 us of ONE lane's instruction stream whatever else the GPU does, and its Fq2 products are two
// independent fused multiply-accumulates (c0 = a0 b0 - a1 b1, c1 = a0 b1 + a1 b0).  With the components on adjacent
// lanes each lane runs ONE mac2 per product after an exchange of the operands (18 v_mov_dpp, quad_perm [1,0,3,2]):
// half the instruction stream per lane, half the registers (36 instead of 72 per point), same bounds -- every
// function below is Fq2L's, component by component, with T = one FqD per lane.
// Control flow must be uniform across a pair: every predicate (identity, P = +-Q) is combined over both lanes.
struct Fq2PL {
    using T = FqD;
    using Mem = Fq2;
    static constexpr int MB = 3;
    static constexpr bool QUAD = false;
    static __device__ __forceinline__ bool hi() { return (threadIdx.x & 1u) != 0; }
    static __device__ __forceinline__ uint32_t swap_u32(uint32_t v) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    }
    static __device__ __forceinline__ T swap(const T& a) {   // the partner lane's value
        T r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = swap_u32(a.l[i]);
        return r;
    }
    static __device__ __forceinline__ T sel(bool c, const T& a, const T& b) {   // c ? a : b, limb-wise (v_cndmask)
        T r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : b.l[i];
        return r;
    }
    // a predicate over the whole Fq2 value; the exchange runs on both lanes unconditionally (no short circuit: a lane
    // that skipped it would hand its partner a stale register)
    static __device__ __forceinline__ bool both(bool own) {
        const uint32_t o = own ? 1u : 0u;
        return (o & swap_u32(o)) != 0u;
    }

    static __device__ __forceinline__ T zero() { return T::zero(); }
    static __device__ __forceinline__ T one() { return hi() ? T::zero() : T::one(); }
    static __device__ __forceinline__ T load_canon(const Mem* p) { return T::load_canon(hi() ? &p->c1 : &p->c0); }
    static __device__ __forceinline__ T load_red(const Mem* p) { return T::load_red(hi() ? &p->c1 : &p->c0); }
    static __device__ __forceinline__ void store_red(Mem* p, const T& v) { v.reduce().store(hi() ? &p->c1 : &p->c0); }
    static __device__ __forceinline__ void store_canon(Mem* p, const T& v) { v.canonical().store(hi() ? &p->c1 : &p->c0); }
    static __device__ __forceinline__ bool is_exact_zero(const T& a) { return both(a.is_exact_zero()); }
    static __device__ __forceinline__ bool is_zero_mod_p(const T& a) { return both(a.is_zero_mod_p()); }
    static __device__ __forceinline__ T add(const T& a, const T& b) { return a + b; }
    template <int K>
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return T::template sub<K>(a, b); }
    // Fq2L::mulx: lane 0 computes a0 b0 + a1 (-b1), lane 1 computes a1 b0 + a0 b1 -- as ONE instruction stream:
    // own a * (own b | partner b)  +  partner a * (-partner b | own b)
    template <int KB>
    static __device__ __forceinline__ T mulx(const T& a, const T& b) {
        const T pa = swap(a), pb = swap(b);
        const T npb = pb.template neg_lazy<KB, 29>();
        const bool h = hi();
        return T::mac2(a, sel(h, pb, b), pa, sel(h, b, npb));
    }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return mulx<18>(a, b).reduce(); }
    // Fq2L::sqrx: (c0 + c1)(c0 - c1) on lane 0, c0 * 2 c1 on lane 1
    template <int K>
    static __device__ __forceinline__ T sqrx(const T& a) {
        const T pa = swap(a);
        const bool h = hi();
        const T s = T::add_lazy(a, pa), d = T::template sub<K>(a, pa), two = T::add_lazy(a, a);
        return sel(h, pa, s) * sel(h, two, d);
    }
    static __device__ __forceinline__ T sqr(const T& a) { return sqrx<16>(a).reduce(); }
    static __device__ __forceinline__ T add_lazy(const T& a, const T& b) { return T::add_lazy(a, b); }
    template <int K, int S>
    static __device__ __forceinline__ T sub_lazy(const T& a, const T& b) { return T::template sub_lazy<K, S>(a, b); }
    static __device__ __forceinline__ T norm(const T& a) { return a.norm(); }
    static __device__ __forceinline__ T neg_canon_lazy(const T& y) { return y.template neg_lazy<2, 29>(); }
    template <int KMAX>
    static __device__ __forceinline__ bool maybe_zero_mod_p(const T& a) { return both(a.template maybe_zero_mod_p<KMAX>()); }
    template <int KX>
    static __device__ __forceinline__ T mul_sub_of_diff(const T& r, const T& qq, const T& x3, const T& y, const T& w) {
        const T t = sub<KX>(qq, x3);
        return sub<MB>(mulx<MB + 2 * MB>(t, r), mulx<MB>(y, w));
    }
    static __device__ __forceinline__ T inv(const T& a) {   // (a0 - a1 u) / (a0^2 + a1^2)
        const T sq = a.sqr();
        const T n = (sq + swap(sq)).inv();
        const T r = a * n;
        return hi() ? r.template neg<2>() : r;
    }
};

// ---- one G1 addition on FOUR lanes.  The latency-bound kernels (the small levels of the bucket reduction, heavy-bucket
// folds) run chains of dependent additions, and one addition is fourteen field multiplications issued one after the other
// by a single wave (~7 us) although at most four of them depend on each other.  Here the four lanes of a quad hold COPIES
// of both operands and each computes one product per round -- u1, u2, s1, s2 | pp, r^2, zz zz', zzz zzz' | ppp, qq, zz3 |
// r (qq - x3), s1 ppp, zzz3 -- with the results broadcast inside the quad by DPP (quad_perm [k,k,k,k]): four rounds
// instead of fourteen products, the same operations with the same bounds as XYZZd::add.  Costs 4 lanes' worth of issue
// slots for ~2.2x the speed: only where lanes are idle anyway (MsmJob picks it per launch).  Memory: loads are
// replicated (same address on the four lanes), stores are the first lane's.
struct FqQL : FqL {
    static constexpr bool QUAD = true;
    static __device__ __forceinline__ uint32_t role() { return threadIdx.x & 3u; }
    static __device__ __forceinline__ void store_red(Mem* p, const T& v) {
        if (role() == 0) v.reduce().store(p);
    }
    static __device__ __forceinline__ T sel4(const T& a0, const T& a1, const T& a2, const T& a3) {
        const uint32_t k = role();
        T r;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const uint32_t lo = k & 1u ? a1.l[i] : a0.l[i], hi = k & 1u ? a3.l[i] : a2.l[i];
            r.l[i] = k & 2u ? hi : lo;
        }
        return r;
    }
    template <int K>
    static __device__ __forceinline__ T bcast(const T& a) {   // lane K's value on all four lanes
        T r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], K * 0x55, 0xF, 0xF, true);
        return r;
    }
};
#endif

template <class L>
struct AffD {
    typename L::T x, y;
    bool inf;
    static ZG_HD AffD load(const Affine<typename L::Mem>* p) {
        AffD r;
        r.x = L::load_canon(&p->x);
        r.y = L::load_canon(&p->y);
        r.inf = L::is_exact_zero(r.x) && L::is_exact_zero(r.y);
        return r;
    }
    ZG_HD AffD negated() const {  // y in [0, p]  ->  p - y
        AffD r = *this;
        if (!inf) r.y = L::template sub<1>(L::zero(), y);
        return r;
    }
    // normalised coordinates again (the rare doubling branch of add_mixed needs sums of y)
    ZG_HD AffD y_canonical() const {
        AffD r = *this;
        r.y = L::norm(y);
        return r;
    }
    // the same with limb-wise subtraction only (9 instructions): y may then only be multiplied -- which is all
    // add_mixed does with it
    ZG_HD AffD negated_lazy() const {
        AffD r = *this;
        if (!inf) r.y = L::neg_canon_lazy(y);
        return r;
    }
};

template <class L>
struct XYZZd {
    using T = typename L::T;
    static constexpr int MB = L::MB, BX = 4 * L::MB, BY = 2 * L::MB;
    T x, y, zz, zzz;

    static ZG_HD XYZZd inf() { return XYZZd{L::zero(), L::zero(), L::zero(), L::zero()}; }
    ZG_HD bool is_inf() const { return L::is_exact_zero(zz); }
    static ZG_HD XYZZd from_affine(const AffD<L>& p) {
        if (p.inf) return inf();
        return XYZZd{p.x, p.y, L::one(), L::one()};
    }
    // memory form: four reduce()d residues (< 2.01 p each)
    static ZG_HD XYZZd load(const XYZZ<typename L::Mem>* p) {
        return XYZZd{L::load_red(&p->x), L::load_red(&p->y), L::load_red(&p->zz), L::load_red(&p->zzz)};
    }
    ZG_HD void store(XYZZ<typename L::Mem>* p) const {
        L::store_red(&p->x, x);
        L::store_red(&p->y, y);
        L::store_red(&p->zz, zz);
        L::store_red(&p->zzz, zzz);
    }
    ZG_HD XYZZd neg() const { return XYZZd{x, L::template sub<BY>(L::zero(), y), zz, zzz}; }

    ZG_HD XYZZd dbl() const {
        if (is_inf()) return *this;
        T u = L::add(y, y);                      // <= 2 BY = 4 MB
        T v = L::sqr(u);
        T w = L::mul(u, v);
        T s = L::mul(x, v);
        T xx = L::sqr(x);
        T m = L::add(L::add(xx, xx), xx);        // <= 3 MB
        T x3 = L::template sub<2 * MB>(L::sqr(m), L::add(s, s));   // <= 3 MB
        T y3 = L::template sub<MB>(L::mul(m, L::template sub<3 * MB>(s, x3)), L::mul(w, y));
        return XYZZd{x3, y3, L::mul(v, zz), L::mul(w, zzz)};
    }
    static ZG_HD XYZZd dbl_affine(const AffD<L>& p) {
        if (p.inf) return inf();
        T u = L::add(p.y, p.y);
        T v = L::sqr(u);
        T w = L::mul(u, v);
        T s = L::mul(p.x, v);
        T xx = L::sqr(p.x);
        T m = L::add(L::add(xx, xx), xx);
        T x3 = L::template sub<2 * MB>(L::sqr(m), L::add(s, s));
        T y3 = L::template sub<MB>(L::mul(m, L::template sub<3 * MB>(s, x3)), L::mul(w, p.y));
        return XYZZd{x3, y3, v, w};
    }
    // this + affine q  (madd-2008-s: 8M + 2S).  q.y may be a lazily negated canonical value.  Sums and differences that
    // only feed a multiplication skip carry propagation (fd.h "lazy limbs"), the y3 = r (qq - x3) - y ppp pair of
    // products shares one Montgomery reduction over Fq, and the doubling / inverse test starts with a one-multiply
    // filter.  All four outputs are normalised: X <= BX, Y <= BY, ZZ, ZZZ <= MB.
    ZG_HD XYZZd add_mixed(const AffD<L>& q) const {
#if defined(__HIPCC__)
        if constexpr (L::QUAD) return add_mixed_quad(q);
#endif
        if (q.inf) return *this;
        if (is_inf()) return from_affine(q.y_canonical());
        T u2 = L::template mulx<MB>(q.x, zz);
        T s2 = L::template mulx<MB>(q.y, zzz);
        T p = L::template sub<BX>(u2, x);        // <= MB + BX
        T r = L::template sub<BY>(s2, y);        // <= MB + BY
        // The candidate flag is made opaque to the optimiser: with a visible `filter && full test` condition clang
        // threads the two tests into the main path and emits ~120 extra multiply-adds and ~240 moves per addition.
        uint32_t cand = L::template maybe_zero_mod_p<BX + MB + 1>(p) ? 1u : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(cand));
#endif
        if (cand) {
            if (L::is_zero_mod_p(p)) {
                if (L::is_zero_mod_p(r)) return dbl_affine(q.y_canonical());
                return inf();
            }
        }
        T pp = L::template sqrx<BX + MB>(p);     // Fq2: components up to ~7 p; only ever a factor
        T ppp = L::template mulx<2 * MB + 2>(p, pp);
        T qq = L::template mulx<2 * MB + 2>(x, pp);
        // x3 = r^2 - (ppp + 2 qq): the subtrahend is a limb-wise sum of three normalised values (limbs < 3 * 2^29)
        T x3 = L::norm(L::template sub_lazy<3 * MB, 31>(L::template sqrx<BY + MB>(r), L::add_lazy(ppp, L::add_lazy(qq, qq))));   // <= 4 MB = BX
        T y3 = L::template mul_sub_of_diff<BX>(r, qq, x3, y, ppp);                                            // <= 2 MB = BY
        return XYZZd{x3, y3, L::template mulx<2 * MB + 2>(zz, pp), L::template mulx<MB>(zzz, ppp)};
    }
#if defined(__HIPCC__)
    // add() with the products of each dependency level on the four lanes of a quad (FqQL)
    __device__ __forceinline__ XYZZd add_quad(const XYZZd& o) const {
        if (o.is_inf()) return *this;
        if (is_inf()) return o;
        const T m1 = L::mul(L::sel4(x, o.x, y, o.y), L::sel4(o.zz, zz, o.zzz, zzz));
        const T u1 = L::template bcast<0>(m1), u2 = L::template bcast<1>(m1), s1 = L::template bcast<2>(m1), s2 = L::template bcast<3>(m1);
        const T p = L::template sub<MB>(u2, u1);
        const T r = L::template sub<MB>(s2, s1);
        if (L::is_zero_mod_p(p)) {
            if (L::is_zero_mod_p(r)) return dbl();
            return inf();
        }
        const T m2 = L::mul(L::sel4(p, r, zz, zzz), L::sel4(p, r, o.zz, o.zzz));
        const T pp = L::template bcast<0>(m2), rr = L::template bcast<1>(m2), zzo = L::template bcast<2>(m2), zzzo = L::template bcast<3>(m2);
        const T m3 = L::mul(L::sel4(p, u1, zzo, zzo), pp);
        const T ppp = L::template bcast<0>(m3), qq = L::template bcast<1>(m3), zz3 = L::template bcast<2>(m3);
        const T x3 = L::template sub<3 * MB>(rr, L::add(ppp, L::add(qq, qq)));
        const T m4 = L::mul(L::sel4(r, s1, zzzo, zzzo), L::sel4(L::template sub<BX>(qq, x3), ppp, ppp, ppp));
        const T t0 = L::template bcast<0>(m4), t1 = L::template bcast<1>(m4), zzz3 = L::template bcast<2>(m4);
        return XYZZd{x3, L::template sub<MB>(t0, t1), zz3, zzz3};
    }
    // add_mixed() the same way: u2, s2 | pp, r^2 | ppp, qq, zz3 | r (qq - x3), y ppp, zzz3 -- four rounds instead of ten
    // serial products.  The operations are add_mixed()'s with add_quad()'s plain (reduced) products: p <= MB + BX and
    // r <= MB + BY feed multiplications only, x3 <= MB + 3 MB = BX, y3 <= 2 MB = BY.  q.y may be a lazily negated
    // canonical value (2p - y limb-wise): it is only ever a factor (round one) -- the identity and doubling branches
    // normalise it first, as add_mixed() does.
    __device__ __forceinline__ XYZZd add_mixed_quad(const AffD<L>& q) const {
        if (q.inf) return *this;
        if (is_inf()) return from_affine(q.y_canonical());
        const T m1 = L::mul(L::sel4(q.x, q.y, q.x, q.y), L::sel4(zz, zzz, zz, zzz));
        const T u2 = L::template bcast<0>(m1), s2 = L::template bcast<1>(m1);
        const T p = L::template sub<BX>(u2, x);
        const T r = L::template sub<BY>(s2, y);
        if (L::is_zero_mod_p(p)) {
            if (L::is_zero_mod_p(r)) return dbl_affine(q.y_canonical());
            return inf();
        }
        const T m2 = L::sqr(L::sel4(p, r, p, r));
        const T pp = L::template bcast<0>(m2), rr = L::template bcast<1>(m2);
        const T m3 = L::mul(L::sel4(p, x, zz, zz), pp);
        const T ppp = L::template bcast<0>(m3), qq = L::template bcast<1>(m3), zz3 = L::template bcast<2>(m3);
        const T x3 = L::template sub<3 * MB>(rr, L::add(ppp, L::add(qq, qq)));
        const T m4 = L::mul(L::sel4(r, y, zzz, zzz), L::sel4(L::template sub<BX>(qq, x3), ppp, ppp, ppp));
        const T t0 = L::template bcast<0>(m4), t1 = L::template bcast<1>(m4), zzz3 = L::template bcast<2>(m4);
        return XYZZd{x3, L::template sub<MB>(t0, t1), zz3, zzz3};
    }
#endif
    ZG_HD XYZZd add(const XYZZd& o) const {
#if defined(__HIPCC__)
        if constexpr (L::QUAD) return add_quad(o);
#endif
        if (o.is_inf()) return *this;
        if (is_inf()) return o;
        T u1 = L::mul(x, o.zz);
        T u2 = L::mul(o.x, zz);
        T s1 = L::mul(y, o.zzz);
        T s2 = L::mul(o.y, zzz);
        T p = L::template sub<MB>(u2, u1);
        T r = L::template sub<MB>(s2, s1);
        if (L::is_zero_mod_p(p)) {
            if (L::is_zero_mod_p(r)) return dbl();
            return inf();
        }
        T pp = L::sqr(p);
        T ppp = L::mul(p, pp);
        T qq = L::mul(u1, pp);
        T x3 = L::template sub<3 * MB>(L::sqr(r), L::add(ppp, L::add(qq, qq)));
        T y3 = L::template sub<MB>(L::mul(r, L::template sub<BX>(qq, x3)), L::mul(s1, ppp));
        return XYZZd{x3, y3, L::mul(L::mul(zz, o.zz), pp), L::mul(L::mul(zzz, o.zzz), ppp)};
    }
    // affine coordinates as canonical residues (x/zz, y/zzz); identity -> (0, 0)
    ZG_HD void to_affine_mem(Affine<typename L::Mem>* out) const {
        if (is_inf()) {
            L::store_canon(&out->x, L::zero());
            L::store_canon(&out->y, L::zero());
            return;
        }
        T t = L::inv(L::mul(zz, zzz));
        T izz = L::mul(t, zzz);
        T izzz = L::mul(t, zz);
        L::store_canon(&out->x, L::mul(x, izz));
        L::store_canon(&out->y, L::mul(y, izzz));
    }
};

using G1L = XYZZd<FqL>;
using G2L = XYZZd<Fq2L>;

}  // namespace zg
