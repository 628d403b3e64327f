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
};

struct Fq2D {
    FqD c0, c1;
};

struct Fq2L {
    using T = Fq2D;
    using Mem = Fq2;
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
    // Karatsuba; component bounds up to 18 p are safe (aa, bb <= 18*18/168.8 + 1 < 3)
    static ZG_HD T mul(const T& a, const T& b) {
        FqD aa = a.c0 * b.c0;
        FqD bb = a.c1 * b.c1;
        FqD o = (a.c0 + a.c1) * (b.c0 + b.c1);
        return T{FqD::sub<3>(aa, bb).reduce(), FqD::sub<6>(o, aa + bb).reduce()};
    }
    // (c0 + c1)(c0 - c1), 2 c0 c1; needs c1 <= 16 p
    static ZG_HD T sqr(const T& a) {
        FqD s = a.c0 + a.c1;
        FqD d = FqD::sub<16>(a.c0, a.c1);
        FqD m = a.c0 * a.c1;
        return T{(s * d).reduce(), m.dbl().reduce()};
    }
    static ZG_HD T inv(const T& a) {
        FqD n = (a.c0.sqr() + a.c1.sqr()).inv();
        FqD c1 = (a.c1 * n);
        return T{a.c0 * n, c1.template neg<2>()};
    }
};

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
    // this + affine q
    ZG_HD XYZZd add_mixed(const AffD<L>& q) const {
        if (q.inf) return *this;
        if (is_inf()) return from_affine(q);
        T u2 = L::mul(q.x, zz);
        T s2 = L::mul(q.y, zzz);
        T p = L::template sub<BX>(u2, x);        // <= MB + BX
        T r = L::template sub<BY>(s2, y);        // <= MB + BY
        if (L::is_zero_mod_p(p)) {
            if (L::is_zero_mod_p(r)) return dbl_affine(q);
            return inf();
        }
        T pp = L::sqr(p);
        T ppp = L::mul(p, pp);
        T qq = L::mul(x, pp);
        T x3 = L::template sub<3 * MB>(L::sqr(r), L::add(ppp, L::add(qq, qq)));       // <= 4 MB = BX
        T y3 = L::template sub<MB>(L::mul(r, L::template sub<BX>(qq, x3)), L::mul(y, ppp));  // <= 2 MB = BY
        return XYZZd{x3, y3, L::mul(zz, pp), L::mul(zzz, ppp)};
    }
    ZG_HD XYZZd add(const XYZZd& o) const {
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
