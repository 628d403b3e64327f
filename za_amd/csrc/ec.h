// Short-Weierstrass a=0 curve arithmetic (BN254 G1 over Fq, G2 over Fq2) for host and device.
//
// Replaces pairing_ce::bn256's CurveAffine/CurveProjective arithmetic that bellman's multiexp and
// create_proof call (reference call site: prover/src/groth16/prover.rs:173).  Group law is exact, so
// any coordinate system gives the same canonical affine result; buckets use extended Jacobian
// "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) because the mixed add is 8M+2S and needs
// no field inversion.
#pragma once
#include "fp.h"

namespace zg {

template <class F>
struct Affine {  // identity encoded as (0, 0), which is not on y^2 = x^3 + b for b != 0
    F x, y;
    ZG_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    static ZG_HD Affine inf() { return Affine{F::zero(), F::zero()}; }
    ZG_HD Affine neg() const { return is_inf() ? *this : Affine{x, y.neg()}; }
};

template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    static ZG_HD XYZZ inf() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
    ZG_HD bool is_inf() const { return zz.is_zero(); }
    static ZG_HD XYZZ from_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        return XYZZ{p.x, p.y, F::one(), F::one()};
    }
    ZG_HD XYZZ neg() const { return XYZZ{x, y.neg(), zz, zzz}; }

    // dbl-2008-s-1
    ZG_HD XYZZ dbl() const {
        if (is_inf()) return *this;
        F u = y.dbl();
        F v = u.sqr();
        F w = u * v;
        F s = x * v;
        F xx = x.sqr();
        F m = xx.dbl() + xx;
        F x3 = m.sqr() - s.dbl();
        F y3 = m * (s - x3) - w * y;
        return XYZZ{x3, y3, v * zz, w * zzz};
    }
    // mdbl-2008-s-1 : 2*P for affine P
    static ZG_HD XYZZ dbl_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        F u = p.y.dbl();
        F v = u.sqr();
        F w = u * v;
        F s = p.x * v;
        F xx = p.x.sqr();
        F m = xx.dbl() + xx;
        F x3 = m.sqr() - s.dbl();
        F y3 = m * (s - x3) - w * p.y;
        return XYZZ{x3, y3, v, w};
    }
    // madd-2008-s : this + affine q   (8M + 2S)
    ZG_HD XYZZ add_mixed(const Affine<F>& q) const {
        if (q.is_inf()) return *this;
        if (is_inf()) return from_affine(q);
        F u2 = q.x * zz;
        F s2 = q.y * zzz;
        F p = u2 - x;
        F r = s2 - y;
        if (p.is_zero()) {
            if (r.is_zero()) return dbl_affine(q);
            return inf();
        }
        F pp = p.sqr();
        F ppp = p * pp;
        F qq = x * pp;
        F x3 = r.sqr() - ppp - qq.dbl();
        F y3 = r * (qq - x3) - y * ppp;
        return XYZZ{x3, y3, zz * pp, zzz * ppp};
    }
    // add-2008-s : this + o   (12M + 2S)
    ZG_HD XYZZ add(const XYZZ& o) const {
        if (o.is_inf()) return *this;
        if (is_inf()) return o;
        F u1 = x * o.zz;
        F u2 = o.x * zz;
        F s1 = y * o.zzz;
        F s2 = o.y * zzz;
        F p = u2 - u1;
        F r = s2 - s1;
        if (p.is_zero()) {
            if (r.is_zero()) return dbl();
            return inf();
        }
        F pp = p.sqr();
        F ppp = p * pp;
        F qq = u1 * pp;
        F x3 = r.sqr() - ppp - qq.dbl();
        F y3 = r * (qq - x3) - s1 * ppp;
        return XYZZ{x3, y3, zz * o.zz * pp, zzz * o.zzz * ppp};
    }
    // k * this for a small non-negative integer k (left-to-right double-and-add)
    ZG_HD XYZZ mul_small(uint32_t k) const {
        if (k == 0) return inf();
        XYZZ acc = *this;
        int top = 31;
        while (!((k >> top) & 1)) top--;
        for (int i = top - 1; i >= 0; i--) {
            acc = acc.dbl();
            if ((k >> i) & 1) acc = acc.add(*this);
        }
        return acc;
    }
    // k * this for a canonical 256-bit scalar (8 x u32 LE)
    ZG_HD XYZZ mul(const uint32_t k[8]) const {
        XYZZ acc = inf();
        for (int i = 255; i >= 0; i--) {
            acc = acc.dbl();
            if ((k[i >> 5] >> (i & 31)) & 1) acc = acc.add(*this);
        }
        return acc;
    }
    ZG_HD Affine<F> to_affine() const {
        if (is_inf()) return Affine<F>::inf();
        // 1/ZZ = ZZZ^2/ZZ^4 ... simplest: invert zz and zzz through one inversion of zz*zzz
        F t = (zz * zzz).inv();
        F izz = t * zzz;
        F izzz = t * zz;
        return Affine<F>{x * izz, y * izzz};
    }
};

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;

// y^2 == x^3 + b
template <class F>
ZG_HD bool on_curve(const Affine<F>& p, const F& b) {
    if (p.is_inf()) return true;
    return p.y.sqr() == p.x.sqr() * p.x + b;
}

}  // namespace zg
