// BN254 G1 (y^2 = x^3 + 3 over Fq) point arithmetic for the MSM kernels.
//
// Bucket accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2;
// identity: ZZ = 0): a mixed addition of an affine base costs 8M + 2S and needs no inversion.  All
// exceptional cases (identity operands, P + P, P + (-P)) are handled: bases of the form [i]G make
// "accumulator equals the incoming point" a normal event, not a corner case.
#pragma once
#include "field.hip.h"

namespace zkhip {

struct g1_affine {
    fe x, y;   // identity = (0, 0), as halo2curves' G1Affine
};
struct g1_xyzz {
    fe x, y, zz, zzz;
};

ZK_DEV bool affine_is_identity(const g1_affine& p) { return Fq::is_zero(p.x) && Fq::is_zero(p.y); }
ZK_DEV bool xyzz_is_identity(const g1_xyzz& p) { return Fq::is_zero(p.zz); }
ZK_DEV g1_xyzz xyzz_identity() {
    g1_xyzz o;
    o.x = Fq::zero();
    o.y = Fq::zero();
    o.zz = Fq::zero();
    o.zzz = Fq::zero();
    return o;
}
ZK_DEV g1_xyzz xyzz_from_affine(const g1_affine& p) {
    g1_xyzz o;
    if (affine_is_identity(p)) return xyzz_identity();
    o.x = p.x;
    o.y = p.y;
    o.zz = Fq::one();
    o.zzz = Fq::one();
    return o;
}

// 2 * (affine p), p != identity
ZK_DEV g1_xyzz xyzz_double_affine(const g1_affine& p) {
    g1_xyzz o;
    const fe u = Fq::dbl(p.y);
    const fe v = Fq::sqr(u);
    const fe w = Fq::mul(u, v);
    const fe s = Fq::mul(p.x, v);
    const fe x2 = Fq::sqr(p.x);
    const fe m = Fq::add(Fq::dbl(x2), x2);
    o.x = Fq::sub(Fq::sqr(m), Fq::dbl(s));
    o.y = Fq::sub(Fq::mul(m, Fq::sub(s, o.x)), Fq::mul(w, p.y));
    o.zz = v;
    o.zzz = w;
    return o;
}

ZK_DEV g1_xyzz xyzz_double(const g1_xyzz& p) {
    if (xyzz_is_identity(p)) return p;
    g1_xyzz o;
    const fe u = Fq::dbl(p.y);
    const fe v = Fq::sqr(u);
    const fe w = Fq::mul(u, v);
    const fe s = Fq::mul(p.x, v);
    const fe x2 = Fq::sqr(p.x);
    const fe m = Fq::add(Fq::dbl(x2), x2);
    o.x = Fq::sub(Fq::sqr(m), Fq::dbl(s));
    o.y = Fq::sub(Fq::mul(m, Fq::sub(s, o.x)), Fq::mul(w, p.y));
    o.zz = Fq::mul(v, p.zz);
    o.zzz = Fq::mul(w, p.zzz);
    return o;
}

// acc += q (affine)
ZK_DEV void xyzz_add_affine(g1_xyzz& acc, const g1_affine& q) {
    if (affine_is_identity(q)) return;
    if (xyzz_is_identity(acc)) {
        acc.x = q.x;
        acc.y = q.y;
        acc.zz = Fq::one();
        acc.zzz = Fq::one();
        return;
    }
    const fe u2 = Fq::mul(q.x, acc.zz);
    const fe s2 = Fq::mul(q.y, acc.zzz);
    const fe p = Fq::sub(u2, acc.x);
    const fe r = Fq::sub(s2, acc.y);
    if (Fq::is_zero(p)) {
        if (Fq::is_zero(r)) acc = xyzz_double_affine(q);
        else acc = xyzz_identity();
        return;
    }
    const fe pp = Fq::sqr(p);
    const fe ppp = Fq::mul(p, pp);
    const fe qq = Fq::mul(acc.x, pp);
    const fe x3 = Fq::sub(Fq::sub(Fq::sqr(r), ppp), Fq::dbl(qq));
    acc.y = Fq::sub(Fq::mul(r, Fq::sub(qq, x3)), Fq::mul(acc.y, ppp));
    acc.x = x3;
    acc.zz = Fq::mul(acc.zz, pp);
    acc.zzz = Fq::mul(acc.zzz, ppp);
}

// a + b
ZK_DEV g1_xyzz xyzz_add(const g1_xyzz& a, const g1_xyzz& b) {
    if (xyzz_is_identity(a)) return b;
    if (xyzz_is_identity(b)) return a;
    const fe u1 = Fq::mul(a.x, b.zz);
    const fe u2 = Fq::mul(b.x, a.zz);
    const fe s1 = Fq::mul(a.y, b.zzz);
    const fe s2 = Fq::mul(b.y, a.zzz);
    const fe p = Fq::sub(u2, u1);
    const fe r = Fq::sub(s2, s1);
    if (Fq::is_zero(p)) {
        if (Fq::is_zero(r)) return xyzz_double(a);
        return xyzz_identity();
    }
    g1_xyzz o;
    const fe pp = Fq::sqr(p);
    const fe ppp = Fq::mul(p, pp);
    const fe qq = Fq::mul(u1, pp);
    o.x = Fq::sub(Fq::sub(Fq::sqr(r), ppp), Fq::dbl(qq));
    o.y = Fq::sub(Fq::mul(r, Fq::sub(qq, o.x)), Fq::mul(s1, ppp));
    o.zz = Fq::mul(Fq::mul(a.zz, b.zz), pp);
    o.zzz = Fq::mul(Fq::mul(a.zzz, b.zzz), ppp);
    return o;
}

// a^(q-2)
ZK_DEV fe fq_invert(const fe& a) {
    fe acc = Fq::one();
    for (int bit = 255; bit >= 0; bit--) {
        acc = Fq::sqr(acc);
        uint32_t word = FqParams::P[bit >> 5];
        if ((bit >> 5) == 0) word -= 2;   // exponent q - 2; the low word of q is > 2, so no borrow
        if ((word >> (bit & 31)) & 1) acc = Fq::mul(acc, a);
    }
    return acc;
}

ZK_DEV g1_affine xyzz_to_affine(const g1_xyzz& p) {
    g1_affine o;
    if (xyzz_is_identity(p)) {
        o.x = Fq::zero();
        o.y = Fq::zero();
        return o;
    }
    // 1/zzz, then 1/zz = (1/zzz)^2 * zz^2  (zz^3 = zzz^2)
    const fe izzz = fq_invert(p.zzz);
    const fe izz = Fq::mul(Fq::sqr(izzz), Fq::sqr(p.zz));
    o.x = Fq::mul(p.x, izz);
    o.y = Fq::mul(p.y, izzz);
    return o;
}

ZK_DEV g1_affine load_affine(const g1_affine* p) {
    g1_affine o;
    o.x = load_fe(&p->x);
    o.y = load_fe(&p->y);
    return o;
}
ZK_DEV void store_affine(g1_affine* p, const g1_affine& v) {
    store_fe(&p->x, v.x);
    store_fe(&p->y, v.y);
}
ZK_DEV g1_xyzz load_xyzz(const g1_xyzz* p) {
    g1_xyzz o;
    o.x = load_fe(&p->x);
    o.y = load_fe(&p->y);
    o.zz = load_fe(&p->zz);
    o.zzz = load_fe(&p->zzz);
    return o;
}
ZK_DEV void store_xyzz(g1_xyzz* p, const g1_xyzz& v) {
    store_fe(&p->x, v.x);
    store_fe(&p->y, v.y);
    store_fe(&p->zz, v.zz);
    store_fe(&p->zzz, v.zzz);
}

}  // namespace zkhip
