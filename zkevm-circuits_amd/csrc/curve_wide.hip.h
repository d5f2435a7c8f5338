// XYZZ bucket accumulation on the lazy signed-digit arithmetic of wide.hip.h (the MSM accumulate kernel's inner loop).
//
// Coordinates live in the 2^261 Montgomery domain (the MSM tables are stored in it; see msm.hip).  A mixed addition
// is 8 mul + 2 sqr (the two products of y3 share one reduction), digit-wise subtractions and one carry
// normalisation; nothing is reduced modulo p.
// Digit-magnitude bookkeeping (see wide.hip.h): acc.x, acc.y, acc.zz, acc.zzz are normalised; every subtraction below
// takes two normalised values, so its result is a valid mul / sqr operand; x3 (four terms) and y3 are normalised
// before they are stored.  Value bookkeeping: with |x| < 3.3 p, |y| < 1.2 p at entry, P and Rr stay below 4.3 p and
// 2.3 p, all products land in (-0.15 p, 1.15 p), and x3, y3 satisfy the entry bounds again.
#pragma once
#include "curve.hip.h"
#include "wide.hip.h"

namespace zkhip {

struct g1_xyzz_w {
    fw x, y, zz, zzz;
    bool inf;   // identity (coordinates unspecified)
};

// 2 * (x, y), an affine point that is not the identity; y may be a lazy negation
template <class W = FqW>
__device__ __forceinline__ void xyzz_w_double_affine(g1_xyzz_w& acc, const fw& x, const fw& y) {
    const fw yn = W::norm(y);
    const fw u = W::norm(W::dbl(yn));
    const fw v = W::sqr(u);
    const fw w = W::mul(u, v);
    const fw s = W::mul(x, v);
    const fw x2 = W::sqr(x);
    const fw m = W::norm(W::add(W::dbl(x2), x2));
    const fw x3 = W::norm(W::sub(W::sqr(m), W::dbl(s)));
    acc.y = W::mul_sub_mul(m, W::sub(s, x3), w, yn);
    acc.x = x3;
    acc.zz = v;
    acc.zzz = w;
    acc.inf = false;
}

// acc = (qx, +-qy); q is in the packed in-memory form (identity = all zero words)
template <class W = FqW>
__device__ __forceinline__ void xyzz_w_set_affine(g1_xyzz_w& acc, const g1_affine& q, bool negate) {
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) any |= q.x.l[i] | q.y.l[i];
    acc.x = W::from_fe(q.x);
    acc.y = W::norm(W::cneg(W::from_fe(q.y), negate));
    acc.zz = W::one();
    acc.zzz = W::one();
    acc.inf = any == 0;
}

// acc += (qx, +-qy); q is in the packed in-memory form (identity = all zero words)
template <class W = FqW>
__device__ __forceinline__ void xyzz_w_add_affine(g1_xyzz_w& acc, const g1_affine& q, bool negate) {
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) any |= q.x.l[i] | q.y.l[i];
    if (any == 0) return;
    if (__builtin_expect(acc.inf, 0)) {
        xyzz_w_set_affine<W>(acc, q, negate);
        return;
    }
    const fw qx = W::from_fe(q.x);
    const fw qy = W::cneg(W::from_fe(q.y), negate);
    const fw u2 = W::mul(qx, acc.zz);
    const fw s2 = W::mul(qy, acc.zzz);
    const fw p = W::sub(u2, acc.x);
    const fw r = W::sub(s2, acc.y);
    if (W::maybe_zero(p)) {          // almost never taken: the exact test costs a full canonicalisation
        if (W::is_zero(p)) {
            if (W::is_zero(r)) xyzz_w_double_affine<W>(acc, qx, qy);
            else acc.inf = true;
            return;
        }
    }
    const fw pp = W::sqr(p);
    const fw ppp = W::mul(p, pp);
    const fw qq = W::mul(acc.x, pp);
    const fw x3 = W::norm(W::sub(W::sub(W::sqr(r), ppp), W::dbl(qq)));
    acc.y = W::mul_sub_mul(r, W::sub(qq, x3), acc.y, ppp);
    acc.x = x3;
    acc.zz = W::mul(acc.zz, pp);
    acc.zzz = W::mul(acc.zzz, ppp);
}

// 2 * a (a not the identity)
__device__ __forceinline__ g1_xyzz_w xyzz_w_double(const g1_xyzz_w& a) {
    g1_xyzz_w o;
    const fw u = FqW::norm(FqW::dbl(a.y));
    const fw v = FqW::sqr(u);
    const fw w = FqW::mul(u, v);
    const fw s = FqW::mul(a.x, v);
    const fw x2 = FqW::sqr(a.x);
    const fw m = FqW::norm(FqW::add(FqW::dbl(x2), x2));
    o.x = FqW::norm(FqW::sub(FqW::sqr(m), FqW::dbl(s)));
    o.y = FqW::mul_sub_mul(m, FqW::sub(s, o.x), w, a.y);
    o.zz = FqW::mul(v, a.zz);
    o.zzz = FqW::mul(w, a.zzz);
    o.inf = false;
    return o;
}

// a + b, both with normalised coordinates
__device__ __forceinline__ g1_xyzz_w xyzz_w_add(const g1_xyzz_w& a, const g1_xyzz_w& b) {
    if (a.inf) return b;
    if (b.inf) return a;
    const fw u1 = FqW::mul(a.x, b.zz);
    const fw u2 = FqW::mul(b.x, a.zz);
    const fw s1 = FqW::mul(a.y, b.zzz);
    const fw s2 = FqW::mul(b.y, a.zzz);
    const fw p = FqW::sub(u2, u1);
    const fw r = FqW::sub(s2, s1);
    if (FqW::maybe_zero(p)) {
        if (FqW::is_zero(p)) {
            if (FqW::is_zero(r)) return xyzz_w_double(a);
            g1_xyzz_w o = a;
            o.inf = true;
            return o;
        }
    }
    g1_xyzz_w o;
    const fw pp = FqW::sqr(p);
    const fw ppp = FqW::mul(p, pp);
    const fw qq = FqW::mul(u1, pp);
    o.x = FqW::norm(FqW::sub(FqW::sub(FqW::sqr(r), ppp), FqW::dbl(qq)));
    o.y = FqW::mul_sub_mul(r, FqW::sub(qq, o.x), s1, ppp);
    o.zz = FqW::mul(FqW::mul(a.zz, b.zz), pp);
    o.zzz = FqW::mul(FqW::mul(a.zzz, b.zzz), ppp);
    o.inf = false;
    return o;
}

// ---- a + b by TWO neighbouring lanes ---------------------------------------------------------------
// For kernels that are one dependency chain of point additions per wave (the MSM's reduce): lanes 2i and 2i + 1 hold
// the same a and b, each computes half of the field products, and both end with the same sum.  Lane role 0 takes the
// x side (u1, u2, zz_a zz_b, p^2, u1 p^2, zz3, r (q - x3)), role 1 the y side (s1, s2, zzz_a zzz_b, r^2, p^3, zzz3,
// s1 p^3): six products and a square per lane instead of twelve and two, with five exchanges of one value between the
// two lanes (data-parallel-primitive moves inside a quad, no LDS).  Both lanes of a pair must be active together.
__device__ __forceinline__ fw fw_pair_other(const fw& v) {
    fw o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.d[i] = __builtin_amdgcn_update_dpp(0, v.d[i], 0xB1, 0xf, 0xf, false);   // quad_perm:[1,0,3,2]
    return o;
}
__device__ __forceinline__ fw fw_sel(bool c, const fw& t, const fw& f) {
    fw o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.d[i] = c ? t.d[i] : f.d[i];
    return o;
}
__device__ __forceinline__ g1_xyzz_w xyzz_w_add_pair(const g1_xyzz_w& a, const g1_xyzz_w& b, bool role) {
    if (a.inf) return b;
    if (b.inf) return a;
    const fw xy1 = fw_sel(role, a.y, a.x), xy2 = fw_sel(role, b.y, b.x);
    const fw z1 = fw_sel(role, a.zzz, a.zz), z2 = fw_sel(role, b.zzz, b.zz);
    const fw t1 = FqW::mul(xy1, z2);            // u1 | s1
    const fw t2 = FqW::mul(xy2, z1);            // u2 | s2
    const fw t3 = FqW::mul(z1, z2);             // zz_a zz_b | zzz_a zzz_b
    const fw d = FqW::sub(t2, t1);              // p | r
    // p == 0 (equal or opposite points) is decided by the x lane for the pair; the pair then takes the one-lane
    // addition with its exact tests, both lanes alike
    const int mz = FqW::maybe_zero(d) ? 1 : 0;
    const int mz_o = __builtin_amdgcn_update_dpp(0, mz, 0xB1, 0xf, 0xf, false);   // read with both lanes active
    const int mz_x = role ? mz_o : mz;
    if (__builtin_expect(mz_x != 0, 0)) return xyzz_w_add(a, b);
    const fw sq = FqW::sqr(d);                  // p^2 | r^2
    const fw sq_o = fw_pair_other(sq), d_o = fw_pair_other(d);
    const fw pp = fw_sel(role, sq_o, sq), rr = fw_sel(role, sq, sq_o);
    const fw m1 = FqW::mul(fw_sel(role, d_o, t1), pp);        // q = u1 p^2 | p^3
    const fw m2 = FqW::mul(t3, fw_sel(role, m1, pp));         // zz3 | zzz3
    const fw m1_o = fw_pair_other(m1);
    const fw qq = fw_sel(role, m1_o, m1), ppp = fw_sel(role, m1, m1_o);
    g1_xyzz_w o;
    o.x = FqW::norm(FqW::sub(FqW::sub(rr, ppp), FqW::dbl(qq)));
    const fw m3 = FqW::mul(fw_sel(role, t1, d_o), fw_sel(role, ppp, FqW::sub(qq, o.x)));   // r (q - x3) | s1 p^3
    const fw m3_o = fw_pair_other(m3), m2_o = fw_pair_other(m2);
    o.y = FqW::norm(role ? FqW::sub(m3_o, m3) : FqW::sub(m3, m3_o));
    o.zz = fw_sel(role, m2_o, m2);
    o.zzz = fw_sel(role, m2, m2_o);
    o.inf = false;
    return o;
}

__device__ __forceinline__ g1_xyzz_w xyzz_w_identity() {
    g1_xyzz_w o;
    o.x = o.y = o.zz = o.zzz = FqW::zero();
    o.inf = true;
    return o;
}

// Packed XYZZ in memory (canonical residues, identity: zz = 0), coordinates in the 2^261 domain.
__device__ __forceinline__ g1_xyzz_w xyzz_w_unpack(const g1_xyzz& v) {
    g1_xyzz_w o;
    o.inf = Fq::is_zero(v.zz);
    o.x = FqW::from_fe(v.x);
    o.y = FqW::from_fe(v.y);
    o.zz = FqW::from_fe(v.zz);
    o.zzz = FqW::from_fe(v.zzz);
    return o;
}
__device__ __forceinline__ g1_xyzz_w xyzz_w_load(const g1_xyzz* p) { return xyzz_w_unpack(load_xyzz(p)); }
__device__ __forceinline__ g1_xyzz xyzz_w_pack(const g1_xyzz_w& a) {
    if (a.inf) return xyzz_identity();
    g1_xyzz o;
    o.x = FqW::to_fe(a.x);
    o.y = FqW::to_fe(a.y);
    o.zz = FqW::to_fe(a.zz);
    o.zzz = FqW::to_fe(a.zzz);
    return o;
}
__device__ __forceinline__ void xyzz_w_store(g1_xyzz* p, const g1_xyzz_w& a) { store_xyzz(p, xyzz_w_pack(a)); }

// c ? t : f, field by field (a conditional expression over whole structs selects between their ADDRESSES, which puts both
// on the stack)
__device__ __forceinline__ g1_xyzz_w xyzz_w_sel(bool c, const g1_xyzz_w& t, const g1_xyzz_w& f) {
    g1_xyzz_w o;
    o.x = fw_sel(c, t.x, f.x);
    o.y = fw_sel(c, t.y, f.y);
    o.zz = fw_sel(c, t.zz, f.zz);
    o.zzz = fw_sel(c, t.zzz, f.zzz);
    o.inf = c ? t.inf : f.inf;
    return o;
}
// the point held by the other lane of a pair (lanes 2 i <-> 2 i + 1)
__device__ __forceinline__ g1_xyzz_w xyzz_w_pair_other(const g1_xyzz_w& v) {
    g1_xyzz_w o;
    o.x = fw_pair_other(v.x);
    o.y = fw_pair_other(v.y);
    o.zz = fw_pair_other(v.zz);
    o.zzz = fw_pair_other(v.zzz);
    o.inf = __builtin_amdgcn_update_dpp(0, (int)v.inf, 0xB1, 0xf, 0xf, false) != 0;
    return o;
}
__device__ __forceinline__ g1_xyzz_w xyzz_w_shfl(const g1_xyzz_w& v, int src_lane) {
    g1_xyzz_w o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        o.x.d[i] = __shfl(v.x.d[i], src_lane, 64);
        o.y.d[i] = __shfl(v.y.d[i], src_lane, 64);
        o.zz.d[i] = __shfl(v.zz.d[i], src_lane, 64);
        o.zzz.d[i] = __shfl(v.zzz.d[i], src_lane, 64);
    }
    o.inf = __shfl((int)v.inf, src_lane, 64) != 0;
    return o;
}

}  // namespace zkhip
