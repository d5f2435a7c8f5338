// Host-side BN254 helpers of libzkhip: 4 x u64 Montgomery field arithmetic and Jacobian G1.
// Used only for O(10^2..10^3)-operation set-up and fold steps (domain constants, twiddle seeds, folding a
// handful of partial sums); every per-element path runs on the device.
#include <cstring>
#include "common.h"

namespace zkhip {

typedef unsigned __int128 u128;

const HostField HFR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
    {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL},
    0xc2e1f593efffffffULL};
const HostField HFQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
    {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL},
    0x87d20782e4866389ULL};

static inline bool geq(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return true;
}
static inline void sub_nb(uint64_t a[4], const uint64_t b[4]) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - br;
        a[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
}
bool h_is_zero(const uint64_t a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
bool h_eq(const uint64_t a[4], const uint64_t b[4]) { return memcmp(a, b, 32) == 0; }

void h_add(const HostField& F, uint64_t o[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t t[4];
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a[i] + b[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || geq(t, F.p)) sub_nb(t, F.p);
    memcpy(o, t, 32);
}
void h_sub(const HostField& F, uint64_t o[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t t[4], br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - br;
        t[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t[i] + F.p[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(o, t, 32);
}
void h_mul(const HostField& F, uint64_t o[4], const uint64_t a[4], const uint64_t b[4]) {
    // Coarsely integrated operand scanning, unrolled.  Both BN254 moduli leave the two top bits of the high limb clear,
    // so the running value fits four limbs plus one carry limb and no second carry chain is needed.
    const uint64_t p0 = F.p[0], p1 = F.p[1], p2 = F.p[2], p3 = F.p[3];
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (int i = 0; i < 4; i++) {
        const uint64_t bi = b[i];
        u128 c = (u128)a[0] * bi + t0;
        t0 = (uint64_t)c;
        c = (u128)a[1] * bi + t1 + (uint64_t)(c >> 64);
        t1 = (uint64_t)c;
        c = (u128)a[2] * bi + t2 + (uint64_t)(c >> 64);
        t2 = (uint64_t)c;
        c = (u128)a[3] * bi + t3 + (uint64_t)(c >> 64);
        t3 = (uint64_t)c;
        const uint64_t t4 = (uint64_t)(c >> 64);
        const uint64_t m = t0 * F.inv;
        c = (u128)m * p0 + t0;
        c = (u128)m * p1 + t1 + (uint64_t)(c >> 64);
        t0 = (uint64_t)c;
        c = (u128)m * p2 + t2 + (uint64_t)(c >> 64);
        t1 = (uint64_t)c;
        c = (u128)m * p3 + t3 + (uint64_t)(c >> 64);
        t2 = (uint64_t)c;
        t3 = t4 + (uint64_t)(c >> 64);
    }
    uint64_t r[4] = {t0, t1, t2, t3};
    if (geq(r, F.p)) sub_nb(r, F.p);
    memcpy(o, r, 32);
}
void h_pow(const HostField& F, uint64_t o[4], const uint64_t a[4], uint64_t e) {
    uint64_t acc[4], base[4];
    memcpy(acc, F.r, 32);
    memcpy(base, a, 32);
    while (e) {
        if (e & 1) h_mul(F, acc, acc, base);
        h_mul(F, base, base, base);
        e >>= 1;
    }
    memcpy(o, acc, 32);
}
void h_inv(const HostField& F, uint64_t o[4], const uint64_t a[4]) {
    // a^(p-2)
    uint64_t e[4];
    memcpy(e, F.p, 32);
    e[0] -= 2;
    uint64_t acc[4];
    memcpy(acc, F.r, 32);
    for (int i = 255; i >= 0; i--) {
        h_mul(F, acc, acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) h_mul(F, acc, acc, a);
    }
    memcpy(o, acc, 32);
}
void h_from_u64(const HostField& F, uint64_t o[4], uint64_t v) {
    const uint64_t a[4] = {v, 0, 0, 0};
    h_mul(F, o, a, F.r2);
}

// ---- Jacobian G1 (y^2 = x^3 + 3), layout x||y||z, identity z = 0 -----------------------------------
void h_g1_double(uint64_t o[12], const uint64_t p[12]) {
    const HostField& F = HFQ;
    if (h_is_zero(p + 8)) {
        memmove(o, p, 96);
        return;
    }
    uint64_t a[4], b[4], c[4], d[4], e[4], f[4], t[4], x3[4], y3[4], z3[4];
    h_mul(F, a, p, p);
    h_mul(F, b, p + 4, p + 4);
    h_mul(F, c, b, b);
    h_add(F, d, p, b);
    h_mul(F, d, d, d);
    h_sub(F, d, d, a);
    h_sub(F, d, d, c);
    h_add(F, d, d, d);
    h_add(F, e, a, a);
    h_add(F, e, e, a);
    h_mul(F, f, e, e);
    h_mul(F, z3, p + 4, p + 8);
    h_add(F, z3, z3, z3);
    h_sub(F, x3, f, d);
    h_sub(F, x3, x3, d);
    h_add(F, c, c, c);
    h_add(F, c, c, c);
    h_add(F, c, c, c);
    h_sub(F, t, d, x3);
    h_mul(F, t, e, t);
    h_sub(F, y3, t, c);
    memcpy(o, x3, 32);
    memcpy(o + 4, y3, 32);
    memcpy(o + 8, z3, 32);
}
void h_g1_add(uint64_t o[12], const uint64_t p[12], const uint64_t q[12]) {
    const HostField& F = HFQ;
    if (h_is_zero(p + 8)) {
        memmove(o, q, 96);
        return;
    }
    if (h_is_zero(q + 8)) {
        memmove(o, p, 96);
        return;
    }
    uint64_t z1z1[4], z2z2[4], u1[4], u2[4], s1[4], s2[4], h[4], i[4], j[4], r[4], v[4], t[4], x3[4], y3[4], z3[4];
    h_mul(F, z1z1, p + 8, p + 8);
    h_mul(F, z2z2, q + 8, q + 8);
    h_mul(F, u1, p, z2z2);
    h_mul(F, u2, q, z1z1);
    h_mul(F, s1, p + 4, q + 8);
    h_mul(F, s1, s1, z2z2);
    h_mul(F, s2, q + 4, p + 8);
    h_mul(F, s2, s2, z1z1);
    if (h_eq(u1, u2)) {
        if (h_eq(s1, s2)) {
            h_g1_double(o, p);
            return;
        }
        memset(o, 0, 96);
        return;
    }
    h_sub(F, h, u2, u1);
    h_add(F, i, h, h);
    h_mul(F, i, i, i);
    h_mul(F, j, h, i);
    h_sub(F, r, s2, s1);
    h_add(F, r, r, r);
    h_mul(F, v, u1, i);
    h_mul(F, x3, r, r);
    h_sub(F, x3, x3, j);
    h_sub(F, x3, x3, v);
    h_sub(F, x3, x3, v);
    h_mul(F, t, s1, j);
    h_add(F, t, t, t);
    h_sub(F, y3, v, x3);
    h_mul(F, y3, y3, r);
    h_sub(F, y3, y3, t);
    h_add(F, z3, p + 8, q + 8);
    h_mul(F, z3, z3, z3);
    h_sub(F, z3, z3, z1z1);
    h_sub(F, z3, z3, z2z2);
    h_mul(F, z3, z3, h);
    memcpy(o, x3, 32);
    memcpy(o + 4, y3, 32);
    memcpy(o + 8, z3, 32);
}
void h_g1_to_affine(uint64_t o[8], const uint64_t p[12]) {
    const HostField& F = HFQ;
    if (h_is_zero(p + 8)) {
        memset(o, 0, 64);
        return;
    }
    uint64_t zi[4], zi2[4], zi3[4];
    h_inv(F, zi, p + 8);
    h_mul(F, zi2, zi, zi);
    h_mul(F, zi3, zi2, zi);
    h_mul(F, o, p, zi2);
    h_mul(F, o + 4, p + 4, zi3);
}

}  // namespace zkhip
