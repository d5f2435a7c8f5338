// BN254 Fr / Fq arithmetic for gfx950.
//
// Storage type `fe` is the in-memory form of `halo2curves::bn256::{Fr,Fq}`: 8 x u32 little-endian limbs
// (= 4 x u64), Montgomery form with R = 2^256.  add/sub run on the packed limbs with the hardware carry
// chain.  mul unpacks both operands into nine 29-bit digits so that the 9x9 schoolbook product and the
// Montgomery reduction accumulate in 64-bit columns with v_mad_u64_u32 and never need a carry: measured on
// MI355X this reaches 1.7x the throughput of a 32-bit-digit CIOS loop, whose every multiply-add drags a
// zero-extension move and a 64-bit add behind it.  The reduction retires eight 29-bit digits and one 24-bit
// digit (232 + 24 = 256 bits), so the result is again in R = 2^256 form and bit-identical to the 4 x u64
// CIOS product of the CPU implementation.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "field_constants.h"

namespace zkhip {

struct alignas(16) fe {
    uint32_t l[8];
};

#define ZK_DEV __host__ __device__ __forceinline__

// ({hi,lo} >> s)[31:0], 0 < s < 32 (v_alignbit_b32 on the device)
ZK_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}

constexpr uint32_t M29 = (1u << 29) - 1;

// Hides a value's known bit range from the optimiser.  hipcc (ROCm 7.2) turns a 64-bit multiply-add whose
// operands are both provably below 2^24 into a 24-bit multiply node, drops the `& mask` that proved it, and
// then emits a full v_mad_u64_u32 on the unmasked register: the product silently gains the masked-off bits.
// Every digit that is narrower than 25 bits passes through here before it is multiplied.
ZK_DEV uint32_t opaque(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// nine 29-bit digits of a packed 256-bit value (value < 2^261 always holds)
struct d29 {
    uint32_t d[9];
};

ZK_DEV d29 unpack29(const fe& a) {
    d29 o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, s = bit & 31;
        uint32_t v;
        if (s == 0) v = a.l[w];
        else if (w + 1 < 8) v = alignbit(a.l[w + 1], a.l[w], s);
        else v = a.l[w] >> s;
        o.d[i] = (i == 8) ? opaque(v & M29) : (v & M29);
    }
    return o;
}

template <class FP>
struct Field {
    ZK_DEV static fe zero() {
        fe o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = 0;
        return o;
    }
    ZK_DEV static fe one() {
        fe o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = FP::ONE[i];
        return o;
    }
    ZK_DEV static bool is_zero(const fe& a) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) v |= a.l[i];
        return v == 0;
    }
    ZK_DEV static bool eq(const fe& a, const fe& b) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) v |= a.l[i] ^ b.l[i];
        return v == 0;
    }
    // t (+ 2^256 * top) reduced once: returns t - p when t >= p
    ZK_DEV static fe cond_sub(const uint32_t t[8], uint32_t top) {
        uint32_t r[8];
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)t[i] - FP::P[i] - br;
            r[i] = (uint32_t)d;
            br = (uint32_t)(d >> 63);
        }
        const bool ge = (top != 0) | (br == 0);
        fe o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = ge ? r[i] : t[i];
        return o;
    }
    ZK_DEV static fe add(const fe& a, const fe& b) {
        uint32_t t[8];
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t s = (uint64_t)a.l[i] + b.l[i] + c;
            t[i] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        return cond_sub(t, c);   // p < 2^254 so c is always 0; kept for generality
    }
    ZK_DEV static fe dbl(const fe& a) { return add(a, a); }
    ZK_DEV static fe sub(const fe& a, const fe& b) {
        uint32_t t[8];
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)a.l[i] - b.l[i] - br;
            t[i] = (uint32_t)d;
            br = (uint32_t)(d >> 63);
        }
        const uint32_t mask = 0u - br;
        uint32_t c = 0;
        fe o;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t s = (uint64_t)t[i] + (FP::P[i] & mask) + c;
            o.l[i] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        return o;
    }
    ZK_DEV static fe neg(const fe& a) {
        const uint32_t mask = is_zero(a) ? 0u : 0xffffffffu;
        uint32_t br = 0;
        fe o;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)FP::P[i] - a.l[i] - br;
            o.l[i] = (uint32_t)d & mask;
            br = (uint32_t)(d >> 63);
        }
        return o;
    }

    // Montgomery reduction of an 18-column product (columns of weight 2^(29k)), R = 2^256.
    ZK_DEV static fe reduce(uint64_t col[18]) {
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const uint32_t dm = (k < 8) ? M29 : ((1u << 24) - 1);
            uint32_t m = ((uint32_t)col[k] * FP::INV29) & dm;
            if (k == 8) m = opaque(m);
#pragma unroll
            for (int j = 0; j < 9; j++) col[k + j] += (uint64_t)m * FP::P29[j];
            if (k < 8) col[k + 1] += col[k] >> 29;
        }
        // col[8] now has 24 zero low bits; V = (sum_{k>=8} col[k] 2^(29(k-8))) >> 24 < 2p
        uint32_t n[10];
        uint64_t c = 0;
#pragma unroll
        for (int k = 8; k < 18; k++) {
            const uint64_t s = col[k] + c;
            n[k - 8] = (uint32_t)s & M29;
            c = s >> 29;
        }
        uint32_t t[8];
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const int bit = 32 * w + 24, i = bit / 29, s = bit % 29;
            uint32_t v = n[i] >> s;
            if (i + 1 < 10) v |= n[i + 1] << (29 - s);
            if (58 - s < 32 && i + 2 < 10) v |= n[i + 2] << (58 - s);
            t[w] = v;
        }
        return cond_sub(t, 0);
    }

    ZK_DEV static fe mul(const fe& a, const fe& b) {
        const d29 x = unpack29(a), y = unpack29(b);
        uint64_t col[18];
#pragma unroll
        for (int k = 0; k < 18; k++) col[k] = 0;
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) col[i + j] += (uint64_t)x.d[i] * y.d[j];
        return reduce(col);
    }
    ZK_DEV static fe sqr(const fe& a) {
        const d29 x = unpack29(a);
        uint64_t col[18];
#pragma unroll
        for (int k = 0; k < 18; k++) col[k] = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            col[2 * i] += (uint64_t)x.d[i] * x.d[i];
            const uint32_t x2 = x.d[i] << 1;
#pragma unroll
            for (int j = i + 1; j < 9; j++) col[i + j] += (uint64_t)x2 * x.d[j];
        }
        return reduce(col);
    }
    // a * b where b is already unpacked (kernels that reuse one factor, e.g. a twiddle, many times)
    ZK_DEV static fe mul_d29(const fe& a, const d29& y) {
        const d29 x = unpack29(a);
        uint64_t col[18];
#pragma unroll
        for (int k = 0; k < 18; k++) col[k] = 0;
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) col[i + j] += (uint64_t)x.d[i] * y.d[j];
        return reduce(col);
    }
    // a^e for a small exponent (host-side setup helpers use the CPU; this is for in-kernel powers)
    ZK_DEV static fe pow_u64(fe a, uint64_t e) {
        fe acc = one();
        while (e) {
            if (e & 1) acc = mul(acc, a);
            a = sqr(a);
            e >>= 1;
        }
        return acc;
    }
};

using Fr = Field<FrParams>;
using Fq = Field<FqParams>;

// 32-byte loads / stores as two 16-byte vector accesses
ZK_DEV fe load_fe(const fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    fe o;
    o.l[0] = a.x; o.l[1] = a.y; o.l[2] = a.z; o.l[3] = a.w;
    o.l[4] = b.x; o.l[5] = b.y; o.l[6] = b.z; o.l[7] = b.w;
    return o;
}
ZK_DEV void store_fe(fe* p, const fe& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

}  // namespace zkhip
