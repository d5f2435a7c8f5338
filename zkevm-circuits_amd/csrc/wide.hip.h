// Lazy signed-digit arithmetic for BN254 Fr / Fq inside kernels ("wide" form).
//
// A value is nine signed 32-bit digits of weight 2^(29 i); the represented integer is only fixed modulo p and
// may be negative or exceed p.  add / sub / neg work digit by digit with no carries and no comparison against
// p.  mul is a Montgomery multiplication with R' = 2^261 (all nine digits retired), computed with signed
// 64-bit multiply-adds (v_mad_i64_i32) in 64-bit columns:
//
//     mul(a, b) = (a b + m p) / 2^261,   0 <= m < 2^261   =>   a b / 2^261 <= result < a b / 2^261 + p
//
// Because 2^261 / p > 147, a product of operands of a few p in magnitude lands back in (-p/10, 1.1 p): the
// multiplications themselves keep every value small and no modular reduction is ever executed between
// operations.  What callers must keep track of is the digit magnitude:
//
//     "normalised"  digits 0..7 in [0, 2^29), digit 8 signed and small       (mul, sqr, norm, from_fe results)
//     "lazy"        sums / differences of normalised values, |digit| < 2^31
//
//   mul / sqr operands: the column sums need  9 |a_i| |b_j| + 9 * 2^58 < 2^63, so a difference of two normalised
//   values (|digit| < 2^29) is a valid operand next to another such difference or a normalised value; a sum of two
//   normalised values (|digit| < 2^30) is valid next to a normalised value only.  Anything larger goes through
//   norm() first (25 instructions, against ~24 for ONE packed modular addition).
//
// Elements enter with from_fe (the packed in-memory form, any Montgomery domain: wide form does not care what
// the residue means) and leave with to_fe, which produces the canonical packed residue in [0, p).
// A value x kept as x 2^261 in wide form is obtained from the in-memory x 2^256 by one packed multiplication
// with FP::TWO5, and returns to x 2^256 through mul(v, r256()).
#pragma once
#include "field.hip.h"

namespace zkhip {

struct fw {
    int32_t d[9];
};

// CHAIN: the multiplier adds every column's terms to its accumulator as ONE chain of dependent multiply-adds (see chain()):
// the shortest instruction stream (the optimiser's own form is two interleaved chains per column joined by a 64-bit
// addition).  The empty asm statements that pin the chain make the compiler put a wait state (s_nop 0) behind each of them
// -- it must assume that an asm statement wrote its output with a partial-register write -- which halves a wave's issue
// rate; the library's build removes those wait states from the assembly (zkevm-circuits_amd/build.py), and with them gone
// the chain form is the faster one in every kernel, the lone-wave dependency chains of the MSM's merge and reduce included
// (with the wait states in place those were 25 % slower in this form, which is why the parameter exists).
#ifndef ZK_CHAIN_DEFAULT
#define ZK_CHAIN_DEFAULT true
#endif
template <class FP, bool CHAIN = ZK_CHAIN_DEFAULT>
struct Wide {
    // Hides from the optimiser what it has proved about a digit (a masked digit is non-negative).  A product of a
    // sign-extended digit and a digit known to be non-negative is otherwise lowered to an UNSIGNED 64-bit multiply-add
    // plus a second multiply-add and two moves that correct the high word for the sign; with the knowledge hidden both
    // factors are plain sign-extended 32-bit values and the product is one v_mad_i64_i32.  Emits no instruction.  Applied
    // where masked digits are made (from_fe, norm, reduce): the unveiled value dies there, so no copy is needed either.
    ZK_DEV static int32_t veil(int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(v));
#endif
        return v;
    }
    ZK_DEV static fw veil(const fw& a) {
        fw o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.d[i] = veil(a.d[i]);
        return o;
    }
    ZK_DEV static fw zero() {
        fw o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.d[i] = 0;
        return o;
    }
    // packed 256-bit value (< 2^256) -> normalised digits
    ZK_DEV static fw from_fe(const fe& a) {
        const d29 u = unpack29(a);
        fw o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.d[i] = veil((int32_t)u.d[i]);
        return o;
    }
    // normalised digits of 32 * a for a packed a < 2^256 / 4: the same residue moved from the 2^256 to the 2^261
    // Montgomery domain (x 2^256 * 32 = x 2^261) at the price of an ordinary unpack; the value may reach 32 p
    ZK_DEV static fw from_fe_shl5(const fe& a) {
        fw o;
        o.d[0] = veil((int32_t)((a.l[0] << 5) & M29));
#pragma unroll
        for (int i = 1; i < 9; i++) {
            const int bit = 29 * i - 5, w = bit >> 5, s = bit & 31;
            uint32_t v;
            if (s == 0) v = a.l[w];
            else if (w + 1 < 8) v = alignbit(a.l[w + 1], a.l[w], s);
            else v = a.l[w] >> s;
            o.d[i] = veil((int32_t)(v & M29));
        }
        return o;
    }
    ZK_DEV static fw from_words(const uint32_t w[8]) {
        fe t;
#pragma unroll
        for (int i = 0; i < 8; i++) t.l[i] = w[i];
        return from_fe(t);
    }
    ZK_DEV static fw one() { return from_words(FP::TWO5); }    // 1 in the 2^261 Montgomery domain
    ZK_DEV static fw r256() { return from_words(FP::ONE); }    // 2^256 mod p as an integer: mul(x 2^261, r256) = x 2^256

    ZK_DEV static fw add(const fw& a, const fw& b) {
        fw o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.d[i] = a.d[i] + b.d[i];
        return o;
    }
    ZK_DEV static fw sub(const fw& a, const fw& b) {
        fw o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.d[i] = a.d[i] - b.d[i];
        return o;
    }
    ZK_DEV static fw neg(const fw& a) {
        fw o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.d[i] = -a.d[i];
        return o;
    }
    ZK_DEV static fw dbl(const fw& a) {
        fw o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.d[i] = a.d[i] << 1;
        return o;
    }
    // negate when flag is set
    ZK_DEV static fw cneg(const fw& a, bool flag) {
        const int32_t m = flag ? -1 : 0;
        fw o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.d[i] = (a.d[i] ^ m) - m;
        return o;
    }
    // carry propagation: digits 0..7 into [0, 2^29), the rest of the value into the signed top digit
    ZK_DEV static fw norm(const fw& a) {
        fw o;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = a.d[i] + c;
            o.d[i] = veil(t & (int32_t)M29);
            c = t >> 29;
        }
        o.d[8] = a.d[8] + c;
        return o;
    }

    // Montgomery reduction by 2^261 of 17 signed product columns; normalised result
    ZK_DEV static fw reduce(int64_t col[18]) {
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int32_t m = (int32_t)(((uint32_t)col[k] * FP::INV29) & M29);
#pragma unroll
            for (int j = 0; j < 9; j++) col[k + j] += (int64_t)m * (int32_t)FP::P29[j];
            col[k + 1] += col[k] >> 29;
        }
        fw o;
#pragma unroll
        for (int k = 9; k < 17; k++) {
            o.d[k - 9] = veil((int32_t)((uint32_t)col[k] & M29));
            col[k + 1] += col[k] >> 29;
        }
        o.d[8] = (int32_t)col[17];
        return o;
    }
    // col += a b, the 81 digit products in their 17 columns (col[17] only takes carries in reduce)
    ZK_DEV static void mac(int64_t col[18], const fw& a, const fw& b) {
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) col[i + j] += (int64_t)a.d[i] * b.d[j];
    }
    // Column-wise Montgomery product with ONE running 64-bit accumulator: for output column k the digit products of
    // weight 2^(29 k) and the reduction terms m_i p_(k-i) are added to the accumulator, the column's multiplier m_k (k < 9)
    // or output digit (k >= 9) is taken from its low 29 bits, and the accumulator is shifted down by one digit.  Against
    // the row-wise form (mac + reduce: 18 columns in registers, every column's carry shifted out AND added into the next
    // column) this keeps two registers instead of thirty-six and saves one 64-bit addition per column.
    // products(k, acc) adds the digit products of column k (0 <= k <= 16) to acc.
    // acc >> 29 as the START of the next column's chain of multiply-adds: the optimiser otherwise re-associates the sum,
    // starts the chain from zero and adds the shifted carry at the end with a 64-bit addition of its own
    ZK_DEV static int64_t carry_out(int64_t acc) {
        acc >>= 29;
#if defined(__HIP_DEVICE_COMPILE__)
        if (CHAIN) asm("" : "+v"(acc));
#endif
        return acc;
    }
    // acc + x y as one link of a chain of multiply-adds (the sum is opaque afterwards, so the chain keeps its order)
    ZK_DEV static void chain(int64_t& acc, int32_t x, int32_t y) {
        acc += (int64_t)x * y;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_CHAIN_VEIL)
        if (CHAIN) asm("" : "+v"(acc));
#endif
    }
    template <class F>
    ZK_DEV static fw comba(F&& products) {
        int64_t acc = 0;
        int32_t m[9];
        fw o;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            products(k, acc);
#pragma unroll
            for (int i = 0; i < k; i++) chain(acc, m[i], (int32_t)FP::P29[k - i]);
            m[k] = (int32_t)(((uint32_t)acc * FP::INV29) & M29);
            chain(acc, m[k], (int32_t)FP::P29[0]);
            acc = carry_out(acc);
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
            products(k, acc);
#pragma unroll
            for (int i = k - 8; i < 9; i++) chain(acc, m[i], (int32_t)FP::P29[k - i]);
            o.d[k - 9] = veil((int32_t)((uint32_t)acc & M29));
            acc = carry_out(acc);
        }
        o.d[8] = (int32_t)acc;
        return o;
    }
    ZK_DEV static fw mul(const fw& a, const fw& b) {
        return comba([&](int k, int64_t& acc) {
#pragma unroll
            for (int i = 0; i < 9; i++)
                if (k - i >= 0 && k - i < 9) chain(acc, a.d[i], b.d[k - i]);
        });
    }
    // (a b - c d) / 2^261 with ONE reduction: both products share the columns.  All four operands must be normalised
    // or differences of normalised values (|digit| < 2^29): 18 terms of 2^58 plus the reduction stay below 2^63.
    ZK_DEV static fw mul_sub_mul(const fw& a, const fw& b, const fw& c, const fw& d) {
        const fw nd = neg(d);
        return comba([&](int k, int64_t& acc) {
#pragma unroll
            for (int i = 0; i < 9; i++)
                if (k - i >= 0 && k - i < 9) chain(acc, a.d[i], b.d[k - i]);
#pragma unroll
            for (int i = 0; i < 9; i++)
                if (k - i >= 0 && k - i < 9) chain(acc, c.d[i], nd.d[k - i]);
        });
    }
    // (a b + c d) / 2^261 with ONE reduction; same operand rule as mul_sub_mul
    ZK_DEV static fw mul_add_mul(const fw& a, const fw& b, const fw& c, const fw& d) {
        return comba([&](int k, int64_t& acc) {
#pragma unroll
            for (int i = 0; i < 9; i++)
                if (k - i >= 0 && k - i < 9) chain(acc, a.d[i], b.d[k - i]);
#pragma unroll
            for (int i = 0; i < 9; i++)
                if (k - i >= 0 && k - i < 9) chain(acc, c.d[i], d.d[k - i]);
        });
    }
    // operand digits must stay below 2^29 in magnitude (the doubled cross terms use 2 a_i)
    ZK_DEV static fw sqr(const fw& a) {
        fw a2;
#pragma unroll
        for (int i = 0; i < 9; i++) a2.d[i] = a.d[i] << 1;
        return comba([&](int k, int64_t& acc) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (j > i && j < 9) chain(acc, a2.d[i], a.d[j]);
                if (j == i) chain(acc, a.d[i], a.d[i]);
            }
        });
    }

    // Cheap necessary condition for value == 0 (mod p) when |value| < 16 p: value = k p forces the low 29 bits to be
    // k p mod 2^29, i.e. low * p^-1 mod 2^29 lies within [-16, 16].  Passes a non-multiple with probability 2^-24.
    ZK_DEV static bool maybe_zero(const fw& a) {
        const uint32_t k = ((uint32_t)a.d[0] * (0u - FP::INV29)) & M29;   // p^-1 = -(-p^-1)
        return ((k + 16u) & M29) <= 32u;
    }

    // canonical packed residue in [0, p) of a value with |value| < 64 p
    ZK_DEV static fe to_fe(const fw& v) {
        fw a = norm(v);
        // q ~ floor(value / p) from the top digit, deliberately two short so that the remainder is positive
        const float qf = floorf((float)a.d[8] * (1.0f / (float)(FP::P29[8] + 1)));
        const int32_t q = (int32_t)qf - 2;
        int64_t c = 0;
        fw w;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int64_t t = (int64_t)a.d[i] - (int64_t)q * (int32_t)FP::P29[i] + c;
            if (i < 8) {
                w.d[i] = (int32_t)((uint32_t)t & M29);
                c = t >> 29;
            } else {
                w.d[i] = (int32_t)t;
            }
        }
        // now p <= w < 4 p (the float estimate may be one off in either direction): subtract p while it fits
#pragma unroll
        for (int r = 0; r < 4; r++) {
            fw t;
            int32_t cc = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int32_t s = w.d[i] - (int32_t)FP::P29[i] + cc;
                t.d[i] = s & (int32_t)M29;
                cc = s >> 29;
            }
            t.d[8] = w.d[8] - (int32_t)FP::P29[8] + cc;
            const bool keep = t.d[8] >= 0;
#pragma unroll
            for (int i = 0; i < 9; i++) w.d[i] = keep ? t.d[i] : w.d[i];
        }
        // pack nine 29-bit digits (top digit < 2^22) into 8 words
        uint32_t n[10];
#pragma unroll
        for (int i = 0; i < 9; i++) n[i] = (uint32_t)w.d[i];
        n[9] = 0;
        fe o;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int bit = 32 * k, i = bit / 29, s = bit % 29;
            uint32_t x = n[i] >> s;
            if (i + 1 < 10) x |= n[i + 1] << (29 - s);
            if (58 - s < 32 && i + 2 < 10) x |= n[i + 2] << (58 - s);
            o.l[k] = x;
        }
        return o;
    }
    // packs a normalised value known to lie in [0, 2^256) without reducing it modulo p
    ZK_DEV static fe pack_nonneg(const fw& w) {
        uint32_t n[10];
#pragma unroll
        for (int i = 0; i < 9; i++) n[i] = (uint32_t)w.d[i];
        n[9] = 0;
        fe o;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int bit = 32 * k, i = bit / 29, s = bit % 29;
            uint32_t x = n[i] >> s;
            if (i + 1 < 10) x |= n[i + 1] << (29 - s);
            if (58 - s < 32 && i + 2 < 10) x |= n[i + 2] << (58 - s);
            o.l[k] = x;
        }
        return o;
    }
    // exact test for value == 0 (mod p), |value| < 64 p
    ZK_DEV static bool is_zero(const fw& v) {
        const fe c = to_fe(v);
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) x |= c.l[i];
        return x == 0;
    }
};

using FrW = Wide<FrParams>;
using FqW = Wide<FqParams>;
using FqWChain = Wide<FqParams, true>;

}  // namespace zkhip
