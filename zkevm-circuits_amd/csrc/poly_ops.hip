// Device versions of the prover steps right next to the hot path: the grand products of the permutation and lookup
// arguments and polynomial evaluation at a point.
//
//   `ff::BatchInvert::batch_invert`                    -> zkhip_batch_invert_dev
//   running product z[i+1] = z[i] * m[i]               -> zkhip_prefix_product_dev
//   `plonk::permutation::prover::Argument::commit`     -> zkhip_permutation_product_dev   (one column set)
//   `plonk::lookup::prover::Permuted::commit_product`  -> zkhip_lookup_product_dev
//   `arithmetic::eval_polynomial`                      -> zkhip_eval_polynomial_dev
//   `arithmetic::kate_division`                        -> zkhip_kate_division_dev
//   sums of scaled polynomials (`poly * scalar`, `+`), as the SHPLONK prover forms them for its quotient and
//   linearisation polynomials (`poly::kzg::multiopen::shplonk::prover`)   -> zkhip_poly_lincomb_dev
//
// The blinding rows are left to the caller: upstream draws them from the prover's RNG in a fixed order, so the host
// overwrites the last `blinding_factors` rows of z after these calls and reads last_z from row n - blinding - 1.
// Field inverses are unique, so any exact inversion scheme reproduces upstream's values bit for bit.
#include <algorithm>
#include <cstring>
#include <vector>
#include "common.h"
#include "domain.h"
#include "wide.hip.h"

using namespace zkhip;

namespace zkhip {

constexpr int PO_THREADS = 256;

// x^(p - 2) in the lazy digit form (2^261 Montgomery domain inside, 2^256 domain in and out): 254 squarings and about
// 127 products of ~245 instructions each instead of ~345 in the packed form -- this chain is the latency of every
// batch inversion
template <class F, class FP>
ZK_DEV fe field_invert(const fe& x) {
    using W = Wide<FP>;
    const fw xw = W::from_fe_shl5(x);      // x 2^256 -> x 2^261
    fw acc = W::one();
    for (int bit = 255; bit >= 0; bit--) {
        acc = W::sqr(acc);
        uint32_t word = FP::P[bit >> 5];
        if ((bit >> 5) == 0) word -= 2;   // exponent p - 2; the low word of p is > 2
        if ((word >> (bit & 31)) & 1) acc = W::mul(acc, xw);
    }
    return W::to_fe(W::mul(acc, W::r256()));
}

// ---- batch inversion ---------------------------------------------------------------------------------
// Lane t owns the elements t, t + S, t + 2S, ... (S = lanes of the launch; coalesced).  Forward: running products of
// its elements into `prefix`; one Fermat inversion of the lane's total; backward: a_k^-1 = prefix_k * inv, inv *= a_k.
// Zeros are skipped and stay zero.
template <class F, class FP>
__global__ void __launch_bounds__(PO_THREADS) batch_invert_kernel(fe* a, fe* prefix, size_t n, size_t lanes) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes) return;
    fe run = F::one();
    for (size_t i = t; i < n; i += lanes) {
        store_fe(prefix + i, run);
        const fe v = load_fe(a + i);
        if (!F::is_zero(v)) run = F::mul(run, v);
    }
    fe inv = field_invert<F, FP>(run);
    // backward over the same elements
    size_t last = t + ((n - 1 - t) / lanes) * lanes;
    for (size_t i = last;; i -= lanes) {
        const fe v = load_fe(a + i);
        if (!F::is_zero(v)) {
            store_fe(a + i, F::mul(load_fe(prefix + i), inv));
            inv = F::mul(inv, v);
        }
        if (i < lanes) break;
    }
}

// The same over `total` elements that live in several vectors of 2^log_n elements each (element i is element
// i mod 2^log_n of vecs[i >> log_n]): the inversions of many grand products share one launch, whose lanes then hold enough
// elements each (64 at 32 vectors of 2^19) that the 380-multiplication chain is a small part of a lane's work and every
// SIMD has four waves to interleave -- a lone vector of 2^19 elements gives a lane 8 elements and a SIMD one wave.
template <class F, class FP>
__global__ void __launch_bounds__(PO_THREADS) batch_invert_multi_kernel(fe* const* vecs, uint32_t log_n, size_t total, fe* prefix,
                                                                        size_t lanes) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes || t >= total) return;
    const size_t mask = ((size_t)1 << log_n) - 1;
    fe run = F::one();
    for (size_t i = t; i < total; i += lanes) {
        store_fe(prefix + i, run);
        const fe v = load_fe(vecs[i >> log_n] + (i & mask));
        if (!F::is_zero(v)) run = F::mul(run, v);
    }
    fe inv = field_invert<F, FP>(run);
    size_t last = t + ((total - 1 - t) / lanes) * lanes;
    for (size_t i = last;; i -= lanes) {
        fe* at = vecs[i >> log_n] + (i & mask);
        const fe v = load_fe(at);
        if (!F::is_zero(v)) {
            store_fe(at, F::mul(load_fe(prefix + i), inv));
            inv = F::mul(inv, v);
        }
        if (i < lanes) break;
    }
}

// ---- prefix product ----------------------------------------------------------------------------------
constexpr int PP_PER = 8;                       // elements per lane
constexpr int PP_BLOCK = PO_THREADS * PP_PER;   // elements per block

__device__ __forceinline__ fe block_exclusive_product(const fe& v, fe* sh /*[PO_THREADS]*/, fe* total) {
    // Hillis-Steele inclusive scan, then shift
    const uint32_t tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    fe cur = v;
    for (uint32_t off = 1; off < PO_THREADS; off <<= 1) {
        fe other;
        const bool take = tid >= off;
        if (take) other = sh[tid - off];
        __syncthreads();
        if (take) {
            cur = Fr::mul(other, cur);
            sh[tid] = cur;
        }
        __syncthreads();
    }
    if (total) *total = sh[PO_THREADS - 1];
    fe excl = tid ? sh[tid - 1] : Fr::one();
    __syncthreads();
    return excl;
}

// phase 1: out[i] = product of m[block start .. i) within the block; totals[block] = product of the block's elements
__global__ void __launch_bounds__(PO_THREADS) prefix_local_kernel(const fe* m, fe* out, size_t n, fe* totals) {
    __shared__ fe sh[PO_THREADS];
    const size_t base = (size_t)blockIdx.x * PP_BLOCK + (size_t)threadIdx.x * PP_PER;
    fe loc[PP_PER];
    fe run = Fr::one();
#pragma unroll
    for (int k = 0; k < PP_PER; k++) {
        loc[k] = run;
        if (base + k < n) run = Fr::mul(run, load_fe(m + base + k));
    }
    fe total;
    const fe excl = block_exclusive_product(run, sh, &total);
#pragma unroll
    for (int k = 0; k < PP_PER; k++)
        if (base + k < n) store_fe(out + base + k, Fr::mul(excl, loc[k]));
    if (threadIdx.x == 0) store_fe(totals + blockIdx.x, total);
}
// phase 2 (one block): totals[b] <- init * product of totals[0 .. b)
__global__ void __launch_bounds__(PO_THREADS) prefix_totals_kernel(fe* totals, size_t nblocks, fe init) {
    __shared__ fe sh[PO_THREADS];
    fe carry = init;
    for (size_t base = 0; base < nblocks; base += PO_THREADS) {
        const size_t i = base + threadIdx.x;
        const fe v = i < nblocks ? load_fe(totals + i) : Fr::one();
        fe total;
        const fe excl = block_exclusive_product(v, sh, &total);
        if (i < nblocks) store_fe(totals + i, Fr::mul(carry, excl));
        carry = Fr::mul(carry, total);
    }
}
// phase 3: out[i] *= totals[block]
__global__ void __launch_bounds__(PO_THREADS) prefix_apply_kernel(fe* out, size_t n, const fe* totals) {
    const fe off = load_fe(totals + blockIdx.x);
    const size_t base = (size_t)blockIdx.x * PP_BLOCK;
    for (int k = 0; k < PP_PER; k++) {
        const size_t i = base + (size_t)k * PO_THREADS + threadIdx.x;
        if (i < n) store_fe(out + i, Fr::mul(load_fe(out + i), off));
    }
}

// ---- permutation / lookup fractions ------------------------------------------------------------------
struct PermArgs {
    const fe* const* values;   // device array of column pointers (Lagrange basis)
    const fe* const* sigmas;
    uint32_t num_cols;
    fe beta, gamma, delta_start;   // delta_start = delta^(index of the set's first column)
    const fe* om_lo;
    const fe* om_hi;
    uint32_t h;
};
// out[i] = prod_j (beta * sigma_j[i] + gamma + v_j[i])
__global__ void __launch_bounds__(PO_THREADS) perm_denominator_kernel(const PermArgs a, fe* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe acc = Fr::one();
    for (uint32_t j = 0; j < a.num_cols; j++) {
        const fe t = Fr::add(Fr::add(Fr::mul(a.beta, load_fe(a.sigmas[j] + i)), a.gamma), load_fe(a.values[j] + i));
        acc = Fr::mul(acc, t);
    }
    store_fe(out + i, acc);
}
// io[i] *= prod_j (delta^(j0 + j) * beta * omega^i + gamma + v_j[i])
__global__ void __launch_bounds__(PO_THREADS) perm_numerator_kernel(const PermArgs a, fe* io, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe w = load_fe(a.om_lo + (i & ((1u << a.h) - 1)));
    if (i >> a.h) w = Fr::mul(w, load_fe(a.om_hi + (i >> a.h)));
    fe cur = Fr::mul(Fr::mul(w, a.beta), a.delta_start);
    fe delta;
#pragma unroll
    for (int k = 0; k < 8; k++) delta.l[k] = FrParams::DELTA[k];
    fe acc = load_fe(io + i);
    for (uint32_t j = 0; j < a.num_cols; j++) {
        acc = Fr::mul(acc, Fr::add(Fr::add(cur, a.gamma), load_fe(a.values[j] + i)));
        cur = Fr::mul(cur, delta);
    }
    store_fe(io + i, acc);
}
// out[i] = (x[i] + beta) * (y[i] + gamma)        (lookup denominators from a', s')
__global__ void __launch_bounds__(PO_THREADS) shifted_product_kernel(const fe* x, const fe* y, fe beta, fe gamma, fe* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_fe(out + i, Fr::mul(Fr::add(load_fe(x + i), beta), Fr::add(load_fe(y + i), gamma)));
}
// io[i] *= (x[i] + beta) * (y[i] + gamma)         (lookup numerators from the compressed expressions)
__global__ void __launch_bounds__(PO_THREADS) shifted_product_mul_kernel(const fe* x, const fe* y, fe beta, fe gamma, fe* io, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe t = Fr::mul(Fr::add(load_fe(x + i), beta), Fr::add(load_fe(y + i), gamma));
    store_fe(io + i, Fr::mul(load_fe(io + i), t));
}

// ---- polynomial evaluation ---------------------------------------------------------------------------
// Lane t evaluates its EP_PER consecutive coefficients by Horner, scales by x^(t * EP_PER) (x_per = x^EP_PER given)
// and the block sums its lanes; partial[block] goes to the host.
constexpr int EP_PER = 16;
__global__ void __launch_bounds__(PO_THREADS) eval_poly_kernel(const fe* coeff, size_t n, fe x, fe x_per, fe* partial) {
    __shared__ fe sh[PO_THREADS];
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t base = t * EP_PER;
    fe acc = Fr::zero();
    if (base < n) {
        const size_t hi = base + EP_PER < n ? base + EP_PER : n;
        for (size_t i = hi; i-- > base;) acc = Fr::add(Fr::mul(acc, x), load_fe(coeff + i));
        // x^(t * EP_PER) = x_per^t
        fe p = Fr::one(), b = x_per;
        for (size_t e = t; e; e >>= 1) {
            if (e & 1) p = Fr::mul(p, b);
            b = Fr::sqr(b);
        }
        acc = Fr::mul(acc, p);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t half = PO_THREADS / 2; half >= 1; half >>= 1) {
        if (threadIdx.x < half) sh[threadIdx.x] = Fr::add(sh[threadIdx.x], sh[threadIdx.x + half]);
        __syncthreads();
    }
    if (threadIdx.x == 0) store_fe(partial + blockIdx.x, sh[0]);
}

// ---- sum of scaled polynomials -----------------------------------------------------------------------
// out[i] = sum_k c_k p_k[i] (p_k zero beyond its own length).  The table holds, per polynomial, its device pointer,
// its length and the coefficient; all lanes of a launch walk it together, so it is read with scalar loads.  Terms
// are taken two at a time: two products share one Montgomery reduction in the lazy digit form.
struct LincombTerm {
    const fe* poly;
    uint64_t len;
    uint32_t coeff[8];
};
constexpr int LC_PER = 2;   // elements per lane
__global__ void __launch_bounds__(PO_THREADS) poly_lincomb_kernel(const LincombTerm* terms_generic, uint32_t num, size_t n, fe* out) {
    const LincombTerm __attribute__((address_space(4)))* terms =
        (const LincombTerm __attribute__((address_space(4)))*)(uintptr_t)terms_generic;
    const size_t i0 = ((size_t)blockIdx.x * PO_THREADS + threadIdx.x) * LC_PER;
    if (i0 >= n) return;
    fw acc[LC_PER];
#pragma unroll
    for (int e = 0; e < LC_PER; e++) acc[e] = FrW::zero();
    uint32_t since = 0;
    for (uint32_t k = 0; k < num; k += 2) {
        const bool two = k + 1 < num;
        fe c0, c1;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            c0.l[w] = terms[k].coeff[w];
            c1.l[w] = two ? terms[k + 1].coeff[w] : 0u;
        }
        const fe* p0 = (const fe*)terms[k].poly;
        const fe* p1 = two ? (const fe*)terms[k + 1].poly : p0;
        const size_t l0 = terms[k].len, l1 = two ? (size_t)terms[k + 1].len : 0;
        const fw w0 = FrW::from_fe_shl5(c0), w1 = FrW::from_fe_shl5(c1);
#pragma unroll
        for (int e = 0; e < LC_PER; e++) {
            const size_t i = i0 + e;
            fe v0 = Fr::zero(), v1 = Fr::zero();
            if (i < l0) v0 = load_fe(p0 + i);
            if (i < l1) v1 = load_fe(p1 + i);
            acc[e] = FrW::norm(FrW::add(acc[e], FrW::mul_add_mul(w0, FrW::from_fe(v0), w1, FrW::from_fe(v1))));
        }
        // a pair adds less than 1.5 p; fold the sum back below p long before the 64 p limit of to_fe
        if (++since == 32) {
            since = 0;
#pragma unroll
            for (int e = 0; e < LC_PER; e++) acc[e] = FrW::from_fe(FrW::to_fe(acc[e]));
        }
    }
#pragma unroll
    for (int e = 0; e < LC_PER; e++)
        if (i0 + e < n) store_fe(out + i0 + e, FrW::to_fe(acc[e]));
}

// ---- division by (X - z) -------------------------------------------------------------------------------
// q_j = S_(j+1) with S_j = a_j + z S_(j+1) (S_n = 0): a backward Horner recurrence.  A lane owns KD_PER consecutive
// coefficients; first every lane runs the recurrence from zero, the lane results are chained by a suffix scan with the
// powers z^(KD_PER 2^d) inside the block and z^(KD_BLOCK 2^d) over the blocks, and a second sweep reruns the recurrence
// from each lane's true incoming value.  pw: [0..8) = z^(KD_PER 2^d), [8..16) = z^(KD_BLOCK 2^d), [16..272) = z^(KD_PER t).
constexpr int KD_PER = 8;
constexpr int KD_BLOCK = PO_THREADS * KD_PER;

// suffix scan over the lanes of a block: on return lane t holds sum_{u >= t} v_u w^(u - t), w = step[0]; step[d] = w^(2^d)
__device__ __forceinline__ fe block_suffix_horner(fe v, const fe* step, fe* sh /*[PO_THREADS]*/) {
    const uint32_t tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    int d = 0;
    for (uint32_t off = 1; off < PO_THREADS; off <<= 1, d++) {
        fe other;
        const bool take = tid + off < PO_THREADS;
        if (take) other = sh[tid + off];
        __syncthreads();
        if (take) {
            v = Fr::add(v, Fr::mul(load_fe(step + d), other));
            sh[tid] = v;
        }
        __syncthreads();
    }
    return v;
}

__global__ void __launch_bounds__(PO_THREADS) kate_local_kernel(const fe* a, size_t n, fe z, const fe* pw, fe* lane_in, fe* block_tot) {
    __shared__ fe sh[PO_THREADS];
    const size_t base = (size_t)blockIdx.x * KD_BLOCK + (size_t)threadIdx.x * KD_PER;
    fe s = Fr::zero();
#pragma unroll
    for (int k = KD_PER - 1; k >= 0; k--) {
        const size_t i = base + k;
        s = Fr::mul(s, z);
        if (i < n) s = Fr::add(s, load_fe(a + i));
    }
    const fe incl = block_suffix_horner(s, pw, sh);   // value at this lane's first coefficient, zero beyond the block
    // value entering this lane from above (at the next lane's first coefficient), still without the block's carry
    __syncthreads();
    sh[threadIdx.x] = incl;
    __syncthreads();
    const fe in = threadIdx.x + 1 < PO_THREADS ? sh[threadIdx.x + 1] : Fr::zero();
    store_fe(lane_in + (size_t)blockIdx.x * PO_THREADS + threadIdx.x, in);
    if (threadIdx.x == 0) store_fe(block_tot + blockIdx.x, incl);
}
// one block: block_tot[b] <- value entering block b from above = sum_{c > b} tot_c z^(KD_BLOCK (c - b - 1))
__global__ void __launch_bounds__(PO_THREADS) kate_blocks_kernel(fe* block_tot, size_t nblocks, const fe* pw) {
    __shared__ fe sh[PO_THREADS];
    __shared__ fe carry_s;
    // chunks of PO_THREADS blocks from the top down; carry = value entering the chunk from above
    fe carry = Fr::zero();
    const size_t nchunks = (nblocks + PO_THREADS - 1) / PO_THREADS;
    fe wchunk = load_fe(pw + 8 + 7);   // z^(KD_BLOCK 128)
    wchunk = Fr::sqr(wchunk);          // z^(KD_BLOCK 256): one chunk further
    for (size_t ch = nchunks; ch-- > 0;) {
        const size_t b = ch * PO_THREADS + threadIdx.x;
        const fe v = b < nblocks ? load_fe(block_tot + b) : Fr::zero();
        const fe incl = block_suffix_horner(v, pw + 8, sh);   // sum_{c >= b in chunk} tot_c w^(c - b)
        __syncthreads();
        sh[threadIdx.x] = incl;
        __syncthreads();
        // entering block b: blocks above it in the chunk, plus the carry moved down over the PO_THREADS - 1 - tid
        // blocks between b and the top of the chunk
        fe in = threadIdx.x + 1 < PO_THREADS ? sh[threadIdx.x + 1] : Fr::zero();
        fe p = Fr::one(), bs = load_fe(pw + 8);
        for (uint32_t e = PO_THREADS - 1 - threadIdx.x; e; e >>= 1) {
            if (e & 1) p = Fr::mul(p, bs);
            bs = Fr::sqr(bs);
        }
        in = Fr::add(in, Fr::mul(p, carry));
        if (b < nblocks) store_fe(block_tot + b, in);
        if (threadIdx.x == 0) carry_s = Fr::add(incl, Fr::mul(wchunk, carry));
        __syncthreads();
        carry = carry_s;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(PO_THREADS) kate_apply_kernel(const fe* a, size_t n, fe z, const fe* pw, const fe* lane_in,
                                                                const fe* block_in, fe* q) {
    const size_t base = (size_t)blockIdx.x * KD_BLOCK + (size_t)threadIdx.x * KD_PER;
    // incoming value: the block-local part plus the block's carry moved down over the lanes above this one
    fe s = Fr::add(load_fe(lane_in + (size_t)blockIdx.x * PO_THREADS + threadIdx.x),
                   Fr::mul(load_fe(pw + 16 + (PO_THREADS - 1 - threadIdx.x)), load_fe(block_in + blockIdx.x)));
#pragma unroll
    for (int k = KD_PER - 1; k >= 0; k--) {
        const size_t i = base + k;
        // here s = S_(i+1); q_i = S_(i+1) for i <= n - 2
        if (i + 1 < n) store_fe(q + i, s);
        s = Fr::mul(s, z);
        if (i < n) s = Fr::add(s, load_fe(a + i));
    }
}

// ---- permutation polynomials at key generation ----------------------------------------------------------
// sigma[i] = delta^(col[i]) * omega^(row[i]): the Lagrange-basis values of one permutation polynomial from the cycle
// mapping of its column (`plonk::permutation::keygen::Assembly::build_pk`).  dpow: delta^c for c < num_cols.
__global__ void __launch_bounds__(PO_THREADS) perm_sigma_kernel(const uint32_t* col, const uint32_t* row, size_t n, uint32_t num_cols,
                                                                const fe* dpow, const fe* om_lo, const fe* om_hi, uint32_t h,
                                                                fe* out, uint32_t* bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = row[i];
    if (r >= n || col[i] >= num_cols) {   // a cell outside the argument: report it instead of reading past the tables
        atomicOr(bad, 1u);
        store_fe(out + i, Fr::zero());
        return;
    }
    fe om = load_fe(om_lo + (r & ((1u << h) - 1)));
    if (r >> h) om = Fr::mul(om, load_fe(om_hi + (r >> h)));
    store_fe(out + i, Fr::mul(om, load_fe(dpow + col[i])));
}

// many (polynomial, point) pairs in one launch: blockIdx.y = query, the same lane work as eval_poly_kernel
struct EvalQuery {
    const fe* coeff;
    fe x, x_per;
};
// Coefficients per lane of the batched kernel: with many queries in flight there are lanes enough, and a longer Horner
// run amortises the lane's x_per^t (up to 2 log2(t) products) over four times as many coefficients.
constexpr int EPB_PER = 64;
__global__ void __launch_bounds__(PO_THREADS) eval_poly_batch_kernel(const EvalQuery* queries, size_t n, fe* partial) {
    __shared__ fe sh[PO_THREADS];
    const EvalQuery q = queries[blockIdx.y];
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t base = t * EPB_PER;
    fe acc = Fr::zero();
    if (base < n) {
        const size_t hi = base + EPB_PER < n ? base + EPB_PER : n;
        for (size_t i = hi; i-- > base;) acc = Fr::add(Fr::mul(acc, q.x), load_fe(q.coeff + i));
        fe p = Fr::one(), b = q.x_per;
        for (size_t e = t; e; e >>= 1) {
            if (e & 1) p = Fr::mul(p, b);
            b = Fr::sqr(b);
        }
        acc = Fr::mul(acc, p);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t half = PO_THREADS / 2; half >= 1; half >>= 1) {
        if (threadIdx.x < half) sh[threadIdx.x] = Fr::add(sh[threadIdx.x], sh[threadIdx.x + half]);
        __syncthreads();
    }
    if (threadIdx.x == 0) store_fe(partial + (size_t)blockIdx.y * gridDim.x + blockIdx.x, sh[0]);
}

// ---- launch helpers ----------------------------------------------------------------------------------
static fe fe_from(const uint64_t v[4]) {
    fe o;
    memcpy(o.l, v, 32);
    return o;
}

static size_t invert_lanes(const zkhip_ctx* ctx, size_t n) {
    // 16 elements per lane amortise the 380-multiplication inversion; never fewer lanes than one wave per SIMD
    size_t lanes = (n + 15) / 16;
    const size_t floor_lanes = (size_t)ctx->num_cus * 256;
    if (lanes < floor_lanes) lanes = std::min(n, floor_lanes);
    return lanes;
}

// a (n elements, device) -> inverses in place; tmp: n elements of scratch
static int batch_invert_dev(zkhip_ctx* ctx, int field, fe* d_a, fe* d_tmp, size_t n) {
    if (n == 0) return ZKHIP_OK;
    const size_t lanes = invert_lanes(ctx, n);
    const dim3 grid((unsigned)((lanes + PO_THREADS - 1) / PO_THREADS));
    if (field == 0)
        hipLaunchKernelGGL((batch_invert_kernel<Fr, FrParams>), grid, dim3(PO_THREADS), 0, ctx->stream, d_a, d_tmp, n, lanes);
    else
        hipLaunchKernelGGL((batch_invert_kernel<Fq, FqParams>), grid, dim3(PO_THREADS), 0, ctx->stream, d_a, d_tmp, n, lanes);
    ZK_HIP(ctx, hipGetLastError());
    return ZKHIP_OK;
}

// out[0] = init, out[i] = init * m[0] * ... * m[i-1]; totals: ceil(n / PP_BLOCK) elements of scratch.  m may equal out.
static int prefix_product_dev(zkhip_ctx* ctx, const fe* d_m, fe* d_out, fe* d_totals, size_t n, const fe& init) {
    if (n == 0) return ZKHIP_OK;
    const size_t nblocks = (n + PP_BLOCK - 1) / PP_BLOCK;
    hipLaunchKernelGGL(prefix_local_kernel, dim3((unsigned)nblocks), dim3(PO_THREADS), 0, ctx->stream, d_m, d_out, n, d_totals);
    hipLaunchKernelGGL(prefix_totals_kernel, dim3(1), dim3(PO_THREADS), 0, ctx->stream, d_totals, nblocks, init);
    hipLaunchKernelGGL(prefix_apply_kernel, dim3((unsigned)nblocks), dim3(PO_THREADS), 0, ctx->stream, d_out, n, d_totals);
    ZK_HIP(ctx, hipGetLastError());
    return ZKHIP_OK;
}

}  // namespace zkhip

// ---- running product for callers that cannot allocate ---------------------------------------------------
// Block totals of zkhip_prefix_product_noalloc_dev: one element per block of PP_BLOCK inputs.  The array belongs to the
// code object, so every GPU the library runs on holds its own copy at its own address for the life of the process.
// Host code never passes the array's name to a kernel (on the host it is a stand-in, not a device address):
// poly_ops_init resolves the address once per device and the entry point hands that pointer to the kernels.
constexpr size_t PPNA_MAX_N = (size_t)1 << 24;
constexpr size_t PPNA_MAX_BLOCKS = PPNA_MAX_N / zkhip::PP_BLOCK;
static __device__ __align__(256) unsigned char scratch[PPNA_MAX_BLOCKS * 32];
constexpr int PPNA_MAX_DEVICES = 64;
static void* g_static_totals[PPNA_MAX_DEVICES];   // device address of `scratch` on each device, nullptr until resolved

namespace zkhip {
// called by zkhip_create with the context's device current
int poly_ops_init(zkhip_ctx* ctx) {
    if (ctx->device < 0 || ctx->device >= PPNA_MAX_DEVICES) return ZKHIP_OK;   // the no-alloc entry point reports it
    if (g_static_totals[ctx->device]) return ZKHIP_OK;
    void* ptr = nullptr;
    ZK_HIP(ctx, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    g_static_totals[ctx->device] = ptr;
    return ZKHIP_OK;
}
}  // namespace zkhip

extern "C" {

int zkhip_prefix_product_noalloc_dev(zkhip_ctx* ctx, const uint64_t* d_m, uint64_t* d_out, size_t n, const uint64_t init[4]) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, (d_m && d_out && init) || n == 0, "prefix_product_noalloc: null pointer");
    ZK_CHECK(ctx, n <= PPNA_MAX_N, "prefix_product_noalloc: %zu elements, the static block-total buffer holds at most %zu",
             n, PPNA_MAX_N);
    ZK_CHECK(ctx, ctx->device >= 0 && ctx->device < PPNA_MAX_DEVICES && g_static_totals[ctx->device],
             "prefix_product_noalloc: the static buffer was not resolved on device %d", ctx->device);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    // every block total the second and third kernel read is written by the first kernel of this same call, so nothing
    // left over from an earlier call is ever read and nothing needs clearing
    return prefix_product_dev(ctx, (const fe*)d_m, (fe*)d_out, (fe*)g_static_totals[ctx->device], n, fe_from(init));
}

int zkhip_batch_invert_dev(zkhip_ctx* ctx, int field, uint64_t* d_a, size_t n) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, field == 0 || field == 1, "batch_invert: field %d", field);
    ZK_CHECK(ctx, d_a || n == 0, "batch_invert: null pointer");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_scratch(ctx, n * 32);
    if (rc) return rc;
    TimedRegion tr(ctx, "batch_invert");
    return batch_invert_dev(ctx, field, (fe*)d_a, (fe*)ctx->scratch, n);
}

int zkhip_prefix_product_dev(zkhip_ctx* ctx, const uint64_t* d_m, uint64_t* d_out, size_t n, const uint64_t init[4]) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, (d_m && d_out && init) || n == 0, "prefix_product: null pointer");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nblocks = (n + PP_BLOCK - 1) / PP_BLOCK;
    int rc = ensure_scratch(ctx, (nblocks + 1) * 32);
    if (rc) return rc;
    TimedRegion tr(ctx, "prefix_product");
    return prefix_product_dev(ctx, (const fe*)d_m, (fe*)d_out, (fe*)ctx->scratch, n, fe_from(init));
}

int zkhip_permutation_product_dev(zkhip_domain* dom, const uint64_t* const* d_values, const uint64_t* const* d_sigmas,
                                  uint32_t num_cols, const uint64_t beta[4], const uint64_t gamma[4],
                                  const uint64_t delta_start[4], const uint64_t z0[4], uint64_t* d_z) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_values && d_sigmas && beta && gamma && delta_start && z0 && d_z, "permutation_product: null pointer");
    ZK_CHECK(ctx, num_cols >= 1 && num_cols <= 64, "permutation_product: %u columns in one set", num_cols);
    for (uint32_t j = 0; j < num_cols; j++)
        ZK_CHECK(ctx, d_values[j] && d_sigmas[j], "permutation_product: null column %u", j);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << dom->k;
    const size_t nblocks = (n + PP_BLOCK - 1) / PP_BLOCK;
    // scratch: [prefix temp n][totals nblocks][pointer tables]
    const size_t ptr_bytes = 2 * (size_t)num_cols * sizeof(void*);
    int rc = ensure_scratch(ctx, (n + nblocks + 1) * 32 + ptr_bytes);
    if (rc) return rc;
    fe* d_tmp = (fe*)ctx->scratch;
    fe* d_totals = d_tmp + n;
    const fe** d_ptrs = (const fe**)(d_totals + nblocks + 1);
    ZK_HIP(ctx, hipMemcpyAsync(d_ptrs, d_values, num_cols * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(d_ptrs + num_cols, d_sigmas, num_cols * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the tables are the caller's pageable host memory, free to die on return
    PermArgs a;
    a.values = d_ptrs;
    a.sigmas = d_ptrs + num_cols;
    a.num_cols = num_cols;
    a.beta = fe_from(beta);
    a.gamma = fe_from(gamma);
    a.delta_start = fe_from(delta_start);
    a.om_lo = (const fe*)dom->plan_fwd->d_lo;
    a.om_hi = (const fe*)dom->plan_fwd->d_hi;
    a.h = dom->plan_fwd->h;
    TimedRegion tr(ctx, "grand_product");
    const dim3 grid((unsigned)((n + PO_THREADS - 1) / PO_THREADS));
    fe* z = (fe*)d_z;
    hipLaunchKernelGGL(perm_denominator_kernel, grid, dim3(PO_THREADS), 0, ctx->stream, a, z, n);
    ZK_HIP(ctx, hipGetLastError());
    rc = batch_invert_dev(ctx, 0, z, d_tmp, n);
    if (rc) return rc;
    hipLaunchKernelGGL(perm_numerator_kernel, grid, dim3(PO_THREADS), 0, ctx->stream, a, z, n);
    ZK_HIP(ctx, hipGetLastError());
    return prefix_product_dev(ctx, z, z, d_totals, n, fe_from(z0));
}

int zkhip_lookup_product_dev(zkhip_domain* dom, const uint64_t* d_compressed_input, const uint64_t* d_compressed_table,
                             const uint64_t* d_permuted_input, const uint64_t* d_permuted_table, const uint64_t beta[4],
                             const uint64_t gamma[4], uint64_t* d_z) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_compressed_input && d_compressed_table && d_permuted_input && d_permuted_table && beta && gamma && d_z,
             "lookup_product: null pointer");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << dom->k;
    const size_t nblocks = (n + PP_BLOCK - 1) / PP_BLOCK;
    int rc = ensure_scratch(ctx, (n + nblocks + 1) * 32);
    if (rc) return rc;
    fe* d_tmp = (fe*)ctx->scratch;
    fe* d_totals = d_tmp + n;
    TimedRegion tr(ctx, "grand_product");
    const dim3 grid((unsigned)((n + PO_THREADS - 1) / PO_THREADS));
    fe* z = (fe*)d_z;
    const fe b = fe_from(beta), g = fe_from(gamma);
    hipLaunchKernelGGL(shifted_product_kernel, grid, dim3(PO_THREADS), 0, ctx->stream, (const fe*)d_permuted_input,
                       (const fe*)d_permuted_table, b, g, z, n);
    ZK_HIP(ctx, hipGetLastError());
    rc = batch_invert_dev(ctx, 0, z, d_tmp, n);
    if (rc) return rc;
    hipLaunchKernelGGL(shifted_product_mul_kernel, grid, dim3(PO_THREADS), 0, ctx->stream, (const fe*)d_compressed_input,
                       (const fe*)d_compressed_table, b, g, z, n);
    ZK_HIP(ctx, hipGetLastError());
    fe one;
    memcpy(one.l, HFR.r, 32);
    return prefix_product_dev(ctx, z, z, d_totals, n, one);
}

int zkhip_lookup_product_batch_dev(zkhip_domain* dom, const uint64_t* const* d_compressed_inputs,
                                   const uint64_t* const* d_compressed_tables, const uint64_t* const* d_permuted_inputs,
                                   const uint64_t* const* d_permuted_tables, const uint64_t beta[4], const uint64_t gamma[4],
                                   uint64_t* const* d_zs, uint32_t count) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    if (count == 0) return ZKHIP_OK;
    ZK_CHECK(ctx, d_compressed_inputs && d_compressed_tables && d_permuted_inputs && d_permuted_tables && beta && gamma && d_zs,
             "lookup_product_batch: null pointer");
    for (uint32_t i = 0; i < count; i++)
        ZK_CHECK(ctx, d_compressed_inputs[i] && d_compressed_tables[i] && d_permuted_inputs[i] && d_permuted_tables[i] && d_zs[i],
                 "lookup_product_batch: null vector of lookup %u", i);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    constexpr uint32_t GROUP = 32;          // lookups whose denominators are inverted by one launch
    const size_t n = (size_t)1 << dom->k;
    const size_t nblocks = (n + PP_BLOCK - 1) / PP_BLOCK;
    const uint32_t group = std::min(count, GROUP);
    // scratch: [inversion prefixes group * n][block totals nblocks + 1][pointers to the z vectors]
    int rc = ensure_scratch(ctx, ((size_t)group * n + nblocks + 1) * 32 + (size_t)count * sizeof(void*));
    if (rc) return rc;
    fe* d_tmp = (fe*)ctx->scratch;
    fe* d_totals = d_tmp + (size_t)group * n;
    fe** d_ptrs = (fe**)(d_totals + nblocks + 1);
    ZK_HIP(ctx, hipMemcpyAsync(d_ptrs, d_zs, (size_t)count * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the table is the caller's pageable host memory, free to die on return
    TimedRegion tr(ctx, "grand_product");
    const dim3 grid((unsigned)((n + PO_THREADS - 1) / PO_THREADS));
    const fe b = fe_from(beta), g = fe_from(gamma);
    fe one;
    memcpy(one.l, HFR.r, 32);
    for (uint32_t g0 = 0; g0 < count; g0 += GROUP) {
        const uint32_t cnt = std::min(GROUP, count - g0);
        for (uint32_t i = g0; i < g0 + cnt; i++)
            hipLaunchKernelGGL(shifted_product_kernel, grid, dim3(PO_THREADS), 0, ctx->stream, (const fe*)d_permuted_inputs[i],
                               (const fe*)d_permuted_tables[i], b, g, (fe*)d_zs[i], n);
        ZK_HIP(ctx, hipGetLastError());
        const size_t total = (size_t)cnt * n;
        // four waves on every SIMD when the group is large enough to give each lane 16 elements, never fewer than one
        const size_t lanes = std::max<size_t>((size_t)ctx->num_cus * 256, std::min<size_t>((size_t)ctx->num_cus * 1024, total / 16));
        hipLaunchKernelGGL((batch_invert_multi_kernel<Fr, FrParams>), dim3((unsigned)((lanes + PO_THREADS - 1) / PO_THREADS)),
                           dim3(PO_THREADS), 0, ctx->stream, d_ptrs + g0, dom->k, total, d_tmp, lanes);
        ZK_HIP(ctx, hipGetLastError());
        for (uint32_t i = g0; i < g0 + cnt; i++) {
            fe* z = (fe*)d_zs[i];
            hipLaunchKernelGGL(shifted_product_mul_kernel, grid, dim3(PO_THREADS), 0, ctx->stream, (const fe*)d_compressed_inputs[i],
                               (const fe*)d_compressed_tables[i], b, g, z, n);
            ZK_HIP(ctx, hipGetLastError());
            if ((rc = prefix_product_dev(ctx, z, z, d_totals, n, one))) return rc;
        }
    }
    return ZKHIP_OK;
}

int zkhip_eval_polynomial_dev(zkhip_ctx* ctx, const uint64_t* d_coeff, size_t n, const uint64_t x[4], uint64_t out[4]) {
    if (!ctx || !out) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, (d_coeff && x) || n == 0, "eval_polynomial: null pointer");
    memset(out, 0, 32);
    if (n == 0) return ZKHIP_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t lanes = (n + EP_PER - 1) / EP_PER;
    const size_t nblocks = (lanes + PO_THREADS - 1) / PO_THREADS;
    int rc = ensure_scratch(ctx, nblocks * 32);
    if (rc) return rc;
    uint64_t x_per[4];
    h_pow(HFR, x_per, x, EP_PER);
    {
        TimedRegion tr(ctx, "eval_polynomial");
        hipLaunchKernelGGL(eval_poly_kernel, dim3((unsigned)nblocks), dim3(PO_THREADS), 0, ctx->stream, (const fe*)d_coeff, n,
                           fe_from(x), fe_from(x_per), (fe*)ctx->scratch);
        ZK_HIP(ctx, hipGetLastError());
    }
    std::vector<uint64_t> part(nblocks * 4);
    ZK_HIP(ctx, hipMemcpyAsync(part.data(), ctx->scratch, nblocks * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // block b holds sum_t p_t x^(t EP_PER) with t global, so the partials simply add up
    uint64_t acc[4] = {0, 0, 0, 0};
    for (size_t b = 0; b < nblocks; b++) h_add(HFR, acc, acc, &part[4 * b]);
    memcpy(out, acc, 32);
    return ZKHIP_OK;
}

int zkhip_eval_polynomials_dev(zkhip_ctx* ctx, const uint64_t* const* d_coeffs, const uint64_t* points, uint32_t count, size_t n,
                               uint64_t* out) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, (d_coeffs && points && out) || count == 0, "eval_polynomials: null pointer");
    if (count == 0) return ZKHIP_OK;
    memset(out, 0, (size_t)count * 32);
    if (n == 0) return ZKHIP_OK;
    for (uint32_t i = 0; i < count; i++) ZK_CHECK(ctx, d_coeffs[i], "eval_polynomials: null polynomial %u", i);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t lanes = (n + EPB_PER - 1) / EPB_PER;
    const size_t nblocks = (lanes + PO_THREADS - 1) / PO_THREADS;
    const uint32_t chunk_max = 16384;   // queries per launch (gridDim.y limit is 65535)
    std::vector<EvalQuery> q(std::min(count, chunk_max));
    std::vector<uint64_t> part;
    for (uint32_t first = 0; first < count; first += chunk_max) {
        const uint32_t c = std::min(chunk_max, count - first);
        const size_t qbytes = (c * sizeof(EvalQuery) + 255) & ~(size_t)255;
        int rc = ensure_scratch(ctx, qbytes + (size_t)c * nblocks * 32);
        if (rc) return rc;
        for (uint32_t i = 0; i < c; i++) {
            q[i].coeff = (const fe*)d_coeffs[first + i];
            memcpy(q[i].x.l, points + 4 * (size_t)(first + i), 32);
            uint64_t x_per[4];
            h_pow(HFR, x_per, points + 4 * (size_t)(first + i), EPB_PER);
            memcpy(q[i].x_per.l, x_per, 32);
        }
        fe* d_part = (fe*)((unsigned char*)ctx->scratch + qbytes);
        ZK_HIP(ctx, hipMemcpyAsync(ctx->scratch, q.data(), c * sizeof(EvalQuery), hipMemcpyHostToDevice, ctx->stream));
        {
            TimedRegion tr(ctx, "eval_polynomial");
            hipLaunchKernelGGL(eval_poly_batch_kernel, dim3((unsigned)nblocks, c), dim3(PO_THREADS), 0, ctx->stream,
                               (const EvalQuery*)ctx->scratch, n, d_part);
            ZK_HIP(ctx, hipGetLastError());
        }
        part.resize((size_t)c * nblocks * 4);
        ZK_HIP(ctx, hipMemcpyAsync(part.data(), d_part, (size_t)c * nblocks * 32, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t i = 0; i < c; i++) {
            uint64_t acc[4] = {0, 0, 0, 0};
            for (size_t b = 0; b < nblocks; b++) h_add(HFR, acc, acc, &part[4 * ((size_t)i * nblocks + b)]);
            memcpy(out + 4 * (size_t)(first + i), acc, 32);
        }
    }
    return ZKHIP_OK;
}

int zkhip_poly_lincomb_dev(zkhip_ctx* ctx, const uint64_t* const* d_polys, const size_t* lens, const uint64_t* coeffs,
                           uint32_t num, size_t n, uint64_t* d_out) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, (d_polys && lens && coeffs) || num == 0, "poly_lincomb: null table");
    ZK_CHECK(ctx, d_out || n == 0, "poly_lincomb: null output");
    if (n == 0) return ZKHIP_OK;
    for (uint32_t k = 0; k < num; k++) {
        ZK_CHECK(ctx, d_polys[k] || lens[k] == 0, "poly_lincomb: null polynomial %u", k);
        ZK_CHECK(ctx, lens[k] <= n, "poly_lincomb: polynomial %u has %zu coefficients, output has %zu", k, lens[k], n);
    }
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_scratch(ctx, ((size_t)num + 1) * sizeof(LincombTerm));
    if (rc) return rc;
    std::vector<LincombTerm> terms(num);
    for (uint32_t k = 0; k < num; k++) {
        terms[k].poly = (const fe*)d_polys[k];
        terms[k].len = lens[k];
        memcpy(terms[k].coeff, coeffs + 4 * (size_t)k, 32);
    }
    if (num) {
        ZK_HIP(ctx, hipMemcpyAsync(ctx->scratch, terms.data(), num * sizeof(LincombTerm), hipMemcpyHostToDevice, ctx->stream));
        // the table is pageable host memory that dies with this call
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    TimedRegion tr(ctx, "poly_lincomb");
    const size_t lanes = (n + LC_PER - 1) / LC_PER;
    hipLaunchKernelGGL(poly_lincomb_kernel, dim3((unsigned)((lanes + PO_THREADS - 1) / PO_THREADS)), dim3(PO_THREADS), 0,
                       ctx->stream, (const LincombTerm*)ctx->scratch, num, n, (fe*)d_out);
    ZK_HIP(ctx, hipGetLastError());
    return ZKHIP_OK;
}

int zkhip_kate_division_dev(zkhip_ctx* ctx, const uint64_t* d_poly, size_t n, const uint64_t point[4], uint64_t* d_out) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, n >= 1, "kate_division: empty polynomial");
    ZK_CHECK(ctx, d_poly && point && (d_out || n == 1), "kate_division: null pointer");
    if (n == 1) return ZKHIP_OK;   // the quotient of a constant has no coefficients
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nblocks = (n + KD_BLOCK - 1) / KD_BLOCK;
    // scratch: [powers 272][lane_in nblocks * PO_THREADS][block totals nblocks]
    const size_t npw = 16 + PO_THREADS;
    int rc = ensure_scratch(ctx, (npw + nblocks * PO_THREADS + nblocks) * 32);
    if (rc) return rc;
    std::vector<uint64_t> pw(npw * 4);
    uint64_t zper[4], zblock[4];
    h_pow(HFR, zper, point, KD_PER);
    h_pow(HFR, zblock, point, KD_BLOCK);
    memcpy(&pw[0], zper, 32);
    for (int d = 1; d < 8; d++) h_mul(HFR, &pw[4 * d], &pw[4 * (d - 1)], &pw[4 * (d - 1)]);
    memcpy(&pw[4 * 8], zblock, 32);
    for (int d = 1; d < 8; d++) h_mul(HFR, &pw[4 * (8 + d)], &pw[4 * (8 + d - 1)], &pw[4 * (8 + d - 1)]);
    memcpy(&pw[4 * 16], HFR.r, 32);
    for (int t = 1; t < PO_THREADS; t++) h_mul(HFR, &pw[4 * (16 + t)], &pw[4 * (16 + t - 1)], zper);
    fe* d_pw = (fe*)ctx->scratch;
    fe* d_lane_in = d_pw + npw;
    fe* d_tot = d_lane_in + nblocks * PO_THREADS;
    ZK_HIP(ctx, hipMemcpyAsync(d_pw, pw.data(), npw * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    TimedRegion tr(ctx, "kate_division");
    const fe z = fe_from(point);
    hipLaunchKernelGGL(kate_local_kernel, dim3((unsigned)nblocks), dim3(PO_THREADS), 0, ctx->stream, (const fe*)d_poly, n, z, d_pw,
                       d_lane_in, d_tot);
    hipLaunchKernelGGL(kate_blocks_kernel, dim3(1), dim3(PO_THREADS), 0, ctx->stream, d_tot, nblocks, d_pw);
    hipLaunchKernelGGL(kate_apply_kernel, dim3((unsigned)nblocks), dim3(PO_THREADS), 0, ctx->stream, (const fe*)d_poly, n, z, d_pw,
                       d_lane_in, d_tot, (fe*)d_out);
    ZK_HIP(ctx, hipGetLastError());
    return ZKHIP_OK;
}

int zkhip_permutation_sigma_dev(zkhip_domain* dom, const uint32_t* d_map_col, const uint32_t* d_map_row, uint32_t num_cols,
                                uint64_t* d_out) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_map_col && d_map_row && d_out, "permutation_sigma: null pointer");
    ZK_CHECK(ctx, num_cols >= 1 && num_cols <= 4096, "permutation_sigma: %u columns", num_cols);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << dom->k;
    int rc = ensure_scratch(ctx, ((size_t)num_cols + 1) * 32);
    if (rc) return rc;
    uint32_t* d_bad = (uint32_t*)((fe*)ctx->scratch + num_cols);
    ZK_HIP(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    std::vector<uint64_t> dpow((size_t)num_cols * 4);
    memcpy(&dpow[0], HFR.r, 32);
    uint64_t delta[4];
    memcpy(delta, FrParams::DELTA, 32);
    for (uint32_t c = 1; c < num_cols; c++) h_mul(HFR, &dpow[4 * c], &dpow[4 * (c - 1)], delta);
    ZK_HIP(ctx, hipMemcpyAsync(ctx->scratch, dpow.data(), dpow.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the table is pageable host memory that dies with this call
    hipLaunchKernelGGL(perm_sigma_kernel, dim3((unsigned)((n + PO_THREADS - 1) / PO_THREADS)), dim3(PO_THREADS), 0, ctx->stream,
                       d_map_col, d_map_row, n, num_cols, (const fe*)ctx->scratch, (const fe*)dom->plan_fwd->d_lo,
                       (const fe*)dom->plan_fwd->d_hi, dom->plan_fwd->h, (fe*)d_out, d_bad);
    ZK_HIP(ctx, hipGetLastError());
    // the kernel reads the table from the scratch buffer: finish before another call may overwrite it
    uint32_t h_bad = 0;
    ZK_HIP(ctx, hipMemcpyAsync(&h_bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ZK_CHECK(ctx, !h_bad, "permutation_sigma: a mapping entry points outside the %u columns x %zu rows", num_cols, n);
    return ZKHIP_OK;
}

}  // extern "C"
