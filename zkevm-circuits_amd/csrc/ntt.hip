// Radix-2^r Stockham NTT over BN254 Fr for gfx950.
//
// A transform of N = 2^L points runs as ceil(L / 8) out-of-place passes.  Pass p has radix R = 2^r and
// Ns = product of the earlier radices; workgroup tiles are R x J elements (R*J = 1024 elements):
//   load   x[t][c] = in[(j0 + c) + t * N/R] * w_N^(t * ((j0+c) mod Ns) * N/(Ns R))      (J*32 B segments)
//   R-point DIF butterflies on every column in LDS, twiddles w_R^k from an LDS table
//   store  out[(j / Ns) * Ns * R + (j mod Ns) + q * Ns] = X[q]                           (natural order)
// so input and output of the whole transform are in natural order, like `best_fft`.
// Fused into the first load: zero padding beyond n_in and the zeta-coset scaling of `coeff_to_extended`;
// fused into the last store: the 1/n (and inverse zeta) scaling and the truncation of `extended_to_coeff`.
// Powers of omega come from two L2-resident tables (w^i for i < 2^h, w^(i 2^h)) and one multiplication.
//
// Transforms of 2^11 points and more use `ntt_tile_kernel` (below); the LDS-per-stage kernel above it serves the
// small ones.  The tile kernel keeps the same pass / tile / index structure and changes the inside of a pass:
//   * arithmetic in the lazy signed-digit form of wide.hip.h: a butterfly is one Montgomery product and 18 digit
//     additions, elements are carry-normalised once per LDS round trip, and nothing is compared against p;
//   * decimation in time on bit-reversed tile rows, two stages at a time in registers: a lane owns four elements
//     of one column, so a radix-2^7 pass crosses LDS four times instead of seven;
//   * the twiddle w_N^(t (j mod Ns) N/(Ns R)) that the NEXT pass would apply while loading is applied by THIS pass
//     while storing.  That product also pulls the value back into (-p/100, 1.01 p), so adding p gives a non-negative
//     residue below 2^256 that is stored as is; only the last pass canonicalises.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "ntt.h"
#include "wide.hip.h"

namespace zkhip {

constexpr int NTT_THREADS = 256;
constexpr int TILE_LOG = 10;


struct NttPassArgs {
    const fe* in;
    fe* out;
    const fe* lo;   // w^i, i < 2^h
    const fe* hi;   // w^(i * 2^h)
    uint32_t h;
    uint32_t logN, r, logNs, logJ;
    uint32_t n_in;     // inputs at index >= n_in read as zero
    uint32_t n_out;    // outputs at index >= n_out are not stored
    uint32_t pre_mode;   // 0 none, 1: multiply input i by pre[i % 3]
    uint32_t post_mode;  // 0 none, 1: multiply output by post[0], 2: multiply output i by post[i % 3]
    fe pre[3];
    fe post[3];
    uint32_t split_out;   // last pass: result i is stored at (i & 1) * N/2 + (i >> 1)
    size_t in_stride, out_stride;   // elements between consecutive batch members
};

ZK_DEV fe omega_pow(const NttPassArgs& a, uint32_t e) {
    const fe l = load_fe(a.lo + (e & ((1u << a.h) - 1)));
    const uint32_t eh = e >> a.h;
    if (eh == 0) return l;
    return Fr::mul(l, load_fe(a.hi + eh));
}

ZK_DEV uint32_t bitrev(uint32_t v, uint32_t bits) { return bits ? (__brev(v) >> (32 - bits)) : 0; }

// The butterfly products as single dependent multiply-add chains per column (see Wide::chain): 2^19 -> 2^21
// 0.289 -> 0.280 ms, 64 columns 0.2325 -> 0.2295 ms each when it was introduced.
#ifndef ZKHIP_NTT_CHAIN_MUL
#define ZKHIP_NTT_CHAIN_MUL 1
#endif
using NttW = Wide<FrParams, ZKHIP_NTT_CHAIN_MUL != 0>;

__global__ void __launch_bounds__(NTT_THREADS) ntt_pass_kernel(const NttPassArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    fe* x = reinterpret_cast<fe*>(smem_raw);                 // [R][J]
    const uint32_t R = 1u << a.r, J = 1u << a.logJ, N = 1u << a.logN;
    fe* tw = x + (R << a.logJ);                              // [R/2]
    const uint32_t tid = threadIdx.x;
    const uint32_t j0 = blockIdx.x << a.logJ;
    const fe* in = a.in + (size_t)blockIdx.y * a.in_stride;
    fe* out = a.out + (size_t)blockIdx.y * a.out_stride;
    const uint32_t rowStride = N >> a.r;                      // N / R
    const uint32_t nsMask = (1u << a.logNs) - 1;
    const uint32_t twShift = a.logN - a.logNs - a.r;          // log2(N / (Ns R))
    const uint32_t tile = R << a.logJ;

    // tile-local twiddles w_R^k = w_N^(k N/R)
    for (uint32_t k = tid; k < (R >> 1); k += NTT_THREADS) x[tile + k] = omega_pow(a, k << (a.logN - a.r));

    for (uint32_t idx = tid; idx < tile; idx += NTT_THREADS) {
        const uint32_t c = idx & (J - 1), t = idx >> a.logJ;
        const uint32_t j = j0 + c;
        const uint32_t gi = j + t * rowStride;
        fe v;
        if (gi < a.n_in) {
            v = load_fe(in + gi);
            if (a.pre_mode == 1) {
                const uint32_t m = gi % 3;
                if (m) v = Fr::mul(v, a.pre[m]);
            }
            if (a.logNs) {
                const uint32_t e = (t * (j & nsMask)) << twShift;
                if (e) v = Fr::mul(v, omega_pow(a, e));
            }
        } else {
            v = Fr::zero();
        }
        x[idx] = v;
    }
    __syncthreads();

    const uint32_t nbf = tile >> 1;
    for (uint32_t s = 0; s < a.r; s++) {
        const uint32_t logHalf = a.r - 1 - s, half = 1u << logHalf;
        for (uint32_t bf = tid; bf < nbf; bf += NTT_THREADS) {
            const uint32_t c = bf & (J - 1), p = bf >> a.logJ;
            const uint32_t i = p & (half - 1), blk = p >> logHalf;
            const uint32_t row0 = (blk << (logHalf + 1)) + i, row1 = row0 + half;
            const fe u = x[(row0 << a.logJ) + c], v = x[(row1 << a.logJ) + c];
            x[(row0 << a.logJ) + c] = Fr::add(u, v);
            fe d = Fr::sub(u, v);
            const uint32_t k = i << s;
            if (k) d = Fr::mul(d, tw[k]);
            x[(row1 << a.logJ) + c] = d;
        }
        __syncthreads();
    }

    // store: X[q] sits in LDS row bitrev(q)
    for (uint32_t idx = tid; idx < tile; idx += NTT_THREADS) {
        uint32_t c, q;
        if (a.logNs == 0) { q = idx & (R - 1); c = idx >> a.r; }     // outputs of one j are contiguous
        else { c = idx & (J - 1); q = idx >> a.logJ; }
        const uint32_t j = j0 + c;
        const uint32_t oi = ((j >> a.logNs) << (a.logNs + a.r)) + (j & nsMask) + (q << a.logNs);
        if (oi < a.n_out) {
            fe v = x[(bitrev(q, a.r) << a.logJ) + c];
            if (a.post_mode == 1) v = Fr::mul(v, a.post[0]);
            else if (a.post_mode == 2) v = Fr::mul(v, a.post[oi % 3]);
            store_fe(out + (a.split_out ? ((oi & 1) << (a.logN - 1)) + (oi >> 1) : oi), v);
        }
    }
}

// ---- tile kernel -------------------------------------------------------------------------------------
constexpr uint32_t TILE = 1u << TILE_LOG;
// Elements per lane.  Four (two butterfly stages per LDS round trip) instead of eight (three stages) costs one more
// round trip per radix-2^7 pass and buys twice the waves for the same LDS footprint at 98 VGPRs.  Measured equal for a
// batch (0.24 ms per 2^21 column) and slightly ahead for a lone transform, whose workgroups start in lockstep and
// therefore overlap their load / compute / store phases poorly whatever the occupancy.
constexpr int TK_LOGE = 2;
constexpr int TK_E = 1 << TK_LOGE;
constexpr int TK_THREADS = (1 << TILE_LOG) / TK_E;
constexpr int TK_MAX_ROUNDS = 4;
constexpr uint32_t NTT_BOUNDARY_TABLE_MAX_LOG = 22;   // 128 MiB

// Largest pass boundary (log2 of its twiddle count) that gets a single-lookup table; larger ones compute
// w^e = lo[e mod 2^h] * hi[e >> h].  ZKHIP_NTT_BOUNDARY_TABLE_MAX_LOG overrides it (measurements).
static uint32_t boundary_table_max_log() {
    const char* e = getenv("ZKHIP_NTT_BOUNDARY_TABLE_MAX_LOG");
    return e && *e ? (uint32_t)atoi(e) : NTT_BOUNDARY_TABLE_MAX_LOG;
}

struct NttTileArgs {
    const fe* in;
    fe* out;
    const fe* lo;      // w^i (i < 2^h) and w^(i 2^h), canonical, 2^261 Montgomery domain
    const fe* hi;
    uint32_t h;
    uint32_t logN, r, logNs, logJ;
    uint32_t n_in, n_out;
    uint32_t pre_mode, post_mode;
    fe pre[3], post[3];           // 2^261 domain
    uint32_t last;                // final pass: canonical output (after the optional post scaling)
    uint32_t r_next;              // radix of the next pass (for the twiddle applied on store)
    const fe* b0;                 // non-null: single-lookup twiddles of this pass boundary, indexed by t * (j mod Ns)
    uint32_t zskip;               // first pass of a zero-padded transform: n_in = N >> zskip (0 otherwise)
    uint32_t split_out;           // last pass: result i is stored at (i & 1) * N/2 + (i >> 1)
    uint32_t nrounds;
    uint32_t pb[TK_MAX_ROUNDS][TK_LOGE];   // round, local bit i -> tile-row bit
    uint32_t sb[TK_MAX_ROUNDS][TK_LOGE];   // the same bits sorted ascending
    uint32_t nst[TK_MAX_ROUNDS];           // butterfly stages of the round (local bits 0 .. nst-1)
    size_t in_stride, out_stride;
};

// w^e in wide form (normalised, value in (-p/100, 1.01 p))
__device__ __forceinline__ fw omega_pow_w(const NttTileArgs& a, uint32_t e) {
    fw l = FrW::from_fe(load_fe(a.lo + (e & ((1u << a.h) - 1))));
    const uint32_t eh = e >> a.h;
    if (eh) l = FrW::mul(l, FrW::from_fe(load_fe(a.hi + eh)));
    return l;
}
// digits 0..7 into [-2^28, 2^28): halves the column sums of a product, so the other factor may carry three
// stages of un-normalised butterfly sums
__device__ __forceinline__ fw balance_digits(const fw& v) {
    fw o;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int32_t t = v.d[i] + c;
        c = (t + (1 << 28)) >> 29;
        o.d[i] = t - (c << 29);
    }
    o.d[8] = v.d[8] + c;
    return o;
}
__device__ __forceinline__ uint32_t spread_bits(uint32_t g, const uint32_t sb[TK_LOGE]) {
#pragma unroll
    for (int i = 0; i < TK_LOGE; i++) g = ((g >> sb[i]) << (sb[i] + 1)) | (g & ((1u << sb[i]) - 1));
    return g;
}
// Word index of tile element (row p, column c) inside one digit plane.  Rows are J words long, so without help a
// wave that walks rows in steps of the register-group size or reads one column down the rows (the transposed
// store of the first pass) lands in a quarter or a single one of the 64 banks.  Within every aligned 64-word line the
// index is XORed with bits of the line number: the high bits of the swizzle move the lane groups of one wave to
// different bank quarters, the low bits permute the columns differently in each line.  A bijection inside each
// line, so the plane stays dense.
__device__ __forceinline__ uint32_t tile_at(uint32_t p, uint32_t c, uint32_t logJ) {
    const uint32_t w = (p << logJ) + c, u = w >> 6;
    uint32_t swz;
    if (logJ >= 6) {
        swz = u & 63;
    } else {
        const uint32_t nb = 6 - logJ;                        // row bits inside the bank index
        swz = (((u >> (3 - nb)) & ((1u << nb) - 1)) << logJ) | (u & ((1u << logJ) - 1));
    }
    return (u << 6) | ((w & 63) ^ swz);
}
__device__ __forceinline__ fw lds_get(const int32_t* x, uint32_t at) {
    fw v;
#pragma unroll
    for (int d = 0; d < 9; d++) v.d[d] = x[d * TILE + at];
    return v;
}
__device__ __forceinline__ void lds_put(int32_t* x, uint32_t at, const fw& v) {
#pragma unroll
    for (int d = 0; d < 9; d++) x[d * TILE + at] = v.d[d];
}
__device__ __forceinline__ fw tw_get(const int32_t* tw, uint32_t half_r, uint32_t k) {
    fw v;
#pragma unroll
    for (int d = 0; d < 9; d++) v.d[d] = tw[d * half_r + k];
    return v;
}

// Butterfly stages of one round on the TK_E elements of a lane.  e[m] sits in tile row pos[m]; local bit i is tile-row
// bit TK_LOGE * round + i.  FIRST: the round of the lowest stages, whose twiddle indices depend on m only.
// zskip: stages below it see a zero partner (zero-padded input: tile rows >= R >> zskip are zero, i.e. every position
// with one of its low zskip bits set), so the butterfly degenerates to a copy.
template <bool FIRST>
__device__ __forceinline__ void tile_stages(fw e[TK_E], const uint32_t pos[TK_E], uint32_t round, uint32_t nst, uint32_t r,
                                            uint32_t zskip, const int32_t* tw) {
    const uint32_t half_r = 1u << (r - 1);
#pragma unroll
    for (int i = 0; i < TK_LOGE; i++) {
        if ((uint32_t)i >= nst) break;
        const uint32_t s = TK_LOGE * round + i;
#pragma unroll
        for (int m = 0; m < TK_E; m++) {
            if (m & (1 << i)) continue;
            const int mb = m + (1 << i);
            if (s < zskip) {
                e[mb] = e[m];
                continue;
            }
            fw t;
            if (FIRST) {
                const uint32_t km = (uint32_t)m & ((1u << i) - 1);   // compile-time
                if (km == 0) t = (i == 0) ? e[mb] : FrW::norm(e[mb]);
                else t = NttW::mul(e[mb], tw_get(tw, half_r, km << (r - 1 - s)));
            } else {
                const uint32_t k = (pos[m] & ((1u << s) - 1)) << (r - 1 - s);
                t = NttW::mul(e[mb], tw_get(tw, half_r, k));
            }
            e[mb] = FrW::sub(e[m], t);
            e[m] = FrW::add(e[m], t);
        }
    }
}

__global__ void __launch_bounds__(TK_THREADS) ntt_tile_kernel(const NttTileArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    int32_t* x = reinterpret_cast<int32_t*>(smem_raw);       // [9][R][J]
    const uint32_t R = 1u << a.r, J = 1u << a.logJ, N = 1u << a.logN;
    int32_t* tw = x + 9 * TILE;                               // [9][R/2]
    const uint32_t tid = threadIdx.x;
    const uint32_t j0 = blockIdx.x << a.logJ;
    const fe* in = a.in + (size_t)blockIdx.y * a.in_stride;
    fe* out = a.out + (size_t)blockIdx.y * a.out_stride;
    const uint32_t rowStride = N >> a.r;
    const uint32_t c = tid & (J - 1), grp = tid >> a.logJ;   // R / TK_E groups of TK_E rows per column
    const uint32_t j = j0 + c;

    // tile-local twiddles w_R^k = w_N^(k N/R), balanced digits
    for (uint32_t k = tid; k < (R >> 1); k += TK_THREADS) {
        const fw w = balance_digits(omega_pow_w(a, k << (a.logN - a.r)));
#pragma unroll
        for (int d = 0; d < 9; d++) tw[d * (R >> 1) + k] = w.d[d];
    }

    fw e[TK_E];
    uint32_t pos[TK_E];
    // round 0: rows come straight from memory; row p of the tile holds input row bitrev(p)
    {
        const uint32_t base = spread_bits(grp, a.sb[0]);
#pragma unroll
        for (int m = 0; m < TK_E; m++) {
            pos[m] = base | (uint32_t)m;                       // round 0 always owns the lowest row bits
            const uint32_t t = bitrev(pos[m], a.r);
            const uint32_t gi = j + t * rowStride;
            if (gi < a.n_in) {
                e[m] = FrW::from_fe(load_fe(in + gi));
                if (a.pre_mode == 1) {
                    const uint32_t q = gi % 3;
                    if (q) e[m] = NttW::mul(e[m], FrW::from_fe(a.pre[q]));
                }
            } else {
                e[m] = FrW::zero();
            }
        }
    }
    __syncthreads();   // twiddle table
    tile_stages<true>(e, pos, 0, a.nst[0], a.r, a.zskip, tw);
#pragma unroll
    for (int m = 0; m < TK_E; m++) lds_put(x, tile_at(pos[m], c, a.logJ), e[m]);
    __syncthreads();
    // (constant round indices: indexing the by-value argument block with a loop variable would move it to scratch)
#pragma unroll
    for (int rd = 1; rd < TK_MAX_ROUNDS; rd++) {
        if ((uint32_t)rd >= a.nrounds) break;
        const uint32_t base = spread_bits(grp, a.sb[rd]);
        uint32_t at[TK_E];
#pragma unroll
        for (int m = 0; m < TK_E; m++) {
            pos[m] = base;
#pragma unroll
            for (int i = 0; i < TK_LOGE; i++) pos[m] |= (uint32_t)((m >> i) & 1) << a.pb[rd][i];
            at[m] = tile_at(pos[m], c, a.logJ);
            e[m] = FrW::norm(lds_get(x, at[m]));
        }
        tile_stages<false>(e, pos, rd, a.nst[rd], a.r, a.zskip, tw);
#pragma unroll
        for (int m = 0; m < TK_E; m++) lds_put(x, at[m], e[m]);
        __syncthreads();
    }

    // store: X[q] sits in tile row q
    const uint32_t nsMask = (1u << a.logNs) - 1;
    const uint32_t logNs2 = a.logNs + a.r;                       // the next pass sees Ns * R
    const uint32_t rowShift2 = a.logN - a.r_next;                // log2 of its row stride
    const uint32_t twShift2 = a.last ? 0 : a.logN - logNs2 - a.r_next;
    auto place = [&](uint32_t idx, uint32_t& q, uint32_t& cc) -> uint32_t {
        if (a.logNs == 0) { q = idx & (R - 1); cc = idx >> a.r; }     // outputs of one j are contiguous
        else { cc = idx & (J - 1); q = idx >> a.logJ; }
        const uint32_t jj = j0 + cc;
        return ((jj >> a.logNs) << (a.logNs + a.r)) + (jj & nsMask) + (q << a.logNs);
    };
    if (a.last) {
        for (uint32_t idx = tid; idx < TILE; idx += TK_THREADS) {
            uint32_t cc, q;
            const uint32_t oi = place(idx, q, cc);
            if (oi >= a.n_out) continue;
            fw v = FrW::norm(lds_get(x, tile_at(q, cc, a.logJ)));
            if (a.post_mode == 1) v = NttW::mul(v, FrW::from_fe(a.post[0]));
            else if (a.post_mode == 2) v = NttW::mul(v, FrW::from_fe(a.post[oi % 3]));
            store_fe(out + (a.split_out ? ((oi & 1) << (a.logN - 1)) + (oi >> 1) : oi), FrW::to_fe(v));
        }
        return;
    }
    // not the last pass (every output is stored): multiply by what the next pass would apply to element oi while
    // loading it (t2 = its tile row, j2 = its column).  The twiddle reads of the next element are issued before this
    // element's products, so their latency is covered even with two waves per SIMD.
    auto twiddle_reads = [&](uint32_t oi, fe& ra, fe& rb) -> bool {
        const uint32_t t2 = oi >> rowShift2, j2 = oi & ((1u << rowShift2) - 1);
        const uint32_t prod = t2 * (j2 & ((1u << logNs2) - 1));
        if (a.b0) {
            ra = load_fe(a.b0 + prod);
            return false;
        }
        const uint32_t ex = prod << twShift2;
        ra = load_fe(a.lo + (ex & ((1u << a.h) - 1)));
        if (ex >> a.h) rb = load_fe(a.hi + (ex >> a.h));
        return (ex >> a.h) != 0;
    };
    fw pp;
#pragma unroll
    for (int d = 0; d < 9; d++) pp.d[d] = (int32_t)FrParams::P29[d];
    uint32_t q, cc;
    uint32_t oi = place(tid, q, cc);
    fe ra, rb;
    bool two = twiddle_reads(oi, ra, rb);
    for (uint32_t idx = tid; idx < TILE; idx += TK_THREADS) {
        const uint32_t cur_oi = oi, cur_q = q, cur_cc = cc;
        const fe ca = ra, cb = rb;
        const bool cur_two = two;
        if (idx + TK_THREADS < TILE) {
            oi = place(idx + TK_THREADS, q, cc);
            two = twiddle_reads(oi, ra, rb);
        }
        fw v = FrW::norm(lds_get(x, tile_at(cur_q, cur_cc, a.logJ)));
        v = NttW::mul(v, FrW::from_fe(ca));
        if (cur_two) v = NttW::mul(v, FrW::from_fe(cb));
        // v in (-p/100, 1.01 p): v + p is a non-negative residue below 2^256
        store_fe(out + cur_oi, FrW::pack_nonneg(FrW::norm(FrW::add(v, pp))));
    }
}

// out[i] = in[i] * scale[i % period]  (divide_by_vanishing_poly and friends)
__global__ void __launch_bounds__(256) scale_periodic_kernel(fe* a, const fe* scale, uint32_t period_mask, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_fe(a + i, Fr::mul(load_fe(a + i), load_fe(scale + (i & period_mask))));
}

// ------------------------------------------------------------------ host side ------------------------
NttPlan::~NttPlan() {
    if (d_lo) (void)hipFree(d_lo);
    if (d_hi) (void)hipFree(d_hi);
    if (d_lo_w) (void)hipFree(d_lo_w);
    if (d_hi_w) (void)hipFree(d_hi_w);
    for (void* t : d_bt)
        if (t) (void)hipFree(t);
}

int ntt_plan_create(zkhip_ctx* ctx, uint32_t log_n, const uint64_t omega[4], NttPlan** out) {
    ZK_CHECK(ctx, log_n >= 1 && log_n <= 28, "ntt: log_n %u outside [1, 28]", log_n);
    // omega must have order exactly 2^log_n
    uint64_t t[4];
    memcpy(t, omega, 32);
    for (uint32_t i = 0; i + 1 < log_n; i++) h_mul(HFR, t, t, t);
    uint64_t minus_one[4], zero[4] = {0, 0, 0, 0};
    h_sub(HFR, minus_one, zero, HFR.r);
    ZK_CHECK(ctx, h_eq(t, minus_one), "ntt: omega does not have order 2^%u", log_n);

    auto* plan = new NttPlan();
    plan->log_n = log_n;
    memcpy(plan->omega, omega, 32);
    plan->h = (log_n + 1) / 2;
    const size_t nlo = (size_t)1 << plan->h, nhi = (size_t)1 << (log_n - plan->h);
    std::vector<uint64_t> lo(nlo * 4), hi(nhi * 4);
    memcpy(&lo[0], HFR.r, 32);
    for (size_t i = 1; i < nlo; i++) h_mul(HFR, &lo[4 * i], &lo[4 * (i - 1)], omega);
    uint64_t step[4];
    h_mul(HFR, step, &lo[4 * (nlo - 1)], omega);   // omega^(2^h)
    memcpy(&hi[0], HFR.r, 32);
    for (size_t i = 1; i < nhi; i++) h_mul(HFR, &hi[4 * i], &hi[4 * (i - 1)], step);
    hipError_t e = hipMalloc(&plan->d_lo, nlo * 32);
    if (e == hipSuccess) e = hipMalloc(&plan->d_hi, nhi * 32);
    if (e == hipSuccess) e = hipMemcpy(plan->d_lo, lo.data(), nlo * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(plan->d_hi, hi.data(), nhi * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess && log_n >= TILE_LOG) {
        // the same tables in the 2^261 Montgomery domain for the tile kernel (x R -> 32 x R)
        uint64_t k32[4];
        memcpy(k32, FrParams::TWO5, 32);
        for (size_t i = 0; i < nlo; i++) h_mul(HFR, &lo[4 * i], &lo[4 * i], k32);
        for (size_t i = 0; i < nhi; i++) h_mul(HFR, &hi[4 * i], &hi[4 * i], k32);
        e = hipMalloc(&plan->d_lo_w, nlo * 32);
        if (e == hipSuccess) e = hipMalloc(&plan->d_hi_w, nhi * 32);
        if (e == hipSuccess) e = hipMemcpy(plan->d_lo_w, lo.data(), nlo * 32, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(plan->d_hi_w, hi.data(), nhi * 32, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        delete plan;
        return set_error(ctx, ZKHIP_ERR_HIP, "ntt plan upload failed: %s", hipGetErrorString(e));
    }
    // split log_n into the fewest passes of radix <= 2^8, as even as possible
    const uint32_t npass = (log_n + 7) / 8;
    uint32_t left = log_n;
    for (uint32_t p = 0; p < npass; p++) {
        const uint32_t r = (left + (npass - p) - 1) / (npass - p);
        plan->radix.push_back(r);
        left -= r;
    }
    if (log_n >= TILE_LOG) {
        uint64_t k32[4];
        memcpy(k32, FrParams::TWO5, 32);
        uint32_t bits = plan->radix[0];
        for (uint32_t p = 0; p + 1 < npass; p++) {
            bits += plan->radix[p + 1];
            plan->d_bt.push_back(nullptr);
            if (bits > boundary_table_max_log()) continue;
            const size_t cnt = (size_t)1 << bits;
            std::vector<uint64_t> bt(cnt * 4);
            uint64_t step[4];
            h_pow(HFR, step, omega, (uint64_t)1 << (log_n - bits));
            h_mul(HFR, &bt[0], HFR.r, k32);                       // one, 2^261 domain
            for (size_t i = 1; i < cnt; i++) h_mul(HFR, &bt[4 * i], &bt[4 * (i - 1)], step);
            hipError_t e2 = hipMalloc(&plan->d_bt[p], cnt * 32);
            if (e2 == hipSuccess) e2 = hipMemcpy(plan->d_bt[p], bt.data(), cnt * 32, hipMemcpyHostToDevice);
            if (e2 != hipSuccess) {
                delete plan;
                return set_error(ctx, ZKHIP_ERR_HIP, "ntt plan upload failed: %s", hipGetErrorString(e2));
            }
        }
    }
    *out = plan;
    return ZKHIP_OK;
}

int ntt_init(zkhip_ctx* ctx) {
    ZK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_pass_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    ZK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_tile_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    return ZKHIP_OK;
}

// Runs the transform.  `src` holds n_in valid elements per batch member (stride src_stride); the result
// (first n_out elements) lands in `dst` (stride dst_stride).  src may equal dst.  tmp0/tmp1: scratch of
// batch * 2^log_n elements each (tmp1 only needed when the pass count parity requires it).
int ntt_run(zkhip_ctx* ctx, const NttPlan* plan, const fe* src, size_t src_stride, uint32_t n_in, fe* dst,
            size_t dst_stride, uint32_t n_out, size_t batch, const NttFuse& fuse, fe* tmp0, fe* tmp1) {
    const uint32_t L = plan->log_n;
    const size_t N = (size_t)1 << L;
    const uint32_t npass = (uint32_t)plan->radix.size();
    ZK_CHECK(ctx, batch >= 1 && batch <= 65535, "ntt: batch %zu outside [1, 65535]", batch);
    ZK_CHECK(ctx, !fuse.split_out || (n_out == N && L >= 1 && src != dst), "ntt: the split output needs all N results, out of place");
    // ping-pong so that the last pass writes dst and no pass runs in place
    // sequence of buffers: src -> b1 -> b2 -> ... -> dst
    std::vector<fe*> bufs(npass + 1);
    std::vector<size_t> strides(npass + 1);
    bufs[0] = const_cast<fe*>(src);
    strides[0] = src_stride;
    bufs[npass] = dst;
    strides[npass] = dst_stride;
    // intermediate buffers alternate between tmp0 and `alt`, walking back from dst; dst itself serves as
    // `alt` when it is a distinct full-size buffer, which saves the second scratch buffer
    fe* alt = (src != dst && dst_stride == N && n_out == N) ? dst : tmp1;
    for (uint32_t p = npass - 1; p >= 1; p--) {
        bufs[p] = ((npass - p) & 1) ? tmp0 : alt;
        strides[p] = N;
    }
    if (npass >= 2 && bufs[1] == src) return set_error(ctx, ZKHIP_ERR_INVALID, "ntt: scratch aliases source");
    uint32_t logNs = 0;
    TimedRegion tr(ctx, "ntt");
    if (L >= TILE_LOG) {
        // tile kernel: every pass has radix 2^3 .. 2^8 and a full tile of 2^TILE_LOG elements
        uint64_t k32[4];
        memcpy(k32, FrParams::TWO5, 32);
        for (uint32_t p = 0; p < npass; p++) {
            NttTileArgs a;
            memset(&a, 0, sizeof a);
            a.in = bufs[p];
            a.out = bufs[p + 1];
            a.in_stride = strides[p];
            a.out_stride = strides[p + 1];
            a.lo = (const fe*)plan->d_lo_w;
            a.hi = (const fe*)plan->d_hi_w;
            a.h = plan->h;
            a.logN = L;
            a.r = plan->radix[p];
            a.logNs = logNs;
            a.logJ = TILE_LOG - a.r;
            a.n_in = (p == 0) ? n_in : (uint32_t)N;
            a.n_out = (p == npass - 1) ? n_out : (uint32_t)N;
            a.pre_mode = (p == 0) ? fuse.pre_mode : 0;
            a.post_mode = (p == npass - 1) ? fuse.post_mode : 0;
            for (int q = 0; q < 3; q++) {
                h_mul(HFR, (uint64_t*)a.pre[q].l, (const uint64_t*)fuse.pre[q].l, k32);
                h_mul(HFR, (uint64_t*)a.post[q].l, (const uint64_t*)fuse.post[q].l, k32);
            }
            a.last = (p == npass - 1);
            a.split_out = (a.last && fuse.split_out) ? 1 : 0;
            a.zskip = 0;
            if (p == 0 && n_in < N && n_in > 0 && (n_in & (n_in - 1)) == 0) {
                uint32_t z = 0;
                while (((size_t)n_in << z) < N) z++;
                a.zskip = std::min(z, a.r);
            }
            a.r_next = a.last ? 0 : plan->radix[p + 1];
            a.b0 = a.last ? nullptr : (const fe*)plan->d_bt[p];
            ZK_CHECK(ctx, a.r >= 3 && a.r <= 8, "ntt: tile pass radix 2^%u unsupported", a.r);
            a.nrounds = (a.r + TK_LOGE - 1) / TK_LOGE;
            for (uint32_t rd = 0; rd < a.nrounds; rd++) {
                const uint32_t nst = std::min<uint32_t>(TK_LOGE, a.r - TK_LOGE * rd);
                a.nst[rd] = nst;
                uint32_t filler = 0;   // lowest tile-row bits ride along when a round has fewer stages than local bits
                uint32_t sorted[TK_LOGE];
                for (uint32_t i = 0; i < TK_LOGE; i++) sorted[i] = a.pb[rd][i] = (i < nst) ? TK_LOGE * rd + i : filler++;
                std::sort(sorted, sorted + TK_LOGE);
                memcpy(a.sb[rd], sorted, sizeof sorted);
            }
            const size_t lds = (9 * (size_t)TILE + 9 * ((size_t)1 << (a.r - 1))) * sizeof(int32_t);
            dim3 grid((uint32_t)(N >> TILE_LOG), (uint32_t)batch);
            hipLaunchKernelGGL(ntt_tile_kernel, grid, dim3(TK_THREADS), lds, ctx->stream, a);
            ZK_HIP(ctx, hipGetLastError());
            logNs += a.r;
        }
        return ZKHIP_OK;
    }
    for (uint32_t p = 0; p < npass; p++) {
        NttPassArgs a;
        a.in = bufs[p];
        a.out = bufs[p + 1];
        a.in_stride = strides[p];
        a.out_stride = strides[p + 1];
        a.lo = (const fe*)plan->d_lo;
        a.hi = (const fe*)plan->d_hi;
        a.h = plan->h;
        a.logN = L;
        a.r = plan->radix[p];
        a.logNs = logNs;
        uint32_t logJ = TILE_LOG - a.r;
        if (logJ > L - a.r) logJ = L - a.r;
        a.logJ = logJ;
        a.n_in = (p == 0) ? n_in : (uint32_t)N;
        a.n_out = (p == npass - 1) ? n_out : (uint32_t)N;
        a.pre_mode = (p == 0) ? fuse.pre_mode : 0;
        a.post_mode = (p == npass - 1) ? fuse.post_mode : 0;
        memcpy(a.pre, fuse.pre, sizeof a.pre);
        memcpy(a.post, fuse.post, sizeof a.post);
        a.split_out = (p == npass - 1 && fuse.split_out) ? 1 : 0;
        const uint32_t tile = 1u << (a.r + logJ);
        const size_t lds = ((size_t)tile + (1u << a.r) / 2 + 1) * sizeof(fe);
        dim3 grid((uint32_t)(N >> (a.r + logJ)), (uint32_t)batch);
        hipLaunchKernelGGL(ntt_pass_kernel, grid, dim3(NTT_THREADS), lds, ctx->stream, a);
        ZK_HIP(ctx, hipGetLastError());
        logNs += a.r;
    }
    return ZKHIP_OK;
}

int scale_periodic(zkhip_ctx* ctx, fe* a, const fe* d_scale, uint32_t period, size_t n) {
    hipLaunchKernelGGL(scale_periodic_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, d_scale,
                       period - 1, n);
    ZK_HIP(ctx, hipGetLastError());
    return ZKHIP_OK;
}

}  // namespace zkhip
