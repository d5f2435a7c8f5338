// Pippenger bucket MSM over BN254 G1 for gfx950 (`arithmetic::best_multiexp`).
//
// Registration (once per SRS): for window width c and W = ceil(255 / c) windows the library stores the
// affine tables T_w[i] = 2^(c w) * G_i for every w (coordinates in the 2^261 Montgomery domain of the accumulate
// kernel's lazy arithmetic, csrc/wide.hip.h).  With 288 GB of HBM the W-fold copy is cheap (k = 19:
// 512 MiB) and it collapses all windows into ONE set of 2^(c-1) signed-digit buckets: no per-window bucket
// sets, no doubling ladder at the end, 16x less bucket-reduction work.
//
// Per MSM:
//   1. digits + histogram : scalars leave Montgomery form, are recoded into W signed c-bit digits; each block
//                           counts its digits per bucket in an LDS-private histogram (heavy buckets such as
//                           "digit 1" of witness-like scalars never touch a contended global atomic)
//   2. scan               : exclusive scan of the [bucket][block] count matrix -> scatter offsets
//   3. scatter            : the same blocks place (point, window, sign) entries with LDS-private cursors
//   4. accumulate         : one lane per (bucket, slot): a bucket of cnt entries is split into ceil(cnt / L) slots of
//                           equal share; L is planned per MSM so that the slots fill a whole number of rounds of
//                           the chip.  Each lane gathers its table points and folds them with XYZZ mixed additions
//   5. merge              : per bucket, two lanes fold the slot sums and join by a shuffle (heavy buckets of skewed
//                           scalars first pass through serial pre-levels of 16 sums per lane)
//   6. reduce             : sum (b + 1) S_b without scalar multiplications: per wave of 64 buckets six rounds of
//                           shuffles and additions give the total and the six "bit sums" (sum of the buckets whose
//                           index has bit j set); a second level folds those over the waves in eight more rounds and
//                           the host joins the 30 leftover points by Horner in powers of two
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include "common.h"
#include "curve.hip.h"
#include "curve_wide.hip.h"

using namespace zkhip;

namespace zkhip {

constexpr int HIST_THREADS = 1024;
constexpr uint32_t ENTRY_IDX_BITS = 26;
constexpr uint32_t ENTRY_IDX_MASK = (1u << ENTRY_IDX_BITS) - 1;

struct MsmTables {
    uint32_t c = 0, W = 0;
    size_t n = 0;
    g1_affine* d_tables = nullptr;   // [W][n]
};

// ---- registration ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) table_next_kernel(const g1_affine* prev, g1_affine* next, size_t n, uint32_t c) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const g1_affine p = load_affine(prev + i);
    g1_xyzz acc = xyzz_from_affine(p);
    for (uint32_t k = 0; k < c; k++) acc = xyzz_double(acc);
    store_affine(next + i, xyzz_to_affine(acc));
}

// R = 2^256 Montgomery coordinates -> the 2^261 domain of the accumulate kernel's arithmetic (x * 32 mod q); the
// identity (0, 0) maps to itself
__global__ void __launch_bounds__(256) table_widen_kernel(g1_affine* pts, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe k;
#pragma unroll
    for (int j = 0; j < 8; j++) k.l[j] = FqParams::TWO5[j];
    g1_affine p = load_affine(pts + i);
    p.x = Fq::mul(p.x, k);
    p.y = Fq::mul(p.y, k);
    store_affine(pts + i, p);
}

// ---- digits ----------------------------------------------------------------------------------------
// Montgomery -> canonical: multiply by the integer 1
ZK_DEV fe fr_canonical(const fe& a) {
    fe one;
#pragma unroll
    for (int i = 0; i < 8; i++) one.l[i] = 0;
    one.l[0] = 1;
    return Fr::mul(a, one);
}
// c-bit field of a 256-bit little-endian integer starting at bit `pos`
// Signed window digits of a canonical scalar: d[w] in (-2^(C-1), 2^(C-1)], sum d[w] 2^(C w) = v.  The window width is a
// template parameter so that, fully unrolled, every word index and shift is a constant and the digits stay in registers.
constexpr uint32_t windows_for(uint32_t c) { return (255 + c - 1) / c; }

template <uint32_t C>
ZK_DEV void signed_digits(const fe& v, int32_t (&d)[windows_for(C)]) {
    uint32_t carry = 0;
#pragma unroll
    for (uint32_t w = 0; w < windows_for(C); w++) {
        const uint32_t pos = w * C, word = pos >> 5, sh = pos & 31;
        uint32_t x = v.l[word] >> sh;
        if (sh + C > 32 && word + 1 < 8) x |= v.l[word + 1] << (32 - sh);
        const uint32_t raw = (x & ((1u << C) - 1)) + carry;
        carry = raw > (1u << (C - 1)) ? 1u : 0u;
        d[w] = (int32_t)raw - (int32_t)(carry << C);
    }
}

template <uint32_t C>
__global__ void __launch_bounds__(HIST_THREADS) digits_hist_kernel(const fe* scalars, size_t n, size_t chunk,
                                                                    uint32_t* counts /*[G][B]*/, uint32_t* bincount /*[G][256]*/) {
    extern __shared__ uint32_t hist[];   // [B] bucket counts, then [256] counts per bin of 128 buckets
    constexpr uint32_t B = 1u << (C - 1), W = windows_for(C);
    const uint32_t g = blockIdx.x;
    uint32_t* binh = hist + B;
    for (uint32_t b = threadIdx.x; b < B + 256; b += HIST_THREADS) hist[b] = 0;
    __syncthreads();
    const size_t lo = (size_t)g * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += HIST_THREADS) {
        int32_t d[W];
        signed_digits<C>(fr_canonical(load_fe(scalars + i)), d);
#pragma unroll
        for (uint32_t w = 0; w < W; w++)
            if (d[w] != 0) atomicAdd(&hist[(d[w] < 0 ? -d[w] : d[w]) - 1], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += HIST_THREADS) {   // a wave covers 64 buckets of one bin
        const uint32_t v = hist[b];
        counts[(size_t)g * B + b] = v;
        uint32_t t = v;
        for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
        if ((threadIdx.x & 63) == 0 && t) atomicAdd(&binh[b >> 7], t);
    }
    __syncthreads();
    if (threadIdx.x < 256) bincount[(size_t)g * 256 + threadIdx.x] = binh[threadIdx.x];
}

// ---- two-level scatter ------------------------------------------------------------------------------
// Writing every 4-byte entry straight to its bucket puts each in a 32-byte HBM sector of its own (8x write
// amplification, measured).  The two-level form first groups a block's entries by coarse bin (128 consecutive buckets) in LDS and writes them as
// runs of (entry, bucket) records into the bin's range of a temporary array (a bin occupies the same range there as
// in the final array); a second kernel then places the records of one bin, whose final range is small enough to stay
// in L2 while it fills.  Any order inside a bucket is fine: bucket sums do not depend on it.
constexpr uint32_t BIN_LOG = 7;                 // buckets per coarse bin = 128
constexpr uint32_t BIN_MASK = (1u << BIN_LOG) - 1;
constexpr uint32_t COARSE_ROUND = 16384;        // records staged per round (128 KiB)
constexpr int COARSE_THREADS = 1024;
constexpr uint32_t FINE_SLICE = 4096;           // records one workgroup of the second pass places at a time
constexpr int FINE_THREADS = 256;
constexpr int FINE_PER_LANE = FINE_SLICE / FINE_THREADS;

// Workgroup g handles the same chunk of scalars as in digits_hist_kernel, T scalars (one per lane) per round: their
// nonzero digits are counted per bin, staged bin by bin in LDS and written out as one run per bin.  The workgroup's
// place in each bin's range follows from the bin counts of the workgroups before it, so the layout of tmp is
// deterministic and no global atomics are needed.
template <uint32_t C>
__global__ void __launch_bounds__(COARSE_THREADS) scatter_coarse_kernel(const fe* scalars, size_t n, size_t chunk, uint32_t T,
                                                                        const uint32_t* bincount, const uint32_t* start,
                                                                        uint2* tmp) {
    extern __shared__ __align__(16) unsigned char smem_sc[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem_sc);        // [256] records of this round per bin
    uint32_t* roff = hist + 256;                                  // [256] exclusive scan of hist
    uint32_t* fill = roff + 256;                                  // [256]
    uint32_t* gdelta = fill + 256;                                // [256] start of this round's run in tmp minus roff
    uint32_t* cursor = gdelta + 256;                              // [256] next free record of this workgroup, per bin
    uint2* staging = reinterpret_cast<uint2*>(cursor + 256);      // [COARSE_ROUND]
    __shared__ uint32_t total_s;
    constexpr uint32_t W = windows_for(C), NB = (1u << (C - 1)) >> BIN_LOG;
    const uint32_t tid = threadIdx.x, g = blockIdx.x;
    // cursor[q] = first record of bin q (= start of its first bucket) + the records of bin q that the workgroups before
    // this one hold.  (Deriving the bin starts from the bin counts instead, so that the bucket scan could run beside this
    // kernel on a second stream, was measured: no gain, the sort phase stayed at 150 us.)
    if (tid < 256) {
        cursor[tid] = tid < NB ? start[tid << BIN_LOG] : 0;
        fill[tid] = 0;
    }
    __syncthreads();
    {
        const uint32_t q = tid & 255, part = tid >> 8;            // four lanes share the rows before g of one bin
        uint32_t sum = 0;
        for (uint32_t r = part; r < g; r += COARSE_THREADS / 256) sum += bincount[(size_t)r * 256 + q];
        if (sum) atomicAdd(&cursor[q], sum);
    }
    const size_t lo = (size_t)g * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (size_t r0 = lo; r0 < hi; r0 += T) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const size_t i = r0 + tid;
        const bool live = tid < T && i < hi;
        int32_t d[W];
        if (live) {
            signed_digits<C>(fr_canonical(load_fe(scalars + i)), d);
#pragma unroll
            for (uint32_t w = 0; w < W; w++)
                if (d[w] != 0) atomicAdd(&hist[((uint32_t)(d[w] < 0 ? -d[w] : d[w]) - 1) >> BIN_LOG], 1u);
        }
        __syncthreads();
        if (tid < 64) {   // one wave scans the 256 counts, four per lane
            const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const uint32_t sum = h0 + h1 + h2 + h3;
            uint32_t inc = sum;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o, 64);
                if ((int)tid >= o) inc += t;
            }
            const uint32_t e = inc - sum;
            roff[4 * tid] = e;
            roff[4 * tid + 1] = e + h0;
            roff[4 * tid + 2] = e + h0 + h1;
            roff[4 * tid + 3] = e + h0 + h1 + h2;
            if (tid == 63) total_s = inc;
        }
        __syncthreads();
        if (tid < 256) {
            const uint32_t cq = cursor[tid];
            gdelta[tid] = cq - roff[tid];
            cursor[tid] = cq + hist[tid];
        }
        if (live) {
#pragma unroll
            for (uint32_t w = 0; w < W; w++) {
                if (d[w] != 0) {
                    const uint32_t bk = (uint32_t)(d[w] < 0 ? -d[w] : d[w]) - 1, q = bk >> BIN_LOG;
                    const uint32_t rank = atomicAdd(&fill[q], 1u);
                    staging[roff[q] + rank] = make_uint2((uint32_t)i | (w << ENTRY_IDX_BITS) | (d[w] < 0 ? 0x80000000u : 0u), bk);
                }
            }
        }
        __syncthreads();
        if (tid < 256) fill[tid] = 0;
        const uint32_t total = total_s;
        for (uint32_t j0 = tid; j0 < total; j0 += 4 * COARSE_THREADS) {
            uint2 rec[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t j = j0 + k * COARSE_THREADS;
                if (j < total) rec[k] = staging[j];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t j = j0 + k * COARSE_THREADS;
                if (j < total) tmp[gdelta[rec[k].y >> BIN_LOG] + j] = rec[k];
            }
        }
    }
}

// Second pass: workgroup (bin q, part) places slices of the bin's records, FINE_SLICE at a time.  A slice is sorted by
// bucket in LDS (ranks from LDS atomics), its per-bucket runs get their place in the bucket ranges with one global atomic
// per (slice, bucket) on cur[] (initialised to the bucket starts), and the runs are written with consecutive lanes on
// consecutive entries.
__global__ void __launch_bounds__(FINE_THREADS) scatter_fine_kernel(const uint2* tmp, const uint32_t* start, const uint32_t* cnt,
                                                                    uint32_t* cur, uint32_t parts, uint32_t* entries) {
    __shared__ uint32_t hist[1u << BIN_LOG], delta[1u << BIN_LOG], loff[1u << BIN_LOG], wave_tot[2];
    __shared__ uint32_t stage_e[FINE_SLICE];
    __shared__ uint8_t stage_l[FINE_SLICE];
    const uint32_t q = blockIdx.x / parts, part = blockIdx.x % parts, tid = threadIdx.x;
    const uint32_t b0 = q << BIN_LOG, b1 = b0 + BIN_MASK;    // the bucket count is a multiple of 128
    const uint32_t lo_bin = start[b0], hi_bin = start[b1] + cnt[b1];
    for (uint32_t lo = lo_bin + part * FINE_SLICE; lo < hi_bin; lo += parts * FINE_SLICE) {
        const uint32_t hi = lo + FINE_SLICE < hi_bin ? lo + FINE_SLICE : hi_bin;
        if (tid < (1u << BIN_LOG)) hist[tid] = 0;
        __syncthreads();
        uint2 rec[FINE_PER_LANE];
        uint32_t rank[FINE_PER_LANE];
#pragma unroll
        for (int k = 0; k < FINE_PER_LANE; k++) {
            const uint32_t j = lo + k * FINE_THREADS + tid;
            if (j < hi) rec[k] = tmp[j];
        }
#pragma unroll
        for (int k = 0; k < FINE_PER_LANE; k++)
            if (lo + k * FINE_THREADS + tid < hi) rank[k] = atomicAdd(&hist[rec[k].y & BIN_MASK], 1u);
        __syncthreads();
        uint32_t v = 0, inc = 0, gb = 0;
        if (tid < (1u << BIN_LOG)) {
            v = hist[tid];
            if (v) gb = atomicAdd(&cur[b0 + tid], v);
            inc = v;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(inc, d, 64);
                if ((int)(tid & 63) >= d) inc += t;
            }
            if ((tid & 63) == 63) wave_tot[tid >> 6] = inc;
        }
        __syncthreads();
        if (tid < (1u << BIN_LOG)) {
            const uint32_t e = inc - v + (tid >= 64 ? wave_tot[0] : 0);
            loff[tid] = e;
            delta[tid] = gb - e;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < FINE_PER_LANE; k++) {
            if (lo + k * FINE_THREADS + tid < hi) {
                const uint32_t l = rec[k].y & BIN_MASK, pos = loff[l] + rank[k];
                stage_e[pos] = rec[k].x;
                stage_l[pos] = (uint8_t)l;
            }
        }
        __syncthreads();
        for (uint32_t j = tid; j < hi - lo; j += FINE_THREADS) entries[delta[stage_l[j]] + j] = stage_e[j];
    }
}

// ---- scans and the slot plan ------------------------------------------------------------------------
// A bucket of cnt entries is folded by f = ceil(cnt / L) lanes ("slots") of near-equal share.  The accumulate phase
// takes rounds * L additions per SIMD, where rounds = ceil(total slots / one slot per lane of the chip): L is picked
// per MSM from a ladder of candidates (2.5 % apart) so that the slots just fill a whole number of rounds.
constexpr uint32_t L_MAX = 256;
constexpr uint32_t PLAN_STRIDE = 128;   // words per block in the plan statistics: [0] max cnt, [1] entries, [2 + i] slots(L_i)
enum { INFO_SLOTS = 0, INFO_MAXCNT = 1, INFO_L = 2, INFO_ENTRIES = 3, INFO_HEAVY = 4 };   // HEAVY: slots in buckets of > 32
constexpr uint32_t MERGE_DIRECT_MAX = 32;   // the final merge level takes buckets of up to this many partial sums
ZK_DEV uint32_t next_slot_len(uint32_t L) { return L + (L < 40 ? 1 : L / 40); }
// the candidates, built once on the host: L[i] for ladder step i (plan word 2 + i), 0 beyond L_MAX
struct SlotLadder {
    uint32_t L[PLAN_STRIDE];
};
static SlotLadder make_ladder(uint32_t l_min) {
    SlotLadder lad;
    uint32_t L = l_min;
    for (uint32_t i = 0; i < PLAN_STRIDE; i++) {
        lad.L[i] = (i + 2 < PLAN_STRIDE && L <= L_MAX) ? L : 0;
        if (L <= L_MAX) L = next_slot_len(L);
    }
    return lad;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// inclusive prefix sum over the 64 lanes with data-parallel-primitive moves (no LDS crossbar round trips): shifts by
// 1, 2, 4, 8 inside each row of 16 lanes (lanes without a source read 0), then the last lane of row 0 / 2 is added to
// row 1 / 3 and the last lane of row 1 to rows 2 and 3
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// cnt[b] = sum over g of counts[g][b].  A block of 1024 lanes covers 256 buckets: a wave reads four buckets per lane
// from one row (1 KiB per load) and the sixteen waves take every sixteenth row; the partial sums meet in LDS.
// Each block also leaves its share of the plan statistics in plan[blockIdx.x][*].
constexpr int COLSUM_THREADS = 1024;
__global__ void __launch_bounds__(COLSUM_THREADS) column_sum_kernel(const uint32_t* counts /*[G][B]*/, uint32_t G, uint32_t B,
                                                                    uint32_t* cnt, const SlotLadder ladder, uint32_t* plan,
                                                                    uint32_t* scan_sync) {
    __shared__ uint32_t sh[PLAN_STRIDE];
    if (blockIdx.x == 0 && threadIdx.x < 2) scan_sync[threadIdx.x] = 0;   // the two counters of bucket_scan_kernel, which runs next
    __shared__ uint32_t part[COLSUM_THREADS / 64][256];
    for (uint32_t i = threadIdx.x; i < PLAN_STRIDE; i += COLSUM_THREADS) sh[i] = 0;
    const uint32_t lane = threadIdx.x & 63, rg = threadIdx.x >> 6, b4 = blockIdx.x * 256 + 4 * lane;
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (b4 < B) {   // B is a multiple of 128, so four buckets are inside together
        for (uint32_t g = rg; g < G; g += COLSUM_THREADS / 64) {
            const uint4 v = *reinterpret_cast<const uint4*>(counts + (size_t)g * B + b4);
            acc.x += v.x;
            acc.y += v.y;
            acc.z += v.z;
            acc.w += v.w;
        }
    }
    *reinterpret_cast<uint4*>(&part[rg][4 * lane]) = acc;
    __syncthreads();
    __shared__ uint32_t cv[256];
    if (threadIdx.x < 256) {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < COLSUM_THREADS / 64; k++) run += part[k][threadIdx.x];
        const uint32_t b = blockIdx.x * 256 + threadIdx.x;
        if (b < B) cnt[b] = run;
        cv[threadIdx.x] = run;
    }
    __syncthreads();
    // statistics: lane i < 2 folds max / sum, lane 2 + i the slots of ladder step i, over the block's 256 bucket sizes
    if (threadIdx.x < PLAN_STRIDE) {
        const uint32_t L = threadIdx.x >= 2 ? ladder.L[threadIdx.x - 2] : 0;
        uint32_t acc = 0;
        if (threadIdx.x == 0) {
            for (int k = 0; k < 256; k++) acc = cv[k] > acc ? cv[k] : acc;
        } else if (threadIdx.x == 1) {
            for (int k = 0; k < 256; k++) acc += cv[k];
        } else if (L) {
            // ceil(c / L) = floor((c + L - 1) * ceil(2^32 / L) / 2^32), exact while (c + L) * L < 2^32
            const uint32_t magic = (uint32_t)((0x100000000ull + L - 1) / L);
            for (int k = 0; k < 256; k++) acc += __umulhi(cv[k] + L - 1, magic);
        }
        sh[threadIdx.x] = acc;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < PLAN_STRIDE; i += COLSUM_THREADS) plan[(size_t)blockIdx.x * PLAN_STRIDE + i] = sh[i];
}

// Block j (256 lanes) covers the buckets [256 j, 256 j + 256) -- the buckets of block j of column_sum_kernel, whose row
// of plan statistics already holds this block's entries and, for every ladder step, its slots.  Every block folds ALL
// rows (64 KiB at 32768 buckets, out of L2): the totals give it the slot length L that minimises rounds * L for
// `round_lanes` lanes per round, the rows before its own the start of its buckets' entries and slots.  With
// f(b) = ceil(cnt[b] / L): f_out[b] = f(b), off_out[b] = exclusive scan of f, off_out[B] = info[INFO_SLOTS] = total,
// start_out = cur_out = exclusive scan of cnt, info[INFO_L, INFO_MAXCNT, INFO_ENTRIES, INFO_HEAVY].  (One block of 1024
// lanes scanning all buckets took 26 us at 32768 buckets, on every MSM's critical path; 128 blocks take 9 us, and a
// 256-lane block of few registers finds room beside the accumulate kernel of the MSM before in a batch.)
// sync[0] collects the blocks' heavy counts, sync[1] counts finished blocks (both zeroed by column_sum_kernel, which runs
// before on the same stream); the last block publishes.
constexpr int SCAN_THREADS = 256;
__global__ void __launch_bounds__(SCAN_THREADS) bucket_scan_kernel(const uint32_t* cnt, uint32_t B, uint32_t* f_out,
                                                                   uint32_t* off_out, uint32_t* start_out, uint32_t* cur_out,
                                                                   const uint32_t* plan, uint32_t plan_blocks,
                                                                   const SlotLadder ladder, uint32_t round_lanes,
                                                                   uint32_t* sync, uint32_t* info) {
    __shared__ uint32_t tot[PLAN_STRIDE], pre[PLAN_STRIDE], wsum[3][SCAN_THREADS / 64];
    __shared__ uint32_t best, best_idx;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = blockIdx.x;
    const uint32_t b = j * SCAN_THREADS + tid;
    const uint32_t c = b < B ? cnt[b] : 0;           // travels while the plan is folded
    if (tid < PLAN_STRIDE) tot[tid] = pre[tid] = 0;
    if (tid == 0) best = 0xffffffffu;
    __syncthreads();
    {
        const uint32_t col = tid & (PLAN_STRIDE - 1);
        uint32_t v = 0, before = 0;
#pragma unroll 8
        for (uint32_t k = tid / PLAN_STRIDE; k < plan_blocks; k += SCAN_THREADS / PLAN_STRIDE) {
            const uint32_t x = plan[(size_t)k * PLAN_STRIDE + col];
            v = col == 0 ? (x > v ? x : v) : v + x;
            before += k < j ? x : 0;
        }
        if (col == 0) atomicMax(&tot[0], v);
        else {
            atomicAdd(&tot[col], v);
            atomicAdd(&pre[col], before);
        }
    }
    __syncthreads();
    if (tid >= 2 && tid < PLAN_STRIDE) {
        // candidate index tid - 2
        const uint32_t L = ladder.L[tid - 2], v = tot[tid];
        if (L) {
            const uint32_t rounds = v == 0 ? 1 : (v + round_lanes - 1) / round_lanes;
            // in 1/16 of a mixed addition per SIMD: the accumulate rounds, the partly filled last round
            // (about 0.7 L, measured) and the merge of v / B partial sums per bucket; ties go to the longer slot
            uint64_t cost = (uint64_t)rounds * L * 16 + (uint64_t)L * 11 + (uint64_t)(v / B) * 8;
            if (cost >= (1u << 20)) cost = (1u << 20) - 1;
            atomicMin(&best, ((uint32_t)cost << 12) | (L_MAX + 1 - L));
        }
    }
    __syncthreads();
    const uint32_t D = L_MAX + 1 - (best & 0xfff);
    if (tid >= 2 && tid < PLAN_STRIDE && ladder.L[tid - 2] == D) best_idx = tid;   // ladder steps are distinct
    __syncthreads();
    const uint32_t col_f = best_idx;
    // ceil(c / D) = floor((c + D - 1) ceil(2^32 / D) / 2^32), one too large at most (c < 2^27, D <= 256): corrected
    const uint32_t magic = (uint32_t)((0x100000000ull + D - 1) / D);
    const uint32_t x = c + D - 1;
    uint32_t f = __umulhi(x, magic);
    if (f * D > x) f--;
    const uint32_t inc_f = wave_inclusive_scan_u32(f), inc_c = wave_inclusive_scan_u32(c);
    const uint32_t heavy = wave_sum_u32(f > MERGE_DIRECT_MAX ? f : 0);
    if (lane == 63) {
        wsum[0][wv] = inc_f;
        wsum[1][wv] = inc_c;
    }
    if (lane == 0) wsum[2][wv] = heavy;
    __syncthreads();
    uint32_t run_f = pre[col_f] + inc_f - f, run_c = pre[1] + inc_c - c, heavy_block = 0;
    for (uint32_t k = 0; k < SCAN_THREADS / 64; k++) {
        if (k < wv) {
            run_f += wsum[0][k];
            run_c += wsum[1][k];
        }
        heavy_block += wsum[2][k];
    }
    if (b < B) {
        f_out[b] = f;
        off_out[b] = run_f;
        start_out[b] = run_c;
        cur_out[b] = run_c;
    }
    if (tid == 0) {
        if (heavy_block) atomicAdd(&sync[0], heavy_block);
        __threadfence();
        if (atomicAdd(&sync[1], 1u) == gridDim.x - 1) {
            __threadfence();
            const uint32_t heavy_total = atomicAdd(&sync[0], 0u);
            off_out[B] = tot[col_f];
            info[INFO_SLOTS] = tot[col_f];
            info[INFO_MAXCNT] = tot[0];
            info[INFO_L] = D;
            info[INFO_ENTRIES] = tot[1];
            info[INFO_HEAVY] = heavy_total;
        }
    }
}

// largest b with off[b] <= t, over the B + 1 entry monotone array off (off[B] = total)
ZK_DEV uint32_t find_bucket(const uint32_t* off, uint32_t B, uint32_t t) {
    uint32_t lo = 0, hi = B;   // invariant: off[lo] <= t < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= t) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ---- accumulate ------------------------------------------------------------------------------------
// lane t = slot s of bucket b: folds entries [s * cnt / f, (s + 1) * cnt / f) of the bucket, f = slots of the bucket
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) accumulate_kernel(const uint32_t* entries, const uint32_t* start, const uint32_t* cnt,
                                                         const uint32_t* slot_off, const uint32_t* slot_cnt, uint32_t B,
                                                         const g1_affine* tables, size_t table_stride, size_t first,
                                                         g1_xyzz* items) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= slot_off[B]) return;
    const uint32_t b = find_bucket(slot_off, B, t);
    const uint32_t s = t - slot_off[b], f = slot_cnt[b], c = cnt[b];
    const uint32_t lo = start[b] + (uint32_t)((uint64_t)s * c / f);
    const uint32_t hi = start[b] + (uint32_t)((uint64_t)(s + 1) * c / f);
    g1_xyzz_w acc;
    if (lo >= hi) {   // cannot happen while a bucket has no more slots than entries; kept so that a bad plan stays harmless
        store_xyzz(items + t, xyzz_identity());
        return;
    }
    uint32_t e = entries[lo];
    g1_affine p = load_affine(tables + (size_t)((e >> ENTRY_IDX_BITS) & 31) * table_stride + first + (e & ENTRY_IDX_MASK));
    // the first point becomes the accumulator (a slot is never empty: lo < hi), so that the loop below starts from a
    // point and "accumulator is the identity" is an exception in it, not a case of every iteration
    {
        const uint32_t ecur = e;
        const g1_affine q = p;
        if (lo + 1 < hi) {
            e = entries[lo + 1];
            p = load_affine(tables + (size_t)((e >> ENTRY_IDX_BITS) & 31) * table_stride + first + (e & ENTRY_IDX_MASK));
        }
        xyzz_w_set_affine<FqWChain>(acc, q, (ecur & 0x80000000u) != 0);
    }
    for (uint32_t k = lo + 1; k < hi; k++) {
        const uint32_t ecur = e;
        const g1_affine q = p;
        if (k + 1 < hi) {   // prefetch the next point while this one is added
            e = entries[k + 1];
            p = load_affine(tables + (size_t)((e >> ENTRY_IDX_BITS) & 31) * table_stride + first + (e & ENTRY_IDX_MASK));
        }
        xyzz_w_add_affine<FqWChain>(acc, q, (ecur & 0x80000000u) != 0);
    }
    xyzz_w_store(items + t, acc);
}

// ---- merge -----------------------------------------------------------------------------------------
// Pre-levels for heavy buckets (skewed scalars: half of all 16-bit scalars carry a digit +1 into window 1, so bucket 0
// can hold hundreds of thousands of entries and thousands of slot sums).  A bucket with more than MERGE_DIRECT_MAX
// items folds them 64 at a time, one WAVE per output item: a butterfly of six dependent additions, where a lane folding
// items one after the other would need one addition per item.  Light buckets are left where they are: the bucket view
// (offset, count) of the next level points either at the old items or at the new ones; both live in one array.
constexpr uint32_t MERGE_WAVE = 64;    // items one wave folds in a pre-level for few, very heavy buckets
constexpr uint32_t MERGE_LANE = 16;    // items one lane folds in a pre-level for many heavy buckets
static_assert(MERGE_WAVE == 1u << 6 && MERGE_LANE == 1u << 4, "heavy_scan_kernel is given log2 of the fold");

// hoff = exclusive scan of f(b) = cnt > direct_max ? ceil(cnt / 2^fold_log) : 0 (hoff[B] = total), and the bucket view
// after the level: heavy buckets at out_base + hoff[b] with f(b) items, light buckets unchanged.  Block j (256 lanes)
// covers the buckets [256 j, 256 j + 256) and sums f over all buckets before its own by itself (at most 128 KiB out of
// L2, four values per load): no block waits for another, and 128 small blocks take 5 us where one block of 1024 lanes
// scanning all 32768 buckets took 27 - 35 us (two such levels sit in the tail of every sparse witness column).
__global__ void __launch_bounds__(SCAN_THREADS) heavy_scan_kernel(const uint32_t* off_in, const uint32_t* cnt_in, uint32_t B,
                                                                  uint32_t fold_log, uint32_t direct_max, uint32_t out_base,
                                                                  uint32_t* hoff, uint32_t* off_out, uint32_t* cnt_out) {
    __shared__ uint32_t wsum[SCAN_THREADS / 64], wpre[SCAN_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = blockIdx.x;
    const uint32_t b = j * SCAN_THREADS + tid, fold_mask = (1u << fold_log) - 1;
    const uint32_t c = b < B ? cnt_in[b] : 0, o = b < B ? off_in[b] : 0;
    uint32_t before = 0;
#pragma unroll 4
    for (uint32_t i = 4 * tid; i < j * SCAN_THREADS; i += 4 * SCAN_THREADS) {   // whole blocks of 256 buckets: no tail
        const uint4 q = *reinterpret_cast<const uint4*>(cnt_in + i);
        before += q.x > direct_max ? (q.x + fold_mask) >> fold_log : 0;
        before += q.y > direct_max ? (q.y + fold_mask) >> fold_log : 0;
        before += q.z > direct_max ? (q.z + fold_mask) >> fold_log : 0;
        before += q.w > direct_max ? (q.w + fold_mask) >> fold_log : 0;
    }
    before = wave_sum_u32(before);
    const bool heavy = c > direct_max;
    const uint32_t f = heavy ? (c + fold_mask) >> fold_log : 0;
    const uint32_t inc = wave_inclusive_scan_u32(f);
    if (lane == 63) wsum[wv] = inc;
    if (lane == 0) wpre[wv] = before;
    __syncthreads();
    uint32_t run = inc - f;
    for (uint32_t k = 0; k < SCAN_THREADS / 64; k++) {
        run += wpre[k];
        if (k < wv) run += wsum[k];
    }
    if (b < B) {
        hoff[b] = run;
        off_out[b] = heavy ? out_base + run : o;
        cnt_out[b] = heavy ? f : c;
        if (b == B - 1) hoff[B] = run + f;
    }
}

// wave t of the launch folds items [lo, lo + 64) of its heavy bucket into items[out_base + t].  Two lanes per point
// (xyzz_w_add_pair): logical lane ll = lane / 2 first adds items 2 ll and 2 ll + 1, then five butterfly rounds join the 32
// logical lanes -- six dependent two-lane additions (~6 us each) where one lane per item took six one-lane additions
// (~9 us each).  One loop, so that the kernel holds one copy of the addition's code.
__global__ void __launch_bounds__(256) merge_wave_kernel(g1_xyzz* items, const uint32_t* off_in, const uint32_t* cnt_in,
                                                         const uint32_t* hoff, uint32_t B, uint32_t out_base) {
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, ll = lane >> 1;
    const bool role = (lane & 1u) != 0;
    if (t >= hoff[B]) return;
    const uint32_t b = find_bucket(hoff, B, t);
    const uint32_t lo = off_in[b] + (t - hoff[b]) * MERGE_WAVE, end = off_in[b] + cnt_in[b];
    g1_xyzz_w p = lo + 2 * ll < end ? xyzz_w_load(items + lo + 2 * ll) : xyzz_w_identity();
    const g1_xyzz_w second = lo + 2 * ll + 1 < end ? xyzz_w_load(items + lo + 2 * ll + 1) : xyzz_w_identity();
#pragma unroll 1
    for (uint32_t r = 0; r < 6; r++) {
        g1_xyzz_w o = second;
        if (r) o = xyzz_w_shfl(p, (int)(((ll ^ (32u >> r)) << 1) | (role ? 1u : 0u)));
        p = xyzz_w_add_pair(p, o, role);
    }
    if (lane == 0) xyzz_w_store(items + out_base + t, p);
}

// the same level for MANY heavy buckets (large uniform MSMs: every bucket holds a hundred slot sums): lane t folds
// MERGE_LANE consecutive items of its bucket one after the other, which costs a sixth of the additions of the wave form
__global__ void __launch_bounds__(256) merge_lane_kernel(g1_xyzz* items, const uint32_t* off_in, const uint32_t* cnt_in,
                                                         const uint32_t* hoff, uint32_t B, uint32_t out_base) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= hoff[B]) return;
    const uint32_t b = find_bucket(hoff, B, t);
    const uint32_t lo = off_in[b] + (t - hoff[b]) * MERGE_LANE, end = off_in[b] + cnt_in[b];
    const uint32_t hi = lo + MERGE_LANE < end ? lo + MERGE_LANE : end;
    g1_xyzz_w acc = xyzz_w_load(items + lo);
    for (uint32_t k = lo + 1; k < hi; k++) acc = xyzz_w_add(acc, xyzz_w_load(items + k));
    xyzz_w_store(items + out_base + t, acc);
}

// ---- reduce ----------------------------------------------------------------------------------------
// sum over b of (b + 1) * S_b without scalar multiplications and with a short dependency chain.  For 2^NB points P_l
// held one per lane, NB rounds of one point addition each leave in lane 0 the total A = sum P_l and in lane 2^j the
// "bit sum" B_j = sum of the P_l with bit j of l set, so that sum l * P_l = sum_j 2^j B_j:
//   round r: every lane whose bit r is clear adds the value of lane l | 2^r.
// The lanes with low r bits clear carry the running totals of their 2^r-blocks; the upper half of each pair keeps its
// total U, which from then on is only summed over the higher bits (it sits in the lanes with lowest set bit r).
// A lone wave needs ~9 us per dependent point addition (~6 us when two lanes share it, below), so the chain length is
// what this phase costs: 6 additions in level 1 and 7 in level 2, against 12 + 12 for a suffix-scan formulation.
constexpr uint32_t RED_STREAMS = 7;    // the total and six bit sums of every wave of level 1

// Both levels run with TWO lanes per point (xyzz_w_add_pair): physical lanes 2 i and 2 i + 1 of a wave are logical lane i
// and hold the same point, an addition costs each lane half of the field products, and the dependency chain that this
// phase consists of shortens by a third (level 1 + level 2: 127 -> 88 us at 2^15 buckets).  A wave holds 32 points: the
// rounds over the low five bits of the logical lane index run inside the wave, the later ones go through LDS.
// one round: logical lanes whose bit `bit` is clear add the value of logical lane ll | 2^bit
__device__ __forceinline__ g1_xyzz_w pairs_round_operand(const g1_xyzz_w& p, uint32_t ll, uint32_t bit, bool role) {
    g1_xyzz_w o = xyzz_w_shfl(p, (int)(((ll | (1u << bit)) << 1) | (role ? 1u : 0u)));
    if ((ll >> bit) & 1u) o.inf = true;
    return o;
}
constexpr size_t REDUCE_OUT = 56;      // four quarters of 14 points each, see reduce_level2_kernel

// The final merge level and level 1 of the reduce in one kernel.
// Merge: the two lanes of a pair share bucket b; lane r folds its items r, r + 2, ... with one-lane additions (the next
// item is in flight during an addition), then the lanes exchange their partial sums and join them with a two-lane
// addition, which leaves the bucket's sum in both lanes -- what the rounds of level 1 start from.  (Four lanes per
// bucket lose: 100 us against 92, and 89 against 74 as two cooperating pairs; their 2048 waves share SIMDs two by two,
// and a lone wave already uses 85 % of its SIMD's issue rate in this code: 8.2 us per addition alone, 14.2 us with a neighbour.)
// Level 1: bucket sums of b < B (B a multiple of 128) -> l1[0][h] = A_h, l1[1 + j][h] = B_{h,j} for the 64-bucket group
// h = b / 64 (J = number of groups = row stride of l1).  A workgroup of four waves takes 128 buckets; waves 2 g and
// 2 g + 1 are the lower and upper 32 logical lanes of the 64-bucket group g, and the round over bit 5 takes the upper
// wave's sums from LDS.
// As two kernels (merge, then level 1 from memory) this took 20 us more of a lone MSM's 830: a launch gap, the
// canonicalisation and store / load and unpack of 32768 bucket sums, and a one-lane addition for the join.
__global__ void __launch_bounds__(256) merge_reduce_level1_kernel(const g1_xyzz* in, const uint32_t* off_in, const uint32_t* cnt_in,
                                                                  uint32_t B, uint32_t J, g1_xyzz* l1) {
    __shared__ g1_xyzz_w ex[2][32];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63, ll = lane >> 1, g = w >> 1, upper = w & 1, r = lane & 1u;
    const bool role = r != 0;
    const uint32_t b = blockIdx.x * 128 + g * 64 + upper * 32 + ll;
    g1_xyzz_w acc = xyzz_w_identity();
    if (b < B) {
        const uint32_t c = cnt_in[b], base = off_in[b];
        if (r < c) {
            acc = xyzz_w_load(in + base + r);
            g1_xyzz nxt;
            if (r + 2 < c) nxt = load_xyzz(in + base + r + 2);
            for (uint32_t i = r + 2; i < c; i += 2) {
                const g1_xyzz cur = nxt;
                if (i + 2 < c) nxt = load_xyzz(in + base + i + 2);   // in flight during the addition
                acc = xyzz_w_add(acc, xyzz_w_unpack(cur));
            }
        }
    }
    // both lanes need the same two operands in the same order: a = the even lane's sum, b = the odd lane's
    const g1_xyzz_w other = xyzz_w_pair_other(acc);
    g1_xyzz_w p = xyzz_w_sel(role, other, acc), o = xyzz_w_sel(role, acc, other);
    // ONE copy of the two-lane addition's code for the join (round 0) and the six rounds of level 1
#pragma unroll 1
    for (uint32_t k = 0; k < 7; k++) {
        if (k >= 1 && k < 6) {
            o = pairs_round_operand(p, ll, k - 1, role);
        } else if (k == 6) {
            if (upper && !role) ex[g][ll] = p;
            __syncthreads();
            o = ex[g][ll];
            if (upper) o.inf = true;
        }
        p = xyzz_w_add_pair(p, o, role);
    }
    if (b < B && !role && (ll & (ll - 1)) == 0) {
        // lower wave: logical lanes 0, 1, 2, 4, 8, 16 = total and bit sums 0..4; upper wave: logical lane 0 = bit sum 5
        const uint32_t stream = upper ? 6u : (ll ? 1u + (uint32_t)__builtin_ctz(ll) : 0u);
        if (!upper || ll == 0) xyzz_w_store(l1 + (size_t)stream * J + (b >> 6), p);
    }
}

// level 2: workgroup (stream, quarter) folds l1[stream][128 quarter .. 128 quarter + 128), J <= 512, with four waves of 32
// logical lanes (one wave per SIMD of its CU); then wave 0 folds the four results of the six groups (total and bit sums
// 0..4 of the waves).  Over the bits of h = 128 quarter + 32 w + ll the totals' bit sums are D_0..D_4 (ll), D_5, D_6 (w);
// D_7 and D_8 are sums of quarter totals, which the host forms.
// out[14 quarter + ..]: [0] total, [1 + i] D_i (i < 7, this quarter's part), [8 + j] this quarter's sum of B_j (j < 6)
__global__ void __launch_bounds__(256) reduce_level2_kernel(const g1_xyzz* l1, uint32_t J, g1_xyzz* out) {
    __shared__ g1_xyzz_w ex[4][6];
    const uint32_t stream = blockIdx.x >> 2, quarter = blockIdx.x & 3, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t ll = lane >> 1;
    const bool role = (lane & 1u) != 0;
    const uint32_t h = quarter * 128 + w * 32 + ll;
    g1_xyzz_w p = h < J ? xyzz_w_load(l1 + (size_t)stream * J + h) : xyzz_w_identity();
    // rounds 0..4 over the bits of ll inside every wave; then wave 0 alone takes logical lane 4 g + u = element u (the wave
    // index of above) of group g (group 0 = totals, group 1 + j = bit sums j) and runs rounds 5, 6 over the bits of u.
    // One loop, so that the kernel holds one copy of the addition's code.
    uint32_t g = 0, u = 0;
#pragma unroll 1
    for (uint32_t r = 0; r < 7; r++) {
        if (r == 5) {
            if (!role && (ll & (ll - 1)) == 0) ex[w][ll ? 1 + __builtin_ctz(ll) : 0] = p;
            __syncthreads();
            if (w != 0) return;
            g = ll >> 2;
            u = ll & 3;
            if (g < 6) p = ex[u][g];
            else p = xyzz_w_identity();
        }
        const g1_xyzz_w o = pairs_round_operand(p, ll, r < 5 ? r : r - 5, role);
        p = xyzz_w_add_pair(p, o, role);
    }
    const g1_xyzz_w q = p;
    if (role) return;
    g1_xyzz* o = out + 14 * quarter;
    if (stream == 0) {
        if (g == 0 && u != 3) xyzz_w_store(o + (u ? 5 + u : 0), q);        // total, D_5 (u = 1), D_6 (u = 2)
        else if (g >= 1 && g < 6 && u == 0) xyzz_w_store(o + g, q);         // D_0..D_4
    } else if (ll == 0) {
        xyzz_w_store(o + 7 + stream, q);                                    // sum of B_{., stream - 1}
    }
}

// out[i] = (first + i) * P for an affine point P: the synthetic SRS of the benchmark and the parity tests
__global__ void __launch_bounds__(256) g1_multiples_kernel(const g1_affine* base, uint64_t first, size_t n, g1_affine* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const g1_affine p = load_affine(base);
    const uint64_t k = first + i;
    g1_xyzz acc = xyzz_identity();
    for (int bit = 63; bit >= 0; bit--) {
        acc = xyzz_double(acc);
        if ((k >> bit) & 1) xyzz_add_affine(acc, p);
    }
    store_affine(out + i, xyzz_to_affine(acc));
}

__global__ void __launch_bounds__(256) g1_add_affine_kernel(const g1_affine* p, const g1_affine* q, g1_affine* o, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1_xyzz acc = xyzz_from_affine(load_affine(p + i));
    xyzz_add_affine(acc, load_affine(q + i));
    store_affine(o + i, xyzz_to_affine(acc));
}

// ---- host side -------------------------------------------------------------------------------------
static uint32_t log2_ceil(size_t n) {
    uint32_t l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = getenv(name);
    return v && *v ? (uint32_t)atoi(v) : dflt;
}
// experiments: force the slot length (0 = planned per MSM) / items folded per lane in a merge pre-level
// a forced slot length (measurements only) of 1 is raised to 2: the kernels divide by the slot length with a 32-bit
// reciprocal, (2^32 + D - 1) / D, which does not fit for D = 1
static const uint32_t FORCE_L = env_u32("ZKHIP_SLOT_L", 0) == 1 ? 2 : env_u32("ZKHIP_SLOT_L", 0);

// Buffers of one in-flight MSM.  A workspace is bound to a stream; several workspaces let the latency-bound tail of
// one MSM (merge, reduce, host fold) overlap the accumulate phase of the next.
struct MsmWorkspace {
    // stream: recode / count / plan / scatter.  acc_stream / tail_stream: the accumulate kernel and merge + reduce when
    // the pipelined batch runs them on streams of their own (null: everything on `stream`).
    hipStream_t stream = nullptr, acc_stream = nullptr, tail_stream = nullptr;
    hipEvent_t ev_info = nullptr;      // the plan has reached h_info
    hipEvent_t ev_sorted = nullptr, ev_acc = nullptr, ev_done = nullptr;   // hand-offs between the three streams
    hipStream_t acc() const { return acc_stream ? acc_stream : stream; }
    hipStream_t tail() const { return tail_stream ? tail_stream : stream; }
    unsigned char* base = nullptr;
    size_t bytes = 0;
    uint32_t* h_info = nullptr;        // pinned copy of info[0..3]
    uint64_t* h_sums = nullptr;        // pinned: the REDUCE_OUT leftover sums of the reduce (XYZZ)
    // per-call state
    size_t n = 0, first = 0;
    const fe* d_scalars = nullptr;
    uint32_t G = 0, l_min = 0, plan_blocks = 0, acc_blocks = 0;
    SlotLadder ladder;
    bool nonzero = false;
    uint32_t *counts, *entries, *start, *cnt, *offA, *cntA, *offB, *cntB, *info, *plan, *cur, *bincount, *hoff;
    uint2* tmp;
    g1_xyzz *itemsA, *itemsB, *l1, *out2;
};

static int ws_prepare(zkhip_ctx* ctx, MsmWorkspace& ws, const MsmTables& tb, size_t n) {
    const uint32_t B = 1u << (tb.c - 1), W = tb.W;
    uint32_t G = (uint32_t)std::min<size_t>((size_t)ctx->num_cus, (n + HIST_THREADS - 1) / HIST_THREADS);
    if (G < 1) G = 1;
    ws.G = G;
    const size_t E_max = n * W, round_lanes = (size_t)ctx->num_cus * 256;
    // shortest slot the plan may pick: keeps the partial-sum buffers within E_max / l_min + B items
    ws.l_min = (uint32_t)std::min<size_t>(64, std::max<size_t>(8, E_max / (8 * round_lanes)));
    if (FORCE_L) ws.l_min = std::min(ws.l_min, FORCE_L);
    ws.ladder = make_ladder(ws.l_min);
    if (FORCE_L) {                       // measurements: the forced length is the only candidate
        for (uint32_t i = 0; i < PLAN_STRIDE; i++) ws.ladder.L[i] = 0;
        ws.ladder.L[0] = std::min(FORCE_L, L_MAX);
    }
    ws.plan_blocks = (B + 255) / 256;
    const size_t T1_max = E_max / ws.l_min + B + 1;
    const size_t J = (B + 63) / 64;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    const size_t o_info = take(64);
    const size_t o_counts = take((size_t)B * G * 4), o_entries = take(E_max * 4), o_tmp = take(E_max * 8),
                 o_cur = take((size_t)B * 4), o_bin = take((size_t)G * 256 * 4), o_start = take(((size_t)B + 1) * 4),
                 o_cnt = take((size_t)B * 4), o_offA = take(((size_t)B + 1) * 4), o_cntA = take((size_t)B * 4),
                 o_offB = take(((size_t)B + 1) * 4), o_cntB = take((size_t)B * 4),
                 o_plan = take((size_t)ws.plan_blocks * PLAN_STRIDE * 4), o_itemsA = take(T1_max * sizeof(g1_xyzz)),
                 o_itemsB = take((T1_max / 8 + 2 * (size_t)B + 64) * sizeof(g1_xyzz)), o_hoff = take(((size_t)B + 1) * 4),
                 o_l1 = take((size_t)RED_STREAMS * J * sizeof(g1_xyzz)),
                 o_out2 = take(REDUCE_OUT * sizeof(g1_xyzz));
    if (off > ws.bytes) {
        if (ws.base) {
            ZK_HIP(ctx, hipStreamSynchronize(ws.stream));
            ZK_HIP(ctx, hipStreamSynchronize(ws.acc()));
            ZK_HIP(ctx, hipStreamSynchronize(ws.tail()));
            ZK_HIP(ctx, hipFree(ws.base));
            ws.base = nullptr;
            ws.bytes = 0;
        }
        ZK_HIP(ctx, hipMalloc((void**)&ws.base, off));
        ws.bytes = off;
    }
    // page-locked and coherent: the plan and the reduce's leftover sums are written there by the kernels themselves
    // (one copy command and its launch gap less on the critical path of a lone MSM, twice)
    if (!ws.h_info) ZK_HIP(ctx, hipHostMalloc((void**)&ws.h_info, 64, hipHostMallocMapped | hipHostMallocCoherent));
    if (!ws.ev_info) {
        ZK_HIP(ctx, hipEventCreateWithFlags(&ws.ev_info, hipEventDisableTiming));
        ZK_HIP(ctx, hipEventCreateWithFlags(&ws.ev_sorted, hipEventDisableTiming));
        ZK_HIP(ctx, hipEventCreateWithFlags(&ws.ev_acc, hipEventDisableTiming));
        ZK_HIP(ctx, hipEventCreateWithFlags(&ws.ev_done, hipEventDisableTiming));
    }
    ws.acc_blocks = (uint32_t)((T1_max + 255) / 256);
    if (!ws.h_sums)
        ZK_HIP(ctx, hipHostMalloc((void**)&ws.h_sums, REDUCE_OUT * sizeof(g1_xyzz), hipHostMallocMapped | hipHostMallocCoherent));
    unsigned char* p = ws.base;
    ws.counts = (uint32_t*)(p + o_counts);
    ws.entries = (uint32_t*)(p + o_entries);
    ws.tmp = (uint2*)(p + o_tmp);
    ws.cur = (uint32_t*)(p + o_cur);
    ws.bincount = (uint32_t*)(p + o_bin);
    ws.start = (uint32_t*)(p + o_start);
    ws.cnt = (uint32_t*)(p + o_cnt);
    ws.offA = (uint32_t*)(p + o_offA);
    ws.cntA = (uint32_t*)(p + o_cntA);
    ws.offB = (uint32_t*)(p + o_offB);
    ws.cntB = (uint32_t*)(p + o_cntB);
    ws.info = (uint32_t*)(p + o_info);
    ws.plan = (uint32_t*)(p + o_plan);
    ws.itemsA = (g1_xyzz*)(p + o_itemsA);
    ws.itemsB = (g1_xyzz*)(p + o_itemsB);
    ws.l1 = (g1_xyzz*)(p + o_l1);
    ws.hoff = (uint32_t*)(p + o_hoff);
    ws.out2 = (g1_xyzz*)(p + o_out2);
    return ZKHIP_OK;
}

// phase 1: recode, count, scan and plan, scatter; asynchronous readback of {slots, max bucket size, slot length}
static int msm_phase1(zkhip_ctx* ctx, MsmWorkspace& ws, const MsmTables& tb, size_t first, const fe* d_scalars, size_t n) {
    int rc = ws_prepare(ctx, ws, tb, n);
    if (rc) return rc;
    ws.n = n;
    ws.first = first;
    ws.d_scalars = d_scalars;
    const uint32_t c = tb.c, W = tb.W, B = 1u << (c - 1), G = ws.G;
    const size_t chunk = (n + G - 1) / G;
    const size_t lds_hist = (size_t)B * 4 + 256 * 4;
    hipStream_t st = ws.stream;
    ZK_CHECK(ctx, tb.W == windows_for(c), "msm: window count %u does not match window width %u", tb.W, c);
#define ZK_FOR_WINDOW_WIDTH(X) \
    switch (c) {               \
        case 8: X(8); break;   \
        case 9: X(9); break;   \
        case 10: X(10); break; \
        case 11: X(11); break; \
        case 12: X(12); break; \
        case 13: X(13); break; \
        case 14: X(14); break; \
        case 15: X(15); break; \
        case 16: X(16); break; \
        default: ZK_CHECK(ctx, false, "msm: unsupported window width %u", c); \
    }
#define ZK_LAUNCH_HIST(C) \
    hipLaunchKernelGGL(digits_hist_kernel<C>, dim3(G), dim3(HIST_THREADS), lds_hist, st, d_scalars, n, chunk, ws.counts, ws.bincount)
    ZK_FOR_WINDOW_WIDTH(ZK_LAUNCH_HIST)
#undef ZK_LAUNCH_HIST
    hipLaunchKernelGGL(column_sum_kernel, dim3(ws.plan_blocks), dim3(COLSUM_THREADS), 0, st, ws.counts, G, B, ws.cnt, ws.ladder, ws.plan, ws.info);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(ws.plan_blocks), dim3(SCAN_THREADS), 0, st, ws.cnt, B, ws.cntA, ws.offA, ws.start,
                       ws.cur, ws.plan, ws.plan_blocks, ws.ladder, (uint32_t)ctx->num_cus * 256u, ws.info, ws.h_info);
    // the plan is in host memory when this event completes, while scatter and accumulate run
    ZK_HIP(ctx, hipEventRecord(ws.ev_info, st));
    {
        const size_t lds_coarse = 5 * 256 * 4 + (size_t)COARSE_ROUND * 8;
        const uint32_t T = std::min<uint32_t>(COARSE_THREADS, COARSE_ROUND / W);
        const uint32_t NB = B >> BIN_LOG;
        // slices of an evenly filled bin, with a quarter to spare; fuller bins take more turns of the slice loop
        const uint32_t parts = (uint32_t)std::max<size_t>(1, (n * W / NB * 5 / 4 + FINE_SLICE - 1) / FINE_SLICE);
#define ZK_LAUNCH_COARSE(C)                                                                                              \
    hipLaunchKernelGGL(scatter_coarse_kernel<C>, dim3(G), dim3(COARSE_THREADS), lds_coarse, st, d_scalars, n, chunk, T, \
                       ws.bincount, ws.start, ws.tmp)
        ZK_FOR_WINDOW_WIDTH(ZK_LAUNCH_COARSE)
#undef ZK_LAUNCH_COARSE
#undef ZK_FOR_WINDOW_WIDTH
        hipLaunchKernelGGL(scatter_fine_kernel, dim3(NB * parts), dim3(FINE_THREADS), 0, st, ws.tmp, ws.start, ws.cnt, ws.cur,
                           parts, ws.entries);
    }
    ZK_HIP(ctx, hipGetLastError());
    if (ws.acc() != ws.stream) ZK_HIP(ctx, hipEventRecord(ws.ev_sorted, ws.stream));
    return ZKHIP_OK;
}

// accumulate: launched for the largest slot count the plan may choose; lanes beyond the planned slots leave at once
static int msm_accumulate(zkhip_ctx* ctx, MsmWorkspace& ws, const MsmTables& tb) {
    const uint32_t B = 1u << (tb.c - 1);
    if (ws.acc() != ws.stream) ZK_HIP(ctx, hipStreamWaitEvent(ws.acc(), ws.ev_sorted, 0));
    hipLaunchKernelGGL(accumulate_kernel, dim3(ws.acc_blocks), dim3(256), 0, ws.acc(), ws.entries, ws.start, ws.cnt, ws.offA,
                       ws.cntA, B, tb.d_tables, tb.n, ws.first, ws.itemsA);
    ZK_HIP(ctx, hipGetLastError());
    if (ws.tail() != ws.acc()) ZK_HIP(ctx, hipEventRecord(ws.ev_acc, ws.acc()));
    return ZKHIP_OK;
}

// phase 2: waits for the plan (the accumulate kernel is already queued), then merge and reduce; asynchronous readback
// of the leftover sums
static int msm_phase2(zkhip_ctx* ctx, MsmWorkspace& ws, const MsmTables& tb, bool timed) {
    hipStream_t st = ws.tail();
    ZK_HIP(ctx, hipEventSynchronize(ws.ev_info));
    if (ws.tail() != ws.acc()) ZK_HIP(ctx, hipStreamWaitEvent(st, ws.ev_acc, 0));
    const uint32_t total_slots = ws.h_info[INFO_SLOTS], max_cnt = ws.h_info[INFO_MAXCNT], L = ws.h_info[INFO_L];
    const uint32_t B = 1u << (tb.c - 1);
    ws.nonzero = total_slots != 0;
    static const bool debug = getenv("ZKHIP_DEBUG") != nullptr;
    if (debug)
        fprintf(stderr, "[zkhip] msm n=%zu: slots=%u max_cnt=%u L=%u entries=%u heavy=%u\n", ws.n, total_slots, max_cnt, L,
                ws.h_info[INFO_ENTRIES], ws.h_info[INFO_HEAVY]);
    if (!ws.nonzero) {                  // every scalar is zero
        ZK_HIP(ctx, hipEventRecord(ws.ev_done, st));
        return ZKHIP_OK;
    }
    // all items live in one array: itemsA (the slot sums) followed by itemsB (the outputs of the pre-levels, appended
    // level after level); bucket views index it from itemsA
    g1_xyzz* items = ws.itemsA;
    uint32_t out_base = (uint32_t)(ws.itemsB - ws.itemsA);
    uint32_t *cur_off = ws.offA, *cur_cnt = ws.cntA, *nxt_off = ws.offB, *nxt_cnt = ws.cntB;
    {
        TimedRegion tr(ctx, timed ? "msm_merge" : nullptr, st);
        // level_heavy: (a bound on) the items in buckets of more than MERGE_DIRECT_MAX items.  Few of them (one or
        // two huge buckets of skewed scalars): one wave folds 64, six dependent additions per level.  Many (every
        // bucket of a large uniform MSM): one lane folds 16, a sixth of the work.
        // The final level folds a bucket with two lanes, so what it costs is its largest bucket: a pre-level pays only
        // where some buckets are far above the average (a large uniform MSM has ~30 slot sums in every bucket).
        uint32_t level_max = (max_cnt + L - 1) / L, level_heavy = ws.h_info[INFO_HEAVY];
        const uint32_t direct_max = std::max(MERGE_DIRECT_MAX, 2 * ((total_slots + B - 1) / B));
        while (level_max > direct_max) {
            const bool by_wave = level_heavy / MERGE_WAVE <= 8u * 1024u;
            const uint32_t fold = by_wave ? MERGE_WAVE : MERGE_LANE;
            hipLaunchKernelGGL(heavy_scan_kernel, dim3((B + SCAN_THREADS - 1) / SCAN_THREADS), dim3(SCAN_THREADS), 0, st, (const uint32_t*)cur_off, (const uint32_t*)cur_cnt, B,
                               by_wave ? 6u : 4u /* log2 fold */, direct_max, out_base, ws.hoff, nxt_off, nxt_cnt);
            // a heavy bucket yields at most cnt / fold + 1 outputs, and there are at most min(B, heavy / 33) of them; all
            // levels together stay below the T1 / 8 + 2 B + 64 items reserved behind the slot sums
            const uint32_t bound = level_heavy / fold + std::min(B, level_heavy / (MERGE_DIRECT_MAX + 1)) + 1;
            if (by_wave)
                hipLaunchKernelGGL(merge_wave_kernel, dim3((bound + 3) / 4), dim3(256), 0, st, items, (const uint32_t*)cur_off,
                                   (const uint32_t*)cur_cnt, (const uint32_t*)ws.hoff, B, out_base);
            else
                hipLaunchKernelGGL(merge_lane_kernel, dim3((bound + 255) / 256), dim3(256), 0, st, items, (const uint32_t*)cur_off,
                                   (const uint32_t*)cur_cnt, (const uint32_t*)ws.hoff, B, out_base);
            ZK_HIP(ctx, hipGetLastError());
            std::swap(cur_off, nxt_off);
            std::swap(cur_cnt, nxt_cnt);
            out_base += bound;
            level_max = (level_max + fold - 1) / fold;
            level_heavy = bound;
        }
        hipLaunchKernelGGL(merge_reduce_level1_kernel, dim3(B / 128), dim3(256), 0, st, (const g1_xyzz*)items,
                           (const uint32_t*)cur_off, (const uint32_t*)cur_cnt, B, B / 64, ws.l1);
        ZK_HIP(ctx, hipGetLastError());
    }
    {
        TimedRegion tr(ctx, timed ? "msm_reduce" : nullptr, st);
        const uint32_t J = B / 64;   // B is a multiple of 128
        hipLaunchKernelGGL(reduce_level2_kernel, dim3(4 * RED_STREAMS), dim3(256), 0, st, ws.l1, J,
                           reinterpret_cast<g1_xyzz*>(ws.h_sums));
        ZK_HIP(ctx, hipGetLastError());
    }
    ZK_HIP(ctx, hipEventRecord(ws.ev_done, st));
    return ZKHIP_OK;
}

// ---- host fold in XYZZ coordinates (values in the host's 2^256 Montgomery domain; identity: zz = 0) ----
struct HXyzz {
    uint64_t x[4], y[4], zz[4], zzz[4];
};
static void hx_double(HXyzz& o, const HXyzz& p) {
    const HostField& F = HFQ;
    if (h_is_zero(p.zz)) {
        o = p;
        return;
    }
    uint64_t u[4], v[4], w[4], s[4], m[4], t[4], x3[4], y3[4];
    h_add(F, u, p.y, p.y);
    h_mul(F, v, u, u);
    h_mul(F, w, u, v);
    h_mul(F, s, p.x, v);
    h_mul(F, t, p.x, p.x);
    h_add(F, m, t, t);
    h_add(F, m, m, t);                 // 3 x^2 (a = 0)
    h_mul(F, x3, m, m);
    h_sub(F, x3, x3, s);
    h_sub(F, x3, x3, s);
    h_sub(F, t, s, x3);
    h_mul(F, y3, m, t);
    h_mul(F, t, w, p.y);
    h_sub(F, y3, y3, t);
    h_mul(F, o.zz, v, p.zz);
    h_mul(F, o.zzz, w, p.zzz);
    memcpy(o.x, x3, 32);
    memcpy(o.y, y3, 32);
}
static void hx_add(HXyzz& o, const HXyzz& a, const HXyzz& b) {
    const HostField& F = HFQ;
    if (h_is_zero(a.zz)) {
        o = b;
        return;
    }
    if (h_is_zero(b.zz)) {
        o = a;
        return;
    }
    uint64_t u1[4], u2[4], s1[4], s2[4], p[4], r[4];
    h_mul(F, u1, a.x, b.zz);
    h_mul(F, u2, b.x, a.zz);
    h_mul(F, s1, a.y, b.zzz);
    h_mul(F, s2, b.y, a.zzz);
    h_sub(F, p, u2, u1);
    h_sub(F, r, s2, s1);
    if (h_is_zero(p)) {
        if (h_is_zero(r)) {
            hx_double(o, a);
        } else {
            memset(&o, 0, sizeof o);
        }
        return;
    }
    uint64_t pp[4], ppp[4], q[4], x3[4], y3[4], t[4];
    h_mul(F, pp, p, p);
    h_mul(F, ppp, p, pp);
    h_mul(F, q, u1, pp);
    h_mul(F, x3, r, r);
    h_sub(F, x3, x3, ppp);
    h_sub(F, x3, x3, q);
    h_sub(F, x3, x3, q);
    h_sub(F, t, q, x3);
    h_mul(F, y3, r, t);
    h_mul(F, t, s1, ppp);
    h_sub(F, y3, y3, t);
    h_mul(F, t, a.zz, b.zz);
    h_mul(F, o.zz, t, pp);
    h_mul(F, t, a.zzz, b.zzz);
    h_mul(F, o.zzz, t, ppp);
    memcpy(o.x, x3, 32);
    memcpy(o.y, y3, 32);
}

// phase 3: waits for the leftover sums of the reduce and joins them on the host:
//   result = total + sum_{t < 15} 2^t X_t,  X_t = sum over the waves of B_t for t < 6, D_(t - 6) for t >= 6
// (bucket b = 64 h + l weighs b + 1 = 64 h + l + 1), by Horner: 14 doublings and 15 additions, after the four quarters
// of level 2 have been added up (D_7, D_8 = sums of quarter totals).  The fold stays in XYZZ coordinates (an addition is 14
// field products, and nothing is converted until the end); the device's x 2^261 become the host's x 2^256 by one
// product with 1 / 32 per coordinate.
static int msm_phase3(zkhip_ctx* ctx, MsmWorkspace& ws, uint64_t out_jacobian[12]) {
    ZK_HIP(ctx, hipEventSynchronize(ws.ev_done));   // this MSM's sums have arrived (the stream may hold later MSMs' work)
    memset(out_jacobian, 0, 96);
    if (!ws.nonzero) return ZKHIP_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    struct Inv32 {
        uint64_t v[4];
        Inv32() {
            uint64_t k32[4];
            h_from_u64(HFQ, k32, 32);
            h_inv(HFQ, v, k32);
        }
    };
    static const Inv32 inv32;
    constexpr int parts = 4, per = 14;           // level 2 leaves four quarters of 14 points each
    HXyzz pt[parts][per];
    for (int hf = 0; hf < parts; hf++)
        for (int i = 0; i < per; i++) {
            const uint64_t* in = ws.h_sums + 16 * (per * hf + i);
            HXyzz& q = pt[hf][i];
            h_mul(HFQ, q.x, in, inv32.v);
            h_mul(HFQ, q.y, in + 4, inv32.v);
            h_mul(HFQ, q.zz, in + 8, inv32.v);
            h_mul(HFQ, q.zzz, in + 12, inv32.v);
        }
    auto over_parts = [&](HXyzz& o, int idx) {
        o = pt[0][idx];
        for (int q = 1; q < parts; q++) hx_add(o, o, pt[q][idx]);
    };
    HXyzz acc, t;
    memset(&acc, 0, sizeof acc);
    for (int k = 14; k >= 0; k--) {
        hx_double(acc, acc);
        if (k == 14) hx_add(t, pt[2][0], pt[3][0]);          // D_8: the totals of quarters 2 and 3
        else if (k == 13) hx_add(t, pt[1][0], pt[3][0]);     // D_7: quarters 1 and 3
        else over_parts(t, k < 6 ? 8 + k : 1 + (k - 6));     // B_k or D_(k - 6)
        hx_add(acc, acc, t);
    }
    over_parts(t, 0);
    hx_add(acc, acc, t);
    // XYZZ -> Jacobian: X' = X ZZ ZZZ^2, Y' = Y ZZ^3 ZZZ^2, Z' = ZZ ZZZ
    if (!h_is_zero(acc.zz)) {
        uint64_t zzz2[4], zz2[4], zz3[4], u[4];
        h_mul(HFQ, zzz2, acc.zzz, acc.zzz);
        h_mul(HFQ, zz2, acc.zz, acc.zz);
        h_mul(HFQ, zz3, zz2, acc.zz);
        h_mul(HFQ, u, acc.x, acc.zz);
        h_mul(HFQ, out_jacobian, u, zzz2);
        h_mul(HFQ, u, acc.y, zz3);
        h_mul(HFQ, out_jacobian + 4, u, zzz2);
        h_mul(HFQ, out_jacobian + 8, acc.zz, acc.zzz);
    }
    if (ctx->timing == 1) {
        auto& slot = ctx->timers["msm_host_fold"];
        slot.total_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
        slot.launches++;
    }
    return ZKHIP_OK;
}

// Workspaces of a pipelined batch: MSM j is sorted and accumulated while MSM j - 1 is merged and reduced and the sums of
// MSM j - 1 - lag .. are joined on the host, so that lag + 1 merge + reduce tails are in flight, each on its workspace's
// own stream.  A tail is a chain of dependent point additions (0.2 - 0.5 ms whatever the size): for sparse witness
// columns, whose other phases take tens of microseconds, the tails in flight set the rate of the batch (64 byte columns
// of 2^19: 0.252 ms per MSM with a lag of 3, 0.228 with 6 or 8; dense columns 0.79 -> 0.765 ms when they are
// cache-resident, unchanged when they stream from HBM).  The lag is MSM_TAIL_LAG (ZKHIP_MSM_TAIL_LAG overrides it for
// measurements), less when the workspaces of that many MSMs would exceed WS_BUDGET_BYTES.
constexpr int MSM_TAIL_LAG_MAX = 8;
constexpr int MSM_PIPE = 2 + MSM_TAIL_LAG_MAX;   // workspaces that exist; 2 + the tail lag in use take part
static const uint32_t MSM_TAIL_LAG = std::min<uint32_t>(MSM_TAIL_LAG_MAX, std::max<uint32_t>(1, env_u32("ZKHIP_MSM_TAIL_LAG", 8)));
constexpr size_t WS_BUDGET_BYTES = (size_t)8 << 30;
// upper estimate of one workspace for n scalars (the large arrays of ws_prepare)
static size_t ws_bytes_estimate(zkhip_ctx* ctx, const MsmTables& tb, size_t n) {
    const size_t B = (size_t)1 << (tb.c - 1), E = n * tb.W, round_lanes = (size_t)ctx->num_cus * 256;
    const size_t l_min = std::min<size_t>(64, std::max<size_t>(8, E / (8 * round_lanes)));
    const size_t G = std::min<size_t>((size_t)ctx->num_cus, (n + HIST_THREADS - 1) / HIST_THREADS);
    const size_t T1 = E / l_min + B + 1;
    return E * 12 + (T1 + T1 / 8 + 3 * B + 64) * sizeof(g1_xyzz) + B * std::max<size_t>(G, 1) * 4 + ((size_t)1 << 20);
}
struct MsmState {
    std::map<uint32_t, MsmTables> tables;
    MsmWorkspace ws[MSM_PIPE];
    // streams: [0] recode / count / plan / scatter and [1] accumulate of the even workspaces, [2] and [3] the same of the
    // odd ones, [4 + i] merge + reduce of workspace i
    static constexpr int N_STREAMS = 4 + MSM_PIPE;
    hipEvent_t ev_in = nullptr, ev_out[N_STREAMS] = {};
    hipStream_t batch_streams[N_STREAMS] = {};
    std::vector<hipStream_t> idle_streams;   // never used: they take turns of the runtime's queue assignment
};
// Per-context MSM state.  The map is shared by all contexts of the process (one per device), so lookups are
// serialised; a context itself is not re-entrant, so the state found is then used without the lock.  std::map never
// moves its nodes: the reference stays valid while other contexts come and go.
static std::mutex& registry_mutex() {
    static std::mutex m;
    return m;
}
static std::map<zkhip_ctx*, MsmState>& registry_unlocked() {
    static std::map<zkhip_ctx*, MsmState> r;
    return r;
}
static MsmState& state_of(zkhip_ctx* ctx) {
    std::lock_guard<std::mutex> lock(registry_mutex());
    return registry_unlocked()[ctx];
}

static int check_args(zkhip_ctx* ctx, const MsmTables& tb, size_t first, size_t n) {
    ZK_CHECK(ctx, first + n <= tb.n, "msm: range [%zu, %zu) exceeds the %zu registered bases", first, first + n, tb.n);
    ZK_CHECK(ctx, n <= ((size_t)1 << ENTRY_IDX_BITS), "msm: n=%zu exceeds 2^26", n);
    return ZKHIP_OK;
}

struct MsmItem {
    size_t first;
    const fe* d_scalars;
    size_t n;
};

// Up to MSM_PIPE MSMs in flight on internal streams, so that one MSM's latency-bound tail (merge, reduce, host join)
// overlaps the accumulate phase of the next.  out: one Jacobian point per item.
static int msm_pipeline(zkhip_ctx* ctx, MsmState& stt, const MsmTables& tb, const std::vector<MsmItem>& items, uint64_t* out,
                        bool timed) {
    const size_t batch = items.size();
    if (!stt.ev_in) {
        ZK_HIP(ctx, hipEventCreateWithFlags(&stt.ev_in, hipEventDisableTiming));
        for (int i = 0; i < MsmState::N_STREAMS; i++) ZK_HIP(ctx, hipEventCreateWithFlags(&stt.ev_out[i], hipEventDisableTiming));
    }
    // Streams by role, shared by the workspaces: [0] recode / count / plan / scatter, [1] accumulate (a second stream of
    // each kind serves the odd workspaces, see below).  The accumulate
    // kernels run back to back on their own stream; the short kernels before them have a stream of their own, so that
    // they queue behind each other and not behind an accumulate (24 us of histogram take 560 us next to one, and delay
    // the next accumulate, which waits for them).  Merge + reduce run on a stream per workspace: the tails of
    // consecutive MSMs are latency chains and overlap.
    // Stream priorities (accumulate lowest, the others highest) gain another 2 % in a batch that starts on an idle
    // device, but a priority stream that waits on an event of the context's normal-priority stream while that stream
    // still has hundreds of milliseconds of work queued was seen never to wake up: they stay off unless
    // ZKHIP_MSM_STREAM_PRIORITY is set (measurements only).
    MsmWorkspace* pipe[MSM_PIPE];
    if (!stt.batch_streams[0]) {
        static const bool with_priority = getenv("ZKHIP_MSM_STREAM_PRIORITY") != nullptr;
        int least = 0, greatest = 0;
        ZK_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        // The runtime places each new stream on one of its few hardware queues (four unless GPU_MAX_HW_QUEUES says
        // otherwise): the first streams of a process open the queues one by one, each later one goes to the queue with
        // the fewest streams, which hands the queues out in turn d, c, b, a, d, ...  Which streams share a queue decides
        // the batch rate: with both sort streams beside an accumulate stream on one queue, as creation in index order
        // gave in a process that had made one stream more beforehand, an MSM of a batch takes 0.79 ms instead of 0.72.
        // So four streams that stay idle are created first, which opens every queue whatever the process did before,
        // and then the others queue by queue, with more idle streams where a queue's turn comes and no stream is left
        // for it.  The layout, one letter per stream in index order (ZKHIP_MSM_LAYOUT overrides it, measurements):
        //   d: accumulate (even), tails 0, 4, 8     c: sort (even), tails 1, 5, 9
        //   b: accumulate (odd), tails 2, 6         a: sort (odd), tails 3, 7
        const char* layout = getenv("ZKHIP_MSM_LAYOUT");
        if (!layout || strlen(layout) != (size_t)MsmState::N_STREAMS || strspn(layout, "abcd") != (size_t)MsmState::N_STREAMS)
            layout = "cdabdcbadcbadc";
        static_assert(sizeof("cdabdcbadcbadc") - 1 == MsmState::N_STREAMS, "one letter per stream");
        // (all tails on the queues of the accumulate streams: the same within 1 %; all on those of the sort streams:
        // dense columns 2 % slower, byte columns 20 to 30 %)
        auto create = [&](hipStream_t* st, int i) {
            if (with_priority) return hipStreamCreateWithPriority(st, hipStreamNonBlocking, (i == 1 || i == 3) ? least : greatest);
            return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
        };
        for (int k = 0; k < 4; k++) {
            hipStream_t idle;
            ZK_HIP(ctx, create(&idle, -1));
            stt.idle_streams.push_back(idle);
        }
        for (int placed = 0, turn = 0; placed < MsmState::N_STREAMS; turn++) {
            const char queue = "dcba"[turn % 4];
            int i = 0;
            while (i < MsmState::N_STREAMS && (stt.batch_streams[i] || layout[i] != queue)) i++;
            if (i < MsmState::N_STREAMS) {
                ZK_HIP(ctx, create(&stt.batch_streams[i], i));
                placed++;
            } else {
                hipStream_t idle;
                ZK_HIP(ctx, create(&idle, -1));
                stt.idle_streams.push_back(idle);
            }
        }
    }
    size_t n_max = 1;
    for (const MsmItem& it : items) n_max = std::max(n_max, it.n);
    const size_t fit = WS_BUDGET_BYTES / ws_bytes_estimate(ctx, tb, n_max);
    const size_t lag = std::min<size_t>(MSM_TAIL_LAG, fit > 3 ? fit - 2 : 1);
    const size_t depth = 2 + lag;   // workspaces in the rotation
    for (int i = 0; i < MSM_PIPE; i++) {
        // Two sort streams, for the workspaces of even and odd MSMs: the sort of a 2^19 column is five kernels of which
        // none fills the device, so two of them overlap well (64 distinct dense columns 0.79 -> 0.765 ms per MSM, sparse
        // columns 0.18 -> 0.148; byte columns, whose heavy buckets keep the tails busy, 0.232 -> 0.246).
        // ZKHIP_MSM_TWO_SORT=0 puts every sort on one stream (measurements).
        static const bool two_sort = env_u32("ZKHIP_MSM_TWO_SORT", 1) != 0;
        stt.ws[i].stream = stt.batch_streams[(two_sort && (i & 1)) ? 2 : 0];
        // ... and two accumulate streams: the accumulate kernel of MSM j + 1 starts while that of MSM j drains (byte
        // columns 0.26 -> 0.198 ms per MSM, dense cache-resident columns 0.774 -> 0.733, dense columns from HBM unchanged).
        // ZKHIP_MSM_TWO_ACC=0: one accumulate stream (measurements).
        static const bool two_acc = env_u32("ZKHIP_MSM_TWO_ACC", 1) != 0;
        stt.ws[i].acc_stream = stt.batch_streams[(two_acc && (i & 1)) ? 3 : 1];
        stt.ws[i].tail_stream = stt.batch_streams[4 + i];
        pipe[i] = &stt.ws[i];
    }
    ZK_HIP(ctx, hipEventRecord(stt.ev_in, ctx->stream));
    for (int i = 0; i < MsmState::N_STREAMS; i++) ZK_HIP(ctx, hipStreamWaitEvent(stt.batch_streams[i], stt.ev_in, 0));
    // software pipeline: phase1(j), phase2(j - 1), phase3(j - 1 - MSM_TAIL_LAG)
    int rc;
    // a failed step leaves work of earlier members queued on the batch's streams: it is waited for before the error
    // goes back, so that the caller may free the columns and the context's stream never runs beside leftovers
    auto drained = [&](int status) {
        for (int i = 0; i < MsmState::N_STREAMS; i++) (void)hipStreamSynchronize(stt.batch_streams[i]);
        return status;
    };
    const size_t lag3 = 1 + lag;
    for (size_t j = 0; j < batch + lag3; j++) {
        if (j < batch) {
            MsmWorkspace& ws = *pipe[j % depth];
            {
                TimedRegion tr(ctx, timed ? "msm_sort" : nullptr, ws.stream);
                rc = msm_phase1(ctx, ws, tb, items[j].first, items[j].d_scalars, items[j].n);
            }
            if (!rc) {
                TimedRegion tr(ctx, timed ? "msm_accumulate" : nullptr, ws.acc());
                rc = msm_accumulate(ctx, ws, tb);
            }
            if (rc) return drained(rc);
        }
        if (j >= 1 && j - 1 < batch) {
            rc = msm_phase2(ctx, *pipe[(j - 1) % depth], tb, timed);
            if (rc) return drained(rc);
        }
        if (j >= lag3 && j - lag3 < batch) {
            rc = msm_phase3(ctx, *pipe[(j - lag3) % depth], out + 12 * (j - lag3));
            if (rc) return drained(rc);
        }
    }
    for (int i = 0; i < MsmState::N_STREAMS; i++) {
        ZK_HIP(ctx, hipEventRecord(stt.ev_out[i], stt.batch_streams[i]));
        ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, stt.ev_out[i], 0));
    }
    return ZKHIP_OK;
}

int msm_run(zkhip_ctx* ctx, MsmState& stt, const MsmTables& tb, size_t first, const fe* d_scalars, size_t n,
            uint64_t out_jacobian[12]) {
    memset(out_jacobian, 0, 96);
    if (n == 0) return ZKHIP_OK;
    int rc = check_args(ctx, tb, first, n);
    if (rc) return rc;
    MsmWorkspace& ws = stt.ws[0];
    ws.stream = ctx->stream;
    ws.acc_stream = ws.tail_stream = nullptr;      // a lone MSM has nothing to overlap with: one stream
    {
        TimedRegion tr(ctx, "msm_sort");
        rc = msm_phase1(ctx, ws, tb, first, d_scalars, n);
    }
    if (!rc) {
        TimedRegion tr(ctx, "msm_accumulate");
        rc = msm_accumulate(ctx, ws, tb);
    }
    if (!rc) rc = msm_phase2(ctx, ws, tb, true);
    if (!rc) rc = msm_phase3(ctx, ws, out_jacobian);
    return rc;
}

int msm_run_batch(zkhip_ctx* ctx, MsmState& stt, const MsmTables& tb, const uint64_t* const* d_scalars, size_t batch, size_t n,
                  uint64_t* out) {
    memset(out, 0, batch * 96);
    if (n == 0 || batch == 0) return ZKHIP_OK;
    int rc = check_args(ctx, tb, 0, n);
    if (rc) return rc;
    std::vector<MsmItem> items;
    for (size_t j = 0; j < batch; j++) items.push_back({0, (const fe*)d_scalars[j], n});
    return msm_pipeline(ctx, stt, tb, items, out, false);
}

void msm_drop_ctx(zkhip_ctx* ctx) {
    std::lock_guard<std::mutex> lock(registry_mutex());
    auto it = registry_unlocked().find(ctx);
    if (it == registry_unlocked().end()) return;
    MsmState& st = it->second;
    for (auto& kv : st.tables)
        if (kv.second.d_tables) (void)hipFree(kv.second.d_tables);
    for (int i = 0; i < MSM_PIPE; i++) {
        MsmWorkspace& w = st.ws[i];
        if (w.base) (void)hipFree(w.base);
        if (w.h_info) (void)hipHostFree(w.h_info);
        if (w.h_sums) (void)hipHostFree(w.h_sums);
        if (w.ev_info) (void)hipEventDestroy(w.ev_info);
        if (w.ev_sorted) (void)hipEventDestroy(w.ev_sorted);
        if (w.ev_acc) (void)hipEventDestroy(w.ev_acc);
        if (w.ev_done) (void)hipEventDestroy(w.ev_done);
    }
    for (hipStream_t idle : st.idle_streams) (void)hipStreamDestroy(idle);
    for (int i = 0; i < MsmState::N_STREAMS; i++) {
        if (st.batch_streams[i]) (void)hipStreamDestroy(st.batch_streams[i]);
        if (st.ev_out[i]) (void)hipEventDestroy(st.ev_out[i]);
    }
    if (st.ev_in) (void)hipEventDestroy(st.ev_in);
    registry_unlocked().erase(it);
}

int msm_init(zkhip_ctx* ctx) {
#define ZK_LDS_ATTR(C)                                                                                  \
    ZK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(digits_hist_kernel<C>),              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024));           \
    ZK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_coarse_kernel<C>),           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    ZK_LDS_ATTR(8) ZK_LDS_ATTR(9) ZK_LDS_ATTR(10) ZK_LDS_ATTR(11) ZK_LDS_ATTR(12) ZK_LDS_ATTR(13) ZK_LDS_ATTR(14)
    ZK_LDS_ATTR(15) ZK_LDS_ATTR(16)
#undef ZK_LDS_ATTR
    return ZKHIP_OK;
}

}  // namespace zkhip

extern "C" {

int zkhip_register_bases(zkhip_ctx* ctx, const uint64_t* affine_xy, size_t n, uint32_t* out_id) {
    if (!ctx || !out_id) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, affine_xy && n >= 1, "register_bases: null pointer or n = 0");
    ZK_CHECK(ctx, n <= ((size_t)1 << ENTRY_IDX_BITS), "register_bases: n=%zu exceeds 2^26", n);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    MsmTables tb;
    tb.n = n;
    const uint32_t ln = log2_ceil(n);
    tb.c = std::min<uint32_t>(16, std::max<uint32_t>(8, ln >= 3 ? ln - 3 : 0));   // W = ceil(255/c) <= 32 windows
    tb.W = (255 + tb.c - 1) / tb.c;
    ZK_HIP(ctx, hipMalloc((void**)&tb.d_tables, (size_t)tb.W * n * sizeof(g1_affine)));
    hipError_t e = hipMemcpyAsync(tb.d_tables, affine_xy, n * sizeof(g1_affine), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        for (uint32_t w = 1; w < tb.W; w++)
            hipLaunchKernelGGL(table_next_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               tb.d_tables + (size_t)(w - 1) * n, tb.d_tables + (size_t)w * n, n, tb.c);
        const size_t total = (size_t)tb.W * n;
        hipLaunchKernelGGL(table_widen_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, tb.d_tables,
                           total);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(tb.d_tables);
        return set_error(ctx, ZKHIP_ERR_HIP, "register_bases failed: %s", hipGetErrorString(e));
    }
    const uint32_t id = ctx->next_bases_id++;
    state_of(ctx).tables[id] = tb;
    *out_id = id;
    return ZKHIP_OK;
}

int zkhip_release_bases(zkhip_ctx* ctx, uint32_t bases_id) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    auto& m = state_of(ctx).tables;
    auto it = m.find(bases_id);
    ZK_CHECK(ctx, it != m.end(), "release_bases: unknown id %u", bases_id);
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(it->second.d_tables);
    m.erase(it);
    return ZKHIP_OK;
}

int zkhip_msm_range_dev(zkhip_ctx* ctx, uint32_t bases_id, size_t first, const uint64_t* d_scalars, size_t n,
                        uint64_t out_jacobian[12]) {
    if (!ctx || !out_jacobian) return ZKHIP_ERR_INVALID;
    MsmState& stt = state_of(ctx);
    auto it = stt.tables.find(bases_id);
    ZK_CHECK(ctx, it != stt.tables.end(), "msm: unknown bases id %u", bases_id);
    ZK_CHECK(ctx, d_scalars || n == 0, "msm: null scalars");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    return msm_run(ctx, stt, it->second, first, (const fe*)d_scalars, n, out_jacobian);
}

int zkhip_msm_batch_dev(zkhip_ctx* ctx, uint32_t bases_id, const uint64_t* const* d_scalars, size_t batch, size_t n,
                        uint64_t* out_jacobian) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, (d_scalars && out_jacobian) || batch == 0, "msm_batch: null pointer");
    MsmState& stt = state_of(ctx);
    auto it = stt.tables.find(bases_id);
    ZK_CHECK(ctx, it != stt.tables.end(), "msm: unknown bases id %u", bases_id);
    for (size_t b = 0; b < batch; b++) ZK_CHECK(ctx, d_scalars[b] || n == 0, "msm_batch: null scalars");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    return msm_run_batch(ctx, stt, it->second, d_scalars, batch, n, out_jacobian);
}

int zkhip_msm(zkhip_ctx* ctx, uint32_t bases_id, const uint64_t* scalars, size_t n, uint64_t out_jacobian[12]) {
    if (!ctx || !out_jacobian) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, scalars || n == 0, "msm: null scalars");
    memset(out_jacobian, 0, 96);
    if (n == 0) return ZKHIP_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    void* d = nullptr;
    ZK_HIP(ctx, hipMalloc(&d, n * 32));
    hipError_t e = hipMemcpyAsync(d, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream);
    int rc = ZKHIP_OK;
    if (e == hipSuccess) rc = zkhip_msm_range_dev(ctx, bases_id, 0, (const uint64_t*)d, n, out_jacobian);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return set_error(ctx, ZKHIP_ERR_HIP, "msm upload failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return set_error(ctx, ZKHIP_ERR_HIP, "msm failed: %s", hipGetErrorString(e2));
    return ZKHIP_OK;
}

int zkhip_g1_add_affine(zkhip_ctx* ctx, const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* out_xy, size_t n) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, n == 0 || (p_xy && q_xy && out_xy), "g1_add_affine: null pointer");
    if (n == 0) return ZKHIP_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_scratch(ctx, 3 * n * 64);
    if (rc) return rc;
    g1_affine* dp = (g1_affine*)ctx->scratch;
    g1_affine* dq = dp + n;
    g1_affine* dout = dq + n;
    ZK_HIP(ctx, hipMemcpyAsync(dp, p_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(dq, q_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(g1_add_affine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dp, dq, dout, n);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipMemcpyAsync(out_xy, dout, n * 64, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

int zkhip_g1_multiples(zkhip_ctx* ctx, const uint64_t base_xy[8], uint64_t first, size_t n, uint64_t* out_xy) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, base_xy && (out_xy || n == 0), "g1_multiples: null pointer");
    if (n == 0) return ZKHIP_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_scratch(ctx, (n + 1) * 64);
    if (rc) return rc;
    g1_affine* dbase = (g1_affine*)ctx->scratch;
    g1_affine* dout = dbase + 1;
    ZK_HIP(ctx, hipMemcpyAsync(dbase, base_xy, 64, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(g1_multiples_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dbase, first, n, dout);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipMemcpyAsync(out_xy, dout, n * 64, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

int zkhip_g1_sum_jacobian(const uint64_t* jacobians, size_t count, uint64_t out_jacobian[12]) {
    if (!out_jacobian || (!jacobians && count)) return ZKHIP_ERR_INVALID;
    uint64_t acc[12];
    memset(acc, 0, sizeof acc);
    for (size_t i = 0; i < count; i++) h_g1_add(acc, acc, jacobians + 12 * i);
    memcpy(out_jacobian, acc, 96);
    return ZKHIP_OK;
}

int zkhip_g1_to_affine(const uint64_t jacobian[12], uint64_t out_xy[8]) {
    if (!jacobian || !out_xy) return ZKHIP_ERR_INVALID;
    h_g1_to_affine(out_xy, jacobian);
    return ZKHIP_OK;
}

}  // extern "C"
