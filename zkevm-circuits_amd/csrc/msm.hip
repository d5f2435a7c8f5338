// Pippenger bucket MSM over BN254 G1 for gfx950 (`arithmetic::best_multiexp`).
//
// Registration (once per SRS): for window width c and W = ceil(255 / c) windows the library stores the
// affine tables T_w[i] = 2^(c w) * G_i for every w.  With 288 GB of HBM the W-fold copy is cheap (k = 19:
// 512 MiB) and it collapses all windows into ONE set of 2^(c-1) signed-digit buckets: no per-window bucket
// sets, no doubling ladder at the end, 16x less bucket-reduction work.
//
// Per MSM:
//   1. digits + histogram : scalars leave Montgomery form, are recoded into W signed c-bit digits; each block
//                           counts its digits per bucket in an LDS-private histogram (heavy buckets such as
//                           "digit 1" of witness-like scalars never touch a contended global atomic)
//   2. scan               : exclusive scan of the [bucket][block] count matrix -> scatter offsets
//   3. scatter            : the same blocks place (point, window, sign) entries with LDS-private cursors
//   4. accumulate         : one lane per (bucket, slot of <= L entries): gathers table points, XYZZ mixed adds
//   5. merge              : per bucket, the slot partial sums are folded F at a time until one remains
//   6. reduce             : every lane multiplies its bucket sum by its weight (b + 1) and a block tree-sum in
//                           LDS folds 512 buckets; the <= 64 block sums are added on the host
#include <algorithm>
#include <cstring>
#include "common.h"
#include "curve.cuh"

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
ZK_DEV uint32_t bits_at(const fe& v, uint32_t pos, uint32_t c) {
    if (pos >= 256) return 0;
    const uint32_t w = pos >> 5, s = pos & 31;
    uint64_t x = v.l[w];
    if (w + 1 < 8) x |= (uint64_t)v.l[w + 1] << 32;
    return (uint32_t)(x >> s) & ((1u << c) - 1);
}

struct DigitIter {
    fe v;
    uint32_t carry = 0;
    // returns signed digit d in (-2^(c-1), 2^(c-1)] for window w (call with w = 0, 1, ... in order)
    ZK_DEV int32_t next(uint32_t w, uint32_t c) {
        uint32_t raw = bits_at(v, w * c, c) + carry;
        if (raw > (1u << (c - 1))) {
            carry = 1;
            return (int32_t)raw - (int32_t)(1u << c);
        }
        carry = 0;
        return (int32_t)raw;
    }
};

__global__ void __launch_bounds__(HIST_THREADS) digits_hist_kernel(const fe* scalars, size_t n, size_t chunk, uint32_t c,
                                                                    uint32_t W, uint32_t* counts /*[B][G]*/) {
    extern __shared__ uint32_t hist[];
    const uint32_t B = 1u << (c - 1), G = gridDim.x, g = blockIdx.x;
    for (uint32_t b = threadIdx.x; b < B; b += HIST_THREADS) hist[b] = 0;
    __syncthreads();
    const size_t lo = (size_t)g * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += HIST_THREADS) {
        DigitIter it;
        it.v = fr_canonical(load_fe(scalars + i));
        for (uint32_t w = 0; w < W; w++) {
            const int32_t d = it.next(w, c);
            if (d != 0) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += HIST_THREADS) counts[(size_t)b * G + g] = hist[b];
}

__global__ void __launch_bounds__(HIST_THREADS) scatter_kernel(const fe* scalars, size_t n, size_t chunk, uint32_t c,
                                                               uint32_t W, const uint32_t* offsets /*[B][G]*/,
                                                               uint32_t* entries) {
    extern __shared__ uint32_t cursor[];
    const uint32_t B = 1u << (c - 1), G = gridDim.x, g = blockIdx.x;
    for (uint32_t b = threadIdx.x; b < B; b += HIST_THREADS) cursor[b] = offsets[(size_t)b * G + g];
    __syncthreads();
    const size_t lo = (size_t)g * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += HIST_THREADS) {
        DigitIter it;
        it.v = fr_canonical(load_fe(scalars + i));
        for (uint32_t w = 0; w < W; w++) {
            const int32_t d = it.next(w, c);
            if (d != 0) {
                const uint32_t pos = atomicAdd(&cursor[(d < 0 ? -d : d) - 1], 1u);
                entries[pos] = (uint32_t)i | (w << ENTRY_IDX_BITS) | (d < 0 ? 0x80000000u : 0u);
            }
        }
    }
}

// ---- scans -----------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds /*[blockDim]*/, uint32_t* total) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    lds[tid] = v;
    __syncthreads();
    for (uint32_t off = 1; off < nt; off <<= 1) {
        uint32_t t = (tid >= off) ? lds[tid - off] : 0;
        __syncthreads();
        lds[tid] += t;
        __syncthreads();
    }
    const uint32_t incl = lds[tid];
    if (total) *total = lds[nt - 1];
    __syncthreads();
    return incl - v;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_tile_sums_kernel(const uint32_t* in, size_t n, uint32_t* tile_sums) {
    __shared__ uint32_t lds[SCAN_THREADS];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) s += in[base + k];
    uint32_t total;
    block_exclusive_scan(s, lds, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
// single block: exclusive scan of `m` values in place (m arbitrary), total to *total_out
__global__ void __launch_bounds__(1024) scan_single_block_kernel(uint32_t* data, size_t m, uint32_t* total_out) {
    __shared__ uint32_t lds[1024];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (size_t base = 0; base < m; base += 1024) {
        const size_t i = base + threadIdx.x;
        const uint32_t v = i < m ? data[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, lds, &total);
        const uint32_t carry = carry_s;
        if (i < m) data[i] = ex + carry;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}
__global__ void __launch_bounds__(SCAN_THREADS) scan_apply_kernel(const uint32_t* in, size_t n, const uint32_t* tile_offsets,
                                                                  uint32_t* out) {
    __shared__ uint32_t lds[SCAN_THREADS];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    uint32_t run = block_exclusive_scan(s, lds, nullptr) + tile_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

// Per-bucket bookkeeping after the big scan.  One block.
//   cnt[b]  = number of entries in bucket b, start[b] = first entry
//   slots[b] = ceil(cnt / L), slot_off = exclusive scan of slots; info = {total entries, max cnt, total slots}
__global__ void __launch_bounds__(1024) bucket_plan_kernel(const uint32_t* offsets /*[B][G]*/, uint32_t G, uint32_t B,
                                                           uint32_t total_entries_last_count, const uint32_t* counts,
                                                           uint32_t L, uint32_t* start, uint32_t* cnt, uint32_t* slot_off,
                                                           uint32_t* info) {
    __shared__ uint32_t lds[1024];
    __shared__ uint32_t carry_s, max_s;
    if (threadIdx.x == 0) { carry_s = 0; max_s = 0; }
    __syncthreads();
    (void)total_entries_last_count;
    // total entries = offsets[last] + counts[last]
    const uint32_t total = offsets[(size_t)B * G - 1] + counts[(size_t)B * G - 1];
    for (uint32_t base = 0; base < B; base += 1024) {
        const uint32_t b = base + threadIdx.x;
        uint32_t s = 0, cn = 0;
        if (b < B) {
            const uint32_t st = offsets[(size_t)b * G];
            const uint32_t en = (b + 1 < B) ? offsets[(size_t)(b + 1) * G] : total;
            cn = en - st;
            start[b] = st;
            cnt[b] = cn;
            s = (cn + L - 1) / L;
            atomicMax(&max_s, cn);
        }
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan(s, lds, &tot);
        const uint32_t carry = carry_s;
        if (b < B) slot_off[b] = ex + carry;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        slot_off[B] = carry_s;
        info[0] = total;
        info[1] = max_s;
        info[2] = carry_s;
    }
}

// next merge level: cnt_out = ceil(cnt_in / F), off_out = exclusive scan.  One block.
__global__ void __launch_bounds__(1024) merge_plan_kernel(const uint32_t* cnt_in, uint32_t B, uint32_t F, uint32_t* cnt_out,
                                                          uint32_t* off_out) {
    __shared__ uint32_t lds[1024];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < B; base += 1024) {
        const uint32_t b = base + threadIdx.x;
        uint32_t s = 0;
        if (b < B) {
            s = (cnt_in[b] + F - 1) / F;
            cnt_out[b] = s;
        }
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan(s, lds, &tot);
        const uint32_t carry = carry_s;
        if (b < B) off_out[b] = ex + carry;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) off_out[B] = carry_s;
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
__global__ void __launch_bounds__(256) accumulate_kernel(const uint32_t* entries, const uint32_t* start, const uint32_t* cnt,
                                                         const uint32_t* slot_off, uint32_t B, uint32_t L,
                                                         const g1_affine* tables, size_t table_stride, size_t first,
                                                         g1_xyzz* items) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= slot_off[B]) return;
    const uint32_t b = find_bucket(slot_off, B, t);
    const uint32_t s = t - slot_off[b];
    const uint32_t lo = start[b] + s * L;
    const uint32_t end = start[b] + cnt[b];
    const uint32_t hi = lo + L < end ? lo + L : end;
    g1_xyzz acc = xyzz_identity();
    uint32_t e = entries[lo];
    g1_affine p = load_affine(tables + (size_t)((e >> ENTRY_IDX_BITS) & 31) * table_stride + first + (e & ENTRY_IDX_MASK));
    for (uint32_t k = lo; k < hi; k++) {
        const uint32_t ecur = e;
        g1_affine q = p;
        if (k + 1 < hi) {   // prefetch the next point while this one is added
            e = entries[k + 1];
            p = load_affine(tables + (size_t)((e >> ENTRY_IDX_BITS) & 31) * table_stride + first + (e & ENTRY_IDX_MASK));
        }
        if (ecur & 0x80000000u) q.y = Fq::neg(q.y);
        xyzz_add_affine(acc, q);
    }
    store_xyzz(items + t, acc);
}

// one lane folds up to F consecutive items of one bucket
__global__ void __launch_bounds__(256) merge_kernel(const g1_xyzz* in, const uint32_t* off_in, const uint32_t* cnt_in,
                                                    const uint32_t* off_out, uint32_t B, uint32_t F, g1_xyzz* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= off_out[B]) return;
    const uint32_t b = find_bucket(off_out, B, t);
    const uint32_t j = t - off_out[b];
    const uint32_t lo = off_in[b] + j * F;
    const uint32_t end = off_in[b] + cnt_in[b];
    const uint32_t hi = lo + F < end ? lo + F : end;
    g1_xyzz acc = load_xyzz(in + lo);
    for (uint32_t k = lo + 1; k < hi; k++) acc = xyzz_add(acc, load_xyzz(in + k));
    store_xyzz(out + t, acc);
}

// ---- reduce ----------------------------------------------------------------------------------------
constexpr int REDUCE_THREADS = 512;
// weight * p by double-and-add from the top bit
ZK_DEV g1_xyzz xyzz_mul_small(const g1_xyzz& p, uint32_t k) {
    g1_xyzz acc = xyzz_identity();
    if (k == 0 || xyzz_is_identity(p)) return acc;
    const int top = 31 - __builtin_clz(k);
    for (int bit = top; bit >= 0; bit--) {
        acc = xyzz_double(acc);
        if ((k >> bit) & 1) acc = xyzz_add(acc, p);
    }
    return acc;
}

__global__ void __launch_bounds__(REDUCE_THREADS) bucket_reduce_kernel(const g1_xyzz* items, const uint32_t* off,
                                                                       const uint32_t* cnt, uint32_t B, g1_xyzz* block_sums) {
    extern __shared__ __align__(16) unsigned char smem[];
    g1_xyzz* sh = reinterpret_cast<g1_xyzz*>(smem);
    const uint32_t b = blockIdx.x * REDUCE_THREADS + threadIdx.x;
    g1_xyzz v = xyzz_identity();
    if (b < B && cnt[b]) v = xyzz_mul_small(load_xyzz(items + off[b]), b + 1);
    for (uint32_t half = REDUCE_THREADS / 2; half >= 1; half >>= 1) {
        if (threadIdx.x >= half && threadIdx.x < 2 * half) sh[threadIdx.x - half] = v;
        __syncthreads();
        if (threadIdx.x < half) v = xyzz_add(v, sh[threadIdx.x]);
        __syncthreads();
    }
    if (threadIdx.x == 0) store_xyzz(block_sums + blockIdx.x, v);
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

static void xyzz_to_jacobian_host(uint64_t out[12], const uint64_t in[16]) {
    const uint64_t *X = in, *Y = in + 4, *ZZ = in + 8, *ZZZ = in + 12;
    if (h_is_zero(ZZ)) {
        memset(out, 0, 96);
        return;
    }
    uint64_t zzz2[4], zz2[4], zz3[4], t[4];
    h_mul(HFQ, zzz2, ZZZ, ZZZ);
    h_mul(HFQ, zz2, ZZ, ZZ);
    h_mul(HFQ, zz3, zz2, ZZ);
    h_mul(HFQ, t, X, ZZ);
    h_mul(HFQ, out, t, zzz2);          // X' = X * ZZ * ZZZ^2
    h_mul(HFQ, t, Y, zz3);
    h_mul(HFQ, out + 4, t, zzz2);      // Y' = Y * ZZ^3 * ZZZ^2
    h_mul(HFQ, out + 8, ZZ, ZZZ);      // Z' = ZZ * ZZZ
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int exclusive_scan_u32(zkhip_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, uint32_t* tile_sums) {
    const size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, in, n, tile_sums);
    hipLaunchKernelGGL(scan_single_block_kernel, dim3(1), dim3(1024), 0, ctx->stream, tile_sums, tiles, (uint32_t*)nullptr);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, in, n, tile_sums, out);
    ZK_HIP(ctx, hipGetLastError());
    return ZKHIP_OK;
}

int msm_run(zkhip_ctx* ctx, const MsmTables& tb, size_t first, const fe* d_scalars, size_t n, uint64_t out_jacobian[12]) {
    memset(out_jacobian, 0, 96);
    if (n == 0) return ZKHIP_OK;
    ZK_CHECK(ctx, first + n <= tb.n, "msm: range [%zu, %zu) exceeds the %zu registered bases", first, first + n, tb.n);
    ZK_CHECK(ctx, n <= ((size_t)1 << ENTRY_IDX_BITS), "msm: n=%zu exceeds 2^26", n);
    const uint32_t c = tb.c, W = tb.W, B = 1u << (c - 1);
    const uint32_t L = 32, F = 16;
    uint32_t G = (uint32_t)std::min<size_t>((size_t)ctx->num_cus, (n + HIST_THREADS - 1) / HIST_THREADS);
    if (G < 1) G = 1;
    const size_t chunk = (n + G - 1) / G;
    const size_t E_max = n * W;
    const size_t T1_max = E_max / L + B + 1;

    // scratch layout
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    const size_t o_counts = take((size_t)B * G * 4), o_offsets = take((size_t)B * G * 4);
    const size_t o_tiles = take((((size_t)B * G + SCAN_TILE - 1) / SCAN_TILE + 1) * 4);
    const size_t o_entries = take(E_max * 4);
    const size_t o_start = take((size_t)B * 4), o_cnt = take((size_t)B * 4);
    const size_t o_offA = take(((size_t)B + 1) * 4), o_cntA = take((size_t)B * 4);
    const size_t o_offB = take(((size_t)B + 1) * 4), o_cntB = take((size_t)B * 4);
    const size_t o_info = take(64);
    const size_t o_itemsA = take(T1_max * sizeof(g1_xyzz));
    const size_t o_itemsB = take((T1_max / F + B + 1) * sizeof(g1_xyzz));
    const size_t n_red_blocks = (B + REDUCE_THREADS - 1) / REDUCE_THREADS;
    const size_t o_bsums = take(n_red_blocks * sizeof(g1_xyzz));
    int rc = ensure_scratch(ctx, off);
    if (rc) return rc;
    unsigned char* base = (unsigned char*)ctx->scratch;
    uint32_t* counts = (uint32_t*)(base + o_counts);
    uint32_t* offsets = (uint32_t*)(base + o_offsets);
    uint32_t* tiles = (uint32_t*)(base + o_tiles);
    uint32_t* entries = (uint32_t*)(base + o_entries);
    uint32_t* start = (uint32_t*)(base + o_start);
    uint32_t* cnt = (uint32_t*)(base + o_cnt);
    uint32_t* offA = (uint32_t*)(base + o_offA);
    uint32_t* cntA = (uint32_t*)(base + o_cntA);
    uint32_t* offB = (uint32_t*)(base + o_offB);
    uint32_t* cntB = (uint32_t*)(base + o_cntB);
    uint32_t* info = (uint32_t*)(base + o_info);
    g1_xyzz* itemsA = (g1_xyzz*)(base + o_itemsA);
    g1_xyzz* itemsB = (g1_xyzz*)(base + o_itemsB);
    g1_xyzz* bsums = (g1_xyzz*)(base + o_bsums);

    const size_t lds_hist = (size_t)B * 4;
    {
        TimedRegion tr(ctx, "msm_sort");
        hipLaunchKernelGGL(digits_hist_kernel, dim3(G), dim3(HIST_THREADS), lds_hist, ctx->stream, d_scalars, n, chunk, c, W,
                           counts);
        ZK_HIP(ctx, hipGetLastError());
        rc = exclusive_scan_u32(ctx, counts, offsets, (size_t)B * G, tiles);
        if (rc) return rc;
        hipLaunchKernelGGL(bucket_plan_kernel, dim3(1), dim3(1024), 0, ctx->stream, offsets, G, B, 0u, counts, L, start, cnt,
                           offA, info);
        hipLaunchKernelGGL(scatter_kernel, dim3(G), dim3(HIST_THREADS), lds_hist, ctx->stream, d_scalars, n, chunk, c, W,
                           offsets, entries);
        ZK_HIP(ctx, hipGetLastError());
    }
    uint32_t h_info[3];
    ZK_HIP(ctx, hipMemcpyAsync(h_info, info, sizeof h_info, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t total_entries = h_info[0], max_cnt = h_info[1], total_slots = h_info[2];
    if (total_entries == 0) return ZKHIP_OK;   // every scalar is zero

    {
        TimedRegion tr(ctx, "msm_accumulate");
        hipLaunchKernelGGL(accumulate_kernel, dim3((total_slots + 255) / 256), dim3(256), 0, ctx->stream, entries, start, cnt,
                           offA, B, L, tb.d_tables, tb.n, first, itemsA);
        ZK_HIP(ctx, hipGetLastError());
    }
    // slot partial sums -> one item per bucket
    g1_xyzz *cur_items = itemsA, *nxt_items = itemsB;
    uint32_t *cur_off = offA, *cur_cnt = cntA, *nxt_off = offB, *nxt_cnt = cntB;
    {
        TimedRegion tr(ctx, "msm_merge");
        // cntA = slots per bucket
        hipLaunchKernelGGL(merge_plan_kernel, dim3(1), dim3(1024), 0, ctx->stream, cnt, B, L, cntA, offA);
        uint32_t level_max = (max_cnt + L - 1) / L;
        uint32_t level_total = total_slots;
        while (level_max > 1) {
            hipLaunchKernelGGL(merge_plan_kernel, dim3(1), dim3(1024), 0, ctx->stream, cur_cnt, B, F, nxt_cnt, nxt_off);
            const uint32_t bound = level_total / F + B + 1;
            hipLaunchKernelGGL(merge_kernel, dim3((bound + 255) / 256), dim3(256), 0, ctx->stream, cur_items, cur_off, cur_cnt,
                               nxt_off, B, F, nxt_items);
            ZK_HIP(ctx, hipGetLastError());
            std::swap(cur_items, nxt_items);
            std::swap(cur_off, nxt_off);
            std::swap(cur_cnt, nxt_cnt);
            level_max = (level_max + F - 1) / F;
            level_total = bound;
        }
    }
    {
        TimedRegion tr(ctx, "msm_reduce");
        hipLaunchKernelGGL(bucket_reduce_kernel, dim3((unsigned)n_red_blocks), dim3(REDUCE_THREADS),
                           (REDUCE_THREADS / 2) * sizeof(g1_xyzz), ctx->stream, cur_items, cur_off, cur_cnt, B, bsums);
        ZK_HIP(ctx, hipGetLastError());
    }
    std::vector<uint64_t> h_sums(n_red_blocks * 16);
    ZK_HIP(ctx, hipMemcpyAsync(h_sums.data(), bsums, n_red_blocks * sizeof(g1_xyzz), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t acc[12], jac[12];
    memset(acc, 0, sizeof acc);
    for (size_t i = 0; i < n_red_blocks; i++) {
        xyzz_to_jacobian_host(jac, &h_sums[16 * i]);
        h_g1_add(acc, acc, jac);
    }
    memcpy(out_jacobian, acc, 96);
    return ZKHIP_OK;
}

static std::map<zkhip_ctx*, std::map<uint32_t, MsmTables>>& registry() {
    static std::map<zkhip_ctx*, std::map<uint32_t, MsmTables>> r;
    return r;
}

void msm_drop_ctx(zkhip_ctx* ctx) {
    auto it = registry().find(ctx);
    if (it == registry().end()) return;
    for (auto& kv : it->second)
        if (kv.second.d_tables) (void)hipFree(kv.second.d_tables);
    registry().erase(it);
}

int msm_init(zkhip_ctx* ctx) {
    ZK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(digits_hist_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    ZK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    ZK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_reduce_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
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
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(tb.d_tables);
        return set_error(ctx, ZKHIP_ERR_HIP, "register_bases failed: %s", hipGetErrorString(e));
    }
    const uint32_t id = ctx->next_bases_id++;
    registry()[ctx][id] = tb;
    *out_id = id;
    return ZKHIP_OK;
}

int zkhip_release_bases(zkhip_ctx* ctx, uint32_t bases_id) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    auto& m = registry()[ctx];
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
    auto& m = registry()[ctx];
    auto it = m.find(bases_id);
    ZK_CHECK(ctx, it != m.end(), "msm: unknown bases id %u", bases_id);
    ZK_CHECK(ctx, d_scalars || n == 0, "msm: null scalars");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    return msm_run(ctx, it->second, first, (const fe*)d_scalars, n, out_jacobian);
}

int zkhip_msm_batch_dev(zkhip_ctx* ctx, uint32_t bases_id, const uint64_t* const* d_scalars, size_t batch, size_t n,
                        uint64_t* out_jacobian) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, (d_scalars && out_jacobian) || batch == 0, "msm_batch: null pointer");
    for (size_t b = 0; b < batch; b++) {
        int rc = zkhip_msm_range_dev(ctx, bases_id, 0, d_scalars[b], n, out_jacobian + 12 * b);
        if (rc) return rc;
    }
    return ZKHIP_OK;
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
