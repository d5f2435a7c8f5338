// The permuted columns of a lookup argument on the device: `plonk::lookup::prover::permute_expression_pair` of the
// halo2 prover (called from `Argument::commit_permuted`, inside create_proof: reached from
// zkevm-circuits/benches/benchmark_list/main.rs:265 and prover/src/chunk/prover.rs:85).
//
// Upstream, over the first `usable_rows` rows of the compressed input A and table S:
//   A' = A sorted (Fr's Ord: by canonical integer value);
//   S'[row] = A'[row] wherever A'[row] is the first occurrence of its value, and that value is taken once out of the
//   multiset of table values (an input value missing from the table is an error);
//   the table values left over, in ascending order, fill the remaining rows from the LAST repeated row backwards
//   (`repeated_input_rows.pop()` per leftover value).
// The same result follows from two sorts and two prefix sums: with T = S sorted, the instance removed for a distinct
// input value is the first element of its run in T; the k-th leftover element of T goes to the k-th repeated row
// counted from the end.  The last n - usable_rows rows are blinding rows drawn from the prover's RNG and are left to
// the caller.
//
// All lookups of a proof go through ONE sequence of launches (grid.y = column): 2 L columns of m rows each.
// The sort is a least-significant-digit radix sort of (48-bit key, row) pairs, eight bits per pass:
//   * a column's key is the 48-bit window of the canonical value that ends at its most significant VARYING bit (the OR and
//     the AND of all values of a column tell which bits vary).  Equal values need no order among themselves, so the window
//     decides the order of a whole column unless two DIFFERENT values agree on all 48 bits: random field elements (what a
//     theta-compressed expression is) do so with probability 2^-48 per pair, values below 2^48 never;
//   * passes whose eight bits do not vary in a column are skipped for that column (small table indices take two passes, not
//     six), which every workgroup decides alike from the column's masks; the ping-pong parity follows the same rule;
//   * a pass is histogram (LDS counters per 4096-key tile) -> per-column scan of the [digit][tile] counts -> stable scatter
//     (wave-level peer masks from eight ballots give a key's rank among equal digits, tile-level running counters in LDS);
//   * the sorted column is gathered and every element compared with its predecessor on all 256 bits; should any column be
//     out of order (colliding windows) the whole batch is re-sorted exactly, six windows from the least significant up.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "domain.h"

using namespace zkhip;

namespace zkhip {

// canonical (non-Montgomery) value as eight words, compared from the most significant word down
ZK_DEV bool canon_less(const fe& a, const fe& b) {
    for (int i = 7; i >= 0; i--) {
        if (a.l[i] != b.l[i]) return a.l[i] < b.l[i];
    }
    return false;
}
ZK_DEV bool fe_same(const fe& a, const fe& b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= a.l[i] ^ b.l[i];
    return d == 0;
}
// index of the first element >= v in the sorted array a[0..n)
ZK_DEV uint32_t lower_bound(const fe* a, uint32_t n, const fe& v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (canon_less(load_fe(a + mid), v)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

constexpr int LP_THREADS = 256;
constexpr int RS_TILE = 4096;                    // keys per workgroup and pass
constexpr int RS_ROUNDS = RS_TILE / LP_THREADS;
constexpr int RS_BINS = 256;
constexpr int RS_PASSES = 6;                     // 48-bit keys
constexpr uint32_t RS_KEY_BITS = 48;

// per column: where its key window starts and which passes have varying bits
struct SegPlan {
    uint32_t shift, active;
};
ZK_DEV uint32_t plan_parity(const SegPlan& p, int pass) { return __popc(p.active & ((1u << pass) - 1)) & 1u; }

// Montgomery -> canonical: multiply by the integer 1
ZK_DEV fe lp_canonical(const fe& a) {
    fe one;
#pragma unroll
    for (int i = 0; i < 8; i++) one.l[i] = 0;
    one.l[0] = 1;
    return Fr::mul(a, one);
}

// canon[s][i] = canonical form of column s; masks[s][0..8) |= words, masks[s][8..16) &= words
__global__ void __launch_bounds__(LP_THREADS) lp_prepare_kernel(const fe* const* src, fe* canon, uint32_t* masks, uint32_t m) {
    __shared__ uint32_t sh[16];
    const uint32_t s = blockIdx.y;
    if (threadIdx.x < 16) sh[threadIdx.x] = threadIdx.x < 8 ? 0u : 0xffffffffu;
    __syncthreads();
    uint32_t o[8], a[8];
#pragma unroll
    for (int w = 0; w < 8; w++) { o[w] = 0; a[w] = 0xffffffffu; }
    // four elements per lane, a workgroup covers 1024 consecutive rows
    for (int e = 0; e < 4; e++) {
        const uint32_t i = blockIdx.x * (LP_THREADS * 4) + e * LP_THREADS + threadIdx.x;
        if (i < m) {
            const fe c = lp_canonical(load_fe(src[s] + i));
            store_fe(canon + (size_t)s * m + i, c);
#pragma unroll
            for (int w = 0; w < 8; w++) { o[w] |= c.l[w]; a[w] &= c.l[w]; }
        }
    }
#pragma unroll
    for (int w = 0; w < 8; w++) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            o[w] |= __shfl_xor(o[w], off);
            a[w] &= __shfl_xor(a[w], off);
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int w = 0; w < 8; w++) {
            atomicOr(&sh[w], o[w]);
            atomicAnd(&sh[8 + w], a[w]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 8) atomicOr(&masks[s * 16 + threadIdx.x], sh[threadIdx.x]);
    else if (threadIdx.x < 16) atomicAnd(&masks[s * 16 + threadIdx.x], sh[threadIdx.x]);
}

// exact: shift >= 0 -> every column gets the window [shift, shift + 48) and all six passes (the exact re-sort);
// otherwise the window ends at the column's most significant varying bit and only passes with varying bits run
__global__ void lp_plan_kernel(const uint32_t* masks, SegPlan* plans, uint32_t segs, int exact_shift) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= segs) return;
    SegPlan p;
    if (exact_shift >= 0) {
        p.shift = (uint32_t)exact_shift;
        p.active = (1u << RS_PASSES) - 1;
    } else {
        uint32_t v[8];
        int top = -1;
        for (int w = 0; w < 8; w++) {
            v[w] = masks[s * 16 + w] & ~masks[s * 16 + 8 + w];
            if (v[w]) top = 32 * w + 31 - __clz(v[w]);
        }
        p.shift = top >= (int)RS_KEY_BITS ? (uint32_t)(top - (int)RS_KEY_BITS + 1) : 0u;
        p.active = 0;
        for (int pass = 0; pass < RS_PASSES; pass++) {
            // any varying bit in [shift + 8 pass, shift + 8 pass + 8)?
            bool any = false;
            for (int b = 0; b < 8; b++) {
                const uint32_t bit = p.shift + 8 * pass + b;
                if (bit < 256 && ((v[bit >> 5] >> (bit & 31)) & 1u)) any = true;
            }
            if (any) p.active |= 1u << pass;
        }
    }
    plans[s] = p;
}

// the 48 bits of a canonical value that start at bit `shift` (bits beyond 255 read as zero)
ZK_DEV uint64_t key_window(const uint32_t* words, uint32_t shift) {
    const uint32_t w = shift >> 5, sh = shift & 31;
    const uint64_t w0 = w < 8 ? words[w] : 0u, w1 = w + 1 < 8 ? words[w + 1] : 0u, w2 = w + 2 < 8 ? words[w + 2] : 0u;
    uint64_t lo = w0 | (w1 << 32);
    uint64_t v = lo >> sh;
    if (sh) v |= w2 << (64 - sh);
    return v & (((uint64_t)1 << RS_KEY_BITS) - 1);
}

// key[s][i] = window of canon[s][row], row = i (first sort) or the row that the sort so far put at position i (re-sort)
__global__ void __launch_bounds__(LP_THREADS) lp_keys_kernel(const fe* canon, const SegPlan* plans, const uint32_t* idx_in,
                                                             uint64_t* key, uint32_t* idx, uint32_t m) {
    const uint32_t s = blockIdx.y, i = blockIdx.x * LP_THREADS + threadIdx.x;
    if (i >= m) return;
    const size_t at = (size_t)s * m + i;
    const uint32_t row = idx_in ? idx_in[at] : i;
    key[at] = key_window(reinterpret_cast<const uint32_t*>(canon + (size_t)s * m + row), plans[s].shift);
    idx[at] = row;
}

// hist[s][digit][tile] = keys of the tile with that digit
__global__ void __launch_bounds__(LP_THREADS) rs_hist_kernel(const uint64_t* key0, const uint64_t* key1, const SegPlan* plans,
                                                             int pass, uint32_t* hist, uint32_t m, uint32_t ntiles) {
    __shared__ uint32_t cnt[RS_BINS];
    const uint32_t s = blockIdx.y;
    const SegPlan p = plans[s];
    if (!((p.active >> pass) & 1u)) return;
    const uint64_t* key = (plan_parity(p, pass) ? key1 : key0) + (size_t)s * m;
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_ROUNDS; r++) {
        const uint32_t i = base + r * LP_THREADS + threadIdx.x;
        if (i < m) atomicAdd(&cnt[(uint32_t)(key[i] >> (8 * pass)) & 255u], 1u);
    }
    __syncthreads();
    hist[((size_t)s * RS_BINS + threadIdx.x) * ntiles + blockIdx.x] = cnt[threadIdx.x];
}

// one workgroup per column: hist[s][digit][tile] <- number of keys that sort before the first key of (digit, tile)
__global__ void __launch_bounds__(LP_THREADS) rs_scan_kernel(const SegPlan* plans, int pass, uint32_t* hist, uint32_t ntiles) {
    __shared__ uint32_t tot[RS_BINS];
    const uint32_t s = blockIdx.x;
    if (!((plans[s].active >> pass) & 1u)) return;
    uint32_t* h = hist + ((size_t)s * RS_BINS + threadIdx.x) * ntiles;
    uint32_t sum = 0;
    for (uint32_t t = 0; t < ntiles; t++) {
        const uint32_t c = h[t];
        h[t] = sum;
        sum += c;
    }
    tot[threadIdx.x] = sum;
    __syncthreads();
    // exclusive scan of the 256 digit totals (Hillis-Steele)
    uint32_t incl = sum;
    for (uint32_t off = 1; off < RS_BINS; off <<= 1) {
        uint32_t other = 0;
        if (threadIdx.x >= off) other = tot[threadIdx.x - off];
        __syncthreads();
        incl += other;
        tot[threadIdx.x] = incl;
        __syncthreads();
    }
    const uint32_t excl = incl - sum;
    for (uint32_t t = 0; t < ntiles; t++) h[t] += excl;
}

// stable scatter of one tile: a key goes to (keys before its (digit, tile)) + (keys of the tile with its digit before it)
__global__ void __launch_bounds__(LP_THREADS) rs_scatter_kernel(const uint64_t* key0, const uint64_t* key1, const uint32_t* idx0,
                                                                const uint32_t* idx1, uint64_t* okey0, uint64_t* okey1,
                                                                uint32_t* oidx0, uint32_t* oidx1, const SegPlan* plans, int pass,
                                                                const uint32_t* hist, uint32_t m, uint32_t ntiles) {
    __shared__ uint32_t cnt[RS_BINS];
    const uint32_t s = blockIdx.y;
    const SegPlan p = plans[s];
    if (!((p.active >> pass) & 1u)) return;
    const uint32_t par = plan_parity(p, pass);
    const size_t seg = (size_t)s * m;
    const uint64_t* key = (par ? key1 : key0) + seg;
    const uint32_t* idx = (par ? idx1 : idx0) + seg;
    uint64_t* okey = (par ? okey0 : okey1) + seg;      // the other buffer
    uint32_t* oidx = (par ? oidx0 : oidx1) + seg;
    cnt[threadIdx.x] = hist[((size_t)s * RS_BINS + threadIdx.x) * ntiles + blockIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    const uint32_t base = blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_ROUNDS; r++) {
        const uint32_t i = base + r * LP_THREADS + threadIdx.x;
        const bool valid = i < m;
        uint64_t k = 0;
        uint32_t row = 0;
        if (valid) {
            k = key[i];
            row = idx[i];
        }
        const uint32_t d = (uint32_t)(k >> (8 * pass)) & 255u;
        // lanes of this wave that hold a valid key with the same digit
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t has = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? has : ~has;
        }
        const uint32_t rank = __popcll(peers & lt), count = __popcll(peers);
        uint32_t at = 0;
        // the four waves take their positions in order: a wave reads the running counters, its first lane per digit
        // advances them (LDS operations of one wave execute in program order)
        for (uint32_t w = 0; w < LP_THREADS / 64; w++) {
            if (wave == w && valid) {
                at = cnt[d] + rank;
                if (rank == 0) cnt[d] = at + count;
            }
            __syncthreads();
        }
        if (valid) {
            okey[at] = k;
            oidx[at] = row;
        }
    }
}

// sorted[s][i] = canon[s][row at position i]; a column whose neighbours are out of order on the full value is marked
__global__ void __launch_bounds__(LP_THREADS) lp_gather_kernel(const fe* canon, const uint32_t* idx0, const uint32_t* idx1,
                                                               const SegPlan* plans, fe* sorted, uint32_t* dirty, uint32_t m) {
    const uint32_t s = blockIdx.y, i = blockIdx.x * LP_THREADS + threadIdx.x;
    if (i >= m) return;
    const SegPlan p = plans[s];
    const uint32_t* idx = ((__popc(p.active) & 1) ? idx1 : idx0) + (size_t)s * m;
    const fe* c = canon + (size_t)s * m;
    const fe v = load_fe(c + idx[i]);
    store_fe(sorted + (size_t)s * m + i, v);
    if (i && canon_less(v, load_fe(c + idx[i - 1]))) atomicOr(dirty, 1u);
}

// per lookup l (columns 2l = input, 2l + 1 = table, both sorted): rep[i] = 1 where A'[i] repeats A'[i-1];
// left[j] = 1 where T[j] is NOT the instance taken out for an input value; bad[l] is raised when a distinct input value
// does not occur in the table
// (flag array 2 l = rep of lookup l, array 2 l + 1 = left, each m words)
__global__ void __launch_bounds__(LP_THREADS) lp_flags_kernel(const fe* sorted, uint32_t m, uint32_t* flags, uint32_t* bad) {
    const uint32_t l = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const fe* as = sorted + (size_t)(2 * l) * m;
    const fe* ts = as + m;
    const fe a = load_fe(as + i);
    const bool first_a = i == 0 || !fe_same(a, load_fe(as + i - 1));
    flags[(size_t)(2 * l) * m + i] = first_a ? 0u : 1u;
    if (first_a) {
        const uint32_t p = lower_bound(ts, m, a);
        if (p >= m || !fe_same(load_fe(ts + p), a)) atomicOr(bad + l, 1u);
    }
    const fe t = load_fe(ts + i);
    const bool first_t = i == 0 || !fe_same(t, load_fe(ts + i - 1));
    bool taken = false;
    if (first_t) {
        const uint32_t p = lower_bound(as, m, t);
        taken = p < m && fe_same(load_fe(as + p), t);
    }
    flags[(size_t)(2 * l + 1) * m + i] = taken ? 0u : 1u;
}

// ---- exclusive prefix sums of the two flag arrays of every lookup (arrays a = 0 / 1 of lookup l at flags + (2 l + a) m) ----
__global__ void __launch_bounds__(LP_THREADS) lp_scan_sums_kernel(const uint32_t* flags, uint32_t* part, uint32_t m, uint32_t ntiles) {
    __shared__ uint32_t sh[LP_THREADS / 64];
    const uint32_t arr = blockIdx.y;
    const uint32_t* f = flags + (size_t)arr * m;
    uint32_t sum = 0;
    for (int r = 0; r < RS_ROUNDS; r++) {
        const uint32_t i = blockIdx.x * RS_TILE + r * LP_THREADS + threadIdx.x;
        if (i < m) sum += f[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)arr * ntiles + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// one workgroup per array: part[arr][tile] <- sum of the tiles before it; total[arr] = sum of all
__global__ void __launch_bounds__(LP_THREADS) lp_scan_tiles_kernel(uint32_t* part, uint32_t* total, uint32_t ntiles) {
    __shared__ uint32_t sh[LP_THREADS];
    uint32_t* p = part + (size_t)blockIdx.x * ntiles;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += LP_THREADS) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = t < ntiles ? p[t] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        uint32_t incl = v;
        for (uint32_t off = 1; off < LP_THREADS; off <<= 1) {
            uint32_t other = 0;
            if (threadIdx.x >= off) other = sh[threadIdx.x - off];
            __syncthreads();
            incl += other;
            sh[threadIdx.x] = incl;
            __syncthreads();
        }
        if (t < ntiles) p[t] = carry + incl - v;
        carry += sh[LP_THREADS - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[blockIdx.x] = carry;
}
// out[arr][i] = number of set flags before i; a lane owns 16 consecutive flags of the tile
__global__ void __launch_bounds__(LP_THREADS) lp_scan_apply_kernel(const uint32_t* flags, const uint32_t* part, uint32_t* out,
                                                                   uint32_t m, uint32_t ntiles) {
    __shared__ uint32_t sh[LP_THREADS];
    const uint32_t arr = blockIdx.y;
    const uint32_t* f = flags + (size_t)arr * m;
    uint32_t* o = out + (size_t)arr * m;
    const uint32_t first = blockIdx.x * RS_TILE + threadIdx.x * RS_ROUNDS;
    uint32_t v[RS_ROUNDS], sum = 0;
#pragma unroll
    for (int e = 0; e < RS_ROUNDS; e++) {
        v[e] = first + e < m ? f[first + e] : 0;
        sum += v[e];
    }
    sh[threadIdx.x] = sum;
    __syncthreads();
    uint32_t incl = sum;
    for (uint32_t off = 1; off < LP_THREADS; off <<= 1) {
        uint32_t other = 0;
        if (threadIdx.x >= off) other = sh[threadIdx.x - off];
        __syncthreads();
        incl += other;
        sh[threadIdx.x] = incl;
        __syncthreads();
    }
    uint32_t run = part[(size_t)arr * ntiles + blockIdx.x] + incl - sum;
#pragma unroll
    for (int e = 0; e < RS_ROUNDS; e++) {
        if (first + e < m) o[first + e] = run;
        run += v[e];
    }
}

// rows[l][k] = k-th repeated row (ascending)
__global__ void __launch_bounds__(LP_THREADS) lp_rows_kernel(const uint32_t* flags, const uint32_t* scans, uint32_t m, uint32_t* rows) {
    const uint32_t l = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rep = (size_t)(2 * l) * m;
    if (i < m && flags[rep + i]) rows[(size_t)l * m + scans[rep + i]] = i;
}

// A' and S' in Montgomery form.  n_rep = number of repeated rows = number of leftover table values used.
__global__ void __launch_bounds__(LP_THREADS) lp_assign_kernel(const fe* sorted, const uint32_t* flags, const uint32_t* scans,
                                                               const uint32_t* totals, const uint32_t* rows, uint32_t m, fe r2,
                                                               fe* const* out_a, fe* const* out_s) {
    const uint32_t l = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const fe* as = sorted + (size_t)(2 * l) * m;
    const fe* ts = as + m;
    const size_t rep = (size_t)(2 * l) * m, left = rep + m;
    const uint32_t n_rep = totals[2 * l];
    const fe a = Fr::mul(load_fe(as + i), r2);   // canonical -> Montgomery
    store_fe(out_a[l] + i, a);
    if (!flags[rep + i]) store_fe(out_s[l] + i, a);
    if (flags[left + i]) {
        const uint32_t k = scans[left + i];
        if (k < n_rep) store_fe(out_s[l] + rows[(size_t)l * m + n_rep - 1 - k], Fr::mul(load_fe(ts + i), r2));
    }
}

// one sort of `segs` columns: keys from the windows in `plans`, then the passes; the launches are the same whatever the
// plans say (inactive passes return at once)
static int radix_sort_columns(zkhip_ctx* ctx, const fe* canon, const SegPlan* plans, const uint32_t* idx_in, uint64_t* key0,
                              uint64_t* key1, uint32_t* idx0, uint32_t* idx1, uint32_t* hist, uint32_t m, uint32_t segs) {
    hipStream_t st = ctx->stream;
    const uint32_t ntiles = (m + RS_TILE - 1) / RS_TILE;
    hipLaunchKernelGGL(lp_keys_kernel, dim3((m + LP_THREADS - 1) / LP_THREADS, segs), dim3(LP_THREADS), 0, st, canon, plans, idx_in,
                       key0, idx0, m);
    for (int pass = 0; pass < RS_PASSES; pass++) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3(ntiles, segs), dim3(LP_THREADS), 0, st, (const uint64_t*)key0, (const uint64_t*)key1,
                           plans, pass, hist, m, ntiles);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(segs), dim3(LP_THREADS), 0, st, plans, pass, hist, ntiles);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(ntiles, segs), dim3(LP_THREADS), 0, st, (const uint64_t*)key0,
                           (const uint64_t*)key1, (const uint32_t*)idx0, (const uint32_t*)idx1, key0, key1, idx0, idx1, plans, pass,
                           (const uint32_t*)hist, m, ntiles);
    }
    ZK_HIP(ctx, hipGetLastError());
    return ZKHIP_OK;
}

constexpr uint32_t LP_MAX_BATCH = 64;   // lookups per sequence of launches (about 100 MB of scratch per lookup at 2^19 rows)

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// one chunk of at most LP_MAX_BATCH lookups; `first` = index of the chunk's first lookup in the caller's batch (for messages)
static int lookup_permute_chunk(zkhip_ctx* ctx, uint32_t first, uint32_t m, uint32_t count, const uint64_t* const* d_inputs,
                                const uint64_t* const* d_tables, uint64_t* const* d_out_a, uint64_t* const* d_out_s) {
    hipStream_t st = ctx->stream;
    const uint32_t segs = 2 * count, ntiles = (m + RS_TILE - 1) / RS_TILE;
    const size_t sm = (size_t)segs * m;
    // scratch layout
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += align256(bytes); return at; };
    const size_t o_canon = take(sm * 32), o_sorted = take(sm * 32), o_key0 = take(sm * 8), o_key1 = take(sm * 8);
    const size_t o_idx0 = take(sm * 4), o_idx1 = take(sm * 4), o_flags = take(sm * 4), o_scans = take(sm * 4);
    const size_t o_rows = take((size_t)count * m * 4), o_hist = take((size_t)segs * RS_BINS * ntiles * 4);
    const size_t o_part = take((size_t)segs * ntiles * 4), o_totals = take((size_t)segs * 4), o_masks = take((size_t)segs * 64);
    const size_t o_plans = take((size_t)segs * sizeof(SegPlan)), o_status = take((size_t)(count + 1) * 4);
    const size_t o_ptrs = take((size_t)4 * count * sizeof(void*));
    int rc = ensure_scratch(ctx, off);
    if (rc) return rc;
    unsigned char* base = (unsigned char*)ctx->scratch;
    fe* canon = (fe*)(base + o_canon);
    fe* sorted = (fe*)(base + o_sorted);
    uint64_t *key0 = (uint64_t*)(base + o_key0), *key1 = (uint64_t*)(base + o_key1);
    uint32_t *idx0 = (uint32_t*)(base + o_idx0), *idx1 = (uint32_t*)(base + o_idx1);
    uint32_t *flags = (uint32_t*)(base + o_flags), *scans = (uint32_t*)(base + o_scans), *rows = (uint32_t*)(base + o_rows);
    uint32_t *hist = (uint32_t*)(base + o_hist), *part = (uint32_t*)(base + o_part), *totals = (uint32_t*)(base + o_totals);
    uint32_t* masks = (uint32_t*)(base + o_masks);
    SegPlan* plans = (SegPlan*)(base + o_plans);
    uint32_t* status = (uint32_t*)(base + o_status);     // [0] = some column out of order, [1 + l] = lookup l's input not in its table
    const fe** src = (const fe**)(base + o_ptrs);         // [2 l] input, [2 l + 1] table
    fe** out_a = (fe**)(src + 2 * count);
    fe** out_s = out_a + count;
    {
        std::vector<const void*> h(4 * (size_t)count);
        for (uint32_t l = 0; l < count; l++) {
            h[2 * l] = d_inputs[l];
            h[2 * l + 1] = d_tables[l];
            h[2 * count + l] = d_out_a[l];
            h[3 * count + l] = d_out_s[l];
        }
        ZK_HIP(ctx, hipMemcpyAsync(src, h.data(), h.size() * sizeof(void*), hipMemcpyHostToDevice, st));
        ZK_HIP(ctx, hipStreamSynchronize(st));   // the table is host memory that dies with this scope
    }
    std::vector<uint32_t> init_masks((size_t)segs * 16);
    for (uint32_t s = 0; s < segs; s++)
        for (int w = 0; w < 16; w++) init_masks[s * 16 + w] = w < 8 ? 0u : 0xffffffffu;
    ZK_HIP(ctx, hipMemcpyAsync(masks, init_masks.data(), init_masks.size() * 4, hipMemcpyHostToDevice, st));
    ZK_HIP(ctx, hipStreamSynchronize(st));
    ZK_HIP(ctx, hipMemsetAsync(status, 0, (size_t)(count + 1) * 4, st));

    const dim3 rows_grid((m + LP_THREADS - 1) / LP_THREADS, segs);
    hipLaunchKernelGGL(lp_prepare_kernel, dim3((m + 4 * LP_THREADS - 1) / (4 * LP_THREADS), segs), dim3(LP_THREADS), 0, st, src, canon,
                       masks, m);
    hipLaunchKernelGGL(lp_plan_kernel, dim3((segs + 63) / 64), dim3(64), 0, st, (const uint32_t*)masks, plans, segs, -1);
    if ((rc = radix_sort_columns(ctx, canon, plans, nullptr, key0, key1, idx0, idx1, hist, m, segs))) return rc;
    hipLaunchKernelGGL(lp_gather_kernel, rows_grid, dim3(LP_THREADS), 0, st, (const fe*)canon, (const uint32_t*)idx0,
                       (const uint32_t*)idx1, (const SegPlan*)plans, sorted, status, m);
    ZK_HIP(ctx, hipGetLastError());
    uint32_t dirty = 0;
    ZK_HIP(ctx, hipMemcpyAsync(&dirty, status, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(ctx, hipStreamSynchronize(st));
    const char* force = getenv("ZKHIP_LOOKUP_EXACT_SORT");   // tests: always take the exact re-sort
    if (dirty || (force && *force == '1')) {
        // two different values of some column agree on its whole key window: sort again on all 256 bits, six windows of 48
        // bits from the least significant up, each a stable sort of the order so far
        for (int c = 0; c < 6; c++) {
            hipLaunchKernelGGL(lp_plan_kernel, dim3((segs + 63) / 64), dim3(64), 0, st, (const uint32_t*)masks, plans, segs, 48 * c);
            // six active passes leave the result in buffer 0 again, which the next window reads its rows from
            if ((rc = radix_sort_columns(ctx, canon, plans, c ? idx0 : nullptr, key0, key1, idx0, idx1, hist, m, segs))) return rc;
        }
        ZK_HIP(ctx, hipMemsetAsync(status, 0, 4, st));
        hipLaunchKernelGGL(lp_gather_kernel, rows_grid, dim3(LP_THREADS), 0, st, (const fe*)canon, (const uint32_t*)idx0,
                           (const uint32_t*)idx1, (const SegPlan*)plans, sorted, status, m);
        ZK_HIP(ctx, hipGetLastError());
    }

    const dim3 lk_grid((m + LP_THREADS - 1) / LP_THREADS, count);
    hipLaunchKernelGGL(lp_flags_kernel, lk_grid, dim3(LP_THREADS), 0, st, (const fe*)sorted, m, flags, status + 1);
    ZK_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(lp_scan_sums_kernel, dim3(ntiles, segs), dim3(LP_THREADS), 0, st, (const uint32_t*)flags, part, m, ntiles);
    hipLaunchKernelGGL(lp_scan_tiles_kernel, dim3(segs), dim3(LP_THREADS), 0, st, part, totals, ntiles);
    hipLaunchKernelGGL(lp_scan_apply_kernel, dim3(ntiles, segs), dim3(LP_THREADS), 0, st, (const uint32_t*)flags, (const uint32_t*)part,
                       scans, m, ntiles);
    hipLaunchKernelGGL(lp_rows_kernel, lk_grid, dim3(LP_THREADS), 0, st, (const uint32_t*)flags, (const uint32_t*)scans, m, rows);
    fe r2;
    memcpy(r2.l, HFR.r2, 32);
    hipLaunchKernelGGL(lp_assign_kernel, lk_grid, dim3(LP_THREADS), 0, st, (const fe*)sorted, (const uint32_t*)flags,
                       (const uint32_t*)scans, (const uint32_t*)totals, (const uint32_t*)rows, m, r2, (fe* const*)out_a,
                       (fe* const*)out_s);
    ZK_HIP(ctx, hipGetLastError());
    std::vector<uint32_t> h_status(count + 1);
    ZK_HIP(ctx, hipMemcpyAsync(h_status.data(), status, (size_t)(count + 1) * 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(ctx, hipStreamSynchronize(st));
    ZK_CHECK(ctx, !h_status[0], "lookup_permute: internal error, a column is not sorted after the exact sort");
    for (uint32_t l = 0; l < count; l++)
        if (h_status[1 + l])
            return set_error(ctx, ZKHIP_ERR_CONSTRAINT, "lookup_permute: an input value of lookup %u does not occur in its table",
                             first + l);
    return ZKHIP_OK;
}

}  // namespace zkhip

extern "C" int zkhip_lookup_permute_batch_dev(zkhip_domain* dom, uint32_t count, const uint64_t* const* d_compressed_inputs,
                                              const uint64_t* const* d_compressed_tables, uint32_t usable_rows,
                                              uint64_t* const* d_permuted_inputs, uint64_t* const* d_permuted_tables) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    if (count == 0) return ZKHIP_OK;
    ZK_CHECK(ctx, d_compressed_inputs && d_compressed_tables && d_permuted_inputs && d_permuted_tables, "lookup_permute: null pointer table");
    const size_t n = (size_t)1 << dom->k;
    ZK_CHECK(ctx, usable_rows >= 1 && usable_rows <= n, "lookup_permute: %u usable rows of %zu", usable_rows, n);
    for (uint32_t l = 0; l < count; l++) {
        ZK_CHECK(ctx, d_compressed_inputs[l] && d_compressed_tables[l] && d_permuted_inputs[l] && d_permuted_tables[l],
                 "lookup_permute: null pointer (lookup %u)", l);
        ZK_CHECK(ctx, (const void*)d_permuted_inputs[l] != (const void*)d_permuted_tables[l], "lookup_permute: outputs overlap");
    }
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    TimedRegion tr(ctx, "lookup_permute");
    for (uint32_t first = 0; first < count; first += LP_MAX_BATCH) {
        const uint32_t c = std::min(LP_MAX_BATCH, count - first);
        int rc = lookup_permute_chunk(ctx, first, usable_rows, c, d_compressed_inputs + first, d_compressed_tables + first,
                                      d_permuted_inputs + first, d_permuted_tables + first);
        if (rc) return rc;
    }
    return ZKHIP_OK;
}

extern "C" int zkhip_lookup_permute_dev(zkhip_domain* dom, const uint64_t* d_compressed_input, const uint64_t* d_compressed_table,
                                        uint32_t usable_rows, uint64_t* d_permuted_input, uint64_t* d_permuted_table) {
    if (!dom) return ZKHIP_ERR_INVALID;
    ZK_CHECK(dom->ctx, d_compressed_input && d_compressed_table && d_permuted_input && d_permuted_table, "lookup_permute: null pointer");
    return zkhip_lookup_permute_batch_dev(dom, 1, &d_compressed_input, &d_compressed_table, usable_rows, &d_permuted_input,
                                          &d_permuted_table);
}
