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
// Sorting and scanning use rocPRIM (merge sort with a 256-bit comparison, exclusive scan); they are library
// primitives next to the hot path, not part of it.
#include <cstring>
#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "common.h"
#include "domain.h"

using namespace zkhip;

namespace zkhip {

// canonical (non-Montgomery) value as eight words, compared from the most significant word down
struct CanonLess {
    __device__ __host__ bool operator()(const fe& a, const fe& b) const {
        for (int i = 7; i >= 0; i--) {
            if (a.l[i] != b.l[i]) return a.l[i] < b.l[i];
        }
        return false;
    }
};
ZK_DEV bool fe_same(const fe& a, const fe& b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= a.l[i] ^ b.l[i];
    return d == 0;
}
// index of the first element >= v in the sorted array a[0..n)
ZK_DEV uint32_t lower_bound(const fe* a, uint32_t n, const fe& v) {
    uint32_t lo = 0, hi = n;
    CanonLess less;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (less(load_fe(a + mid), v)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

constexpr int LP_THREADS = 256;

// Montgomery -> canonical: multiply by the integer 1
ZK_DEV fe lp_canonical(const fe& a) {
    fe one;
#pragma unroll
    for (int i = 0; i < 8; i++) one.l[i] = 0;
    one.l[0] = 1;
    return Fr::mul(a, one);
}

__global__ void __launch_bounds__(LP_THREADS) lp_from_mont_kernel(const fe* a, const fe* s, fe* ac, fe* sc, uint32_t m) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    store_fe(ac + i, lp_canonical(load_fe(a + i)));
    store_fe(sc + i, lp_canonical(load_fe(s + i)));
}

// rep[i] = 1 where A'[i] repeats A'[i-1]; left[j] = 1 where T[j] is NOT the instance taken out for an input value;
// bad is raised when a distinct input value does not occur in the table
__global__ void __launch_bounds__(LP_THREADS) lp_flags_kernel(const fe* as, const fe* ts, uint32_t m, uint32_t* rep,
                                                              uint32_t* left, uint32_t* bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const fe a = load_fe(as + i);
    const bool first_a = i == 0 || !fe_same(a, load_fe(as + i - 1));
    rep[i] = first_a ? 0u : 1u;
    if (first_a) {
        const uint32_t p = lower_bound(ts, m, a);
        if (p >= m || !fe_same(load_fe(ts + p), a)) atomicOr(bad, 1u);
    }
    const fe t = load_fe(ts + i);
    const bool first_t = i == 0 || !fe_same(t, load_fe(ts + i - 1));
    bool taken = false;
    if (first_t) {
        const uint32_t p = lower_bound(as, m, t);
        taken = p < m && fe_same(load_fe(as + p), t);
    }
    left[i] = taken ? 0u : 1u;
}

// rows[k] = k-th repeated row (ascending)
__global__ void __launch_bounds__(LP_THREADS) lp_rows_kernel(const uint32_t* rep, const uint32_t* rep_idx, uint32_t m, uint32_t* rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m && rep[i]) rows[rep_idx[i]] = i;
}

// A' and S' in Montgomery form.  n_rep = number of repeated rows = number of leftover table values.
__global__ void __launch_bounds__(LP_THREADS) lp_assign_kernel(const fe* as, const fe* ts, const uint32_t* rep, const uint32_t* left,
                                                               const uint32_t* left_idx, const uint32_t* rows,
                                                               const uint32_t* rep_idx, uint32_t m, fe r2, fe* out_a, fe* out_s) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t n_rep = rep_idx[m - 1] + rep[m - 1];
    const fe a = Fr::mul(load_fe(as + i), r2);   // canonical -> Montgomery
    store_fe(out_a + i, a);
    if (!rep[i]) store_fe(out_s + i, a);
    if (left[i]) {
        const uint32_t k = left_idx[i];
        if (k < n_rep) store_fe(out_s + rows[n_rep - 1 - k], Fr::mul(load_fe(ts + i), r2));
    }
}

}  // namespace zkhip

extern "C" int zkhip_lookup_permute_dev(zkhip_domain* dom, const uint64_t* d_compressed_input, const uint64_t* d_compressed_table,
                                        uint32_t usable_rows, uint64_t* d_permuted_input, uint64_t* d_permuted_table) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_compressed_input && d_compressed_table && d_permuted_input && d_permuted_table, "lookup_permute: null pointer");
    const size_t n = (size_t)1 << dom->k;
    ZK_CHECK(ctx, usable_rows >= 1 && usable_rows <= n, "lookup_permute: %u usable rows of %zu", usable_rows, n);
    ZK_CHECK(ctx, (const void*)d_permuted_input != (const void*)d_permuted_table, "lookup_permute: outputs overlap");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t m = usable_rows;
    hipStream_t st = ctx->stream;

    // temporary storage sizes of the library primitives
    size_t sort_bytes = 0, scan_bytes = 0;
    ZK_HIP(ctx, rocprim::merge_sort(nullptr, sort_bytes, (fe*)nullptr, (fe*)nullptr, (size_t)m, CanonLess(), st));
    ZK_HIP(ctx, rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)m,
                                        rocprim::plus<uint32_t>(), st));
    const size_t tmp_bytes = ((sort_bytes > scan_bytes ? sort_bytes : scan_bytes) + 255) & ~(size_t)255;
    // scratch: [Ac m][Sc m][As m][Ts m] elements, five word arrays of m, one status word, primitive scratch
    const size_t words = 5 * (size_t)m + 64;
    int rc = ensure_scratch(ctx, 4 * (size_t)m * 32 + words * 4 + tmp_bytes + 512);
    if (rc) return rc;
    fe* ac = (fe*)ctx->scratch;
    fe *sc = ac + m, *as = sc + m, *ts = as + m;
    uint32_t* rep = (uint32_t*)(ts + m);
    uint32_t *left = rep + m, *rep_idx = left + m, *left_idx = rep_idx + m, *rows = left_idx + m, *bad = rows + m;
    void* tmp = (void*)(((uintptr_t)(bad + 64) + 255) & ~(uintptr_t)255);

    TimedRegion tr(ctx, "lookup_permute");
    const dim3 grid((m + LP_THREADS - 1) / LP_THREADS);
    ZK_HIP(ctx, hipMemsetAsync(bad, 0, 4, st));
    hipLaunchKernelGGL(lp_from_mont_kernel, grid, dim3(LP_THREADS), 0, st, (const fe*)d_compressed_input,
                       (const fe*)d_compressed_table, ac, sc, m);
    ZK_HIP(ctx, hipGetLastError());
    size_t sb = tmp_bytes;
    ZK_HIP(ctx, rocprim::merge_sort(tmp, sb, ac, as, (size_t)m, CanonLess(), st));
    sb = tmp_bytes;
    ZK_HIP(ctx, rocprim::merge_sort(tmp, sb, sc, ts, (size_t)m, CanonLess(), st));
    hipLaunchKernelGGL(lp_flags_kernel, grid, dim3(LP_THREADS), 0, st, (const fe*)as, (const fe*)ts, m, rep, left, bad);
    ZK_HIP(ctx, hipGetLastError());
    sb = tmp_bytes;
    ZK_HIP(ctx, rocprim::exclusive_scan(tmp, sb, rep, rep_idx, 0u, (size_t)m, rocprim::plus<uint32_t>(), st));
    sb = tmp_bytes;
    ZK_HIP(ctx, rocprim::exclusive_scan(tmp, sb, left, left_idx, 0u, (size_t)m, rocprim::plus<uint32_t>(), st));
    hipLaunchKernelGGL(lp_rows_kernel, grid, dim3(LP_THREADS), 0, st, (const uint32_t*)rep, (const uint32_t*)rep_idx, m, rows);
    fe r2;
    memcpy(r2.l, HFR.r2, 32);
    hipLaunchKernelGGL(lp_assign_kernel, grid, dim3(LP_THREADS), 0, st, (const fe*)as, (const fe*)ts, (const uint32_t*)rep,
                       (const uint32_t*)left, (const uint32_t*)left_idx, (const uint32_t*)rows, (const uint32_t*)rep_idx, m, r2,
                       (fe*)d_permuted_input, (fe*)d_permuted_table);
    ZK_HIP(ctx, hipGetLastError());
    uint32_t h_bad = 0;
    ZK_HIP(ctx, hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(ctx, hipStreamSynchronize(st));
    if (h_bad) return set_error(ctx, ZKHIP_ERR_CONSTRAINT, "lookup_permute: an input value does not occur in the table");
    return ZKHIP_OK;
}
