// `arithmetic::best_fft` over G1 points: the transform `ParamsKZG::setup` / `g_to_lagrange` applies to the SRS to obtain
// `g_lagrange` (the reference builds its parameters at zkevm-circuits/benches/benchmark_list/main.rs:184 and
// prover/src/test/create_all_file.rs:118-131).  Same radix-2 decimation-in-time network as the scalar transform: bit
// reversal, then log n stages of (a, b) -> (a + w b, a - w b), where "w b" is a scalar multiplication of a point by a
// power of omega.  This runs once per parameter set, so it is written for clarity: one lane per butterfly, packed field
// arithmetic, plain double-and-add; every stage is one launch over n / 2 butterflies.
#include <cstring>
#include "common.h"
#include "curve.cuh"
#include "ntt.h"

using namespace zkhip;

extern "C" int zkhip_internal_get_plan(zkhip_ctx* ctx, uint32_t log_n, const uint64_t omega[4], zkhip::NttPlan** out);

namespace zkhip {

constexpr int GF_THREADS = 256;

ZK_DEV g1_xyzz xyzz_neg(const g1_xyzz& p) {
    g1_xyzz o = p;
    o.y = Fq::neg(p.y);
    return o;
}

// k * p for a canonical (non-Montgomery) 256-bit scalar k, most significant bit first
ZK_DEV g1_xyzz xyzz_scalar_mul(const g1_xyzz& p, const fe& k) {
    g1_xyzz acc = xyzz_identity();
    bool started = false;
    for (int bit = 255; bit >= 0; bit--) {
        if (started) acc = xyzz_double(acc);
        if ((k.l[bit >> 5] >> (bit & 31)) & 1u) {
            acc = started ? xyzz_add(acc, p) : p;
            started = true;
        }
    }
    return acc;
}

ZK_DEV fe fr_from_montgomery(const fe& a) {
    fe one;
#pragma unroll
    for (int i = 0; i < 8; i++) one.l[i] = 0;
    one.l[0] = 1;
    return Fr::mul(a, one);
}

__global__ void __launch_bounds__(GF_THREADS) g1fft_load_kernel(const g1_affine* in, g1_xyzz* data, uint32_t log_n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log_n)) return;
    const uint32_t r = log_n ? __brev(i) >> (32 - log_n) : 0;
    store_xyzz(data + r, xyzz_from_affine(load_affine(in + i)));
}

// stage s (half-size m = 2^s): butterfly t = (group, j) works on a = group * 2m + j and b = a + m with w = omega^(j n / 2m)
__global__ void __launch_bounds__(GF_THREADS) g1fft_stage_kernel(g1_xyzz* data, uint32_t log_n, uint32_t s, const fe* om_lo,
                                                                 const fe* om_hi, uint32_t h) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (1u << (log_n - 1))) return;
    const uint32_t m = 1u << s, j = t & (m - 1), a = ((t >> s) << (s + 1)) + j, b = a + m;
    const uint32_t e = j << (log_n - 1 - s);
    g1_xyzz pb = load_xyzz(data + b);
    if (e) {
        fe w = load_fe(om_lo + (e & ((1u << h) - 1)));
        if (e >> h) w = Fr::mul(w, load_fe(om_hi + (e >> h)));
        pb = xyzz_scalar_mul(pb, fr_from_montgomery(w));
    }
    const g1_xyzz pa = load_xyzz(data + a);
    store_xyzz(data + a, xyzz_add(pa, pb));
    store_xyzz(data + b, xyzz_add(pa, xyzz_neg(pb)));
}

// out[i] = affine(scale * data[i]); scale canonical, has_scale = 0 leaves the points as they are
__global__ void __launch_bounds__(GF_THREADS) g1fft_finish_kernel(const g1_xyzz* data, uint32_t n, fe scale, int has_scale,
                                                                  g1_affine* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1_xyzz p = load_xyzz(data + i);
    if (has_scale) p = xyzz_scalar_mul(p, scale);
    store_affine(out + i, xyzz_to_affine(p));
}

}  // namespace zkhip

extern "C" int zkhip_g1_fft(zkhip_ctx* ctx, const uint64_t* affine_xy, uint32_t log_n, const uint64_t omega[4],
                            const uint64_t* scale, uint64_t* out_xy) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, affine_xy && omega && out_xy, "g1_fft: null pointer");
    ZK_CHECK(ctx, log_n >= 1 && log_n <= 26, "g1_fft: log_n = %u outside [1, 26]", log_n);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << log_n;
    NttPlan* plan = nullptr;
    int rc = zkhip_internal_get_plan(ctx, log_n, omega, &plan);   // also checks that omega has order 2^log_n
    if (rc) return rc;
    rc = ensure_scratch(ctx, n * (sizeof(g1_xyzz) + sizeof(g1_affine)));
    if (rc) return rc;
    g1_xyzz* data = (g1_xyzz*)ctx->scratch;
    g1_affine* pts = (g1_affine*)(data + n);
    hipStream_t st = ctx->stream;
    ZK_HIP(ctx, hipMemcpyAsync(pts, affine_xy, n * sizeof(g1_affine), hipMemcpyHostToDevice, st));
    const dim3 grid_n((unsigned)((n + GF_THREADS - 1) / GF_THREADS)), grid_h((unsigned)((n / 2 + GF_THREADS - 1) / GF_THREADS));
    hipLaunchKernelGGL(g1fft_load_kernel, grid_n, dim3(GF_THREADS), 0, st, (const g1_affine*)pts, data, log_n);
    for (uint32_t s = 0; s < log_n; s++)
        hipLaunchKernelGGL(g1fft_stage_kernel, grid_h, dim3(GF_THREADS), 0, st, data, log_n, s, (const fe*)plan->d_lo,
                           (const fe*)plan->d_hi, plan->h);
    fe sc;
    memset(&sc, 0, sizeof sc);
    if (scale) {   // Montgomery -> canonical on the host: multiply by the integer 1
        uint64_t one[4] = {1, 0, 0, 0}, canon[4];
        h_mul(HFR, canon, scale, one);
        memcpy(sc.l, canon, 32);
    }
    hipLaunchKernelGGL(g1fft_finish_kernel, grid_n, dim3(GF_THREADS), 0, st, (const g1_xyzz*)data, (uint32_t)n, sc, scale ? 1 : 0, pts);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipMemcpyAsync(out_xy, pts, n * sizeof(g1_affine), hipMemcpyDeviceToHost, st));
    ZK_HIP(ctx, hipStreamSynchronize(st));
    return ZKHIP_OK;
}
