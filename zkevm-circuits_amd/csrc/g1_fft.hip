// `arithmetic::best_fft` over G1 points: the transform `ParamsKZG::setup` / `g_to_lagrange` applies to the SRS to obtain
// `g_lagrange` (the reference builds its parameters at zkevm-circuits/benches/benchmark_list/main.rs:184 and
// prover/src/test/create_all_file.rs:118-131).  Same radix-2 decimation-in-time network as the scalar transform: bit
// reversal, then log n stages of (a, b) -> (a + w b, a - w b), where "w b" is a scalar multiplication of a point by a
// power of omega; every stage is one launch over n / 2 butterflies, one lane per butterfly.  Runs once per parameter set.
//
// Points travel between the stages as XYZZ with coordinates in the 2^261 Montgomery domain of the lazy signed-digit
// arithmetic (csrc/wide.hip.h).  The scalar multiplication walks the non-adjacent form of the twiddle, most significant
// digit first (digit i = bit i + 1 of 3 k minus bit i + 1 of k; a third of the digits are nonzero).  A stage has n / 2m
// butterflies per twiddle: while that is at least 64, the 64 lanes of a wave take butterflies of ONE twiddle, so the digits
// are wave-uniform and a zero digit skips the addition for the whole wave (in a wave of mixed twiddles some lane always
// needs it); the last six stages have fewer butterflies per twiddle than lanes and pay the addition at every digit.
#include <cstring>
#include "common.h"
#include "curve.hip.h"
#include "curve_wide.hip.h"
#include "ntt.h"

using namespace zkhip;

extern "C" int zkhip_internal_get_plan(zkhip_ctx* ctx, uint32_t log_n, const uint64_t omega[4], zkhip::NttPlan** out);

namespace zkhip {

constexpr int GF_THREADS = 256;

// k * p for a canonical (non-Montgomery) 256-bit scalar k, by its non-adjacent form, most significant digit first.
// When k is the same in every lane of the wave (the caller made it so with readfirstlane) the digit branches are scalar.
__device__ __forceinline__ g1_xyzz_w xyzz_w_scalar_mul(const g1_xyzz_w& p, const fe& k) {
    uint32_t h[9];   // 3 k
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += 3ull * k.l[i];
        h[i] = (uint32_t)c;
        c >>= 32;
    }
    h[8] = (uint32_t)c;
    g1_xyzz_w np = p;
    np.y = FqW::neg(p.y);
    g1_xyzz_w acc = xyzz_w_identity();
    // k < r < 2^254, so 3 k < 2^256: the top digit is at most digit 254
#pragma unroll 1
    for (int i = 254; i >= 0; i--) {
        if (!acc.inf) acc = xyzz_w_double(acc);   // the identity stays (also before a lane's first digit)
        const int b = i + 1;   // bit i + 1 of 3 k and of k
        const int d = (int)((h[b >> 5] >> (b & 31)) & 1u) - (int)((k.l[b >> 5] >> (b & 31)) & 1u);
        if (d != 0) {           // a scalar branch when k is wave-uniform
            g1_xyzz_w q = p;
            if (d < 0) q = np;
            acc = xyzz_w_add(acc, q);   // the identity passes q through
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

// affine (2^256 domain) -> XYZZ in the 2^261 domain, bit-reversed position
__global__ void __launch_bounds__(GF_THREADS) g1fft_load_kernel(const g1_affine* in, g1_xyzz* data, uint32_t log_n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log_n)) return;
    const uint32_t r = log_n ? __brev(i) >> (32 - log_n) : 0;
    const g1_affine p = load_affine(in + i);
    g1_xyzz_w w;
    w.inf = Fq::is_zero(p.x) && Fq::is_zero(p.y);
    w.x = FqW::from_fe_shl5(p.x);
    w.y = FqW::from_fe_shl5(p.y);
    w.zz = FqW::one();
    w.zzz = FqW::one();
    xyzz_w_store(data + r, w);
}

// stage s (half-size m = 2^s): butterfly (group, j) works on a = group * 2m + j and b = a + m with w = omega^(j n / 2m).
// by_twiddle: consecutive lanes take consecutive GROUPS of one j (needs at least 64 groups), else consecutive j.
__global__ void __launch_bounds__(GF_THREADS) g1fft_stage_kernel(g1_xyzz* data, uint32_t log_n, uint32_t s, const fe* om_lo,
                                                                 const fe* om_hi, uint32_t h, int by_twiddle) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (1u << (log_n - 1))) return;
    const uint32_t log_groups = log_n - 1 - s;
    uint32_t j, group;
    if (by_twiddle) {
        group = t & ((1u << log_groups) - 1);
        j = t >> log_groups;
    } else {
        j = t & ((1u << s) - 1);
        group = t >> s;
    }
    const uint32_t a = (group << (s + 1)) + j, b = a + (1u << s);
    const uint32_t e = j << log_groups;
    g1_xyzz_w pb = xyzz_w_load(data + b);
    if (e) {
        fe w = load_fe(om_lo + (e & ((1u << h) - 1)));
        if (e >> h) w = Fr::mul(w, load_fe(om_hi + (e >> h)));
        fe k = fr_from_montgomery(w);
        if (by_twiddle) {
#pragma unroll
            for (int i = 0; i < 8; i++) k.l[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)k.l[i]);
        }
        pb = xyzz_w_scalar_mul(pb, k);
    }
    const g1_xyzz_w pa = xyzz_w_load(data + a);
    g1_xyzz_w nb = pb;
    nb.y = FqW::neg(pb.y);
    xyzz_w_store(data + a, xyzz_w_add(pa, pb));
    xyzz_w_store(data + b, xyzz_w_add(pa, nb));
}

// out[i] = affine(scale * data[i]); scale canonical, has_scale = 0 leaves the points as they are
__global__ void __launch_bounds__(GF_THREADS) g1fft_finish_kernel(const g1_xyzz* data, uint32_t n, fe scale, int has_scale,
                                                                  g1_affine* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1_xyzz_w p = xyzz_w_load(data + i);
    if (has_scale) p = xyzz_w_scalar_mul(p, scale);   // one scalar for every lane
    // back to the 2^256 domain, then the packed conversion to affine
    g1_xyzz q = xyzz_identity();
    if (!p.inf) {
        q.x = FqW::to_fe(FqW::mul(FqW::norm(p.x), FqW::r256()));
        q.y = FqW::to_fe(FqW::mul(FqW::norm(p.y), FqW::r256()));
        q.zz = FqW::to_fe(FqW::mul(FqW::norm(p.zz), FqW::r256()));
        q.zzz = FqW::to_fe(FqW::mul(FqW::norm(p.zzz), FqW::r256()));
    }
    store_affine(out + i, xyzz_to_affine(q));
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
                           (const fe*)plan->d_hi, plan->h, (log_n - 1 - s) >= 6 ? 1 : 0);
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
