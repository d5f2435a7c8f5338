// Development tool: how long a wave takes for a chain of dependent point additions, alone on its SIMD and with one or
// three neighbours.  The MSM's merge and reduce phases are such chains; whether two of them on one SIMD run side by side
// or take turns decides how many lanes a bucket may be given.
// Every lane adds the same affine point K times to an accumulator that starts from another point (no doubling, no
// identity: the common path of xyzz_w_add), in the one-lane and the two-lane form of curve_wide.hip.h.
// Build: hipcc --offload-arch=gfx950 -O3 -I zkevm-circuits_amd/csrc -o tools/chain_probe tools/chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "curve_wide.hip.h"

using namespace zkhip;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// start[i]: XYZZ points (any valid ones); out keeps the result alive
template <bool PAIRS>
__global__ void __launch_bounds__(256) chain_kernel(const g1_xyzz* start, uint32_t n_start, uint32_t K, g1_xyzz* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const uint32_t logical = PAIRS ? t >> 1 : t;
    g1_xyzz_w p = xyzz_w_load(start + logical % n_start);
    const g1_xyzz_w q = xyzz_w_load(start + (logical + 7) % n_start);
    const bool role = (lane & 1u) != 0;
#pragma unroll 1
    for (uint32_t k = 0; k < K; k++) p = PAIRS ? xyzz_w_add_pair(p, q, role) : xyzz_w_add(p, q);
    if (!PAIRS || !role) xyzz_w_store(out + logical, p);
}

// P_i = (i + 1) G in XYZZ form, made with the one-lane addition itself
__global__ void make_points_kernel(g1_xyzz* pts, uint32_t n) {
    if (blockIdx.x || threadIdx.x) return;
    g1_affine g;
    for (int i = 0; i < 8; i++) g.x.l[i] = FqParams::ONE[i], g.y.l[i] = 0;
    // G = (1, 2) in the 2^256 Montgomery domain, moved to the 2^261 domain of the wide form
    const fe two = Fq::add(g.x, g.x);
    fe five;
    for (int i = 0; i < 8; i++) five.l[i] = FqParams::TWO5[i];
    g1_xyzz_w acc;
    acc.x = FqW::from_fe(Fq::mul(g.x, five));
    acc.y = FqW::from_fe(Fq::mul(two, five));
    acc.zz = FqW::one();
    acc.zzz = FqW::one();
    acc.inf = false;
    const g1_xyzz_w gen = acc;
    acc = xyzz_w_double(acc);
    for (uint32_t i = 0; i < n; i++) {
        xyzz_w_store(pts + i, acc);
        acc = xyzz_w_add(acc, gen);
    }
}

int main(int argc, char** argv) {
    const uint32_t K = argc > 1 ? (uint32_t)atoi(argv[1]) : 32, n_start = 64;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const uint32_t simds = (uint32_t)prop.multiProcessorCount * 4;
    g1_xyzz *pts, *out;
    CK(hipMalloc(&pts, n_start * sizeof(g1_xyzz)));
    CK(hipMalloc(&out, (size_t)simds * 8 * 64 * sizeof(g1_xyzz)));
    hipLaunchKernelGGL(make_points_kernel, dim3(1), dim3(64), 0, 0, pts, n_start);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"simds\": %u, \"additions_per_chain\": %u, \"rows\": [\n", prop.gcnArchName, simds, K);
    bool first = true;
    for (int pairs = 0; pairs < 2; pairs++)
        for (uint32_t per_simd : {1u, 2u, 4u}) {
            // workgroups of four waves: the dispatcher gives each SIMD of a compute unit one of them
            const uint32_t waves = simds * per_simd, blocks = waves / 4;
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0, 0));
                if (pairs) hipLaunchKernelGGL(chain_kernel<true>, dim3(blocks), dim3(256), 0, 0, pts, n_start, K, out);
                else hipLaunchKernelGGL(chain_kernel<false>, dim3(blocks), dim3(256), 0, 0, pts, n_start, K, out);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%s  {\"form\": \"%s\", \"waves_per_simd\": %u, \"kernel_us\": %.1f, \"us_per_addition\": %.2f}", first ? "" : ",\n",
                   pairs ? "two lanes per point" : "one lane per point", per_simd, best * 1e3, best * 1e3 / K);
            first = false;
        }
    printf("\n]}\n");
    return 0;
}
