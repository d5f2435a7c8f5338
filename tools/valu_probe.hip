// VALU integer-multiply throughput probe for gfx950: measures issue rate of the instructions a
// Montgomery multiplier can be built from.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;   // independent chains per lane

template <int KIND>
__global__ void __launch_bounds__(256) probe(uint64_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[UNROLL];
    double dacc[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) { acc[u] = a + u; dacc[u] = (double)(a + u); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (KIND == 0) {  // v_mad_u64_u32
                acc[u] = (uint64_t)(uint32_t)acc[u] * (uint64_t)b + acc[u];
            } else if (KIND == 1) {  // v_mul_lo_u32
                acc[u] = (uint32_t)acc[u] * b;
            } else if (KIND == 2) {  // v_mul_hi_u32
                acc[u] = __umulhi((uint32_t)acc[u], b);
            } else if (KIND == 3) {  // v_mad_u32_u24
                acc[u] = __umul24((uint32_t)acc[u], b) + a;
            } else if (KIND == 4) {  // 64-bit add (v_add_co + v_addc)
                acc[u] = acc[u] + (((uint64_t)b << 32) | a);
            } else if (KIND == 5) {  // v_fma_f64
                dacc[u] = __builtin_fma(dacc[u], 1.0000001, 0.5);
            } else if (KIND == 6) {  // 32-bit add
                acc[u] = (uint32_t)acc[u] + b;
            } else if (KIND == 8) {  // v_mad_i64_i32
                acc[u] = (uint64_t)((int64_t)(int32_t)acc[u] * (int64_t)(int32_t)b + (int64_t)acc[u]);
            } else if (KIND == 9) {  // v_ashrrev_i64
                acc[u] = (uint64_t)((int64_t)acc[u] >> (b & 31)) ^ a;
            } else if (KIND == 10) {  // v_lshl_add_u64
                acc[u] = (acc[u] << 1) + (((uint64_t)b << 32) | a);
            } else if (KIND == 7) {  // v_mad_u64_u32 with carry-out consumed (mad + addc)
                uint64_t r = (uint64_t)(uint32_t)acc[u] * (uint64_t)b;
                uint64_t s = r + acc[u];
                acc[u] = s + (s < r ? 1u : 0u);
            }
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) r += acc[u] + (uint64_t)dacc[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
int run(const char* name, uint64_t* d_out, int blocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<KIND><<<blocks, 256>>>(d_out, 12345);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) probe<KIND><<<blocks, 256>>>(d_out, 12345 + r);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double ops = 5.0 * blocks * 256.0 * ITERS * UNROLL;
    printf("%-28s %8.2f Gop/s/lane-total (%.3f ms/launch)\n", name, ops / (ms * 1e-3) / 1e9, ms / 5);
    return 0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    int blocks = p.multiProcessorCount * 8;
    uint64_t* d; CK(hipMalloc(&d, (size_t)blocks * 256 * 8));
    run<0>("v_mad_u64_u32", d, blocks);
    run<1>("v_mul_lo_u32", d, blocks);
    run<2>("v_mul_hi_u32", d, blocks);
    run<3>("v_mad_u32_u24", d, blocks);
    run<4>("add_u64 (add_co+addc)", d, blocks);
    run<5>("v_fma_f64", d, blocks);
    run<6>("v_add_u32", d, blocks);
    run<7>("mad_u64 + carry fold", d, blocks);
    run<8>("v_mad_i64_i32", d, blocks);
    run<9>("v_ashrrev_i64 (+xor)", d, blocks);
    run<10>("v_lshl_add_u64", d, blocks);
    CK(hipFree(d));
    return 0;
}
