// Montgomery-multiplier variant probe for gfx950 (development tool, not part of the library).
// Build: hipcc --offload-arch=gfx950 -O3 -o mont_probe mont_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__constant__ uint32_t P32[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
constexpr uint32_t INV32 = 0xefffffffu;

struct fe { uint32_t l[8]; };

// ---- V1: CIOS, 32-bit limbs, straightforward u64 arithmetic
__device__ __forceinline__ fe mul_v1(const fe& a, const fe& b) {
    const uint32_t p[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    uint32_t t[10] = {0};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t s = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint32_t)s; c = s >> 32;
        }
        uint64_t s = (uint64_t)t[8] + c;
        t[8] = (uint32_t)s; t[9] = (uint32_t)(s >> 32);
        uint32_t m = t[0] * INV32;
        c = ((uint64_t)m * p[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            uint64_t s2 = (uint64_t)m * p[j] + t[j] + c;
            t[j - 1] = (uint32_t)s2; c = s2 >> 32;
        }
        s = (uint64_t)t[8] + c;
        t[7] = (uint32_t)s; t[8] = t[9] + (uint32_t)(s >> 32);
    }
    // conditional subtract
    uint32_t r[8]; uint64_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { uint64_t d = (uint64_t)t[j] - p[j] - br; r[j] = (uint32_t)d; br = (d >> 32) & 1; }
    bool ge = t[8] || !br;
    fe o;
#pragma unroll
    for (int j = 0; j < 8; j++) o.l[j] = ge ? r[j] : t[j];
    return o;
}

// ---- V2: separated product (operand scanning with row products) then Montgomery reduction, lazy
// wide columns: accumulate 64-bit columns as two 32-bit halves summed into 64-bit accumulators.
// Column-wise with 64-bit accumulators split lo/hi: each product p=a_i*b_j contributes lo(p) to col k and hi(p) to col k+1.
// Implemented as: col[k] (u64) += (u32)lo ; col[k+1] (u64) += hi, via mad_u64_u32(a,b,0) then two 64-bit adds -> too many adds.
// Instead V2 = 9x29-bit limbs, no carries in the accumulation.
struct fe29 { uint32_t l[9]; };
__device__ __forceinline__ fe29 to29(const fe& a) {
    fe29 o;
    const uint32_t M = (1u << 29) - 1;
    o.l[0] = a.l[0] & M;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        int bit = 29 * i; int w = bit / 32, s = bit % 32;
        uint32_t lo = a.l[w] >> s;
        uint32_t hi = (w + 1 < 8 && s) ? (a.l[w + 1] << (32 - s)) : 0;
        o.l[i] = (lo | hi) & M;
    }
    return o;
}
__device__ __forceinline__ fe from29(const fe29& a) {
    fe o;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        // bits [32w, 32w+32)
        int i = (32 * w) / 29, s = (32 * w) % 29;   // starts in limb i at offset s
        uint32_t v = a.l[i] >> s;
        int got = 29 - s;
        if (i + 1 < 9) v |= a.l[i + 1] << got;
        got += 29;
        if (got < 32 && i + 2 < 9) v |= a.l[i + 2] << got;
        o.l[w] = v;
    }
    return o;
}
// Montgomery with R = 2^261 (9*29). Inputs/outputs in 29-bit limbs, values < p.
__device__ __forceinline__ fe29 mul_v2(const fe29& a, const fe29& b, const uint32_t* p29, uint32_t inv29) {
    const uint32_t M = (1u << 29) - 1;
    uint64_t col[18];
#pragma unroll
    for (int k = 0; k < 18; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) col[i + j] += (uint64_t)a.l[i] * b.l[j];
    // reduction: for each k in 0..8: m = (col[k] * inv) mod 2^29; col[k..k+8] += m*p; carry col[k]>>29 into col[k+1]
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint32_t m = ((uint32_t)col[k] * inv29) & M;
#pragma unroll
        for (int j = 0; j < 9; j++) col[k + j] += (uint64_t)m * p29[j];
        col[k + 1] += col[k] >> 29;
    }
    fe29 o;
    uint64_t c = 0;
#pragma unroll
    for (int k = 9; k < 18; k++) { uint64_t s = col[k] + c; o.l[k - 9] = (uint32_t)s & M; c = s >> 29; }
    // result < 2p; conditional subtract
    uint32_t r[9]; int32_t br = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) { int32_t d = (int32_t)o.l[j] - (int32_t)p29[j] + br; r[j] = (uint32_t)d & M; br = d >> 29; }
    bool ge = (br == 0);
#pragma unroll
    for (int j = 0; j < 9; j++) o.l[j] = ge ? r[j] : o.l[j];
    return o;
}

constexpr int ITERS = 512;

__global__ void __launch_bounds__(256) k_v1(fe* io) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe x = io[g], y = io[g ^ 1];
    for (int i = 0; i < ITERS; i++) { x = mul_v1(x, y); y = mul_v1(y, x); }
    io[g] = x;
}
__global__ void __launch_bounds__(256) k_v2(fe* io, const uint32_t* p29g, uint32_t inv29) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t p29[9];
#pragma unroll
    for (int j = 0; j < 9; j++) p29[j] = p29g[j];
    fe29 x = to29(io[g]), y = to29(io[g ^ 1]);
    for (int i = 0; i < ITERS; i++) { x = mul_v2(x, y, p29, inv29); y = mul_v2(y, x, p29, inv29); }
    io[g] = from29(x);
}

int main() {
    int blocks = 256 * 8;
    size_t n = (size_t)blocks * 256;
    fe* d; CK(hipMalloc(&d, n * sizeof(fe)));
    fe* h = (fe*)malloc(n * sizeof(fe));
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) for (int j = 0; j < 8; j++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].l[j] = (uint32_t)s; if (j == 7) h[i].l[j] &= 0x1fffffffu; }
    // p in 29-bit limbs
    uint32_t p32[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    uint32_t p29[9];
    for (int i = 0; i < 9; i++) { int bit = 29 * i, w = bit / 32, sh = bit % 32; uint64_t v = p32[w] >> sh; if (w + 1 < 8) v |= (uint64_t)p32[w + 1] << (32 - sh); p29[i] = (uint32_t)(v & ((1u << 29) - 1)); }
    uint32_t inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - p29[0] * inv; uint32_t inv29 = (0u - inv) & ((1u << 29) - 1);
    uint32_t* dp; CK(hipMalloc(&dp, 36)); CK(hipMemcpy(dp, p29, 36, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int v = 1; v <= 2; v++) {
        CK(hipMemcpy(d, h, n * sizeof(fe), hipMemcpyHostToDevice));
        if (v == 1) k_v1<<<blocks, 256>>>(d); else k_v2<<<blocks, 256>>>(d, dp, inv29);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 3; r++) { if (v == 1) k_v1<<<blocks, 256>>>(d); else k_v2<<<blocks, 256>>>(d, dp, inv29); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double muls = 3.0 * n * ITERS * 2;
        printf("V%d: %.2f G Fr-mul/s (%.3f ms/launch)\n", v, muls / (ms * 1e-3) / 1e9, ms / 3);
    }
    return 0;
}
