// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the MSM accumulate kernel: every
// kernel here moves a known number of bytes, so the counter value per launch divided by that number is the factor to
// apply to the counter for that shape.
//   stream32   32 contiguous bytes per lane (two 16-byte loads): how scalars and field elements are read
//   stream4    4 contiguous bytes per lane: how the sorted entry lists are read
//   gather64   64 bytes per lane (four 16-byte loads) at a random 64-byte-aligned offset of a 512 MiB table: how window
//              table points are gathered
//   store32    32 contiguous bytes per lane written
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip
// Run each counter in its own pass: rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) stream32(const u32x4* in, uint32_t* out, size_t lanes) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= lanes) return;
    const u32x4 a = in[2 * g], b = in[2 * g + 1];
    const uint32_t s = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
    if (s == 0x12345678u) out[0] = s;
}
__global__ void __launch_bounds__(256) stream4(const uint32_t* in, uint32_t* out, size_t lanes) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= lanes) return;
    const uint32_t s = in[g];
    if (s == 0x12345678u) out[0] = s;
}
__global__ void __launch_bounds__(256) gather64(const u32x4* table, uint32_t* out, size_t lanes, uint32_t mask) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= lanes) return;
    uint32_t h = (uint32_t)g * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const u32x4* p = table + 4 * (size_t)(h & mask);
    const u32x4 a = p[0], b = p[1], c = p[2], d = p[3];
    const uint32_t s = a.x ^ b.y ^ c.z ^ d.w;
    if (s == 0x12345678u) out[0] = s;
}
__global__ void __launch_bounds__(256) store32(u32x4* outv, size_t lanes) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= lanes) return;
    u32x4 v;
    v.x = (uint32_t)g; v.y = 1; v.z = 2; v.w = 3;
    outv[2 * g] = v;
    outv[2 * g + 1] = v;
}

int main() {
    const size_t bytes = (size_t)1 << 30;       // 1 GiB buffer: four times the Infinity Cache
    void *buf, *out;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    const int reps = 4;
    for (int r = 0; r < reps; r++) {
        size_t lanes = bytes / 32;
        stream32<<<(unsigned)(lanes / 256), 256>>>((const u32x4*)buf, (uint32_t*)out, lanes);
        lanes = bytes / 4 / 4;                  // 256 MiB of 4-byte reads
        stream4<<<(unsigned)(lanes / 256), 256>>>((const uint32_t*)buf, (uint32_t*)out, lanes);
        lanes = (size_t)1 << 23;                // 2^23 gathers of 64 B = 512 MiB out of a 512 MiB table
        gather64<<<(unsigned)(lanes / 256), 256>>>((const u32x4*)buf, (uint32_t*)out, lanes, (1u << 23) - 1);
        lanes = bytes / 32;
        store32<<<(unsigned)(lanes / 256), 256>>>((u32x4*)buf, lanes);
    }
    CK(hipDeviceSynchronize());
    printf("{\"stream32_bytes\": %zu, \"stream4_bytes\": %zu, \"gather64_bytes\": %zu, \"store32_bytes\": %zu, \"reps\": %d}\n", bytes,
           bytes / 4, ((size_t)1 << 23) * 64, bytes, reps);
    CK(hipFree(buf));
    CK(hipFree(out));
    return 0;
}
