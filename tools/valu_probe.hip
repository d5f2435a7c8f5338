// VALU issue-rate probe for gfx950: measures, per instruction kind, how many wave64 instructions a SIMD
// retires per cycle at 1, 2, 4 and 8 waves per SIMD.  Every instruction is written as inline asm on
// UNROLL independent registers, so the compiler can neither fold a chain nor change the instruction.
// The shader clock during each run is read from s_memtime / s_memrealtime (100 MHz), so rates are
// reported both as T lane-operations/s and as cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip ; run: ./valu_probe [out.json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 16;   // independent chains per lane

enum Kind {
    K_MAD_U64_U32, K_MAD_I64_I32, K_MUL_LO_U32, K_MUL_HI_U32, K_MAD_U32_U24, K_FMA_F64, K_FMA_F32, K_PK_FMA_F32,
    K_ADD_U32, K_ADD3_U32, K_ADD_CO, K_ADDC_CO, K_AND_B32, K_AND_OR_B32, K_LSHLREV_B32, K_BFE_U32, K_ALIGNBIT,
    K_CNDMASK, K_MOV_B32, K_LSHRREV_B64, K_ASHRREV_I64, K_LSHL_ADD_U64, K_MIX_MAD_ADD, K_MIX_MAD_2ADD, K_COUNT
};

static const char* kind_name[K_COUNT] = {
    "v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_fma_f64", "v_fma_f32",
    "v_pk_fma_f32", "v_add_u32", "v_add3_u32", "v_add_co_u32", "v_addc_co_u32", "v_and_b32", "v_and_or_b32",
    "v_lshlrev_b32", "v_bfe_u32", "v_alignbit_b32", "v_cndmask_b32", "v_mov_b32", "v_lshrrev_b64",
    "v_ashrrev_i64", "v_lshl_add_u64", "mix: 1 mad_u64_u32 + 1 add_u32", "mix: 1 mad_u64_u32 + 2 add_u32"
};
// wave-instructions issued per chain step
static const int kind_insts[K_COUNT] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 3};

template <int KIND>
__global__ void __launch_bounds__(256) probe(uint64_t* out, uint64_t* clocks, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = (seed * 3 + blockIdx.x) | 1;
    uint64_t acc[UNROLL];
    uint32_t lo[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) { acc[u] = ((uint64_t)(a + u) << 20) | (b + u); lo[u] = a * (u + 3); }
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (KIND == K_MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u]) : "v"(lo[u]), "v"(b) : "vcc");
            else if (KIND == K_MAD_I64_I32) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[u]) : "v"(lo[u]), "v"(b) : "vcc");
            else if (KIND == K_MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[u]) : "v"(b));
            else if (KIND == K_MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo[u]) : "v"(b));
            else if (KIND == K_MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo[u]) : "v"(b), "v"(a));
            else if (KIND == K_FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(acc[u]) : "v"(acc[(u + 1) % UNROLL]));
            else if (KIND == K_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(lo[u]) : "v"(b), "v"(a));
            else if (KIND == K_PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(acc[u]) : "v"(acc[(u + 1) % UNROLL]));
            else if (KIND == K_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[u]) : "v"(b));
            else if (KIND == K_ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo[u]) : "v"(b), "v"(a));
            else if (KIND == K_ADD_CO) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(lo[u]) : "v"(b) : "vcc");
            else if (KIND == K_ADDC_CO) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo[u]) : "v"(b) : "vcc");
            else if (KIND == K_AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo[u]) : "v"(b));
            else if (KIND == K_AND_OR_B32) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(lo[u]) : "v"(b), "v"(a));
            else if (KIND == K_LSHLREV_B32) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(lo[u]));
            else if (KIND == K_BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(lo[u]));
            else if (KIND == K_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(lo[u]) : "v"(b));
            else if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo[u]) : "v"(b) : "vcc");
            else if (KIND == K_MOV_B32) asm volatile("v_mov_b32 %0, %1" : "+v"(lo[u]) : "v"(b));
            else if (KIND == K_LSHRREV_B64) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[u]));
            else if (KIND == K_ASHRREV_I64) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(acc[u]));
            else if (KIND == K_LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(acc[u]) : "v"(acc[(u + 1) % UNROLL]));
            else if (KIND == K_MIX_MAD_ADD) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[u]) : "v"(b));
            } else if (KIND == K_MIX_MAD_2ADD) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[u]) : "v"(b));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[u]) : "v"(a));
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t r = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) r += acc[u] + lo[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

struct Result { std::string name; int waves_per_simd; double tops, cyc_per_inst, ghz; };

template <int KIND>
int run(uint64_t* d_out, uint64_t* d_clk, int cus, int blocks_per_cu, std::vector<Result>& res) {
    const int blocks = cus * blocks_per_cu;   // a 256-lane block puts one wave on each SIMD of its CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; r++) probe<KIND><<<blocks, 256>>>(d_out, d_clk, 12345);
    CK(hipDeviceSynchronize());
    const int reps = 8;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) probe<KIND><<<blocks, 256>>>(d_out, d_clk, 12345 + r);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> clk(2 * (size_t)blocks);
    CK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; b++) { cyc += (double)clk[2 * b]; real += (double)clk[2 * b + 1]; }
    const double ghz = cyc / real * 0.1;                      // s_memrealtime ticks at 100 MHz
    const double lane_ops = (double)reps * blocks * 256.0 * ITERS * UNROLL * kind_insts[KIND];
    const double tops = lane_ops / (ms * 1e-3) / 1e12;
    // in-kernel: cycles a wave needs per instruction, times the waves sharing the SIMD -> SIMD cycles per wave-instruction
    const double per_wave = (cyc / blocks) / ((double)ITERS * UNROLL * kind_insts[KIND]);
    const double simd_cyc = per_wave / blocks_per_cu;
    printf("%-34s waves/SIMD %d  %7.2f T lane-ops/s  %5.2f cycles per wave-instruction per SIMD  clock %.2f GHz\n",
           kind_name[KIND], blocks_per_cu, tops, simd_cyc, ghz);
    res.push_back({kind_name[KIND], blocks_per_cu, tops, simd_cyc, ghz});
    return 0;
}

template <int KIND>
int sweep(uint64_t* d_out, uint64_t* d_clk, int cus, std::vector<Result>& res) {
    for (int w : {1, 2, 4, 8}) if (run<KIND>(d_out, d_clk, cus, w, res)) return 1;
    return 0;
}

template <int K>
int sweep_all(uint64_t* d_out, uint64_t* d_clk, int cus, std::vector<Result>& res) {
    if constexpr (K < K_COUNT) {
        if (sweep<K>(d_out, d_clk, cus, res)) return 1;
        return sweep_all<K + 1>(d_out, d_clk, cus, res);
    } else {
        return 0;
    }
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int cus = p.multiProcessorCount;
    uint64_t *d, *c;
    CK(hipMalloc(&d, (size_t)cus * 8 * 256 * 8));
    CK(hipMalloc(&c, (size_t)cus * 8 * 2 * 8));
    std::vector<Result> res;
    if (sweep_all<0>(d, c, cus, res)) return 1;
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        if (!f) { printf("cannot write %s\n", argv[1]); return 1; }
        fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"chains_per_lane\": %d, \"rows\": [\n", p.gcnArchName, cus, UNROLL);
        for (size_t i = 0; i < res.size(); i++)
            fprintf(f, "  {\"inst\": \"%s\", \"waves_per_simd\": %d, \"t_lane_ops_per_s\": %.3f, \"simd_cycles_per_wave_inst\": %.3f, \"clock_ghz\": %.3f}%s\n",
                    res[i].name.c_str(), res[i].waves_per_simd, res[i].tops, res[i].cyc_per_inst, res[i].ghz, i + 1 < res.size() ? "," : "");
        fprintf(f, "]}\n");
        fclose(f);
    }
    CK(hipFree(d)); CK(hipFree(c));
    return 0;
}
