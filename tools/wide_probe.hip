// Development tool: checks the lazy signed-digit arithmetic (csrc/wide.hip.h) against the packed Montgomery arithmetic
// (csrc/field.hip.h) on the host and on the GPU, and measures its multiplication rate.
// Build: hipcc --offload-arch=gfx950 -O3 -I zkevm-circuits_amd/csrc -o tools/wide_probe tools/wide_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "wide.hip.h"

using namespace zkhip;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static fe rand_fe() {
    fe a;
    for (int i = 0; i < 8; i++) a.l[i] = (uint32_t)rnd();
    a.l[7] &= 0x0fffffffu;   // < 2^252 < p
    return a;
}

// one pass of mixed checks on (a, b, c, d); returns a bit mask of failed checks
template <class FP>
ZK_DEV uint32_t check4(const fe& a, const fe& b, const fe& c, const fe& d) {
    using F = Field<FP>;
    using W = Wide<FP>;
    uint32_t bad = 0;
    const fw A = W::from_fe(a), B = W::from_fe(b), C = W::from_fe(c), D = W::from_fe(d);
    const fe k32 = W::to_fe(W::one());   // 2^261 mod p as an integer = 32 in the packed Montgomery domain
    // round trip
    if (!F::eq(W::to_fe(A), a)) bad |= 1;
    // mul: packed(ab / 2^256) == packed_mul(wide(ab / 2^261), 32 R)
    if (!F::eq(F::mul(W::to_fe(W::mul(A, B)), k32), F::mul(a, b))) bad |= 2;
    if (!F::eq(F::mul(W::to_fe(W::sqr(A)), k32), F::sqr(a))) bad |= 4;
    // lazy differences as operands
    const fw AB = W::sub(A, B), CD = W::sub(C, D);
    if (!F::eq(F::mul(W::to_fe(W::mul(AB, CD)), k32), F::mul(F::sub(a, b), F::sub(c, d)))) bad |= 8;
    if (!F::eq(F::mul(W::to_fe(W::sqr(AB)), k32), F::sqr(F::sub(a, b)))) bad |= 16;
    // lazy sum next to a normalised value
    if (!F::eq(F::mul(W::to_fe(W::mul(W::add(A, B), C)), k32), F::mul(F::add(a, b), c))) bad |= 32;
    // to_fe of sums / negative values
    fw s = W::sub(W::sub(W::sub(A, B), C), W::dbl(D));
    if (!F::eq(W::to_fe(s), F::sub(F::sub(F::sub(a, b), c), F::dbl(d)))) bad |= 64;
    s = W::add(W::add(A, B), C);
    if (!F::eq(W::to_fe(s), F::add(F::add(a, b), c))) bad |= 128;
    // a chain with no reduction in between
    fw ch = W::sub(W::mul(AB, CD), W::mul(A, B));
    ch = W::mul(ch, C);
    const fe ref = F::mul(F::sub(F::mul(F::sub(a, b), F::sub(c, d)), F::mul(a, b)), c);
    if (!F::eq(F::mul(F::mul(W::to_fe(ch), k32), k32), ref)) bad |= 256;
    // zero tests
    if (!W::maybe_zero(W::sub(A, A)) || !W::is_zero(W::sub(A, A))) bad |= 512;
    fw kp;
    for (int i = 0; i < 9; i++) kp.d[i] = -3 * (int32_t)FP::P29[i];
    if (!W::maybe_zero(kp) || !W::is_zero(kp)) bad |= 1024;
    if (W::is_zero(AB) != F::eq(a, b)) bad |= 2048;
    if (!F::eq(W::to_fe(W::cneg(A, true)), F::neg(a)) || !F::eq(W::to_fe(W::cneg(A, false)), a)) bad |= 4096;
    // leaving the 2^261 domain: mul(x 2^261, 2^256 mod p) = x 2^256
    fe two5;
    for (int i = 0; i < 8; i++) two5.l[i] = FP::TWO5[i];
    const fw Aw = W::from_fe(F::mul(a, two5));            // a 2^256 -> a 2^261 (packed multiplication by 32 R)
    if (!F::eq(W::to_fe(W::mul(Aw, W::r256())), a)) bad |= 8192;
    return bad;
}

template <class FP>
__global__ void check_kernel(const fe* in, size_t n, uint32_t* bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = check4<FP>(in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]);
    if (r) atomicOr(bad, r);
}

template <class FP, bool WIDE>
__global__ void __launch_bounds__(256) rate_kernel(const fe* in, fe* out, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (WIDE) {
        using W = Wide<FP>;
        fw x = W::from_fe(in[2 * i]);
        const fw y = W::from_fe(in[2 * i + 1]);
        for (int k = 0; k < iters; k++) x = W::mul(x, y);
        out[i] = W::to_fe(x);
    } else {
        using F = Field<FP>;
        fe x = in[2 * i];
        const fe y = in[2 * i + 1];
        for (int k = 0; k < iters; k++) x = F::mul(x, y);
        out[i] = x;
    }
}

// issue-rate probe: CHAINS independent multiplication chains per lane, launched at one wave per SIMD
template <int CHAINS>
__global__ void __launch_bounds__(256) ilp_kernel(const fe* in, fe* out, int iters) {
    using W = Wide<FqParams>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fw x[CHAINS];
    const fw y = W::from_fe(in[2 * i + 1]);
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { x[c] = W::from_fe(in[2 * i]); x[c].d[0] ^= c; }
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) x[c] = W::mul(x[c], y);
    }
    fw r = x[0];
#pragma unroll
    for (int c = 1; c < CHAINS; c++) r = W::add(r, x[c]);
    out[i] = W::to_fe(W::norm(r));
}

int main() {
    const size_t n = 1 << 16;
    std::vector<fe> h(4 * n);
    for (auto& v : h) v = rand_fe();
    // edge values
    for (int i = 0; i < 8; i++) h[0].l[i] = 0;
    for (int i = 0; i < 8; i++) h[5].l[i] = FqParams::P[i];
    h[5].l[0] -= 1;   // p - 1 (Fq)
    h[9] = h[8];      // a == b
    uint32_t bad_q = 0, bad_r = 0;
    for (size_t i = 0; i < 4096; i++) {
        bad_q |= check4<FqParams>(h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
        bad_r |= check4<FrParams>(h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
    }
    printf("host checks: Fq bad mask %#x, Fr bad mask %#x\n", bad_q, bad_r);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        printf("no GPU: device part skipped\n");
        return (bad_q | bad_r) ? 1 : 0;
    }
    fe *d_in, *d_out;
    uint32_t* d_bad;
    CK(hipMalloc(&d_in, 4 * n * sizeof(fe)));
    CK(hipMalloc(&d_bad, 8));
    CK(hipMemcpy(d_in, h.data(), 4 * n * sizeof(fe), hipMemcpyHostToDevice));
    CK(hipMemset(d_bad, 0, 8));
    check_kernel<FqParams><<<(n + 255) / 256, 256>>>(d_in, n, d_bad);
    check_kernel<FrParams><<<(n + 255) / 256, 256>>>(d_in, n, d_bad + 1);
    uint32_t hb[2];
    CK(hipMemcpy(hb, d_bad, 8, hipMemcpyDeviceToHost));
    printf("device checks: Fq bad mask %#x, Fr bad mask %#x\n", hb[0], hb[1]);

    const size_t lanes = 256 * 256 * 12;   // 12 waves per SIMD offered; occupancy set by registers
    std::vector<fe> hin(2 * lanes);
    for (auto& v : hin) v = rand_fe();
    fe* d_rin;
    CK(hipMalloc(&d_rin, 2 * lanes * sizeof(fe)));
    CK(hipMalloc(&d_out, lanes * sizeof(fe)));
    CK(hipMemcpy(d_rin, hin.data(), 2 * lanes * sizeof(fe), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int wide = 0; wide < 2; wide++) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            if (wide) rate_kernel<FqParams, true><<<lanes / 256, 256>>>(d_rin, d_out, iters);
            else rate_kernel<FqParams, false><<<lanes / 256, 256>>>(d_rin, d_out, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%s mul chain: %.3f ms, %.1f G mul/s\n", wide ? "wide  " : "packed", best, (double)lanes * iters / best / 1e6);
    }
    // one wave per SIMD (256 CUs x 4): how much of the issue rate can a lone wave use, and does ILP help?
    {
        const int blocks = 256;
        for (int chains = 1; chains <= 4; chains *= 2) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                if (chains == 1) ilp_kernel<1><<<blocks, 256>>>(d_rin, d_out, iters);
                else if (chains == 2) ilp_kernel<2><<<blocks, 256>>>(d_rin, d_out, iters);
                else ilp_kernel<4><<<blocks, 256>>>(d_rin, d_out, iters);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("one wave per SIMD, %d chain(s): %.3f ms, %.1f G mul/s\n", chains, best,
                   (double)blocks * 256 * iters * chains / best / 1e6);
        }
    }
    return (bad_q | bad_r | hb[0] | hb[1]) ? 1 : 0;
}
