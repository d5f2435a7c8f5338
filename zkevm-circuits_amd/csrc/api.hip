// libzkhip C ABI: context, memory, field vector ops, NTT and EvaluationDomain entry points.
#include <cstring>
#include <memory>
#include "common.h"
#include "ntt.h"
#include "wide.hip.h"

using namespace zkhip;

namespace zkhip {
int msm_init(zkhip_ctx* ctx);
void msm_drop_ctx(zkhip_ctx* ctx);
int poly_ops_init(zkhip_ctx* ctx);
void evalh_drop_ctx(zkhip_ctx* ctx);

int set_error(zkhip_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    return code;
}

int ensure_scratch(zkhip_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return ZKHIP_OK;
    if (ctx->scratch) {
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ZK_HIP(ctx, hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    bytes = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    ZK_HIP(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return ZKHIP_OK;
}

TimedRegion::TimedRegion(zkhip_ctx* c, const char* f, hipStream_t s) : ctx(c), family(f), stream(s ? s : c->stream) {
    if (!ctx->timing || !family) return;
    // every bracket costs the stream two event packets (a few microseconds of idle queue each): level 2 keeps only the two
    // kernels a roofline is quoted for, so that a measured step is disturbed as little as possible
    if (ctx->timing == 2 && strcmp(family, "msm_accumulate") != 0 && strcmp(family, "ntt") != 0) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        e0 = e1 = nullptr;
        return;
    }
    (void)hipEventRecord(e0, stream);
}
TimedRegion::~TimedRegion() {
    if (!e0) return;
    (void)hipEventRecord(e1, stream);
    ctx->timers[family].pending.emplace_back(e0, e1);
}

// ---- element-wise kernels --------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) binop_kernel(const fe* a, const fe* b, fe* o, size_t n, int op) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe x = load_fe(a + i), y = load_fe(b + i);
    fe r;
    if (op == 0) r = F::add(x, y);
    else if (op == 1) r = F::sub(x, y);
    else if (op == 2) r = F::mul(x, y);
    else r = F::mul(F::sqr(x), y);   // op 3: x^2 * y (exercises the squaring path)
    store_fe(o + i, r);
}

// a^(p-2) by square-and-multiply; one element per lane (batch inversion for callers that only need a few
// thousand inverses; the grand-product scans use the prefix-product form instead)
template <class F, class FP>
__global__ void __launch_bounds__(256) invert_kernel(const fe* a, fe* o, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe x = load_fe(a + i);
    fe acc = F::one();
    uint32_t e[8];
#pragma unroll
    for (int k = 0; k < 8; k++) e[k] = FP::P[k];
    e[0] -= 2;
    for (int bit = 255; bit >= 0; bit--) {
        acc = F::sqr(acc);
        if ((e[bit >> 5] >> (bit & 31)) & 1) acc = F::mul(acc, x);
    }
    store_fe(o + i, acc);
}

constexpr int PROBE_CHAINS = 4;
__global__ void __launch_bounds__(256) probe_kernel(fe* io, int iters) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe x[PROBE_CHAINS];
    const fe y = load_fe(io + g);
#pragma unroll
    for (int c = 0; c < PROBE_CHAINS; c++) { x[c] = y; x[c].l[0] ^= c; x[c].l[7] &= 0x0fffffffu; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < PROBE_CHAINS; c++) x[c] = Fr::mul(x[c], y);
    }
    fe r = x[0];
#pragma unroll
    for (int c = 1; c < PROBE_CHAINS; c++) r = Fr::add(r, x[c]);
    store_fe(io + g, r);
}

// ---- self-test of the lazy arithmetic ----------------------------------------------------------------
// The multiplier of wide.hip.h leans on how the compiler lowers it (hidden digit signs, one chain of multiply-adds per
// column): this checks every form of it against the packed Montgomery arithmetic on the device it will run on.
// Returns a bit mask of the failed checks of (a, b, c, d).
template <class FP, bool CHAIN>
__device__ uint32_t wide_check(const fe& a, const fe& b, const fe& c, const fe& d) {
    using F = Field<FP>;
    using W = Wide<FP, CHAIN>;
    uint32_t bad = 0;
    const fw A = W::from_fe(a), B = W::from_fe(b), C = W::from_fe(c), D = W::from_fe(d);
    const fe k32 = W::to_fe(W::one());   // 2^261 mod p: packed_mul(wide result, 32 R) undoes the wider radix
    if (!F::eq(W::to_fe(A), a)) bad |= 1u;
    if (!F::eq(F::mul(W::to_fe(W::mul(A, B)), k32), F::mul(a, b))) bad |= 2u;
    if (!F::eq(F::mul(W::to_fe(W::sqr(A)), k32), F::sqr(a))) bad |= 4u;
    const fw AB = W::sub(A, B), CD = W::sub(C, D);   // lazy differences as operands
    if (!F::eq(F::mul(W::to_fe(W::mul(AB, CD)), k32), F::mul(F::sub(a, b), F::sub(c, d)))) bad |= 8u;
    if (!F::eq(F::mul(W::to_fe(W::sqr(AB)), k32), F::sqr(F::sub(a, b)))) bad |= 16u;
    if (!F::eq(F::mul(W::to_fe(W::mul(W::add(A, B), C)), k32), F::mul(F::add(a, b), c))) bad |= 32u;
    if (!F::eq(F::mul(W::to_fe(W::mul_sub_mul(AB, C, D, CD)), k32),
               F::sub(F::mul(F::sub(a, b), c), F::mul(d, F::sub(c, d))))) bad |= 64u;
    if (!F::eq(F::mul(W::to_fe(W::mul_add_mul(A, CD, AB, D)), k32),
               F::add(F::mul(a, F::sub(c, d)), F::mul(F::sub(a, b), d)))) bad |= 128u;
    // a chain with no reduction in between, negated operands
    fw ch = W::sub(W::mul(AB, CD), W::mul(W::cneg(A, true), B));
    ch = W::mul(W::norm(ch), C);
    const fe ref = F::mul(F::add(F::mul(F::sub(a, b), F::sub(c, d)), F::mul(a, b)), c);
    if (!F::eq(F::mul(F::mul(W::to_fe(ch), k32), k32), ref)) bad |= 256u;
    if (!W::maybe_zero(W::sub(A, A)) || !W::is_zero(W::sub(A, A)) || W::is_zero(AB) != F::eq(a, b)) bad |= 512u;
    return bad;
}
__global__ void __launch_bounds__(256) wide_selftest_kernel(const fe* in, size_t quads, uint32_t* bad /*[4]*/) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= quads) return;
    fe v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = load_fe(in + 4 * i + k);
    // every eighth quadruple repeats an operand, the next one uses the extremes 0 and p - 1
    if ((i & 7) == 1) v[1] = v[0];
    if ((i & 7) == 2) {
        v[2] = Fr::zero();
#pragma unroll
        for (int k = 0; k < 8; k++) v[3].l[k] = 0;
    }
    fe q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = v[k];
    if ((i & 7) == 2) q[3] = Fq::neg(Fq::one()), v[3] = Fr::neg(Fr::one());
    uint32_t r;
    if ((r = wide_check<FrParams, false>(v[0], v[1], v[2], v[3]))) atomicOr(bad + 0, r);
    if ((r = wide_check<FrParams, true>(v[0], v[1], v[2], v[3]))) atomicOr(bad + 1, r);
    if ((r = wide_check<FqParams, false>(q[0], q[1], q[2], q[3]))) atomicOr(bad + 2, r);
    if ((r = wide_check<FqParams, true>(q[0], q[1], q[2], q[3]))) atomicOr(bad + 3, r);
}

}  // namespace zkhip

// ====================================================================================================
extern "C" {

const char* zkhip_version(void) { return "zkhip 0.1 (gfx950)"; }

int zkhip_create(int device, zkhip_ctx** out) {
    if (!out) return ZKHIP_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return ZKHIP_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return ZKHIP_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ZKHIP_ERR_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ZKHIP_ERR_UNSUPPORTED;   // kernels are built for gfx950 only
    auto* ctx = new zkhip_ctx();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return ZKHIP_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    if (ntt_init(ctx) != ZKHIP_OK || msm_init(ctx) != ZKHIP_OK || poly_ops_init(ctx) != ZKHIP_OK) {
        delete ctx;
        return ZKHIP_ERR_HIP;
    }
    *out = ctx;
    return ZKHIP_OK;
}

void zkhip_destroy(zkhip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    msm_drop_ctx(ctx);
    evalh_drop_ctx(ctx);
    for (auto& kv : ctx->timers)
        for (auto& pr : kv.second.pending) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    for (NttPlan* p : ctx->plans) delete p;
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->upload_stream) {
        (void)hipStreamSynchronize(ctx->upload_stream);
        (void)hipStreamDestroy(ctx->upload_stream);
        (void)hipEventDestroy(ctx->ev_upload_base);
        (void)hipEventDestroy(ctx->ev_upload_done);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* zkhip_last_error(const zkhip_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int zkhip_set_stream(zkhip_ctx* ctx, void* hip_stream) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    int rc = zkhip_upload_fence(ctx);
    if (rc) return rc;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return ZKHIP_OK;
}
int zkhip_sync(zkhip_ctx* ctx) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    int rc = zkhip_upload_fence(ctx);     // overlapped copies still open count as the context's work
    if (rc) return rc;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}
int zkhip_dev_alloc(zkhip_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return ZKHIP_ERR_INVALID;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    ZK_HIP(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return ZKHIP_OK;
}
int zkhip_dev_free(zkhip_ctx* ctx, void* dptr) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    if (ctx->upload_stream) ZK_HIP(ctx, hipStreamSynchronize(ctx->upload_stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ZK_HIP(ctx, hipFree(dptr));
    return ZKHIP_OK;
}
int zkhip_mem_info(zkhip_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
    if (!ctx || !free_bytes || !total_bytes) return ZKHIP_ERR_INVALID;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    ZK_HIP(ctx, hipMemGetInfo(free_bytes, total_bytes));
    return ZKHIP_OK;
}
int zkhip_host_alloc(zkhip_ctx* ctx, size_t bytes, void** hptr) {
    if (!ctx || !hptr) return ZKHIP_ERR_INVALID;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    ZK_HIP(ctx, hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return ZKHIP_OK;
}
int zkhip_host_free(zkhip_ctx* ctx, void* hptr) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    if (ctx->upload_stream) ZK_HIP(ctx, hipStreamSynchronize(ctx->upload_stream));   // a copy may still read the buffer
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ZK_HIP(ctx, hipHostFree(hptr));
    return ZKHIP_OK;
}
int zkhip_upload_async(zkhip_ctx* ctx, void* dptr, const void* host, size_t bytes) {
    if (!ctx || (!dptr && bytes) || (!host && bytes)) return ZKHIP_ERR_INVALID;
    ZK_HIP(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return ZKHIP_OK;
}
int zkhip_upload_overlapped(zkhip_ctx* ctx, void* dptr, const void* host, size_t bytes) {
    if (!ctx || (!dptr && bytes) || (!host && bytes)) return ZKHIP_ERR_INVALID;
    if (!ctx->upload_stream) {
        ZK_HIP(ctx, hipSetDevice(ctx->device));
        ZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
        ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_upload_base, hipEventDisableTiming));
        ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_upload_done, hipEventDisableTiming));
    }
    if (!ctx->upload_open) {
        // the first copy after a fence: behind everything the compute stream holds now, which may still read the target
        ZK_HIP(ctx, hipEventRecord(ctx->ev_upload_base, ctx->stream));
        ZK_HIP(ctx, hipStreamWaitEvent(ctx->upload_stream, ctx->ev_upload_base, 0));
        ctx->upload_open = true;
    }
    ZK_HIP(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->upload_stream));
    return ZKHIP_OK;
}
int zkhip_upload_fence(zkhip_ctx* ctx) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    if (!ctx->upload_open) return ZKHIP_OK;
    ZK_HIP(ctx, hipEventRecord(ctx->ev_upload_done, ctx->upload_stream));
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_upload_done, 0));
    ctx->upload_open = false;
    return ZKHIP_OK;
}
int zkhip_upload(zkhip_ctx* ctx, void* dptr, const void* host, size_t bytes) {
    if (!ctx || (!dptr && bytes) || (!host && bytes)) return ZKHIP_ERR_INVALID;
    ZK_HIP(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}
int zkhip_copy_dev(zkhip_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    if (!ctx || ((!d_dst || !d_src) && bytes)) return ZKHIP_ERR_INVALID;
    if (bytes) ZK_HIP(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return ZKHIP_OK;
}
int zkhip_download(zkhip_ctx* ctx, void* host, const void* dptr, size_t bytes) {
    if (!ctx || (!dptr && bytes) || (!host && bytes)) return ZKHIP_ERR_INVALID;
    ZK_HIP(ctx, hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

// ---- timing ---------------------------------------------------------------------------------------
int zkhip_timing_enable(zkhip_ctx* ctx, int on) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ctx->timing = on == 2 ? 2 : (on != 0 ? 1 : 0);
    return ZKHIP_OK;
}
int zkhip_timing_reset(zkhip_ctx* ctx) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->timers) {
        for (auto& pr : kv.second.pending) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        kv.second.pending.clear();
        kv.second.total_us = 0;
        kv.second.launches = 0;
    }
    return ZKHIP_OK;
}
int zkhip_timing_get(zkhip_ctx* ctx, const char* family, double* avg_us, uint64_t* launches) {
    if (!ctx || !family) return ZKHIP_ERR_INVALID;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    auto& slot = ctx->timers[family];
    for (auto& pr : slot.pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            slot.total_us += 1e3 * ms;
            slot.launches++;
        }
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    slot.pending.clear();
    if (avg_us) *avg_us = slot.launches ? slot.total_us / (double)slot.launches : 0.0;
    if (launches) *launches = slot.launches;
    return ZKHIP_OK;
}

// ---- field vector ops -----------------------------------------------------------------------------
int zkhip_field_binop(zkhip_ctx* ctx, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, (field == 0 || field == 1) && op >= 0 && op <= 3, "field_binop: bad field/op");
    ZK_CHECK(ctx, n == 0 || (a && b && out), "field_binop: null pointer");
    if (n == 0) return ZKHIP_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_scratch(ctx, 3 * n * 32);
    if (rc) return rc;
    fe* da = (fe*)ctx->scratch;
    fe* db = da + n;
    fe* dout = db + n;
    ZK_HIP(ctx, hipMemcpyAsync(da, a, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(db, b, n * 32, hipMemcpyHostToDevice, ctx->stream));
    const dim3 grid((unsigned)((n + 255) / 256));
    if (field == 0) hipLaunchKernelGGL(binop_kernel<Fr>, grid, dim3(256), 0, ctx->stream, da, db, dout, n, op);
    else hipLaunchKernelGGL(binop_kernel<Fq>, grid, dim3(256), 0, ctx->stream, da, db, dout, n, op);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

int zkhip_field_batch_invert(zkhip_ctx* ctx, int field, const uint64_t* a, uint64_t* out, size_t n) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, field == 0 || field == 1, "batch_invert: bad field");
    ZK_CHECK(ctx, n == 0 || (a && out), "batch_invert: null pointer");
    if (n == 0) return ZKHIP_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_scratch(ctx, 2 * n * 32);
    if (rc) return rc;
    fe* da = (fe*)ctx->scratch;
    fe* dout = da + n;
    ZK_HIP(ctx, hipMemcpyAsync(da, a, n * 32, hipMemcpyHostToDevice, ctx->stream));
    const dim3 grid((unsigned)((n + 255) / 256));
    if (field == 0) hipLaunchKernelGGL((invert_kernel<Fr, FrParams>), grid, dim3(256), 0, ctx->stream, da, dout, n);
    else hipLaunchKernelGGL((invert_kernel<Fq, FqParams>), grid, dim3(256), 0, ctx->stream, da, dout, n);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

int zkhip_probe_mad_rate(zkhip_ctx* ctx, int iters, double* mads_per_second, double* fr_mul_per_second) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, iters > 0 && iters <= (1 << 20), "probe: iters out of range");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const int blocks = ctx->num_cus * 8;
    const size_t n = (size_t)blocks * 256;
    int rc = ensure_scratch(ctx, n * 32);
    if (rc) return rc;
    std::vector<uint64_t> h(n * 4);
    uint64_t s = 0x9E3779B97F4A7C15ULL;
    for (size_t i = 0; i < n * 4; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (i % 4 == 3) ? (s & 0x0fffffffffffffffULL) : s;
    }
    ZK_HIP(ctx, hipMemcpyAsync(ctx->scratch, h.data(), n * 32, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (fe*)ctx->scratch, 8);
    hipEvent_t e0, e1;
    ZK_HIP(ctx, hipEventCreate(&e0));
    ZK_HIP(ctx, hipEventCreate(&e1));
    ZK_HIP(ctx, hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (fe*)ctx->scratch, iters);
    ZK_HIP(ctx, hipEventRecord(e1, ctx->stream));
    ZK_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0;
    ZK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double muls = (double)n * iters * PROBE_CHAINS;
    const double mps = muls / (ms * 1e-3);
    if (fr_mul_per_second) *fr_mul_per_second = mps;
    if (mads_per_second) *mads_per_second = mps * 171.0;   // 81 product + 81 reduction + 9 digit multiplies
    return ZKHIP_OK;
}

int zkhip_selftest_arithmetic(zkhip_ctx* ctx, uint64_t seed, size_t quads, uint32_t out_failed[4]) {
    if (!ctx || !out_failed) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, quads >= 1 && quads <= ((size_t)1 << 24), "selftest: quads = %zu outside [1, 2^24]", quads);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_scratch(ctx, quads * 4 * 32 + 256);
    if (rc) return rc;
    std::vector<uint64_t> h(quads * 16);
    uint64_t s = seed ? seed : 0x9E3779B97F4A7C15ULL;
    for (size_t i = 0; i < h.size(); i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (i % 4 == 3) ? (s & 0x0fffffffffffffffULL) : s;   // below 2^252, i.e. below both moduli
    }
    uint32_t* d_bad = (uint32_t*)((unsigned char*)ctx->scratch + quads * 4 * 32);
    ZK_HIP(ctx, hipMemcpyAsync(ctx->scratch, h.data(), quads * 4 * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipMemsetAsync(d_bad, 0, 16, ctx->stream));
    hipLaunchKernelGGL(wide_selftest_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const fe*)ctx->scratch, quads, d_bad);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipMemcpyAsync(out_failed, d_bad, 16, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the pageable vector h must outlive its copy
    return ZKHIP_OK;
}

// ---- NTT ------------------------------------------------------------------------------------------
int zkhip_internal_get_plan(zkhip_ctx* ctx, uint32_t log_n, const uint64_t omega[4], zkhip::NttPlan** out);
int zkhip_internal_get_plan(zkhip_ctx* ctx, uint32_t log_n, const uint64_t omega[4], NttPlan** out) {
    for (NttPlan* p : ctx->plans)
        if (p->log_n == log_n && memcmp(p->omega, omega, 32) == 0) {
            *out = p;
            return ZKHIP_OK;
        }
    NttPlan* plan = nullptr;
    int rc = ntt_plan_create(ctx, log_n, omega, &plan);
    if (rc) return rc;
    ctx->plans.push_back(plan);
    *out = plan;
    return ZKHIP_OK;
}

int zkhip_ntt_dev(zkhip_ctx* ctx, uint64_t* d_a, uint32_t log_n, const uint64_t omega[4], size_t batch) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, d_a && omega, "ntt: null pointer");
    ZK_CHECK(ctx, batch >= 1, "ntt: empty batch");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    if (log_n == 0) return ZKHIP_OK;
    NttPlan* plan = nullptr;
    int rc = zkhip_internal_get_plan(ctx, log_n, omega, &plan);
    if (rc) return rc;
    const size_t N = (size_t)1 << log_n;
    rc = ensure_scratch(ctx, 2 * batch * N * 32);
    if (rc) return rc;
    fe* tmp0 = (fe*)ctx->scratch;
    fe* tmp1 = tmp0 + batch * N;
    NttFuse fuse;
    return ntt_run(ctx, plan, (const fe*)d_a, N, (uint32_t)N, (fe*)d_a, N, (uint32_t)N, batch, fuse, tmp0, tmp1);
}

int zkhip_ntt(zkhip_ctx* ctx, uint64_t* a, uint32_t log_n, const uint64_t omega[4]) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_CHECK(ctx, a && omega, "ntt: null pointer");
    ZK_CHECK(ctx, log_n <= 28, "ntt: log_n %u > 28", log_n);
    if (log_n == 0) return ZKHIP_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)1 << log_n;
    void* d = nullptr;
    ZK_HIP(ctx, hipMalloc(&d, N * 32));
    int rc = ZKHIP_OK;
    hipError_t e = hipMemcpyAsync(d, a, N * 32, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        rc = zkhip_ntt_dev(ctx, (uint64_t*)d, log_n, omega, 1);
        if (rc == ZKHIP_OK) e = hipMemcpyAsync(a, d, N * 32, hipMemcpyDeviceToHost, ctx->stream);
    }
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return set_error(ctx, ZKHIP_ERR_HIP, "ntt copy failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return set_error(ctx, ZKHIP_ERR_HIP, "ntt failed: %s", hipGetErrorString(e2));
    return ZKHIP_OK;
}

}  // extern "C"
