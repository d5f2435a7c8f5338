// Radix-2^r Stockham NTT over BN254 Fr for gfx950.
//
// A transform of N = 2^L points runs as ceil(L / 8) out-of-place passes.  Pass p has radix R = 2^r and
// Ns = product of the earlier radices; workgroup tiles are R x J elements (R*J = 2048, 64 KiB of LDS):
//   load   x[t][c] = in[(j0 + c) + t * N/R] * w_N^(t * ((j0+c) mod Ns) * N/(Ns R))      (J*32 B segments)
//   R-point DIF butterflies on every column in LDS, twiddles w_R^k from an LDS table
//   store  out[(j / Ns) * Ns * R + (j mod Ns) + q * Ns] = X[q]                           (natural order)
// so input and output of the whole transform are in natural order, like `best_fft`.
// Fused into the first load: zero padding beyond n_in and the zeta-coset scaling of `coeff_to_extended`;
// fused into the last store: the 1/n (and inverse zeta) scaling and the truncation of `extended_to_coeff`.
// Powers of omega come from two L2-resident tables (w^i for i < 2^h, w^(i 2^h)) and one multiplication.
#include <cstring>
#include "common.h"
#include "ntt.h"

namespace zkhip {

constexpr int NTT_THREADS = 256;
constexpr int TILE_LOG = 11;


struct NttPassArgs {
    const fe* in;
    fe* out;
    const fe* lo;   // w^i, i < 2^h
    const fe* hi;   // w^(i * 2^h)
    uint32_t h;
    uint32_t logN, r, logNs, logJ;
    uint32_t n_in;     // inputs at index >= n_in read as zero
    uint32_t n_out;    // outputs at index >= n_out are not stored
    uint32_t pre_mode;   // 0 none, 1: multiply input i by pre[i % 3]
    uint32_t post_mode;  // 0 none, 1: multiply output by post[0], 2: multiply output i by post[i % 3]
    fe pre[3];
    fe post[3];
    size_t in_stride, out_stride;   // elements between consecutive batch members
};

ZK_DEV fe omega_pow(const NttPassArgs& a, uint32_t e) {
    const fe l = load_fe(a.lo + (e & ((1u << a.h) - 1)));
    const uint32_t eh = e >> a.h;
    if (eh == 0) return l;
    return Fr::mul(l, load_fe(a.hi + eh));
}

ZK_DEV uint32_t bitrev(uint32_t v, uint32_t bits) { return bits ? (__brev(v) >> (32 - bits)) : 0; }

__global__ void __launch_bounds__(NTT_THREADS) ntt_pass_kernel(const NttPassArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    fe* x = reinterpret_cast<fe*>(smem_raw);                 // [R][J]
    const uint32_t R = 1u << a.r, J = 1u << a.logJ, N = 1u << a.logN;
    fe* tw = x + (R << a.logJ);                              // [R/2]
    const uint32_t tid = threadIdx.x;
    const uint32_t j0 = blockIdx.x << a.logJ;
    const fe* in = a.in + (size_t)blockIdx.y * a.in_stride;
    fe* out = a.out + (size_t)blockIdx.y * a.out_stride;
    const uint32_t rowStride = N >> a.r;                      // N / R
    const uint32_t nsMask = (1u << a.logNs) - 1;
    const uint32_t twShift = a.logN - a.logNs - a.r;          // log2(N / (Ns R))
    const uint32_t tile = R << a.logJ;

    // tile-local twiddles w_R^k = w_N^(k N/R)
    for (uint32_t k = tid; k < (R >> 1); k += NTT_THREADS) x[tile + k] = omega_pow(a, k << (a.logN - a.r));

    for (uint32_t idx = tid; idx < tile; idx += NTT_THREADS) {
        const uint32_t c = idx & (J - 1), t = idx >> a.logJ;
        const uint32_t j = j0 + c;
        const uint32_t gi = j + t * rowStride;
        fe v;
        if (gi < a.n_in) {
            v = load_fe(in + gi);
            if (a.pre_mode == 1) {
                const uint32_t m = gi % 3;
                if (m) v = Fr::mul(v, a.pre[m]);
            }
            if (a.logNs) {
                const uint32_t e = (t * (j & nsMask)) << twShift;
                if (e) v = Fr::mul(v, omega_pow(a, e));
            }
        } else {
            v = Fr::zero();
        }
        x[idx] = v;
    }
    __syncthreads();

    const uint32_t nbf = tile >> 1;
    for (uint32_t s = 0; s < a.r; s++) {
        const uint32_t logHalf = a.r - 1 - s, half = 1u << logHalf;
        for (uint32_t bf = tid; bf < nbf; bf += NTT_THREADS) {
            const uint32_t c = bf & (J - 1), p = bf >> a.logJ;
            const uint32_t i = p & (half - 1), blk = p >> logHalf;
            const uint32_t row0 = (blk << (logHalf + 1)) + i, row1 = row0 + half;
            const fe u = x[(row0 << a.logJ) + c], v = x[(row1 << a.logJ) + c];
            x[(row0 << a.logJ) + c] = Fr::add(u, v);
            fe d = Fr::sub(u, v);
            const uint32_t k = i << s;
            if (k) d = Fr::mul(d, tw[k]);
            x[(row1 << a.logJ) + c] = d;
        }
        __syncthreads();
    }

    // store: X[q] sits in LDS row bitrev(q)
    for (uint32_t idx = tid; idx < tile; idx += NTT_THREADS) {
        uint32_t c, q;
        if (a.logNs == 0) { q = idx & (R - 1); c = idx >> a.r; }     // outputs of one j are contiguous
        else { c = idx & (J - 1); q = idx >> a.logJ; }
        const uint32_t j = j0 + c;
        const uint32_t oi = ((j >> a.logNs) << (a.logNs + a.r)) + (j & nsMask) + (q << a.logNs);
        if (oi < a.n_out) {
            fe v = x[(bitrev(q, a.r) << a.logJ) + c];
            if (a.post_mode == 1) v = Fr::mul(v, a.post[0]);
            else if (a.post_mode == 2) v = Fr::mul(v, a.post[oi % 3]);
            store_fe(out + oi, v);
        }
    }
}

// out[i] = in[i] * scale[i % period]  (divide_by_vanishing_poly and friends)
__global__ void __launch_bounds__(256) scale_periodic_kernel(fe* a, const fe* scale, uint32_t period_mask, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_fe(a + i, Fr::mul(load_fe(a + i), load_fe(scale + (i & period_mask))));
}

// ------------------------------------------------------------------ host side ------------------------
NttPlan::~NttPlan() {
    if (d_lo) (void)hipFree(d_lo);
    if (d_hi) (void)hipFree(d_hi);
}

int ntt_plan_create(zkhip_ctx* ctx, uint32_t log_n, const uint64_t omega[4], NttPlan** out) {
    ZK_CHECK(ctx, log_n >= 1 && log_n <= 28, "ntt: log_n %u outside [1, 28]", log_n);
    // omega must have order exactly 2^log_n
    uint64_t t[4];
    memcpy(t, omega, 32);
    for (uint32_t i = 0; i + 1 < log_n; i++) h_mul(HFR, t, t, t);
    uint64_t minus_one[4], zero[4] = {0, 0, 0, 0};
    h_sub(HFR, minus_one, zero, HFR.r);
    ZK_CHECK(ctx, h_eq(t, minus_one), "ntt: omega does not have order 2^%u", log_n);

    auto* plan = new NttPlan();
    plan->log_n = log_n;
    memcpy(plan->omega, omega, 32);
    plan->h = (log_n + 1) / 2;
    const size_t nlo = (size_t)1 << plan->h, nhi = (size_t)1 << (log_n - plan->h);
    std::vector<uint64_t> lo(nlo * 4), hi(nhi * 4);
    memcpy(&lo[0], HFR.r, 32);
    for (size_t i = 1; i < nlo; i++) h_mul(HFR, &lo[4 * i], &lo[4 * (i - 1)], omega);
    uint64_t step[4];
    h_mul(HFR, step, &lo[4 * (nlo - 1)], omega);   // omega^(2^h)
    memcpy(&hi[0], HFR.r, 32);
    for (size_t i = 1; i < nhi; i++) h_mul(HFR, &hi[4 * i], &hi[4 * (i - 1)], step);
    hipError_t e = hipMalloc(&plan->d_lo, nlo * 32);
    if (e == hipSuccess) e = hipMalloc(&plan->d_hi, nhi * 32);
    if (e == hipSuccess) e = hipMemcpy(plan->d_lo, lo.data(), nlo * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(plan->d_hi, hi.data(), nhi * 32, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        delete plan;
        return set_error(ctx, ZKHIP_ERR_HIP, "ntt plan upload failed: %s", hipGetErrorString(e));
    }
    // split log_n into the fewest passes of radix <= 2^8, as even as possible
    const uint32_t npass = (log_n + 7) / 8;
    uint32_t left = log_n;
    for (uint32_t p = 0; p < npass; p++) {
        const uint32_t r = (left + (npass - p) - 1) / (npass - p);
        plan->radix.push_back(r);
        left -= r;
    }
    *out = plan;
    return ZKHIP_OK;
}

int ntt_init(zkhip_ctx* ctx) {
    ZK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_pass_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    return ZKHIP_OK;
}

// Runs the transform.  `src` holds n_in valid elements per batch member (stride src_stride); the result
// (first n_out elements) lands in `dst` (stride dst_stride).  src may equal dst.  tmp0/tmp1: scratch of
// batch * 2^log_n elements each (tmp1 only needed when the pass count parity requires it).
int ntt_run(zkhip_ctx* ctx, const NttPlan* plan, const fe* src, size_t src_stride, uint32_t n_in, fe* dst,
            size_t dst_stride, uint32_t n_out, size_t batch, const NttFuse& fuse, fe* tmp0, fe* tmp1) {
    const uint32_t L = plan->log_n;
    const size_t N = (size_t)1 << L;
    const uint32_t npass = (uint32_t)plan->radix.size();
    ZK_CHECK(ctx, batch >= 1 && batch <= 65535, "ntt: batch %zu outside [1, 65535]", batch);
    // ping-pong so that the last pass writes dst and no pass runs in place
    // sequence of buffers: src -> b1 -> b2 -> ... -> dst
    std::vector<fe*> bufs(npass + 1);
    std::vector<size_t> strides(npass + 1);
    bufs[0] = const_cast<fe*>(src);
    strides[0] = src_stride;
    bufs[npass] = dst;
    strides[npass] = dst_stride;
    // intermediate buffers alternate between tmp0 and `alt`, walking back from dst; dst itself serves as
    // `alt` when it is a distinct full-size buffer, which saves the second scratch buffer
    fe* alt = (src != dst && dst_stride == N && n_out == N) ? dst : tmp1;
    for (uint32_t p = npass - 1; p >= 1; p--) {
        bufs[p] = ((npass - p) & 1) ? tmp0 : alt;
        strides[p] = N;
    }
    if (npass >= 2 && bufs[1] == src) return set_error(ctx, ZKHIP_ERR_INVALID, "ntt: scratch aliases source");
    uint32_t logNs = 0;
    TimedRegion tr(ctx, "ntt");
    for (uint32_t p = 0; p < npass; p++) {
        NttPassArgs a;
        a.in = bufs[p];
        a.out = bufs[p + 1];
        a.in_stride = strides[p];
        a.out_stride = strides[p + 1];
        a.lo = (const fe*)plan->d_lo;
        a.hi = (const fe*)plan->d_hi;
        a.h = plan->h;
        a.logN = L;
        a.r = plan->radix[p];
        a.logNs = logNs;
        uint32_t logJ = TILE_LOG - a.r;
        if (logJ > L - a.r) logJ = L - a.r;
        a.logJ = logJ;
        a.n_in = (p == 0) ? n_in : (uint32_t)N;
        a.n_out = (p == npass - 1) ? n_out : (uint32_t)N;
        a.pre_mode = (p == 0) ? fuse.pre_mode : 0;
        a.post_mode = (p == npass - 1) ? fuse.post_mode : 0;
        memcpy(a.pre, fuse.pre, sizeof a.pre);
        memcpy(a.post, fuse.post, sizeof a.post);
        const uint32_t tile = 1u << (a.r + logJ);
        const size_t lds = ((size_t)tile + (1u << a.r) / 2 + 1) * sizeof(fe);
        dim3 grid((uint32_t)(N >> (a.r + logJ)), (uint32_t)batch);
        hipLaunchKernelGGL(ntt_pass_kernel, grid, dim3(NTT_THREADS), lds, ctx->stream, a);
        ZK_HIP(ctx, hipGetLastError());
        logNs += a.r;
    }
    return ZKHIP_OK;
}

int scale_periodic(zkhip_ctx* ctx, fe* a, const fe* d_scale, uint32_t period, size_t n) {
    hipLaunchKernelGGL(scale_periodic_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, d_scale,
                       period - 1, n);
    ZK_HIP(ctx, hipGetLastError());
    return ZKHIP_OK;
}

}  // namespace zkhip
