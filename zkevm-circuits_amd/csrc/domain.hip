// `poly::EvaluationDomain<Fr>` on the device: domain constants on the host (a few hundred field
// operations), every per-element step as an NTT pass or a fused scaling.
#include <cstring>
#include "domain.h"

using namespace zkhip;

namespace zkhip {

static void to_fe(fe& o, const uint64_t v[4]) { memcpy(o.l, v, 32); }

int domain_coeff_to_extended(zkhip_domain* dom, const fe* d_coeff, fe* d_out, size_t batch, fe* tmp0, fe* tmp1, bool split) {
    NttFuse fuse;
    fuse.pre_mode = 1;
    fuse.split_out = split ? 1 : 0;
    to_fe(fuse.pre[0], HFR.r);
    to_fe(fuse.pre[1], dom->g_coset);
    to_fe(fuse.pre[2], dom->g_coset_inv);
    const size_t n = (size_t)1 << dom->k, en = (size_t)1 << dom->extended_k;
    return ntt_run(dom->ctx, dom->plan_ext_fwd, d_coeff, n, (uint32_t)n, d_out, en, (uint32_t)en, batch, fuse, tmp0, tmp1);
}

// ---- coset subsets (the unit of the multi-GPU split of evaluate_h) -------------------------------------------
// t[i] = c[i] * zeta^(i mod 3) * extended_omega^(j i): the coefficients of p(zeta extended_omega^j X)
// t[i] = (zeta * extended_omega^j)^i: the factor that moves coefficient i to coset j
__global__ void __launch_bounds__(256) coset_powers_kernel(fe* t, uint32_t n, uint32_t j, fe one, fe zeta, fe zeta2,
                                                           const fe* om_lo, const fe* om_hi, uint32_t h) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe v = one;
    const uint32_t e = i * j;   // < 2^extended_k
    if (e) {
        v = load_fe(om_lo + (e & ((1u << h) - 1)));
        if (e >> h) v = Fr::mul(v, load_fe(om_hi + (e >> h)));
    }
    const uint32_t r = i % 3;
    if (r == 1) v = Fr::mul(v, zeta);
    else if (r == 2) v = Fr::mul(v, zeta2);
    store_fe(t + i, v);
}
// A group of (coefficient vector, coset) pairs handled by one launch: blockIdx.y = member of the group.
constexpr int COSET_GROUP = 32;
struct CosetGroup {
    const fe* src[COSET_GROUP];    // coefficient vector of the member
    const fe* pow[COSET_GROUP];    // power table of its coset
    fe* dst[COSET_GROUP];          // compact output of its vector
    uint32_t slot[COSET_GROUP];    // position of its coset in the compact layout
};
// t[g][i] = src_g[i] * pow_g[i]
__global__ void __launch_bounds__(256) coset_prescale_kernel(const CosetGroup grp, fe* t, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (i >= n) return;
    store_fe(t + (size_t)g * n + i, Fr::mul(load_fe(grp.src[g] + i), load_fe(grp.pow[g] + i)));
}
// dst_g[q * num + slot_g] = t[g][q]
__global__ void __launch_bounds__(256) coset_interleave_kernel(const CosetGroup grp, const fe* t, uint32_t n, uint32_t num) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (q < n) store_fe(grp.dst[g] + (size_t)q * num + grp.slot[g], load_fe(t + (size_t)g * n + q));
}
// to_compact: dst[q * num + s] = src[q * 2^e + j[s]]; otherwise the reverse (rows of other cosets are left alone)
__global__ void __launch_bounds__(256) coset_gather_kernel(const fe* src, fe* dst, size_t rows, CosetList cl, uint32_t e, int to_compact) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= rows) return;
    const size_t q = c / cl.num, full = (q << e) + cl.j[c % cl.num];
    if (to_compact) store_fe(dst + c, load_fe(src + full));
    else store_fe(dst + full, load_fe(src + c));
}
__global__ void __launch_bounds__(256) coset_xpoints_kernel(fe* x, size_t rows, CosetList cl, uint32_t e, const fe* om_lo,
                                                            const fe* om_hi, uint32_t h) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= rows) return;
    const uint32_t idx = (uint32_t)(((c / cl.num) << e) + cl.j[c % cl.num]);
    fe zeta;
#pragma unroll
    for (int i = 0; i < 8; i++) zeta.l[i] = FrParams::ZETA[i];
    fe om = load_fe(om_lo + (idx & ((1u << h) - 1)));
    if (idx >> h) om = Fr::mul(om, load_fe(om_hi + (idx >> h)));
    store_fe(x + c, Fr::mul(om, zeta));
}

// a[m] *= s * w^m with w^m = lo[m mod 2^h] * hi[m >> h]: moves the coefficients of P(zeta X) to those of P(zeta w X)
__global__ void __launch_bounds__(256) scale_by_powers_kernel(fe* a, size_t n, fe s, const fe* om_lo, const fe* om_hi, uint32_t h) {
    const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    fe f = Fr::mul(s, load_fe(om_lo + (m & (((size_t)1 << h) - 1))));
    if (m >> h) f = Fr::mul(f, load_fe(om_hi + (m >> h)));
    store_fe(a + m, Fr::mul(load_fe(a + m), f));
}

int coset_list_check(zkhip_domain* dom, const uint32_t* cosets, uint32_t num, CosetList* out) {
    zkhip_ctx* ctx = dom->ctx;
    const uint32_t e = dom->extended_k - dom->k;
    ZK_CHECK(ctx, cosets && num >= 1 && num <= (1u << e) && (num & (num - 1)) == 0,
             "coset list: %u cosets of %u (a power of two, at most all of them)", num, 1u << e);
    ZK_CHECK(ctx, e <= 4, "coset list: 2^%u cosets", e);
    uint32_t seen = 0;
    out->num = num;
    for (uint32_t s = 0; s < 16; s++) out->j[s] = 0;
    for (uint32_t s = 0; s < num; s++) {
        ZK_CHECK(ctx, cosets[s] < (1u << e) && !((seen >> cosets[s]) & 1), "coset list: coset %u out of range or repeated", cosets[s]);
        seen |= 1u << cosets[s];
        out->j[s] = cosets[s];
    }
    return ZKHIP_OK;
}

// members per launch: as many as fit a scratch budget of 1 GiB (four vectors of 2^k elements per member)
static size_t coset_group_size(const zkhip_domain* dom, size_t members) {
    const size_t n = (size_t)1 << dom->k;
    const size_t fit = std::max<size_t>(1, ((size_t)1 << 30) / (4 * n * 32));
    return std::min<size_t>(std::min<size_t>(members, (size_t)COSET_GROUP), fit);
}
size_t domain_coeff_to_cosets_tmp(const zkhip_domain* dom, const CosetList& cl, size_t batch) {
    return 4 * coset_group_size(dom, batch * cl.num) * ((size_t)1 << dom->k);
}
static int coset_power_table(zkhip_domain* dom, uint32_t j, const fe** out) {
    zkhip_ctx* ctx = dom->ctx;
    auto it = dom->d_coset_powers.find(j);
    if (it == dom->d_coset_powers.end()) {
        const uint32_t n = 1u << dom->k;
        void* d;
        ZK_HIP(ctx, hipMalloc(&d, (size_t)n * 32));
        fe one, zeta, zeta2;
        to_fe(one, HFR.r);
        to_fe(zeta, dom->g_coset);
        to_fe(zeta2, dom->g_coset_inv);   // zeta^2 = zeta^-1
        hipLaunchKernelGGL(coset_powers_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (fe*)d, n, j, one, zeta, zeta2,
                           (const fe*)dom->plan_ext_fwd->d_lo, (const fe*)dom->plan_ext_fwd->d_hi, dom->plan_ext_fwd->h);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            (void)hipFree(d);
            return set_error(ctx, ZKHIP_ERR_HIP, "coset powers: %s", hipGetErrorString(e));
        }
        it = dom->d_coset_powers.emplace(j, d).first;
    }
    *out = (const fe*)it->second;
    return ZKHIP_OK;
}

// Per (vector, coset) member: scale coefficient i by (zeta * extended_omega^j)^i, one size-2^k NTT, interleave into the
// compact layout.  Members run in groups: one scaling launch, one batched NTT and one interleave launch per group.
int domain_coeff_to_cosets(zkhip_domain* dom, const fe* d_coeff, size_t batch, const CosetList& cl, fe* d_out, fe* tmp) {
    zkhip_ctx* ctx = dom->ctx;
    const uint32_t n = 1u << dom->k;
    const fe* pow[16];
    for (uint32_t s = 0; s < cl.num; s++) {
        int rc = coset_power_table(dom, cl.j[s], &pow[s]);
        if (rc) return rc;
    }
    const size_t members = batch * cl.num, gs = coset_group_size(dom, members);
    fe *t_in = tmp, *t_out = tmp + gs * n, *t0 = tmp + 2 * gs * n, *t1 = tmp + 3 * gs * n;
    NttFuse fuse;
    for (size_t m0 = 0; m0 < members; m0 += gs) {
        const size_t cnt = std::min(gs, members - m0);
        CosetGroup grp;
        memset(&grp, 0, sizeof grp);
        for (size_t g = 0; g < cnt; g++) {
            const size_t b = (m0 + g) / cl.num, s = (m0 + g) % cl.num;
            grp.src[g] = d_coeff + b * n;
            grp.pow[g] = pow[s];
            grp.dst[g] = d_out + b * n * cl.num;
            grp.slot[g] = (uint32_t)s;
        }
        const dim3 grid((n + 255) / 256, (unsigned)cnt);
        hipLaunchKernelGGL(coset_prescale_kernel, grid, dim3(256), 0, ctx->stream, grp, t_in, n);
        ZK_HIP(ctx, hipGetLastError());
        int rc = ntt_run(ctx, dom->plan_fwd, t_in, n, n, t_out, n, n, cnt, fuse, t0, t1);
        if (rc) return rc;
        hipLaunchKernelGGL(coset_interleave_kernel, grid, dim3(256), 0, ctx->stream, grp, (const fe*)t_out, n, cl.num);
        ZK_HIP(ctx, hipGetLastError());
    }
    return ZKHIP_OK;
}

int domain_coset_xpoints(zkhip_domain* dom, const CosetList& cl, fe* d_x) {
    zkhip_ctx* ctx = dom->ctx;
    const size_t rows = ((size_t)1 << dom->k) * cl.num;
    hipLaunchKernelGGL(coset_xpoints_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, d_x, rows, cl,
                       dom->extended_k - dom->k, (const fe*)dom->plan_ext_fwd->d_lo, (const fe*)dom->plan_ext_fwd->d_hi,
                       dom->plan_ext_fwd->h);
    ZK_HIP(ctx, hipGetLastError());
    return ZKHIP_OK;
}

}  // namespace zkhip

extern "C" {

int zkhip_internal_get_plan(zkhip_ctx* ctx, uint32_t log_n, const uint64_t omega[4], zkhip::NttPlan** out);

int zkhip_domain_create(zkhip_ctx* ctx, uint32_t j, uint32_t k, zkhip_domain** out) {
    if (!ctx || !out) return ZKHIP_ERR_INVALID;
    *out = nullptr;
    ZK_CHECK(ctx, j >= 2, "domain: constraint degree j=%u must be >= 2", j);
    ZK_CHECK(ctx, k >= 1 && k <= 27, "domain: k=%u outside [1, 27]", k);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    auto dom = new zkhip_domain();
    dom->ctx = ctx;
    dom->k = k;
    dom->quotient_poly_degree = j - 1;
    const uint64_t n = 1ULL << k;
    uint32_t ek = k;
    while ((1ULL << ek) < n * (uint64_t)(j - 1)) ek++;
    if (ek > FrParams::S) {
        delete dom;
        return set_error(ctx, ZKHIP_ERR_INVALID, "domain: extended_k=%u exceeds the field's 2-adicity 28", ek);
    }
    dom->extended_k = ek;
    uint64_t w[4];
    memcpy(w, FrParams::ROOT_OF_UNITY, 32);
    for (uint32_t i = ek; i < FrParams::S; i++) h_mul(HFR, w, w, w);
    memcpy(dom->extended_omega, w, 32);
    h_inv(HFR, dom->extended_omega_inv, w);
    for (uint32_t i = k; i < ek; i++) h_mul(HFR, w, w, w);
    memcpy(dom->omega, w, 32);
    h_inv(HFR, dom->omega_inv, w);
    memcpy(dom->g_coset, FrParams::ZETA, 32);
    h_mul(HFR, dom->g_coset_inv, dom->g_coset, dom->g_coset);
    uint64_t t[4];
    h_from_u64(HFR, t, n);
    h_inv(HFR, dom->ifft_divisor, t);
    h_from_u64(HFR, t, 1ULL << ek);
    h_inv(HFR, dom->extended_ifft_divisor, t);
    // t_evaluations[i] = 1 / ((zeta * extended_omega^i)^n - 1)
    const size_t m = (size_t)1 << (ek - k);
    dom->t_evaluations.resize(4 * m);
    uint64_t cur[4];
    memcpy(cur, dom->g_coset, 32);
    for (size_t i = 0; i < m; i++) {
        uint64_t v[4];
        h_pow(HFR, v, cur, n);
        h_sub(HFR, v, v, HFR.r);
        h_inv(HFR, &dom->t_evaluations[4 * i], v);
        h_mul(HFR, cur, cur, dom->extended_omega);
    }
    hipError_t e = hipMalloc(&dom->d_t_evaluations, m * 32);
    if (e == hipSuccess) e = hipMemcpy(dom->d_t_evaluations, dom->t_evaluations.data(), m * 32, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        delete dom;
        return set_error(ctx, ZKHIP_ERR_HIP, "domain: t_evaluations upload failed: %s", hipGetErrorString(e));
    }
    int rc = zkhip_internal_get_plan(ctx, k, dom->omega, &dom->plan_fwd);
    if (!rc) rc = zkhip_internal_get_plan(ctx, k, dom->omega_inv, &dom->plan_inv);
    if (!rc) rc = zkhip_internal_get_plan(ctx, ek, dom->extended_omega, &dom->plan_ext_fwd);
    if (!rc) rc = zkhip_internal_get_plan(ctx, ek, dom->extended_omega_inv, &dom->plan_ext_inv);
    if (rc) {
        zkhip_domain_destroy(dom);
        return rc;
    }
    *out = dom;
    return ZKHIP_OK;
}

void zkhip_domain_destroy(zkhip_domain* dom) {
    if (!dom) return;
    if (dom->d_t_evaluations) (void)hipFree(dom->d_t_evaluations);
    if (dom->d_xpoints) (void)hipFree(dom->d_xpoints);
    for (auto& kv : dom->d_coset_powers) (void)hipFree(kv.second);
    delete dom;   // NTT plans are owned by the context
}

uint32_t zkhip_domain_k(const zkhip_domain* dom) { return dom ? dom->k : 0; }
uint32_t zkhip_domain_extended_k(const zkhip_domain* dom) { return dom ? dom->extended_k : 0; }
uint32_t zkhip_domain_quotient_poly_degree(const zkhip_domain* dom) { return dom ? dom->quotient_poly_degree : 0; }

int zkhip_domain_constant(const zkhip_domain* dom, int which, uint64_t out[4]) {
    if (!dom || !out || which < 0 || which > 7) return ZKHIP_ERR_INVALID;
    const uint64_t* src[8] = {dom->omega, dom->omega_inv, dom->extended_omega, dom->extended_omega_inv,
                              dom->g_coset, dom->g_coset_inv, dom->ifft_divisor, dom->extended_ifft_divisor};
    memcpy(out, src[which], 32);
    return ZKHIP_OK;
}

// ---- device-resident, batched ---------------------------------------------------------------------
static int scratch2(zkhip_domain* dom, size_t batch, uint32_t log_n, fe** t0, fe** t1) {
    const size_t N = (size_t)1 << log_n;
    int rc = ensure_scratch(dom->ctx, 2 * batch * N * 32);
    if (rc) return rc;
    *t0 = (fe*)dom->ctx->scratch;
    *t1 = *t0 + batch * N;
    return ZKHIP_OK;
}

int zkhip_lagrange_to_coeff_dev(zkhip_domain* dom, uint64_t* d_a, size_t batch) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_a && batch >= 1, "lagrange_to_coeff: null pointer or empty batch");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    fe *t0, *t1;
    int rc = scratch2(dom, batch, dom->k, &t0, &t1);
    if (rc) return rc;
    NttFuse fuse;
    fuse.post_mode = 1;
    memcpy(fuse.post[0].l, dom->ifft_divisor, 32);
    const size_t n = (size_t)1 << dom->k;
    return ntt_run(ctx, dom->plan_inv, (const fe*)d_a, n, (uint32_t)n, (fe*)d_a, n, (uint32_t)n, batch, fuse, t0, t1);
}

int zkhip_coeff_to_lagrange_dev(zkhip_domain* dom, uint64_t* d_a, size_t batch) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_a && batch >= 1, "coeff_to_lagrange: null pointer or empty batch");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    fe *t0, *t1;
    int rc = scratch2(dom, batch, dom->k, &t0, &t1);
    if (rc) return rc;
    NttFuse fuse;
    const size_t n = (size_t)1 << dom->k;
    return ntt_run(ctx, dom->plan_fwd, (const fe*)d_a, n, (uint32_t)n, (fe*)d_a, n, (uint32_t)n, batch, fuse, t0, t1);
}

int zkhip_coeff_to_extended_dev(zkhip_domain* dom, const uint64_t* d_coeff, uint64_t* d_out, size_t batch) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_coeff && d_out && batch >= 1, "coeff_to_extended: null pointer or empty batch");
    ZK_CHECK(ctx, (const void*)d_coeff != (const void*)d_out, "coeff_to_extended: in and out must differ");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    fe *t0, *t1;
    int rc = scratch2(dom, batch, dom->extended_k, &t0, &t1);
    if (rc) return rc;
    return domain_coeff_to_extended(dom, (const fe*)d_coeff, (fe*)d_out, batch, t0, t1);
}

int zkhip_coeff_to_extended_split_dev(zkhip_domain* dom, const uint64_t* d_coeff, uint64_t* d_out, size_t batch) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_coeff && d_out && batch >= 1, "coeff_to_extended_split: null pointer or empty batch");
    ZK_CHECK(ctx, (const void*)d_coeff != (const void*)d_out, "coeff_to_extended_split: in and out must differ");
    ZK_CHECK(ctx, dom->extended_k > dom->k, "coeff_to_extended_split: the extended domain has a single coset");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    fe *t0, *t1;
    int rc = scratch2(dom, batch, dom->extended_k, &t0, &t1);
    if (rc) return rc;
    return domain_coeff_to_extended(dom, (const fe*)d_coeff, (fe*)d_out, batch, t0, t1, true);
}

static int half_plans(zkhip_domain* dom) {
    if (dom->plan_half_fwd) return ZKHIP_OK;
    uint64_t w[4], wi[4];
    h_mul(HFR, w, dom->extended_omega, dom->extended_omega);
    h_mul(HFR, wi, dom->extended_omega_inv, dom->extended_omega_inv);
    int rc = zkhip_internal_get_plan(dom->ctx, dom->extended_k - 1, w, &dom->plan_half_fwd);
    if (!rc) rc = zkhip_internal_get_plan(dom->ctx, dom->extended_k - 1, wi, &dom->plan_half_inv);
    return rc;
}

int zkhip_coeff_to_extended_half_dev(zkhip_domain* dom, const uint64_t* d_coeff, uint64_t* d_out, size_t out_stride, size_t batch) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_coeff && d_out && batch >= 1, "coeff_to_extended_half: null pointer or empty batch");
    ZK_CHECK(ctx, (const void*)d_coeff != (const void*)d_out, "coeff_to_extended_half: in and out must differ");
    ZK_CHECK(ctx, dom->extended_k > dom->k, "coeff_to_extended_half: the extended domain has a single coset");
    const size_t n = (size_t)1 << dom->k, hn = (size_t)1 << (dom->extended_k - 1);
    ZK_CHECK(ctx, out_stride >= hn, "coeff_to_extended_half: members %zu elements apart, each %zu long", out_stride, hn);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = half_plans(dom);
    if (rc) return rc;
    if ((rc = ensure_scratch(ctx, 2 * batch * hn * 32))) return rc;
    fe* t0 = (fe*)ctx->scratch;
    fe* t1 = t0 + batch * hn;
    // values on zeta <w^2>: the same zeta^(i mod 3) scaling of the coefficients, then the transform of half the size
    NttFuse fuse;
    fuse.pre_mode = 1;
    to_fe(fuse.pre[0], HFR.r);
    to_fe(fuse.pre[1], dom->g_coset);
    to_fe(fuse.pre[2], dom->g_coset_inv);
    return ntt_run(ctx, dom->plan_half_fwd, (const fe*)d_coeff, n, (uint32_t)n, (fe*)d_out, out_stride, (uint32_t)hn, batch, fuse,
                   t0, t1);
}

int zkhip_extend_half_dev(zkhip_domain* dom, const uint64_t* d_even, uint64_t* d_odd, size_t batch) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_even && d_odd && batch >= 1, "extend_half: null pointer or empty batch");
    ZK_CHECK(ctx, (const void*)d_even != (const void*)d_odd, "extend_half: in and out must differ");
    ZK_CHECK(ctx, dom->extended_k > dom->k, "extend_half: the extended domain has a single coset");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t hk = dom->extended_k - 1;
    const size_t hn = (size_t)1 << hk;
    int rc = half_plans(dom);
    if (rc) return rc;
    // d_odd = inverse transform of the values on zeta <w^2> (the coefficients of P(zeta X), not yet divided by their
    // count); coefficient m times w^m / count; forward transform: P on zeta w <w^2>, the odd-indexed points
    rc = ensure_scratch(ctx, 2 * batch * hn * 32);
    if (rc) return rc;
    fe* t0 = (fe*)ctx->scratch;
    fe* t1 = t0 + batch * hn;
    NttFuse none;
    rc = ntt_run(ctx, dom->plan_half_inv, (const fe*)d_even, hn, (uint32_t)hn, (fe*)d_odd, hn, (uint32_t)hn, batch, none, t0, t1);
    if (rc) return rc;
    uint64_t cnt[4], inv[4];
    h_from_u64(HFR, cnt, (uint64_t)hn);
    h_inv(HFR, inv, cnt);
    fe s;
    memcpy(s.l, inv, 32);
    for (size_t b = 0; b < batch; b++) {
        hipLaunchKernelGGL(scale_by_powers_kernel, dim3((unsigned)((hn + 255) / 256)), dim3(256), 0, ctx->stream,
                           (fe*)d_odd + b * hn, hn, s, (const fe*)dom->plan_ext_fwd->d_lo, (const fe*)dom->plan_ext_fwd->d_hi,
                           dom->plan_ext_fwd->h);
    }
    ZK_HIP(ctx, hipGetLastError());
    return ntt_run(ctx, dom->plan_half_fwd, (const fe*)d_odd, hn, (uint32_t)hn, (fe*)d_odd, hn, (uint32_t)hn, batch, none, t0, t1);
}

int zkhip_extended_to_coeff_dev(zkhip_domain* dom, uint64_t* d_ext, uint64_t* d_out, size_t batch) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_ext && d_out && batch >= 1, "extended_to_coeff: null pointer or empty batch");
    ZK_CHECK(ctx, (const void*)d_ext != (const void*)d_out, "extended_to_coeff: in and out must differ");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    fe *t0, *t1;
    int rc = scratch2(dom, batch, dom->extended_k, &t0, &t1);
    if (rc) return rc;
    NttFuse fuse;
    fuse.post_mode = 2;   // x extended_ifft_divisor x [1, zeta^2, zeta][i % 3]
    uint64_t c[4];
    memcpy(fuse.post[0].l, dom->extended_ifft_divisor, 32);
    h_mul(HFR, c, dom->extended_ifft_divisor, dom->g_coset_inv);
    memcpy(fuse.post[1].l, c, 32);
    h_mul(HFR, c, dom->extended_ifft_divisor, dom->g_coset);
    memcpy(fuse.post[2].l, c, 32);
    const size_t n = (size_t)1 << dom->k, en = (size_t)1 << dom->extended_k, no = n * dom->quotient_poly_degree;
    return ntt_run(ctx, dom->plan_ext_inv, (const fe*)d_ext, en, (uint32_t)en, (fe*)d_out, no, (uint32_t)no, batch, fuse,
                   t0, t1);
}

int zkhip_divide_by_vanishing_poly_dev(zkhip_domain* dom, uint64_t* d_ext, size_t batch) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_ext && batch >= 1, "divide_by_vanishing_poly: null pointer or empty batch");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t en = (size_t)1 << dom->extended_k;
    return scale_periodic(ctx, (fe*)d_ext, (const fe*)dom->d_t_evaluations, 1u << (dom->extended_k - dom->k), batch * en);
}

// ---- coset subsets ------------------------------------------------------------------------------------
int zkhip_coeff_to_cosets_dev(zkhip_domain* dom, const uint64_t* d_coeff, const uint32_t* cosets, uint32_t num_cosets,
                              uint64_t* d_out, size_t batch) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_coeff && d_out && batch >= 1, "coeff_to_cosets: null pointer or empty batch");
    CosetList cl;
    int rc = coset_list_check(dom, cosets, num_cosets, &cl);
    if (rc) return rc;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    rc = ensure_scratch(ctx, domain_coeff_to_cosets_tmp(dom, cl, batch) * 32);
    if (rc) return rc;
    return domain_coeff_to_cosets(dom, (const fe*)d_coeff, batch, cl, (fe*)d_out, (fe*)ctx->scratch);
}

static int coset_copy(zkhip_domain* dom, const uint64_t* d_src, const uint32_t* cosets, uint32_t num_cosets, uint64_t* d_dst,
                      int to_compact, const char* what) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, d_src && d_dst && (const void*)d_src != (const void*)d_dst, "%s: null or overlapping pointers", what);
    CosetList cl;
    int rc = coset_list_check(dom, cosets, num_cosets, &cl);
    if (rc) return rc;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rows = ((size_t)1 << dom->k) * cl.num;
    hipLaunchKernelGGL(coset_gather_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, (const fe*)d_src,
                       (fe*)d_dst, rows, cl, dom->extended_k - dom->k, to_compact);
    ZK_HIP(ctx, hipGetLastError());
    return ZKHIP_OK;
}
int zkhip_extended_to_cosets_dev(zkhip_domain* dom, const uint64_t* d_ext, const uint32_t* cosets, uint32_t num_cosets,
                                 uint64_t* d_out) {
    return coset_copy(dom, d_ext, cosets, num_cosets, d_out, 1, "extended_to_cosets");
}
int zkhip_cosets_to_extended_dev(zkhip_domain* dom, const uint64_t* d_compact, const uint32_t* cosets, uint32_t num_cosets,
                                 uint64_t* d_ext) {
    return coset_copy(dom, d_compact, cosets, num_cosets, d_ext, 0, "cosets_to_extended");
}

// ---- host variants: upload, run, download ---------------------------------------------------------
static int host_roundtrip(zkhip_domain* dom, const uint64_t* in, size_t n_in, uint64_t* out, size_t n_out, int which) {
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, in && out, "null pointer");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    void *d_in = nullptr, *d_out = nullptr;
    ZK_HIP(ctx, hipMalloc(&d_in, n_in * 32));
    hipError_t e = hipSuccess;
    const bool separate_out = (which == 2 || which == 3);
    if (separate_out) e = hipMalloc(&d_out, n_out * 32);
    int rc = ZKHIP_OK;
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, in, n_in * 32, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        switch (which) {
            case 0: rc = zkhip_lagrange_to_coeff_dev(dom, (uint64_t*)d_in, 1); break;
            case 1: rc = zkhip_coeff_to_lagrange_dev(dom, (uint64_t*)d_in, 1); break;
            case 2: rc = zkhip_coeff_to_extended_dev(dom, (const uint64_t*)d_in, (uint64_t*)d_out, 1); break;
            case 3: rc = zkhip_extended_to_coeff_dev(dom, (uint64_t*)d_in, (uint64_t*)d_out, 1); break;
            default: rc = zkhip_divide_by_vanishing_poly_dev(dom, (uint64_t*)d_in, 1); break;
        }
        if (rc == ZKHIP_OK)
            e = hipMemcpyAsync(out, separate_out ? d_out : d_in, n_out * 32, hipMemcpyDeviceToHost, ctx->stream);
    }
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) return set_error(ctx, ZKHIP_ERR_HIP, "domain op failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return set_error(ctx, ZKHIP_ERR_HIP, "domain op failed: %s", hipGetErrorString(e2));
    return ZKHIP_OK;
}

int zkhip_lagrange_to_coeff(zkhip_domain* dom, uint64_t* a) {
    if (!dom) return ZKHIP_ERR_INVALID;
    const size_t n = (size_t)1 << dom->k;
    return host_roundtrip(dom, a, n, a, n, 0);
}
int zkhip_coeff_to_lagrange(zkhip_domain* dom, uint64_t* a) {
    if (!dom) return ZKHIP_ERR_INVALID;
    const size_t n = (size_t)1 << dom->k;
    return host_roundtrip(dom, a, n, a, n, 1);
}
int zkhip_coeff_to_extended(zkhip_domain* dom, const uint64_t* coeff, uint64_t* out) {
    if (!dom) return ZKHIP_ERR_INVALID;
    return host_roundtrip(dom, coeff, (size_t)1 << dom->k, out, (size_t)1 << dom->extended_k, 2);
}
int zkhip_extended_to_coeff(zkhip_domain* dom, const uint64_t* ext, uint64_t* out) {
    if (!dom) return ZKHIP_ERR_INVALID;
    return host_roundtrip(dom, ext, (size_t)1 << dom->extended_k, out,
                          ((size_t)1 << dom->k) * dom->quotient_poly_degree, 3);
}
int zkhip_divide_by_vanishing_poly(zkhip_domain* dom, uint64_t* ext) {
    if (!dom) return ZKHIP_ERR_INVALID;
    const size_t en = (size_t)1 << dom->extended_k;
    return host_roundtrip(dom, ext, en, ext, en, 4);
}

}  // extern "C"
