// EvaluationDomain state (internal to libzkhip).
#pragma once
#include <map>
#include "common.h"
#include "ntt.h"

struct zkhip_domain {
    zkhip_ctx* ctx = nullptr;
    uint32_t k = 0, extended_k = 0, quotient_poly_degree = 0;
    uint64_t omega[4], omega_inv[4], extended_omega[4], extended_omega_inv[4];
    uint64_t g_coset[4], g_coset_inv[4], ifft_divisor[4], extended_ifft_divisor[4];
    std::vector<uint64_t> t_evaluations;   // 2^(extended_k - k) inverted values, Montgomery limbs
    void* d_t_evaluations = nullptr;
    void* d_xpoints = nullptr;             // zeta * extended_omega^idx per extended row (built on first use by evaluate_h)
    std::map<uint32_t, void*> d_coset_powers;   // coset j -> (zeta * extended_omega^j)^i for i < 2^k (built on first use)
    zkhip::NttPlan *plan_fwd = nullptr, *plan_inv = nullptr, *plan_ext_fwd = nullptr, *plan_ext_inv = nullptr;
    // transforms over the even-indexed half of the extended domain, zeta * <extended_omega^2> (built on first use)
    zkhip::NttPlan *plan_half_fwd = nullptr, *plan_half_inv = nullptr;
};

namespace zkhip {
// A subset of the 2^(extended_k - k) cosets zeta * extended_omega^j * <omega> that make up the extended domain
// (extended row idx lies in coset idx mod 2^e).  "Compact" vectors hold only these cosets, interleaved the same way:
// compact row q * num + s is extended row q * 2^e + j[s].  num is a power of two.
struct CosetList {
    uint32_t num;
    uint32_t j[16];
};
int coset_list_check(zkhip_domain* dom, const uint32_t* cosets, uint32_t num, CosetList* out);
// `batch` coefficient vectors (2^k each, back to back, device) -> their values on the listed cosets, compact layout
// (2^k * num per vector, back to back); tmp: domain_coeff_to_cosets_tmp(...) elements
size_t domain_coeff_to_cosets_tmp(const zkhip_domain* dom, const CosetList& cl, size_t batch);
int domain_coeff_to_cosets(zkhip_domain* dom, const fe* d_coeff, size_t batch, const CosetList& cl, fe* d_out, fe* tmp);
// x[q * num + s] = zeta * extended_omega^(q * 2^e + j[s])
int domain_coset_xpoints(zkhip_domain* dom, const CosetList& cl, fe* d_x);
// coeff (2^k per member, stride 2^k) -> extended coset (2^extended_k per member); tmp0/tmp1: batch * 2^extended_k each
int domain_coeff_to_extended(zkhip_domain* dom, const fe* d_coeff, fe* d_out, size_t batch, fe* tmp0, fe* tmp1,
                             bool split = false);
}  // namespace zkhip
