// zkhip.hpp — header-only C++17 mirror, above the C ABI of zkhip.h, of the halo2_proofs interfaces the accelerator
// replaces.  Names, argument meaning and failure behaviour follow the crate the reference links
// (/root/reference/zkevm-circuits/Cargo.toml:15) as it is called from
// /root/reference/zkevm-circuits/benches/benchmark_list/main.rs:265-281 and /root/reference/prover/src/chunk/prover.rs:85:
//
//     halo2::arithmetic::best_multiexp(coeffs, bases)            arithmetic::best_multiexp
//     halo2::arithmetic::best_fft(a, omega, log_n)               arithmetic::best_fft
//     halo2::poly::EvaluationDomain                              poly::EvaluationDomain<Fr>::{new, k, extended_k, extended_len,
//                                                                get_omega, get_extended_omega, get_quotient_poly_degree,
//                                                                lagrange_to_coeff, coeff_to_lagrange, coeff_to_extended,
//                                                                extended_to_coeff, divide_by_vanishing_poly}
//     halo2::poly::ParamsKZG::{commit, commit_lagrange}          poly::kzg::commitment::ParamsKZG (the two MSM call sites)
//     halo2::plonk::Evaluator::evaluate_h                        plonk::evaluation::Evaluator::evaluate_h
//     halo2::arithmetic::{eval_polynomial, kate_division}        arithmetic::{eval_polynomial, kate_division}
//     halo2::arithmetic::batch_invert                            ff::BatchInvert::batch_invert
//     halo2::plonk::lookup::permute_expression_pair              plonk::lookup::prover::permute_expression_pair
//
// Upstream's functions panic on misuse (length mismatch, wrong domain); here every failure of the C ABI throws
// halo2::Panic carrying zkhip_last_error's text.  As in the reference (`env_info: &mut Option<JitProverEnv>`,
// /root/reference/prover/src/test/proof_test.rs:297-306) the accelerator is a process-wide option: Accelerator::get()
// creates the context for device 0 on first use and throws when there is no gfx950 device — there is no CPU path here.
#ifndef ZKHIP_HPP
#define ZKHIP_HPP

#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "zkhip.h"

namespace halo2 {

struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// halo2curves::bn256::{Fr, Fq}: four little-endian limbs in Montgomery form
struct Fr {
    uint64_t l[4];
    bool operator==(const Fr& o) const { return std::memcmp(l, o.l, 32) == 0; }
    bool operator!=(const Fr& o) const { return !(*this == o); }
};
using Fq = Fr;
struct G1Affine { Fq x, y; };      // identity: all zero
struct G1 { Fq x, y, z; };         // Jacobian, identity: z = 0
static_assert(sizeof(Fr) == 32 && sizeof(G1Affine) == 64 && sizeof(G1) == 96, "limb layout of the C ABI");

class Accelerator {
public:
    // Lives until the process ends and is never destroyed: a destructor running among the static destructors would
    // call into a HIP runtime that may already have shut down.
    static Accelerator& get() {
        static Accelerator* a = new Accelerator(0);
        return *a;
    }
    zkhip_ctx* ctx() const { return ctx_; }
    std::mutex& lock() { return mu_; }     // a context is not re-entrant
    void check(int rc, const char* what) const {
        if (rc != ZKHIP_OK) throw Panic(std::string(what) + ": " + zkhip_last_error(ctx_));
    }
    // SRS tables are registered once and found again by address and length, as `commit` passes `&self.g[..len]`
    uint32_t bases_id(const G1Affine* bases, size_t n) {
        auto it = bases_.lower_bound(reinterpret_cast<uintptr_t>(bases));
        if (it != bases_.end() && it->first == reinterpret_cast<uintptr_t>(bases) && it->second.first >= n) return it->second.second;
        uint32_t id;
        check(zkhip_register_bases(ctx_, reinterpret_cast<const uint64_t*>(bases), n, &id), "register_bases");
        if (it != bases_.end() && it->first == reinterpret_cast<uintptr_t>(bases)) {
            zkhip_release_bases(ctx_, it->second.second);      // the same table, registered shorter before
            it->second = {n, id};
        } else {
            bases_.emplace(reinterpret_cast<uintptr_t>(bases), std::make_pair(n, id));
        }
        return id;
    }
    void forget_bases(const G1Affine* bases) {
        auto it = bases_.find(reinterpret_cast<uintptr_t>(bases));
        if (it == bases_.end()) return;
        zkhip_release_bases(ctx_, it->second.second);
        bases_.erase(it);
    }
    Accelerator(const Accelerator&) = delete;
    Accelerator& operator=(const Accelerator&) = delete;

private:
    explicit Accelerator(int device) {
        const int rc = zkhip_create(device, &ctx_);
        if (rc != ZKHIP_OK) throw Panic("zkhip_create failed (no usable gfx950 device?): status " + std::to_string(rc));
    }
    ~Accelerator() { zkhip_destroy(ctx_); }
    zkhip_ctx* ctx_ = nullptr;
    std::mutex mu_;
    std::map<uintptr_t, std::pair<size_t, uint32_t>> bases_;
};

namespace arithmetic {

// sum_i coeffs[i] * bases[i]; panics when the lengths differ (upstream: assert_eq!(coeffs.len(), bases.len()))
inline G1 best_multiexp(const std::vector<Fr>& coeffs, const G1Affine* bases, size_t bases_len) {
    if (coeffs.size() != bases_len) throw Panic("best_multiexp: coeffs and bases differ in length");
    Accelerator& acc = Accelerator::get();
    std::lock_guard<std::mutex> g(acc.lock());
    G1 out;
    const uint32_t id = acc.bases_id(bases, bases_len);
    acc.check(zkhip_msm(acc.ctx(), id, reinterpret_cast<const uint64_t*>(coeffs.data()), coeffs.size(),
                        reinterpret_cast<uint64_t*>(&out)), "best_multiexp");
    return out;
}
inline G1 best_multiexp(const std::vector<Fr>& coeffs, const std::vector<G1Affine>& bases) {
    return best_multiexp(coeffs, bases.data(), bases.size());
}

// in place, natural order: a[i] <- sum_j a[j] omega^(i j); panics unless a.len() == 1 << log_n
inline void best_fft(std::vector<Fr>& a, const Fr& omega, uint32_t log_n) {
    if (log_n >= 64 || a.size() != (size_t(1) << log_n)) throw Panic("best_fft: a.len() != 1 << log_n");
    Accelerator& acc = Accelerator::get();
    std::lock_guard<std::mutex> g(acc.lock());
    acc.check(zkhip_ntt(acc.ctx(), reinterpret_cast<uint64_t*>(a.data()), log_n, omega.l), "best_fft");
}

// A device buffer of field elements for the duration of one call (the context's allocator; freed on scope exit)
class DeviceVec {
public:
    DeviceVec(Accelerator& acc, size_t elements) : acc_(acc) {
        acc.check(zkhip_dev_alloc(acc.ctx(), (elements ? elements : 1) * 32, &p_), "device allocation");
    }
    DeviceVec(Accelerator& acc, const std::vector<Fr>& host) : DeviceVec(acc, host.size()) {
        if (!host.empty()) acc.check(zkhip_upload(acc.ctx(), p_, host.data(), host.size() * 32), "upload");
    }
    ~DeviceVec() { zkhip_dev_free(acc_.ctx(), p_); }
    DeviceVec(const DeviceVec&) = delete;
    DeviceVec& operator=(const DeviceVec&) = delete;
    uint64_t* ptr() const { return static_cast<uint64_t*>(p_); }
    std::vector<Fr> download(size_t elements) const {
        std::vector<Fr> out(elements);
        if (elements) acc_.check(zkhip_download(acc_.ctx(), out.data(), p_, elements * 32), "download");
        return out;
    }

private:
    Accelerator& acc_;
    void* p_ = nullptr;
};

// sum_i poly[i] point^i
inline Fr eval_polynomial(const std::vector<Fr>& poly, const Fr& point) {
    Accelerator& acc = Accelerator::get();
    std::lock_guard<std::mutex> g(acc.lock());
    Fr out{{0, 0, 0, 0}};
    if (poly.empty()) return out;
    const DeviceVec d(acc, poly);
    acc.check(zkhip_eval_polynomial_dev(acc.ctx(), d.ptr(), poly.size(), point.l, out.l), "eval_polynomial");
    return out;
}

// quotient of a(X) by (X - b): a.len() - 1 coefficients, the remainder a(b) is dropped
inline std::vector<Fr> kate_division(const std::vector<Fr>& a, const Fr& b) {
    if (a.empty()) throw Panic("kate_division: empty polynomial");
    Accelerator& acc = Accelerator::get();
    std::lock_guard<std::mutex> g(acc.lock());
    const DeviceVec d(acc, a), q(acc, a.size());
    acc.check(zkhip_kate_division_dev(acc.ctx(), d.ptr(), a.size(), b.l, q.ptr()), "kate_division");
    acc.check(zkhip_sync(acc.ctx()), "kate_division");
    return q.download(a.size() - 1);
}

// every non-zero element replaced by its inverse, zeros kept
inline void batch_invert(std::vector<Fr>& v) {
    if (v.empty()) return;
    Accelerator& acc = Accelerator::get();
    std::lock_guard<std::mutex> g(acc.lock());
    const DeviceVec d(acc, v);
    acc.check(zkhip_batch_invert_dev(acc.ctx(), 0, d.ptr(), v.size()), "batch_invert");
    acc.check(zkhip_sync(acc.ctx()), "batch_invert");
    v = d.download(v.size());
}

inline G1Affine to_affine(const G1& p) {
    G1Affine out;
    if (zkhip_g1_to_affine(reinterpret_cast<const uint64_t*>(&p), reinterpret_cast<uint64_t*>(&out)) != ZKHIP_OK)
        throw Panic("to_affine");
    return out;
}

}  // namespace arithmetic

namespace poly {

// Polynomial<Fr, B>: the basis is a matter of the caller's bookkeeping, as the marker type is upstream
struct Polynomial {
    std::vector<Fr> values;
};

class EvaluationDomain {
public:
    // EvaluationDomain::new(j, k): j = constraint system degree, 2^k rows
    EvaluationDomain(uint32_t j, uint32_t k) {
        Accelerator& acc = Accelerator::get();
        std::lock_guard<std::mutex> g(acc.lock());
        acc.check(zkhip_domain_create(acc.ctx(), j, k, &dom_), "EvaluationDomain::new");
    }
    ~EvaluationDomain() {
        if (dom_) zkhip_domain_destroy(dom_);
    }
    EvaluationDomain(const EvaluationDomain&) = delete;
    EvaluationDomain& operator=(const EvaluationDomain&) = delete;

    uint32_t k() const { return zkhip_domain_k(dom_); }
    uint32_t extended_k() const { return zkhip_domain_extended_k(dom_); }
    size_t extended_len() const { return size_t(1) << extended_k(); }
    uint64_t get_quotient_poly_degree() const { return zkhip_domain_quotient_poly_degree(dom_); }
    Fr get_omega() const { return constant(0); }
    Fr get_omega_inv() const { return constant(1); }
    Fr get_extended_omega() const { return constant(2); }

    Polynomial lagrange_to_coeff(Polynomial a) const {
        expect(a.values.size() == (size_t(1) << k()), "lagrange_to_coeff: a.len() != 1 << k");
        const auto g = guard();
        run(zkhip_lagrange_to_coeff(dom_, u64(a)), "lagrange_to_coeff");
        return a;
    }
    Polynomial coeff_to_lagrange(Polynomial a) const {
        expect(a.values.size() == (size_t(1) << k()), "coeff_to_lagrange: a.len() != 1 << k");
        const auto g = guard();
        run(zkhip_coeff_to_lagrange(dom_, u64(a)), "coeff_to_lagrange");
        return a;
    }
    Polynomial coeff_to_extended(const Polynomial& a) const {
        expect(a.values.size() == (size_t(1) << k()), "coeff_to_extended: a.len() != 1 << k");
        Polynomial out{std::vector<Fr>(extended_len())};
        const auto g = guard();
        run(zkhip_coeff_to_extended(dom_, u64(a), u64(out)), "coeff_to_extended");
        return out;
    }
    // the result keeps n * quotient_poly_degree coefficients, as upstream truncates it
    Polynomial extended_to_coeff(const Polynomial& a) const {
        expect(a.values.size() == extended_len(), "extended_to_coeff: a.len() != extended_len()");
        Polynomial out{std::vector<Fr>((size_t(1) << k()) * get_quotient_poly_degree())};
        const auto g = guard();
        run(zkhip_extended_to_coeff(dom_, u64(a), u64(out)), "extended_to_coeff");
        return out;
    }
    Polynomial divide_by_vanishing_poly(Polynomial a) const {
        expect(a.values.size() == extended_len(), "divide_by_vanishing_poly: a.len() != extended_len()");
        const auto g = guard();
        run(zkhip_divide_by_vanishing_poly(dom_, u64(a)), "divide_by_vanishing_poly");
        return a;
    }
    zkhip_domain* raw() const { return dom_; }

private:
    static uint64_t* u64(Polynomial& p) { return reinterpret_cast<uint64_t*>(p.values.data()); }
    static const uint64_t* u64(const Polynomial& p) { return reinterpret_cast<const uint64_t*>(p.values.data()); }
    static void expect(bool ok, const char* what) {
        if (!ok) throw Panic(what);
    }
    void run(int rc, const char* what) const { Accelerator::get().check(rc, what); }
    static std::unique_lock<std::mutex> guard() { return std::unique_lock<std::mutex>(Accelerator::get().lock()); }
    Fr constant(int which) const {
        Fr out;
        run(zkhip_domain_constant(dom_, which, out.l), "domain constant");
        return out;
    }
    zkhip_domain* dom_ = nullptr;
};

// The two commitment entry points of ParamsKZG; g and g_lagrange stay owned by the caller (the parameter file's tables)
class ParamsKZG {
public:
    ParamsKZG(uint32_t k, const G1Affine* g, const G1Affine* g_lagrange) : k_(k), g_(g), g_lagrange_(g_lagrange) {}
    ~ParamsKZG() {
        Accelerator& acc = Accelerator::get();
        std::lock_guard<std::mutex> lk(acc.lock());
        acc.forget_bases(g_);
        acc.forget_bases(g_lagrange_);
    }
    uint32_t k() const { return k_; }
    size_t n() const { return size_t(1) << k_; }
    // commit(poly: Polynomial<Fr, Coeff>): the first poly.len() bases of g
    G1 commit(const Polynomial& p) const { return msm(p, g_); }
    // commit_lagrange(poly: Polynomial<Fr, LagrangeCoeff>): against g_lagrange
    G1 commit_lagrange(const Polynomial& p) const { return msm(p, g_lagrange_); }

private:
    G1 msm(const Polynomial& p, const G1Affine* table) const {
        if (p.values.size() > n()) throw Panic("commit: polynomial longer than the parameters");
        Accelerator& acc = Accelerator::get();
        std::lock_guard<std::mutex> lk(acc.lock());
        const uint32_t id = acc.bases_id(table, n());      // whole table once; an MSM uses its first poly.len() bases
        G1 out;
        acc.check(zkhip_msm(acc.ctx(), id, reinterpret_cast<const uint64_t*>(p.values.data()), p.values.size(),
                            reinterpret_cast<uint64_t*>(&out)), "commit");
        return out;
    }
    uint32_t k_;
    const G1Affine *g_, *g_lagrange_;
};

}  // namespace poly

namespace plonk {

// GraphEvaluator { constants, rotations, calculations }, the calculations in the word stream zkhip.h documents
struct GraphEvaluator {
    std::vector<Fr> constants;
    std::vector<int32_t> rotations;
    std::vector<uint32_t> calc_words;
    uint32_t num_calculations = 0;
    zkhip_graph view() const {
        zkhip_graph g;
        g.constants = reinterpret_cast<const uint64_t*>(constants.data());
        g.num_constants = (uint32_t)constants.size();
        g.rotations = rotations.data();
        g.num_rotations = (uint32_t)rotations.size();
        g.calc_words = calc_words.data();
        g.num_calc_words = (uint32_t)calc_words.size();
        g.num_calculations = num_calculations;
        return g;
    }
};

enum class Any : uint32_t { Fixed = 2, Advice = 3, Instance = 4 };

// the parts of ProvingKey / ConstraintSystem evaluate_h reads
struct ProvingKey {
    const poly::EvaluationDomain* domain = nullptr;              // vk.domain
    std::vector<poly::Polynomial> fixed_cosets;                  // extended
    poly::Polynomial l0, l_last, l_active_row;                   // extended
    std::vector<poly::Polynomial> permutation_cosets;            // pk.permutation.cosets, extended
    std::vector<std::pair<Any, uint32_t>> permutation_columns;   // cs.permutation.columns
    uint32_t cs_degree = 0;                                      // cs.degree()
    uint32_t blinding_factors = 0;                               // cs.blinding_factors()
};
// lookup::prover::Committed and permutation::prover::Committed of one circuit instance
struct LookupCommitted {
    poly::Polynomial product_poly, permuted_input_poly, permuted_table_poly;   // coefficient form
};
struct PermutationCommitted {
    std::vector<poly::Polynomial> permutation_product_cosets;                  // extended, one per column set
};

class Evaluator {
public:
    GraphEvaluator custom_gates;
    std::vector<GraphEvaluator> lookups;     // one graph per lookup argument

    // Numerator of the quotient polynomial over the extended domain, for one circuit instance; advice / instance /
    // lookup polynomials in coefficient form, as upstream passes them.
    poly::Polynomial evaluate_h(const ProvingKey& pk, const std::vector<poly::Polynomial>& advice_polys,
                                const std::vector<poly::Polynomial>& instance_polys, const std::vector<Fr>& challenges,
                                const Fr& y, const Fr& beta, const Fr& gamma, const Fr& theta,
                                const std::vector<LookupCommitted>& lookup_committed,
                                const PermutationCommitted& permutation) const;
};

// The C descriptor of one evaluate_h call together with the pointer tables it refers to.  `poly_form` says in which form
// the advice / instance / lookup polynomials are given (ZKHIP_FORM_COEFF is upstream's calling convention).
struct EvaluateHCall {
    zkhip_evalh_desc desc;
    EvaluateHCall(const Evaluator& ev, const ProvingKey& pk, const std::vector<poly::Polynomial>& advice_polys,
                  const std::vector<poly::Polynomial>& instance_polys, const std::vector<Fr>& challenges, const Fr& y,
                  const Fr& beta, const Fr& gamma, const Fr& theta, const std::vector<LookupCommitted>& lookup_committed,
                  const PermutationCommitted& permutation, uint32_t poly_form = ZKHIP_FORM_COEFF) {
        if (!pk.domain) throw Panic("evaluate_h: proving key without a domain");
        if (lookup_committed.size() != ev.lookups.size()) throw Panic("evaluate_h: lookups and their graphs differ in number");
        const size_t n = size_t(1) << pk.domain->k(), en = pk.domain->extended_len();
        const size_t plen = poly_form == ZKHIP_FORM_COEFF ? n : en;
        auto ptrs = [](const std::vector<poly::Polynomial>& v, size_t len, const char* what) {
            std::vector<const uint64_t*> p;
            for (const auto& q : v) {
                if (q.values.size() != len) throw Panic(std::string("evaluate_h: wrong length of ") + what);
                p.push_back(reinterpret_cast<const uint64_t*>(q.values.data()));
            }
            return p;
        };
        fixed_ = ptrs(pk.fixed_cosets, en, "a fixed coset");
        advice_ = ptrs(advice_polys, plen, "an advice polynomial");
        instance_ = ptrs(instance_polys, plen, "an instance polynomial");
        sigma_ = ptrs(pk.permutation_cosets, en, "a permutation coset");
        prod_ = ptrs(permutation.permutation_product_cosets, en, "a permutation product coset");
        for (const poly::Polynomial* p : {&pk.l0, &pk.l_last, &pk.l_active_row})
            if (p->values.size() != en) throw Panic("evaluate_h: wrong length of l0 / l_last / l_active_row");
        lk_.resize(ev.lookups.size());
        for (size_t i = 0; i < lk_.size(); i++) {
            const LookupCommitted& c = lookup_committed[i];
            if (c.product_poly.values.size() != plen || c.permuted_input_poly.values.size() != plen ||
                c.permuted_table_poly.values.size() != plen)
                throw Panic("evaluate_h: wrong length of a lookup polynomial");
            lk_[i].graph = ev.lookups[i].view();
            lk_[i].product = reinterpret_cast<const uint64_t*>(c.product_poly.values.data());
            lk_[i].permuted_input = reinterpret_cast<const uint64_t*>(c.permuted_input_poly.values.data());
            lk_[i].permuted_table = reinterpret_cast<const uint64_t*>(c.permuted_table_poly.values.data());
        }
        for (const auto& c : pk.permutation_columns) {
            kind_.push_back((uint32_t)c.first);
            index_.push_back(c.second);
        }
        if (sigma_.size() != kind_.size()) throw Panic("evaluate_h: permutation cosets and columns differ in number");
        gates_ = ev.custom_gates.view();
        zkhip_evalh_desc& d = desc;
        std::memset(&d, 0, sizeof d);
        d.mem = ZKHIP_MEM_HOST;
        d.poly_form = poly_form;
        d.num_fixed = (uint32_t)fixed_.size();
        d.num_advice = (uint32_t)advice_.size();
        d.num_instance = (uint32_t)instance_.size();
        d.num_challenges = (uint32_t)challenges.size();
        d.fixed_cosets = fixed_.data();
        d.advice = advice_.data();
        d.instance = instance_.data();
        d.challenges = reinterpret_cast<const uint64_t*>(challenges.data());
        std::memcpy(d.y, y.l, 32);
        std::memcpy(d.beta, beta.l, 32);
        std::memcpy(d.gamma, gamma.l, 32);
        std::memcpy(d.theta, theta.l, 32);
        d.l0 = reinterpret_cast<const uint64_t*>(pk.l0.values.data());
        d.l_last = reinterpret_cast<const uint64_t*>(pk.l_last.values.data());
        d.l_active_row = reinterpret_cast<const uint64_t*>(pk.l_active_row.values.data());
        d.custom_gates = &gates_;
        d.num_lookups = (uint32_t)lk_.size();
        d.lookups = lk_.data();
        d.num_perm_columns = (uint32_t)kind_.size();
        d.perm_column_kind = kind_.data();
        d.perm_column_index = index_.data();
        d.perm_sigma_cosets = sigma_.data();
        d.num_perm_sets = (uint32_t)prod_.size();
        d.perm_product_cosets = prod_.data();
        d.perm_chunk_len = pk.cs_degree >= 3 ? pk.cs_degree - 2 : 1;
        d.blinding_factors = pk.blinding_factors;
    }
    EvaluateHCall(const EvaluateHCall&) = delete;
    EvaluateHCall& operator=(const EvaluateHCall&) = delete;

private:
    std::vector<const uint64_t*> fixed_, advice_, instance_, sigma_, prod_;
    std::vector<zkhip_lookup> lk_;
    std::vector<uint32_t> kind_, index_;
    zkhip_graph gates_;
};

inline poly::Polynomial Evaluator::evaluate_h(const ProvingKey& pk, const std::vector<poly::Polynomial>& advice_polys,
                                              const std::vector<poly::Polynomial>& instance_polys,
                                              const std::vector<Fr>& challenges, const Fr& y, const Fr& beta, const Fr& gamma,
                                              const Fr& theta, const std::vector<LookupCommitted>& lookup_committed,
                                              const PermutationCommitted& permutation) const {
    const EvaluateHCall call(*this, pk, advice_polys, instance_polys, challenges, y, beta, gamma, theta, lookup_committed,
                             permutation);
    poly::Polynomial out{std::vector<Fr>(pk.domain->extended_len())};
    Accelerator& acc = Accelerator::get();
    std::lock_guard<std::mutex> g(acc.lock());
    acc.check(zkhip_evaluate_h(pk.domain->raw(), &call.desc, reinterpret_cast<uint64_t*>(out.values.data())), "evaluate_h");
    return out;
}

namespace lookup {

// Rows [0, usable_rows) of the permuted input A' (the compressed input sorted) and of the permuted table S' (S'[i] =
// A'[i] where A'[i] first occurs, leftover table values in the repeated rows); usable_rows = n - (blinding_factors + 1).
// The blinding rows beyond are drawn from the prover's RNG by the caller and are returned as zero here.  An input value
// missing from the table is upstream's Error::ConstraintSystemFailure: thrown as ConstraintSystemFailure.
struct ConstraintSystemFailure : Panic {
    using Panic::Panic;
};
inline std::pair<poly::Polynomial, poly::Polynomial> permute_expression_pair(const poly::EvaluationDomain& domain,
                                                                             uint32_t blinding_factors,
                                                                             const poly::Polynomial& input_expression,
                                                                             const poly::Polynomial& table_expression) {
    const size_t n = size_t(1) << domain.k();
    if (input_expression.values.size() != n || table_expression.values.size() != n)
        throw Panic("permute_expression_pair: expressions must have 2^k rows");
    if (blinding_factors + 1 >= n) throw Panic("permute_expression_pair: no usable rows");
    const uint32_t usable_rows = (uint32_t)(n - (blinding_factors + 1));
    Accelerator& acc = Accelerator::get();
    std::lock_guard<std::mutex> g(acc.lock());
    const std::vector<Fr> zeros(n, Fr{{0, 0, 0, 0}});
    const arithmetic::DeviceVec in(acc, input_expression.values), tab(acc, table_expression.values), pin(acc, zeros), ptab(acc, zeros);
    const int rc = zkhip_lookup_permute_dev(domain.raw(), in.ptr(), tab.ptr(), usable_rows, pin.ptr(), ptab.ptr());
    if (rc == ZKHIP_ERR_CONSTRAINT) throw ConstraintSystemFailure(zkhip_last_error(acc.ctx()));
    acc.check(rc, "permute_expression_pair");
    return {poly::Polynomial{pin.download(n)}, poly::Polynomial{ptab.download(n)}};
}

}  // namespace lookup

}  // namespace plonk
}  // namespace halo2

#endif  // ZKHIP_HPP
