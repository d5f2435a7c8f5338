// Parity of the C++ mirror (include/zkhip.hpp) against the CPU oracle, on the GPU: reads like the reference's own
// prover tests would if they compared results -- same calls, same argument order -- with the oracle as the checker.
// Built by __graft_entry__.build(); run by tests/test_cpp_mirror.py on a machine with a gfx950 device.
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../include/zkhip.hpp"

extern "C" {
typedef struct odomain odomain;
void oracle_field_from_canonical(int field, const uint64_t* a, uint64_t* out, size_t n);
void oracle_field_binop(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
void oracle_g1_to_affine(const uint64_t* jac, uint64_t* out_xy);
void oracle_g1_bases(uint64_t* out_xy, size_t n);
void oracle_best_multiexp(const uint64_t* coeffs, const uint64_t* bases, size_t n, int threads, uint64_t* out_jac);
void oracle_best_fft(uint64_t* a, const uint64_t* omega, uint32_t log_n, int threads);
odomain* oracle_domain_new(uint32_t j, uint32_t k, int threads);
void oracle_domain_free(odomain* d);
uint32_t oracle_domain_extended_k(const odomain* d);
void oracle_domain_constant(const odomain* d, int which, uint64_t* out);
void oracle_lagrange_to_coeff(const odomain* d, uint64_t* a);
void oracle_coeff_to_lagrange(const odomain* d, uint64_t* a);
void oracle_coeff_to_extended(const odomain* d, const uint64_t* coeff, uint64_t* out);
void oracle_extended_to_coeff(const odomain* d, const uint64_t* ext, uint64_t* out);
void oracle_divide_by_vanishing_poly(const odomain* d, uint64_t* ext);
int oracle_evaluate_h(const odomain* d, const zkhip_evalh_desc* desc, uint64_t* out);
}

using namespace halo2;
using poly::Polynomial;

static int failures = 0;
#define CHECK(cond, name)                                  \
    do {                                                   \
        if (cond) std::printf("ok    %s\n", name);         \
        else { std::printf("FAIL  %s\n", name); failures++; } \
    } while (0)

static std::mt19937_64 rng(20250523);
static Fr rand_fr() {                       // any value below 2^252 is a reduced residue
    Fr f;
    for (auto& w : f.l) w = rng();
    f.l[3] &= 0x0fffffffffffffffull;
    return f;
}
static Polynomial rand_poly(size_t n) {
    Polynomial p;
    p.values.resize(n);
    for (auto& v : p.values) v = rand_fr();
    return p;
}
static uint64_t* u64(Polynomial& p) { return reinterpret_cast<uint64_t*>(p.values.data()); }
static const uint64_t* u64(const Polynomial& p) { return reinterpret_cast<const uint64_t*>(p.values.data()); }
static Fr mont(uint64_t v) {
    const uint64_t c[4] = {v, 0, 0, 0};
    Fr out;
    oracle_field_from_canonical(0, c, out.l, 1);
    return out;
}
template <class F>
static bool panics(F f) {
    try { f(); } catch (const Panic&) { return true; }
    return false;
}

static void test_best_multiexp() {
    const size_t n = 3001;
    std::vector<G1Affine> bases(n);
    oracle_g1_bases(reinterpret_cast<uint64_t*>(bases.data()), n);
    std::vector<Fr> coeffs(n);
    for (auto& c : coeffs) c = rand_fr();
    coeffs[7] = Fr{{0, 0, 0, 0}};
    const G1 got = arithmetic::best_multiexp(coeffs, bases);
    uint64_t exp_jac[12], exp_xy[8];
    oracle_best_multiexp(reinterpret_cast<const uint64_t*>(coeffs.data()), reinterpret_cast<const uint64_t*>(bases.data()), n, 8, exp_jac);
    oracle_g1_to_affine(exp_jac, exp_xy);
    const G1Affine a = arithmetic::to_affine(got);
    CHECK(std::memcmp(&a, exp_xy, 64) == 0, "best_multiexp(coeffs, bases) == oracle");
    // ParamsKZG::commit uses the first poly.len() bases of the registered table
    poly::ParamsKZG params(11, bases.data(), bases.data());      // k = 11: 2048 of the points
    Polynomial p = rand_poly(1500);
    const G1Affine c = arithmetic::to_affine(params.commit(p));
    oracle_best_multiexp(u64(p), reinterpret_cast<const uint64_t*>(bases.data()), 1500, 8, exp_jac);
    oracle_g1_to_affine(exp_jac, exp_xy);
    CHECK(std::memcmp(&c, exp_xy, 64) == 0, "ParamsKZG::commit(poly) == oracle over the first poly.len() bases");
    CHECK(panics([&] { coeffs.pop_back(); arithmetic::best_multiexp(coeffs, bases); }), "best_multiexp panics on a length mismatch");
    CHECK(panics([&] { params.commit(rand_poly(4096)); }), "commit panics on a polynomial longer than the parameters");
}

static void test_best_fft_and_domain() {
    const uint32_t j = 5, k = 10;
    poly::EvaluationDomain domain(j, k);
    odomain* od = oracle_domain_new(j, k, 8);
    CHECK(domain.extended_k() == oracle_domain_extended_k(od) && domain.k() == k && domain.get_quotient_poly_degree() == j - 1,
          "EvaluationDomain::new(j, k): k, extended_k, quotient_poly_degree");
    Fr om;
    oracle_domain_constant(od, 0, om.l);
    CHECK(domain.get_omega() == om, "get_omega");
    oracle_domain_constant(od, 2, om.l);
    CHECK(domain.get_extended_omega() == om, "get_extended_omega");

    Polynomial a = rand_poly(size_t(1) << k), b = a;
    arithmetic::best_fft(a.values, domain.get_omega(), k);
    const Fr w = domain.get_omega();
    oracle_best_fft(u64(b), w.l, k, 8);
    CHECK(a.values == b.values, "best_fft(a, omega, log_n) == oracle");
    CHECK(panics([&] { arithmetic::best_fft(a.values, w, k + 1); }), "best_fft panics when a.len() != 1 << log_n");
    CHECK(panics([&] { arithmetic::best_fft(a.values, domain.get_extended_omega(), k); }), "best_fft panics on a root of the wrong order");

    Polynomial lag = rand_poly(size_t(1) << k), exp = lag;
    oracle_lagrange_to_coeff(od, u64(exp));
    const Polynomial coeff = domain.lagrange_to_coeff(lag);
    CHECK(coeff.values == exp.values, "lagrange_to_coeff");
    CHECK(domain.coeff_to_lagrange(coeff).values == lag.values, "coeff_to_lagrange inverts it");
    Polynomial ext_exp{std::vector<Fr>(domain.extended_len())};
    oracle_coeff_to_extended(od, u64(coeff), u64(ext_exp));
    const Polynomial ext = domain.coeff_to_extended(coeff);
    CHECK(ext.values == ext_exp.values, "coeff_to_extended");
    Polynomial div_exp = ext_exp;
    oracle_divide_by_vanishing_poly(od, u64(div_exp));
    const Polynomial div = domain.divide_by_vanishing_poly(ext);
    CHECK(div.values == div_exp.values, "divide_by_vanishing_poly");
    Polynomial back_exp{std::vector<Fr>((size_t(1) << k) * (j - 1))};
    oracle_extended_to_coeff(od, u64(div_exp), u64(back_exp));
    CHECK(domain.extended_to_coeff(div).values == back_exp.values, "extended_to_coeff");
    CHECK(panics([&] { domain.coeff_to_extended(ext); }), "coeff_to_extended panics on a polynomial of the wrong length");
    oracle_domain_free(od);
}

// value sources and operations of the serialised GraphEvaluator (zkhip.h)
enum : uint32_t { CONST = 0, INTER, FIXED, ADVICE, INSTANCE, CHALLENGE, BETA, GAMMA, THETA, Y, PREV };
enum : uint32_t { ADD = 0, SUB, MUL, SQUARE, DOUBLE, NEGATE, HORNER, STORE };
static void push(plonk::GraphEvaluator& g, std::initializer_list<uint32_t> words) {
    g.calc_words.insert(g.calc_words.end(), words);
    g.num_calculations++;
}

static void test_evaluate_h() {
    const uint32_t degree = 5, k = 8;
    poly::EvaluationDomain domain(degree, k);
    odomain* od = oracle_domain_new(degree, k, 8);
    const size_t n = size_t(1) << k, en = domain.extended_len();

    plonk::Evaluator ev;
    // gates: f1 * (a0 * a1(w X) - f0)  and  (c * a1(w^-1 X) + instance0)^2 - challenge0, folded by y
    plonk::GraphEvaluator& g = ev.custom_gates;
    g.constants = {mont(0), mont(1), mont(2), rand_fr()};
    g.rotations = {0, 1, -1};
    push(g, {MUL, ADVICE, 0, 0, ADVICE, 1, 1});
    push(g, {SUB, INTER, 0, 0, FIXED, 0, 0});
    push(g, {MUL, FIXED, 1, 0, INTER, 1, 0});
    push(g, {MUL, ADVICE, 1, 2, CONST, 3, 0});
    push(g, {ADD, INTER, 3, 0, INSTANCE, 0, 0});
    push(g, {SQUARE, INTER, 4, 0});
    push(g, {SUB, INTER, 5, 0, CHALLENGE, 0, 0});
    push(g, {DOUBLE, INTER, 6, 0});
    push(g, {NEGATE, INTER, 7, 0});
    push(g, {HORNER, PREV, 0, 0, Y, 0, 0, 3, INTER, 2, 0, INTER, 6, 0, INTER, 8, 0});
    // one lookup: inputs (a0, a1(w X)) into table (f0, a1), compressed by theta
    ev.lookups.emplace_back();
    plonk::GraphEvaluator& l = ev.lookups.back();
    l.constants = {mont(0), mont(1), mont(2)};
    l.rotations = {0, 1};
    push(l, {HORNER, CONST, 0, 0, THETA, 0, 0, 2, ADVICE, 0, 0, ADVICE, 1, 1});
    push(l, {HORNER, CONST, 0, 0, THETA, 0, 0, 2, FIXED, 0, 0, ADVICE, 1, 0});
    push(l, {ADD, INTER, 0, 0, BETA, 0, 0});
    push(l, {ADD, INTER, 1, 0, GAMMA, 0, 0});
    push(l, {MUL, INTER, 2, 0, INTER, 3, 0});

    plonk::ProvingKey pk;
    pk.domain = &domain;
    pk.fixed_cosets = {rand_poly(en), rand_poly(en)};
    pk.l0 = rand_poly(en);
    pk.l_last = rand_poly(en);
    pk.l_active_row = rand_poly(en);
    pk.permutation_columns = {{plonk::Any::Advice, 0}, {plonk::Any::Fixed, 0}, {plonk::Any::Advice, 1}, {plonk::Any::Instance, 0}};
    pk.permutation_cosets = {rand_poly(en), rand_poly(en), rand_poly(en), rand_poly(en)};
    pk.cs_degree = degree;          // chunks of degree - 2 = 3 columns: two sets
    pk.blinding_factors = 5;
    const std::vector<Polynomial> advice = {rand_poly(n), rand_poly(n)}, instance = {rand_poly(n)};
    const std::vector<Fr> challenges = {rand_fr()};
    const Fr y = rand_fr(), beta = rand_fr(), gamma = rand_fr(), theta = rand_fr();
    const std::vector<plonk::LookupCommitted> lookups = {{rand_poly(n), rand_poly(n), rand_poly(n)}};
    plonk::PermutationCommitted perm;
    perm.permutation_product_cosets = {rand_poly(en), rand_poly(en)};

    const Polynomial h = ev.evaluate_h(pk, advice, instance, challenges, y, beta, gamma, theta, lookups, perm);

    // the oracle takes extended cosets: extend what upstream passes in coefficient form
    auto extend = [&](const Polynomial& p) {
        Polynomial e{std::vector<Fr>(en)};
        oracle_coeff_to_extended(od, u64(p), u64(e));
        return e;
    };
    const std::vector<Polynomial> advice_e = {extend(advice[0]), extend(advice[1])}, instance_e = {extend(instance[0])};
    const std::vector<plonk::LookupCommitted> lookups_e = {
        {extend(lookups[0].product_poly), extend(lookups[0].permuted_input_poly), extend(lookups[0].permuted_table_poly)}};
    const plonk::EvaluateHCall call(ev, pk, advice_e, instance_e, challenges, y, beta, gamma, theta, lookups_e, perm,
                                    ZKHIP_FORM_EXTENDED);
    Polynomial exp{std::vector<Fr>(en)};
    CHECK(oracle_evaluate_h(od, &call.desc, u64(exp)) == 0, "oracle evaluate_h runs");
    CHECK(h.values == exp.values, "Evaluator::evaluate_h == oracle (gates, permutation, lookup)");
    bool nonzero = false;
    for (const Fr& v : h.values) nonzero |= v != Fr{{0, 0, 0, 0}};
    CHECK(nonzero, "the quotient numerator is not trivially zero");
    CHECK(panics([&] { ev.evaluate_h(pk, {advice[0]}, instance, challenges, y, beta, gamma, theta, lookups, perm); }),
          "evaluate_h panics when the graph names an advice column that was not passed");
    CHECK(panics([&] { ev.evaluate_h(pk, advice, instance, challenges, y, beta, gamma, theta, {}, perm); }),
          "evaluate_h panics when lookups and their graphs differ in number");
    oracle_domain_free(od);
}

// field arithmetic of the checker: op 0 add, 1 sub, 2 mul
static Fr fop(int op, const Fr& a, const Fr& b) {
    Fr o;
    oracle_field_binop(0, op, a.l, b.l, o.l, 1);
    return o;
}
static Fr horner(const std::vector<Fr>& p, const Fr& x) {
    Fr acc = mont(0);
    for (size_t i = p.size(); i-- > 0;) acc = fop(0, fop(2, acc, x), p[i]);
    return acc;
}
static std::vector<std::array<uint64_t, 4>> multiset(const std::vector<Fr>& v, size_t rows) {
    std::vector<std::array<uint64_t, 4>> m(rows);
    for (size_t i = 0; i < rows; i++) std::memcpy(m[i].data(), v[i].l, 32);
    std::sort(m.begin(), m.end());
    return m;
}

static void test_openings_and_lookup() {
    const Polynomial a = rand_poly(1537);
    const Fr x = rand_fr(), b = rand_fr();
    CHECK(arithmetic::eval_polynomial(a.values, x) == horner(a.values, x), "eval_polynomial(poly, point) == Horner");
    CHECK(arithmetic::eval_polynomial({}, x) == mont(0), "eval_polynomial of the empty polynomial is zero");
    // a(X) - a(b) = q(X) (X - b), checked at a random point
    const std::vector<Fr> q = arithmetic::kate_division(a.values, b);
    CHECK(q.size() == a.values.size() - 1, "kate_division returns len - 1 coefficients");
    CHECK(fop(1, horner(a.values, x), horner(a.values, b)) == fop(2, horner(q, x), fop(1, x, b)), "kate_division(a, b): a(x) - a(b) == q(x) (x - b)");
    CHECK(panics([&] { arithmetic::kate_division({}, b); }), "kate_division panics on the empty polynomial");

    std::vector<Fr> v = rand_poly(1000).values, orig;
    v[3] = v[999] = mont(0);
    orig = v;
    arithmetic::batch_invert(v);
    bool ok = v[3] == mont(0) && v[999] == mont(0);
    for (size_t i = 0; i < v.size(); i++)
        if (!(orig[i] == mont(0))) ok = ok && fop(2, v[i], orig[i]) == mont(1);
    CHECK(ok, "batch_invert: x * x^-1 == 1, zeros stay zero");

    // lookup: every input value occurs in the table
    const uint32_t k = 9, blinding = 5;
    const size_t n = size_t(1) << k, usable = n - (blinding + 1);
    poly::EvaluationDomain domain(4, k);
    Polynomial table = rand_poly(n), input;
    for (size_t i = 0; i < n; i++)
        if (i % 7 == 3) table.values[i] = table.values[i - 1];       // repeated table values
    input.values.resize(n);
    for (size_t i = 0; i < n; i++) input.values[i] = table.values[(i * i + 3 * i) % (usable / 4)];
    const auto permuted = plonk::lookup::permute_expression_pair(domain, blinding, input, table);
    const Polynomial &pa = permuted.first, &ps = permuted.second;
    CHECK(multiset(pa.values, usable) == multiset(input.values, usable), "permuted input is a permutation of the input's usable rows");
    CHECK(multiset(ps.values, usable) == multiset(table.values, usable), "permuted table is a permutation of the table's usable rows");
    ok = pa.values[0] == ps.values[0];
    for (size_t i = 1; i < usable; i++) ok = ok && (pa.values[i] == ps.values[i] || pa.values[i] == pa.values[i - 1]);
    CHECK(ok, "A'[i] == S'[i] or A'[i] == A'[i-1] on every usable row");
    ok = true;
    for (size_t i = usable; i < n; i++) ok = ok && pa.values[i] == mont(0) && ps.values[i] == mont(0);
    CHECK(ok, "blinding rows are left to the caller");
    input.values[17] = rand_fr();
    bool failure = false;
    try { plonk::lookup::permute_expression_pair(domain, blinding, input, table); }
    catch (const plonk::lookup::ConstraintSystemFailure&) { failure = true; }
    CHECK(failure, "an input value missing from the table is a ConstraintSystemFailure");
}

int main(int argc, char** argv) {
    std::setvbuf(stdout, nullptr, _IOLBF, 0);
    if (argc > 1 && std::string(argv[1]) == "--list") {     // no device needed: proves the binary links and loads
        std::printf("best_multiexp best_fft EvaluationDomain ParamsKZG evaluate_h (%s)\n", zkhip_version());
        return 0;
    }
    try {
        test_best_multiexp();
        test_best_fft_and_domain();
        test_evaluate_h();
        test_openings_and_lookup();
    } catch (const std::exception& e) {
        std::printf("FAIL  uncaught: %s\n", e.what());
        return 2;
    }
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "all ok", failures);
    return failures ? 1 : 0;
}
