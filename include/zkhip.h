/*
 * zkhip.h — C ABI of the MI355X (gfx950) accelerator for the halo2 `create_proof` polynomial pipeline.
 *
 * Each entry point replaces one function of the `halo2_proofs 0.3.0` fork that the reference links as an
 * out-of-tree path dependency (/root/reference/zkevm-circuits/Cargo.toml:15, /root/reference/Cargo.lock:2189-2211)
 * and reaches through `create_proof` (/root/reference/zkevm-circuits/benches/benchmark_list/main.rs:265,315;
 * /root/reference/prover/src/chunk/prover.rs:85; /root/reference/prover/src/batch/prover.rs:149-150) and
 * `keygen_vk/keygen_pk` (/root/reference/zkevm-circuits/benches/benchmark_list/main.rs:121,132,189,194).
 * The comments name the replaced symbol of that crate.
 *
 * Conventions
 *  - Field elements (`halo2curves::bn256::{Fr,Fq}`) are 4 x uint64_t little-endian limbs in Montgomery form
 *    (R = 2^256), exactly the in-memory `Fr([u64;4])`.  G1Affine = x||y (8 x u64, identity = all zero),
 *    G1 (Jacobian X/Z^2, Y/Z^3) = x||y||z (12 x u64, identity has z = 0).
 *  - Every function returns 0 on success or a negative zkhip_status; nothing unwinds or aborts across
 *    the boundary.  zkhip_last_error(ctx) returns a description of the last failure on that context.
 *  - One context per device.  A context is not re-entrant: serialise calls per context; different
 *    contexts are independent.  All pointers stay owned by the caller.
 *  - `_dev` variants take device pointers (hipMalloc'd or zkhip_dev_alloc'd) and enqueue on the context's
 *    stream without synchronising; host variants copy in, run, copy out and synchronise.
 *  - There is no CPU fallback: every call fails with ZKHIP_ERR_HIP when no gfx950 device is usable.
 */
#ifndef ZKHIP_H
#define ZKHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ZKHIP_OK = 0,
    ZKHIP_ERR_INVALID = -1,   /* bad argument (null pointer, size mismatch, unsupported log_n ...) */
    ZKHIP_ERR_HIP = -2,       /* HIP runtime failure, text in zkhip_last_error */
    ZKHIP_ERR_OOM = -3,       /* device allocation failed */
    ZKHIP_ERR_UNSUPPORTED = -4,
    ZKHIP_ERR_CONSTRAINT = -5 /* the witness violates a constraint the call depends on (upstream: Error::ConstraintSystemFailure) */
} zkhip_status;

typedef struct zkhip_ctx zkhip_ctx;
typedef struct zkhip_domain zkhip_domain;

/* ------------------------------------------------------------------ context / memory ---------- */
int zkhip_create(int device, zkhip_ctx** out);
void zkhip_destroy(zkhip_ctx* ctx);
const char* zkhip_last_error(const zkhip_ctx* ctx);
const char* zkhip_version(void);
/* Use an existing hipStream_t (e.g. the caller's current stream) instead of the context's own. NULL restores it. */
int zkhip_set_stream(zkhip_ctx* ctx, void* hip_stream);
int zkhip_sync(zkhip_ctx* ctx);
int zkhip_dev_alloc(zkhip_ctx* ctx, size_t bytes, void** dptr);
int zkhip_dev_free(zkhip_ctx* ctx, void* dptr);
/* free and total device memory of the context's GPU: callers size their batches of resident cosets with it (no reference
 * counterpart; the reference's runtime is given a fixed budget, benches/benchmark_list/main.rs:300-303). */
int zkhip_mem_info(zkhip_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
/* Page-locked host memory for the buffers a caller hands over (witness columns): the device reads it by DMA at the
 * link's rate, and zkhip_upload_async from it returns at once (the copy is ordered on the context's stream before the
 * calls that follow; the memory must stay untouched until they have run or zkhip_sync returns).  From ordinary host
 * memory zkhip_upload_async behaves like zkhip_upload without the final synchronisation. */
int zkhip_host_alloc(zkhip_ctx* ctx, size_t bytes, void** hptr);
int zkhip_host_free(zkhip_ctx* ctx, void* hptr);
int zkhip_upload_async(zkhip_ctx* ctx, void* dptr, const void* host, size_t bytes);
/* Copies that run BESIDE the context's work, on a transfer stream of the context's own: the witness columns of the next
 * commitments arrive while the current ones are computed (the reference's prover holds the witness in host memory and
 * commits column after column, benches/benchmark_list/main.rs:277-285).  zkhip_upload_overlapped returns at once; the
 * first copy after a fence is ordered behind the work the context holds at that moment, the following ones behind each
 * other only.  Calls made after zkhip_upload_fence see every copy issued before it; calls made before it must not touch
 * the targets.  zkhip_sync implies a fence.  The host memory should be page-locked (zkhip_host_alloc) and stays
 * untouched until a fence has been followed by zkhip_sync or by a call that returns results. */
int zkhip_upload_overlapped(zkhip_ctx* ctx, void* dptr, const void* host, size_t bytes);
int zkhip_upload_fence(zkhip_ctx* ctx);
int zkhip_upload(zkhip_ctx* ctx, void* dptr, const void* host, size_t bytes);
int zkhip_download(zkhip_ctx* ctx, void* host, const void* dptr, size_t bytes);
/* device-to-device copy on the context's stream, asynchronous (a column kept in two forms at once) */
int zkhip_copy_dev(zkhip_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);

/* ------------------------------------------------------------------ field / curve vector ops -----
 * Element-wise helpers used by the parity tests and by callers that keep polynomials on the device
 * (`Polynomial<F,B>` +,-,* in halo2_proofs::poly).  field: 0 = Fr, 1 = Fq.  op: 0 add, 1 sub, 2 mul. */
int zkhip_field_binop(zkhip_ctx* ctx, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* Batch inversion (`ff::BatchInvert`, zero maps to zero). */
int zkhip_field_batch_invert(zkhip_ctx* ctx, int field, const uint64_t* a, uint64_t* out, size_t n);
/* out[i] = affine(P[i] + Q[i]) for affine inputs — exercises the device G1 group law on its own. */
int zkhip_g1_add_affine(zkhip_ctx* ctx, const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* out_xy, size_t n);
/* out[i] = affine((first + i) * P): synthetic SRS ([1]G, [2]G, ...) for benchmarks and tests. */
int zkhip_g1_multiples(zkhip_ctx* ctx, const uint64_t base_xy[8], uint64_t first, size_t n, uint64_t* out_xy);
/* Throughput probe: runs `iters` dependent Montgomery multiplications per lane on every CU and returns
 * achieved 32x32->64 multiply-adds per second (the VALU integer peak the MSM roofline is quoted against). */
int zkhip_probe_mad_rate(zkhip_ctx* ctx, int iters, double* mads_per_second, double* fr_mul_per_second);
/* Self-test of the device arithmetic (no counterpart in the reference; for integrators and the test suite): the lazy
 * signed-digit multiplier in every form the kernels use -- product, square, a b - c d, a b + c d, lazy and negated
 * operands, zero tests -- is checked against the packed Montgomery arithmetic on `quads` seeded quadruples of Fr and of
 * Fq elements, edge values included.  out_failed[0..3]: bit masks of failed checks for Fr, Fr with chained columns, Fq,
 * Fq with chained columns; all zero on a correct build. */
int zkhip_selftest_arithmetic(zkhip_ctx* ctx, uint64_t seed, size_t quads, uint32_t out_failed[4]);

/* ------------------------------------------------------------------ MSM ---------------------------
 * Replaces `halo2_proofs::arithmetic::best_multiexp(coeffs, bases) -> C::Curve` as called by
 * `ParamsKZG::{commit, commit_lagrange}`.  Bases (the SRS `g` or `g_lagrange`) are registered once. */
int zkhip_register_bases(zkhip_ctx* ctx, const uint64_t* affine_xy, size_t n, uint32_t* out_id);
int zkhip_release_bases(zkhip_ctx* ctx, uint32_t bases_id);
/* scalars: n x 4 Montgomery Fr on the host; uses bases[0..n).  out: Jacobian, 12 x u64. */
int zkhip_msm(zkhip_ctx* ctx, uint32_t bases_id, const uint64_t* scalars, size_t n, uint64_t out_jacobian[12]);
/* batch of `batch` scalar vectors (device pointers, each n x 4) against the same bases; out: batch x 12 (host). */
int zkhip_msm_batch_dev(zkhip_ctx* ctx, uint32_t bases_id, const uint64_t* const* d_scalars, size_t batch, size_t n,
                        uint64_t* out_jacobian);
/* MSM over a base sub-range [first, first+n): the unit of the multi-GPU base-range split. */
int zkhip_msm_range_dev(zkhip_ctx* ctx, uint32_t bases_id, size_t first, const uint64_t* d_scalars, size_t n,
                        uint64_t out_jacobian[12]);
/* `Curve::batch_normalize` + point addition helpers for folding per-GPU partial sums. */
int zkhip_g1_sum_jacobian(const uint64_t* jacobians, size_t count, uint64_t out_jacobian[12]);
int zkhip_g1_to_affine(const uint64_t jacobian[12], uint64_t out_xy[8]);

/* ------------------------------------------------------------------ NTT ---------------------------
 * Replaces `halo2_proofs::arithmetic::best_fft(a, omega, log_n)`: in place, natural order in and out,
 * out[i] = sum_j a[j] * omega^(i*j).  omega must have order exactly 2^log_n. */
int zkhip_ntt(zkhip_ctx* ctx, uint64_t* a, uint32_t log_n, const uint64_t omega[4]);
/* batch of device-resident vectors laid out back to back (stride = 2^log_n elements). */
int zkhip_ntt_dev(zkhip_ctx* ctx, uint64_t* d_a, uint32_t log_n, const uint64_t omega[4], size_t batch);

/* ------------------------------------------------------------------ EvaluationDomain --------------
 * Replaces `halo2_proofs::poly::EvaluationDomain<Fr>`.  zkhip_domain_create = `EvaluationDomain::new(j, k)`. */
int zkhip_domain_create(zkhip_ctx* ctx, uint32_t j, uint32_t k, zkhip_domain** out);
void zkhip_domain_destroy(zkhip_domain* dom);
uint32_t zkhip_domain_k(const zkhip_domain* dom);
uint32_t zkhip_domain_extended_k(const zkhip_domain* dom);
uint32_t zkhip_domain_quotient_poly_degree(const zkhip_domain* dom);
/* which: 0 omega, 1 omega_inv, 2 extended_omega, 3 extended_omega_inv, 4 g_coset, 5 g_coset_inv,
 *        6 ifft_divisor, 7 extended_ifft_divisor   (`get_omega`, `get_extended_omega`, ...) */
int zkhip_domain_constant(const zkhip_domain* dom, int which, uint64_t out[4]);
/* host, in place / out of place, one polynomial: */
int zkhip_lagrange_to_coeff(zkhip_domain* dom, uint64_t* a /* 2^k */);
int zkhip_coeff_to_lagrange(zkhip_domain* dom, uint64_t* a /* 2^k */);
int zkhip_coeff_to_extended(zkhip_domain* dom, const uint64_t* coeff /* 2^k */, uint64_t* out /* 2^ext_k */);
int zkhip_extended_to_coeff(zkhip_domain* dom, const uint64_t* ext /* 2^ext_k */, uint64_t* out /* 2^k * (j-1) */);
int zkhip_divide_by_vanishing_poly(zkhip_domain* dom, uint64_t* ext /* 2^ext_k */);
/* device-resident, batched (vectors back to back); cosets never round-trip to the host: */
int zkhip_lagrange_to_coeff_dev(zkhip_domain* dom, uint64_t* d_a, size_t batch);
int zkhip_coeff_to_lagrange_dev(zkhip_domain* dom, uint64_t* d_a, size_t batch);
int zkhip_coeff_to_extended_dev(zkhip_domain* dom, const uint64_t* d_coeff, uint64_t* d_out, size_t batch);
int zkhip_extended_to_coeff_dev(zkhip_domain* dom, uint64_t* d_ext /* clobbered */, uint64_t* d_out, size_t batch);
/* coeff_to_extended with the result split by the parity of the extended row: per member the 2^(ext_k-1) even-indexed rows
 * first, then the odd-indexed ones.  Each half is the compact layout (see the coset subsets below) of the even / the odd
 * cosets, in ascending order, so zkhip_evaluate_h reads a half through desc.cosets = {0, 2, ..} or {1, 3, ..}.  No
 * reference counterpart: upstream keeps one layout because it evaluates every term on every row. */
int zkhip_coeff_to_extended_split_dev(zkhip_domain* dom, const uint64_t* d_coeff, uint64_t* d_out, size_t batch);
/* Values of a polynomial of degree < 2^(ext_k-1) on the even-indexed rows of the extended domain -> its values on the
 * odd-indexed rows (one inverse and one forward transform of half the extended size).  batch members back to back. */
int zkhip_extend_half_dev(zkhip_domain* dom, const uint64_t* d_even, uint64_t* d_odd, size_t batch);
/* coeff_to_extended onto the even-indexed rows only: per member the 2^(ext_k-1) values that the first half of
 * zkhip_coeff_to_extended_split_dev holds, written with a stride of out_stride elements between members (2^ext_k to fill
 * the even halves of vectors in the split layout and leave their odd halves alone), by one transform of half the size. */
int zkhip_coeff_to_extended_half_dev(zkhip_domain* dom, const uint64_t* d_coeff, uint64_t* d_out, size_t out_stride, size_t batch);
int zkhip_divide_by_vanishing_poly_dev(zkhip_domain* dom, uint64_t* d_ext, size_t batch);

/* ------------------------------------------------------------------ evaluate_h --------------------
 * Replaces `halo2_proofs::plonk::evaluation::Evaluator::evaluate_h(pk, advice_polys, instance_polys,
 * challenges, y, beta, gamma, theta, lookups, shuffles, permutations)` for one circuit instance.
 *
 * zkhip_graph is the serialised `GraphEvaluator { constants, rotations, calculations }`.
 * calc_words: for each calculation, in order (calculation i writes intermediate i):
 *      op, then its ValueSources as (kind, a, b) triples:
 *        ADD/SUB/MUL: two;  SQUARE/DOUBLE/NEGATE/STORE: one;
 *        HORNER: start, factor, nparts, then nparts part triples.
 *   kind: 0 Constant(a) 1 Intermediate(a) 2 Fixed(col a, rotation index b) 3 Advice(a,b) 4 Instance(a,b)
 *         5 Challenge(a) 6 Beta 7 Gamma 8 Theta 9 Y 10 PreviousValue
 *   op:   0 Add 1 Sub 2 Mul 3 Square 4 Double 5 Negate 6 Horner 7 Store
 */
typedef struct {
    const uint64_t* constants;   /* num_constants x 4 */
    uint32_t num_constants;
    const int32_t* rotations;
    uint32_t num_rotations;
    const uint32_t* calc_words;
    uint32_t num_calc_words;
    uint32_t num_calculations;
} zkhip_graph;

typedef struct {
    zkhip_graph graph;                 /* last calculation = (a_compressed + beta) * (s_compressed + gamma) */
    const uint64_t* product;           /* lookup::prover::Committed::product_poly        */
    const uint64_t* permuted_input;    /* ...::permuted_input_poly                       */
    const uint64_t* permuted_table;    /* ...::permuted_table_poly                       */
} zkhip_lookup;

enum { ZKHIP_MEM_HOST = 0, ZKHIP_MEM_DEVICE = 1 };
enum { ZKHIP_FORM_COEFF = 0, ZKHIP_FORM_EXTENDED = 1 };
enum { ZKHIP_BASIS_EXTENDED = 0, ZKHIP_BASIS_LAGRANGE = 1 };

typedef struct {
    uint32_t mem;            /* where every polynomial pointer below (and `out`) lives */
    uint32_t poly_form;      /* form of advice/instance/lookup polynomials: COEFF (2^k, as upstream passes them;
                                the library runs coeff_to_extended) or EXTENDED (2^ext_k cosets) */
    uint32_t num_fixed, num_advice, num_instance, num_challenges;
    const uint64_t* const* fixed_cosets;   /* pk.fixed_cosets, always extended        */
    const uint64_t* const* advice;
    const uint64_t* const* instance;
    const uint64_t* challenges;            /* host, num_challenges x 4                 */
    uint64_t y[4], beta[4], gamma[4], theta[4];
    const uint64_t* l0;                    /* pk.l0, pk.l_last, pk.l_active_row (extended) */
    const uint64_t* l_last;
    const uint64_t* l_active_row;
    const zkhip_graph* custom_gates;       /* may be NULL                              */
    uint32_t num_lookups;
    const zkhip_lookup* lookups;
    uint32_t num_perm_columns;             /* permutation argument columns, in order   */
    const uint32_t* perm_column_kind;      /* 2 Fixed, 3 Advice, 4 Instance            */
    const uint32_t* perm_column_index;
    const uint64_t* const* perm_sigma_cosets;    /* pk.permutation.cosets (extended)   */
    uint32_t num_perm_sets;
    const uint64_t* const* perm_product_cosets;  /* permutation_product_coset per set (extended) */
    uint32_t perm_chunk_len;               /* cs.degree() - 2                          */
    uint32_t blinding_factors;             /* cs.blinding_factors()                    */
    /* Multi-GPU split (no reference counterpart: upstream splits rows over rayon threads): num_cosets = 0 evaluates the
     * whole extended domain.  Otherwise only the listed cosets (see zkhip_coeff_to_cosets_dev) are evaluated; every
     * "extended" vector above and `out` are then in the compact layout of that list, 2^k * num_cosets rows.         */
    uint32_t num_cosets;
    const uint32_t* cosets;                /* host, num_cosets distinct values < 2^(extended_k - k) */
    /* ZKHIP_BASIS_LAGRANGE: evaluate `custom_gates` alone over the 2^k rows of the circuit (every column given with 2^k
     * rows, rotations move single rows, out has 2^k rows): `plonk::lookup::prover::Argument::commit_permuted`'s
     * compress_expressions, with the theta-Horner of the expressions as the graph's last calculation.                 */
    uint32_t basis;
    /* row_count != 0: evaluate only the rows [row_first, row_first + row_count) of the domain the call works on (the
     * extended domain, or the compact layout of `cosets`); every column is still given whole, `out` has row_count rows.
     * More GPUs than cosets share a coset this way without exchanging neighbouring rows (no reference counterpart). */
    uint32_t row_first, row_count;
    /* Which terms of h are evaluated (no reference counterpart): h = sum_i y^(N-1-i) term_i over the gate polynomials and
     * the permutation / lookup identities in upstream's order.  ZKHIP_TERMS_LOW keeps the terms whose degree as a polynomial
     * in the columns is at most term_degree_max, ZKHIP_TERMS_HIGH the others; every kept term keeps its power of y, so the
     * two results add up to the ZKHIP_TERMS_ALL result.  A sum of terms of degree <= d is determined by its values on
     * d * 2^k points: with d = 2^(e-1) the low part is evaluated on the even-indexed half of the extended domain only
     * (the compact layout of the even cosets) and extended to the other half by zkhip_extend_half_dev. */
    uint32_t term_select;
    uint32_t term_degree_max;
} zkhip_evalh_desc;
enum { ZKHIP_TERMS_ALL = 0, ZKHIP_TERMS_LOW = 1, ZKHIP_TERMS_HIGH = 2 };

/* out: 2^ext_k x 4 (host or device per desc->mem) = numerator of the quotient on the extended coset (2^k * num_cosets rows
 * with a coset list, row_count rows with a row window). */
int zkhip_evaluate_h(zkhip_domain* dom, const zkhip_evalh_desc* desc, uint64_t* out);
/* Size of the fused program the last zkhip_evaluate_h (or zkhip_evaluate_h_prepare) call on this context ran (built) per extended row: out[0] instructions,
 * out[1] field multiplications, out[2] column reads, out[3] value slots (for throughput accounting; no reference
 * counterpart). */
int zkhip_evaluate_h_last_stats(zkhip_ctx* ctx, uint64_t out[4]);
/* The fused program depends on the graphs and counts of the descriptor only, so the context keeps it per shape (upstream
 * builds `Evaluator` once per proving key, `ProvingKey::ev`); a later call of the same shape links it against its column
 * pointers and allocates nothing for device-resident extended columns.  out[0] = programs built, out[1] = calls served
 * from the cache, on this context so far (no reference counterpart). */
int zkhip_evaluate_h_cache_stats(zkhip_ctx* ctx, uint64_t out[2]);
/* Builds and keeps the program of the descriptor's shape without evaluating anything: `Evaluator::new(cs)` as key
 * generation runs it (`ProvingKey::ev`).  Only the counts, the graphs and the permutation column kinds / indices of the
 * descriptor are read; every polynomial pointer may be NULL. */
int zkhip_evaluate_h_prepare(zkhip_domain* dom, const zkhip_evalh_desc* desc);
/* Several graphs over the same Lagrange-basis columns in one call: what `plonk::lookup::prover::Argument::commit_permuted`
 * (reached from /root/reference/zkevm-circuits/benches/benchmark_list/main.rs:353-354 through create_proof) does lookup after
 * lookup with `compress_expressions` -- one theta-compressed input and one table expression per lookup, each a graph whose value
 * over the 2^k rows goes to d_outs[i].  desc: basis = ZKHIP_BASIS_LAGRANGE, mem = ZKHIP_MEM_DEVICE, the fixed / advice /
 * instance columns (2^k rows each), challenges and theta; its custom_gates, lookup and permutation fields are not read.
 * Equal to `count` calls of zkhip_evaluate_h with custom_gates = &graphs[i], without the host waiting for the device
 * between them: the programs are staged together and launched back to back. */
int zkhip_evaluate_graphs_dev(zkhip_domain* dom, const zkhip_evalh_desc* desc, const zkhip_graph* graphs, uint32_t count,
                              uint64_t* const* d_outs);
/* Which columns the program of the descriptor's shape reads (built and kept like zkhip_evaluate_h_prepare does): used[c] = 1
 * or 0 for column c of the order  fixed | advice | instance | l0, l_last, l_active | X (only with a permutation argument) |
 * permutation products | permutation sigmas | per lookup (product, permuted input, permuted table);  *count = their number
 * (<= capacity).  With desc.term_select = ZKHIP_TERMS_HIGH this names the columns whose odd-indexed extended rows the split
 * evaluation reads at all: the others are extended to the even-indexed half only (zkhip_coeff_to_extended_half_dev).  No
 * reference counterpart. */
int zkhip_evaluate_h_columns_used(zkhip_domain* dom, const zkhip_evalh_desc* desc, uint8_t* used, uint32_t capacity,
                                  uint32_t* count);

/* ------------------------------------------------------------------ grand products and openings -----
 * The prover steps adjacent to the hot path (the rest of `plonk::prover::create_proof`, reached from
 * zkevm-circuits/benches/benchmark_list/main.rs:265 and prover/src/chunk/prover.rs:85).  All pointers are device
 * pointers to Montgomery Fr vectors of n = 2^k elements on the domain's context; calls are asynchronous on the
 * context's stream except where a host result is returned.
 *
 * zkhip_batch_invert_dev       `ff::BatchInvert::batch_invert`: every non-zero element replaced by its inverse, zeros kept.
 * zkhip_prefix_product_dev     out[0] = init, out[i] = init * m[0] * ... * m[i-1]  (n outputs; m may alias out).
 * zkhip_permutation_product_dev  one column set of `plonk::permutation::prover::Argument::commit`:
 *     z[0] = z0, z[i+1] = z[i] * prod_j (v_j[i] + delta^(j0+j) beta omega^i + gamma) / (v_j[i] + beta sigma_j[i] + gamma)
 *     d_values / d_sigmas: HOST arrays of num_cols device pointers (column values and `pk.permutation.permutations`,
 *     Lagrange basis); delta_start = delta^j0 for the set's first column j0; z0 = 1 for the first set, last_z after.
 * zkhip_lookup_product_dev     `plonk::lookup::prover::Permuted::commit_product`:
 *     z[0] = 1, z[i+1] = z[i] * (a[i] + beta)(s[i] + gamma) / ((a'[i] + beta)(s'[i] + gamma))
 * Both write all n rows of z; upstream then overwrites the last `blinding_factors` rows with values from the prover's
 * RNG, which stays on the caller's side (last_z is row n - blinding_factors - 1).
 * zkhip_eval_polynomial_dev    `arithmetic::eval_polynomial`: sum_i coeff[i] x^i, result on the host (synchronises).
 *
 * The polynomial arithmetic of the SHPLONK opening (`poly::kzg::multiopen::shplonk::prover::ProverSHPLONK::create_proof`,
 * selected at zkevm-circuits/benches/benchmark_list/main.rs:26,267 and prover/src/test/proof_test.rs:297):
 * zkhip_poly_lincomb_dev       out[i] = sum_k coeffs[k] * p_k[i] for i < n, p_k read as zero beyond lens[k] <= n.  This is
 *     what upstream's chains of `poly * scalar` and `acc + &poly` compute for the quotient and linearisation
 *     polynomials.  d_polys / lens: HOST arrays of num device pointers / lengths; coeffs: HOST, num x 4 limbs.
 *     d_out may be one of the inputs (every output element depends on the same element of the inputs only).
 * zkhip_kate_division_dev      `arithmetic::kate_division(a, point)`: the n - 1 coefficients of the quotient of
 *     a(X) by (X - point); the remainder a(point) is dropped, as upstream drops it.  d_out must not overlap d_poly. */
int zkhip_batch_invert_dev(zkhip_ctx* ctx, int field, uint64_t* d_a, size_t n);
int zkhip_prefix_product_dev(zkhip_ctx* ctx, const uint64_t* d_m, uint64_t* d_out, size_t n, const uint64_t init[4]);
/* zkhip_prefix_product_dev for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments and results as
 * zkhip_prefix_product_dev; the per-block totals live in a static device array of 256 KiB per GPU instead of the
 * context's growable scratch buffer, so the call performs no hipMalloc / hipFree.  n is limited to 2^24 elements
 * (ZKHIP_ERR_INVALID beyond).  All contexts on one GPU share that array: calls that overlap on one GPU race. */
int zkhip_prefix_product_noalloc_dev(zkhip_ctx* ctx, const uint64_t* d_m, uint64_t* d_out, size_t n, const uint64_t init[4]);
int zkhip_permutation_product_dev(zkhip_domain* dom, const uint64_t* const* d_values, const uint64_t* const* d_sigmas,
                                  uint32_t num_cols, const uint64_t beta[4], const uint64_t gamma[4],
                                  const uint64_t delta_start[4], const uint64_t z0[4], uint64_t* d_z);
int zkhip_lookup_product_dev(zkhip_domain* dom, const uint64_t* d_compressed_input, const uint64_t* d_compressed_table,
                             const uint64_t* d_permuted_input, const uint64_t* d_permuted_table, const uint64_t beta[4],
                             const uint64_t gamma[4], uint64_t* d_z);
/* zkhip_lookup_product_dev for `count` lookups (the loop over `lookups` in `plonk::prover::create_proof`, between the beta /
 * gamma challenges and the commitments of the products): the same values, with the denominators of up to 32 lookups inverted
 * by one launch.  The z vectors must be distinct. */
int zkhip_lookup_product_batch_dev(zkhip_domain* dom, const uint64_t* const* d_compressed_inputs,
                                   const uint64_t* const* d_compressed_tables, const uint64_t* const* d_permuted_inputs,
                                   const uint64_t* const* d_permuted_tables, const uint64_t beta[4], const uint64_t gamma[4],
                                   uint64_t* const* d_zs, uint32_t count);
int zkhip_eval_polynomial_dev(zkhip_ctx* ctx, const uint64_t* d_coeff, size_t n, const uint64_t x[4], uint64_t out[4]);
/* `count` calls of `arithmetic::eval_polynomial` in one launch and one synchronisation: out[i] = d_coeffs[i](points[i]),
 * every polynomial with n coefficients (the evaluations `create_proof` writes to the transcript before the multi-open:
 * one per (polynomial, rotation) query).  d_coeffs: HOST array of device pointers; points, out: HOST, count x 4 limbs. */
int zkhip_eval_polynomials_dev(zkhip_ctx* ctx, const uint64_t* const* d_coeffs, const uint64_t* points, uint32_t count, size_t n,
                               uint64_t* out);
/* `plonk::lookup::prover::permute_expression_pair`: rows [0, usable_rows) of the permuted input A' (the compressed input
 * sorted by canonical value) and of the permuted table S' (S'[i] = A'[i] where A'[i] first occurs; the table values left
 * over, ascending, fill the repeated rows from the last one backwards).  usable_rows = n - (blinding_factors + 1); the
 * remaining rows are upstream's blinding rows, drawn from the prover's RNG by the caller.  Returns ZKHIP_ERR_CONSTRAINT
 * when an input value is missing from the table (upstream: Error::ConstraintSystemFailure).  Synchronises. */
int zkhip_lookup_permute_dev(zkhip_domain* dom, const uint64_t* d_compressed_input, const uint64_t* d_compressed_table,
                             uint32_t usable_rows, uint64_t* d_permuted_input, uint64_t* d_permuted_table);
/* The same for `count` lookups in one sequence of launches (all lookups of a proof: no challenge is squeezed between them,
 * so `create_proof` can permute them together).  Every argument of zkhip_lookup_permute_dev becomes a HOST array of
 * `count` device pointers; the error names the first lookup whose input is not contained in its table. */
int zkhip_lookup_permute_batch_dev(zkhip_domain* dom, uint32_t count, const uint64_t* const* d_compressed_inputs,
                                   const uint64_t* const* d_compressed_tables, uint32_t usable_rows,
                                   uint64_t* const* d_permuted_inputs, uint64_t* const* d_permuted_tables);
/* `arithmetic::best_fft` on G1 points (host pointers, n = 2^log_n affine points in and out): out[i] = scale * sum_j
 * omega^(i j) in[j].  `poly::kzg::commitment::ParamsKZG::setup` / `g_to_lagrange` obtain `g_lagrange` from `g` this way with
 * omega = omega_n^-1 and scale = n^-1 (parameters are built at zkevm-circuits/benches/benchmark_list/main.rs:184).
 * scale may be NULL (no scaling).  Synchronises. */
int zkhip_g1_fft(zkhip_ctx* ctx, const uint64_t* affine_xy, uint32_t log_n, const uint64_t omega[4], const uint64_t* scale,
                 uint64_t* out_xy);
/* One permutation polynomial of `plonk::permutation::keygen::Assembly::build_pk` (key generation, reached from
 * zkevm-circuits/benches/benchmark_list/main.rs:132 `keygen_pk`): d_out[i] = delta^(col[i]) * omega^(row[i]) for the
 * 2^k rows of a column, (col[i], row[i]) = the cell that the copy-constraint cycles map row i of this column to
 * (device arrays of 2^k words each; every col[i] < num_cols, every row[i] < 2^k; Lagrange basis).  Synchronises. */
int zkhip_permutation_sigma_dev(zkhip_domain* dom, const uint32_t* d_map_col, const uint32_t* d_map_row, uint32_t num_cols,
                                uint64_t* d_out);
int zkhip_poly_lincomb_dev(zkhip_ctx* ctx, const uint64_t* const* d_polys, const size_t* lens, const uint64_t* coeffs,
                           uint32_t num, size_t n, uint64_t* d_out);
int zkhip_kate_division_dev(zkhip_ctx* ctx, const uint64_t* d_poly, size_t n, const uint64_t point[4], uint64_t* d_out);

/* ------------------------------------------------------------------ coset subsets (multi-GPU evaluate_h) ---
 * The extended domain zeta * <extended_omega> is the union of the 2^e cosets zeta * extended_omega^j * <omega>,
 * e = extended_k - k: extended row idx belongs to coset idx mod 2^e, and a rotation by r rows of the circuit moves
 * r * 2^e extended rows, i.e. stays inside its coset.  A GPU that owns whole cosets therefore evaluates its rows of h
 * without any exchange of neighbouring rows.  (Upstream's evaluate_h splits contiguous row ranges over rayon threads
 * in shared memory; this is the same computation partitioned so that it needs no shared memory.)
 * A "compact" vector for a list of num_cosets cosets (a power of two, distinct) has 2^k * num_cosets rows:
 * compact row q * num_cosets + s  =  extended row q * 2^e + cosets[s].
 * zkhip_coeff_to_cosets_dev     coefficients (2^k per member) -> compact values on the listed cosets: per coset one
 *                               size-2^k NTT of c_i (zeta extended_omega^j)^i instead of the size-2^extended_k transform.
 * zkhip_extended_to_cosets_dev  extended vector -> compact (gather; for pk.fixed_cosets, l0, ... computed at keygen).
 * zkhip_cosets_to_extended_dev  compact -> rows of the listed cosets in an extended vector (other rows untouched): how
 *                               the pieces of h from all GPUs are put together before divide_by_vanishing_poly. */
int zkhip_coeff_to_cosets_dev(zkhip_domain* dom, const uint64_t* d_coeff, const uint32_t* cosets, uint32_t num_cosets,
                              uint64_t* d_out, size_t batch);
int zkhip_extended_to_cosets_dev(zkhip_domain* dom, const uint64_t* d_ext, const uint32_t* cosets, uint32_t num_cosets,
                                 uint64_t* d_out);
int zkhip_cosets_to_extended_dev(zkhip_domain* dom, const uint64_t* d_compact, const uint32_t* cosets, uint32_t num_cosets,
                                 uint64_t* d_ext);

/* Kernel timing hook: average duration in microseconds of the launches of the named kernel family issued on
 * this context since the last zkhip_timing_reset (measured with HIP events on the launch stream).
 * families: "ntt", "msm_sort", "msm_accumulate", "msm_merge", "msm_reduce", "evalh", "batch_invert",
 * "prefix_product", "grand_product", "eval_polynomial", "poly_lincomb", "kate_division", "lookup_permute".  A family brackets every launch of its stage in one call (e.g. the passes of one NTT).
 * on = 1 times every family; on = 2 only "msm_accumulate" and "ntt" (each bracket puts two event packets on the
 * stream, about 30 us per MSM when all four of its stages are bracketed); on = 0 none. */
int zkhip_timing_enable(zkhip_ctx* ctx, int on);
int zkhip_timing_reset(zkhip_ctx* ctx);
int zkhip_timing_get(zkhip_ctx* ctx, const char* family, double* avg_us, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_H */
