//! `extern "C"` declarations of `include/zkhip.h`, one per function, in the header's order.
//!
//! Field elements are `[u64; 4]` little-endian limbs in Montgomery form (the in-memory `halo2curves::bn256::{Fr, Fq}`),
//! `G1Affine` is x ‖ y (8 limbs, identity all zero), a Jacobian `G1` is x ‖ y ‖ z (12 limbs, z = 0 for the identity).
//! Every function returns 0 or a negative `zkhip_status`; nothing unwinds across the boundary.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_double, c_int, c_void};

pub const ZKHIP_OK: c_int = 0;
pub const ZKHIP_ERR_INVALID: c_int = -1;
pub const ZKHIP_ERR_HIP: c_int = -2;
pub const ZKHIP_ERR_OOM: c_int = -3;
pub const ZKHIP_ERR_UNSUPPORTED: c_int = -4;
pub const ZKHIP_ERR_CONSTRAINT: c_int = -5;

pub const ZKHIP_MEM_HOST: u32 = 0;
pub const ZKHIP_MEM_DEVICE: u32 = 1;
pub const ZKHIP_FORM_COEFF: u32 = 0;
pub const ZKHIP_FORM_EXTENDED: u32 = 1;
pub const ZKHIP_BASIS_EXTENDED: u32 = 0;
pub const ZKHIP_BASIS_LAGRANGE: u32 = 1;
pub const ZKHIP_TERMS_ALL: u32 = 0;
pub const ZKHIP_TERMS_LOW: u32 = 1;
pub const ZKHIP_TERMS_HIGH: u32 = 2;

#[repr(C)]
pub struct zkhip_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkhip_domain {
    _private: [u8; 0],
}

/// Serialised `GraphEvaluator { constants, rotations, calculations }` (word stream documented in the header).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkhip_graph {
    pub constants: *const u64,
    pub num_constants: u32,
    pub rotations: *const i32,
    pub num_rotations: u32,
    pub calc_words: *const u32,
    pub num_calc_words: u32,
    pub num_calculations: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkhip_lookup {
    pub graph: zkhip_graph,
    pub product: *const u64,
    pub permuted_input: *const u64,
    pub permuted_table: *const u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkhip_evalh_desc {
    pub mem: u32,
    pub poly_form: u32,
    pub num_fixed: u32,
    pub num_advice: u32,
    pub num_instance: u32,
    pub num_challenges: u32,
    pub fixed_cosets: *const *const u64,
    pub advice: *const *const u64,
    pub instance: *const *const u64,
    pub challenges: *const u64,
    pub y: [u64; 4],
    pub beta: [u64; 4],
    pub gamma: [u64; 4],
    pub theta: [u64; 4],
    pub l0: *const u64,
    pub l_last: *const u64,
    pub l_active_row: *const u64,
    pub custom_gates: *const zkhip_graph,
    pub num_lookups: u32,
    pub lookups: *const zkhip_lookup,
    pub num_perm_columns: u32,
    pub perm_column_kind: *const u32,
    pub perm_column_index: *const u32,
    pub perm_sigma_cosets: *const *const u64,
    pub num_perm_sets: u32,
    pub perm_product_cosets: *const *const u64,
    pub perm_chunk_len: u32,
    pub blinding_factors: u32,
    pub num_cosets: u32,
    pub cosets: *const u32,
    pub basis: u32,
    pub row_first: u32,
    pub row_count: u32,
    pub term_select: u32,
    pub term_degree_max: u32,
}

extern "C" {
    // ---- context / memory
    pub fn zkhip_create(device: c_int, out: *mut *mut zkhip_ctx) -> c_int;
    pub fn zkhip_destroy(ctx: *mut zkhip_ctx);
    pub fn zkhip_last_error(ctx: *const zkhip_ctx) -> *const c_char;
    pub fn zkhip_version() -> *const c_char;
    pub fn zkhip_set_stream(ctx: *mut zkhip_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn zkhip_sync(ctx: *mut zkhip_ctx) -> c_int;
    pub fn zkhip_dev_alloc(ctx: *mut zkhip_ctx, bytes: usize, dptr: *mut *mut c_void) -> c_int;
    pub fn zkhip_dev_free(ctx: *mut zkhip_ctx, dptr: *mut c_void) -> c_int;
    pub fn zkhip_mem_info(ctx: *mut zkhip_ctx, free_bytes: *mut usize, total_bytes: *mut usize) -> c_int;
    pub fn zkhip_host_alloc(ctx: *mut zkhip_ctx, bytes: usize, hptr: *mut *mut c_void) -> c_int;
    pub fn zkhip_host_free(ctx: *mut zkhip_ctx, hptr: *mut c_void) -> c_int;
    pub fn zkhip_upload_async(ctx: *mut zkhip_ctx, dptr: *mut c_void, host: *const c_void, bytes: usize) -> c_int;
    pub fn zkhip_upload_overlapped(ctx: *mut zkhip_ctx, dptr: *mut c_void, host: *const c_void, bytes: usize) -> c_int;
    pub fn zkhip_upload_fence(ctx: *mut zkhip_ctx) -> c_int;
    pub fn zkhip_upload(ctx: *mut zkhip_ctx, dptr: *mut c_void, host: *const c_void, bytes: usize) -> c_int;
    pub fn zkhip_download(ctx: *mut zkhip_ctx, host: *mut c_void, dptr: *const c_void, bytes: usize) -> c_int;
    pub fn zkhip_copy_dev(ctx: *mut zkhip_ctx, d_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;

    // ---- field / curve vector helpers
    pub fn zkhip_field_binop(ctx: *mut zkhip_ctx, field: c_int, op: c_int, a: *const u64, b: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn zkhip_field_batch_invert(ctx: *mut zkhip_ctx, field: c_int, a: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn zkhip_g1_add_affine(ctx: *mut zkhip_ctx, p_xy: *const u64, q_xy: *const u64, out_xy: *mut u64, n: usize) -> c_int;
    pub fn zkhip_g1_multiples(ctx: *mut zkhip_ctx, base_xy: *const u64, first: u64, n: usize, out_xy: *mut u64) -> c_int;
    pub fn zkhip_probe_mad_rate(ctx: *mut zkhip_ctx, iters: c_int, mads_per_second: *mut c_double, fr_mul_per_second: *mut c_double) -> c_int;
    pub fn zkhip_selftest_arithmetic(ctx: *mut zkhip_ctx, seed: u64, quads: usize, out_failed: *mut u32) -> c_int;

    // ---- MSM: arithmetic::best_multiexp behind ParamsKZG::{commit, commit_lagrange}
    pub fn zkhip_register_bases(ctx: *mut zkhip_ctx, affine_xy: *const u64, n: usize, out_id: *mut u32) -> c_int;
    pub fn zkhip_release_bases(ctx: *mut zkhip_ctx, bases_id: u32) -> c_int;
    pub fn zkhip_msm(ctx: *mut zkhip_ctx, bases_id: u32, scalars: *const u64, n: usize, out_jacobian: *mut u64) -> c_int;
    pub fn zkhip_msm_batch_dev(ctx: *mut zkhip_ctx, bases_id: u32, d_scalars: *const *const u64, batch: usize, n: usize, out_jacobian: *mut u64) -> c_int;
    pub fn zkhip_msm_range_dev(ctx: *mut zkhip_ctx, bases_id: u32, first: usize, d_scalars: *const u64, n: usize, out_jacobian: *mut u64) -> c_int;
    pub fn zkhip_g1_sum_jacobian(jacobians: *const u64, count: usize, out_jacobian: *mut u64) -> c_int;
    pub fn zkhip_g1_to_affine(jacobian: *const u64, out_xy: *mut u64) -> c_int;

    // ---- NTT: arithmetic::best_fft
    pub fn zkhip_ntt(ctx: *mut zkhip_ctx, a: *mut u64, log_n: u32, omega: *const u64) -> c_int;
    pub fn zkhip_ntt_dev(ctx: *mut zkhip_ctx, d_a: *mut u64, log_n: u32, omega: *const u64, batch: usize) -> c_int;

    // ---- poly::EvaluationDomain
    pub fn zkhip_domain_create(ctx: *mut zkhip_ctx, j: u32, k: u32, out: *mut *mut zkhip_domain) -> c_int;
    pub fn zkhip_domain_destroy(dom: *mut zkhip_domain);
    pub fn zkhip_domain_k(dom: *const zkhip_domain) -> u32;
    pub fn zkhip_domain_extended_k(dom: *const zkhip_domain) -> u32;
    pub fn zkhip_domain_quotient_poly_degree(dom: *const zkhip_domain) -> u32;
    pub fn zkhip_domain_constant(dom: *const zkhip_domain, which: c_int, out: *mut u64) -> c_int;
    pub fn zkhip_lagrange_to_coeff(dom: *mut zkhip_domain, a: *mut u64) -> c_int;
    pub fn zkhip_coeff_to_lagrange(dom: *mut zkhip_domain, a: *mut u64) -> c_int;
    pub fn zkhip_coeff_to_extended(dom: *mut zkhip_domain, coeff: *const u64, out: *mut u64) -> c_int;
    pub fn zkhip_extended_to_coeff(dom: *mut zkhip_domain, ext: *const u64, out: *mut u64) -> c_int;
    pub fn zkhip_divide_by_vanishing_poly(dom: *mut zkhip_domain, ext: *mut u64) -> c_int;
    pub fn zkhip_lagrange_to_coeff_dev(dom: *mut zkhip_domain, d_a: *mut u64, batch: usize) -> c_int;
    pub fn zkhip_coeff_to_lagrange_dev(dom: *mut zkhip_domain, d_a: *mut u64, batch: usize) -> c_int;
    pub fn zkhip_coeff_to_extended_dev(dom: *mut zkhip_domain, d_coeff: *const u64, d_out: *mut u64, batch: usize) -> c_int;
    pub fn zkhip_extended_to_coeff_dev(dom: *mut zkhip_domain, d_ext: *mut u64, d_out: *mut u64, batch: usize) -> c_int;
    pub fn zkhip_divide_by_vanishing_poly_dev(dom: *mut zkhip_domain, d_ext: *mut u64, batch: usize) -> c_int;
    pub fn zkhip_coeff_to_extended_split_dev(dom: *mut zkhip_domain, d_coeff: *const u64, d_out: *mut u64, batch: usize) -> c_int;
    pub fn zkhip_extend_half_dev(dom: *mut zkhip_domain, d_even: *const u64, d_odd: *mut u64, batch: usize) -> c_int;
    pub fn zkhip_coeff_to_extended_half_dev(dom: *mut zkhip_domain, d_coeff: *const u64, d_out: *mut u64, out_stride: usize, batch: usize) -> c_int;

    // ---- plonk::evaluation::Evaluator::evaluate_h
    pub fn zkhip_evaluate_h(dom: *mut zkhip_domain, desc: *const zkhip_evalh_desc, out: *mut u64) -> c_int;
    pub fn zkhip_evaluate_h_last_stats(ctx: *mut zkhip_ctx, out: *mut u64) -> c_int;
    pub fn zkhip_evaluate_h_cache_stats(ctx: *mut zkhip_ctx, out: *mut u64) -> c_int;
    pub fn zkhip_evaluate_h_prepare(dom: *mut zkhip_domain, desc: *const zkhip_evalh_desc) -> c_int;
    pub fn zkhip_evaluate_graphs_dev(dom: *mut zkhip_domain, desc: *const zkhip_evalh_desc, graphs: *const zkhip_graph, count: u32, d_outs: *const *mut u64) -> c_int;
    pub fn zkhip_evaluate_h_columns_used(dom: *mut zkhip_domain, desc: *const zkhip_evalh_desc, used: *mut u8, capacity: u32, count: *mut u32) -> c_int;

    // ---- grand products and openings (the rest of plonk::prover::create_proof)
    pub fn zkhip_batch_invert_dev(ctx: *mut zkhip_ctx, field: c_int, d_a: *mut u64, n: usize) -> c_int;
    pub fn zkhip_prefix_product_dev(ctx: *mut zkhip_ctx, d_m: *const u64, d_out: *mut u64, n: usize, init: *const u64) -> c_int;
    pub fn zkhip_prefix_product_noalloc_dev(ctx: *mut zkhip_ctx, d_m: *const u64, d_out: *mut u64, n: usize, init: *const u64) -> c_int;
    pub fn zkhip_permutation_product_dev(dom: *mut zkhip_domain, d_values: *const *const u64, d_sigmas: *const *const u64, num_cols: u32,
        beta: *const u64, gamma: *const u64, delta_start: *const u64, z0: *const u64, d_z: *mut u64) -> c_int;
    pub fn zkhip_lookup_product_dev(dom: *mut zkhip_domain, d_compressed_input: *const u64, d_compressed_table: *const u64,
        d_permuted_input: *const u64, d_permuted_table: *const u64, beta: *const u64, gamma: *const u64, d_z: *mut u64) -> c_int;
    pub fn zkhip_lookup_product_batch_dev(dom: *mut zkhip_domain, d_compressed_inputs: *const *const u64, d_compressed_tables: *const *const u64,
        d_permuted_inputs: *const *const u64, d_permuted_tables: *const *const u64, beta: *const u64, gamma: *const u64,
        d_zs: *const *mut u64, count: u32) -> c_int;
    pub fn zkhip_eval_polynomial_dev(ctx: *mut zkhip_ctx, d_coeff: *const u64, n: usize, x: *const u64, out: *mut u64) -> c_int;
    pub fn zkhip_eval_polynomials_dev(ctx: *mut zkhip_ctx, d_coeffs: *const *const u64, points: *const u64, count: u32, n: usize,
        out: *mut u64) -> c_int;
    pub fn zkhip_lookup_permute_dev(dom: *mut zkhip_domain, d_compressed_input: *const u64, d_compressed_table: *const u64,
        usable_rows: u32, d_permuted_input: *mut u64, d_permuted_table: *mut u64) -> c_int;
    pub fn zkhip_lookup_permute_batch_dev(dom: *mut zkhip_domain, count: u32, d_compressed_inputs: *const *const u64,
        d_compressed_tables: *const *const u64, usable_rows: u32, d_permuted_inputs: *const *mut u64,
        d_permuted_tables: *const *mut u64) -> c_int;
    pub fn zkhip_g1_fft(ctx: *mut zkhip_ctx, affine_xy: *const u64, log_n: u32, omega: *const u64, scale: *const u64,
        out_xy: *mut u64) -> c_int;
    pub fn zkhip_permutation_sigma_dev(dom: *mut zkhip_domain, d_map_col: *const u32, d_map_row: *const u32, num_cols: u32,
        d_out: *mut u64) -> c_int;
    pub fn zkhip_poly_lincomb_dev(ctx: *mut zkhip_ctx, d_polys: *const *const u64, lens: *const usize, coeffs: *const u64, num: u32,
        n: usize, d_out: *mut u64) -> c_int;
    pub fn zkhip_kate_division_dev(ctx: *mut zkhip_ctx, d_poly: *const u64, n: usize, point: *const u64, d_out: *mut u64) -> c_int;

    // ---- coset subsets (multi-GPU evaluate_h)
    pub fn zkhip_coeff_to_cosets_dev(dom: *mut zkhip_domain, d_coeff: *const u64, cosets: *const u32, num_cosets: u32,
        d_out: *mut u64, batch: usize) -> c_int;
    pub fn zkhip_extended_to_cosets_dev(dom: *mut zkhip_domain, d_ext: *const u64, cosets: *const u32, num_cosets: u32,
        d_out: *mut u64) -> c_int;
    pub fn zkhip_cosets_to_extended_dev(dom: *mut zkhip_domain, d_compact: *const u64, cosets: *const u32, num_cosets: u32,
        d_ext: *mut u64) -> c_int;

    // ---- kernel timing hook
    pub fn zkhip_timing_enable(ctx: *mut zkhip_ctx, on: c_int) -> c_int;
    pub fn zkhip_timing_reset(ctx: *mut zkhip_ctx) -> c_int;
    pub fn zkhip_timing_get(ctx: *mut zkhip_ctx, family: *const c_char, avg_us: *mut c_double, launches: *mut u64) -> c_int;
}
