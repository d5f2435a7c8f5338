//! The shim the halo2_proofs fork calls at its three seams.  Everything here works on plain limbs -- `[u64; 4]` field
//! elements in Montgomery form, `[u64; 8]` affine points, `[u64; 12]` Jacobian points -- so this crate does not depend on
//! halo2curves; the fork converts at the call site (`Fr` is `#[repr(transparent)]`-like over `[u64; 4]`; the fork asserts
//! `size_of::<Fr>() == 32` and `size_of::<G1Affine>() == 64` once and casts slices, or copies if a later halo2curves
//! changes the layout).
//!
//! Switch: the reference selects its accelerator with `env_info: &mut Option<JitProverEnv>` on `create_proof`
//! (prover/src/test/proof_test.rs:297-306 of the reference; `None` = CPU).  The same convention here is process-wide:
//! `zkhip::enable(device)` once, then `zkhip::get()` returns `Some` and the seams below take the device path; without it
//! they return `None` and the caller runs upstream's rayon code unchanged.
//!
//! Seam 1, `halo2_proofs::arithmetic::best_multiexp`:
//! ```ignore
//! pub fn best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve {
//!     if let Some(hip) = zkhip::get() {
//!         if let Some(j) = hip.best_multiexp(as_limbs4(coeffs), as_limbs8(bases)) { return from_jacobian_limbs(j); }
//!     }
//!     best_multiexp_cpu(coeffs, bases)
//! }
//! ```
//! (`ParamsKZG::{setup, read}` call `hip.register_bases` for `g` and `g_lagrange`; `commit` / `commit_lagrange` pass
//! `&self.g[..len]` / `&self.g_lagrange[..len]`, which `bases_id` recognises by address.)
//!
//! Seam 2, `halo2_proofs::arithmetic::best_fft` and `poly::EvaluationDomain::{lagrange_to_coeff, coeff_to_lagrange,
//! coeff_to_extended, extended_to_coeff, divide_by_vanishing_poly}`: `hip.best_fft(a, omega, log_n)` and the methods of
//! `Domain`, which `EvaluationDomain::new(j, k)` creates next to its own constants.
//!
//! Seam 3, `plonk::evaluation::Evaluator::evaluate_h`: serialise every `GraphEvaluator` with `Graph::from_calculations`
//! once per proving key, fill `EvaluateH` from the method's own arguments, `run`.
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::c_int;
use std::ptr;
use std::sync::{Mutex, OnceLock};

pub use zkhip_sys as sys;
use zkhip_sys::*;

pub type Limbs4 = [u64; 4];
pub type Affine = [u64; 8];
pub type Jacobian = [u64; 12];

#[derive(Debug)]
pub struct Error {
    pub status: i32,
    pub message: String,
}

/// One device context.  The C ABI is not re-entrant per context, so every call takes the lock.
pub struct ZkHip {
    ctx: *mut zkhip_ctx,
    lock: Mutex<()>,
    bases: Mutex<HashMap<(usize, usize), u32>>, // (address, length) of a registered SRS slice -> id
}
// the context is only ever used under `lock`
unsafe impl Send for ZkHip {}
unsafe impl Sync for ZkHip {}

static INSTANCE: OnceLock<Option<ZkHip>> = OnceLock::new();

/// Turns the accelerator on for this process (the counterpart of passing `Some(JitProverEnv)`).
pub fn enable(device: i32) -> Result<(), Error> {
    let mut failure = None;
    INSTANCE.get_or_init(|| match ZkHip::new(device) {
        Ok(h) => Some(h),
        Err(e) => {
            failure = Some(e);
            None
        }
    });
    match failure {
        Some(e) => Err(e),
        None => Ok(()),
    }
}

/// `None`: take upstream's CPU path.
pub fn get() -> Option<&'static ZkHip> {
    INSTANCE.get().and_then(|o| o.as_ref())
}

impl ZkHip {
    pub fn new(device: i32) -> Result<Self, Error> {
        let mut ctx = ptr::null_mut();
        let rc = unsafe { zkhip_create(device as c_int, &mut ctx) };
        if rc != ZKHIP_OK {
            return Err(Error { status: rc, message: format!("zkhip_create({device}) failed") });
        }
        Ok(ZkHip { ctx, lock: Mutex::new(()), bases: Mutex::new(HashMap::new()) })
    }

    pub fn raw(&self) -> *mut zkhip_ctx {
        self.ctx
    }

    /// Maps a status to `Err` with the library's message; never panics, so callers decide (the infallible seams of
    /// upstream panic on misuse: `hip.check(rc).unwrap_or_else(|e| panic!("{}", e.message))`).
    pub fn check(&self, rc: c_int) -> Result<(), Error> {
        if rc == ZKHIP_OK {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(zkhip_last_error(self.ctx)) }.to_string_lossy().into_owned();
        Err(Error { status: rc, message: msg })
    }

    /// `ParamsKZG::{setup, read}`: uploads an SRS slice once and builds its window tables.
    pub fn register_bases(&self, bases: &[Affine]) -> Result<u32, Error> {
        let _g = self.lock.lock().unwrap();
        let mut id = 0u32;
        self.check(unsafe { zkhip_register_bases(self.ctx, bases.as_ptr() as *const u64, bases.len(), &mut id) })?;
        self.bases.lock().unwrap().insert((bases.as_ptr() as usize, bases.len()), id);
        Ok(id)
    }

    pub fn release_bases(&self, id: u32) -> Result<(), Error> {
        let _g = self.lock.lock().unwrap();
        self.bases.lock().unwrap().retain(|_, v| *v != id);
        self.check(unsafe { zkhip_release_bases(self.ctx, id) })
    }

    /// The registered set that `bases` is a prefix of (same address, at most its length).
    pub fn bases_id(&self, bases: &[Affine]) -> Option<u32> {
        let at = bases.as_ptr() as usize;
        self.bases.lock().unwrap().iter().find(|((a, n), _)| *a == at && bases.len() <= *n).map(|(_, id)| *id)
    }

    /// Seam 1.  `None` when the bases are not a registered SRS prefix (the caller falls back to the CPU path).
    pub fn best_multiexp(&self, coeffs: &[Limbs4], bases: &[Affine]) -> Option<Jacobian> {
        assert_eq!(coeffs.len(), bases.len()); // upstream asserts the same
        let id = self.bases_id(bases)?;
        let _g = self.lock.lock().unwrap();
        let mut out = [0u64; 12];
        let rc = unsafe { zkhip_msm(self.ctx, id, coeffs.as_ptr() as *const u64, coeffs.len(), out.as_mut_ptr()) };
        self.check(rc).unwrap_or_else(|e| panic!("zkhip_msm: {}", e.message));
        Some(out)
    }

    /// Seam 2a: in place, natural order in and out, `omega` of order exactly 2^log_n.
    pub fn best_fft(&self, a: &mut [Limbs4], omega: &Limbs4, log_n: u32) {
        assert_eq!(a.len(), 1usize << log_n);
        let _g = self.lock.lock().unwrap();
        let rc = unsafe { zkhip_ntt(self.ctx, a.as_mut_ptr() as *mut u64, log_n, omega.as_ptr()) };
        self.check(rc).unwrap_or_else(|e| panic!("zkhip_ntt: {}", e.message));
    }

    /// `EvaluationDomain::new(j, k)`.
    pub fn domain(&'static self, j: u32, k: u32) -> Result<Domain, Error> {
        let _g = self.lock.lock().unwrap();
        let mut dom = ptr::null_mut();
        self.check(unsafe { zkhip_domain_create(self.ctx, j, k, &mut dom) })?;
        Ok(Domain { hip: self, dom })
    }
}

impl Drop for ZkHip {
    fn drop(&mut self) {
        unsafe { zkhip_destroy(self.ctx) }
    }
}

/// Seam 2b: `poly::EvaluationDomain<Fr>` (host vectors in, host vectors out, as upstream's methods).
pub struct Domain {
    hip: &'static ZkHip,
    dom: *mut zkhip_domain,
}
unsafe impl Send for Domain {}
unsafe impl Sync for Domain {}

impl Domain {
    pub fn raw(&self) -> *mut zkhip_domain {
        self.dom
    }
    pub fn k(&self) -> u32 {
        unsafe { zkhip_domain_k(self.dom) }
    }
    pub fn extended_k(&self) -> u32 {
        unsafe { zkhip_domain_extended_k(self.dom) }
    }
    pub fn quotient_poly_degree(&self) -> u32 {
        unsafe { zkhip_domain_quotient_poly_degree(self.dom) }
    }
    fn ok(&self, what: &str, rc: c_int) {
        self.hip.check(rc).unwrap_or_else(|e| panic!("{what}: {}", e.message));
    }
    pub fn lagrange_to_coeff(&self, a: &mut [Limbs4]) {
        assert_eq!(a.len(), 1usize << self.k());
        let _g = self.hip.lock.lock().unwrap();
        self.ok("lagrange_to_coeff", unsafe { zkhip_lagrange_to_coeff(self.dom, a.as_mut_ptr() as *mut u64) });
    }
    pub fn coeff_to_lagrange(&self, a: &mut [Limbs4]) {
        assert_eq!(a.len(), 1usize << self.k());
        let _g = self.hip.lock.lock().unwrap();
        self.ok("coeff_to_lagrange", unsafe { zkhip_coeff_to_lagrange(self.dom, a.as_mut_ptr() as *mut u64) });
    }
    pub fn coeff_to_extended(&self, coeff: &[Limbs4]) -> Vec<Limbs4> {
        assert_eq!(coeff.len(), 1usize << self.k());
        let mut out = vec![[0u64; 4]; 1usize << self.extended_k()];
        let _g = self.hip.lock.lock().unwrap();
        self.ok("coeff_to_extended", unsafe {
            zkhip_coeff_to_extended(self.dom, coeff.as_ptr() as *const u64, out.as_mut_ptr() as *mut u64)
        });
        out
    }
    pub fn extended_to_coeff(&self, ext: &[Limbs4]) -> Vec<Limbs4> {
        assert_eq!(ext.len(), 1usize << self.extended_k());
        let mut out = vec![[0u64; 4]; (self.quotient_poly_degree() as usize) << self.k()];
        let _g = self.hip.lock.lock().unwrap();
        self.ok("extended_to_coeff", unsafe {
            zkhip_extended_to_coeff(self.dom, ext.as_ptr() as *const u64, out.as_mut_ptr() as *mut u64)
        });
        out
    }
    pub fn divide_by_vanishing_poly(&self, ext: &mut [Limbs4]) {
        assert_eq!(ext.len(), 1usize << self.extended_k());
        let _g = self.hip.lock.lock().unwrap();
        self.ok("divide_by_vanishing_poly", unsafe { zkhip_divide_by_vanishing_poly(self.dom, ext.as_mut_ptr() as *mut u64) });
    }
}

impl Drop for Domain {
    fn drop(&mut self) {
        unsafe { zkhip_domain_destroy(self.dom) }
    }
}

/// `plonk::evaluation::ValueSource`, with the (kind, a, b) encoding of the header.
#[derive(Clone, Copy, Debug)]
pub enum ValueSource {
    Constant(u32),
    Intermediate(u32),
    Fixed(u32, u32),
    Advice(u32, u32),
    Instance(u32, u32),
    Challenge(u32),
    Beta,
    Gamma,
    Theta,
    Y,
    PreviousValue,
}

impl ValueSource {
    fn words(self) -> [u32; 3] {
        match self {
            ValueSource::Constant(a) => [0, a, 0],
            ValueSource::Intermediate(a) => [1, a, 0],
            ValueSource::Fixed(a, b) => [2, a, b],
            ValueSource::Advice(a, b) => [3, a, b],
            ValueSource::Instance(a, b) => [4, a, b],
            ValueSource::Challenge(a) => [5, a, 0],
            ValueSource::Beta => [6, 0, 0],
            ValueSource::Gamma => [7, 0, 0],
            ValueSource::Theta => [8, 0, 0],
            ValueSource::Y => [9, 0, 0],
            ValueSource::PreviousValue => [10, 0, 0],
        }
    }
}

/// `plonk::evaluation::Calculation`.
#[derive(Clone, Debug)]
pub enum Calculation {
    Add(ValueSource, ValueSource),
    Sub(ValueSource, ValueSource),
    Mul(ValueSource, ValueSource),
    Square(ValueSource),
    Double(ValueSource),
    Negate(ValueSource),
    Horner(ValueSource, Vec<ValueSource>, ValueSource),
    Store(ValueSource),
}

/// A serialised `GraphEvaluator`; calculation i writes intermediate i, as `CalculationInfo::target` does upstream.
pub struct Graph {
    pub constants: Vec<Limbs4>,
    pub rotations: Vec<i32>,
    pub calc_words: Vec<u32>,
    pub num_calculations: u32,
}

impl Graph {
    pub fn from_calculations(constants: Vec<Limbs4>, rotations: Vec<i32>, calculations: &[Calculation]) -> Self {
        let mut w = Vec::new();
        for c in calculations {
            match c {
                Calculation::Add(a, b) => { w.push(0); w.extend(a.words()); w.extend(b.words()); }
                Calculation::Sub(a, b) => { w.push(1); w.extend(a.words()); w.extend(b.words()); }
                Calculation::Mul(a, b) => { w.push(2); w.extend(a.words()); w.extend(b.words()); }
                Calculation::Square(a) => { w.push(3); w.extend(a.words()); }
                Calculation::Double(a) => { w.push(4); w.extend(a.words()); }
                Calculation::Negate(a) => { w.push(5); w.extend(a.words()); }
                Calculation::Horner(start, parts, factor) => {
                    w.push(6);
                    w.extend(start.words());
                    w.extend(factor.words());
                    w.push(parts.len() as u32);
                    for p in parts { w.extend(p.words()); }
                }
                Calculation::Store(a) => { w.push(7); w.extend(a.words()); }
            }
        }
        Graph { constants, rotations, calc_words: w, num_calculations: calculations.len() as u32 }
    }

    fn raw(&self) -> zkhip_graph {
        zkhip_graph {
            constants: self.constants.as_ptr() as *const u64,
            num_constants: self.constants.len() as u32,
            rotations: self.rotations.as_ptr(),
            num_rotations: self.rotations.len() as u32,
            calc_words: self.calc_words.as_ptr(),
            num_calc_words: self.calc_words.len() as u32,
            num_calculations: self.num_calculations,
        }
    }
}

/// One lookup argument of `evaluate_h`: its graph and the three committed polynomials (coefficient form).
pub struct LookupArg<'a> {
    pub graph: &'a Graph,
    pub product: &'a [Limbs4],
    pub permuted_input: &'a [Limbs4],
    pub permuted_table: &'a [Limbs4],
}

/// Seam 3: the arguments of `Evaluator::evaluate_h(pk, advice_polys, instance_polys, challenges, y, beta, gamma, theta,
/// lookups, shuffles, permutations)` for one circuit, host memory, polynomials in coefficient form as upstream passes
/// them (`pk.fixed_cosets`, `pk.l0 / l_last / l_active_row`, `pk.permutation.cosets` and each set's
/// `permutation_product_coset` are extended vectors).  Shuffle arguments are not supported: a constraint system that
/// has any takes the CPU path.
pub struct EvaluateH<'a> {
    pub fixed_cosets: &'a [&'a [Limbs4]],
    pub advice_polys: &'a [&'a [Limbs4]],
    pub instance_polys: &'a [&'a [Limbs4]],
    pub challenges: &'a [Limbs4],
    pub y: Limbs4,
    pub beta: Limbs4,
    pub gamma: Limbs4,
    pub theta: Limbs4,
    pub l0: &'a [Limbs4],
    pub l_last: &'a [Limbs4],
    pub l_active_row: &'a [Limbs4],
    pub custom_gates: Option<&'a Graph>,
    pub lookups: &'a [LookupArg<'a>],
    /// permutation argument columns in order: (kind, index) with kind 2 Fixed, 3 Advice, 4 Instance
    pub perm_columns: &'a [(u32, u32)],
    pub perm_sigma_cosets: &'a [&'a [Limbs4]],
    pub perm_product_cosets: &'a [&'a [Limbs4]],
    pub perm_chunk_len: u32,   // cs.degree() - 2
    pub blinding_factors: u32, // cs.blinding_factors()
}

impl<'a> EvaluateH<'a> {
    fn desc(&self, keep: &mut Keep) -> zkhip_evalh_desc {
        let ptrs = |v: &[&[Limbs4]]| v.iter().map(|s| s.as_ptr() as *const u64).collect::<Vec<_>>();
        keep.fixed = ptrs(self.fixed_cosets);
        keep.advice = ptrs(self.advice_polys);
        keep.instance = ptrs(self.instance_polys);
        keep.sigma = ptrs(self.perm_sigma_cosets);
        keep.product = ptrs(self.perm_product_cosets);
        keep.kinds = self.perm_columns.iter().map(|c| c.0).collect();
        keep.indices = self.perm_columns.iter().map(|c| c.1).collect();
        keep.gates = self.custom_gates.map(|g| g.raw());
        keep.lookups = self
            .lookups
            .iter()
            .map(|l| zkhip_lookup {
                graph: l.graph.raw(),
                product: l.product.as_ptr() as *const u64,
                permuted_input: l.permuted_input.as_ptr() as *const u64,
                permuted_table: l.permuted_table.as_ptr() as *const u64,
            })
            .collect();
        zkhip_evalh_desc {
            mem: ZKHIP_MEM_HOST,
            poly_form: ZKHIP_FORM_COEFF,
            num_fixed: keep.fixed.len() as u32,
            num_advice: keep.advice.len() as u32,
            num_instance: keep.instance.len() as u32,
            num_challenges: self.challenges.len() as u32,
            fixed_cosets: keep.fixed.as_ptr(),
            advice: keep.advice.as_ptr(),
            instance: keep.instance.as_ptr(),
            challenges: self.challenges.as_ptr() as *const u64,
            y: self.y,
            beta: self.beta,
            gamma: self.gamma,
            theta: self.theta,
            l0: self.l0.as_ptr() as *const u64,
            l_last: self.l_last.as_ptr() as *const u64,
            l_active_row: self.l_active_row.as_ptr() as *const u64,
            custom_gates: keep.gates.as_ref().map_or(ptr::null(), |g| g as *const zkhip_graph),
            num_lookups: keep.lookups.len() as u32,
            lookups: keep.lookups.as_ptr(),
            num_perm_columns: keep.kinds.len() as u32,
            perm_column_kind: keep.kinds.as_ptr(),
            perm_column_index: keep.indices.as_ptr(),
            perm_sigma_cosets: keep.sigma.as_ptr(),
            num_perm_sets: keep.product.len() as u32,
            perm_product_cosets: keep.product.as_ptr(),
            perm_chunk_len: self.perm_chunk_len,
            blinding_factors: self.blinding_factors,
            num_cosets: 0,
            cosets: ptr::null(),
            basis: ZKHIP_BASIS_EXTENDED,
            row_first: 0,
            row_count: 0,
            term_select: ZKHIP_TERMS_ALL,
            term_degree_max: 0,
        }
    }

    /// `Evaluator::new` at key generation (`ProvingKey::ev`): builds and caches the fused device program of this shape.
    pub fn prepare(&self, dom: &Domain) -> Result<(), Error> {
        let mut keep = Keep::default();
        let desc = self.desc(&mut keep);
        let _g = dom.hip.lock.lock().unwrap();
        dom.hip.check(unsafe { zkhip_evaluate_h_prepare(dom.dom, &desc) })
    }

    /// The numerator of the quotient on the extended domain, 2^extended_k values.
    pub fn run(&self, dom: &Domain) -> Result<Vec<Limbs4>, Error> {
        let mut keep = Keep::default();
        let desc = self.desc(&mut keep);
        let mut out = vec![[0u64; 4]; 1usize << dom.extended_k()];
        let _g = dom.hip.lock.lock().unwrap();
        dom.hip.check(unsafe { zkhip_evaluate_h(dom.dom, &desc, out.as_mut_ptr() as *mut u64) })?;
        Ok(out)
    }
}

/// pointer tables that must outlive the descriptor
#[derive(Default)]
struct Keep {
    fixed: Vec<*const u64>,
    advice: Vec<*const u64>,
    instance: Vec<*const u64>,
    sigma: Vec<*const u64>,
    product: Vec<*const u64>,
    kinds: Vec<u32>,
    indices: Vec<u32>,
    gates: Option<zkhip_graph>,
    lookups: Vec<zkhip_lookup>,
}
