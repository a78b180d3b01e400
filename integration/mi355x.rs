// mi355x.rs -- the reference-side glue for proxima-one/kzg (crate kzg 0.8.0-beta.1), cargo feature `mi355x`:
// safe wrappers over integration/mi355x_sys.rs and the FIVE splices that route the crate's hot call sites through
// libkzg_mi355x.so.  New code only -- nothing here is copied from the crate; every function names the method body it replaces
// (file:line in the reference tree).  The public surface of the crate (KZGProver, KZGProverEvalForm, EvaluationDomain,
// KZGParams, setup) stays as it is: each splice is the new body of an existing method.
//
// Layout assumptions (blstrs::Scalar = blst_fr, G1Affine = blst_p1_affine, G1Projective = blst_p1) are pinned by the
// `mi355x_pin` test of INTEGRATION.md section 4; a binding that prefers not to rely on them passes the canonical formats
// (KZG_FR_CANONICAL_LE_32, KZG_G1_ZCASH_COMPRESSED_48) instead.
//
// Not compiled in the engine's own image (no Rust toolchain there); tests/test_rust_shim.py checks every extern "C" signature
// used here against include/kzg_mi355x.h.
#![cfg(feature = "mi355x")]
use crate::mi355x_sys as sys;
use crate::mi355x_sys::{KZG_FR_MONT_LE_32, KZG_G1_AFFINE_MONT_96, KZG_G1_JACOBIAN_MONT_144};
use crate::{KZGError, KZGParams};
use blstrs::{G1Affine, Scalar};
use std::ffi::CStr;
use std::os::raw::c_void;

/// One engine per GPU for the whole process.  `Send + Sync`: the blocking prover calls run concurrently on one context (each
/// leases a lane inside the library), so a `KZGProver` cloned into many threads keeps the GPU as full as a batch call does.
pub struct Mi355x {
    ctx: *mut sys::kzg_ctx,
}
unsafe impl Send for Mi355x {}
unsafe impl Sync for Mi355x {}

impl Mi355x {
    /// Call once at start-up, BEFORE anything in the process touches HIP: asks the runtime for the hardware queues of the
    /// pipelined paths (it sets GPU_MAX_HW_QUEUES unless the host exported a value).
    pub fn init_hw_queues() {
        unsafe { sys::kzg_init_hw_queues(0) };
    }
    pub fn new(device: i32) -> Result<Self, i32> {
        let mut ctx = std::ptr::null_mut();
        match unsafe { sys::kzg_ctx_create(device, &mut ctx) } {
            0 => Ok(Mi355x { ctx }),
            e => Err(e), // KZG_ERR_NO_DEVICE: there is no CPU fallback
        }
    }
    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(sys::kzg_last_error(self.ctx)) }.to_string_lossy().into_owned()
    }
    fn fail(&self, rc: i32) -> ! {
        // status 3 = a condition on which the reference panics (slice index out of range, failed assert!, ...)
        if rc == sys::KZG_ERR_SHAPE {
            panic!("{}", self.last_error())
        }
        panic!("kzg_mi355x error {}: {}", rc, self.last_error())
    }
}
impl Drop for Mi355x {
    fn drop(&mut self) {
        unsafe { sys::kzg_ctx_destroy(self.ctx) }
    }
}

/// `KZGParams.gs` (or a Lagrange basis) resident in HBM, uploaded once.  Immutable: usable from every thread and context.
pub struct DeviceSrs {
    srs: *mut sys::kzg_srs,
    ctx: *mut sys::kzg_ctx,
}
unsafe impl Send for DeviceSrs {}
unsafe impl Sync for DeviceSrs {}
impl Drop for DeviceSrs {
    fn drop(&mut self) {
        unsafe { sys::kzg_srs_free(self.ctx, self.srs) }
    }
}

impl KZGParams {
    /// `gs: Vec<G1Projective>` is 144-byte Jacobian Montgomery memory (= blst_p1): uploaded as it lies.
    pub fn to_device(&self, eng: &Mi355x) -> DeviceSrs {
        let mut srs = std::ptr::null_mut();
        let rc = unsafe {
            sys::kzg_srs_upload_g1(eng.ctx, self.gs.as_ptr() as *const c_void, self.gs.len(), KZG_G1_JACOBIAN_MONT_144, &mut srs)
        };
        if rc != 0 {
            eng.fail(rc)
        }
        DeviceSrs { srs, ctx: eng.ctx }
    }
}

/// Lagrange basis `lagrange_basis_g: Vec<G1Affine>` of KZGProverEvalForm (96-byte affine Montgomery = blst_p1_affine).
pub fn lagrange_to_device(eng: &Mi355x, basis: &[G1Affine]) -> DeviceSrs {
    let mut srs = std::ptr::null_mut();
    let rc = unsafe { sys::kzg_srs_upload_g1(eng.ctx, basis.as_ptr() as *const c_void, basis.len(), KZG_G1_AFFINE_MONT_96, &mut srs) };
    if rc != 0 {
        eng.fail(rc)
    }
    DeviceSrs { srs, ctx: eng.ctx }
}

// ---- splice 1: KZGProver::commit, src/coeff_form.rs:59-64 ------------------------------------------------------------------
// replaces  G1Projective::multi_exp(&self.parameters.gs[..n], polynomial.slice_coeffs()).to_affine()
pub fn commit(eng: &Mi355x, gs: &DeviceSrs, coeffs: &[Scalar]) -> G1Affine {
    let mut out = G1Affine::identity();
    let rc = unsafe {
        sys::kzg_commit_coeff(eng.ctx, gs.srs, coeffs.as_ptr() as *const c_void, coeffs.len(), KZG_FR_MONT_LE_32, 0,
                              &mut out as *mut G1Affine as *mut c_void, KZG_G1_AFFINE_MONT_96)
    };
    if rc != 0 {
        eng.fail(rc)
    }
    out
}

// ---- splice 2: KZGProver::create_witness, src/coeff_form.rs:66-81 ----------------------------------------------------------
// replaces  the clone, long_division by (X - x) and multi_exp of the quotient; status 1 is the method's own error
pub fn create_witness(eng: &Mi355x, gs: &DeviceSrs, coeffs: &[Scalar], x: &Scalar, y: &Scalar) -> Result<G1Affine, KZGError> {
    let mut out = G1Affine::identity();
    let rc = unsafe {
        sys::kzg_witness_coeff(eng.ctx, gs.srs, coeffs.as_ptr() as *const c_void, coeffs.len(), x as *const Scalar as *const c_void,
                               y as *const Scalar as *const c_void, KZG_FR_MONT_LE_32, 0, &mut out as *mut G1Affine as *mut c_void,
                               KZG_G1_AFFINE_MONT_96)
    };
    match rc {
        0 => Ok(out),
        1 => Err(KZGError::PointNotOnPolynomial),
        e => eng.fail(e),
    }
}

// ---- splice 3: KZGProver::create_witness_batched, src/coeff_form.rs:83-111 --------------------------------------------------
// replaces  SubProductTree::new_from_points, linear_interpolation, long_division by Z, multi_exp; returns (w, coefficients of r)
// -- the caller wraps them as KZGBatchWitness { r: Polynomial::new_from_coeffs(r, r.len() - 1), w }
pub fn create_witness_batched(eng: &Mi355x, gs: &DeviceSrs, coeffs: &[Scalar], xs: &[Scalar], ys: &[Scalar])
                              -> Result<(G1Affine, Vec<Scalar>), KZGError> {
    assert_eq!(xs.len(), ys.len());
    let mut w = G1Affine::identity();
    let mut r = vec![Scalar::from(0u64); xs.len().max(2)];
    let mut r_len = 0usize;
    let rc = unsafe {
        sys::kzg_witness_coeff_batched(eng.ctx, gs.srs, coeffs.as_ptr() as *const c_void, coeffs.len(), xs.as_ptr() as *const c_void,
                                       ys.as_ptr() as *const c_void, xs.len(), KZG_FR_MONT_LE_32, 0,
                                       &mut w as *mut G1Affine as *mut c_void, KZG_G1_AFFINE_MONT_96, r.as_mut_ptr() as *mut c_void,
                                       &mut r_len)
    };
    match rc {
        0 => {
            r.truncate(r_len);
            Ok((w, r))
        }
        1 => Err(KZGError::PointNotOnPolynomial),
        e => eng.fail(e),
    }
}

// ---- splice 4: KZGProverEvalForm::commit / create_witness, src/eval_form.rs:114-140 -----------------------------------------
// replaces  multi_exp over lagrange_basis_g, and div_by_omega_i + multi_exp
pub fn commit_eval(eng: &Mi355x, lagrange: &DeviceSrs, evals: &[Scalar]) -> G1Affine {
    let mut out = G1Affine::identity();
    let rc = unsafe {
        sys::kzg_commit_eval(eng.ctx, lagrange.srs, evals.as_ptr() as *const c_void, evals.len(), KZG_FR_MONT_LE_32, 0,
                             &mut out as *mut G1Affine as *mut c_void, KZG_G1_AFFINE_MONT_96)
    };
    if rc != 0 {
        eng.fail(rc) // status 3: assert!(self.d == evals.d)
    }
    out
}
pub fn create_witness_eval(eng: &Mi355x, lagrange: &DeviceSrs, evals: &[Scalar], i: usize) -> G1Affine {
    let mut out = G1Affine::identity();
    let rc = unsafe {
        sys::kzg_witness_eval(eng.ctx, lagrange.srs, evals.as_ptr() as *const c_void, evals.len(), i, KZG_FR_MONT_LE_32, 0,
                              &mut out as *mut G1Affine as *mut c_void, KZG_G1_AFFINE_MONT_96)
    };
    if rc != 0 {
        eng.fail(rc)
    }
    out
}

// ---- splice 5: EvaluationDomain::fft / ifft, src/ft.rs:111-140 --------------------------------------------------------------
// replaces  best_fft(&mut self.coeffs, &self.omega, self.exp)  (and the minv scaling of ifft); in place, natural order
pub fn fft_in_place(eng: &Mi355x, coeffs: &mut [Scalar], exp: u32, inverse: bool) {
    assert_eq!(coeffs.len(), 1usize << exp);
    let rc = unsafe { sys::kzg_ntt_fr(eng.ctx, coeffs.as_mut_ptr() as *mut c_void, exp, inverse as i32, 0) };
    if rc != 0 {
        eng.fail(rc)
    }
}

// ---- multi-GPU: the same prover with KZGParams.gs sharded over every GPU of the node -------------
/// What a ONE-NODE host exports before the first RCCL call of the process (values already exported are kept): RCCL bootstraps
/// every communicator over TCP on the first non-loopback interface it finds; on a host whose interface swallows packets that
/// stalls formation for minutes.  The library bounds formation ("comm_timeout_ms", KZG_COMM_TIMEOUT_MS) but cannot choose the
/// interface for RCCL -- the environment is the host's.
pub fn single_node_rccl_env() {
    for (k, v) in [("NCCL_SOCKET_IFNAME", "lo"), ("NCCL_RAS_ENABLE", "0"), ("NCCL_IB_DISABLE", "1"), ("NCCL_NET_PLUGIN", "none")] {
        if std::env::var_os(k).is_none() {
            std::env::set_var(k, v);
        }
    }
}

/// An SRS (monomial `gs` or a Lagrange basis) sharded contiguously over the group's GPUs.  It BORROWS the group that made it:
/// kzg_msrs_free reads the group's state (its devices, contexts and whether it is dead), so the borrow checker -- not a comment --
/// keeps the group alive until every one of its SRSs has been dropped (with a raw pointer, safe code could drop or move
/// the group first).
pub struct ShardedSrs<'g> {
    s: *mut sys::kzg_msrs,
    group: &'g Mi355xGroup,
}
unsafe impl<'g> Send for ShardedSrs<'g> {}
unsafe impl<'g> Sync for ShardedSrs<'g> {}
impl<'g> ShardedSrs<'g> {
    pub fn len(&self) -> usize {
        unsafe { sys::kzg_msrs_len(self.s) }
    }
}
impl<'g> Drop for ShardedSrs<'g> {
    fn drop(&mut self) {
        unsafe { sys::kzg_msrs_free(self.group.m, self.s) }
    }
}

pub struct Mi355xGroup {
    m: *mut sys::kzg_mctx,
}
unsafe impl Send for Mi355xGroup {}
unsafe impl Sync for Mi355xGroup {}
impl Mi355xGroup {
    fn create_error() -> String {
        unsafe { CStr::from_ptr(sys::kzg_mctx_create_error()) }.to_string_lossy().into_owned()
    }
    /// One host process driving every GPU of the node (ncclCommInitAll inside, bounded by KZG_COMM_TIMEOUT_MS).
    pub fn all_gpus() -> Result<Self, String> {
        single_node_rccl_env();
        let n = unsafe { sys::kzg_device_count() };
        let devs: Vec<i32> = (0..n).collect();
        let mut m = std::ptr::null_mut();
        match unsafe { sys::kzg_mctx_create(devs.as_ptr(), n, &mut m) } {
            0 => Ok(Mi355xGroup { m }),
            e => Err(format!("kzg_mctx_create failed with {}: {}", e, Self::create_error())),
        }
    }
    /// One process per GPU: rank 0 draws the id, the host carries the 128 bytes to the other ranks by its own means.
    pub fn unique_id(single_node: bool) -> Result<[u8; 128], String> {
        if single_node {
            single_node_rccl_env();
        }
        let mut id = [0u8; 128];
        match unsafe { sys::kzg_mctx_unique_id(id.as_mut_ptr() as *mut c_void) } {
            0 => Ok(id),
            e => Err(format!("kzg_mctx_unique_id failed with {}: {}", e, Self::create_error())),
        }
    }
    /// `single_node`: the whole world lives on this node (the xGMI case) -- the loopback bootstrap knobs of single_node_rccl_env()
    /// are applied; a world that spans nodes passes false and brings its own NCCL_* environment (with the loopback knobs it could
    /// not form its communicator at all).
    pub fn for_rank(device: i32, rank: i32, world: i32, id: &[u8; 128], single_node: bool) -> Result<Self, String> {
        if single_node {
            single_node_rccl_env();
        }
        let mut m = std::ptr::null_mut();
        match unsafe { sys::kzg_mctx_create_rank(device, rank, world, id.as_ptr() as *const c_void, &mut m) } {
            0 => Ok(Mi355xGroup { m }),
            e => Err(format!("kzg_mctx_create_rank failed with {}: {}", e, Self::create_error())),
        }
    }
    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(sys::kzg_mctx_last_error(self.m)) }.to_string_lossy().into_owned()
    }
    fn fail(&self, rc: i32) -> ! {
        // every rank returns the same status (mgpu.hip); status 3 = a condition on which the reference panics
        panic!("kzg_mi355x group error {}: {}", rc, self.last_error())
    }
    /// "rccl=... formation_ms=... phases_ms=[...]": which RCCL the group runs on and what forming the communicator cost.
    pub fn info(&self) -> String {
        let mut buf = vec![0u8; 2048];
        let rc = unsafe { sys::kzg_mctx_info(self.m, buf.as_mut_ptr() as *mut std::os::raw::c_char, buf.len()) };
        if rc != 0 {
            self.fail(rc)
        }
        unsafe { CStr::from_ptr(buf.as_ptr() as *const std::os::raw::c_char) }.to_string_lossy().into_owned()
    }
    /// `KZGParams.gs` (or `lagrange_basis_g`) sharded over the group: rank r uploads its contiguous range of the caller's vector.
    pub fn upload<'g>(&'g self, gs: &[blstrs::G1Projective]) -> ShardedSrs<'g> {
        let mut s = std::ptr::null_mut();
        let rc = unsafe { sys::kzg_srs_upload_g1_sharded(self.m, gs.as_ptr() as *const c_void, gs.len(), KZG_G1_JACOBIAN_MONT_144, &mut s) };
        if rc != 0 {
            self.fail(rc)
        }
        ShardedSrs { s, group: self }
    }
    /// KZGProver::commit over the group (src/coeff_form.rs:59-64).
    pub fn commit(&self, srs: &ShardedSrs, coeffs: &[Scalar]) -> G1Affine {
        let mut out = G1Affine::identity();
        let rc = unsafe {
            sys::kzg_commit_coeff_sharded(self.m, srs.s, coeffs.as_ptr() as *const c_void, coeffs.len(), KZG_FR_MONT_LE_32, 0,
                                          &mut out as *mut G1Affine as *mut c_void, KZG_G1_AFFINE_MONT_96)
        };
        if rc != 0 {
            self.fail(rc)
        }
        out
    }
    /// `polys.len() / n` polynomials of n coefficients each (contiguous), one exchange for all of them: the throughput form of
    /// KZGProver::commit (src/coeff_form.rs:59-64) -- commitments of a whole batch of blobs in one call.
    pub fn commit_batch(&self, srs: &ShardedSrs, polys: &[Scalar], n: usize) -> Vec<G1Affine> {
        assert!(n > 0 && polys.len() % n == 0);
        let batch = polys.len() / n;
        let mut out = vec![G1Affine::identity(); batch];
        let rc = unsafe {
            sys::kzg_commit_coeff_sharded_batch(self.m, srs.s, polys.as_ptr() as *const c_void, n, batch, KZG_FR_MONT_LE_32, 0,
                                                out.as_mut_ptr() as *mut c_void, KZG_G1_AFFINE_MONT_96)
        };
        if rc != 0 {
            self.fail(rc)
        }
        out
    }
    /// KZGProver::create_witness over the group (src/coeff_form.rs:66-81): replicated quotient scan, sharded MSM.
    pub fn create_witness(&self, srs: &ShardedSrs, coeffs: &[Scalar], x: &Scalar, y: &Scalar) -> Result<G1Affine, KZGError> {
        let mut out = G1Affine::identity();
        let rc = unsafe {
            sys::kzg_witness_coeff_sharded(self.m, srs.s, coeffs.as_ptr() as *const c_void, coeffs.len(), x as *const Scalar as *const c_void,
                                           y as *const Scalar as *const c_void, KZG_FR_MONT_LE_32, 0,
                                           &mut out as *mut G1Affine as *mut c_void, KZG_G1_AFFINE_MONT_96)
        };
        match rc {
            0 => Ok(out),
            1 => Err(KZGError::PointNotOnPolynomial),
            e => self.fail(e),
        }
    }
    /// KZGProver::create_witness_batched over the group (src/coeff_form.rs:83-111): interpolant and quotient replicated on every
    /// GPU, the quotient's MSM sharded; returns (w, coefficients of r) like the single-GPU splice 3.
    pub fn create_witness_batched(&self, srs: &ShardedSrs, coeffs: &[Scalar], xs: &[Scalar], ys: &[Scalar])
                                  -> Result<(G1Affine, Vec<Scalar>), KZGError> {
        assert_eq!(xs.len(), ys.len());
        let mut w = G1Affine::identity();
        let mut r = vec![Scalar::from(0u64); xs.len().max(2)];
        let mut r_len = 0usize;
        let rc = unsafe {
            sys::kzg_witness_coeff_batched_sharded(self.m, srs.s, coeffs.as_ptr() as *const c_void, coeffs.len(),
                                                   xs.as_ptr() as *const c_void, ys.as_ptr() as *const c_void, xs.len(), KZG_FR_MONT_LE_32, 0,
                                                   &mut w as *mut G1Affine as *mut c_void, KZG_G1_AFFINE_MONT_96,
                                                   r.as_mut_ptr() as *mut c_void, &mut r_len)
        };
        match rc {
            0 => {
                r.truncate(r_len);
                Ok((w, r))
            }
            1 => Err(KZGError::PointNotOnPolynomial),
            e => self.fail(e),
        }
    }
    /// KZGProverEvalForm::create_witness over the group (src/eval_form.rs:124-140): `lagrange` = upload(&lagrange_basis_g);
    /// div_by_omega_i replicated, MSM sharded.
    pub fn create_witness_eval(&self, lagrange: &ShardedSrs, evals: &[Scalar], i: usize) -> G1Affine {
        let mut out = G1Affine::identity();
        let rc = unsafe {
            sys::kzg_witness_eval_sharded(self.m, lagrange.s, evals.as_ptr() as *const c_void, evals.len(), i, KZG_FR_MONT_LE_32, 0,
                                          &mut out as *mut G1Affine as *mut c_void, KZG_G1_AFFINE_MONT_96)
        };
        if rc != 0 {
            self.fail(rc) // status 3: index out of range / d not a power of two (the reference's panics)
        }
        out
    }
    /// e.g. ("gather_timeout_ms", 60000), ("comm_timeout_ms", 60000), ("always_gather", 1), or any kzg_ctx_set_option key
    pub fn set_option(&self, key: &str, value: i64) {
        let k = std::ffi::CString::new(key).unwrap();
        let rc = unsafe { sys::kzg_mctx_set_option(self.m, k.as_ptr(), value) };
        if rc != 0 {
            self.fail(rc)
        }
    }
}
impl Drop for Mi355xGroup {
    fn drop(&mut self) {
        unsafe { sys::kzg_mctx_destroy(self.m) } // bounded: a dead group's stuck stream is left behind, not waited for
    }
}
