//! The solver traits of `optimization-solvers` (src/ls_solver.rs:3-112, src/line_search/mod.rs:14-23) implemented on
//! libqn_hip.so -- hand-written gfx950 kernels behind the C ABI of include/qn_hip.h.
//!
//! NOT COMPILED in the build image (no Rust toolchain; see ../Cargo.toml).  The FFI declarations are checked against the
//! header mechanically; every call sequence below is exercised on the GPU from Python (tests/test_gpu_trait_hooks.py,
//! tests/test_gpu_parity.py) and C (examples/ffi_consumer.c).
//!
//! Two ways in, both with the reference's names and signatures:
//!
//! * **the whole loop on the device** -- `GpuBFGS::minimize_on_device(&mut ls, oracle, max_iter_solver, max_iter_line_search,
//!   callback)` has the argument list of `LineSearchSolver::minimize` (ls_solver.rs:66-111) and takes a GPU line search.  One
//!   `qn_minimize` call runs every iteration; the closure is called through a trampoline in exactly the reference's order.
//!   `minimize_objective` takes a device-resident objective instead (no host round trip at all: the benchmark path).
//!   (It is NOT called `minimize`: an inherent method of that name would shadow the trait's for every call site, including
//!   those that pass the reference's own `MoreThuente` / `BackTracking`, which are not GPU line searches -- a compile error
//!   where a fall-back was meant.  `solver.minimize(..)` therefore always means the trait method below.)
//! * **hook by hook** -- `impl ComputeDirection` / `impl LineSearchSolver` provide `compute_direction`, `has_converged`,
//!   `update_next_iterate`, `xk`, `k`, ... so generic code written against the traits (`fn run<S: LineSearchSolver>(..)`)
//!   drives the reference's own template loop with the matrix work (H g, the secant update) on the GPU, and
//!   `impl LineSearch for GpuMoreThuente / GpuBackTracking` runs the line search's state machine on the GPU.
pub mod ffi;

use ffi::*;
use nalgebra::{DMatrix, DVector};
use optimization_solvers::{
    ComputeDirection, CurvatureCondition, Floating, FuncEvalMultivariate, LineSearch, LineSearchSolver, MoreThuente, SolverError,
    SufficientDecreaseCondition,
};
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::sync::OnceLock;

// ------------------------------------------------------------------------------------------------
// status <-> SolverError (ls_solver.rs:10-20)
// ------------------------------------------------------------------------------------------------
fn status_to_result(code: c_int) -> Result<(), SolverError> {
    match code {
        QN_OK => Ok(()),
        QN_MAX_ITER_REACHED => Err(SolverError::MaxIterReached),
        QN_OUT_OF_DOMAIN => Err(SolverError::OutOfDomain),
        QN_ERROR_INPUT_PARAMS => Err(SolverError::ErrorInputParams),
        _ => Err(SolverError::AbnormalTermination),
    }
}

/// Thread-local detail of the last non-OK return.
pub fn last_error() -> String {
    unsafe { CStr::from_ptr(qn_last_error_message()).to_string_lossy().into_owned() }
}

// ------------------------------------------------------------------------------------------------
// context: one GPU, one stream.  There is no CPU fallback: without a usable MI355X this panics.
// ------------------------------------------------------------------------------------------------
struct CtxPtr(*mut qn_context);
unsafe impl Send for CtxPtr {}
unsafe impl Sync for CtxPtr {}
static DEFAULT_CTX: OnceLock<CtxPtr> = OnceLock::new();

/// The process-wide context on device 0 (created on first use, never destroyed).
pub fn default_context() -> *mut qn_context {
    DEFAULT_CTX
        .get_or_init(|| {
            let mut ctx = std::ptr::null_mut();
            let code = unsafe { qn_context_create(0, &mut ctx) };
            assert_eq!(code, QN_OK, "qn_context_create: {} (there is no CPU fallback)", last_error());
            CtxPtr(ctx)
        })
        .0
}

// ------------------------------------------------------------------------------------------------
// the oracle closure `impl FnMut(&DVector<Floating>) -> FuncEvalMultivariate` behind a `void* user`
// ------------------------------------------------------------------------------------------------
unsafe extern "C" fn oracle_trampoline<F>(user: *mut c_void, x: *const f64, n: usize, f: *mut f64, g: *mut f64) -> c_int
where
    F: FnMut(&DVector<Floating>) -> FuncEvalMultivariate,
{
    let oracle = &mut *(user as *mut F);
    let xv = DVector::from_column_slice(std::slice::from_raw_parts(x, n));
    let eval = oracle(&xv);
    if eval.g().len() != n {
        return 1; // aborts the run with QN_ABNORMAL_TERMINATION
    }
    *f = *eval.f();
    std::ptr::copy_nonoverlapping(eval.g().as_ptr(), g, n);
    0
}

fn host_oracle<F>(oracle: &mut F, memoize: bool) -> qn_oracle
where
    F: FnMut(&DVector<Floating>) -> FuncEvalMultivariate,
{
    qn_oracle {
        kind: QN_ORACLE_HOST,
        memoize: memoize as i32, // 0: the reference's call sequence, call for call
        host_fn: Some(oracle_trampoline::<F>),
        host_user: oracle as *mut F as *mut c_void,
        device_fn: None,
        device_user: std::ptr::null_mut(),
        objective: std::ptr::null_mut(),
        host_hessian_fn: None,
    }
}

// ------------------------------------------------------------------------------------------------
// line searches
// ------------------------------------------------------------------------------------------------
/// A line search the device state machine knows: hands its parameters over as the plain `qn_linesearch` struct.
pub trait GpuLineSearch: LineSearch {
    fn ffi(&mut self) -> &mut qn_linesearch;
}

/// `MoreThuente` (morethuente.rs:6-62) with the state machine of `compute_step_len` (:165-297) on the GPU.
#[derive(Clone, Debug)]
pub struct GpuMoreThuente {
    ls: qn_linesearch,
}

impl Default for GpuMoreThuente {
    fn default() -> Self {
        let mut ls = unsafe { std::mem::zeroed::<qn_linesearch>() };
        unsafe { qn_morethuente_default(&mut ls) }; // c1 1e-4, c2 0.9, t in [0, inf), deltas 0.58333333 / 0.66 / 1.1
        GpuMoreThuente { ls }
    }
}

impl From<&MoreThuente> for GpuMoreThuente {
    /// Same parameters as a reference line search (its fields are readable through derive_getters).
    fn from(m: &MoreThuente) -> Self {
        let mut out = GpuMoreThuente::default();
        out.ls.c1 = *m.c1();
        out.ls.c2 = *m.c2();
        out.ls.t_min = *m.t_min();
        out.ls.t_max = *m.t_max();
        out.ls.delta_min = *m.delta_min();
        out.ls.delta = *m.delta();
        out.ls.delta_max = *m.delta_max();
        out
    }
}

impl GpuMoreThuente {
    // the builder methods of morethuente.rs:31-62; the reference's assert!s stay panics
    pub fn with_deltas(mut self, delta_min: Floating, delta: Floating, delta_max: Floating) -> Self {
        assert_eq!(unsafe { qn_morethuente_with_deltas(&mut self.ls, delta_min, delta, delta_max) }, QN_OK, "{}", last_error());
        self
    }
    pub fn with_t_min(mut self, t_min: Floating) -> Self {
        assert_eq!(unsafe { qn_morethuente_with_t_min(&mut self.ls, t_min) }, QN_OK, "{}", last_error());
        self
    }
    pub fn with_t_max(mut self, t_max: Floating) -> Self {
        assert_eq!(unsafe { qn_morethuente_with_t_max(&mut self.ls, t_max) }, QN_OK, "{}", last_error());
        self
    }
    pub fn with_c1(mut self, c1: Floating) -> Self {
        assert_eq!(unsafe { qn_morethuente_with_c1(&mut self.ls, c1) }, QN_OK, "{}", last_error());
        self
    }
    pub fn with_c2(mut self, c2: Floating) -> Self {
        assert_eq!(unsafe { qn_morethuente_with_c2(&mut self.ls, c2) }, QN_OK, "{}", last_error());
        self
    }
}

/// `BackTracking` (backtracking.rs:3-58).  The reference keeps `beta` private without a getter, so this is constructed
/// from the same two numbers rather than converted from a `BackTracking`.
#[derive(Clone, Debug)]
pub struct GpuBackTracking {
    ls: qn_linesearch,
}

impl GpuBackTracking {
    pub fn new(c1: Floating, beta: Floating) -> Self {
        let mut ls = unsafe { std::mem::zeroed::<qn_linesearch>() };
        unsafe { qn_backtracking_new(&mut ls, c1, beta) };
        GpuBackTracking { ls }
    }
}

fn step_len_on_device<F>(
    ls: &mut qn_linesearch,
    x_k: &DVector<Floating>,
    eval_x_k: &FuncEvalMultivariate,
    direction_k: &DVector<Floating>,
    oracle: &mut F,
    max_iter: usize,
) -> Floating
where
    F: FnMut(&DVector<Floating>) -> FuncEvalMultivariate,
{
    let o = host_oracle(oracle, false);
    let mut t = 0.0;
    let code = unsafe {
        qn_compute_step_len(
            default_context(),
            ls,
            x_k.as_ptr(),
            *eval_x_k.f(),
            eval_x_k.g().as_ptr(),
            direction_k.as_ptr(),
            x_k.len(),
            &o,
            max_iter,
            &mut t,
        )
    };
    // the trait returns a bare Floating (line_search/mod.rs:22): a HIP failure can only panic
    assert_eq!(code, QN_OK, "qn_compute_step_len: {}", last_error());
    t
}

macro_rules! impl_line_search {
    ($t:ty) => {
        impl LineSearch for $t {
            fn compute_step_len(
                &mut self,
                x_k: &DVector<Floating>,
                eval_x_k: &FuncEvalMultivariate,
                direction_k: &DVector<Floating>,
                oracle: &mut impl FnMut(&DVector<Floating>) -> FuncEvalMultivariate,
                max_iter: usize,
            ) -> Floating {
                step_len_on_device(&mut self.ls, x_k, eval_x_k, direction_k, oracle, max_iter)
            }
        }
        impl GpuLineSearch for $t {
            fn ffi(&mut self) -> &mut qn_linesearch {
                &mut self.ls
            }
        }
    };
}
impl_line_search!(GpuMoreThuente);
impl_line_search!(GpuBackTracking);

// The condition traits of line_search/mod.rs:25-83, as the reference implements them for its own structs (morethuente.rs,
// backtracking.rs:13-18): only the sensitivities are supplied, the tests themselves are the traits' default methods (and
// `WolfeConditions` follows from the blanket impl, mod.rs:85-86).
impl SufficientDecreaseCondition for GpuMoreThuente {
    fn c1(&self) -> Floating {
        self.ls.c1
    }
}
impl CurvatureCondition for GpuMoreThuente {
    fn c2(&self) -> Floating {
        self.ls.c2
    }
}
impl SufficientDecreaseCondition for GpuBackTracking {
    fn c1(&self) -> Floating {
        self.ls.bt_c1
    }
}

// ------------------------------------------------------------------------------------------------
// a device-resident objective (the benchmark's dense quadratic; log-sum-exp): evaluated by the library's own kernels
// ------------------------------------------------------------------------------------------------
pub struct DeviceObjective {
    h: *mut qn_objective,
    n: usize,
}

impl DeviceObjective {
    /// f = 1/2 x'Qx - b'x, g = Qx - b; `q_rowmajor` is the full symmetric n x n matrix.
    pub fn quadratic(q_rowmajor: &[Floating], b: &DVector<Floating>) -> Result<Self, SolverError> {
        let n = b.len();
        if q_rowmajor.len() != n * n {
            return Err(SolverError::ErrorInputParams);
        }
        let mut h = std::ptr::null_mut();
        status_to_result(unsafe { qn_quadratic_create(default_context(), n, q_rowmajor.as_ptr(), b.as_ptr(), &mut h) })?;
        Ok(DeviceObjective { h, n })
    }
    /// f = log sum_i exp(a_i'x + c_i) + mu/2 ||x||^2; `a_rowmajor` is m x n.
    pub fn log_sum_exp(a_rowmajor: &[Floating], c: &DVector<Floating>, n: usize, mu: Floating) -> Result<Self, SolverError> {
        let m = c.len();
        if a_rowmajor.len() != m * n {
            return Err(SolverError::ErrorInputParams);
        }
        let mut h = std::ptr::null_mut();
        status_to_result(unsafe { qn_logsumexp_create(default_context(), m, n, a_rowmajor.as_ptr(), c.as_ptr(), mu, &mut h) })?;
        Ok(DeviceObjective { h, n })
    }
    /// One evaluation at a host point (`let eval = f_and_g(&x)` after `minimize`).
    pub fn eval(&self, x: &DVector<Floating>) -> Result<FuncEvalMultivariate, SolverError> {
        if x.len() != self.n {
            return Err(SolverError::ErrorInputParams);
        }
        let mut f = 0.0;
        let mut g = DVector::zeros(self.n);
        status_to_result(unsafe { qn_objective_eval(self.h, x.as_ptr(), &mut f, g.as_mut_ptr()) })?;
        Ok(FuncEvalMultivariate::new(f, g))
    }
}

impl Drop for DeviceObjective {
    fn drop(&mut self) {
        unsafe { qn_objective_destroy(self.h) }
    }
}

// ------------------------------------------------------------------------------------------------
// solvers
// ------------------------------------------------------------------------------------------------
/// What GpuBFGS / GpuDFP / GpuGradientDescent share: the device handle and the host mirror of (x, k) that `xk()` / `k()`
/// hand out by reference (ls_solver.rs:24-27).
struct Core {
    h: *mut qn_solver,
    n: usize,
    tol: Floating,
    x: DVector<Floating>,
    k: usize,
}

impl Core {
    fn new(method: c_int, tol: Floating, x0: DVector<Floating>) -> Self {
        let mut h = std::ptr::null_mut();
        let code = unsafe { qn_solver_create(default_context(), method, tol, x0.as_ptr(), x0.len(), &mut h) };
        assert_eq!(code, QN_OK, "qn_solver_create: {}", last_error());
        Core { h, n: x0.len(), tol, x: x0, k: 0 }
    }
    fn option(&self, f: unsafe extern "C" fn(*mut qn_solver, *mut f64, *mut c_int) -> c_int) -> Option<Floating> {
        let (mut v, mut some) = (0.0, 0);
        unsafe { f(self.h, &mut v, &mut some) };
        if some != 0 {
            Some(v)
        } else {
            None
        }
    }
    fn flag(&self, f: unsafe extern "C" fn(*mut qn_solver, *mut c_int) -> c_int) -> bool {
        let mut v = 0;
        unsafe { f(self.h, &mut v) };
        v != 0
    }
    /// host mirror -> device (the caller may have written through `xk_mut()` / `k_mut()`)
    fn push(&mut self) -> Result<(), SolverError> {
        status_to_result(unsafe { qn_solver_set_x(self.h, self.x.as_ptr()) })?;
        status_to_result(unsafe { qn_solver_set_k(self.h, self.k) })
    }
    /// device -> host mirror
    fn pull(&mut self) {
        unsafe {
            qn_solver_get_x(self.h, self.x.as_mut_ptr());
            self.k = qn_solver_k(self.h);
        }
    }
    fn approx_inv_hessian(&self) -> DMatrix<Floating> {
        let mut m = DMatrix::zeros(self.n, self.n);
        let code = unsafe { qn_solver_get_inv_hessian(self.h, m.as_mut_ptr(), 1) }; // column-major, like DMatrix
        assert_eq!(code, QN_OK, "qn_solver_get_inv_hessian: {}", last_error());
        m
    }
    fn direction(&mut self, eval: &FuncEvalMultivariate) -> Result<DVector<Floating>, SolverError> {
        if eval.g().len() != self.n {
            return Err(SolverError::ErrorInputParams);
        }
        let mut d = DVector::zeros(self.n);
        status_to_result(unsafe { qn_solver_compute_direction(self.h, eval.g().as_ptr(), d.as_mut_ptr()) })?;
        Ok(d)
    }
    fn minimize_with(&mut self, ls: &mut qn_linesearch, o: &qn_oracle, max_iter_solver: usize, max_iter_line_search: usize,
                     callback: qn_callback_fn, callback_user: *mut c_void) -> Result<(), SolverError> {
        self.push()?;
        let code = unsafe { qn_minimize(self.h, ls, o, max_iter_solver, max_iter_line_search, callback, callback_user) };
        self.pull();
        status_to_result(code)
    }
}

impl Drop for Core {
    fn drop(&mut self) {
        unsafe { qn_solver_destroy(self.h) }
    }
}

macro_rules! gpu_solver {
    ($(#[$doc:meta])* $name:ident, $method:expr, $quasi_newton:expr) => {
        $(#[$doc])*
        pub struct $name {
            core: Core,
        }

        impl $name {
            pub fn new(tol: Floating, x0: DVector<Floating>) -> Self {
                $name { core: Core::new($method, tol, x0) }
            }
            // the getters derive_getters generates on the reference struct (bfgs.rs:3-12)
            pub fn x(&self) -> &DVector<Floating> {
                &self.core.x
            }
            pub fn tol(&self) -> &Floating {
                &self.core.tol
            }

            /// `LineSearchSolver::minimize` (ls_solver.rs:66-111), the whole loop in ONE `qn_minimize` call.  Same argument
            /// list as the trait method, but under ANOTHER NAME on purpose: Rust resolves `solver.minimize(..)` to an
            /// inherent method of that name before it looks at traits, so an inherent `minimize<LS: GpuLineSearch>` would have
            /// turned every call site that passes one of the reference's own line searches (`MoreThuente`, `BackTracking`:
            /// not `GpuLineSearch`) into a compile error instead of a fall-back.  As it is, `solver.minimize(..)` is the
            /// trait's default method for ANY `LS: LineSearch` (hook by hook through the ABI, see `impl LineSearchSolver`
            /// below), and `solver.minimize_on_device(..)` is the fast path for the GPU line searches.  The closure is
            /// called in exactly the reference's order: loop top (:79), every line-search evaluation, the gradient at the
            /// accepted point (bfgs.rs:98).
            pub fn minimize_on_device<LS: GpuLineSearch>(
                &mut self,
                line_search: &mut LS,
                mut oracle: impl FnMut(&DVector<Floating>) -> FuncEvalMultivariate,
                max_iter_solver: usize,
                max_iter_line_search: usize,
                mut callback: Option<&mut dyn FnMut(&Self)>,
            ) -> Result<(), SolverError> {
                struct CallbackEnv<'a, 'b> {
                    me: *mut $name,
                    f: &'a mut Option<&'b mut dyn FnMut(&$name)>,
                }
                unsafe extern "C" fn callback_trampoline(user: *mut c_void, _solver: *mut qn_solver) {
                    let env = &mut *(user as *mut CallbackEnv);
                    (*env.me).core.pull(); // the callback sees the state after `k += 1` (ls_solver.rs:104-107)
                    if let Some(f) = env.f.as_mut() {
                        f(&*env.me)
                    }
                }
                let o = host_oracle(&mut oracle, false);
                let has_callback = callback.is_some();
                let mut env = CallbackEnv { me: self as *mut $name, f: &mut callback };
                let core = &mut self.core as *mut Core; // (`env.me` aliases self for the duration of the call)
                unsafe {
                    (*core).minimize_with(
                        line_search.ffi(),
                        &o,
                        max_iter_solver,
                        max_iter_line_search,
                        if has_callback { Some(callback_trampoline) } else { None },
                        &mut env as *mut CallbackEnv as *mut c_void,
                    )
                }
            }

            /// The benchmark path: a device-resident objective, nothing crosses PCIe per iteration.  Each DISTINCT point is
            /// evaluated once (`qn_oracle.memoize = 1`): the values fed to the algorithm are those of the reference's call
            /// sequence, only the number of evaluations differs (5 -> 2 per iteration for More-Thuente on a quadratic).
            pub fn minimize_objective<LS: GpuLineSearch>(
                &mut self,
                line_search: &mut LS,
                objective: &DeviceObjective,
                max_iter_solver: usize,
                max_iter_line_search: usize,
            ) -> Result<(), SolverError> {
                if objective.n != self.core.n {
                    return Err(SolverError::ErrorInputParams);
                }
                let o = qn_oracle {
                    kind: QN_ORACLE_OBJECTIVE,
                    memoize: 1,
                    host_fn: None,
                    host_user: std::ptr::null_mut(),
                    device_fn: None,
                    device_user: std::ptr::null_mut(),
                    objective: objective.h,
                    host_hessian_fn: None,
                };
                self.core.minimize_with(line_search.ffi(), &o, max_iter_solver, max_iter_line_search, None, std::ptr::null_mut())
            }
        }

        impl ComputeDirection for $name {
            /// bfgs.rs:42-49 / dfp.rs:42-49 (`-H g`, one pass over the device-resident matrix); gradient_descent.rs:24-30 (`-g`)
            fn compute_direction(&mut self, eval_x_k: &FuncEvalMultivariate) -> Result<DVector<Floating>, SolverError> {
                self.core.direction(eval_x_k)
            }
        }

        impl LineSearchSolver for $name {
            fn xk(&self) -> &DVector<Floating> {
                &self.core.x
            }
            fn xk_mut(&mut self) -> &mut DVector<Floating> {
                &mut self.core.x
            }
            fn k(&self) -> &usize {
                &self.core.k
            }
            fn k_mut(&mut self) -> &mut usize {
                &mut self.core.k
            }
            fn has_converged(&self, eval: &FuncEvalMultivariate) -> bool {
                if $quasi_newton {
                    // bfgs.rs:64-76
                    self.core.flag(qn_solver_next_iterate_too_close)
                        || self.core.flag(qn_solver_gradient_next_iterate_too_close)
                        || eval.g().norm() < self.core.tol
                } else {
                    // gradient_descent.rs:46-53: infinity norm
                    eval.g().iter().fold(Floating::NEG_INFINITY, |acc, x| x.abs().max(acc)) < self.core.tol
                }
            }

            /// The hook of bfgs.rs:78-130 / dfp.rs:78-118: line search, next iterate, s and y on the host (n-vectors), the
            /// secant update of the n x n matrix on the device.  With a `GpuMoreThuente` / `GpuBackTracking` the line
            /// search's state machine runs on the GPU as well.
            fn update_next_iterate<LS: LineSearch>(
                &mut self,
                line_search: &mut LS,
                eval_x_k: &FuncEvalMultivariate,
                oracle: &mut impl FnMut(&DVector<Floating>) -> FuncEvalMultivariate,
                direction: &DVector<Floating>,
                max_iter_line_search: usize,
            ) -> Result<(), SolverError> {
                let step = line_search.compute_step_len(self.xk(), eval_x_k, direction, oracle, max_iter_line_search);
                let next_iterate = self.xk() + step * direction;
                if $quasi_newton {
                    let s = &next_iterate - self.xk();
                    let y = oracle(&next_iterate).g() - eval_x_k.g();
                    *self.xk_mut() = next_iterate;
                    status_to_result(unsafe { qn_solver_set_x(self.core.h, self.core.x.as_ptr()) })?;
                    // s_norm / y_norm, the two "too close" early returns and the update itself (bfgs.rs:96-127)
                    status_to_result(unsafe { qn_solver_secant_update(self.core.h, s.as_ptr(), y.as_ptr()) })
                } else {
                    *self.xk_mut() = next_iterate; // ls_solver.rs:60-62
                    Ok(())
                }
            }
        }
    };
}

gpu_solver!(
    /// Drop-in for `BFGS` (quasi_newton/bfgs.rs:4-130): the dense inverse Hessian lives in HBM and never leaves it.
    GpuBFGS, QN_BFGS, true
);
gpu_solver!(
    /// Drop-in for `DFP` (quasi_newton/dfp.rs).
    GpuDFP, QN_DFP, true
);
gpu_solver!(
    /// Drop-in for `GradientDescent` (steepest_descent/gradient_descent.rs:7-82); `tol` is its `grad_tol`.
    GpuGradientDescent, QN_GRADIENT_DESCENT, false
);

macro_rules! quasi_newton_getters {
    ($name:ident) => {
        impl $name {
            pub fn s_norm(&self) -> Option<Floating> {
                self.core.option(qn_solver_s_norm)
            }
            pub fn y_norm(&self) -> Option<Floating> {
                self.core.option(qn_solver_y_norm)
            }
            pub fn next_iterate_too_close(&self) -> bool {
                self.core.flag(qn_solver_next_iterate_too_close) // bfgs.rs:15-20
            }
            pub fn gradient_next_iterate_too_close(&self) -> bool {
                self.core.flag(qn_solver_gradient_next_iterate_too_close) // bfgs.rs:21-26
            }
            /// Lazy download (applies the pending rank-2 update first); the reference's getter returns a reference to a
            /// host matrix, this one an owned copy of the device one.
            pub fn approx_inv_hessian(&self) -> DMatrix<Floating> {
                self.core.approx_inv_hessian()
            }
            pub fn identity(&self) -> DMatrix<Floating> {
                DMatrix::identity(self.core.n, self.core.n) // bfgs.rs:9; the GPU solver keeps no copy of it
            }
        }
    };
}
quasi_newton_getters!(GpuBFGS);
quasi_newton_getters!(GpuDFP);
