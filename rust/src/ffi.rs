//! `extern "C"` declarations for include/qn_hip.h (QN_ABI_VERSION 5): one `pub fn` per entry point, parameter for parameter.
//! tests/test_abi_load.py parses this file and the header and compares names, arity and every parameter / return type.
//! NOT COMPILED in the build image (no Rust toolchain) -- see ../Cargo.toml.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const QN_ABI_VERSION: c_int = 5;

// qn_status == SolverError (ls_solver.rs:10-20); 0 is Ok(())
pub const QN_OK: c_int = 0;
pub const QN_MAX_ITER_REACHED: c_int = 1;
pub const QN_OUT_OF_DOMAIN: c_int = 2;
pub const QN_ERROR_INPUT_PARAMS: c_int = 3;
pub const QN_ABNORMAL_TERMINATION: c_int = 4;

pub const QN_LS_MORETHUENTE: i32 = 0;
pub const QN_LS_BACKTRACKING: i32 = 1;
pub const QN_LS_MORETHUENTE_B: i32 = 2;
pub const QN_LS_BACKTRACKING_B: i32 = 3;

pub const QN_ORACLE_HOST: i32 = 0;
pub const QN_ORACLE_DEVICE_FN: i32 = 1;
pub const QN_ORACLE_OBJECTIVE: i32 = 2;

pub const QN_BFGS: c_int = 0;
pub const QN_DFP: c_int = 1;
pub const QN_GRADIENT_DESCENT: c_int = 2;
pub const QN_NEWTON: c_int = 3;
pub const QN_SR1: c_int = 4;

// qn_option (ABI 5): what rounds 1-5 selected through negative codes of qn_solver_set_tiling; value != 0 on, 0 off
pub const QN_OPT_GENERIC_KERNELS: c_int = 1;
pub const QN_OPT_DEFERRED_UPDATE_STEP: c_int = 2;
pub const QN_OPT_SYMMETRIC_STORAGE: c_int = 3;
pub const QN_OPT_SECOND_GENERATION: c_int = 4;
pub const QN_OPT_FOLDED_ACCEPT_REDUCE: c_int = 5;
pub const QN_OPT_ROW_SLIVERS: c_int = 6;
pub const QN_OPT_EVAL_PAIR_INSTANCE: c_int = 7;
pub const QN_OPT_EVAL_MOVER_MULTIPLIER: c_int = 8;
pub const QN_OPT_TAIL_REDUCE: c_int = 9;
pub const QN_OPT_BOUNDED_SECOND_GENERATION: c_int = 10;
pub const QN_OPT_NEWTON_PIVOTED_LU: c_int = 11;
pub const QN_OPT_LU_PER_COLUMN_PANEL: c_int = 12;
pub const QN_OPT_LU_LOOKAHEAD: c_int = 13;
pub const QN_OPT_LU_ONE_LAUNCH_PANEL: c_int = 14;
pub const QN_OPT_LU_FORCE_WAIT_EXPIRY: c_int = 15;
pub const QN_OPT_CHUNKS_PER_TRIP: c_int = 16;
pub const QN_OPT_LU_SPLIT_ROLE_A: c_int = 17;
pub const QN_OPT_LU_SPLIT_MIN_ROWS: c_int = 18;
pub const QN_OPT_BTB_PROJECT_IN_EVAL: c_int = 19;
pub const QN_OPT_EVAL_ZIGZAG: c_int = 20;
pub const QN_OPT_TOUCH_H_ROWS: c_int = 21;
pub const QN_OPT_TOUCH_Q_ROWS: c_int = 22;

pub const QN_UNIQUE_ID_BYTES: usize = 128;
pub const QN_TRACE_LS_MODIFIED: i32 = 1 << 30;
pub const QN_PATH_FUSED: u32 = 1;
pub const QN_PATH_SYM: u32 = 2;
pub const QN_PATH_SYM_GENERIC: u32 = 4;
pub const QN_PATH_PIPELINED: u32 = 8;
pub const QN_PATH_SYM2: u32 = 16;
pub const QN_PATH_TILES1: u32 = 32;

#[repr(C)] pub struct qn_context { _p: [u8; 0] }
#[repr(C)] pub struct qn_solver { _p: [u8; 0] }
#[repr(C)] pub struct qn_objective { _p: [u8; 0] }

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct qn_linesearch {
    pub kind: i32,
    pub _pad: i32,
    pub c1: f64,
    pub c2: f64,
    pub t_min: f64,
    pub t_max: f64,
    pub delta_min: f64,
    pub delta: f64,
    pub delta_max: f64,
    pub bt_c1: f64,
    pub bt_beta: f64,
    pub lower_bound_host: *const f64,
    pub upper_bound_host: *const f64,
}

pub type qn_host_oracle_fn = Option<unsafe extern "C" fn(user: *mut c_void, x_host: *const f64, n: usize, f: *mut f64, g_host: *mut f64) -> c_int>;
pub type qn_device_oracle_fn = Option<unsafe extern "C" fn(user: *mut c_void, stream: *mut c_void, x_dev: *const f64, n: usize, f_dev: *mut f64, g_dev: *mut f64) -> c_int>;
pub type qn_host_hessian_fn = Option<unsafe extern "C" fn(user: *mut c_void, x_host: *const f64, n: usize, h_colmajor_host: *mut f64) -> c_int>;
pub type qn_host_allgather_fn = Option<unsafe extern "C" fn(user: *mut c_void, sendbuf: *const f64, recvbuf: *mut f64, count: usize) -> c_int>;
pub type qn_callback_fn = Option<unsafe extern "C" fn(user: *mut c_void, solver: *mut qn_solver)>;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct qn_oracle {
    pub kind: i32,
    pub memoize: i32,
    pub host_fn: qn_host_oracle_fn,
    pub host_user: *mut c_void,
    pub device_fn: qn_device_oracle_fn,
    pub device_user: *mut c_void,
    pub objective: *mut qn_objective,
    pub host_hessian_fn: qn_host_hessian_fn,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct qn_trace_rec {
    pub f: f64,
    pub gnorm: f64,
    pub t: f64,
    pub s_norm: f64,
    pub y_norm: f64,
    pub n_evals: i32,
    pub ls_iters: i32,
    pub ls_cases: i32,
    pub updated: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct qn_stats {
    pub iterations: u64,
    pub oracle_calls: u64,
    pub oracle_evals: u64,
    pub h_passes: u64,
    pub h_bytes: u64,
    pub obj_bytes: u64,
    pub launches: u64,
    pub host_syncs: u64,
    pub t_hpass_ms: f64,
    pub t_eval_ms: f64,
    pub t_ctl_ms: f64,
    pub t_comm_ms: f64,
    pub n_hpass_timed: u64,
    pub n_eval_timed: u64,
    pub n_ctl_timed: u64,
    pub n_comm_timed: u64,
    pub matrix_bytes_per_pass: u64,
    pub total_minimize_calls: u64,
    pub total_iterations: u64,
    pub total_oracle_calls: u64,
    pub total_oracle_evals: u64,
    pub total_h_passes: u64,
    pub total_h_bytes: u64,
    pub total_obj_bytes: u64,
    pub path: u32,
    pub _pad: u32,
    pub t_hreduce_ms: f64,
    pub t_ereduce_ms: f64,
    pub n_hreduce_timed: u64,
    pub n_ereduce_timed: u64,
    pub total_xchg_vector: u64,
    pub total_xchg_scalar: u64,
    pub t_newton_ms: f64,
    pub n_newton_timed: u64,
    pub newton_lu_sync_timeouts: u64,
}

extern "C" {
    pub fn qn_status_string(status: c_int) -> *const c_char;
    pub fn qn_last_error_message() -> *const c_char;
    pub fn qn_abi_version() -> c_int;

    // ---- context ----
    pub fn qn_device_count(out: *mut c_int) -> c_int;
    pub fn qn_context_create(device: c_int, out: *mut *mut qn_context) -> c_int;
    pub fn qn_comm_unique_id(out_128_bytes: *mut c_void) -> c_int;
    pub fn qn_context_create_sharded(device: c_int, rank: c_int, world: c_int, unique_id: *const c_void, out: *mut *mut qn_context) -> c_int;
    pub fn qn_context_create_sharded_host_exchange(device: c_int, rank: c_int, world: c_int, fn_: qn_host_allgather_fn, user: *mut c_void, out: *mut *mut qn_context) -> c_int;
    pub fn qn_context_set_allreduce(ctx: *mut qn_context, on: c_int) -> c_int;
    pub fn qn_context_set_host_exchange_async(ctx: *mut qn_context, on: c_int) -> c_int;
    pub fn qn_context_set_trial_vector_exchange(ctx: *mut qn_context, on: c_int) -> c_int;
    pub fn qn_context_destroy(ctx: *mut qn_context);
    pub fn qn_partition(n: usize, world: c_int, rows_per_rank: *mut usize, n_pad: *mut usize) -> c_int;
    pub fn qn_comm_selftest(ctx: *mut qn_context) -> c_int;
    pub fn qn_context_comm_check(ctx: *mut qn_context) -> c_int;
    pub fn qn_context_exchange_probe(ctx: *mut qn_context, count: usize, reps: c_int, out_us: *mut f64) -> c_int;
    pub fn qn_context_event_bracket_overhead(ctx: *mut qn_context, reps: c_int, out_ms: *mut f64) -> c_int;
    pub fn qn_context_synchronize(ctx: *mut qn_context) -> c_int;
    pub fn qn_context_rank(ctx: *const qn_context) -> c_int;
    pub fn qn_context_world(ctx: *const qn_context) -> c_int;
    pub fn qn_context_stream(ctx: *mut qn_context) -> *mut c_void;

    // ---- line searches (MoreThuente::default / with_*, BackTracking::new, the *B variants) ----
    pub fn qn_morethuente_default(ls: *mut qn_linesearch);
    pub fn qn_morethuente_with_deltas(ls: *mut qn_linesearch, dmin: f64, d: f64, dmax: f64) -> c_int;
    pub fn qn_morethuente_with_t_min(ls: *mut qn_linesearch, t_min: f64) -> c_int;
    pub fn qn_morethuente_with_t_max(ls: *mut qn_linesearch, t_max: f64) -> c_int;
    pub fn qn_morethuente_with_c1(ls: *mut qn_linesearch, c1: f64) -> c_int;
    pub fn qn_morethuente_with_c2(ls: *mut qn_linesearch, c2: f64) -> c_int;
    pub fn qn_backtracking_new(ls: *mut qn_linesearch, c1: f64, beta: f64);
    pub fn qn_morethuente_b_new(ls: *mut qn_linesearch);
    pub fn qn_backtracking_b_new(ls: *mut qn_linesearch, c1: f64, beta: f64, lower_bound_host: *const f64, upper_bound_host: *const f64);
    pub fn qn_linesearch_with_lower_bound(ls: *mut qn_linesearch, lower_bound_host: *const f64);
    pub fn qn_linesearch_with_upper_bound(ls: *mut qn_linesearch, upper_bound_host: *const f64);

    // ---- device-resident objectives ----
    pub fn qn_quadratic_create(ctx: *mut qn_context, n: usize, q_rowmajor_host: *const f64, b_host: *const f64, out: *mut *mut qn_objective) -> c_int;
    pub fn qn_quadratic_create_synthetic(ctx: *mut qn_context, n: usize, seed: u64, diag_host: *const f64, b_host: *const f64, out: *mut *mut qn_objective) -> c_int;
    pub fn qn_logsumexp_create(ctx: *mut qn_context, m: usize, n: usize, a_rowmajor_host: *const f64, c_host: *const f64, mu: f64, out: *mut *mut qn_objective) -> c_int;
    pub fn qn_objective_destroy(obj: *mut qn_objective);
    pub fn qn_objective_eval(obj: *mut qn_objective, x_host: *const f64, f: *mut f64, g_host: *mut f64) -> c_int;
    pub fn qn_objective_get_rows(obj: *mut qn_objective, row0: usize, nrows: usize, out_host: *mut f64) -> c_int;

    // ---- solvers ----
    pub fn qn_solver_create(ctx: *mut qn_context, method: c_int, tol: f64, x0_host: *const f64, n: usize, out: *mut *mut qn_solver) -> c_int;
    pub fn qn_solver_destroy(s: *mut qn_solver);
    pub fn qn_solver_set_bounds(s: *mut qn_solver, lower_bound_host: *const f64, upper_bound_host: *const f64) -> c_int;
    pub fn qn_solver_reset(s: *mut qn_solver, x0_host: *const f64) -> c_int;
    pub fn qn_minimize(s: *mut qn_solver, ls: *mut qn_linesearch, oracle: *const qn_oracle, max_iter_solver: usize, max_iter_line_search: usize, callback: qn_callback_fn, callback_user: *mut c_void) -> c_int;
    pub fn qn_compute_step_len(ctx: *mut qn_context, ls: *mut qn_linesearch, x_k_host: *const f64, f_k: f64, g_k_host: *const f64, direction_host: *const f64, n: usize, oracle: *const qn_oracle, max_iter: usize, step_out: *mut f64) -> c_int;
    pub fn qn_solver_n(s: *const qn_solver) -> usize;
    pub fn qn_solver_k(s: *const qn_solver) -> usize;
    pub fn qn_solver_set_k(s: *mut qn_solver, k: usize) -> c_int;
    pub fn qn_solver_tol(s: *const qn_solver) -> f64;
    pub fn qn_solver_get_x(s: *mut qn_solver, out_host: *mut f64) -> c_int;
    pub fn qn_solver_set_x(s: *mut qn_solver, x_host: *const f64) -> c_int;
    pub fn qn_solver_s_norm(s: *mut qn_solver, out: *mut f64, is_some: *mut c_int) -> c_int;
    pub fn qn_solver_y_norm(s: *mut qn_solver, out: *mut f64, is_some: *mut c_int) -> c_int;
    pub fn qn_solver_next_iterate_too_close(s: *mut qn_solver, out: *mut c_int) -> c_int;
    pub fn qn_solver_gradient_next_iterate_too_close(s: *mut qn_solver, out: *mut c_int) -> c_int;
    pub fn qn_solver_decrement_squared(s: *mut qn_solver, out: *mut f64, is_some: *mut c_int) -> c_int;
    pub fn qn_solver_get_inv_hessian(s: *mut qn_solver, out_colmajor_host: *mut f64, all_ranks: c_int) -> c_int;
    pub fn qn_solver_set_inv_hessian(s: *mut qn_solver, h_colmajor_host: *const f64) -> c_int;
    pub fn qn_solver_compute_direction(s: *mut qn_solver, g_host: *const f64, d_host: *mut f64) -> c_int;
    pub fn qn_solver_secant_update(s: *mut qn_solver, s_host: *const f64, y_host: *const f64) -> c_int;

    // ---- instrumentation ----
    pub fn qn_solver_set_trace(s: *mut qn_solver, cap: usize, with_x: c_int) -> c_int;
    pub fn qn_solver_get_trace(s: *mut qn_solver, out_host: *mut qn_trace_rec, cap: usize, len: *mut usize, x_trace_host: *mut f64) -> c_int;
    pub fn qn_solver_get_stats(s: *mut qn_solver, out: *mut qn_stats) -> c_int;
    pub fn qn_solver_set_profiling(s: *mut qn_solver, on: c_int) -> c_int;
    pub fn qn_solver_set_sync_mode(s: *mut qn_solver, sync: c_int) -> c_int;
    pub fn qn_solver_set_tiling(s: *mut qn_solver, rows_per_block: c_int, col_splits: c_int) -> c_int;
    pub fn qn_solver_set_option(s: *mut qn_solver, option: c_int, value: c_int) -> c_int;

    // ---- kernel-level primitives on device buffers ----
    pub fn qn_dev_alloc(ctx: *mut qn_context, bytes: usize, out_dev: *mut *mut c_void) -> c_int;
    pub fn qn_dev_free(ctx: *mut qn_context, dev: *mut c_void) -> c_int;
    pub fn qn_h2d(ctx: *mut qn_context, dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> c_int;
    pub fn qn_d2h(ctx: *mut qn_context, dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> c_int;
    pub fn qn_gemv(ctx: *mut qn_context, a_dev: *const f64, ld: usize, nrows: usize, ncols: usize, x_dev: *const f64, y_dev: *mut f64) -> c_int;
    pub fn qn_rank2_update(ctx: *mut qn_context, h_dev: *mut f64, ld: usize, row0: usize, nrows: usize, n: usize, s_dev: *const f64, u_dev: *const f64, c_ss: f64, c_su: f64, c_uu: f64) -> c_int;
    pub fn qn_axpy(ctx: *mut qn_context, n: usize, x_dev: *const f64, t: f64, d_dev: *const f64, out_dev: *mut f64) -> c_int;
    pub fn qn_dot(ctx: *mut qn_context, n: usize, a_dev: *const f64, b_dev: *const f64, out_host: *mut f64) -> c_int;
    pub fn qn_nrm2(ctx: *mut qn_context, n: usize, a_dev: *const f64, out_host: *mut f64) -> c_int;
}
