/*
 * qn_hip.h -- C ABI of libqn_hip.so: the MI355X (gfx950) quasi-Newton / line-search inner loop.
 *
 * This is the drop-in boundary for the hot path of fedemagnani/optimization-solvers
 * (src/ls_solver.rs, src/quasi_newton/{bfgs,dfp}.rs, src/line_search/{mod,backtracking,morethuente}.rs).
 * The reference has no FFI on this path -- its "operator API" is the Rust trait surface -- so every entry
 * point below names the reference item it replaces (file:line relative to the reference repo).  A Rust
 * shim implementing `ComputeDirection` / `LineSearchSolver` on top of these is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; all host matrices are dense f64; `*_host` pointers are host
 * memory, `*_dev` device memory of the context's GPU.  Every function returns a qn_status unless stated;
 * on QN_ABNORMAL_TERMINATION / QN_ERROR_INPUT_PARAMS qn_last_error_message() explains.  There is no CPU
 * fallback: without a usable GPU qn_context_create fails.  All calls on one context must come from one
 * thread at a time (the reference is `&mut self` everywhere: ls_solver.rs:23-112); host callbacks run on
 * the calling thread.
 */
#ifndef QN_HIP_H
#define QN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QN_ABI_VERSION 5

/* SolverError (ls_solver.rs:10-20); 0 is Ok(()) */
typedef enum {
    QN_OK = 0,
    QN_MAX_ITER_REACHED = 1,     /* "Max iter reached"          ls_solver.rs:12-13,110 */
    QN_OUT_OF_DOMAIN = 2,        /* "Out of domain"             ls_solver.rs:14-15,37-40 */
    QN_ERROR_INPUT_PARAMS = 3,   /* "Error in input parameters" ls_solver.rs:16-17 */
    QN_ABNORMAL_TERMINATION = 4  /* "Abnormal termination"      ls_solver.rs:18-19 (HIP / RCCL failures map here) */
} qn_status;

const char* qn_status_string(int status); /* the Display strings of ls_solver.rs:12-19 */
const char* qn_last_error_message(void);  /* thread-local detail of the last non-OK return */
int qn_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Context: one GPU, one HIP stream, optionally one rank of a row-sharded group.
 * The reference has no device or process boundary; this is new.
 * ------------------------------------------------------------------------------------------- */
typedef struct qn_context qn_context;

int qn_device_count(int* out);
int qn_context_create(int device, qn_context** out);

/* Row-sharded group: rank p of `world` owns rows [p*rpr, (p+1)*rpr) of H and of the objective's matrix;
 * all n-vectors are replicated; mat-vec slices are exchanged with one RCCL all-gather per pass.
 * `unique_id` is the 128-byte ncclUniqueId produced by qn_comm_unique_id on rank 0 and distributed by the
 * caller (torch.distributed / MPI / a file). */
#define QN_UNIQUE_ID_BYTES 128
int qn_comm_unique_id(void* out_128_bytes);
int qn_context_create_sharded(int device, int rank, int world, const void* unique_id, qn_context** out);

/* Test / bring-up alternative to RCCL: the caller performs the exchange on host buffers (e.g. with
 * torch.distributed gloo).  `fn(user, sendbuf, recvbuf, count)`: recvbuf[r*count .. (r+1)*count) must
 * receive rank r's sendbuf (an all-gather of `count` doubles per rank).  Returns 0 on success. */
typedef int (*qn_host_allgather_fn)(void* user, const double* sendbuf, double* recvbuf, size_t count);
int qn_context_create_sharded_host_exchange(int device, int rank, int world, qn_host_allgather_fn fn, void* user,
                                            qn_context** out);
/* Row-sharded runs on symmetric storage exchange per-rank PARTIAL n-vectors.  Default (on = 0): one all-gather, every rank adds
 * the P contributions in rank order -- bitwise identical on all ranks, and identical between RCCL and the host-staged exchange.
 * on != 0: ncclAllReduce(ncclSum) instead (the operation north_star names; 1/P of the bytes, RCCL's summation order). */
int qn_context_set_allreduce(qn_context* ctx, int on);
/* Host-exchange contexts only: on != 0 makes every exchange a stream-ordered triple (device-to-pinned copy, the callback as a
 * host node of the stream, pinned-to-device copy) with no synchronisation, so the pipelined launch logic -- which RCCL runs use
 * -- can be rehearsed with several ranks on one GPU.  The callback then runs on a runtime thread, not on the caller's. */
int qn_context_set_host_exchange_async(qn_context* ctx, int on);
/* Row-sharded second-generation runs on a quadratic objective (ABI 5, round 6; DESIGN.md 9.1's fallback): on != 0 lets every evaluation's
 * collective carry the rank's partial n-vector of the TRIAL point beside its 8 KB of scalars (one grouped all-gather), so that an accepted
 * evaluation needs no exchange of its own: E + 1 collectives per iteration instead of E + 2, n doubles per rank more per trial.  For nodes
 * where a small collective between two launches costs much more than its bytes (bench.py --gpus N prints the probed latencies).  The same
 * iterates, bit for bit.  Not together with qn_context_set_allreduce (QN_ERROR_INPUT_PARAMS). */
int qn_context_set_trial_vector_exchange(qn_context* ctx, int on);
void qn_context_destroy(qn_context* ctx);
/* the row partition used for H and the objective's matrix: rows per rank (a multiple of 16) and padded dimension */
int qn_partition(size_t n, int world, size_t* rows_per_rank, size_t* n_pad);
/* diagnostics: create a 1-rank RCCL communicator on this context's GPU, all-gather a small buffer in place on the
 * context's stream and verify it (checks that librccl loads and that the calling convention matches) */
int qn_comm_selftest(qn_context* ctx);
/* collective (every rank calls it): one rank-tagged all-gather through the context's own exchange, then three of different
 * sizes as ONE group (the form the row kernels' exchanges take), each verified on every rank */
int qn_context_comm_check(qn_context* ctx);
/* The latency of ONE exchange (all-gather) of `count` doubles per rank on this context's exchange, between other work on its stream: `reps` timed
 * repetitions bracketed by HIP events; out_us[0] = median, [1] = minimum, [2] = maximum, in microseconds (zeros on a one-rank context).  Collective:
 * call on every rank with the same arguments.  (No counterpart in the reference, which is one thread: bench.py --gpus N reports these in front
 * of its timed region, for the three sizes a row-sharded BFGS iteration exchanges -- 8 KB of scalars, n and 2 n doubles.) */
int qn_context_exchange_probe(qn_context* ctx, size_t count, int reps, double* out_us);
/* measurement aid: the fixed part (ms) of what a hipEventRecord / launch / hipEventRecord bracket on the idle context stream
 * reports beyond the bracketed kernel's own duration: 2 * bracket(one empty kernel) - bracket(two empty kernels) */
int qn_context_event_bracket_overhead(qn_context* ctx, int reps, double* out_ms);
int qn_context_synchronize(qn_context* ctx);
int qn_context_rank(const qn_context* ctx);
int qn_context_world(const qn_context* ctx);
void* qn_context_stream(qn_context* ctx); /* hipStream_t */

/* ---------------------------------------------------------------------------------------------
 * Line searches.  Plain structs by value, as the reference's are plain data.
 *   QN_LS_MORETHUENTE : MoreThuente        (morethuente.rs:6-62, compute_step_len :165-297)
 *   QN_LS_BACKTRACKING: BackTracking       (backtracking.rs:3-58)
 * ------------------------------------------------------------------------------------------- */
enum { QN_LS_MORETHUENTE = 0, QN_LS_BACKTRACKING = 1,
       QN_LS_MORETHUENTE_B = 2 /* MoreThuenteB, morethuente_b.rs */, QN_LS_BACKTRACKING_B = 3 /* BackTrackingB, backtracking_b.rs */ };
typedef struct {
    int32_t kind;
    int32_t _pad;
    double c1, c2, t_min, t_max, delta_min, delta, delta_max; /* More-Thuente fields, morethuente.rs:6-14 */
    double bt_c1, bt_beta;                                    /* BackTracking{c1, beta}, backtracking.rs:3-6 */
    /* the *_B variants hold their own box (morethuente_b.rs:14-15, backtracking_b.rs:7-8): host vectors of length n,
     * read by qn_minimize; NULL = unbounded side */
    const double* lower_bound_host;
    const double* upper_bound_host;
} qn_linesearch;

void qn_morethuente_default(qn_linesearch* ls);                      /* MoreThuente::default, morethuente.rs:16-28 */
int qn_morethuente_with_deltas(qn_linesearch* ls, double dmin, double d, double dmax); /* :31-41 */
int qn_morethuente_with_t_min(qn_linesearch* ls, double t_min);      /* :42-45 */
int qn_morethuente_with_t_max(qn_linesearch* ls, double t_max);      /* :46-49 */
int qn_morethuente_with_c1(qn_linesearch* ls, double c1);            /* :50-55; the assert!s become QN_ERROR_INPUT_PARAMS */
int qn_morethuente_with_c2(qn_linesearch* ls, double c2);            /* :56-62 */
void qn_backtracking_new(qn_linesearch* ls, double c1, double beta); /* BackTracking::new, backtracking.rs:8-10 */
void qn_morethuente_b_new(qn_linesearch* ls);                          /* MoreThuenteB::new(n): defaults, bounds -inf/+inf, morethuente_b.rs:18-31 */
void qn_backtracking_b_new(qn_linesearch* ls, double c1, double beta, const double* lower_bound_host, const double* upper_bound_host); /* backtracking_b.rs:10-23 */
void qn_linesearch_with_lower_bound(qn_linesearch* ls, const double* lower_bound_host); /* morethuente_b.rs:32-35 */
void qn_linesearch_with_upper_bound(qn_linesearch* ls, const double* upper_bound_host); /* morethuente_b.rs:36-39 */

/* ---------------------------------------------------------------------------------------------
 * Oracle: `impl FnMut(&DVector<f64>) -> FuncEvalMultivariate` (ls_solver.rs:69, func_eval.rs:4-41).
 * ------------------------------------------------------------------------------------------- */
/* host closure: writes *f and g[0..n); return value 0 (the reference closure cannot fail; non-zero aborts
 * the run with QN_ABNORMAL_TERMINATION). */
typedef int (*qn_host_oracle_fn)(void* user, const double* x_host, size_t n, double* f, double* g_host);
/* device closure: enqueue work on `stream` that reads x_dev[0..n) and writes *f_dev and g_dev[0..n). */
typedef int (*qn_device_oracle_fn)(void* user, void* stream, const double* x_dev, size_t n, double* f_dev, double* g_dev);
/* Newton only: the Hessian part of the FuncEval (func_eval.rs:8,27-33) at x, column-major n x n (like DMatrix). */
typedef int (*qn_host_hessian_fn)(void* user, const double* x_host, size_t n, double* h_colmajor_host);

typedef struct qn_objective qn_objective; /* a device-resident objective owned by the library */

enum { QN_ORACLE_HOST = 0, QN_ORACLE_DEVICE_FN = 1, QN_ORACLE_OBJECTIVE = 2 };
typedef struct {
    int32_t kind;
    /* 0: reproduce the reference's call sequence exactly (loop-top call ls_solver.rs:79, every line-search
     *    call incl. the per-iteration re-evaluation at tl morethuente.rs:217, and bfgs.rs:98);
     * 1: evaluate each DISTINCT point once (the oracle must then be a pure function of x).  The values fed
     *    to the algorithm are identical; only the number of evaluations changes (5 -> 2 per iteration for
     *    More-Thuente on a quadratic). */
    int32_t memoize;
    qn_host_oracle_fn host_fn;
    void* host_user;
    qn_device_oracle_fn device_fn;
    void* device_user;
    qn_objective* objective;
    qn_host_hessian_fn host_hessian_fn; /* Newton with a host closure; device objectives supply their own Hessian */
} qn_oracle;

/* Built-in benchmark objective (build-defined): f = 1/2 x'Qx - b'x, g = Qx - b.
 * `q_rowmajor_host` is the full symmetric n x n matrix; each rank uploads only its own rows.  (A matrix that is not symmetric
 * bit for bit is multiplied as given -- the symmetric-storage evaluation is then not used.) */
int qn_quadratic_create(qn_context* ctx, size_t n, const double* q_rowmajor_host, const double* b_host, qn_objective** out);
/* Same objective with Q generated on the device, shard-locally: off-diagonal Q_ij = u(seed,min,max)/n,
 * u in [-1,1) from a counter-based splitmix64 hash; diagonal given (SPD if diag_i >= 1). */
int qn_quadratic_create_synthetic(qn_context* ctx, size_t n, uint64_t seed, const double* diag_host, const double* b_host,
                                  qn_objective** out);
/* f = log sum_i exp(a_i'x + c_i) + mu/2 ||x||^2; A is m x n row-major, rows sharded. */
int qn_logsumexp_create(qn_context* ctx, size_t m, size_t n, const double* a_rowmajor_host, const double* c_host, double mu,
                        qn_objective** out);
void qn_objective_destroy(qn_objective* obj);
/* one evaluation at a host point (tests, and `let eval = f_and_g(x)` after minimize as in examples/quadratic.rs:39) */
int qn_objective_eval(qn_objective* obj, const double* x_host, double* f, double* g_host);
/* download rows [row0,row0+nrows) of this rank's matrix shard (row-major, n columns); rows outside the shard fail */
int qn_objective_get_rows(qn_objective* obj, size_t row0, size_t nrows, double* out_host);

/* ---------------------------------------------------------------------------------------------
 * Solvers: BFGS (bfgs.rs:4-127), DFP (dfp.rs), GradientDescent (gradient_descent.rs:7-82), Newton (newton/mod.rs).
 * ------------------------------------------------------------------------------------------- */
enum { QN_BFGS = 0, QN_DFP = 1, QN_GRADIENT_DESCENT = 2, QN_NEWTON = 3 /* newton/mod.rs:8-69, SURVEY.md 8(f) row f2 */,
       QN_SR1 = 4 /* sr1_b.rs (row f4; SR1B once qn_solver_set_bounds is called) */ };
typedef struct qn_solver qn_solver;

/* BFGS::new(tol, x0) / DFP::new / GradientDescent::new(grad_tol, x0): H = I (no identity copy is kept) */
int qn_solver_create(qn_context* ctx, int method, double tol, const double* x0_host, size_t n, qn_solver** out);
void qn_solver_destroy(qn_solver* s);
/* Row f4: BFGSB / DFPB / SR1B (bfgs_b.rs:43-77, dfp_b.rs, sr1_b.rs): box bounds on a BFGS / DFP / SR1 solver.  The current x
 * is projected (bfgs_b.rs:49) and every direction becomes P(x - H g) - x (bfgs_b.rs:72-75). */
int qn_solver_set_bounds(qn_solver* s, const double* lower_bound_host, const double* upper_bound_host);
/* back to the state right after BFGS::new(tol, x0): x = x0, H = I, k = 0, s_norm = y_norm = None */
int qn_solver_reset(qn_solver* s, const double* x0_host);

/* callback: Option<&mut dyn FnMut(&Self)> (ls_solver.rs:72,105-107); called after k += 1 */
typedef void (*qn_callback_fn)(void* user, qn_solver* solver);

/* LineSearchSolver::minimize (ls_solver.rs:66-111).  Resets k only (warm restart keeps x, H, s_norm, y_norm). */
/* `ls` is not const: MoreThuenteB clips its t_max to the box and keeps it clipped (morethuente_b.rs:201 `self.t_max = ...`). */
int qn_minimize(qn_solver* s, qn_linesearch* ls, const qn_oracle* oracle, size_t max_iter_solver,
                size_t max_iter_line_search, qn_callback_fn callback, void* callback_user);

/* LineSearch::compute_step_len (line_search/mod.rs:14-23) on its own, as the reference's line-search tests call it
 * (backtracking.rs:65-113, morethuente.rs:303-352, morethuente_b.rs:330-385): step length along `direction` from `x_k`, where
 * (f_k, g_k) is the evaluation at x_k.  Runs the device state machine from the line search's first statement to its return;
 * `ls` is updated as by qn_minimize (MoreThuenteB keeps its clipped t_max). */
int qn_compute_step_len(qn_context* ctx, qn_linesearch* ls, const double* x_k_host, double f_k, const double* g_k_host,
                        const double* direction_host, size_t n, const qn_oracle* oracle, size_t max_iter, double* step_out);

/* getters generated by derive_getters on bfgs.rs:3-12, plus LineSearchSolver::xk/k (bfgs.rs:52-63) */
size_t qn_solver_n(const qn_solver* s);
size_t qn_solver_k(const qn_solver* s);                    /* k() */
int qn_solver_set_k(qn_solver* s, size_t k);               /* k_mut() (bfgs.rs:58-63) */
double qn_solver_tol(const qn_solver* s);                  /* tol() */
int qn_solver_get_x(qn_solver* s, double* out_host);       /* x() / xk() */
int qn_solver_set_x(qn_solver* s, const double* x_host);   /* xk_mut() */
int qn_solver_s_norm(qn_solver* s, double* out, int* is_some); /* s_norm(): Option<f64> */
int qn_solver_y_norm(qn_solver* s, double* out, int* is_some); /* y_norm(): Option<f64> */
int qn_solver_next_iterate_too_close(qn_solver* s, int* out);          /* bfgs.rs:15-20 */
int qn_solver_gradient_next_iterate_too_close(qn_solver* s, int* out); /* bfgs.rs:21-26 */
int qn_solver_decrement_squared(qn_solver* s, double* out, int* is_some); /* Newton: decrement_squared(): Option<f64>, newton/mod.rs:10 */
/* approx_inv_hessian(): column-major n x n.  Lazy: applies the pending rank-2 update and gathers this rank's
 * rows; rows of other ranks are left untouched unless `all_ranks` (then an all-gather fills everything). */
int qn_solver_get_inv_hessian(qn_solver* s, double* out_colmajor_host, int all_ranks);
int qn_solver_set_inv_hessian(qn_solver* s, const double* h_colmajor_host);
/* ComputeDirection::compute_direction on its own (bfgs.rs:42-49, dfp.rs:42-49: -H g; gradient_descent.rs:24-30: -g), for a
 * binding that implements the trait; qn_minimize computes its directions on the device and never calls this. */
int qn_solver_compute_direction(qn_solver* s, const double* g_host, double* d_host);
/* The inverse-Hessian half of the `update_next_iterate` hook on its own (bfgs.rs:92-130, dfp.rs:92-118), again for a binding
 * that implements LineSearchSolver hook by hook: records s_norm / y_norm, skips the update when either is below tol
 * (bfgs.rs:103-109), else applies the BFGS / DFP secant update to the device-resident matrix. */
int qn_solver_secant_update(qn_solver* s, const double* s_host, const double* y_host);

/* ---- build-side instrumentation (not in the reference) ---- */
#define QN_TRACE_LS_MODIFIED (1 << 30)
typedef struct {
    double f, gnorm, t, s_norm, y_norm; /* f(x_k), ||g_k|| (inf-norm for gradient descent), step, ||s||, ||y|| */
    int32_t n_evals;   /* oracle calls of the reference's sequence in this iteration (memoised ones included) */
    int32_t ls_iters;  /* line-search inner iterations started */
    int32_t ls_cases;  /* More-Thuente: base-8 digits, one per inner iteration: 1..4 trial case, 0 returned;
                        * bit 30 (QN_TRACE_LS_MODIFIED): the modified-updating switch (morethuente.rs:212-215) was thrown */
    int32_t updated;   /* 1 if the inverse Hessian was updated */
} qn_trace_rec;
/* record up to `cap` iterations of the next qn_minimize calls; x_trace (optional) gets x_{k+1} rows */
int qn_solver_set_trace(qn_solver* s, size_t cap, int with_x);
int qn_solver_get_trace(qn_solver* s, qn_trace_rec* out_host, size_t cap, size_t* len, double* x_trace_host);

typedef struct {
    uint64_t iterations;       /* outer iterations completed by the last qn_minimize */
    uint64_t oracle_calls;     /* calls of the reference's sequence */
    uint64_t oracle_evals;     /* evaluations actually performed (distinct points when memoize = 1) */
    uint64_t h_passes;         /* passes over the inverse Hessian */
    uint64_t h_bytes;          /* algorithmic bytes moved over H by those passes (this rank) */
    uint64_t obj_bytes;        /* algorithmic bytes read from the objective's matrix (this rank) */
    uint64_t launches;         /* kernels enqueued */
    uint64_t host_syncs;       /* stream synchronisations */
    double   t_hpass_ms, t_eval_ms, t_ctl_ms, t_comm_ms; /* HIP-event time per kernel class (profiling mode) */
    uint64_t n_hpass_timed, n_eval_timed, n_ctl_timed, n_comm_timed;
    uint64_t matrix_bytes_per_pass; /* bytes of H (or Q) one pass of the last run streams on this rank: the rank's rows, or -- on the
                                       symmetric-storage path -- the 128 x 128 tiles of the symmetric half this rank owns */
    /* The counters above describe the LAST qn_minimize call (the device control block restarts them with k, ls_solver.rs:74).
     * These accumulate over every qn_minimize call made on this solver, so a harness that makes several calls (warm-up,
     * timed region, restarts after convergence) can difference them. */
    uint64_t total_minimize_calls, total_iterations, total_oracle_calls, total_oracle_evals, total_h_passes, total_h_bytes,
        total_obj_bytes;
    uint32_t path; /* QN_PATH_* flags of the last call: which kernels serviced it */
    uint32_t _pad;
    /* profiling mode, symmetric-storage path: the small second launch of a pass (slot sums + epilogue), timed on its own;
     * t_hpass_ms / t_eval_ms then hold the tile kernels alone */
    double   t_hreduce_ms, t_ereduce_ms;
    uint64_t n_hreduce_timed, n_ereduce_timed;
    /* row-sharded runs: collectives enqueued by every qn_minimize call of this solver so far -- of n-vectors (mat-vec partials:
     * the all-reduce north_star names, as an all-gather + rank-order sum or as ncclAllReduce), and of per-workgroup scalars
     * (8 KB per rank: what an evaluation hands the line search on the second-generation path) */
    uint64_t total_xchg_vector, total_xchg_scalar;
    /* Newton (newton/mod.rs:26-49), profiling mode: HIP-event time of the direction requests (staging, factorisation, the four
     * triangular sweeps of d = -H^-1 g and of the decrement's second solve) */
    double   t_newton_ms;
    uint64_t n_newton_timed;
    /* pivoted-LU path: times a bounded wait of the one-launch panel / sweep kernels expired and the factorisation was run again
     * with one launch per step (a co-tenant on the GPU can do that; the result is the same, the iteration slower) */
    uint64_t newton_lu_sync_timeouts;
} qn_stats;
#define QN_PATH_FUSED 1u       /* fused fast path (device quadratic, memoised): no kernel but the streaming ones touches an n-vector */
#define QN_PATH_SYM 2u         /* ... on the symmetric half of H and Q only */
#define QN_PATH_SYM_GENERIC 4u /* generic path whose H pass runs on the symmetric half */
#define QN_PATH_PIPELINED 8u   /* predicated kernels enqueued ahead of the device-side decisions (no host sync per step) */
#define QN_PATH_SYM2 16u       /* QN_PATH_SYM with the solver's decisions taken in every kernel's prologue (5 launches per iteration) */
#define QN_PATH_TILES1 32u     /* QN_PATH_SYM2 whose update pass streams H through the first-generation tile kernel (one workgroup per tile) behind a
                                  one-workgroup launch that runs the state machine: H's share past the Infinity Cache, and the log-sum-exp objective */
int qn_solver_get_stats(qn_solver* s, qn_stats* out);
/* profiling != 0: bracket every launch with HIP events on the solver's stream (slower; for roofline reports) */
int qn_solver_set_profiling(qn_solver* s, int on);
/* 0 = pipelined (device-resident control, no host sync per decision; default for memoised device objectives),
 * 1 = synchronous (host reads the control block after every step; always used with host oracles / callbacks) */
int qn_solver_set_sync_mode(qn_solver* s, int sync);
/* tuning of the fused ROW kernels: rows per workgroup tile (2, 4, 8 or 16), column splits (1 .. 64); 0 keeps what is set.  (ABI <= 4 also took
 * negative `rows_per_block` codes that selected diagnostic paths: those are the named options below since ABI 5.) */
int qn_solver_set_tiling(qn_solver* s, int rows_per_block, int col_splits);
/* Named options (ABI 5).  None of them is needed to USE the library -- the reference has no counterpart: ls_solver.rs:66-111 takes a solver, a line
 * search and a closure -- they select, for tests and measurements, a path the library would not take by itself, or switch a step of the default path
 * off.  value != 0 switches the named thing ON, 0 OFF; [default] in brackets.  Unknown option: QN_ERROR_INPUT_PARAMS. */
typedef enum {
    QN_OPT_GENERIC_KERNELS = 1,            /* [0] the generic path: every O(n) vector operation in the one-workgroup control kernel, synchronous requests */
    QN_OPT_DEFERRED_UPDATE_STEP = 2,       /* [1] fused row kernels: the accepting step also opens the next line search (one launch less per iteration) */
    QN_OPT_SYMMETRIC_STORAGE = 3,          /* [1] stream only the upper block triangle of H and Q; 0: the fused ROW kernels on the full matrices */
    QN_OPT_SECOND_GENERATION = 4,          /* [1] the state machine in every kernel's prologue (5 launches per iteration); 0: first-generation tile kernels */
    QN_OPT_FOLDED_ACCEPT_REDUCE = 5,       /* [0] the accept-reduce inside the update-tile launch (n <= 4096; measured neutral) */
    QN_OPT_ROW_SLIVERS = 6,                /* [1] left-over diagonal tiles cut into 8-row slivers, one per workgroup (n = 4096); 0: whole tiles only */
    QN_OPT_EVAL_PAIR_INSTANCE = 7,         /* [1] the evaluation kernel's two-items-and-a-sliver instance where the work lists allow it; 0: the general body */
    QN_OPT_EVAL_MOVER_MULTIPLIER = 8,      /* [1] ... as mover + multiplier waves (round 6, csrc/qn_sym2r.hip.h); 0: round 5's kernel -- the same bits */
    QN_OPT_TAIL_REDUCE = 9,                /* [0] the update-reduce in the tail of the update-tile launch (bit-identical, measured slower) */
    QN_OPT_BOUNDED_SECOND_GENERATION = 10, /* [1] BFGSB / DFPB / SR1B with More-Thuente(B) on the second-generation path; 0: the generic path */
    QN_OPT_NEWTON_PIVOTED_LU = 11,         /* [0] Newton: the pivoted LU the reference's try_inverse stands for (newton/mod.rs:31-40) even for an SPD Hessian */
    QN_OPT_LU_PER_COLUMN_PANEL = 12,       /* [0] ... its panel with two launches per column (rounds 1-2) */
    QN_OPT_LU_LOOKAHEAD = 13,              /* [1] ... look-ahead on a second stream; 0: one stream */
    QN_OPT_LU_ONE_LAUNCH_PANEL = 14,       /* [1] ... the panel and the substitution sweeps as one launch each; 0: one launch per sub-panel / block */
    QN_OPT_LU_FORCE_WAIT_EXPIRY = 15,      /* [0] ... every bounded wait of the one-launch kernels expires at once (exercises the fallback) */
    QN_OPT_CHUNKS_PER_TRIP = 16,           /* [1] fused ROW kernels: column chunks per loop trip (1, 2 or 4) */
    QN_OPT_LU_SPLIT_ROLE_A = 17,           /* [4] ... the one-launch panel's pivot chain shared by 1 (0 too), 2 or 4 workgroups of one XCD (csrc/qn_lu_split.hip.h) -- the same bits */
    QN_OPT_LU_SPLIT_MIN_ROWS = 18,         /* [4160] ... for panels of at least `value` rows (a NUMBER, not a switch; shorter panels: one workgroup) */
    QN_OPT_BTB_PROJECT_IN_EVAL = 19,       /* [1] BackTrackingB on the second-generation path: the trial point projected inside the evaluation kernel (n = 4096's mover + multiplier kernel); 0: a projection launch per trial -- the same bits */
    QN_OPT_EVAL_ZIGZAG = 20,               /* [1] the mover + multiplier kernel streams its two tiles in the other order in launches of odd parity: an evaluation launch right behind another one starts with the tile the XCD's L2 still holds (csrc/qn_sym2r.hip.h, ZIG-ZAG); 0: the same order in every launch -- the same bits */
    QN_OPT_TOUCH_H_ROWS = 21,              /* [8] n = 4096: the accept-reduce launch carries workgroups that only load the first `value` rows (of a wave's 16; 0, 4, 6, 8, 10, 12 or 16) of the tile the update-tile launch's workgroup of the same index streams first -- into the L2 of the XCD both run on (csrc/qn_sym2.hip.h, TOUCH WORKGROUPS); a NUMBER; loads only -- the same bits */
    QN_OPT_TOUCH_Q_ROWS = 22               /* [6] ... and the update-reduce launch for the evaluation launch behind it (rows of Q's tiles) */
} qn_option;
int qn_solver_set_option(qn_solver* s, int option, int value);

/* ---------------------------------------------------------------------------------------------
 * Thin kernel-level FFI (what a Rust host loop would bind if it keeps `minimize` in Rust):
 * the BLAS-2 / rank-2 / vector primitives as individual calls on device buffers.
 * ------------------------------------------------------------------------------------------- */
int qn_dev_alloc(qn_context* ctx, size_t bytes, void** out_dev);
int qn_dev_free(qn_context* ctx, void* dev);
int qn_h2d(qn_context* ctx, void* dst_dev, const void* src_host, size_t bytes);
int qn_d2h(qn_context* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* y = A x for `nrows` rows of a row-major matrix with leading dimension ld (bfgs.rs:47 H*g, dfp.rs:118 H*y) */
int qn_gemv(qn_context* ctx, const double* a_dev, size_t ld, size_t nrows, size_t ncols, const double* x_dev, double* y_dev);
/* H_rows += c_su (s u' + u s') + c_ss s s' + c_uu u u' on rows [row0,row0+nrows) (bfgs.rs:115-124, dfp.rs:115-120) */
int qn_rank2_update(qn_context* ctx, double* h_dev, size_t ld, size_t row0, size_t nrows, size_t n, const double* s_dev,
                    const double* u_dev, double c_ss, double c_su, double c_uu);
/* out = x + t*d, two roundings per element (ls_solver.rs:60, bfgs.rs:94, backtracking.rs:32, morethuente.rs:182) */
int qn_axpy(qn_context* ctx, size_t n, const double* x_dev, double t, const double* d_dev, double* out_dev);
int qn_dot(qn_context* ctx, size_t n, const double* a_dev, const double* b_dev, double* out_host);  /* mod.rs:35,47,55 */
int qn_nrm2(qn_context* ctx, size_t n, const double* a_dev, double* out_host);                      /* bfgs.rs:74,97,99 */

#ifdef __cplusplus
}
#endif
#endif /* QN_HIP_H */
