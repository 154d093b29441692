/*
 * qn_oracle.h -- CPU restatement of the reference's quasi-Newton / line-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the timed CPU baseline.  The product (libqn_hip.so) never links or calls it.
 *
 * PARITY STATUS: **parity unpinned** at the iterate level.  The reference is a Rust crate whose
 * arithmetic lives in nalgebra 0.33.2 / matrixmultiply 0.3.9 (Cargo.lock:360-361,398-399), which
 * are not vendored under /root/reference, and no Rust toolchain exists in this image, so the
 * reference cannot be executed here.  This restatement follows the reference sources line by line
 * (file:line cited at each function) and the nalgebra operation order as recalled in SURVEY.md
 * section 8(a); it is pinned only by the reference's own known-answer tests.
 */
#ifndef QN_ORACLE_H
#define QN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SolverError, src/ls_solver.rs:10-20 (0 = Ok(())) */
enum {
    QO_OK = 0,
    QO_MAX_ITER_REACHED = 1,
    QO_OUT_OF_DOMAIN = 2,
    QO_ERROR_INPUT_PARAMS = 3,
    QO_ABNORMAL_TERMINATION = 4
};

/* solver families on the path (src/quasi_newton/bfgs.rs, dfp.rs, src/steepest_descent/gradient_descent.rs) */
enum { QO_BFGS = 0, QO_DFP = 1, QO_GRADIENT_DESCENT = 2, QO_NEWTON = 3 /* src/newton/mod.rs */,
       QO_SR1 = 4 /* src/quasi_newton/sr1_b.rs (with bounds: SR1B) */ };

/* how the inverse-Hessian update is evaluated */
enum {
    QO_UPDATE_AS_WRITTEN = 0, /* bfgs.rs:115-124 / dfp.rs:115-120 literally: dense n x n products, O(n^3) */
    QO_UPDATE_RANK2 = 1       /* algebraically identical symmetric rank-2 form, O(n^2) (what the GPU path computes) */
};

enum { QO_LS_MORETHUENTE = 0, QO_LS_BACKTRACKING = 1,
       QO_LS_MORETHUENTE_B = 2 /* morethuente_b.rs */, QO_LS_BACKTRACKING_B = 3 /* backtracking_b.rs */ };

/* MoreThuente, src/line_search/morethuente.rs:6-28 ; BackTracking, src/line_search/backtracking.rs:3-11 */
typedef struct {
    int kind;
    /* More-Thuente */
    double c1, c2, t_min, t_max, delta_min, delta, delta_max;
    /* Backtracking */
    double bt_c1, bt_beta;
    /* the *_B variants: box bounds held by the line search itself (morethuente_b.rs:14-15, backtracking_b.rs:7-8) */
    const double *lower_bound, *upper_bound;
} qo_linesearch;

void qo_morethuente_default(qo_linesearch* ls);                 /* morethuente.rs:16-28 */
void qo_backtracking_new(qo_linesearch* ls, double c1, double beta); /* backtracking.rs:8-10 */

/* the oracle closure: impl FnMut(&DVector<f64>) -> FuncEvalMultivariate (ls_solver.rs:69).
 * Writes f and g[n]; return value is ignored (the reference closure cannot fail). */
typedef int (*qo_oracle_fn)(void* user, const double* x, size_t n, double* f, double* g);

typedef struct qo_solver qo_solver;

/* Newton only: the Hessian part of FuncEvalMultivariate (func_eval.rs:8,27-33), column-major n x n, evaluated at the
 * loop-top point right after the oracle call (newton/mod.rs:31-35 `.expect("Hessian not available in the oracle")`). */
typedef int (*qo_hessian_fn)(void* user, const double* x, size_t n, double* h_colmajor);
void qo_solver_set_hessian_fn(qo_solver* s, qo_hessian_fn fn, void* user);
int qo_solver_decrement_squared(const qo_solver* s, double* out); /* Option<f64>, newton/mod.rs:10 */

/* callback: Option<&mut dyn FnMut(&Self)> (ls_solver.rs:72,105-107) */
typedef void (*qo_callback_fn)(void* user, const qo_solver* solver);

#define QO_LS_MODIFIED_BIT (1 << 30)
/* per-outer-iteration trace record (build-side instrumentation, not in the reference) */
typedef struct {
    double f;        /* f(x_k) at loop top */
    double gnorm;    /* ||g(x_k)||_2 */
    double t;        /* step length returned by the line search */
    double s_norm, y_norm;
    int32_t n_evals; /* oracle calls made during this outer iteration (loop-top call included) */
    int32_t ls_iters;/* line-search inner iterations started */
    int32_t ls_cases;/* More-Thuente: base-8 digits, one per inner iteration: 1..4 = trial case, 0 = returned;
                      * bit 30 (QO_LS_MODIFIED_BIT): the modified-updating switch (morethuente.rs:212-215) was thrown */
    int32_t updated; /* 1 if the inverse Hessian was updated */
} qo_trace_rec;

typedef struct {
    qo_trace_rec* rec; /* caller-owned array */
    size_t cap;        /* capacity */
    size_t len;        /* filled */
    double* xs;        /* optional: cap*n iterates x_{k+1} after each iteration (may be NULL) */
    size_t n_oracle_calls;
} qo_trace;

/* BFGS::new / DFP::new / GradientDescent::new (bfgs.rs:27-39, dfp.rs, gradient_descent.rs:14-21) */
qo_solver* qo_solver_create(int method, double tol, const double* x0, size_t n, int update_mode, int nthreads);
void qo_solver_destroy(qo_solver* s);
/* BFGSB / DFPB / SR1B (bfgs_b.rs:43-63): box bounds; the current x is projected onto them, directions become
 * P(x - H g) - x (bfgs_b.rs:66-77) */
void qo_solver_set_bounds(qo_solver* s, const double* lower_bound, const double* upper_bound);

/* LineSearchSolver::minimize (ls_solver.rs:66-111) */
int qo_minimize(qo_solver* s, const qo_linesearch* ls, qo_oracle_fn oracle, void* oracle_user,
                size_t max_iter_solver, size_t max_iter_line_search,
                qo_callback_fn callback, void* callback_user, qo_trace* trace);

/* LineSearch::compute_step_len on its own (line_search/mod.rs:14-23), as used by backtracking.rs:65-113
 * and morethuente.rs:303-352.  f0/g0 = eval at x. */
double qo_compute_step_len(const qo_linesearch* ls, const double* x, double f0, const double* g0,
                           const double* d, size_t n, qo_oracle_fn oracle, void* oracle_user, size_t max_iter);

/* getters (derive_getters on bfgs.rs:3-12) */
size_t qo_solver_n(const qo_solver* s);
size_t qo_solver_k(const qo_solver* s);
const double* qo_solver_x(const qo_solver* s);
const double* qo_solver_inv_hessian(const qo_solver* s); /* column-major n x n, NULL for gradient descent */
double qo_solver_bytes_streamed(const qo_solver* s);    /* matrix bytes the solver's own sweeps moved so far (mat-vecs with H: 8 n^2 each, the rank-2 update: 16 n^2) */
int qo_solver_s_norm(const qo_solver* s, double* out);   /* returns 0 for Option::None */
int qo_solver_y_norm(const qo_solver* s, double* out);
void qo_solver_set_inv_hessian(qo_solver* s, const double* h_colmajor);

/* nalgebra-order primitives (exposed so tests can compare device kernels one op at a time) */
double qo_dot(const double* a, const double* b, size_t n);                  /* 8-accumulator dot */
double qo_norm(const double* a, size_t n);                                   /* sqrt(dot(a,a)) */
void qo_gemv_colsweep(const double* a_colmajor, const double* x, double* y, size_t n); /* y = A x */
void qo_axpy_new(const double* x, double t, const double* d, double* out, size_t n);    /* out = x + t*d */

/* ---- benchmark objective (build-defined): f = 1/2 x'Qx - b'x, g = Qx - b ---- */
typedef struct {
    size_t n;
    const double* q; /* row-major n x n (symmetric) */
    const double* b;
    int nthreads;
    size_t calls;
} qo_quadratic;
int qo_quadratic_eval(void* user /* qo_quadratic* */, const double* x, size_t n, double* f, double* g);
int qo_quadratic_hessian(void* user /* qo_quadratic* */, const double* x, size_t n, double* h_colmajor); /* = Q */

/* synthetic SPD generator: off-diagonal Q_ij = Q_ji = u(seed, min(i,j), max(i,j)) * inv_n with u in [-1,1),
 * diagonal supplied by the caller.  Fills rows [row0, row0+nrows) of a row-major block with leading dim n. */
double qo_synth_u(uint64_t seed, uint64_t i, uint64_t j);
void qo_synth_fill_rows(double* q_rows, size_t n, size_t row0, size_t nrows, uint64_t seed,
                        const double* diag, int nthreads);

/* log-sum-exp objective: f = log sum_i exp(a_i'x + c_i) + mu/2 ||x||^2 */
typedef struct {
    size_t m, n;
    const double* a; /* row-major m x n */
    const double* c; /* m */
    double mu;
    int nthreads;
    size_t calls;
} qo_logsumexp;
int qo_logsumexp_eval(void* user /* qo_logsumexp* */, const double* x, size_t n, double* f, double* g);

int qo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
