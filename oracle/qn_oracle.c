/*
 * qn_oracle.c -- CPU restatement of the reference's BFGS / DFP / line-search path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see qn_oracle.h).  PARITY STATUS: parity unpinned beyond the reference's
 * own known-answer tests -- the Rust reference cannot be built or run in this image.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * "[nalgebra]" marks operation orders that live in the un-vendored nalgebra 0.33.2 crate and are
 * restated from its published algorithm: column-sweep gemv, 8-accumulator
 * dot, norm = sqrt(dot), single-rounded outer products, no FMA contraction (build with
 * -ffp-contract=off), matrix products through per-column gemv.  For n > 5 nalgebra hands n x n
 * products to matrixmultiply::dgemm whose blocked summation order is not reproduced here.
 */
#include "qn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int qo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* [nalgebra] vector primitives                                                                */
/* ------------------------------------------------------------------------------------------ */

/* [nalgebra] Matrix::dot on a dynamic column: 8 strided accumulators, folded as
 * res += a0+a4; res += a1+a5; res += a2+a6; res += a3+a7; then a sequential scalar tail.
 * Call sites: line_search/mod.rs:35,47,55 ; morethuente.rs:137 ; bfgs.rs:74,97,99,115 ; dfp.rs:117-118 */
double qo_dot(const double* a, const double* b, size_t n) {
    double res = 0.0;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    size_t i = 0;
    while (n - i >= 8) {
        a0 += a[i + 0] * b[i + 0];
        a1 += a[i + 1] * b[i + 1];
        a2 += a[i + 2] * b[i + 2];
        a3 += a[i + 3] * b[i + 3];
        a4 += a[i + 4] * b[i + 4];
        a5 += a[i + 5] * b[i + 5];
        a6 += a[i + 6] * b[i + 6];
        a7 += a[i + 7] * b[i + 7];
        i += 8;
    }
    res += a0 + a4;
    res += a1 + a5;
    res += a2 + a6;
    res += a3 + a7;
    for (; i < n; ++i) res += a[i] * b[i];
    return res;
}

/* [nalgebra] Matrix::norm = sqrt(norm_squared) = sqrt(dot(v, v)).  bfgs.rs:74,97,99 */
double qo_norm(const double* a, size_t n) { return sqrt(qo_dot(a, a, n)); }

/* [nalgebra] Matrix * Vector -> gemv: y = A[:,0]*x0 ; y += A[:,j]*x_j for j = 1..n-1.
 * Each y_i is a strict left-to-right sum over j of fl(A_ij * x_j).  bfgs.rs:47, dfp.rs:118 */
void qo_gemv_colsweep(const double* a, const double* x, double* y, size_t n) {
    if (n == 0) return;
    for (size_t i = 0; i < n; ++i) y[i] = a[i] * x[0];
    for (size_t j = 1; j < n; ++j) {
        const double* col = a + j * n;
        const double xj = x[j];
        for (size_t i = 0; i < n; ++i) y[i] = col[i] * xj + y[i];
    }
}

/* same sums as qo_gemv_colsweep, evaluated row-block-parallel (each y_i keeps its j order) */
static void gemv_colmajor_mt(const double* a, const double* x, double* y, size_t n, int nthreads) {
    if (nthreads <= 1 || n < 256) {
        qo_gemv_colsweep(a, x, y, n);
        return;
    }
#pragma omp parallel num_threads(nthreads)
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int tid = 0, nt = 1;
#endif
        size_t chunk = (n + (size_t)nt - 1) / (size_t)nt;
        size_t i0 = (size_t)tid * chunk, i1 = i0 + chunk > n ? n : i0 + chunk;
        if (i0 < i1) {
            for (size_t i = i0; i < i1; ++i) y[i] = a[i] * x[0];
            for (size_t j = 1; j < n; ++j) {
                const double* col = a + j * n;
                const double xj = x[j];
                for (size_t i = i0; i < i1; ++i) y[i] = col[i] * xj + y[i];
            }
        }
    }
}

/* The same sums once more for a matrix that is SYMMETRIC BIT FOR BIT (H under the rank-2 form, the benchmark's Q): a_ij is
 * also stored at a[j + i*n], so y_i = sum_j a[j + i*n] x_j sweeps row i = column i contiguously, in the same strict left-to-right
 * order over j and with the same two roundings per term (no FMA): bit-identical to the column sweep, but every thread streams
 * its own contiguous block of memory instead of striding through all n columns.  Four rows per thread are in flight (four
 * independent dependent-add chains; a single chain is latency-bound at one element per ~4 cycles).  This is what makes the
 * OpenMP port a BANDWIDTH-bound CPU baseline (CPU-B) rather than a cache-miss-bound one. */
static void gemv_symmetric_rows_mt(const double* a, const double* x, double* y, size_t n, int nthreads) {
    if (n == 0) return;
#pragma omp parallel for num_threads(nthreads) schedule(static) if (nthreads > 1 && n >= 256)
    for (size_t ib = 0; ib < (n + 3) / 4; ++ib) {
        const size_t i = 4 * ib;
        if (i + 4 <= n) {
            const double *r0 = a + i * n, *r1 = r0 + n, *r2 = r1 + n, *r3 = r2 + n;
            double a0 = r0[0] * x[0], a1 = r1[0] * x[0], a2 = r2[0] * x[0], a3 = r3[0] * x[0];
            for (size_t j = 1; j < n; ++j) {
                const double xj = x[j];
                a0 = r0[j] * xj + a0; a1 = r1[j] * xj + a1; a2 = r2[j] * xj + a2; a3 = r3[j] * xj + a3;
            }
            y[i] = a0; y[i + 1] = a1; y[i + 2] = a2; y[i + 3] = a3;
        } else {
            for (size_t k = i; k < n; ++k) {
                const double* r = a + k * n;
                double acc = r[0] * x[0];
                for (size_t j = 1; j < n; ++j) acc = r[j] * x[j] + acc;
                y[k] = acc;
            }
        }
    }
}

/* x + t*d : `step * direction` materialises fl(t*d_i), then the sum rounds again.
 * ls_solver.rs:60 ; bfgs.rs:94 ; backtracking.rs:32 ; morethuente.rs:182,217,276 */
void qo_axpy_new(const double* x, double t, const double* d, double* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        double td = t * d[i];
        out[i] = x[i] + td;
    }
}

/* [nalgebra] general n x n product C = A*B through per-column gemv (column-major):
 * C[:,j] = A[:,0]*B[0,j] ; C[:,j] += A[:,k]*B[k,j].  Exact order for n <= 5; for n > 5 the reference
 * uses matrixmultiply::dgemm (order not reproduced).  bfgs.rs:123-124 ; dfp.rs:119-120 */
static void matmul_colmajor(const double* a, const double* b, double* c, size_t n, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static) if (nthreads > 1 && n >= 64)
    for (size_t j = 0; j < n; ++j) {
        double* cj = c + j * n;
        const double b0 = b[j * n];
        for (size_t i = 0; i < n; ++i) cj[i] = a[i] * b0;
        for (size_t k = 1; k < n; ++k) {
            const double* ak = a + k * n;
            const double bkj = b[k + j * n];
            for (size_t i = 0; i < n; ++i) cj[i] = ak[i] * bkj + cj[i];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* line searches                                                                                */
/* ------------------------------------------------------------------------------------------ */

void qo_morethuente_default(qo_linesearch* ls) { /* morethuente.rs:16-28 */
    memset(ls, 0, sizeof(*ls));
    ls->kind = QO_LS_MORETHUENTE;
    ls->c1 = 1e-4;
    ls->c2 = 0.9;
    ls->t_min = 0.0;
    ls->t_max = INFINITY;
    ls->delta_min = 0.58333333;
    ls->delta = 0.66;
    ls->delta_max = 1.1;
}

void qo_backtracking_new(qo_linesearch* ls, double c1, double beta) { /* backtracking.rs:8-10 */
    memset(ls, 0, sizeof(*ls));
    ls->kind = QO_LS_BACKTRACKING;
    ls->bt_c1 = c1;
    ls->bt_beta = beta;
}

typedef struct {
    qo_oracle_fn fn;
    void* user;
    size_t calls;
} oracle_t;

static void call_oracle(oracle_t* o, const double* x, size_t n, double* f, double* g) {
    o->fn(o->user, x, n, f, g);
    o->calls++;
}

typedef struct {
    int32_t iters;
    int32_t cases;
    int32_t ndigits;
} ls_stats;

static void ls_push_case(ls_stats* st, int c) {
    if (!st) return;
    if (st->ndigits < 10) {
        int32_t mul = 1;
        for (int i = 0; i < st->ndigits; ++i) mul *= 8;
        st->cases += mul * c;
    }
    st->ndigits++;
}

/* SufficientDecreaseCondition::sufficient_decrease, line_search/mod.rs:27-36:
 *   f_kp1 - f_k <= c1 * t * grad_k.dot(direction_k)     (the dot is recomputed on every call) */
static int sufficient_decrease(double c1, double f_k, double f_kp1, const double* g_k, double t,
                               const double* d, size_t n) {
    return f_kp1 - f_k <= c1 * t * qo_dot(g_k, d, n);
}

/* CurvatureCondition::strong_curvature_condition, line_search/mod.rs:49-56 */
static int strong_curvature(double c2, const double* g_k, const double* g_kp1, const double* d, size_t n) {
    return fabs(qo_dot(g_kp1, d, n)) <= c2 * fabs(qo_dot(g_k, d, n));
}

/* BackTracking::compute_step_len, backtracking.rs:20-58 */
static double backtracking_step(const qo_linesearch* ls, const double* x, double f0, const double* g0,
                                const double* d, size_t n, oracle_t* o, size_t max_iter, ls_stats* st,
                                double* xt, double* gt) {
    double t = 1.0;
    size_t i = 0;
    while (max_iter > i) {
        if (st) st->iters++;
        qo_axpy_new(x, t, d, xt, n); /* :32 */
        double ft;
        call_oracle(o, xt, n, &ft, gt); /* :34 */
        if (isnan(ft) || isinf(ft)) {   /* :37-41 shrink WITHOUT consuming an iteration */
            t *= ls->bt_beta;
            continue;
        }
        if (sufficient_decrease(ls->bt_c1, f0, ft, g0, t, d, n)) return t; /* :44-47 */
        t *= ls->bt_beta; /* :50 */
        i += 1;           /* :51 */
    }
    return t; /* :54 */
}

/* MoreThuente::update_interval, morethuente.rs:64-91 */
static int mt_update_interval(double f_tl, double f_t, double g_t, double* tl, double t, double* tu) {
    if (f_t > f_tl) {
        *tu = t;
        return 0;
    } else if (g_t * (*tl - t) > 0.) {
        *tl = t;
        return 0;
    } else if (g_t * (*tl - t) < 0.) {
        *tu = *tl;
        *tl = t;
        return 0;
    }
    return 1;
}

/* MoreThuente::cubic_minimizer, morethuente.rs:93-108 */
static double mt_cubic_minimizer(double ta, double tb, double f_ta, double f_tb, double g_ta, double g_tb) {
    double s = 3. * (f_tb - f_ta) / (tb - ta);
    double z = s - g_ta - g_tb;
    double w = sqrt(z * z - g_ta * g_tb); /* z.powi(2) == z*z ; sqrt of a negative is NaN and flows on */
    return ta + ((tb - ta) * ((w - g_ta - z) / (g_tb - g_ta + 2. * w)));
}

/* MoreThuente::quadratic_minimzer_1 [sic], morethuente.rs:110-121 */
static double mt_quadratic_minimizer_1(double ta, double tb, double f_ta, double f_tb, double g_ta) {
    double lin_int = (f_ta - f_tb) / (ta - tb);
    return ta - 0.5 * ((ta - tb) * g_ta / (g_ta - lin_int));
}

/* MoreThuente::quadratic_minimizer_2, morethuente.rs:123-132 */
static double mt_quadratic_minimizer_2(double ta, double tb, double g_ta, double g_tb) {
    return ta - g_ta * ((ta - tb) / (g_ta - g_tb));
}

/* MoreThuente::compute_step_len, morethuente.rs:165-297 (quirks of SURVEY.md 3.2 kept as they are) */
static double morethuente_step(const qo_linesearch* ls, const double* x, double f0, const double* g0,
                               const double* d, size_t n, oracle_t* o, size_t max_iter, ls_stats* st,
                               double* xt, double* gt, double* xw, double* gw) {
    int use_modified_updating = 0;
    int interval_converged = 0;
    double t = fmin(fmax(1.0, ls->t_min), ls->t_max); /* :176 Rust max/min return the non-NaN operand, as fmax/fmin */
    double tl = ls->t_min;
    double tu = ls->t_max;

    for (size_t i = 0; i < max_iter; ++i) {
        if (st) st->iters++;
        double f_et;
        qo_axpy_new(x, t, d, xt, n);
        call_oracle(o, xt, n, &f_et, gt); /* :182 */
        /* :184-193 strong Wolfe (mod.rs:72-83), short-circuit */
        if (sufficient_decrease(ls->c1, f0, f_et, g0, t, d, n) && strong_curvature(ls->c2, g0, gt, d, n)) {
            ls_push_case(st, 0);
            return t;
        } else if (interval_converged) { /* :194 */
            ls_push_case(st, 0);
            return t;
        } else if (t == tl) { /* :198 */
            ls_push_case(st, 0);
            return t;
        } else if (t == tu) { /* :202 */
            ls_push_case(st, 0);
            return t;
        }
        /* :207-210 phi (134-139) and psi (140-149) */
        double phi_t_f = f_et, phi_t_g = qo_dot(gt, d, n);
        double phi_0_f = f0, phi_0_g = qo_dot(g0, d, n);
        double psi_t_f = phi_t_f - phi_0_f - ls->c1 * t * phi_0_g;
        double psi_t_g = phi_t_g - ls->c1 * phi_0_g;

        if (!use_modified_updating && psi_t_f <= 0. && phi_t_g > 0.) { /* :212-215 */
            use_modified_updating = 1;
            if (st) st->cases |= QO_LS_MODIFIED_BIT; /* instrumentation: the sticky switch was thrown in this line search */
        }

        double f_etl;
        qo_axpy_new(x, tl, d, xw, n);
        call_oracle(o, xw, n, &f_etl, gw); /* :217 re-evaluated on every inner iteration */
        double phi_tl_f = f_etl, phi_tl_g = qo_dot(gw, d, n);

        double f_tl, g_tl, f_t, g_t; /* :221-226 */
        if (use_modified_updating) {
            f_tl = phi_tl_f; g_tl = phi_tl_g; f_t = phi_t_f; g_t = phi_t_g;
        } else {
            f_tl = phi_tl_f - phi_0_f - ls->c1 * tl * phi_0_g;
            g_tl = phi_tl_g - ls->c1 * phi_0_g;
            f_t = psi_t_f; g_t = psi_t_g;
        }

        if (f_t > f_tl) { /* case 1, :230-241 */
            double tc = mt_cubic_minimizer(tl, t, f_tl, f_t, g_tl, g_t);
            double tq = mt_quadratic_minimizer_1(tl, t, f_tl, f_t, g_tl);
            ls_push_case(st, 1);
            if (fabs(tc - tl) < fabs(tq - tl)) t = tc; else t = 0.5 * (tq + tc);
        } else if (g_t * g_tl < 0.) { /* case 2, :243-254 */
            double tc = mt_cubic_minimizer(tl, t, f_tl, f_t, g_tl, g_t);
            double ts = mt_quadratic_minimizer_2(tl, t, g_tl, g_t);
            ls_push_case(st, 2);
            if (fabs(tc - t) >= fabs(ts - t)) t = tc; else t = ts;
        } else if (fabs(g_t) <= fabs(g_tl)) { /* case 3, :256-272 */
            double tc = mt_cubic_minimizer(tl, t, f_tl, f_t, g_tl, g_t);
            double ts = mt_quadratic_minimizer_2(tl, t, g_tl, g_t);
            ls_push_case(st, 3);
            double t_plus = (fabs(tc - t) < fabs(ts - t)) ? tc : ts;
            if (t > tl) t = fmin(t_plus, t + ls->delta * (tu - t));
            else t = fmax(t_plus, t + ls->delta * (tu - t));
        } else { /* case 4, :274-287 (tu may be +inf: the oracle then sees non-finite input) */
            double f_etu;
            qo_axpy_new(x, tu, d, xw, n);
            call_oracle(o, xw, n, &f_etu, gw); /* :276 */
            double phi_tu_f = f_etu, phi_tu_g = qo_dot(gw, d, n);
            double f_tu, g_tu;
            if (use_modified_updating) {
                f_tu = phi_tu_f; g_tu = phi_tu_g;
            } else {
                f_tu = phi_tu_f - phi_0_f - ls->c1 * tu * phi_0_g;
                g_tu = phi_tu_g - ls->c1 * phi_0_g;
            }
            ls_push_case(st, 4);
            t = mt_cubic_minimizer(tu, t, f_t, f_tu, g_t, g_tu); /* :286 argument order as written */
        }
        t = fmin(fmax(t, ls->t_min), ls->t_max); /* :290 */
        /* :293 -- the NEW t with the OLD trial's f_t, g_t */
        interval_converged = mt_update_interval(f_tl, f_t, g_t, &tl, t, &tu);
    }
    return t; /* :295-296 */
}

/* BackTrackingB::compute_step_len, backtracking_b.rs:52-88 (the trial point is projected; modified Armijo rule :24-34) */
static double backtracking_b_step(const qo_linesearch* ls, const double* x, double f0, const double* d, size_t n, oracle_t* o,
                                  size_t max_iter, ls_stats* st, double* xt, double* gt) {
    double t = 1.0;
    size_t i = 0;
    while (max_iter > i) {
        if (st) st->iters++;
        qo_axpy_new(x, t, d, xt, n);
        for (size_t j = 0; j < n; ++j) xt[j] = fmin(fmax(xt[j], ls->lower_bound[j]), ls->upper_bound[j]); /* :67 */
        double ft;
        call_oracle(o, xt, n, &ft, gt);
        if (isnan(ft) || isinf(ft)) { t *= ls->bt_beta; continue; }
        /* sufficient_decrease_with_bounds: f - f0 <= (-c1 / t) * diff.dot(&diff) */
        for (size_t j = 0; j < n; ++j) gt[j] = xt[j] - x[j]; /* gt reused as diff */
        if (ft - f0 <= (-ls->bt_c1 / t) * qo_dot(gt, gt, n)) return t;
        t *= ls->bt_beta;
        i += 1;
    }
    return t;
}

static double compute_step_len(const qo_linesearch* ls, const double* x, double f0, const double* g0,
                               const double* d, size_t n, oracle_t* o, size_t max_iter, ls_stats* st,
                               double* work /* 4n */) {
    if (ls->kind == QO_LS_BACKTRACKING)
        return backtracking_step(ls, x, f0, g0, d, n, o, max_iter, st, work, work + n);
    if (ls->kind == QO_LS_BACKTRACKING_B)
        return backtracking_b_step(ls, x, f0, d, n, o, max_iter, st, work, work + n);
    if (ls->kind == QO_LS_MORETHUENTE_B) { /* morethuente_b.rs:185-201: t_max is clipped to the box -- and STAYS clipped (self.t_max) */
        double cand = INFINITY;
        for (size_t i = 0; i < n; ++i) {
            double v = INFINITY;
            if (d[i] > 0.0) v = (ls->upper_bound[i] - x[i]) / d[i];
            else if (d[i] < 0.0) v = (ls->lower_bound[i] - x[i]) / d[i];
            cand = fmin(v, cand); /* fold(INFINITY, |acc, x| x.min(acc)) */
        }
        ((qo_linesearch*)ls)->t_max = fmin(ls->t_max, cand);
    }
    return morethuente_step(ls, x, f0, g0, d, n, o, max_iter, st, work, work + n, work + 2 * n, work + 3 * n);
}

double qo_compute_step_len(const qo_linesearch* ls, const double* x, double f0, const double* g0,
                           const double* d, size_t n, qo_oracle_fn oracle, void* oracle_user, size_t max_iter) {
    oracle_t o = {oracle, oracle_user, 0};
    double* work = (double*)malloc(sizeof(double) * 4 * (n ? n : 1));
    double t = compute_step_len(ls, x, f0, g0, d, n, &o, max_iter, NULL, work);
    free(work);
    return t;
}

/* ------------------------------------------------------------------------------------------ */
/* solver state                                                                                 */
/* ------------------------------------------------------------------------------------------ */

struct qo_solver {
    int method, update_mode, nthreads;
    int h_symmetric; /* H == H' bit for bit (identity, kept so by the rank-2 form): its mat-vec may sweep rows = columns contiguously */
    double bytes_streamed; /* matrix bytes this solver's own sweeps moved (mat-vecs with H, the rank-2 update); the oracle counts its own */
    size_t n, k;
    double tol;
    double* x;
    double* h; /* approx_inv_hessian, column-major n x n (bfgs.rs:5) */
    int has_s_norm, has_y_norm;
    double s_norm, y_norm;
    /* scratch */
    double *g, *d, *xn, *gn, *s, *y, *u, *work;
    double *m0, *m1, *m2, *m3, *m4; /* n x n temporaries for the as-written update */
    double *lb, *ub; /* BFGSB / DFPB / SR1B bounds (NULL: unbounded solver) */
    /* Newton (newton/mod.rs:8-13) */
    qo_hessian_fn hess_fn;
    void* hess_user;
    int has_decrement;
    double decrement_squared;
};

/* BoxProjection::box_projection, number.rs:19: sup(lower).inf(upper) */
static double box1(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

void qo_solver_set_bounds(qo_solver* s, const double* lb, const double* ub) { /* bfgs_b.rs:43-63 */
    const size_t n = s->n;
    if (!s->lb) { s->lb = (double*)malloc(sizeof(double) * (n ? n : 1)); s->ub = (double*)malloc(sizeof(double) * (n ? n : 1)); }
    memcpy(s->lb, lb, sizeof(double) * n);
    memcpy(s->ub, ub, sizeof(double) * n);
    for (size_t i = 0; i < n; ++i) s->x[i] = box1(s->x[i], lb[i], ub[i]); /* :49 x0.box_projection(..) */
}

void qo_solver_set_hessian_fn(qo_solver* s, qo_hessian_fn fn, void* user) { s->hess_fn = fn; s->hess_user = user; }
int qo_solver_decrement_squared(const qo_solver* s, double* out) { if (s->has_decrement && out) *out = s->decrement_squared; return s->has_decrement; }

qo_solver* qo_solver_create(int method, double tol, const double* x0, size_t n, int update_mode, int nthreads) {
    qo_solver* s = (qo_solver*)calloc(1, sizeof(*s));
    s->method = method;
    s->update_mode = update_mode;
    s->nthreads = nthreads > 0 ? nthreads : 1;
    s->n = n;
    s->tol = tol;
    size_t nn = n ? n : 1;
    s->x = (double*)malloc(sizeof(double) * nn);
    memcpy(s->x, x0, sizeof(double) * n);
    double** vecs[] = {&s->g, &s->d, &s->xn, &s->gn, &s->s, &s->y, &s->u};
    for (size_t i = 0; i < sizeof(vecs) / sizeof(vecs[0]); ++i) *vecs[i] = (double*)calloc(nn, sizeof(double));
    s->work = (double*)calloc(4 * nn, sizeof(double));
    if (method == QO_BFGS || method == QO_DFP || method == QO_SR1) {
        /* bfgs.rs:27-39: H = I (the reference also keeps a second identity matrix; not needed here) */
        s->h = (double*)malloc(nn * nn * sizeof(double));
        /* first touch in parallel, in blocks of four columns as the sweeps partition them: on a multi-socket host the pages of a
         * thread's columns land on its own memory node (a single-threaded calloc + identity put all of H on one node) */
#pragma omp parallel for num_threads(s->nthreads) schedule(static) if (s->nthreads > 1 && n >= 256)
        for (size_t jb = 0; jb < (n + 3) / 4; ++jb)
            for (size_t j = 4 * jb; j < n && j < 4 * jb + 4; ++j) {
                double* hj = s->h + j * n;
                for (size_t i = 0; i < n; ++i) hj[i] = 0.0;
                hj[j] = 1.0;
            }
        s->h_symmetric = 1;
    }
    return s;
}

void qo_solver_destroy(qo_solver* s) {
    if (!s) return;
    free(s->x); free(s->h); free(s->g); free(s->d); free(s->xn); free(s->gn);
    free(s->s); free(s->y); free(s->u); free(s->work);
    free(s->m0); free(s->m1); free(s->m2); free(s->m3); free(s->m4);
    free(s->lb); free(s->ub);
    free(s);
}

size_t qo_solver_n(const qo_solver* s) { return s->n; }
size_t qo_solver_k(const qo_solver* s) { return s->k; }
const double* qo_solver_x(const qo_solver* s) { return s->x; }
const double* qo_solver_inv_hessian(const qo_solver* s) { return s->h; }
int qo_solver_s_norm(const qo_solver* s, double* out) { if (s->has_s_norm && out) *out = s->s_norm; return s->has_s_norm; }
int qo_solver_y_norm(const qo_solver* s, double* out) { if (s->has_y_norm && out) *out = s->y_norm; return s->has_y_norm; }
void qo_solver_set_inv_hessian(qo_solver* s, const double* h) {
    if (!s->h) return;
    memcpy(s->h, h, sizeof(double) * s->n * s->n);
    s->h_symmetric = 1;
    for (size_t i = 0; i < s->n && s->h_symmetric; ++i)
        for (size_t j = i + 1; j < s->n; ++j)
            if (h[i + j * s->n] != h[j + i * s->n]) { s->h_symmetric = 0; break; }
}
double qo_solver_bytes_streamed(const qo_solver* s) { return s->bytes_streamed; }

/* has_converged: bfgs.rs:64-76 (dfp.rs identical) ; gradient_descent.rs:46-53 */
static int has_converged(const qo_solver* s, const double* g) {
    if (s->method == QO_NEWTON) /* newton/mod.rs:64-69 */
        return s->has_decrement ? (s->decrement_squared * 0.5 < s->tol) : 0;
    if (s->method == QO_GRADIENT_DESCENT) {
        double acc = -INFINITY; /* fold(NEG_INFINITY, |acc, x| x.abs().max(acc)) */
        for (size_t i = 0; i < s->n; ++i) acc = fmax(fabs(g[i]), acc);
        return acc < s->tol;
    }
    if (s->has_s_norm && s->s_norm < s->tol) return 1; /* next_iterate_too_close, bfgs.rs:15-20 */
    if (s->has_y_norm && s->y_norm < s->tol) return 1; /* gradient_next_iterate_too_close, bfgs.rs:21-26 */
    return qo_norm(g, s->n) < s->tol;
}

static void ensure_mats(qo_solver* s) {
    if (s->m0) return;
    size_t nn = s->n * s->n;
    if (!nn) nn = 1;
    s->m0 = (double*)malloc(sizeof(double) * nn);
    s->m1 = (double*)malloc(sizeof(double) * nn);
    s->m2 = (double*)malloc(sizeof(double) * nn);
    s->m3 = (double*)malloc(sizeof(double) * nn);
    s->m4 = (double*)malloc(sizeof(double) * nn);
}

/* bfgs.rs:115-124 literally */
static void bfgs_update_as_written(qo_solver* so, const double* s, const double* y) {
    const size_t n = so->n;
    ensure_mats(so);
    double ys = qo_dot(y, s, n); /* :115 */
    double rho = 1.0 / ys;       /* :116 no sign check */
    double *left = so->m0, *right = so->m1, *tmp = so->m2, *res = so->m3;
    for (size_t j = 0; j < n; ++j)
        for (size_t i = 0; i < n; ++i) {
            double w_a = s[i] * y[j]; /* :117 w_a = s y' ; :119 w_b = w_a' */
            double w_b = s[j] * y[i];
            double id = (i == j) ? 1.0 : 0.0;
            left[i + j * n] = id - (w_a * rho);  /* :121 */
            right[i + j * n] = id - (w_b * rho); /* :122 */
        }
    matmul_colmajor(left, so->h, tmp, n, so->nthreads);  /* :124 left_term * H */
    matmul_colmajor(tmp, right, res, n, so->nthreads);   /* (...) * right_term */
    for (size_t j = 0; j < n; ++j)
        for (size_t i = 0; i < n; ++i) {
            double innov = s[i] * s[j]; /* :120 */
            so->h[i + j * n] = res[i + j * n] + innov * rho;
        }
}

/* dfp.rs:115-120 literally */
static void dfp_update_as_written(qo_solver* so, const double* s, const double* y) {
    const size_t n = so->n;
    ensure_mats(so);
    double *yy = so->m0, *t1 = so->m1, *t2 = so->m2;
    double sy = qo_dot(s, y, n);            /* :117 */
    qo_gemv_colsweep(so->h, y, so->u, n);   /* :118 H*y */
    double yhy = qo_dot(y, so->u, n);
    for (size_t j = 0; j < n; ++j)
        for (size_t i = 0; i < n; ++i) yy[i + j * n] = y[i] * y[j]; /* :116 */
    matmul_colmajor(so->h, yy, t1, n, so->nthreads); /* :120 H * yy */
    matmul_colmajor(t1, so->h, t2, n, so->nthreads); /* (...) * H */
    for (size_t j = 0; j < n; ++j)
        for (size_t i = 0; i < n; ++i) {
            double ss = s[i] * s[j]; /* :115 */
            double delta = ss / sy - t2[i + j * n] / yhy;
            so->h[i + j * n] += delta;
        }
}

/* sr1_b.rs:143-146 literally: hy = H y ; shy = s - hy ; H += shy shy' / shy.dot(y) */
static void sr1_update_as_written(qo_solver* so, const double* s, const double* y) {
    const size_t n = so->n;
    double* shy = so->u;
    gemv_colmajor_mt(so->h, y, shy, n, so->nthreads);
    for (size_t i = 0; i < n; ++i) shy[i] = s[i] - shy[i];
    const double den = qo_dot(shy, y, n);
    for (size_t j = 0; j < n; ++j)
        for (size_t i = 0; i < n; ++i) so->h[i + j * n] += (shy[i] * shy[j]) / den;
}

/* The algebraically identical symmetric rank-2 form:
 *   BFGS: H + c_su (s u' + u s') + c_ss s s',  u = H y, c_su = -rho, c_ss = rho^2 (y.u) + rho
 *   DFP : H + c_ss s s' + c_uu u u',           c_ss = 1/(s.y), c_uu = -1/(y.u)
 * evaluated element-wise as ((H + c_su*t1) + c_ss*(s_i s_j)) + c_uu*(u_i u_j), t1 = s_i u_j + u_i s_j,
 * i.e. with commutative inner sums, so H stays bitwise symmetric.  This is the formula the HIP path uses. */
static void solver_gemv_h(qo_solver* so, const double* x, double* y) {
    if (so->h_symmetric && so->update_mode == QO_UPDATE_RANK2) gemv_symmetric_rows_mt(so->h, x, y, so->n, so->nthreads);
    else gemv_colmajor_mt(so->h, x, y, so->n, so->nthreads);
    so->bytes_streamed += 8.0 * (double)so->n * (double)so->n;
}

static void rank2_update(qo_solver* so, const double* s, const double* y) {
    const size_t n = so->n;
    double* u = so->u;
    solver_gemv_h(so, y, u);
    so->bytes_streamed += 16.0 * (double)n * (double)n; /* the update below reads and writes every entry */
    double ys = qo_dot(y, s, n);
    double yu = qo_dot(y, u, n);
    double c_ss, c_su, c_uu;
    if (so->method == QO_BFGS) {
        double rho = 1.0 / ys;
        c_su = -rho;
        c_ss = rho * rho * yu + rho;
        c_uu = 0.0;
    } else if (so->method == QO_SR1) { /* (s-u)(s-u)'/((s-u).y) = c (s s' - (s u' + u s') + u u') */
        double den = 0.0;
        for (size_t i = 0; i < n; ++i) den += (s[i] - u[i]) * y[i];
        c_ss = 1.0 / den; c_su = -c_ss; c_uu = c_ss;
        (void)ys; (void)yu;
    } else {
        c_ss = 1.0 / ys;
        c_su = 0.0;
        c_uu = -1.0 / yu;
    }
    const int bfgs = so->method == QO_BFGS;
#pragma omp parallel for num_threads(so->nthreads) schedule(static) if (so->nthreads > 1 && n >= 256)
    for (size_t j = 0; j < n; ++j) {
        double* hj = so->h + j * n;
        const double sj = s[j], uj = u[j];
        if (bfgs) {
            for (size_t i = 0; i < n; ++i) {
                double t1 = s[i] * uj + u[i] * sj;
                double hn = hj[i] + c_su * t1;
                hj[i] = hn + c_ss * (s[i] * sj);
            }
        } else if (so->method == QO_SR1) {
            for (size_t i = 0; i < n; ++i) {
                double t1 = s[i] * uj + u[i] * sj;
                double hn = hj[i] + c_su * t1;
                hn = hn + c_ss * (s[i] * sj);
                hj[i] = hn + c_uu * (u[i] * uj);
            }
        } else {
            for (size_t i = 0; i < n; ++i) {
                double hn = hj[i] + c_ss * (s[i] * sj);
                hj[i] = hn + c_uu * (u[i] * uj);
            }
        }
    }
}

/* [nalgebra] Matrix::try_inverse on a dynamic square matrix: closed forms for n <= 2 (restated: 1x1 reciprocal, 2x2 by
 * the determinant), LU with partial pivoting otherwise (nalgebra also has closed forms for 3x3 / 4x4 and its LU order
 * is not reproduced: tolerance-level).  Returns 0 when singular (determinant / pivot exactly zero).  Column-major. */
static int try_inverse(const double* a, double* inv, size_t n) {
    if (n == 1) {
        if (a[0] == 0.0) return 0;
        inv[0] = 1.0 / a[0];
        return 1;
    }
    if (n == 2) {
        const double m11 = a[0], m21 = a[1], m12 = a[2], m22 = a[3];
        const double det = m11 * m22 - m21 * m12;
        if (det == 0.0) return 0;
        inv[0] = m22 / det; inv[2] = -m12 / det; inv[1] = -m21 / det; inv[3] = m11 / det;
        return 1;
    }
    double* lu = (double*)malloc(sizeof(double) * n * n);
    size_t* piv = (size_t*)malloc(sizeof(size_t) * n);
    memcpy(lu, a, sizeof(double) * n * n);
    int ok = 1;
    for (size_t k = 0; k < n && ok; ++k) {
        size_t p = k;
        double best = fabs(lu[k + k * n]);
        for (size_t i = k + 1; i < n; ++i) if (fabs(lu[i + k * n]) > best) { best = fabs(lu[i + k * n]); p = i; }
        piv[k] = p;
        if (best == 0.0) { ok = 0; break; }
        if (p != k) for (size_t j = 0; j < n; ++j) { double t = lu[k + j * n]; lu[k + j * n] = lu[p + j * n]; lu[p + j * n] = t; }
        const double d = lu[k + k * n];
        for (size_t i = k + 1; i < n; ++i) lu[i + k * n] /= d;
        for (size_t j = k + 1; j < n; ++j) {
            const double ukj = lu[k + j * n];
            for (size_t i = k + 1; i < n; ++i) lu[i + j * n] -= lu[i + k * n] * ukj;
        }
    }
    if (ok) {
        for (size_t c = 0; c < n; ++c) {
            double* x = inv + c * n;
            for (size_t i = 0; i < n; ++i) x[i] = (i == c) ? 1.0 : 0.0;
            for (size_t k = 0; k < n; ++k) if (piv[k] != k) { double t = x[k]; x[k] = x[piv[k]]; x[piv[k]] = t; }
            for (size_t k = 0; k < n; ++k) for (size_t i = k + 1; i < n; ++i) x[i] -= lu[i + k * n] * x[k];
            for (size_t kk = n; kk-- > 0;) { x[kk] /= lu[kk + kk * n]; for (size_t i = 0; i < kk; ++i) x[i] -= lu[i + kk * n] * x[kk]; }
        }
    }
    free(lu); free(piv);
    return ok;
}

/* Newton::compute_direction, newton/mod.rs:26-49 */
static void newton_direction(qo_solver* so) {
    const size_t n = so->n;
    ensure_mats(so);
    double *hess = so->m0, *inv = so->m1;
    so->hess_fn(so->hess_user, so->x, n, hess); /* eval.hessian().clone().expect(...) */
    if (try_inverse(hess, inv, n)) {
        qo_gemv_colsweep(inv, so->g, so->d, n); /* -&hessian_inv * eval.g() */
        for (size_t i = 0; i < n; ++i) so->d[i] = -so->d[i];
        qo_gemv_colsweep(inv, so->d, so->u, n); /* (hessian_inv * &direction).dot(&direction) -- reproduce, do not 'fix' */
        so->decrement_squared = qo_dot(so->u, so->d, n);
        so->has_decrement = 1;
    } else {
        for (size_t i = 0; i < n; ++i) so->d[i] = -so->g[i]; /* singular: gradient-descent direction */
    }
}

/* LineSearchSolver::minimize, ls_solver.rs:66-111, with BFGS::update_next_iterate (bfgs.rs:78-127),
 * DFP (dfp.rs:78-123) or the default hook (ls_solver.rs:44-64 / gradient_descent.rs:55-82). */
int qo_minimize(qo_solver* so, const qo_linesearch* ls, qo_oracle_fn oracle, void* oracle_user,
                size_t max_iter_solver, size_t max_iter_line_search,
                qo_callback_fn callback, void* callback_user, qo_trace* trace) {
    const size_t n = so->n;
    oracle_t o = {oracle, oracle_user, 0};
    so->k = 0; /* :74 -- H, s_norm, y_norm are NOT reset (warm restart) */
    if (trace) { trace->len = 0; trace->n_oracle_calls = 0; }
    int status = QO_MAX_ITER_REACHED;

    while (max_iter_solver > so->k) { /* :78 */
        size_t calls0 = o.calls;
        double f;
        call_oracle(&o, so->x, n, &f, so->g); /* :79 evaluate_x_k -> :36 */
        if (isnan(f) || isinf(f)) { status = QO_OUT_OF_DOMAIN; goto done; } /* :37-40 */
        if (has_converged(so, so->g)) { status = QO_OK; goto done; }       /* :81-88 */

        /* compute_direction */
        if (so->method == QO_GRADIENT_DESCENT) {
            for (size_t i = 0; i < n; ++i) so->d[i] = -so->g[i]; /* gradient_descent.rs:29 */
        } else if (so->method == QO_NEWTON) {
            newton_direction(so);
        } else {
            /* bfgs.rs:47: (-&H) * g.  Negating every H_ij first gives bit-for-bit -(H g). */
            solver_gemv_h(so, so->g, so->d);
            if (so->lb) { /* bfgs_b.rs:72-75: P(x - H g) - x */
                for (size_t i = 0; i < n; ++i) {
                    double t = so->x[i] - so->d[i];
                    t = box1(t, so->lb[i], so->ub[i]);
                    so->d[i] = t - so->x[i];
                }
            } else {
                for (size_t i = 0; i < n; ++i) so->d[i] = -so->d[i];
            }
        }

        ls_stats st = {0, 0, 0};
        double step = compute_step_len(ls, so->x, f, so->g, so->d, n, &o, max_iter_line_search, &st, so->work);
        qo_axpy_new(so->x, step, so->d, so->xn, n); /* bfgs.rs:94 / ls_solver.rs:60 */

        int updated = 0;
        if (so->method == QO_GRADIENT_DESCENT || so->method == QO_NEWTON) {
            memcpy(so->x, so->xn, sizeof(double) * n); /* gradient_descent.rs:79 ; Newton keeps the default hook ls_solver.rs:44-64 */
        } else {
            for (size_t i = 0; i < n; ++i) so->s[i] = so->xn[i] - so->x[i]; /* bfgs.rs:96 s = x+ - x (not t*d) */
            so->s_norm = qo_norm(so->s, n); so->has_s_norm = 1;             /* :97 */
            double fn;
            call_oracle(&o, so->xn, n, &fn, so->gn);                        /* :98 */
            for (size_t i = 0; i < n; ++i) so->y[i] = so->gn[i] - so->g[i];
            so->y_norm = qo_norm(so->y, n); so->has_y_norm = 1;             /* :99 */
            memcpy(so->x, so->xn, sizeof(double) * n);                      /* :102 */
            if (!(so->s_norm < so->tol) && !(so->y_norm < so->tol)) {       /* :106-112 */
                if (so->update_mode == QO_UPDATE_RANK2) rank2_update(so, so->s, so->y);
                else if (so->method == QO_SR1) sr1_update_as_written(so, so->s, so->y);
                else if (so->method == QO_BFGS) bfgs_update_as_written(so, so->s, so->y);
                else dfp_update_as_written(so, so->s, so->y);
                updated = 1;
            }
        }

        if (trace && trace->len < trace->cap) {
            qo_trace_rec* r = &trace->rec[trace->len];
            r->f = f;
            r->gnorm = qo_norm(so->g, n);
            r->t = step;
            r->s_norm = so->has_s_norm ? so->s_norm : NAN;
            r->y_norm = so->has_y_norm ? so->y_norm : NAN;
            r->n_evals = (int32_t)(o.calls - calls0);
            r->ls_iters = st.iters;
            r->ls_cases = st.cases;
            r->updated = updated;
            if (trace->xs) memcpy(trace->xs + trace->len * n, so->x, sizeof(double) * n);
            trace->len++;
        }

        so->k += 1; /* :104 */
        if (callback) callback(callback_user, so); /* :105-107 */
    }
done:
    if (trace) trace->n_oracle_calls = o.calls;
    return status;
}

/* ------------------------------------------------------------------------------------------ */
/* benchmark objectives (build-defined; the reference only has user closures)                   */
/* ------------------------------------------------------------------------------------------ */

int qo_quadratic_hessian(void* user, const double* x, size_t n, double* h) {
    (void)x;
    qo_quadratic* p = (qo_quadratic*)user;
    for (size_t j = 0; j < n; ++j)
        for (size_t i = 0; i < n; ++i) h[i + j * n] = p->q[i * n + j];
    return 0;
}

/* f = 1/2 x'(Qx) - b'x ; g = Qx - b.  Row sums run left to right over j. */
int qo_quadratic_eval(void* user, const double* x, size_t n, double* f, double* g) {
    qo_quadratic* p = (qo_quadratic*)user;
    const double* q = p->q;
    p->calls++;
    /* (four rows per thread in flight: one dependent-add chain per row is latency-bound; same sums, same order, per row) */
#pragma omp parallel for num_threads(p->nthreads) schedule(static) if (p->nthreads > 1 && n >= 256)
    for (size_t ib = 0; ib < (n + 3) / 4; ++ib) {
        const size_t i = 4 * ib;
        if (i + 4 <= n) {
            const double *r0 = q + i * n, *r1 = r0 + n, *r2 = r1 + n, *r3 = r2 + n;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (size_t j = 0; j < n; ++j) {
                const double xj = x[j];
                a0 += r0[j] * xj; a1 += r1[j] * xj; a2 += r2[j] * xj; a3 += r3[j] * xj;
            }
            g[i] = a0; g[i + 1] = a1; g[i + 2] = a2; g[i + 3] = a3; /* holds (Qx)_i for now */
        } else {
            for (size_t k = i; k < n; ++k) {
                const double* qi = q + k * n;
                double acc = 0.0;
                for (size_t j = 0; j < n; ++j) acc += qi[j] * x[j];
                g[k] = acc;
            }
        }
    }
    double xq = qo_dot(x, g, n);
    double bx = qo_dot(p->b, x, n);
    *f = 0.5 * xq - bx;
    for (size_t i = 0; i < n; ++i) g[i] = g[i] - p->b[i];
    return 0;
}

static uint64_t splitmix64_mix(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

/* counter-based uniform in [-1, 1): key = (min(i,j) << 32 | max(i,j)) */
double qo_synth_u(uint64_t seed, uint64_t i, uint64_t j) {
    uint64_t lo = i < j ? i : j, hi = i < j ? j : i;
    uint64_t key = (lo << 32) | hi;
    uint64_t z = splitmix64_mix(seed + key * 0x9E3779B97F4A7C15ull);
    double r = (double)(z >> 11) * 0x1.0p-53; /* [0,1) */
    return 2.0 * r - 1.0;
}

void qo_synth_fill_rows(double* q_rows, size_t n, size_t row0, size_t nrows, uint64_t seed,
                        const double* diag, int nthreads) {
    const double inv_n = 1.0 / (double)n;
    /* (blocks of four rows per thread: the first touch of the pages follows the evaluation's partition) */
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static) if (nthreads > 1)
    for (size_t rb = 0; rb < (nrows + 3) / 4; ++rb)
        for (size_t r = 4 * rb; r < nrows && r < 4 * rb + 4; ++r) {
            size_t i = row0 + r;
            double* qi = q_rows + r * n;
            for (size_t j = 0; j < n; ++j) qi[j] = (i == j) ? diag[i] : qo_synth_u(seed, i, j) * inv_n;
        }
}

/* f = log sum_i exp(a_i'x + c_i) + mu/2 x'x ; g = A' softmax(Ax + c) + mu x (max-shifted) */
int qo_logsumexp_eval(void* user, const double* x, size_t n, double* f, double* g) {
    qo_logsumexp* p = (qo_logsumexp*)user;
    const size_t m = p->m;
    p->calls++;
    double* z = (double*)malloc(sizeof(double) * (m ? m : 1));
#pragma omp parallel for num_threads(p->nthreads) schedule(static) if (p->nthreads > 1 && m >= 256)
    for (size_t i = 0; i < m; ++i) {
        const double* ai = p->a + i * n;
        double acc = 0.0;
        for (size_t j = 0; j < n; ++j) acc += ai[j] * x[j];
        z[i] = acc + p->c[i];
    }
    double zmax = -INFINITY;
    for (size_t i = 0; i < m; ++i) zmax = fmax(zmax, z[i]);
    double sum = 0.0;
    for (size_t i = 0; i < m; ++i) { z[i] = exp(z[i] - zmax); sum += z[i]; }
    *f = zmax + log(sum) + 0.5 * p->mu * qo_dot(x, x, n);
    for (size_t i = 0; i < m; ++i) z[i] = z[i] / sum;
#pragma omp parallel for num_threads(p->nthreads) schedule(static) if (p->nthreads > 1 && n >= 256)
    for (size_t j = 0; j < n; ++j) {
        double acc = 0.0;
        for (size_t i = 0; i < m; ++i) acc += z[i] * p->a[i * n + j];
        g[j] = acc + p->mu * x[j];
    }
    free(z);
    return 0;
}
