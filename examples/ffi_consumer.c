/* ffi_consumer.c -- the C ABI driven from plain C99, call for call the way the Rust shim of INTEGRATION.md drives it:
 * a boxed closure behind a `void* user` trampoline (ls_solver.rs:69 `impl FnMut(&DVector<f64>) -> FuncEvalMultivariate`),
 * the per-iteration callback (ls_solver.rs:72,105-107), Result<(), SolverError> as the return code, getters afterwards.
 * Problem: the reference's bfgs.rs:141-188 unit test (f = 1/2((x0+1)^2 + (x1-1)^2), x0 = (180,152), tol 1e-12,
 * MoreThuente::default(), caps 1000 / 100000) and bfgs_example.rs:11-52 (x^2+2y^2+3z^2+xy+yz from (1,1,1)).
 * Built by __graft_entry__.build(); run by tests/test_gpu_ffi_consumer.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qn_hip.h"

typedef struct { /* what the Rust side boxes: the closure's captured state */
    int which;
    size_t calls;
} closure_env;

static int oracle_trampoline(void* user, const double* x, size_t n, double* f, double* g) {
    closure_env* env = (closure_env*)user;
    env->calls++;
    if (env->which == 0) {
        if (n != 2) return 1;
        *f = 0.5 * (pow(x[0] + 1.0, 2.0) + pow(x[1] - 1.0, 2.0));
        g[0] = x[0] + 1.0;
        g[1] = x[1] - 1.0;
    } else {
        if (n != 3) return 1;
        *f = x[0] * x[0] + 2.0 * x[1] * x[1] + 3.0 * x[2] * x[2] + x[0] * x[1] + x[1] * x[2];
        g[0] = 2.0 * x[0] + x[1];
        g[1] = 4.0 * x[1] + x[0] + x[2];
        g[2] = 6.0 * x[2] + x[1];
    }
    return 0;
}

typedef struct { size_t seen; size_t last_k; } callback_env;
static void callback_trampoline(void* user, qn_solver* solver) {
    callback_env* env = (callback_env*)user;
    env->seen++;
    env->last_k = qn_solver_k(solver);
}

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int st_ = (call);                                                                            \
        if (st_ != QN_OK) {                                                                          \
            fprintf(stderr, "%s -> %s (%s)\n", #call, qn_status_string(st_), qn_last_error_message()); \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

static int run(qn_context* ctx, int which, const double* x0, size_t n, double tol, size_t max_iter, size_t max_ls, double* f_out,
               size_t* k_out, size_t* calls_out) {
    qn_solver* s = NULL;
    CHECK(qn_solver_create(ctx, QN_BFGS, tol, x0, n, &s));
    qn_linesearch ls;
    qn_morethuente_default(&ls);
    closure_env env = {which, 0};
    callback_env cb = {0, 0};
    qn_oracle o;
    memset(&o, 0, sizeof(o));
    o.kind = QN_ORACLE_HOST;
    o.memoize = 0; /* the reference's exact call sequence */
    o.host_fn = oracle_trampoline;
    o.host_user = &env;
    int st = qn_minimize(s, &ls, &o, max_iter, max_ls, callback_trampoline, &cb);
    if (st != QN_OK) { fprintf(stderr, "minimize: %s\n", qn_status_string(st)); return 1; }
    double x[3], g[3], f;
    CHECK(qn_solver_get_x(s, x));
    oracle_trampoline(&env, x, n, &f, g);
    *f_out = f;
    *k_out = qn_solver_k(s);
    *calls_out = env.calls - 1;
    if (cb.seen != *k_out || cb.last_k != *k_out) { fprintf(stderr, "callback count %zu vs k %zu\n", cb.seen, *k_out); return 1; }
    double sn = 0.0;
    int some = 0;
    CHECK(qn_solver_s_norm(s, &sn, &some));
    if (!some) { fprintf(stderr, "s_norm is None after an update\n"); return 1; }
    qn_solver_destroy(s);
    return 0;
}

int main(void) {
    qn_context* ctx = NULL;
    CHECK(qn_context_create(0, &ctx));
    double f;
    size_t k, calls;
    const double x0a[2] = {180.0, 152.0};
    if (run(ctx, 0, x0a, 2, 1e-12, 1000, 100000, &f, &k, &calls)) return 1;
    printf("bfgs_morethuente: f = %.3e, k = %zu, oracle calls = %zu\n", f, k, calls);
    if (!(fabs(f) < 1e-6)) return 2; /* bfgs.rs:186 assert!((eval.f() - 0.0).abs() < 1e-6) */
    const double x0b[3] = {1.0, 1.0, 1.0};
    if (run(ctx, 1, x0b, 3, 1e-8, 50, 20, &f, &k, &calls)) return 1;
    printf("bfgs_example: f = %.3e, k = %zu, oracle calls = %zu\n", f, k, calls);
    if (!(fabs(f) < 1e-12)) return 2;
    /* error paths the shim maps onto SolverError */
    qn_solver* s = NULL;
    if (qn_solver_create(ctx, 99, 1e-6, x0a, 2, &s) != QN_ERROR_INPUT_PARAMS) return 3;
    printf("error path: %s\n", qn_status_string(QN_ERROR_INPUT_PARAMS));
    qn_context_destroy(ctx);
    printf("ok\n");
    return 0;
}
