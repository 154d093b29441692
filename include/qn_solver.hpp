// qn_solver.hpp -- header-only C++17 host mirror of the reference's trait surface over the C ABI (qn_hip.h).
//
// The reference is a compiled (Rust) crate; this is the compiled-language host side a user would write
// against: same names, argument meaning and error behaviour as the crate, so code reads like the reference's
// examples and tests.  Nothing numeric happens here -- every call forwards to libqn_hip.so (GPU only).
//
//   reference (Rust)                                   here (C++)
//   BFGS::new(tol, x0)                                 BFGS::new_(tol, x0)            bfgs.rs:27-39
//   DFP::new / GradientDescent::new                    DFP::new_ / GradientDescent::new_
//   MoreThuente::default().with_c1(..)                 MoreThuente::default_().with_c1(..)   morethuente.rs:16-62
//   BackTracking::new(c1, beta)                        BackTracking::new_(c1, beta)   backtracking.rs:8-10
//   FuncEvalMultivariate::new(f, g)                    FuncEvalMultivariate(f, g)     func_eval.rs:4-41
//   solver.minimize(&mut ls, oracle, a, b, callback)   solver.minimize(ls, oracle, a, b, callback)  ls_solver.rs:66-111
//   Result<(), SolverError> / .unwrap()                Result / .unwrap() (throws SolverError)      ls_solver.rs:10-20
#pragma once
#include <cmath>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "qn_hip.h"

namespace optimization_solvers {

using Floating = double;                 // number.rs:3
using DVector = std::vector<Floating>;   // nalgebra::DVector<f64>

// ls_solver.rs:10-20
struct SolverError : std::runtime_error {
    enum Kind { MaxIterReached = QN_MAX_ITER_REACHED, OutOfDomain = QN_OUT_OF_DOMAIN, ErrorInputParams = QN_ERROR_INPUT_PARAMS,
                AbnormalTermination = QN_ABNORMAL_TERMINATION };
    Kind kind;
    explicit SolverError(int code, const std::string& detail = "")
        : std::runtime_error(std::string(qn_status_string(code)) + (detail.empty() ? "" : ": " + detail)), kind(static_cast<Kind>(code)) {}
};

// Result<(), SolverError>
class Result {
    std::optional<SolverError> err_;
  public:
    Result() = default;
    explicit Result(SolverError e) : err_(std::move(e)) {}
    bool is_ok() const { return !err_; }
    bool is_err() const { return bool(err_); }
    const SolverError& unwrap_err() const { return *err_; }
    void unwrap() const { if (err_) throw *err_; }
};

inline Result make_result(int status) {
    if (status == QN_OK) return Result();
    const bool detail = status == QN_ERROR_INPUT_PARAMS || status == QN_ABNORMAL_TERMINATION;
    return Result(SolverError(status, detail ? qn_last_error_message() : ""));
}
inline void check(int status) { make_result(status).unwrap(); }

// func_eval.rs:4-41
class FuncEvalMultivariate {
    Floating f_;
    DVector g_;
  public:
    FuncEvalMultivariate(Floating f, DVector g) : f_(f), g_(std::move(g)) {}
    const Floating& f() const { return f_; }
    const DVector& g() const { return g_; }
};

// one GPU (optionally one rank of a row-sharded group)
class Context {
    qn_context* h_ = nullptr;
  public:
    explicit Context(int device = 0) { check(qn_context_create(device, &h_)); }
    Context(int device, int rank, int world, const void* unique_id) { check(qn_context_create_sharded(device, rank, world, unique_id, &h_)); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    ~Context() { qn_context_destroy(h_); }
    qn_context* handle() const { return h_; }
    static Context& default_context() { static Context c(0); return c; }
};

// LineSearch::compute_step_len (line_search/mod.rs:14-23) for a line-search mirror `LS` and a host closure
template <class LS, class Oracle>
inline Floating compute_step_len(LS& ls, const DVector& x_k, const FuncEvalMultivariate& eval_x_k, const DVector& direction_k, Oracle&& oracle,
                                 size_t max_iter, Context& ctx = Context::default_context()) {
    using OracleT = std::remove_reference_t<Oracle>;
    OracleT* op = &oracle;
    auto tramp = [](void* user, const double* x, size_t n, double* f, double* g) -> int {
        DVector xv(x, x + n);
        FuncEvalMultivariate ev = (*static_cast<OracleT*>(user))(xv);
        *f = ev.f();
        for (size_t i = 0; i < n; ++i) g[i] = ev.g()[i];
        return 0;
    };
    qn_oracle o{};
    o.kind = QN_ORACLE_HOST;
    o.memoize = 0;
    o.host_fn = tramp;
    o.host_user = op;
    Floating t = 0;
    check(qn_compute_step_len(ctx.handle(), &ls.ffi(), x_k.data(), eval_x_k.f(), eval_x_k.g().data(), direction_k.data(), x_k.size(), &o, max_iter, &t));
    return t;
}

// morethuente.rs:6-62
class MoreThuente {
    qn_linesearch s_;
  public:
    template <class Oracle>
    Floating compute_step_len(const DVector& x_k, const FuncEvalMultivariate& eval_x_k, const DVector& direction_k, Oracle&& oracle, size_t max_iter) {
        return optimization_solvers::compute_step_len(*this, x_k, eval_x_k, direction_k, oracle, max_iter);
    }
    MoreThuente() { qn_morethuente_default(&s_); }
    static MoreThuente default_() { return MoreThuente(); }
    MoreThuente with_deltas(Floating dmin, Floating d, Floating dmax) && { check(qn_morethuente_with_deltas(&s_, dmin, d, dmax)); return *this; }
    MoreThuente with_t_min(Floating v) && { check(qn_morethuente_with_t_min(&s_, v)); return *this; }
    MoreThuente with_t_max(Floating v) && { check(qn_morethuente_with_t_max(&s_, v)); return *this; }
    MoreThuente with_c1(Floating v) && { check(qn_morethuente_with_c1(&s_, v)); return *this; } // assert!s -> ErrorInputParams
    MoreThuente with_c2(Floating v) && { check(qn_morethuente_with_c2(&s_, v)); return *this; }
    Floating c1() const { return s_.c1; }
    Floating c2() const { return s_.c2; }
    Floating t_max() const { return s_.t_max; }
    qn_linesearch& ffi() { return s_; }
};

// backtracking.rs:3-11
class BackTracking {
    qn_linesearch s_;
  public:
    template <class Oracle>
    Floating compute_step_len(const DVector& x_k, const FuncEvalMultivariate& eval_x_k, const DVector& direction_k, Oracle&& oracle, size_t max_iter) {
        return optimization_solvers::compute_step_len(*this, x_k, eval_x_k, direction_k, oracle, max_iter);
    }
    BackTracking(Floating c1, Floating beta) { qn_backtracking_new(&s_, c1, beta); }
    static BackTracking new_(Floating c1, Floating beta) { return BackTracking(c1, beta); }
    qn_linesearch& ffi() { return s_; }
};

// a device-resident objective (built-in quadratic f = 1/2 x'Qx - b'x)
class Quadratic {
    qn_objective* h_ = nullptr;
    size_t n_;
  public:
    Quadratic(const DVector& q_rowmajor, const DVector& b, Context& ctx = Context::default_context()) : n_(b.size()) {
        check(qn_quadratic_create(ctx.handle(), n_, q_rowmajor.data(), b.data(), &h_));
    }
    Quadratic(const Quadratic&) = delete;
    ~Quadratic() { qn_objective_destroy(h_); }
    qn_objective* handle() const { return h_; }
    FuncEvalMultivariate operator()(const DVector& x) const {
        DVector g(n_);
        Floating f = 0;
        check(qn_objective_eval(h_, x.data(), &f, g.data()));
        return FuncEvalMultivariate(f, std::move(g));
    }
};

template <int METHOD>
class LineSearchSolver { // ls_solver.rs:23-112 for the three solvers on the path
    qn_solver* h_ = nullptr;
    size_t n_;
    mutable DVector x_cache_;
    mutable size_t k_cache_ = 0;
    mutable std::optional<Floating> opt_cache_[2];
  public:
    using Self = LineSearchSolver<METHOD>;
    LineSearchSolver(Floating tol, const DVector& x0, Context& ctx = Context::default_context()) : n_(x0.size()) {
        check(qn_solver_create(ctx.handle(), METHOD, tol, x0.data(), x0.size(), &h_));
    }
    static Self new_(Floating tol, const DVector& x0) { return Self(tol, x0); }
    LineSearchSolver(const Self&) = delete;
    LineSearchSolver(Self&& o) noexcept : h_(o.h_), n_(o.n_) { o.h_ = nullptr; }
    ~LineSearchSolver() { if (h_) qn_solver_destroy(h_); }

    // getters generated by derive_getters (bfgs.rs:3-12) and LineSearchSolver::xk/k (bfgs.rs:52-63)
    const DVector& x() const { x_cache_.resize(n_); check(qn_solver_get_x(h_, x_cache_.data())); return x_cache_; }
    const DVector& xk() const { return x(); }
    const size_t& k() const { k_cache_ = qn_solver_k(h_); return k_cache_; }
    Floating tol() const { return qn_solver_tol(h_); }
    std::optional<Floating> s_norm() const { Floating v; int some; check(qn_solver_s_norm(h_, &v, &some)); return some ? std::optional<Floating>(v) : std::nullopt; }
    std::optional<Floating> y_norm() const { Floating v; int some; check(qn_solver_y_norm(h_, &v, &some)); return some ? std::optional<Floating>(v) : std::nullopt; }
    bool next_iterate_too_close() const { int v; check(qn_solver_next_iterate_too_close(h_, &v)); return v != 0; }                   // bfgs.rs:15-20
    bool gradient_next_iterate_too_close() const { int v; check(qn_solver_gradient_next_iterate_too_close(h_, &v)); return v != 0; } // bfgs.rs:21-26
    DVector approx_inv_hessian() const { DVector m(n_ * n_); check(qn_solver_get_inv_hessian(h_, m.data(), 1)); return m; }          // column-major, like DMatrix
    bool has_converged(const FuncEvalMultivariate& eval) const { // bfgs.rs:64-76 / gradient_descent.rs:46-53
        if (METHOD == QN_GRADIENT_DESCENT) {
            Floating acc = -INFINITY;
            for (Floating v : eval.g()) acc = std::fmax(std::fabs(v), acc);
            return acc < tol();
        }
        if (next_iterate_too_close() || gradient_next_iterate_too_close()) return true;
        Floating s = 0;
        for (Floating v : eval.g()) s += v * v;
        return std::sqrt(s) < tol();
    }

    // minimize with a host closure: the reference's exact oracle-call sequence (ls_solver.rs:66-111)
    template <class LS, class Oracle>
    Result minimize(LS& line_search, Oracle&& oracle, size_t max_iter_solver, size_t max_iter_line_search,
                    std::optional<std::function<void(const Self&)>> callback = std::nullopt) {
        using OracleT = std::remove_reference_t<Oracle>;
        struct Ctx { OracleT* o; size_t n; } octx{&oracle, n_};
        auto tramp = [](void* user, const double* x, size_t n, double* f, double* g) -> int {
            Ctx* c = static_cast<Ctx*>(user);
            DVector xv(x, x + n);
            FuncEvalMultivariate ev = (*c->o)(xv);
            *f = ev.f();
            for (size_t i = 0; i < n; ++i) g[i] = ev.g()[i];
            return 0;
        };
        struct Cb { Self* me; std::function<void(const Self&)>* f; } cb{this, callback ? &*callback : nullptr};
        auto cb_tramp = [](void* user, qn_solver*) { Cb* c = static_cast<Cb*>(user); (*c->f)(*c->me); };
        qn_oracle o{};
        o.kind = QN_ORACLE_HOST;
        o.memoize = 0;
        o.host_fn = tramp;
        o.host_user = &octx;
        const int st = qn_minimize(h_, &line_search.ffi(), &o, max_iter_solver, max_iter_line_search,
                                   callback ? static_cast<qn_callback_fn>(cb_tramp) : nullptr, &cb);
        return make_result(st);
    }

    // minimize with a device-resident objective: no host round trip per oracle call, distinct points evaluated once
    template <class LS>
    Result minimize(LS& line_search, const Quadratic& objective, size_t max_iter_solver, size_t max_iter_line_search) {
        qn_oracle o{};
        o.kind = QN_ORACLE_OBJECTIVE;
        o.memoize = 1;
        o.objective = objective.handle();
        return make_result(qn_minimize(h_, &line_search.ffi(), &o, max_iter_solver, max_iter_line_search, nullptr, nullptr));
    }
};

using BFGS = LineSearchSolver<QN_BFGS>;                       // quasi_newton/bfgs.rs
using DFP = LineSearchSolver<QN_DFP>;                         // quasi_newton/dfp.rs
using GradientDescent = LineSearchSolver<QN_GRADIENT_DESCENT>; // steepest_descent/gradient_descent.rs

} // namespace optimization_solvers
