// examples/quadratic.cpp -- the reference's examples/quadratic.rs through the C++ host mirror (qn_solver.hpp).
// BFGS + MoreThuente::default on f = x'Ix, x0 = (1, 1); ends with the reference's own assertion f == 0.0.
#include <cstdio>
#include <cstdlib>

#include "qn_solver.hpp"

using namespace optimization_solvers;

int main() {
    // Setting up the oracle (examples/quadratic.rs:10-16)
    const DVector matrix = {1., 0., 0., 1.};
    auto f_and_g = [&](const DVector& x) -> FuncEvalMultivariate {
        const DVector mx = {matrix[0] * x[0] + matrix[2] * x[1], matrix[1] * x[0] + matrix[3] * x[1]};
        const Floating f = x[0] * mx[0] + x[1] * mx[1];
        return FuncEvalMultivariate(f, {2. * mx[0], 2. * mx[1]});
    };
    // Setting up the line search and the solver (:18-22)
    auto ls = MoreThuente::default_();
    const Floating tol = 1e-6;
    const DVector x0 = {1., 1.};
    auto solver = BFGS::new_(tol, x0);
    // Running the solver (:25-36)
    const size_t max_iter_solver = 100, max_iter_line_search = 10;
    solver.minimize(ls, f_and_g, max_iter_solver, max_iter_line_search, std::nullopt).unwrap();
    // Printing the result (:38-43)
    const DVector& x = solver.x();
    const auto eval = f_and_g(x);
    std::printf("x: [%g, %g]\n", x[0], x[1]);
    std::printf("f(x): %g\n", eval.f());
    std::printf("g(x): [%g, %g]\n", eval.g()[0], eval.g()[1]);
    std::printf("k: %zu\n", solver.k());
    if (eval.f() != 0.0) { std::printf("assert_eq!(eval.f(), &0.0) FAILED\n"); return 1; } // examples/quadratic.rs:43

    // bfgs.rs:190-239 bfgs_backtracking, and the MaxIterReached / OutOfDomain error paths (ls_solver.rs:10-20)
    const Floating gamma = 1.;
    auto fg2 = [&](const DVector& v) {
        return FuncEvalMultivariate(0.5 * ((v[0] + 1.) * (v[0] + 1.) + gamma * (v[1] - 1.) * (v[1] - 1.)), {v[0] + 1., gamma * (v[1] - 1.)});
    };
    auto bt = BackTracking::new_(1e-4, 0.5);
    auto gd = BFGS::new_(1e-12, {180.0, 152.0});
    gd.minimize(bt, fg2, 1000, 100000, std::nullopt).unwrap();
    if (!(std::fabs(fg2(gd.xk()).f()) < 1e-6) || !gd.has_converged(fg2(gd.xk()))) { std::printf("bfgs_backtracking FAILED\n"); return 1; }
    auto gd2 = BFGS::new_(1e-12, {180.0, 152.0});
    size_t seen = 0;
    Result r = gd2.minimize(bt, fg2, 1, 10, std::function<void(const BFGS&)>([&](const BFGS& s) { seen = s.k(); }));
    if (!r.is_err() || r.unwrap_err().kind != SolverError::MaxIterReached || seen != 1) { std::printf("MaxIterReached path FAILED\n"); return 1; }
    std::printf("error path: %s\n", r.unwrap_err().what());
    bool threw = false;
    try { auto bad = MoreThuente::default_().with_c1(2.0); (void)bad; } catch (const SolverError& e) { threw = e.kind == SolverError::ErrorInputParams; }
    if (!threw) { std::printf("builder assert FAILED\n"); return 1; }
    // backtracking.rs:65-113: LineSearch::compute_step_len on its own inside a hand-rolled gradient descent (gamma = 90)
    {
        const Floating g90 = 90.0;
        auto fg3 = [&](const DVector& v) { return FuncEvalMultivariate(0.5 * (v[0] * v[0] + g90 * v[1] * v[1]), {v[0], g90 * v[1]}); };
        DVector it{180.0, 152.0};
        auto ls = BackTracking::new_(1e-4, 0.5);
        const size_t max_iter = 1000;
        size_t k3 = 1;
        while (max_iter > k3) {
            FuncEvalMultivariate ev = fg3(it);
            if (ev.g()[0] * ev.g()[0] + ev.g()[1] * ev.g()[1] < 1e-12) break;
            DVector dir{-ev.g()[0], -ev.g()[1]};
            const Floating t = ls.compute_step_len(it, ev, dir, fg3, max_iter);
            it[0] += t * dir[0];
            it[1] += t * dir[1];
            ++k3;
        }
        if (!(std::fabs(it[0]) < 1e-6)) { std::printf("backtracking compute_step_len FAILED\n"); return 1; } // backtracking.rs:111
        std::printf("line search alone: %zu steps\n", k3);
    }
    std::printf("ok\n");
    return 0;
}
