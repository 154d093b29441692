"""The reference's convenience facade over the three solvers it exports to JavaScript (src/wasm.rs:7-290): same names,
same defaults, same result fields -- row f3's "wasm-parity result struct" (SURVEY.md 8(f)).  Only the host-language
glue differs: the objective is a Python callable returning the flat sequence the JS function returns,
`[f, g_0 .. g_{n-1}]` (wasm.rs:95-105) or, for Newton, `[f, g_0 .. g_{n-1}, H column-major]` (wasm.rs:215-243)."""
import numpy as np

from .solver import BFGS, BackTracking, FuncEvalMultivariate, GradientDescent, MoreThuente, Newton, SolverError


class OptimizationResult:
    """wasm.rs:7-56"""

    def __init__(self):  # OptimizationResult::new, wasm.rs:17-26
        self.x = []
        self.f_value = 0.0
        self.gradient_norm = 0.0
        self.iterations = 0
        self.success = False
        self.error_message = ""

    def get_x(self):
        return list(self.x)

    def get_f_value(self):
        return self.f_value

    def get_gradient_norm(self):
        return self.gradient_norm

    def get_iterations(self):
        return self.iterations

    def get_success(self):
        return self.success

    def get_error_message(self):
        return self.error_message


class OptimizationSolver:
    """wasm.rs:58-290: `new(tolerance, max_iterations)`; every solve uses max_iter_line_search = 20 and no callback."""

    def __init__(self, tolerance, max_iterations, ctx=None):
        self.tolerance, self.max_iterations, self.ctx = float(tolerance), int(max_iterations), ctx

    @staticmethod
    def _objective(fn, n, with_hessian):
        def objective(x):
            flat = np.asarray(fn(np.asarray(x, dtype=np.float64)), dtype=np.float64).ravel()
            f, g = flat[0], flat[1:1 + n]
            if not with_hessian:
                return FuncEvalMultivariate(f, g)
            if flat.size < 1 + n + n * n:  # wasm.rs:238 panic!("Expected Hessian component at index {}")
                raise RuntimeError(f"Expected Hessian component at index {flat.size}")
            return FuncEvalMultivariate(f, g).with_hessian(flat[1 + n:1 + n + n * n].reshape((n, n), order="F"))
        return objective

    def _run(self, solver, ls, objective):
        result = OptimizationResult()
        try:
            solver.minimize(ls, objective, self.max_iterations, 20)
        except SolverError as e:  # wasm.rs:130-133: format!("Optimization failed: {:?}", e) -- the Debug name of the variant
            result.error_message = f"Optimization failed: {type(e).__name__}"
            result.success = False
            return result
        x = solver.x()
        ev = objective(x)
        result.x = [float(v) for v in x]
        result.f_value = float(ev.f())
        result.gradient_norm = float(np.sqrt(np.dot(ev.g(), ev.g())))
        result.iterations = int(solver.k())
        result.success = True
        return result

    def solve_gradient_descent(self, x0, f_and_g_fn):  # wasm.rs:73-137: GradientDescent + BackTracking::new(1e-4, 0.5)
        x0 = np.asarray(x0, dtype=np.float64)
        return self._run(GradientDescent(self.tolerance, x0, ctx=self.ctx), BackTracking(1e-4, 0.5), self._objective(f_and_g_fn, x0.size, False))

    def solve_bfgs(self, x0, f_and_g_fn):  # wasm.rs:139-190: BFGS + MoreThuente::default()
        x0 = np.asarray(x0, dtype=np.float64)
        return self._run(BFGS(self.tolerance, x0, ctx=self.ctx), MoreThuente(), self._objective(f_and_g_fn, x0.size, False))

    def solve_newton(self, x0, f_and_g_and_h_fn):  # wasm.rs:192-272: Newton + MoreThuente::default()
        x0 = np.asarray(x0, dtype=np.float64)
        return self._run(Newton(self.tolerance, x0, ctx=self.ctx), MoreThuente(), self._objective(f_and_g_and_h_fn, x0.size, True))
