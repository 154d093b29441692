"""GPU test of the hook-level entry points a trait-by-trait binding uses (rust/src/lib.rs: `impl ComputeDirection`, `impl
LineSearchSolver` with the reference's default `minimize` template, ls_solver.rs:66-111): qn_solver_compute_direction,
qn_compute_step_len and qn_solver_secant_update driven from the host in the template's order must reproduce the device-resident
qn_minimize run and the CPU oracle -- same iterates, same step lengths, same s_norm / y_norm, same inverse Hessian."""
import numpy as np
import pytest

import problems as P
from test_gpu_parity import T_TOL, X_TOL, _ls

pytestmark = pytest.mark.gpu


def _template_minimize(qn, solver, ls, oracle, max_iter_solver, max_iter_ls, inf_norm=False):
    """ls_solver.rs:66-111 with bfgs.rs:52-130's hooks, each hook one ABI call"""
    x = solver.x()
    k = 0
    steps = []
    while max_iter_solver > k:
        f, g = oracle(x)                                            # evaluate_x_k
        if not np.isfinite(f):
            return "out_of_domain", k, x, steps
        gnorm = np.max(np.abs(g)) if inf_norm else np.linalg.norm(g)  # gradient_descent.rs:46-53 uses the infinity norm
        if solver.next_iterate_too_close() or solver.gradient_next_iterate_too_close() or gnorm < solver.tol():
            return "ok", k, x, steps                                # has_converged
        d = solver.compute_direction((f, g))                        # ComputeDirection
        t = ls.compute_step_len(x, (f, g), d, oracle, max_iter_ls)  # LineSearch (GPU line search, host closure)
        x_next = x + t * d
        s = x_next - x
        y = oracle(x_next)[1] - g                                   # bfgs.rs:98
        solver.set_x(x_next)                                        # *self.xk_mut() = next_iterate
        solver.secant_update(s, y)                                  # s_norm, y_norm, too-close exits, H update
        x = x_next
        k += 1
        steps.append(t)
    return "max_iter", k, x, steps


@pytest.mark.parametrize("method", ["bfgs", "dfp", "gd"])
@pytest.mark.parametrize("lsname", ["mt", "bt"])
@pytest.mark.parametrize("n", [7, 200])
def test_hook_by_hook_run_equals_device_resident_run(qn, qo, method, lsname, n):
    q, b, x0, _ = P.synth_problem(qo, n)
    iters = 12

    def oracle(x):
        return 0.5 * x @ (q @ x) - b @ x, q @ x - b
    mk = {"bfgs": qn.BFGS, "dfp": qn.DFP, "gd": qn.GradientDescent}[method]
    a = mk(1e-10, x0)
    if method == "gd":  # no secant update: the hook is a no-op for steepest descent
        a.secant_update = lambda s, y: None
        a.next_iterate_too_close = lambda: False
        a.gradient_next_iterate_too_close = lambda: False
    status, k, x, steps = _template_minimize(qn, a, _ls(qn, lsname), oracle, iters, 20, inf_norm=method == "gd")

    r = mk(1e-10, x0)
    r.set_trace(iters, with_x=True)
    st = "ok"
    try:
        r.minimize(_ls(qn, lsname), oracle, iters, 20)
    except qn.MaxIterReached:
        st = "max_iter"
    tr, xs = r.trace()
    assert (status, k) == (st, r.k())
    assert len(steps) == len(tr)
    for t, rec in zip(steps, tr):
        assert abs(t - rec["t"]) <= T_TOL * abs(rec["t"])
    assert np.linalg.norm(x - r.x()) <= X_TOL * max(1.0, np.linalg.norm(r.x()))
    if method != "gd":
        assert abs(a.s_norm() - r.s_norm()) <= 1e-9 * r.s_norm() and abs(a.y_norm() - r.y_norm()) <= 1e-9 * r.y_norm()
        ha, hr = a.approx_inv_hessian(), r.approx_inv_hessian()
        assert np.array_equal(ha, ha.T)
        assert np.linalg.norm(ha - hr) <= 1e-8 * np.linalg.norm(hr)


def test_compute_direction_and_secant_update_vs_oracle_formulas(qn, qo):
    n = 130
    rng = np.random.default_rng(2)
    m = rng.standard_normal((n, n))
    h0 = m @ m.T / n + np.eye(n)
    g = rng.standard_normal(n)
    s_ = rng.standard_normal(n)
    y = s_ + 0.1 * rng.standard_normal(n)  # y's > 0
    for mk, method in ((qn.BFGS, "bfgs"), (qn.DFP, "dfp")):
        sol = mk(1e-10, np.zeros(n))
        sol.set_approx_inv_hessian(h0)
        d = sol.compute_direction((0.0, g))
        assert np.linalg.norm(d + h0 @ g) <= 1e-12 * np.linalg.norm(h0 @ g)
        sol.secant_update(s_, y)
        ys = y @ s_
        if method == "bfgs":  # bfgs.rs:112-127 as written
            rho = 1.0 / ys
            want = (np.eye(n) - rho * np.outer(s_, y)) @ h0 @ (np.eye(n) - rho * np.outer(y, s_)) + rho * np.outer(s_, s_)
        else:  # dfp.rs:110-116
            want = h0 + np.outer(s_, s_) / ys - (h0 @ np.outer(y, y) @ h0) / (y @ h0 @ y)
        got = sol.approx_inv_hessian()
        assert np.linalg.norm(got - want) <= 1e-11 * np.linalg.norm(want)
        assert np.linalg.norm(got @ y - s_) <= 1e-10 * np.linalg.norm(s_)  # secant equation
        assert sol.s_norm() == pytest.approx(np.linalg.norm(s_), rel=1e-14)
        # too-close exits: the norms are recorded, the matrix is left alone (bfgs.rs:103-109)
        sol.secant_update(1e-12 * s_, y)
        assert sol.next_iterate_too_close() and np.array_equal(sol.approx_inv_hessian(), got)
    with pytest.raises(qn.ErrorInputParams):
        qn.GradientDescent(1e-8, np.zeros(4)).secant_update(np.ones(4), np.ones(4))
    assert np.array_equal(qn.GradientDescent(1e-8, np.zeros(4)).compute_direction((0.0, np.arange(4.0))), -np.arange(4.0))


def test_bounded_solver_direction_hook_is_projected(qn, qo):
    """BFGSB / DFPB / SR1B::compute_direction is P(x - H g) - x with the solver's box (bfgs_b.rs:66-77, dfp_b.rs, sr1_b.rs), not
    -H g: the hook must not hand a binding a direction that leaves the box."""
    n = 130
    rng = np.random.default_rng(5)
    m = rng.standard_normal((n, n))
    h0 = m @ m.T / n + np.eye(n)
    g = rng.standard_normal(n)
    lb, ub = -0.3 * np.ones(n), 0.4 * np.ones(n)
    x0 = rng.uniform(-1.0, 1.0, n)  # projected onto the box by the constructor (bfgs_b.rs:49)
    for mk in (qn.BFGSB, qn.DFPB, qn.SR1B):
        sol = mk(1e-10, x0, lb, ub)
        sol.set_approx_inv_hessian(h0)
        x = sol.x()
        assert np.array_equal(x, np.minimum(np.maximum(x0, lb), ub))
        d = sol.compute_direction((0.0, g))
        want = np.minimum(np.maximum(x - h0 @ g, lb), ub) - x
        assert np.linalg.norm(d - want) <= 1e-12 * max(1.0, np.linalg.norm(want))
        assert np.all(x + d >= lb - 1e-15) and np.all(x + d <= ub + 1e-15)
        assert np.linalg.norm(d + h0 @ g) > 1e-3  # (the box is active: the unprojected direction is something else)
