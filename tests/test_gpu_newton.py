"""GPU tests of row f2: Newton (src/newton/mod.rs) with the dense Hessian solve on the GPU
(blocked Cholesky with f64 MFMA + triangular solves), against the oracle's restatement (inverse by LU), the
reference's own unit tests, and BASELINE.json config 4's size (n = 8192) through properties."""
import numpy as np
import pytest

import problems as P
from test_gpu_parity import _ls

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lsname", ["mt", "bt"])
def test_newton_rs_unit_tests(qn, qo, lsname):
    """newton/mod.rs:76-163 newton_morethuente / newton_backtracking: gamma = 1222, tol 1e-8, caps 1000/100."""
    gamma = 1222.0
    n_calls = [0]

    def oracle(x):
        n_calls[0] += 1
        f = 0.5 * (x[0] ** 2 + gamma * x[1] ** 2)
        g = np.array([x[0], gamma * x[1]])
        hessian = np.array([[1.0, 0.0], [0.0, gamma]])
        return qn.FuncEvalMultivariate.new(f, g).with_hessian(hessian)

    nt = qn.Newton.new(1e-8, [1.0, 1.0])
    nt.minimize(_ls(qn, lsname), oracle, 1000, 100, None)  # .unwrap()
    ev = oracle(nt.xk())
    assert nt.has_converged(ev)
    assert abs(ev.f() - 0.0) < 1e-6
    # the oracle restatement: x = (0, 0) exactly after 2 iterations, 5 closure calls
    ref = qo.Solver(qo.NEWTON, 1e-8, [1.0, 1.0])
    ref.set_hessian(lambda x: np.array([[1.0, 0.0], [0.0, gamma]]))
    o = qo.PyOracle(lambda x: (0.5 * (x[0] ** 2 + gamma * x[1] ** 2), np.array([x[0], gamma * x[1]])))
    assert ref.minimize(_ls(qo, lsname), o, 1000, 100) == qo.OK
    assert list(nt.xk()) == list(ref.x) == [0.0, 0.0]
    assert nt.k() == ref.k == 2
    assert nt.decrement_squared() == ref.decrement_squared


@pytest.mark.parametrize("n,kappa", [(3, 10.0), (5, 50.0), (17, 100.0), (64, 1e3), (200, 1e3), (777, 1e3)])
def test_newton_on_quadratic_vs_oracle(qn, qo, n, kappa):
    q, b, x0, diag = P.synth_problem(qo, n, kappa)
    oq = qo.QuadraticOracle(q, b)
    ref = qo.Solver(qo.NEWTON, 1e-8, x0)
    ref.set_hessian(oq)
    st_ref = ref.minimize(qo.morethuente(), oq, 50, 20, trace_cap=50, trace_x=True)
    obj = qn.Quadratic(q, b)
    s = qn.Newton(1e-8, x0)
    s.set_trace(50, with_x=True)
    s.minimize(qn.MoreThuente(), obj, 50, 20)
    assert st_ref == qo.OK and s.k() == ref.k
    tr, xs = s.trace()
    xstar = np.linalg.solve(q, b)
    assert np.linalg.norm(s.x() - xstar) <= 1e-9 * max(1.0, np.linalg.norm(xstar))
    # the first Newton step solves the quadratic: t = 1 and x_1 = Q^-1 b to rounding, as in the restatement
    assert tr[0]["t"] == ref.trace[0]["t"] == 1.0
    assert np.linalg.norm(xs[0] - ref.trace_x[0]) <= 1e-9 * max(1.0, np.linalg.norm(ref.trace_x[0]))
    assert abs(tr[0]["f"] - ref.trace[0]["f"]) <= 1e-10 * max(1.0, abs(ref.trace[0]["f"]))


def test_newton_host_closure_with_hessian_general_path(qn, qo):
    n = 40
    q, b, x0, _ = P.synth_problem(qo, n, 100.0)
    calls = [0]

    def oracle(x):
        calls[0] += 1
        qx = q @ x
        return qn.FuncEvalMultivariate(0.5 * x @ qx - b @ x, qx - b).with_hessian(q)

    s = qn.Newton(1e-8, x0)
    s.minimize(qn.MoreThuente(), oracle, 50, 20)
    oq = qo.QuadraticOracle(q, b)
    ref = qo.Solver(qo.NEWTON, 1e-8, x0)
    ref.set_hessian(oq)
    ref.minimize(qo.morethuente(), oq, 50, 20)
    assert s.k() == ref.k and calls[0] >= oq.calls  # every call of the reference sequence is made (plus the Hessian fetches)
    assert np.linalg.norm(s.x() - ref.x) <= 1e-9 * max(1.0, np.linalg.norm(ref.x))


def test_newton_singular_hessian_falls_back_to_gradient_direction(qn, qo):
    """newton/mod.rs:43-46: a singular Hessian gives d = -g and leaves the decrement untouched."""
    g5 = P.g5_ill_conditioned()

    def oracle(x):
        f, g = g5["fn"](x)
        return qn.FuncEvalMultivariate(f, g).with_hessian(np.zeros((2, 2)))

    s = qn.Newton(1e-12, g5["x0"])
    with pytest.raises(qn.MaxIterReached):
        s.minimize(qn.BackTracking(1e-4, 0.5), oracle, 25, 100)
    ref = qo.Solver(qo.NEWTON, 1e-12, g5["x0"])
    ref.set_hessian(lambda x: np.zeros((2, 2)))
    assert ref.minimize(qo.backtracking(1e-4, 0.5), qo.PyOracle(g5["fn"]), 25, 100) == qo.MAX_ITER_REACHED
    assert s.decrement_squared() is None and ref.decrement_squared is None
    assert np.array_equal(s.x(), ref.x)  # same gradient-descent iterates, bit for bit (n <= 5: reference order)


def test_newton_requires_a_hessian(qn):
    s = qn.Newton(1e-8, [1.0, 1.0])
    with pytest.raises((qn.SolverError, RuntimeError)):
        s.minimize(qn.MoreThuente(), lambda x: (x @ x, 2 * x), 5, 5)  # "Hessian not available in the oracle"


def test_config4_newton_n8192(qn, qo):
    """BASELINE.json config 4: Newton on an n = 8192 quadratic, dense Hessian solve on the GPU."""
    n = 8192
    diag = P.synth_diag(n)
    b, x0 = P.synth_vectors(n)
    obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
    s = qn.Newton(1e-8, x0)
    s.set_trace(10, with_x=True)
    s.minimize(qn.MoreThuente(), obj, 10, 20)  # Ok(())
    tr, xs = s.trace()
    assert s.k() == 2 and tr[0]["t"] == 1.0
    g1 = obj(xs[0]).g()
    g0 = obj(x0).g()
    assert np.linalg.norm(g1) <= 1e-10 * np.linalg.norm(g0)  # one Newton step solves Q x = b
    assert s.decrement_squared() * 0.5 < 1e-8 and s.has_converged()
    # direction check against an independent solve of the same system (numpy LAPACK on the host)
    q = qo.synth_rows(n, 0, n, P.SEED, diag, nthreads=qo.max_threads())
    xstar = np.linalg.solve(q, b)
    assert np.linalg.norm(xs[0] - xstar) <= 1e-9 * np.linalg.norm(xstar)
