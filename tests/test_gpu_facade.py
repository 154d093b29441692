"""GPU test of the facade that mirrors src/wasm.rs (OptimizationSolver / OptimizationResult, SURVEY.md 8(c) KAT G9):
BFGS + MoreThuente::default() with max_iter_line_search 20, the result struct's fields, and the error string."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _quad(x):  # x^2 + 2 y^2 + 3 z^2 + xy + yz as the flat [f, g...] sequence a JS objective returns
    f = x[0] ** 2 + 2 * x[1] ** 2 + 3 * x[2] ** 2 + x[0] * x[1] + x[1] * x[2]
    return [f, 2 * x[0] + x[1], 4 * x[1] + x[0] + x[2], 6 * x[2] + x[1]]


def _quad_h(x):
    h = np.array([[2.0, 1.0, 0.0], [1.0, 4.0, 1.0], [0.0, 1.0, 6.0]])
    return _quad(x) + list(h.ravel(order="F"))


def test_solve_bfgs_result_fields_match_the_oracle(qn, qo):
    r = qn.OptimizationSolver(1e-8, 50).solve_bfgs([1.0, 1.0, 1.0], _quad)
    ref = qo.Solver(qo.BFGS, 1e-8, np.array([1.0, 1.0, 1.0]))
    assert ref.minimize(qo.morethuente(), lambda x: (_quad(x)[0], np.array(_quad(x)[1:])), 50, 20) == 0
    assert r.get_success() and r.get_error_message() == ""
    assert r.get_iterations() == ref.k and np.array_equal(r.get_x(), ref.x)  # n <= 5: reference-order arithmetic, bit for bit
    assert r.get_f_value() == _quad(ref.x)[0] and r.get_gradient_norm() == float(np.sqrt(np.dot(_quad(ref.x)[1:], _quad(ref.x)[1:])))


def test_failure_is_reported_in_the_struct_not_raised(qn):
    r = qn.OptimizationSolver(1e-30, 2).solve_bfgs([1.0, 1.0, 1.0], _quad)
    assert not r.get_success() and r.get_error_message() == "Optimization failed: MaxIterReached"
    assert r.get_x() == [] and r.get_iterations() == 0 and r.get_f_value() == 0.0  # the struct keeps OptimizationResult::new()'s values


def test_solve_gradient_descent_and_newton(qn):
    s = qn.OptimizationSolver(1e-6, 2000)
    r = s.solve_gradient_descent([1.0, 1.0, 1.0], _quad)
    assert r.get_success() and r.get_gradient_norm() < 1e-5 and np.allclose(r.get_x(), 0.0, atol=1e-5)
    r = s.solve_newton([1.0, 1.0, 1.0], _quad_h)
    assert r.get_success() and r.get_iterations() <= 3 and abs(r.get_f_value()) < 1e-20
    with pytest.raises(Exception):
        s.solve_newton([1.0, 1.0, 1.0], _quad)  # no Hessian in the returned sequence: the reference panics
