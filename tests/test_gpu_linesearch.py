"""GPU tests of `LineSearch::compute_step_len` on its own (line_search/mod.rs:14-23), the way the reference's own
line-search unit tests call it: a hand-rolled gradient descent around the line search (backtracking.rs:65-113,
morethuente.rs:303-352, morethuente_b.rs:330-385).  Expected values: the reference's assertion and the oracle's step
sequence (n = 2: reference-order arithmetic, so every step length must be bit-identical)."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu
INF = float("inf")


def _descent(step, fn, x0, max_iter, cap):
    it = np.array(x0, dtype=float)
    ts = []
    k = 1
    while max_iter > k and len(ts) < cap:
        f, g = fn(it)
        if g @ g < 1e-12:
            break
        d = -g
        t = step(it, f, g, d)
        ts.append(t)
        it = it + t * d
        k += 1
    return it, ts


@pytest.mark.parametrize("lsname,max_iter,cap", [("bt", 1000, 1000), ("mt", 10000, 400), ("mtb", 10000, 400), ("btb", 1000, 1000)])
def test_reference_line_search_unit_tests(qn, qo, lsname, max_iter, cap):
    prob = P.g5_ill_conditioned()  # f = 1/2 (x0^2 + 90 x1^2), x0 = (180, 152)
    fn = prob["fn"]
    lo, hi = [-INF, -INF], [INF, INF]
    gpu_ls = {"bt": lambda: qn.BackTracking(1e-4, 0.5), "mt": qn.MoreThuente, "mtb": lambda: qn.MoreThuenteB.new(2),
              "btb": lambda: qn.BackTrackingB.new(1e-4, 0.5, lo, hi)}[lsname]()
    ref_ls = {"bt": lambda: qo.backtracking(1e-4, 0.5), "mt": qo.morethuente, "mtb": lambda: qo.morethuente_b(2, lo, hi),
              "btb": lambda: qo.backtracking_b(1e-4, 0.5, lo, hi)}[lsname]()
    x_gpu, t_gpu = _descent(lambda x, f, g, d: gpu_ls.compute_step_len(x, (f, g), d, fn, max_iter), fn, prob["x0"], max_iter, cap)
    x_ref, t_ref = _descent(lambda x, f, g, d: qo.compute_step_len(ref_ls, x, f, g, d, fn, max_iter), fn, prob["x0"], max_iter, cap)
    assert t_gpu == t_ref and np.array_equal(x_gpu, x_ref)
    if len(t_gpu) < cap:  # ran to the reference's stopping rule: its assertion
        assert abs(x_gpu[0]) < 1e-6


@pytest.mark.parametrize("n", [64, 700])
@pytest.mark.parametrize("lsname", ["mt", "bt"])
def test_compute_step_len_on_a_device_objective(qn, qo, n, lsname):
    q, b, x0, _ = P.synth_problem(qo, n)
    obj = qn.Quadratic(q, b)
    oq = qo.QuadraticOracle(q, b)
    rng = np.random.default_rng(3)
    gpu_ls = qn.MoreThuente() if lsname == "mt" else qn.BackTracking(1e-4, 0.5)
    ref_ls = qo.morethuente() if lsname == "mt" else qo.backtracking(1e-4, 0.5)
    x = x0.copy()
    for trial in range(4):
        f, g = oq(x)
        d = -g * rng.uniform(0.2, 3.0)  # a descent direction with a step scale the line search has to find
        t = gpu_ls.compute_step_len(x, (f, g), d, obj, 30)
        t_ref = qo.compute_step_len(ref_ls, x, f, g, d, oq, 30)
        assert t > 0 and abs(t - t_ref) <= 1e-9 * abs(t_ref)
        assert oq(x + t * d)[0] < f  # Armijo at least
        x = x + t * d


def test_more_thuente_b_keeps_its_clipped_t_max(qn, qo):
    """morethuente_b.rs:185-201: t_max = min(t_max, distance to the box along d) and the clip persists in the object."""
    fn = lambda x: (0.5 * (x[0] ** 2 + 3.0 * x[1] ** 2), np.array([x[0], 3.0 * x[1]]))  # noqa: E731
    lo, hi = [-1.0, -INF], [INF, INF]
    ls = qn.MoreThuenteB.new(2).with_lower_bound(lo).with_upper_bound(hi)
    ref = qo.morethuente_b(2, lo, hi)
    x = np.array([4.0, 1.0])
    f, g = fn(x)
    d = np.array([-10.0, -1.0])  # reaches x0 = -1 at t = 0.5
    t = ls.compute_step_len(x, (f, g), d, fn, 20)
    t_ref = qo.compute_step_len(ref, x, f, g, d, fn, 20)
    assert t == t_ref and t <= 0.5
    assert ls.s.t_max == ref.t_max == 0.5


def test_wolfe_condition_predicates_of_the_line_search_structs(qn):
    """SufficientDecreaseCondition / CurvatureCondition / WolfeConditions (line_search/mod.rs:25-83) on the mirror's line-search
    structs: the weak and the strong curvature test, Armijo, and both conjunctions, against the formulas"""
    rng = np.random.default_rng(12)
    n = 300
    g0, g1, d = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    d = -g0 + 0.1 * d  # a descent direction
    mt = qn.MoreThuente().with_c1(1e-3).with_c2(0.8)
    bt = qn.BackTracking(0.2, 0.5)
    assert mt.c1() == 1e-3 and mt.c2() == 0.8 and bt.c1() == 0.2
    gd0, gd1 = float(g0 @ d), float(g1 @ d)
    for t, f0, f1 in [(1.0, 3.0, 3.0 + 1e-3 * gd0 * 0.5), (0.5, 3.0, 2.0), (1.0, 3.0, 3.5), (2.0, 0.0, 2.0 * 0.2 * gd0)]:
        assert mt.sufficient_decrease(f0, f1, g0, t, d) == (f1 - f0 <= 1e-3 * t * gd0)
        assert bt.sufficient_decrease(f0, f1, g0, t, d) == (f1 - f0 <= 0.2 * t * gd0)
    for scale in (0.1, 0.79, 0.81, 1.5, -0.5, -1.2):
        gk1 = g1 - ((gd1 - scale * gd0) / (d @ d)) * d  # g1 moved so that g1.d = scale * g0.d
        weak, strong = mt.curvature_condition(g0, gk1, d), mt.strong_curvature_condition(g0, gk1, d)
        assert weak == (scale * gd0 >= 0.8 * gd0) and strong == (abs(scale * gd0) <= 0.8 * abs(gd0)), scale
        assert mt.wolfe_conditions_with_directional_derivative(3.0, 2.0, g0, gk1, 0.5, d) == weak
        assert mt.strong_wolfe_conditions_with_directional_derivative(3.0, 2.0, g0, gk1, 0.5, d) == strong
        assert not mt.strong_wolfe_conditions_with_directional_derivative(3.0, 3.5, g0, gk1, 1.0, d)
