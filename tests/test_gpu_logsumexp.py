"""GPU tests of row f1: DFP / BFGS + More-Thuente on the log-sum-exp objective
f = log sum_i exp(a_i'x + c_i) + mu/2 ||x||^2 -- a non-quadratic objective, evaluated by its own kernels on the generic
path -- against the CPU oracle, and BASELINE.json config 5's size through size-independent properties.  (On this family
More-Thuente accepts t = 1 in almost every iteration; its cases 2-4 are driven by tests/test_gpu_mt_cases.py.)"""
import numpy as np
import pytest

from test_gpu_parity import _compare, _ls

pytestmark = pytest.mark.gpu


def _problem(m, n, seed=3, scale=1.0):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((m, n)) * scale / np.sqrt(n)
    c = rng.standard_normal(m)
    x0 = rng.standard_normal(n)
    return a, c, x0


@pytest.mark.parametrize("m,n", [(7, 5), (40, 13), (300, 200), (1030, 515)])
def test_logsumexp_evaluation_vs_oracle(qn, qo, m, n):
    a, c, x0 = _problem(m, n)
    mu = 0.05
    obj = qn.LogSumExp(a, c, mu)
    ev = obj(x0)
    f_ref, g_ref = qo.LogSumExpOracle(a, c, mu)(x0)
    assert abs(ev.f() - f_ref) <= 1e-12 * max(1.0, abs(f_ref))
    assert np.linalg.norm(ev.g() - g_ref) <= 1e-12 * max(1.0, np.linalg.norm(g_ref))
    assert np.array_equal(obj.rows(0, m), a)


@pytest.mark.parametrize("method", ["dfp", "bfgs"])
@pytest.mark.parametrize("lsname", ["mt", "bt"])
def test_quasi_newton_on_logsumexp_vs_oracle(qn, qo, method, lsname):
    m, n = 600, 384
    a, c, x0 = _problem(m, n, scale=3.0)
    mu, iters = 0.1, 30
    ref = qo.Solver(qo.DFP if method == "dfp" else qo.BFGS, 1e-10, x0, qo.UPDATE_AS_WRITTEN)
    o = qo.LogSumExpOracle(a, c, mu, nthreads=4)
    st_ref = ref.minimize(_ls(qo, lsname), o, iters, 20, trace_cap=iters, trace_x=True)
    obj = qn.LogSumExp(a, c, mu)
    s = (qn.DFP if method == "dfp" else qn.BFGS)(1e-10, x0)
    s.set_trace(iters, with_x=True)
    try:
        s.minimize(_ls(qn, lsname), obj, iters, 20)
        st = 0
    except qn.MaxIterReached:
        st = 1
    tr, xs = s.trace()
    w = _compare(tr, xs, ref.trace, ref.trace_x)
    if w == len(ref.trace):
        assert st == st_ref
    # the non-quadratic objective takes the cheap path most of the time: t = 1 accepted => 3 calls per iteration
    if lsname == "mt":
        assert sum(1 for r in tr[:w] if r["n_evals"] == 3) >= 1
    st_ = s.stats()
    assert st_["oracle_evals"] < st_["oracle_calls"]  # memoised: loop-top and bfgs.rs:98 calls are not re-evaluated


def test_config5_size_properties_dfp_morethuente_n16384(qn, qo):
    """BASELINE.json config 5: DFP + More-Thuente, n = 16384 log-sum-exp, f64 (one GPU here; rows shard across ranks)."""
    n = m = 16384
    rng = np.random.default_rng(11)
    a = rng.standard_normal((m, n)) * (2.0 / np.sqrt(n))
    c = rng.standard_normal(m)
    x0 = rng.standard_normal(n)
    mu, iters = 0.1, 6
    obj = qn.LogSumExp(a, c, mu)
    s = qn.DFP(1e-10, x0)
    s.set_trace(iters, with_x=True)
    with pytest.raises(qn.MaxIterReached):
        s.minimize(qn.MoreThuente(), obj, iters, 20)
    tr, xs = s.trace()
    f = np.array([r["f"] for r in tr])
    assert len(tr) == iters and np.all(np.diff(f) < 0)  # sufficient decrease every iteration
    # f and g at the last iterate agree with the threaded CPU oracle at full size
    o = qo.LogSumExpOracle(a, c, mu, nthreads=qo.max_threads())
    f_ref, g_ref = o(xs[-1])
    ev = obj(xs[-1])
    assert abs(ev.f() - f_ref) <= 1e-11 * max(1.0, abs(f_ref))
    assert np.linalg.norm(ev.g() - g_ref) <= 1e-11 * np.linalg.norm(g_ref)
    # secant equation of the DFP update on the last pair: H+ y = s
    sk = xs[-1] - xs[-2]
    yk = ev.g() - obj(xs[-2]).g()
    h = s.approx_inv_hessian()
    assert np.array_equal(h, h.T)
    assert np.linalg.norm(h @ yk - sk) <= 1e-8 * np.linalg.norm(sk)
    # first iterations against the threaded rank-2 CPU restatement
    ref = qo.Solver(qo.DFP, 1e-10, x0, qo.UPDATE_RANK2, nthreads=qo.max_threads())
    ref.minimize(qo.morethuente(), o, iters, 20, trace_cap=iters, trace_x=True)  # (every iteration of the run)
    assert _compare(tr, xs, ref.trace, ref.trace_x) == iters


@pytest.mark.parametrize("m,n", [(9, 2), (300, 200), (1030, 515), (64, 5000), (2100, 1024), (130, 16384)])
def test_one_pass_evaluation_equals_two_pass_and_oracle(qn, qo, m, n, monkeypatch):
    """n_pad <= 16384: the objective is evaluated in ONE pass over A (running-maximum softmax, lse_onepass_kernel); the
    two-pass evaluation of round 1 (QN_LSE_TWO_PASS=1 at creation) and the oracle are the checkers.  Also points with a huge
    spread of exponents, where an unshifted softmax would overflow."""
    a, c, x0 = _problem(m, n)
    mu = 0.05
    one = qn.LogSumExp(a, c, mu)
    monkeypatch.setenv("QN_LSE_TWO_PASS", "1")
    two = qn.LogSumExp(a, c, mu)
    monkeypatch.delenv("QN_LSE_TWO_PASS")
    o = qo.LogSumExpOracle(a, c, mu, nthreads=4)
    for scale in (1.0, 40.0, 3000.0):
        x = x0 * scale
        e1, e2 = one(x), two(x)
        f_ref, g_ref = o(x)
        for e in (e1, e2):
            assert abs(e.f() - f_ref) <= 1e-12 * max(1.0, abs(f_ref)), (scale, e.f(), f_ref)
            assert np.linalg.norm(e.g() - g_ref) <= 1e-12 * max(1.0, np.linalg.norm(g_ref)), scale
        assert abs(e1.f() - e2.f()) <= 1e-13 * max(1.0, abs(e2.f()))
    # the same pass as the oracle of a solver run
    outs = []
    for obj in (one, two):
        s = qn.DFP(1e-10, x0)
        s.set_trace(8, with_x=True)
        try:
            s.minimize(qn.MoreThuente(), obj, 8, 20)
        except qn.MaxIterReached:
            pass
        outs.append(s.trace())
    (t1, x1), (t2, x2) = outs
    # (two evaluations that agree to 1e-13 give DFP trajectories that drift apart with the conditioning of the line search's
    # interpolation -- differences of nearly equal f values: same decisions, iterates loosely)
    assert [r["ls_cases"] for r in t1] == [r["ls_cases"] for r in t2]
    assert np.linalg.norm(x1 - x2) <= 1e-6 * max(1.0, np.linalg.norm(x2))
