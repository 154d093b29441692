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


@pytest.mark.parametrize("method", ["dfp", "bfgs"])
@pytest.mark.parametrize("lsname", ["mt", "bt"])
@pytest.mark.parametrize("m,n", [(600, 1024), (3000, 2176)])
def test_second_generation_structure_for_logsumexp_vs_oracle_and_generic_path(qn, qo, method, lsname, m, n):
    """Round 5 (qn_sym2g.hip.h): n a multiple of 128, >= 1024, one rank -- the objective runs in the structure of the second-generation
    path: the state machine in one-workgroup launches on the device (pipelined: no host round trip per request), the trial point formed
    by the pass over A from the lazy direction, an evaluation's combine launch staging g+, y, x+, s and their sums, the update pass on
    the symmetric tiles.  Against the oracle (decisions exact, steps / iterates / f to the parity tolerance), pipelined = synchronous
    bit for bit, against the generic path (set_option("second_generation", 0)), the launch contract counted, and a continued call."""
    a, c, x0 = _problem(m, n, scale=3.0)
    mu, iters = 0.1, 25
    ref = qo.Solver(qo.DFP if method == "dfp" else qo.BFGS, 1e-10, x0, qo.UPDATE_RANK2, nthreads=4)
    o = qo.LogSumExpOracle(a, c, mu, nthreads=4)
    st_ref = ref.minimize(_ls(qo, lsname), o, iters, 20, trace_cap=iters, trace_x=True)
    obj = qn.LogSumExp(a, c, mu)
    runs = []
    for tiling, sync in ((None, 0), (None, 1), (("second_generation", 0), None)):
        s = (qn.DFP if method == "dfp" else qn.BFGS)(1e-10, x0)
        s.set_trace(iters, with_x=True)
        if tiling:
            s.configure(*tiling)
        if sync is not None:
            s.set_sync_mode(sync)
        try:
            s.minimize(_ls(qn, lsname), obj, iters, 20)
            st = 0
        except qn.MaxIterReached:
            st = 1
        runs.append((s, st, *s.trace()))
    (s0, st0, tr0, xs0), (s1, st1, tr1, xs1), (sg, stg, trg, xsg) = runs
    p0, pg = s0.stats()["path"], sg.stats()["path"]
    assert p0 & 16 and p0 & 8 and p0 & 2 and not pg & 16 and pg & 4  # second-generation structure, pipelined / the generic path on the tiles
    assert st0 == st1 == stg == st_ref and tr0 == tr1 and np.array_equal(xs0, xs1)  # pipelined = synchronous, bit for bit
    assert _compare(tr0, xs0, ref.trace, ref.trace_x) == min(len(ref.trace), 25)
    _compare(tr0, xs0, trg, xsg)
    h = s0.approx_inv_hessian()
    assert np.array_equal(h, h.T) and np.abs(h - sg.approx_inv_hessian()).max() <= 1e-9 * np.abs(h).max()
    # the launch contract: per evaluation 3 (machine, pass over A, combine), per update pass 3 (machine, tiles, reduce); a period of the
    # pipelined pattern carries one evaluation slot, more once the run has needed more; the host reads the control block once per batch
    st = s0.stats()
    assert st["host_syncs"] <= 3 and sg.stats()["host_syncs"] > iters  # (the generic path: at least one round trip per request)
    assert st["launches"] <= 3 * st["oracle_evals"] + 3 * (st["h_passes"] + 1) + 6 * (st["oracle_evals"] - iters) + 8
    assert st["oracle_evals"] == sg.stats()["oracle_evals"]
    # a continued call is the same run (the memo and the lazy direction stay on the device)
    two = (qn.DFP if method == "dfp" else qn.BFGS)(1e-10, x0)
    two.set_trace(iters, with_x=True)
    for k in (10, iters - 10):
        try:
            two.minimize(_ls(qn, lsname), obj, k, 20)
        except qn.MaxIterReached:
            pass
    assert np.array_equal(two.x(), s0.x()) and np.array_equal(two.approx_inv_hessian(), h)


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
