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


# ---- Hessians that are not symmetric positive definite: the reference inverts them (LU), it does not reject them ----
def _double_well_chain(n, seed=4, c=0.05):
    """f = sum 1/4 (x_i^2 - a_i)^2 + c/2 sum (x_{i+1} - x_i)^2: the Hessian diag(3 x^2 - a) + c * tridiag(-1, 2, -1) is indefinite
    wherever some |x_i| < sqrt(a_i / 3).  The start puts a third of the coordinates deep inside the concave region and the rest
    well outside it, so the spectrum splits into [-2, -0.3] and [2, 6]: indefinite AND well conditioned (the comparison with the
    oracle is then a tolerance-level statement, not a conditioning lottery)."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(0.5, 2.0, n)
    x0 = rng.uniform(1.2, 1.5, n) * rng.choice([-1.0, 1.0], n)
    x0[::3] = rng.uniform(-0.1, 0.1, x0[::3].size)

    def fn(x):
        w = x * x - a
        d = np.diff(x)
        g = x * w
        g[:-1] -= c * d
        g[1:] += c * d
        return 0.25 * np.sum(w * w) + 0.5 * c * np.sum(d * d), g

    def hess(x):
        h = np.diag(3.0 * x * x - a)
        lap = 2.0 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
        lap[0, 0] = lap[-1, -1] = 1.0
        return h + c * lap
    return fn, hess, x0


def _newton_pair(qn, qo, fn, hess, x0, ls_gpu, ls_ref, iters, tol=1e-10, force_lu=False):
    ref = qo.Solver(qo.NEWTON, tol, x0)
    ref.set_hessian(hess)
    st_ref = ref.minimize(ls_ref, qo.PyOracle(fn), iters, 20, trace_cap=iters, trace_x=True)
    s = qn.Newton(tol, x0)
    if force_lu:
        s.set_option("newton_pivoted_lu", 1)
    s.set_trace(iters, with_x=True)
    st = qo.OK
    try:
        s.minimize(ls_gpu, lambda x: qn.FuncEvalMultivariate(*fn(x)).with_hessian(hess(x)), iters, 20)
    except qn.MaxIterReached:
        st = qo.MAX_ITER_REACHED
    return s, st, ref, st_ref


@pytest.mark.parametrize("n", [2, 64, 777])
@pytest.mark.parametrize("kind", ["indefinite", "nonsymmetric"])
def test_newton_inverts_indefinite_and_nonsymmetric_hessians(qn, qo, n, kind):
    """newton/mod.rs:36-41: `try_inverse` (LU with partial pivoting) succeeds on any non-singular matrix; only an exactly
    singular one takes the -g branch.  n = 2 runs the reference-order kernel, 64 and 777 the blocked pivoted LU (qn_lu.hip.h):
    after the Cholesky attempt reports a non-positive pivot (indefinite) or straight away (not symmetric bit for bit)."""
    fn, hess0, x0 = _double_well_chain(n)
    if kind == "nonsymmetric":  # the oracle hands over H + a skew part: still inverted as given
        rng = np.random.default_rng(8)
        k = 0.2 * np.triu(rng.standard_normal((n, n)), 1) / np.sqrt(n)
        hess = lambda x: hess0(x) + k - k.T  # noqa: E731
    else:
        hess = hess0
    ev = np.linalg.eigvalsh(hess0(x0))
    assert ev[0] < 0.0 < ev[-1]  # indefinite at the start
    iters = 4
    s, st, ref, st_ref = _newton_pair(qn, qo, fn, hess, x0, qn.MoreThuente(), qo.morethuente(), iters)
    tr, xs = s.trace()
    assert st == st_ref and len(tr) == len(ref.trace) >= 1
    # the first direction is the Newton direction -H^-1 g of the indefinite matrix, not -g
    f0, g0 = fn(x0)
    d_newton = -np.linalg.solve(hess(x0), g0)
    step0 = xs[0] - x0
    if tr[0]["t"] > 0:
        cosang = step0 @ d_newton / (np.linalg.norm(step0) * np.linalg.norm(d_newton))
        assert cosang > 1.0 - 1e-9
        assert abs(step0 @ g0) < (1.0 - 1e-6) * np.linalg.norm(step0) * np.linalg.norm(g0)  # not the gradient direction
    for k_, (a, b) in enumerate(zip(tr, ref.trace)):
        assert (a["ls_cases"], a["n_evals"]) == (b["ls_cases"], b["n_evals"]), (k_, a, b)
        assert abs(a["t"] - b["t"]) <= 1e-8 * abs(b["t"]), (k_, a["t"], b["t"])
        assert np.linalg.norm(xs[k_] - ref.trace_x[k_]) <= 1e-8 * max(1.0, np.linalg.norm(ref.trace_x[k_])), k_
    assert s.decrement_squared() is not None and ref.decrement_squared is not None
    assert abs(s.decrement_squared() - ref.decrement_squared) <= 1e-7 * max(1.0, abs(ref.decrement_squared))


def test_newton_device_objective_with_indefinite_matrix(qn, qo):
    """device-resident quadratic whose matrix has negative eigenvalues: Cholesky reports it, the pivoted LU solves it"""
    n = 200
    q, b, x0, _ = P.synth_problem(qo, n, 100.0)
    q = q.copy()
    idx = np.arange(0, n, 7)
    q[idx, idx] *= -1.0
    assert np.linalg.eigvalsh(q)[0] < 0
    oq = qo.QuadraticOracle(q, b)
    ref = qo.Solver(qo.NEWTON, 1e-8, x0)
    ref.set_hessian(oq)
    # ONE iteration: the step lands on the saddle point Q^-1 b, where the next gradient is rounding noise
    st_ref = ref.minimize(qo.backtracking(1e-4, 0.5), oq, 1, 30, trace_cap=3, trace_x=True)
    s = qn.Newton(1e-8, x0)
    s.set_trace(3, with_x=True)
    st = qo.OK
    try:
        s.minimize(qn.BackTracking(1e-4, 0.5), qn.Quadratic(q, b), 1, 30)
    except qn.MaxIterReached:
        st = qo.MAX_ITER_REACHED
    tr, xs = s.trace()
    assert st == st_ref and len(tr) == len(ref.trace)
    for k_, (a, r) in enumerate(zip(tr, ref.trace)):
        assert a["n_evals"] == r["n_evals"] and abs(a["t"] - r["t"]) <= 1e-9 * abs(r["t"])
        assert np.linalg.norm(xs[k_] - ref.trace_x[k_]) <= 1e-8 * max(1.0, np.linalg.norm(ref.trace_x[k_]))
    xs_saddle = np.linalg.solve(q, b)
    assert tr[0]["t"] == 1.0 and st == qo.MAX_ITER_REACHED
    assert np.linalg.norm(xs[0] - xs_saddle) <= 1e-8 * np.linalg.norm(xs_saddle)


@pytest.mark.parametrize("n", [64, 300, 700])
def test_newton_exactly_singular_large_hessian_takes_the_gradient_direction(qn, qo, n):
    """n > 5: two identical rows make a pivot column exactly zero during the elimination -> d = -g, decrement untouched.
    n = 700: the zero pivot column turns up in the TENTH panel -- with the look-ahead's second stream running and inside the
    one-launch panel, whose other workgroups are waiting on counters at that moment: they must all leave (round 4)."""
    fn, hess0, x0 = _double_well_chain(n, seed=6)
    r1, r2 = (5, 9) if n <= 300 else (n - 100, n - 40)

    def hess(x):
        h = np.zeros((n, n))
        h[np.arange(n), np.arange(n)] = 1.0 + np.arange(n) % 3
        h[r1, :] = 0.0
        h[r2, :] = 0.0
        h[r1, r1] = h[r1, r2] = h[r2, r1] = h[r2, r2] = 2.0  # rows r1 and r2 identical: singular, symmetric, PSD
        return h
    s, st, ref, st_ref = _newton_pair(qn, qo, fn, hess, x0, qn.BackTracking(1e-4, 0.5), qo.backtracking(1e-4, 0.5), 3)
    assert st == st_ref
    assert s.decrement_squared() is None and ref.decrement_squared is None
    tr, xs = s.trace()
    g0 = fn(x0)[1]
    step0 = xs[0] - x0
    assert abs(step0 @ g0 + np.linalg.norm(step0) * np.linalg.norm(g0)) <= 1e-12 * np.linalg.norm(step0) * np.linalg.norm(g0)
    assert np.linalg.norm(xs[-1] - ref.trace_x[-1]) <= 1e-9 * max(1.0, np.linalg.norm(ref.trace_x[-1]))


@pytest.mark.parametrize("split_min_rows", [4160, 0])  # (0: every panel's pivot chain shared by four workgroups, csrc/qn_lu_split.hip.h -- its waits expire too)
def test_one_launch_panel_whose_waits_expire_falls_back_to_step_launches(qn, qo, split_min_rows):
    """qn_lu.hip.h: the one-launch panel / sweep kernels wait for each other on counters with BOUNDED waits; when one expires (their
    workgroups were not placed together) the kernel sets *fail = 2, everybody leaves, and the host runs the factorisation again with one
    launch per sub-panel -- for good.  set_option("lu_force_wait_expiry", 1) makes every wait that is not satisfied at once expire: same iterates, bit
    for bit, and more launches (the abandoned attempt plus 17 per panel)."""
    n = 700
    fn, hess0, x0 = _double_well_chain(n)
    rng = np.random.default_rng(8)
    k = 0.2 * np.triu(rng.standard_normal((n, n)), 1) / np.sqrt(n)
    hess = lambda x: hess0(x) + k - k.T  # noqa: E731
    runs = []
    for forced in (False, True):
        s = qn.Newton(1e-10, x0)
        s.set_option("lu_split_min_rows", split_min_rows)
        if forced:
            s.set_option("lu_force_wait_expiry", 1)
        s.set_trace(2, with_x=True)
        try:
            s.minimize(qn.MoreThuente(), lambda x: qn.FuncEvalMultivariate(*fn(x)).with_hessian(hess(x)), 2, 20)
        except qn.MaxIterReached:
            pass
        tr, xs = s.trace()
        runs.append((tr, xs, s.stats()["launches"]))
        # the expiry is visible to the caller (qn_stats.newton_lu_sync_timeouts, ABI version 4; one line on stderr per process)
        assert (s.stats()["newton_lu_sync_timeouts"] >= 1) == forced
    assert np.array_equal(runs[0][1], runs[1][1]) and [r["f"] for r in runs[0][0]] == [r["f"] for r in runs[1][0]]
    assert runs[1][2] > runs[0][2] + 100


def test_pivoted_lu_with_a_panel_taller_than_the_panel_buffer(qn, qo):
    """n = 8256: the first panel has 8256 rows -- more than the column-major panel buffer holds (8192) -- and is factorised by the
    per-column kernels in W itself, the other 128 panels in the buffer with round 4's one-launch panel and two-launch look-ahead: the
    hand-over between the two (look-ahead through W, the next panel loaded into the buffer) against the Cholesky path on the same
    SPD matrix."""
    n = 8256
    diag = P.synth_diag(n)
    b, x0 = P.synth_vectors(n)
    obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
    out = []
    for force in (False, True):
        s = qn.Newton(1e-8, x0)
        if force:
            s.set_option("newton_pivoted_lu", 1)
        s.set_trace(3, with_x=True)
        s.minimize(qn.MoreThuente(), obj, 10, 20)
        out.append((s.k(), s.trace()[1][0]))
    assert out[0][0] == out[1][0] == 2
    assert np.linalg.norm(out[0][1] - out[1][1]) <= 1e-9 * np.linalg.norm(out[0][1])
    g1 = obj(out[1][1]).g()
    assert np.linalg.norm(g1) <= 1e-10 * np.linalg.norm(obj(x0).g())


@pytest.mark.parametrize("n", [64, 130, 777, 1500])
def test_pivoted_lu_path_agrees_with_the_cholesky_path_on_spd(qn, qo, n):
    """diagnostics knob rows = -5: the same convex problem through both factorisations"""
    q, b, x0, _ = P.synth_problem(qo, n, 1e3)
    out = []
    for force in (False, True):
        s = qn.Newton(1e-8, x0)
        if force:
            s.set_option("newton_pivoted_lu", 1)
        s.set_trace(5, with_x=True)
        s.minimize(qn.MoreThuente(), qn.Quadratic(q, b), 50, 20)
        out.append((s.k(), s.x(), s.decrement_squared(), s.trace()[1][0]))
    assert out[0][0] == out[1][0]
    assert np.linalg.norm(out[0][3] - out[1][3]) <= 1e-9 * np.linalg.norm(out[0][3])
    xstar = np.linalg.solve(q, b)
    assert np.linalg.norm(out[1][1] - xstar) <= 1e-9 * np.linalg.norm(xstar)


@pytest.mark.parametrize("n", [64, 200, 777, 1300, 4200])
def test_panel_lu_equals_the_per_column_lu_bit_for_bit(qn, qo, n):
    """qn_lu.hip.h: the 64-column panel is factorised four columns at a time by one workgroup that holds the rows in registers
    (19 launches per panel); set_option("lu_per_column_panel", 1) selects rounds 1-2's two launches per column.  Same pivots (first maximum), same
    arithmetic in the same order: the iterates are the same bits.  Non-symmetric, indefinite Hessian (row swaps do occur).
    n = 4200: panels of more than 2048 and more than 4096 rows -- the 8- and 16-rows-per-thread instantiations of the step kernel,
    which are the ones config 4 (n = 8192) runs.
    Round 4: from 8 panels on the default path also runs the LOOK-AHEAD (the trailing update split between the solver's stream and a
    CU-masked second stream, the U12 solve row by row with scalar multipliers, the update as a resident grid that loops);
    set_option("lu_lookahead", 0) is the single-stream path with rounds 1-3's kernels -- the same bits again.  And the panel itself is ONE launch
    (58 workgroups waiting for each other on counters, lu_panel_persist_kernel); set_option("lu_one_launch_panel", 0) is round 3's launch per sub-panel.
    Round 6: the pivot chain of that launch shared by the workgroups of one XCD (records through memory, one hop per pivot step):
    set_option("lu_split_role_a", 1 | 2 | 4)."""
    fn, hess0, x0 = _double_well_chain(n)
    iters = 3 if n <= 2000 else 1
    rng = np.random.default_rng(8)
    k = 0.2 * np.triu(rng.standard_normal((n, n)), 1) / np.sqrt(n)
    hess = lambda x: hess0(x) + k - k.T  # noqa: E731
    runs = []
    # (round 6) the default panel shares its pivot chain between FOUR workgroups (csrc/qn_lu_split.hip.h); parts = 1 is rounds 4-5's one workgroup, 2 the other split
    for percol, no_la, no_persist, parts in ((False, False, False, 4), (True, True, True, 4), (False, True, False, 4), (False, False, True, 4),
                                             (False, False, False, 1), (False, False, False, 2), (False, True, False, 2)):
        s = qn.Newton(1e-10, x0)
        s.set_option("lu_split_role_a", parts)
        s.set_option("lu_split_min_rows", 0)  # (every panel; by default only panels of 4160 rows and more are split)
        if percol:
            s.set_option("lu_per_column_panel", 1)
        if no_la:
            s.set_option("lu_lookahead", 0)
        if no_persist:
            s.set_option("lu_one_launch_panel", 0)
        s.set_trace(iters, with_x=True)
        try:
            s.minimize(qn.MoreThuente(), lambda x: qn.FuncEvalMultivariate(*fn(x)).with_hessian(hess(x)), iters, 20)
        except qn.MaxIterReached:
            pass
        tr, xs = s.trace()
        runs.append((tr, xs, s.stats()["launches"]))
        assert s.stats()["newton_lu_sync_timeouts"] == 0  # (no path got here through an expired wait's fallback)
    key = lambda tr: [(r["f"], r["gnorm"], r["t"], r["n_evals"], r["ls_cases"]) for r in tr]  # noqa: E731  (s_norm / y_norm are NaN: Newton has none)
    assert key(runs[0][0]) == key(runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    assert key(runs[0][0]) == key(runs[2][0]) and np.array_equal(runs[0][1], runs[2][1])
    assert key(runs[0][0]) == key(runs[3][0]) and np.array_equal(runs[0][1], runs[3][1])
    for other in runs[4:]:
        assert key(runs[0][0]) == key(other[0]) and np.array_equal(runs[0][1], other[1])
    assert runs[0][2] == runs[4][2]  # (the same launches: the split changes a grid, not a count)
    assert runs[0][2] < runs[1][2]
    # and the direction is the Newton direction of the matrix as given
    f0, g0 = fn(x0)
    d_newton = -np.linalg.solve(hess(x0), g0)
    step0 = runs[0][1][0] - x0
    assert step0 @ d_newton / (np.linalg.norm(step0) * np.linalg.norm(d_newton)) > 1.0 - 1e-9
