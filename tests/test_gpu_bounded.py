"""GPU tests of row f4: the bounded quasi-Newton solvers BFGSB / DFPB / SR1B (bfgs_b.rs, dfp_b.rs, sr1_b.rs)
and the bounded line searches MoreThuenteB / BackTrackingB, against the oracle's restatement."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu
INF = float("inf")


def _box(qo, n, kappa=50.0):
    q, b, x0, _ = P.synth_problem(qo, n, kappa)
    xs = np.linalg.solve(q, b)
    lb = xs - 0.3 * np.abs(xs) - 0.05
    ub = xs + 0.1
    k = max(1, n // 4)
    lb[:k] = xs[:k] + 0.2  # these bounds are active at the constrained optimum
    ub[:k] = xs[:k] + 1.0
    return q, b, x0, lb, ub


def _make_ls(mod, name, n, lb, ub):
    oracle = not hasattr(mod, "MoreThuenteB")
    if name == "mt":
        return mod.morethuente() if oracle else mod.MoreThuente()
    if name == "mtb":
        return mod.morethuente_b(n, lb, ub) if oracle else mod.MoreThuenteB.new(n).with_lower_bound(lb).with_upper_bound(ub)
    return mod.backtracking_b(1e-4, 0.5, lb, ub) if oracle else mod.BackTrackingB.new(1e-4, 0.5, lb, ub)


def test_bfgs_b_rs_unit_test(qn, qo):
    """bfgs_b.rs:160-212 constrained_grad_desc_backtracking: gamma = 999, infinite bounds, BackTrackingB, caps 10000/1000."""
    gamma = 999.0
    fn = lambda x: (0.5 * (x[0] ** 2 + gamma * x[1] ** 2), np.array([x[0], gamma * x[1]]))  # noqa: E731
    lower, upper = [-INF, -INF], [INF, INF]
    ls = qn.BackTrackingB.new(1e-4, 0.5, lower, upper)
    gd = qn.BFGSB.new(1e-12, [180.0, 152.0], lower, upper)
    gd.minimize(ls, fn, 10000, 1000, None)  # .unwrap()
    ev = qn.FuncEvalMultivariate(*fn(gd.xk()))
    assert gd.has_converged(ev)
    assert np.max(np.abs(gd.projected_gradient(ev))) < 1e-6
    ref = qo.Solver(qo.BFGS, 1e-12, [180.0, 152.0])
    ref.set_bounds(lower, upper)
    assert ref.minimize(qo.backtracking_b(1e-4, 0.5, lower, upper), qo.PyOracle(fn), 10000, 1000) == qo.OK
    assert gd.k() == ref.k and np.array_equal(gd.xk(), ref.x)  # n = 2: reference order, bit for bit


@pytest.mark.parametrize("method", ["bfgsb", "dfpb", "sr1b"])
@pytest.mark.parametrize("lsname", ["mt", "mtb", "btb"])
@pytest.mark.parametrize("n", [3, 40, 300])
def test_bounded_solvers_vs_oracle(qn, qo, method, lsname, n):
    q, b, x0, lb, ub = _box(qo, n)
    iters = 40
    ref = qo.Solver({"bfgsb": qo.BFGS, "dfpb": qo.DFP, "sr1b": qo.SR1}[method], 1e-9, x0)
    ref.set_bounds(lb, ub)
    ls_ref = _make_ls(qo, lsname, n, lb, ub)
    st_ref = ref.minimize(ls_ref, qo.QuadraticOracle(q, b), iters, 30, trace_cap=iters, trace_x=True)
    cls = {"bfgsb": qn.BFGSB, "dfpb": qn.DFPB, "sr1b": qn.SR1B}[method]
    if n <= 5:  # reference-order path + host closure: every bit must agree with the restatement
        fn = lambda x: qo.QuadraticOracle(q, b)(x)  # noqa: E731
        s = cls.new(1e-9, x0, lb, ub)
        s.set_trace(iters, with_x=True)
        try:
            s.minimize(_make_ls(qn, lsname, n, lb, ub), fn, iters, 30)
        except qn.MaxIterReached:
            pass
        tr, xs = s.trace()
        assert len(tr) == len(ref.trace)
        assert [r["n_evals"] for r in tr] == [r["n_evals"] for r in ref.trace]
        assert np.array_equal(xs, ref.trace_x)
        return
    for memo in (1, 0):
        s = cls.new(1e-9, x0, lb, ub)
        s.memoize = memo
        s.set_trace(iters, with_x=True)
        ls = _make_ls(qn, lsname, n, lb, ub)
        st = 0
        try:
            s.minimize(ls, qn.Quadratic(q, b), iters, 30)
        except qn.MaxIterReached:
            st = 1
        tr, xs = s.trace()
        x = s.x()
        assert np.all(x >= lb - 1e-12) and np.all(x <= ub + 1e-12)  # projected directions keep every iterate feasible (to rounding)
        w = min(len(tr), len(ref.trace), 12)
        assert w >= min(3, len(ref.trace))
        for k in range(w):
            assert tr[k]["n_evals"] == ref.trace[k]["n_evals"], (k, memo)
            assert abs(tr[k]["t"] - ref.trace[k]["t"]) <= 1e-8 * abs(ref.trace[k]["t"]), (k, memo)
            assert np.linalg.norm(xs[k] - ref.trace_x[k]) <= 1e-8 * max(1.0, np.linalg.norm(ref.trace_x[k])), (k, memo)
        if len(tr) == len(ref.trace) and w == len(tr):
            assert st == st_ref
        if lsname == "mtb":  # morethuente_b.rs:201: the clipped t_max is kept by the line search object
            assert ls.t_max() <= 1e300 and abs(ls.t_max() - ls_ref.t_max) <= 1e-8 * max(1.0, abs(ls_ref.t_max))


@pytest.mark.parametrize("method", ["bfgsb", "dfpb", "sr1b"])
def test_bounded_run_to_termination_matches_oracle(qn, qo, method):
    """Full runs (the reference stops on ||s|| / ||y|| < tol, not on optimality): same exit, same point, same active set."""
    n = 64
    q, b, x0, lb, ub = _box(qo, n)
    ref = qo.Solver({"bfgsb": qo.BFGS, "dfpb": qo.DFP, "sr1b": qo.SR1}[method], 1e-9, x0)
    ref.set_bounds(lb, ub)
    oq = qo.QuadraticOracle(q, b)
    st_ref = ref.minimize(qo.morethuente(), oq, 300, 30)
    s = {"bfgsb": qn.BFGSB, "dfpb": qn.DFPB, "sr1b": qn.SR1B}[method].new(1e-9, x0, lb, ub)
    st = 0
    try:
        s.minimize(qn.MoreThuente(), qn.Quadratic(q, b), 300, 30)
    except qn.MaxIterReached:
        st = 1
    x = s.x()
    assert st == st_ref and abs(s.k() - ref.k) <= 2
    assert np.all(x >= lb - 1e-12) and np.all(x <= ub + 1e-12)
    assert abs(oq(x)[0] - oq(ref.x)[0]) <= 1e-6 * max(1.0, abs(oq(ref.x)[0]))
    active = np.isclose(x, lb) | np.isclose(x, ub)
    active_ref = np.isclose(ref.x, lb) | np.isclose(ref.x, ub)
    assert active.sum() >= n // 4 and np.array_equal(active, active_ref)


@pytest.mark.parametrize("method", ["bfgsb", "dfpb", "sr1b"])
@pytest.mark.parametrize("lsname", ["mt", "mtb", "btb"])
@pytest.mark.parametrize("n", [1024, 1408])  # (work lists read from memory; the two-items-and-a-sliver instance: the test below)
def test_bounded_second_generation_path_vs_oracle_and_generic(qn, qo, method, lsname, n):
    """BFGSB / DFPB / SR1B with More-Thuente(B) -- and, round 6, BackTrackingB: its projected trial points stored by s2_proj_kernel, the evaluation
    AT the stored point, ||P(x + t d) - x||^2 through the table (backtracking_b.rs:24-34, 52-88) -- on the second-generation symmetric path
    (s2_dir_kernel, qn_sym2.hip.h: the direction stored and
    projected by one more launch per iteration, t_max clipped where that request is consumed): against the oracle's restatement, against the
    generic path (set_option("bounded_second_generation", 0)), and pipelined against synchronous bit for bit."""
    q, b, x0, lb, ub = _box(qo, n)
    iters = 25
    # (the oracle's rank-2 form of the update -- pinned against the as-written one by the CPU tests -- on all host threads: the as-written
    # update is five n x n products per iteration)
    ref = qo.Solver({"bfgsb": qo.BFGS, "dfpb": qo.DFP, "sr1b": qo.SR1}[method], 1e-9, x0, qo.UPDATE_RANK2, nthreads=qo.max_threads())
    ref.set_bounds(lb, ub)
    ls_ref = _make_ls(qo, lsname, n, lb, ub)
    ref.minimize(ls_ref, qo.QuadraticOracle(q, b, nthreads=qo.max_threads()), iters, 30, trace_cap=iters, trace_x=True)
    cls = {"bfgsb": qn.BFGSB, "dfpb": qn.DFPB, "sr1b": qn.SR1B}[method]
    obj = qn.Quadratic(q, b)
    runs = {}
    for mode in ("pipelined", "sync", "generic"):
        s = cls.new(1e-9, x0, lb, ub)
        s.set_trace(iters, with_x=True)
        if mode == "sync":
            s.set_sync_mode(1)
        if mode == "generic":
            s.set_option("bounded_second_generation", 0)
        ls = _make_ls(qn, lsname, n, lb, ub)
        try:
            s.minimize(ls, obj, iters, 30)
        except qn.MaxIterReached:
            pass
        tr, xs = s.trace()
        st = s.stats()
        runs[mode] = (tr, xs, s.x(), ls.t_max() if lsname == "mtb" else None, st)
        assert bool(st["path"] & 16) == (mode != "generic"), (mode, st["path"])  # QN_PATH_SYM2
        x = s.x()
        assert np.all(x >= lb - 1e-12) and np.all(x <= ub + 1e-12)
        w = min(len(tr), len(ref.trace), 20)
        assert w >= min(10, len(ref.trace))
        for k in range(w):
            assert tr[k]["n_evals"] == ref.trace[k]["n_evals"], (mode, k)
            assert abs(tr[k]["t"] - ref.trace[k]["t"]) <= 1e-8 * abs(ref.trace[k]["t"]), (mode, k)
            assert np.linalg.norm(xs[k] - ref.trace_x[k]) <= 1e-8 * max(1.0, np.linalg.norm(ref.trace_x[k])), (mode, k)
        if lsname == "mtb":
            assert abs(ls.t_max() - ls_ref.t_max) <= 1e-8 * max(1.0, abs(ls_ref.t_max)), mode
    assert np.array_equal(runs["pipelined"][2], runs["sync"][2]) and np.array_equal(runs["pipelined"][1], runs["sync"][1])
    assert runs["pipelined"][3] == runs["sync"][3]
    # one launch more per iteration than the unbounded pattern, and no host round trip per request
    st = runs["pipelined"][4]
    assert st["host_syncs"] <= 6


def test_bounded_second_generation_path_continues_across_calls(qn, qo):
    """Two calls of 8 iterations = one of 16 (the second call continues warm: the stored direction and its step to the box carry over)."""
    n = 1024
    q, b, x0, lb, ub = _box(qo, n)
    obj = qn.Quadratic(q, b)
    outs = []
    for split in (False, True):
        s = qn.BFGSB.new(1e-9, x0, lb, ub)
        ls = _make_ls(qn, "mtb", n, lb, ub)
        for k in ((8, 8) if split else (16,)):
            try:
                s.minimize(ls, obj, k, 30)
            except qn.MaxIterReached:
                pass
        outs.append((s.x(), ls.t_max()))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]


def test_warm_call_with_another_line_search_box_forms_its_direction_again(qn, qo):
    """a warm bounded call kept the stored direction and its step to the line search's box (QnCtl.dir_ready, mtb_cand) whatever line
    search the call brought.  BFGSB + plain More-Thuente for 6 iterations, then the SAME solver with MoreThuenteB and a box of its own: the
    second call must clip its first t_max by THAT box (morethuente_b.rs:185-201 recomputes the candidate in every compute_step_len) -- i.e.
    equal, bit for bit on the second-generation path's own terms, a run whose second call is not warm -- and stay inside the box."""
    n = 1024
    q, b, x0, lb, ub = _box(qo, n)
    obj = qn.Quadratic(q, b)
    outs = {}
    for mode in ("second generation", "generic"):
        s = qn.BFGSB.new(1e-9, x0, lb, ub)
        if mode == "generic":
            s.set_option("bounded_second_generation", 0)
        try:
            s.minimize(qn.MoreThuente(), obj, 6, 30)
        except qn.MaxIterReached:
            pass
        x6 = s.x()
        # the second call's line search has a box of its own, tight around where the run stands: the very first step is clipped by it
        llb, lub = x6 - 1e-3, x6 + 1e-3
        ls = qn.MoreThuenteB.new(n).with_lower_bound(llb).with_upper_bound(lub)
        try:
            s.minimize(ls, obj, 1, 30)
        except qn.MaxIterReached:
            pass
        x7 = s.x()
        assert np.all(x7 >= llb - 1e-9) and np.all(x7 <= lub + 1e-9), mode  # (with the stale candidate -- +inf -- the step left the box)
        assert np.any(np.abs(x7 - x6) > 0), mode
        outs[mode] = (x7, ls.t_max())
    assert np.linalg.norm(outs["second generation"][0] - outs["generic"][0]) <= 1e-7 * max(1.0, np.linalg.norm(outs["generic"][0]))
    assert abs(outs["second generation"][1] - outs["generic"][1]) <= 1e-8 * max(1.0, abs(outs["generic"][1]))


def test_backtracking_b_at_the_benchmark_size_on_the_second_generation_path(qn, qo):
    """Round 6: BFGSB + BackTrackingB at n = 4096 -- the trial points projected INSIDE the mover / multiplier kernel's bounded
    instantiation (s2_evalr_kernel<true>: clamped where the trial point is formed, ||P(x + t d) - x||^2 as column 6 of the launch's table; default) or
    stored by a s2_proj_kernel launch per trial and evaluated AT the stored point (set_option("btb_project_in_eval", 0); and, with
    set_option("eval_mover_multiplier", 0), by round 5's two-items-and-a-sliver instance): both equal the oracle's restatement (evaluation counts, steps, iterates over the window) and the generic
    path, and each other bit for bit; pipelined equals synchronous bit for bit; no host round trip per request."""
    n = 4096
    q, b, x0, lb, ub = _box(qo, n)
    iters = 16
    ref = qo.Solver(qo.BFGS, 1e-9, x0, qo.UPDATE_RANK2, nthreads=qo.max_threads())
    ref.set_bounds(lb, ub)
    ref.minimize(_make_ls(qo, "btb", n, lb, ub), qo.QuadraticOracle(q, b, nthreads=qo.max_threads()), iters, 30, trace_cap=iters, trace_x=True)
    obj = qn.Quadratic(q, b)
    got = {}
    for mode in ("ring", "ring sync", "ring proj-launch", "pair", "generic"):
        s = qn.BFGSB.new(1e-9, x0, lb, ub)
        s.set_trace(iters, with_x=True)
        if mode == "ring sync":
            s.set_sync_mode(1)
        if mode == "ring proj-launch":  # (the projection as a launch per trial in front of the mover / multiplier kernel: the first version of this round)
            s.set_option("btb_project_in_eval", 0)
        if mode == "pair":
            s.set_option("eval_mover_multiplier", 0)
        if mode == "generic":
            s.set_option("bounded_second_generation", 0)
        try:
            s.minimize(_make_ls(qn, "btb", n, lb, ub), obj, iters, 30)
        except qn.MaxIterReached:
            pass
        tr, xs = s.trace()
        st = s.stats()
        assert bool(st["path"] & 16) == (mode != "generic")
        w = min(len(tr), len(ref.trace))
        assert w >= 12
        for k in range(w):
            assert tr[k]["n_evals"] == ref.trace[k]["n_evals"], (mode, k)
            assert abs(tr[k]["t"] - ref.trace[k]["t"]) <= 1e-8 * abs(ref.trace[k]["t"]), (mode, k)
            assert np.linalg.norm(xs[k] - ref.trace_x[k]) <= 1e-8 * max(1.0, np.linalg.norm(ref.trace_x[k])), (mode, k)
        x = s.x()
        assert np.all(x >= lb - 1e-12) and np.all(x <= ub + 1e-12)
        got[mode] = (tr, xs, x, st["launches"], st["oracle_evals"])
        if mode == "ring":
            assert st["host_syncs"] <= 8  # (one per batch of periods, not one per request: the run makes ~130 requests)
    assert got["ring"][0] == got["pair"][0] == got["ring sync"][0] == got["ring proj-launch"][0]
    assert np.array_equal(got["ring"][1], got["pair"][1]) and np.array_equal(got["ring"][1], got["ring sync"][1])
    assert np.array_equal(got["ring"][1], got["ring proj-launch"][1])
    # the projection inside the evaluation kernel: one launch per evaluation SLOT of the pipelined pattern fewer (at least one per evaluation)
    assert got["ring"][4] == got["ring proj-launch"][4] and got["ring proj-launch"][3] - got["ring"][3] >= got["ring"][4]
    assert np.linalg.norm(got["ring"][2] - got["generic"][2]) <= 1e-8 * max(1.0, np.linalg.norm(got["generic"][2]))


def test_bounded_second_generation_path_at_the_benchmark_size(qn, qo):
    """n = 4096 -- every workgroup two tiles and a sliver: s2_eval_kernel<true, .., BND> -- BFGSB + MoreThuenteB against the oracle and the
    generic path, the launch count of the pattern (one stored-direction launch per iteration on top of evaluation, accept-reduce, update
    tiles, update-reduce), and no host round trip per request."""
    n = 4096
    q, b, x0, lb, ub = _box(qo, n)
    iters = 20
    ref = qo.Solver(qo.BFGS, 1e-9, x0, qo.UPDATE_RANK2, nthreads=qo.max_threads())
    ref.set_bounds(lb, ub)
    ls_ref = _make_ls(qo, "mtb", n, lb, ub)
    ref.minimize(ls_ref, qo.QuadraticOracle(q, b, nthreads=qo.max_threads()), iters, 30, trace_cap=iters, trace_x=True)
    obj = qn.Quadratic(q, b)
    xs_by_mode = {}
    for mode in ("second generation", "generic"):
        s = qn.BFGSB.new(1e-9, x0, lb, ub)
        s.set_trace(iters, with_x=True)
        if mode == "generic":
            s.set_option("bounded_second_generation", 0)
        ls = _make_ls(qn, "mtb", n, lb, ub)
        try:
            s.minimize(ls, obj, iters, 30)
        except qn.MaxIterReached:
            pass
        tr, xs = s.trace()
        st = s.stats()
        assert bool(st["path"] & 16) == (mode != "generic")
        w = min(len(tr), len(ref.trace))
        assert w >= 15
        for k in range(w):
            assert tr[k]["n_evals"] == ref.trace[k]["n_evals"], (mode, k)
            assert abs(tr[k]["t"] - ref.trace[k]["t"]) <= 1e-8 * abs(ref.trace[k]["t"]), (mode, k)
            assert np.linalg.norm(xs[k] - ref.trace_x[k]) <= 1e-8 * max(1.0, np.linalg.norm(ref.trace_x[k])), (mode, k)
        assert abs(ls.t_max() - ls_ref.t_max) <= 1e-8 * max(1.0, abs(ls_ref.t_max))
        x = s.x()
        assert np.all(x >= lb - 1e-12) and np.all(x <= ub + 1e-12)
        xs_by_mode[mode] = x
        if mode == "second generation":
            evals = sum(r["n_evals"] for r in tr)
            assert st["host_syncs"] <= 4
            assert st["launches"] <= 2 * (len(tr) * 4 + evals) + 16  # (dir + vec + tiles + reduce per iteration, the evaluations; slack for unused slots)
    assert np.linalg.norm(xs_by_mode["second generation"] - xs_by_mode["generic"]) <= 1e-8 * max(1.0, np.linalg.norm(xs_by_mode["generic"]))


def test_lds_only_barriers_equal_full_barriers_bit_for_bit(qn, qo):
    """the generic path's control kernel orders its vector states with barriers that wait for LDS only (qn_lds_barrier: no wait for the wave's
    global stores) -- correct because an n-vector entry is touched by one thread per launch, an invariant nothing checks.  The same library built with
    -DQN_CTL_FULL_BARRIERS (`make -C csrc fullbar`, part of build()) runs the bounded generic path -- BFGSB + BackTrackingB and MoreThuenteB, projections
    and all -- in a child process; traces and iterates must be equal bit for bit."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "optimization-solvers_amd", "lib", "libqn_hip_fullbar.so")
    if not os.path.exists(lib):
        pytest.skip("libqn_hip_fullbar.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    code = r"""
import sys, os, hashlib
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import __graft_entry__ as ge
qn = ge.load_package()
from oracle import qn_oracle as qo
import test_gpu_bounded as T
h = hashlib.sha256()
for n, lsname in ((1024, "btb"), (1100, "mtb"), (640, "btb")):
    q, b, x0, lb, ub = T._box(qo, n)
    s = qn.BFGSB.new(1e-9, x0, lb, ub)
    s.set_option("bounded_second_generation", 0)
    s.set_trace(12, with_x=True)
    try:
        s.minimize(T._make_ls(qn, lsname, n, lb, ub), qn.Quadratic(q, b), 12, 30)
    except qn.MaxIterReached:
        pass
    tr, xs = s.trace()
    assert not (s.stats()["path"] & 16)
    h.update(repr(tr).encode()); h.update(np.ascontiguousarray(xs).tobytes()); h.update(np.ascontiguousarray(s.x()).tobytes())
print("DIGEST", h.hexdigest())
""" % (root, root)
    digests = []
    for env_lib in (None, lib):
        env = dict(os.environ)
        if env_lib:
            env["QN_HIP_LIB"] = env_lib
        else:
            env.pop("QN_HIP_LIB", None)
        cp = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert cp.returncode == 0, cp.stderr[-2000:]
        digests.append([l for l in cp.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert digests[0] == digests[1]
