"""GPU parity tests that drive More-Thuente through cases 2, 3 and 4, the sticky modified-updating switch and the evaluation at
`x + tu d` with tu = +inf (morethuente.rs:212-215, 243-293, 64-91) on EVERY path of the HIP library -- the symmetric-storage
tiles and the fused row kernels (the bench path), with and without the deferred update step, pipelined and synchronous, and the
generic path with the memo on and off -- and assert that the cases were really taken (on the oracle's trace and on the GPU's).

Workloads and why they reach those branches: tests/mt_workloads.py.  Tolerances: the parity sweep's (test_gpu_parity.py)."""
import numpy as np
import pytest

import mt_workloads as W
from test_gpu_parity import F_TOL, T_TOL, X_TOL

pytestmark = pytest.mark.gpu

_REF = {}


def _ref(qo, n, name, method):
    key = (n, name, method)
    if key not in _REF:
        s, st, _ = W.run_oracle(qo, n, name, method, qo.UPDATE_RANK2, threads=min(qo.max_threads(), 32))
        _REF[key] = (s.trace, s.trace_x, st)
    return _REF[key]


def _run_gpu(qn, n, name, method, tiling=None, sync=None, memoize=None, tiling2=None):
    w = W.WORKLOADS[name]
    diag, b, x0 = W.inputs(n, name)
    obj = qn.Quadratic.synthetic(n, W.P.SEED, diag, b)  # generated on the device, bit-identical to the oracle's matrix
    s = (qn.BFGS if method == "bfgs" else qn.DFP)(1e-10, x0)
    if w["h0"] is not None:
        s.set_approx_inv_hessian(w["h0"] * np.eye(n))
    s.set_trace(w["iters"], with_x=True)
    if tiling:
        s.set_tiling(*tiling)
    if tiling2:
        s.set_tiling(*tiling2)
    if sync is not None:
        s.set_sync_mode(sync)
    if memoize is not None:
        s.memoize = memoize
    ls = qn.MoreThuente()
    if w["t_max"] is not None:
        ls = ls.with_t_max(w["t_max"])
    st = 0
    try:
        s.minimize(ls, obj, w["iters"], 20)
    except qn.MaxIterReached:
        st = 1
    tr, xs = s.trace()
    return s, st, tr, xs


def _check(tr, xs, st, ref, label, name=None):
    rtr, rxs, rst = ref
    assert st == rst and len(tr) == len(rtr), (label, st, rst, len(tr), len(rtr))
    for k, (a, b) in enumerate(zip(tr, rtr)):
        assert a["ls_cases"] == b["ls_cases"], (label, k, oct(a["ls_cases"]), oct(b["ls_cases"]))
        assert (a["n_evals"], a["ls_iters"], a["updated"]) == (b["n_evals"], b["ls_iters"], b["updated"]), (label, k, a, b)
        t_tol = W.t_tol(name, b["gnorm"], rtr[0]["gnorm"], T_TOL) if name else T_TOL
        assert abs(a["t"] - b["t"]) <= t_tol * abs(b["t"]), (label, k, a["t"], b["t"], t_tol)
        assert abs(a["f"] - b["f"]) <= F_TOL * max(1.0, abs(b["f"])), (label, k, a["f"], b["f"])
        assert np.linalg.norm(xs[k] - rxs[k]) <= X_TOL * max(1.0, np.linalg.norm(rxs[k])), (label, k)


PATHS = {
    # name: (solver knobs, path flags the run must report: (fused, sym, sym_generic, pipelined))
    "sym": (dict(), (1, 1, 0, 1)),
    "sym_sync": (dict(sync=1), (1, 1, 0, 0)),
    "sym_v1": (dict(tiling=(-4, 0)), (1, 1, 0, 1)),  # first-generation tile kernels (separate control launches, deferred update step)
    "sym_v1_no_defer": (dict(tiling=(-2, 0), tiling2=(-4, 0)), (1, 1, 0, 1)),
    "rows": (dict(tiling=(-3, 0)), (1, 0, 0, 1)),
    "rows_sync": (dict(tiling=(-3, 0), sync=1), (1, 0, 0, 0)),
    "generic": (dict(tiling=(-1, 0)), (0, 0, 1, 1)),
    "generic_no_memo": (dict(tiling=(-1, 0), memoize=0), (0, 0, 1, 0)),
}


@pytest.mark.parametrize("n,path,methods", [
    (1024, "sym", ("bfgs", "dfp")), (1024, "sym_sync", ("bfgs",)), (1024, "sym_v1", ("bfgs",)), (1024, "sym_v1_no_defer", ("bfgs",)),
    (1024, "rows", ("bfgs", "dfp")), (1024, "rows_sync", ("bfgs",)),
    (1024, "generic", ("bfgs", "dfp")), (1024, "generic_no_memo", ("bfgs",)),
    (4096, "sym", ("bfgs",)), (4096, "rows", ("bfgs",)),
])
def test_morethuente_cases_2_3_4_and_modified_updating(qn, qo, n, path, methods):
    knobs, want = PATHS[path]
    tot = {1: 0, 2: 0, 3: 0, 4: 0}
    mods = 0
    for method in methods:
        for name, w in W.WORKLOADS.items():
            ref = _ref(qo, n, name, method)
            cnt_ref, mod_ref = W.count_cases(ref[0])
            for digit in w["expect"]:  # the oracle's trace takes the branch this workload exists for
                assert cnt_ref[digit] >= 1, (name, method, cnt_ref)
            s, st, tr, xs = _run_gpu(qn, n, name, method, **knobs)
            flags = s.stats()["path"]
            got = (flags & 1, (flags >> 1) & 1, (flags >> 2) & 1, (flags >> 3) & 1)
            assert got == want, (path, name, got, want)
            assert bool(flags & 16) == (path in ("sym", "sym_sync")), (path, flags)  # second-generation kernels (qn_sym2.hip.h)
            _check(tr, xs, st, ref, (n, path, method, name), name)
            cnt, mod = W.count_cases(tr)
            assert (cnt, mod) == (cnt_ref, mod_ref)
            for d in tot:
                tot[d] += cnt[d]
            mods += mod
    # the point of this test: cases 2, 3 and 4 each at least three times and the switch at least once ON THIS PATH
    assert tot[2] >= 3 and tot[3] >= 3 and tot[4] >= 3 and mods >= 1, (path, tot, mods)


@pytest.mark.parametrize("path", ["rows", "generic"])
def test_cases_on_a_ragged_dimension(qn, qo, path):
    """n = 1100 is no multiple of 128: fused row kernels / h_pass_kernel (no symmetric tiles), ragged last column chunk"""
    n = 1100
    knobs = dict(tiling=(-1, 0)) if path == "generic" else {}
    for name in ("case2_mod", "case3_tmax", "case4_inf", "case4_tmax2"):
        ref = _ref(qo, n, name, "bfgs")
        s, st, tr, xs = _run_gpu(qn, n, name, "bfgs", **knobs)
        flags = s.stats()["path"]
        assert (flags & 1) == (0 if path == "generic" else 1) and (flags & 6) == 0
        _check(tr, xs, st, ref, (n, path, name), name)
        for digit in W.WORKLOADS[name]["expect"]:
            assert W.count_cases(tr)[0][digit] >= 1


@pytest.mark.parametrize("memoize", [0, 1])
def test_infinite_tu_trial_point_reaches_the_oracle_and_the_memo_survives_it(qn, qo, memoize):
    """Case 4 with t_max = +inf on the generic path through a HOST closure: the closure must be handed the non-finite point
    x + inf d (morethuente.rs:276) exactly once, between the tl = 0 re-evaluation and the final t = 0 evaluations, and the
    call sequence must be the reference's point for point (memoize = 0) or its distinct points (memoize = 1)."""
    n = 64
    diag, b, x0 = W.inputs(n, "case4_inf")
    q = qo.synth_rows(n, 0, n, W.P.SEED, diag)
    seen_ref, seen = [], []

    def fn_ref(x):
        seen_ref.append(x.copy())
        with np.errstate(invalid="ignore"):
            return 0.5 * x @ (q @ x) - b @ x, q @ x - b

    def fn(x):
        seen.append(x.copy())
        with np.errstate(invalid="ignore"):
            return 0.5 * x @ (q @ x) - b @ x, q @ x - b

    ref = qo.Solver(qo.BFGS, 1e-10, x0, qo.UPDATE_AS_WRITTEN)
    st_ref = ref.minimize(qo.morethuente(), fn_ref, 8, 20, trace_cap=8, trace_x=True)
    s = qn.BFGS(1e-10, x0)
    s.memoize = memoize
    s.set_trace(8, with_x=True)
    s.minimize(qn.MoreThuente(), fn, 8, 20)  # Ok(()): would raise otherwise
    assert st_ref == 0
    tr, xs = s.trace()
    assert [r["ls_cases"] for r in tr] == [r["ls_cases"] for r in ref.trace]
    assert [r["n_evals"] for r in tr] == [r["n_evals"] for r in ref.trace]
    assert W.case_digits(tr[-1]["ls_cases"]) == [4] and tr[-1]["t"] == 0.0
    bad = [i for i, p in enumerate(seen) if not np.all(np.isfinite(p))]
    bad_ref = [i for i, p in enumerate(seen_ref) if not np.all(np.isfinite(p))]
    assert len(bad) == 1 and len(bad_ref) == 1
    if memoize == 0:  # the reference's call sequence, point for point
        assert len(seen) == len(seen_ref) and bad == bad_ref
        for p, r in zip(seen, seen_ref):
            fin = np.isfinite(r)
            assert np.array_equal(np.isfinite(p), fin) and np.linalg.norm(p[fin] - r[fin]) <= 1e-9 * max(1.0, np.linalg.norm(r[fin]))
    else:
        assert len(seen) < len(seen_ref)
    assert np.linalg.norm(xs[-1] - ref.trace_x[-1]) <= X_TOL * max(1.0, np.linalg.norm(ref.trace_x[-1]))
