"""CPU tests: the More-Thuente workloads of tests/mt_workloads.py really do reach cases 2, 3 and 4, throw the
modified-updating switch and evaluate the oracle at `x + inf d` (morethuente.rs:212-215, 243-293) -- asserted on the oracle's
own trace so the GPU parity tests that use them cannot silently degrade to case 1 -- and the two update formulations of the
oracle (as written, O(n^3) / rank-2, O(n^2)) agree on them to the stated tolerance."""
import numpy as np
import pytest

import mt_workloads as W


@pytest.mark.parametrize("n", [64, 256])
@pytest.mark.parametrize("method", ["bfgs", "dfp"])
@pytest.mark.parametrize("name", list(W.WORKLOADS))
def test_workloads_reach_their_cases_and_both_update_forms_agree(qo, n, method, name):
    w = W.WORKLOADS[name]
    a, st_a, q = W.run_oracle(qo, n, name, method, qo.UPDATE_AS_WRITTEN)
    r, st_r, _ = W.run_oracle(qo, n, name, method, qo.UPDATE_RANK2, memo_q=q)
    cnt, mod = W.count_cases(a.trace)
    for digit, least in w["expect"].items():
        assert cnt[digit] >= min(least, 2 if (method == "dfp" and digit == 3) else least), (name, cnt)
    assert mod >= w["expect_mod"], (name, mod)
    if w["expect_mod"]:  # the switch is thrown exactly in the line searches that take case 2 here
        assert all(bool(rec["ls_cases"] & W.MOD_BIT) == (W.case_digits(rec["ls_cases"])[:1] == [2]) for rec in a.trace)
    assert st_a == st_r and len(a.trace) == len(r.trace)
    for k, (x, y) in enumerate(zip(a.trace, r.trace)):
        assert (x["ls_cases"], x["n_evals"], x["ls_iters"]) == (y["ls_cases"], y["n_evals"], y["ls_iters"]), (name, k)
        assert abs(x["t"] - y["t"]) <= 1e-9 * abs(y["t"]), (name, k, x["t"], y["t"])
        assert np.linalg.norm(a.trace_x[k] - r.trace_x[k]) <= 1e-9 * max(1.0, np.linalg.norm(r.trace_x[k])), (name, k)


def test_case4_at_infinite_tu_ends_the_run_with_a_zero_step(qo):
    """morethuente.rs:276 with tu = +inf: the oracle sees a non-finite point, f = NaN, cubic_minimizer returns NaN, the clamp
    `t.max(t_min).min(t_max)` (:290) drops the NaN and leaves t = t_min = 0; update_interval reports convergence (g_t*(tl - t)
    == 0), the next inner iteration returns t = 0, so s = 0 and the next loop top exits through `next iterate too close`."""
    n = 64
    seen = []
    diag, b, x0 = W.inputs(n, "case4_inf")
    q = qo.synth_rows(n, 0, n, W.P.SEED, diag)

    def fn(x):
        seen.append(x.copy())
        with np.errstate(invalid="ignore"):  # the line search hands over x + inf d once: inf - inf inside the products
            return 0.5 * x @ (q @ x) - b @ x, q @ x - b

    s = qo.Solver(qo.BFGS, 1e-10, x0, qo.UPDATE_AS_WRITTEN)
    st = s.minimize(qo.morethuente(), fn, 8, 20, trace_cap=8, trace_x=True)
    assert st == 0  # Ok(()): bfgs.rs:64-68
    last = s.trace[-1]
    assert W.case_digits(last["ls_cases"]) == [4] and last["t"] == 0.0 and last["n_evals"] == 6 and last["updated"] == 0
    nonfinite = [p for p in seen if not np.all(np.isfinite(p))]
    assert len(nonfinite) == 1 and np.all(np.isinf(nonfinite[0]) | np.isnan(nonfinite[0]))
    assert s.s_norm == 0.0 and np.array_equal(s.trace_x[-1], s.trace_x[-2])


def test_workload_counts_at_the_gpu_test_size(qo):
    """n = 1024 (the smallest symmetric-storage size), rank-2 mode: the totals the GPU parity test asserts"""
    tot = {1: 0, 2: 0, 3: 0, 4: 0}
    mods = 0
    for name in W.WORKLOADS:
        s, _, _ = W.run_oracle(qo, 1024, name, "bfgs", qo.UPDATE_RANK2, threads=min(qo.max_threads(), 8))
        cnt, mod = W.count_cases(s.trace)
        for d in tot:
            tot[d] += cnt[d]
        mods += mod
    assert tot[2] >= 3 and tot[3] >= 3 and tot[4] >= 3 and mods >= 1, (tot, mods)
