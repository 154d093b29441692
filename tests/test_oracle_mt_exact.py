"""CPU test of the 60-digit fixture tests/golden/mt_exact_n1024.json (generator: tests/golden/make_mt_exact.py): the generator
reproduces its own first records, its exact integer mat-vec / implicit inverse Hessian agree with the dense 60-digit evaluation
of tests/test_oracle_mpmath.py at n = 64, and the C oracle at n = 1024 takes the same More-Thuente cases with the same evaluation
counts and is within the workloads' stated step tolerances of the truth (the GPU tests bound the HIP path against the same file)."""
import json
import os
import sys

import mpmath as mp
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_mt_exact as M  # noqa: E402
import mt_workloads as W  # noqa: E402
import test_oracle_mpmath as T  # noqa: E402

mp.mp.dps = 60


def _fixture():
    return json.load(open(os.path.join(HERE, "golden", "mt_exact_n1024.json")))["cases"]


def _undd(p):
    return mp.mpf(float.fromhex(p[0])) + mp.mpf(float.fromhex(p[1]))


@pytest.mark.parametrize("name", ["case2_mod", "case3_inf"])
@pytest.mark.parametrize("method", ["bfgs", "dfp"])
def test_exact_evaluation_equals_dense_high_precision_at_n64(qo, name, method):
    n = 64
    w = W.WORKLOADS[name]
    diag, b, x0 = W.inputs(n, name)
    q = qo.synth_rows(n, 0, n, W.P.SEED, diag)
    a = M.run_exact(method, q, b, x0, w["iters"], w["h0"], mp.inf)
    d = T._run_mp(method, "mt", q, b, x0, w["iters"], h0=w["h0"])
    assert len(a) == len(d) >= 2
    for u, v in zip(a, d):
        assert u["digits"] == v["digits"] and u["n_evals"] == v["n_evals"]
        assert abs(u["t"] - v["t"]) <= mp.mpf("1e-50") * abs(v["t"])
        assert max(abs(p - r) for p, r in zip(u["x"], v["x"])) <= mp.mpf("1e-50")


def test_generator_reproduces_the_committed_fixture(qo):
    cases = M.generate(1024, names=("case3_inf",), methods=("bfgs",))
    want = [c for c in _fixture() if (c["workload"], c["method"]) == ("case3_inf", "bfgs")][0]
    assert json.loads(json.dumps(cases[0])) == want


@pytest.mark.parametrize("mode", ["as_written", "rank2"])
def test_oracle_against_the_60_digit_trace_n1024(qo, mode):
    for c in _fixture():
        name, method = c["workload"], c["method"]
        s, _, _ = W.run_oracle(qo, 1024, name, method, mode=qo.UPDATE_AS_WRITTEN if mode == "as_written" else qo.UPDATE_RANK2,
                               threads=min(qo.max_threads(), 8))
        g0 = s.trace[0]["gnorm"]
        for k, e in enumerate(c["records"]):
            r = s.trace[k]
            assert W.case_digits(r["ls_cases"]) == e["digits"] and r["n_evals"] == e["n_evals"], (name, method, k)
            te = _undd(e["t_dd"])
            assert abs(mp.mpf(r["t"]) - te) <= W.t_tol(name, r["gnorm"], g0, 1e-9) * abs(te), (name, method, k)
            xe = [_undd(v) for v in e["x_dd"]]
            dx = mp.sqrt(mp.fsum((mp.mpf(float(a)) - b) ** 2 for a, b in zip(s.trace_x[k], xe)))
            assert dx <= 1e-9 * max(1, mp.sqrt(mp.fsum(v * v for v in xe))), (name, method, k)
