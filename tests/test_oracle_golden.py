"""CPU test: the oracle reproduces the committed golden traces bit for bit (regression pin; the vectors are
build-generated, see tests/golden/make_golden.py)."""
import json
import os

import numpy as np

import mt_workloads as W
import problems as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traces.json")


def unhx(v):
    return np.array([float.fromhex(x) for x in v])


def load_cases():
    with open(GOLD) as fh:
        return json.load(fh)["cases"]


def test_golden_inputs_regenerate(qo):
    for c in load_cases():
        if c.get("workload"):
            diag, b, x0 = W.inputs(c["n"], c["workload"], c["seed"])
        else:
            q, b, x0, diag = P.synth_problem(qo, c["n"], c["kappa"], c["seed"])
        assert np.array_equal(b, unhx(c["b"])) and np.array_equal(x0, unhx(c["x0"])) and np.array_equal(diag, unhx(c["diag"]))


def test_oracle_reproduces_golden_traces(qo):
    for c in load_cases():
        diag, b, x0 = unhx(c["diag"]), unhx(c["b"]), unhx(c["x0"])
        q = qo.synth_rows(c["n"], 0, c["n"], c["seed"], diag)
        ls = qo.morethuente(**({"t_max": c["t_max"]} if c.get("t_max") is not None else {})) if c["ls"] == "mt" else qo.backtracking(1e-4, 0.5)
        s = qo.Solver(qo.BFGS if c["method"] == "bfgs" else qo.DFP, c["tol"], x0, qo.UPDATE_AS_WRITTEN)
        if c.get("h0") is not None:
            s.set_inv_hessian(c["h0"] * np.eye(c["n"]))
        o = qo.QuadraticOracle(q, b)
        st = s.minimize(ls, o, c["max_iter"], c["max_iter_ls"], trace_cap=c["max_iter"], trace_x=True)
        assert st == c["status"] and s.k == c["k"] and o.calls == c["oracle_calls"]
        assert np.array_equal(np.array([r["t"] for r in s.trace]), unhx(c["t"]))
        assert np.array_equal(np.array([r["f"] for r in s.trace]), unhx(c["f"]))
        assert [r["ls_cases"] for r in s.trace] == c["ls_cases"]
        assert np.array_equal(s.x, unhx(c["x_final"]))


def test_golden_traces_cover_every_morethuente_case():
    """the fixtures hold cases 2, 3 and 4 and the modified-updating switch, not only case 1 (morethuente.rs:212-215, 243-287)"""
    tot, mods = {1: 0, 2: 0, 3: 0, 4: 0}, 0
    for c in load_cases():
        cnt, mod = W.count_cases([dict(ls_cases=v) for v in c["ls_cases"]])
        for d in tot:
            tot[d] += cnt[d]
        mods += mod
    assert min(tot[2], tot[3], tot[4]) >= 3 and mods >= 1, (tot, mods)
