import sys, time, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as ge
qn = ge.load_package()
from oracle import qn_oracle as qo
def problem(m, n, seed=3, scale=1.0):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((m, n)) * scale / np.sqrt(n), rng.standard_normal(m), rng.standard_normal(n)
for (m, n, method, lsn) in [(600, 1024, "dfp", "mt"), (600, 1024, "bfgs", "mt"), (1500, 1280, "dfp", "bt"), (2000, 2048, "dfp", "mt")]:
    a, c, x0 = problem(m, n, scale=3.0)
    mu, iters = 0.1, 25
    ref = qo.Solver(qo.DFP if method == "dfp" else qo.BFGS, 1e-10, x0, qo.UPDATE_RANK2, nthreads=8)
    o = qo.LogSumExpOracle(a, c, mu, nthreads=8)
    ls_o = qo.morethuente() if lsn == "mt" else qo.backtracking(1e-4, 0.5)
    ref.minimize(ls_o, o, iters, 20, trace_cap=iters, trace_x=True)
    obj = qn.LogSumExp(a, c, mu)
    outs = []
    for sync in (0, 1):
        s = (qn.DFP if method == "dfp" else qn.BFGS)(1e-10, x0)
        s.set_trace(iters, with_x=True)
        s.set_sync_mode(sync)
        ls = qn.MoreThuente() if lsn == "mt" else qn.BackTracking(1e-4, 0.5)
        try:
            s.minimize(ls, obj, iters, 20)
        except qn.MaxIterReached:
            pass
        tr, xs = s.trace()
        st = s.stats()
        outs.append((tr, xs))
        ok = len(tr) == len(ref.trace)
        worst = 0.0
        for k in range(min(len(tr), len(ref.trace))):
            a_, b_ = tr[k], ref.trace[k]
            ok = ok and a_["ls_cases"] == b_["ls_cases"] and a_["n_evals"] == b_["n_evals"]
            worst = max(worst, abs(a_["t"] - b_["t"]) / abs(b_["t"]), np.linalg.norm(xs[k] - ref.trace_x[k]) / max(1, np.linalg.norm(ref.trace_x[k])))
        print(m, n, method, lsn, "sync" if sync else "pipelined", "path", st["path"], "iters", len(tr), "decisions equal", ok, "worst rel", worst, "launches", st["launches"], "evals", st["oracle_evals"])
    print("  pipelined == sync bitwise:", outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1]))
