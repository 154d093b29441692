"""Row f4, BackTrackingB on the second-generation path: bench.py's bounded leg (extra_bounded) by itself, with the projection inside the evaluation
kernel (default) and as a launch per trial (QN_S2_PROJ_FOLD=0 in the environment, or both in one run: this tool alternates the two through the solver option).
usage: python tools/bench_btb.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import __graft_entry__ as ge
qn = ge.load_package()
ctx = qn.default_context()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 4096
diag, b, x0 = bench.synth_inputs(n)
obj = qn.Quadratic.synthetic(n, bench.SEED, diag, b, ctx=ctx)
nt = qn.Newton(1e-12, x0, ctx=ctx)
try:
    nt.minimize(qn.MoreThuente(), obj, 3, 20)
except qn.MaxIterReached:
    pass
xs = np.array(nt.x(), dtype=np.float64)
del nt
lb = xs - 0.3 * np.abs(xs) - 0.05; ub = xs + 0.1
k4 = n // 4
lb[:k4] = xs[:k4] + 0.2; ub[:k4] = xs[:k4] + 1.0
def one(name):
    if name == "backtracking":
        s, ls = qn.BFGS(1e-10, x0, ctx=ctx), qn.BackTracking(1e-4, 0.5)
    else:
        s, ls = qn.BFGSB.new(1e-10, x0, lb, ub, ctx=ctx), qn.BackTrackingB.new(1e-4, 0.5, lb, ub)
        s.set_option("btb_project_in_eval", 1 if name == "btb in-eval" else 0)
    def run(k):
        try:
            s.minimize(ls, obj, k, 20)
        except qn.MaxIterReached:
            pass
    run(5); ctx.synchronize()
    t0 = time.perf_counter(); run(200); ctx.synchronize(); dt = time.perf_counter() - t0
    st = s.stats(); its = max(int(st["iterations"]), 1)
    return 1e6 * dt / its, st["launches"] / its, st["oracle_evals"] / its
for r in range(reps):
    res = {k: one(k) for k in ("backtracking", "btb in-eval", "btb proj-launch")}
    base = res["backtracking"][0]
    print(" | ".join("%s %.1f us/it (%.2f x; %.1f launches, %.2f evals per it)" % (k, v[0], v[0] / base, v[1], v[2]) for k, v in res.items()), flush=True)
