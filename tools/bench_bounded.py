"""Row f4 measurement: BFGSB + MoreThuenteB (box bounds, some of them active at the solution) against BFGS + MoreThuente on the same
n-dim synthetic quadratic, one GPU: iterations/s of each and their ratio (VERDICT r4 item 7 asks for <= 1.3).  usage: bench_bounded.py [n] [iters]"""
import json, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as ge
qn = ge.load_package()
import problems as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
from oracle import qn_oracle as qo
q = qo.synth_rows(n, 0, n, P.SEED, diag, nthreads=qo.max_threads())
xs = np.linalg.solve(q, b)  # the box of tests/test_gpu_bounded.py: a quarter of the bounds active at the constrained optimum
lb = xs - 0.3 * np.abs(xs) - 0.05
ub = xs + 0.1
k4 = max(1, n // 4)
lb[:k4] = xs[:k4] + 0.2
ub[:k4] = xs[:k4] + 1.0
x0b = x0
out = {}
for name, mk, ls in (("unbounded", lambda: qn.BFGS(1e-10, x0b), qn.MoreThuente), ("bounded", lambda: qn.BFGSB.new(1e-10, x0b, lb, ub), lambda: qn.MoreThuenteB.new(n).with_lower_bound(lb).with_upper_bound(ub))):
    s = mk()
    l = ls()
    def run(k):
        try:
            s.minimize(l, obj, k, 20)
        except qn.MaxIterReached:
            pass
        except Exception as e:  # noqa: BLE001
            print(name, "ended with", type(e).__name__, "after", s.k(), "iterations", file=sys.stderr)
    run(5)
    qn.default_context().synchronize()
    t0 = time.perf_counter(); run(iters); qn.default_context().synchronize(); dt = time.perf_counter() - t0
    st = s.stats()
    out[name] = {"it_per_s": st["iterations"] / dt, "us_per_iteration": 1e6 * dt / max(st["iterations"], 1), "iterations": st["iterations"], "path": st["path"],
                 "launches_per_iteration": st["launches"] / max(st["iterations"], 1), "host_syncs": st["host_syncs"]}
out["ratio_bounded_over_unbounded_time"] = out["bounded"]["us_per_iteration"] / out["unbounded"]["us_per_iteration"]
print(json.dumps(out))
