#!/usr/bin/env python3
"""What a qn_minimize call costs beyond its iterations (n = 4096, the driver's protocol): wall time of warm calls of K iterations for
several K, a straight-line fit -- slope = one iteration, intercept = the call.  usage: python tools/call_cost.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import bench
qn = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = qn.Context(0)
diag, b, x0 = bench.synth_inputs(n)
obj = qn.Quadratic.synthetic(n, bench.SEED, diag, b, ctx=ctx)
s = qn.BFGS(1e-10, x0, ctx=ctx)
ls = qn.MoreThuente()
Ks = [1, 2, 5, 10, 20, 40, 80]
res = {}
for K in Ks:
    ts = []
    for rep in range(12):
        s.reset(x0)
        bench.run_iterations(qn, s, ls, obj, x0, 5)
        ctx.synchronize()
        t0 = time.perf_counter()
        bench.run_iterations(qn, s, ls, obj, x0, K)
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    res[K] = np.median(ts) * 1e6
    print(f"K = {K:3d}: {res[K]:9.1f} us  ({res[K] / K:7.2f} us per iteration)")
A = np.vstack([Ks, np.ones(len(Ks))]).T
slope, icpt = np.linalg.lstsq(A, np.array([res[k] for k in Ks]), rcond=None)[0]
print(f"fit: {slope:.2f} us per iteration + {icpt:.1f} us per call")
