#!/usr/bin/env python3
"""Is the update kernel's slow mode (28-30 us instead of 23.5 at n = 4096, about one process in five) a property of the
PROCESS or of the ALLOCATION of H?  One process, several solvers one after the other -- each with an inverse Hessian of its own,
the earlier ones kept alive so that the allocator cannot hand the same pages out again --, the same 60 iterations on each, the
update kernel's HIP-event average per solver.  usage: python tools/slow_mode_probe.py [solvers] [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import bench
qn = ge.load_package()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = qn.Context(0)
diag, b, x0 = bench.synth_inputs(n)
obj = qn.Quadratic.synthetic(n, bench.SEED, diag, b, ctx=ctx)
ls = qn.MoreThuente()
keep, out = [], []
for k in range(K):
    s = qn.BFGS(1e-10, x0, ctx=ctx)
    keep.append(s)
    bench.run_iterations(qn, s, ls, obj, x0, 10)
    s.reset(x0)
    s.set_profiling(True)
    p0 = s.stats()
    bench.run_iterations(qn, s, ls, obj, x0, 60)
    p1 = s.stats()
    s.set_profiling(False)
    h = (p1["t_hpass_ms"] - p0["t_hpass_ms"]) / max(1, p1["n_hpass_timed"] - p0["n_hpass_timed"]) * 1e3
    e = (p1["t_eval_ms"] - p0["t_eval_ms"]) / max(1, p1["n_eval_timed"] - p0["n_eval_timed"]) * 1e3
    out.append((h, e))
print("pid %d: update kernel us per solver: %s | evaluation: %s" % (os.getpid(), " ".join("%.1f" % h for h, _ in out), " ".join("%.1f" % e for _, e in out)))
