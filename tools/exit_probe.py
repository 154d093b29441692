"""Leaves live handles (a Newton solver that ran the pivoted-LU path, a device objective, a traceback that references both) for
interpreter exit to clean up; the process must end with the exit code it was given, not with a crash in a destructor."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

qn = ge.load_package()
n = 200
rng = np.random.default_rng(0)
m = rng.standard_normal((n, n))
q = m + m.T + np.diag(np.where(np.arange(n) % 7 == 0, -30.0, 30.0))
b = rng.standard_normal(n)
s = qn.Newton(1e-8, rng.standard_normal(n))
obj = qn.Quadratic(q, b)
try:
    s.minimize(qn.BackTracking(1e-4, 0.5), obj, 1, 30)
except qn.MaxIterReached:
    pass
print("x ok", np.linalg.norm(s.x() - np.linalg.solve(q, b)) <= 1e-8 * np.linalg.norm(np.linalg.solve(q, b)))
keep = []
try:
    raise RuntimeError("keep a traceback that references the handles")
except RuntimeError as e:
    keep.append((e, s, obj))
sys.last_value = keep[0][0]
print("exiting with live handles")
sys.exit(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
