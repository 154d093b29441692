"""Row f2 measurement: wall time of Newton.minimize on the n = 8192 synthetic quadratic (config 4) and its kernel split.
usage: newton_time.py [n] [lu]   -- `lu` forces the pivoted-LU path (qn_lu.hip.h) on the same SPD matrix"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as ge
qn = ge.load_package()
import problems as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
force_lu = len(sys.argv) > 2 and sys.argv[2] == "lu"
diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
for rep in range(3):
    s = qn.Newton(1e-8, x0)
    if force_lu:
        s.set_option("newton_pivoted_lu", 1)
    qn.default_context().synchronize()
    t0 = time.perf_counter()
    s.minimize(qn.MoreThuente(), obj, 10, 20)
    qn.default_context().synchronize()
    dt = time.perf_counter() - t0
    print(f"n={n} {'lu' if force_lu else 'cholesky'} rep={rep} iterations={s.k()} wall={dt*1e3:.2f} ms  ({dt*1e3/max(s.k(),1):.2f} ms per Newton iteration; "
          f"factorisation flops n^3/3 = {n**3/3:.3e})")
