"""Diagnostic: in-kernel time stamps of ctl_step on a BOUNDED run of the GENERIC path (BFGSB + MoreThuenteB, the box of tools/bench_bounded.py); needs
optimization-solvers_amd/lib/libqn_hip_ctlstamps.so built with -DQN_CTL_STAMPS.  usage: ctl_stamps_bounded.py [n]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
qn = ge.load_package()
A = qn._abi
A.LIB_PATH = os.path.join(ROOT, "optimization-solvers_amd", "lib", "libqn_hip_ctlstamps.so")
import problems as P
from oracle import qn_oracle as qo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
q = qo.synth_rows(n, 0, n, P.SEED, diag, nthreads=qo.max_threads())
xs = np.linalg.solve(q, b)
lb = xs - 0.3 * np.abs(xs) - 0.05
ub = xs + 0.1
k4 = max(1, n // 4)
lb[:k4] = xs[:k4] + 0.2
ub[:k4] = xs[:k4] + 1.0
s = qn.BFGSB.new(1e-10, x0, lb, ub)
s.set_option("bounded_second_generation", 0)  # the generic path (bounded runs of this shape default to the second-generation one since round 5)
ls = qn.MoreThuenteB.new(n).with_lower_bound(lb).with_upper_bound(ub)
L = A.lib()
L.qn_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for k in (10, 30):
    try:
        s.minimize(ls, obj, k, 20)
    except qn.MaxIterReached:
        pass
    if k == 10:
        L.qn_debug_stamps(s.h, None, 0)
cnt = 16 * 400
buf = np.zeros(cnt, dtype=np.uint64)
L.qn_debug_stamps(s.h, buf.ctypes.data_as(C.c_void_p), cnt)
ncalls = int(buf[0])
print("calls", ncalls, "path", s.stats()["path"])
for k in range(max(1, ncalls - 14), ncalls):
    rec = buf[16 + 16 * k: 32 + 16 * k].astype(np.int64)
    t0 = rec[0]
    rel = [(int(v - t0) * 10 if v else None) for v in rec[:15]]
    print(k, "expect", int(rec[15]), "ns:", rel)
