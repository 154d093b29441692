"""Diagnostic: in-kernel time stamps of ctl_step (build csrc/qn_hip.hip with -DQN_CTL_STAMPS into optimization-solvers_amd/lib/libqn_hip_stamps.so)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
qn = ge.load_package()
A = qn._abi
A.LIB_PATH = os.path.join(ROOT, "optimization-solvers_amd", "lib", "libqn_hip_stamps.so")
import problems as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
s = qn.BFGS(1e-10, x0)
L = A.lib()
L.qn_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
L.qn_debug_stamps(s.h, None, 0)
try:
    s.minimize(qn.MoreThuente(), obj, 40, 20)
except qn.MaxIterReached:
    pass
cnt = 16 * 400
buf = np.zeros(cnt, dtype=np.uint64)
L.qn_debug_stamps(s.h, buf.ctypes.data_as(C.c_void_p), cnt)
ncalls = int(buf[0])
print("calls", ncalls)
for k in range(max(1, ncalls - 12), ncalls):
    rec = buf[16 + 16 * k: 32 + 16 * k].astype(np.int64)
    t0 = rec[0]
    rel = [(int(v - t0) * 10 if v else None) for v in rec[:15]]
    print(k, "expect", int(rec[15]), "ns:", rel)
