"""Diagnostic: in-kernel time stamps of the sym2 evaluation-tile kernel (build csrc/qn_hip.hip with -DQN_S2_STAMPS into
optimization-solvers_amd/lib/libqn_hip_stamps.so).  Prints, per stamped launch, the median over workgroups of each phase."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
qn = ge.load_package()
A = qn._abi
A.LIB_PATH = os.path.join(ROOT, "optimization-solvers_amd", "lib", "libqn_hip_stamps.so")
import problems as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
s = qn.BFGS(1e-10, x0)
L = A.lib()
L.qn_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
L.qn_debug_stamps(s.h, None, 0)
try:
    s.minimize(qn.MoreThuente(), obj, iters, 20)
except qn.MaxIterReached:
    pass
cnt = 64 * 256 * 16
buf = np.zeros(cnt, dtype=np.uint64)
L.qn_debug_stamps(s.h, buf.ctypes.data_as(C.c_void_p), cnt)
st = buf.reshape(64, 256, 16).astype(np.int64)
names = ["entry->loads issued", "prologue", "vec loads 1", "rows 1", "tail 1", "vec loads 2", "rows 2", "tail 2"]
for slot in range(64):
    t = st[slot]
    if t[0, 15] == 0 or t[0, 0] == 0:
        continue  # not an evaluation launch that did work
    wg = t[:, 0] > 0
    t = t[wg]
    span = (t[:, 15].max() - t[:, 0].min()) * 10
    d = []
    for k in range(1, 9):
        ok = (t[:, k] > 0) & (t[:, k - 1] > 0)
        d.append(int(np.median((t[ok, k] - t[ok, k - 1]) * 10)) if ok.any() else None)
    tot = int(np.median((t[:, 15] - t[:, 0]) * 10))
    start_spread = int((t[:, 0].max() - t[:, 0].min()) * 10)
    pro = [int(np.median((t[:, k] - t[:, 1]) * 10)) for k in (9, 10, 11, 2)]
    print(f"slot {slot:2d}: span {span} ns, start spread {start_spread}, median wg total {tot}; " + ", ".join(f"{nm} {v}" for nm, v in zip(names, d))
          + f"; prologue from its start: ctl in LDS {pro[0]}, sums {pro[1]}, machine done {pro[2]}, end {pro[3]}"
          + "; tail 1 from rows-done: folds %d, barrier %d, sums %d" % tuple(int(np.median((t[:, k] - t[:, 4]) * 10)) for k in (12, 13, 14)))
