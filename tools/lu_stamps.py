"""Diagnostic: in-kernel time stamps of role A of the LU panel (lu_panel_persist_kernel, or lu_panel_step_kernel with QN_LU_PERSIST=0):
build csrc/qn_hip.hip with -DQN_LU_STAMPS [-DQN_LU_STAMP_P0=<first column of the panel to stamp, default 640>] into
optimization-solvers_amd/lib/libqn_hip_lustamps.so; prints, per sub-panel of that panel, ns since the sub-panel's first stamp."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
qn = ge.load_package()
A = qn._abi
A.LIB_PATH = os.environ.get("QN_LU_STAMPS_LIB") or os.path.join(ROOT, "optimization-solvers_amd", "lib", "libqn_hip_lustamps.so")
import problems as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
s = qn.Newton(1e-8, x0)
s.set_option("newton_pivoted_lu", 1)
s.minimize(qn.MoreThuente(), obj, 1, 20) if False else None
try:
    s.minimize(qn.MoreThuente(), obj, 1, 20)
except qn.MaxIterReached:
    pass
buf = np.zeros(64 * 16, dtype=np.uint64)
L = A.lib()
L.qn_debug_lu_stamps.argtypes = [C.c_void_p]
L.qn_debug_lu_stamps(buf.ctypes.data_as(C.c_void_p))
st = buf.reshape(64, 16).astype(np.int64)
names = ["entry", "flag read / waited", "columns loaded/updated", "search done", "after barrier", "rows exchanged", "step 0 done", "step 3 done", "stores issued", "stores complete",
         "record stored", "records seen", "after barrier 2"]  # (the last three: role A split over workgroups, qn_lu_split.hip.h, part 0)
for s_ in range(17):
    t = st[s_]
    if t[0] == 0:
        continue
    print("launch s=%2d:" % s_, ", ".join("%s %d" % (names[k], (t[k] - t[0]) * 10) for k in (1, 2, 3, 4, 10, 11, 12, 5, 6, 7, 8, 9) if t[k] > 0))
