"""Diagnostic: in-kernel time stamps of the sym2 evaluation-tile kernel (build csrc/qn_hip.hip with -DQN_S2_STAMPS into
optimization-solvers_amd/lib/libqn_hip_stamps.so).  Prints, per stamped launch, the median over workgroups of each phase."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
qn = ge.load_package()
A = qn._abi
A.LIB_PATH = os.path.join(ROOT, "optimization-solvers_amd", "lib", os.environ.get("QN_STAMPS_LIB", "libqn_hip_stamps.so"))
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
# stamps (ns after the workgroup's entry, median over workgroups): 9 control block in LDS, 10 sums of the previous launch's table,
# 11 machine done, 1 wave 0 has requested its rows, 2 after the workgroup barrier (prologue end), 3 first pair: rows consumed and
# folded, 4 after the pair's barrier, 5 pair's slots stored, 15 end
order = [9, 10, 11, 1, 6, 7, 8, 2, 13, 14, 12, 3, 4, 5, 15]
if os.environ.get("QN_STAMPS_RING"):  # the mover / multiplier kernel (qn_sym2r.hip.h) stamps other things under the same numbers
    order = [11, 9, 6, 8, 2, 5, 7, 13, 1, 10, 14, 12, 3, 4, 15]
    label = {11: "machine + flag", 9: "m7 trial point staged", 6: "w1 parked", 8: "w4 parked", 2: "w7 4 rows parked", 5: "w7 12 rows parked", 7: "w7 first item parked, second requested", 13: "first item done: m7", 1: "m0", 10: "m4",
             14: "w7 second item done", 12: "w7 folded", 3: "w0 folded", 4: "exchange barrier", 15: "end"}
else:
  label = {12: "w7 pair folded", 13: "w7 item a done", 14: "w7 item b done", 6: "w7 requested", 7: "w7 parked", 8: "w7 at barrier", 9: "ctl", 10: "sums", 11: "machine", 1: "w0 at barrier", 2: "barrier", 3: "w0 pair folded", 4: "pair barrier", 5: "pair stored", 15: "end"}
for slot in range(64):
    t = st[slot]
    if t[0, 14] != 0 and t[0, 0] != 0 and t[0, 3] == 0:  # an accept-reduce launch that did work (32 workgroups; stamp 3 is the evaluation's)
        w = t[:32]
        print("slot %2d (accept-reduce): ctl %d, table %d, sums %d, machine %d, barrier %d, end %d; last workgroup's end %d" % (
            slot, *[int(np.median((w[:, k] - w[:, 0]) * 10)) for k in (9, 1, 10, 11, 2, 14)], int((w[:, 14].max() - w[:, 0].min()) * 10)))
        continue
    if t[0, 12] != 0 and t[0, 0] != 0 and t[0, 3] == 0 and t[0, 14] == 0 and t[0, 15] == 0:  # an update-reduce launch that did work (32 workgroups)
        w = t[:32]
        print("slot %2d (update-reduce): ctl %d, table %d, sums %d, machine %d, barrier %d, totals %d, end %d; last workgroup's end %d" % (
            slot, *[int(np.median((w[:, k] - w[:, 0]) * 10)) for k in (9, 1, 10, 11, 2, 13, 12)], int((w[:, 12].max() - w[:, 0].min()) * 10)))
        continue
    if t[0, 6] != 0 and t[0, 15] != 0 and t[0, 0] != 0 and t[0, 4] == 0:  # an update-tile launch (stamps 3 / 6: rows of item 1 / 2 consumed by wave 0, 12 / 13: by wave 7)
        wg = t[:, 15] > 0
        w = t[wg]
        end = (w[:, 15] - w[:, 0].min()) * 10
        print("slot %2d (update tiles): " % slot + ", ".join("%s %d" % (nm, int(np.median((w[:, k] - w[:, 0]) * 10))) for nm, k in (("ctl", 9), ("table", 1), ("sums", 10), ("machine", 11), ("barrier", 2), ("w0 item 1 rows", 3), ("w7 item 1 rows", 12), ("item 1 stored", 5), ("w0 item 2 rows", 6), ("w7 item 2 rows", 13), ("end", 15), ("tail: stores acknowledged", 7), ("tail: adds returned", 8), ("tail: done", 14)))
              + "; workgroup ends p50 %d p90 %d max %d" % (np.median(end), np.percentile(end, 90), end.max())
              + ("; tail (from the first entry): stores acknowledged p50 %d max %d, adds returned p50 %d max %d, done max %d" % tuple(
                  int(f((w[:, k] - w[:, 0].min()) * 10)) for k, f in ((7, np.median), (7, np.max), (8, np.median), (8, np.max), (14, np.max))) if w[0, 7] else ""))
        continue
    if t[0, 15] == 0 or t[0, 0] == 0 or t[0, 3] == 0:
        continue  # not an evaluation launch that did work
    wg = (t[:, 0] > 0) & (t[:, 15] > 0)
    t = t[wg]
    span = (t[:, 15].max() - t[:, 0].min()) * 10
    start_spread = int((t[:, 0].max() - t[:, 0].min()) * 10)
    parts = []
    for k in order:
        ok = (t[:, k] > 0) & (t[:, k] >= t[:, 0]) & (t[:, k] - t[:, 0] < 100000)
        parts.append("%s %d" % (label[k], int(np.median((t[ok, k] - t[ok, 0]) * 10))) if ok.any() else "%s -" % label[k])
    end = (t[:, 15] - t[:, 0].min()) * 10
    late = np.argsort(end)[-8:]
    print(f"slot {slot:2d}: span {span} ns, start spread {start_spread}; " + ", ".join(parts)
          + "; workgroup ends p50 %d p90 %d max %d, last: %s" % (np.median(end), np.percentile(end, 90), end.max(), " ".join("%d@%d" % (g, end[g]) for g in late)))

# which workgroups end late?  mean end (ns after the launch's first entry) by blockIdx.x % 8 -- the XCD under round-robin placement
if os.environ.get("QN_STAMPS_BY_XCD"):
    for name, sel in (("evaluation", lambda t: t[0, 15] != 0 and t[0, 0] != 0 and t[0, 3] != 0), ("update tiles", lambda t: t[0, 6] != 0 and t[0, 15] != 0 and t[0, 0] != 0 and t[0, 4] == 0)):
        acc = np.zeros(8); cnt = 0
        for slot in range(64):
            t = st[slot]
            if not sel(t):
                continue
            end = (t[:, 15] - t[:, 0].min()) * 10.0
            acc += np.array([end[x::8].mean() for x in range(8)]); cnt += 1
        if cnt:
            m = acc / cnt
            print("%s: mean workgroup end by blockIdx %% 8 over %d launches: %s  (spread %.0f ns)" % (name, cnt, " ".join("%.0f" % v for v in m), m.max() - m.min()))

