"""Soak of the one-launch LU kernels (panel, sweeps) and the look-ahead: the same non-symmetric Newton step many times over through the
default path and through round 3's launches (set_option("lu_lookahead", 0), ("lu_one_launch_panel", 0): one launch per sub-panel / per block) -- every
repetition must give the same bits.  A stale read between workgroups (the panel buffer and the solution vectors cross workgroups as
write-through stores and sc1 loads, no cache flush) would show up here as a mismatch.
usage: python tools/lu_soak.py [n] [reps] [synth]   (synth: the device-resident SPD benchmark matrix through the forced LU path -- n = 8192 without a host matrix)"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as ge
qn = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
synth = len(sys.argv) > 3 and sys.argv[3] == "synth"
if synth:
    import problems as P
    diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
    fn = qn.Quadratic.synthetic(n, P.SEED, diag, b)
else:
    rng = np.random.default_rng(3)
    a = rng.standard_normal((n, n)) / np.sqrt(n) + 2.0 * np.diag(rng.choice([-1.0, 1.0], n))  # indefinite, non-symmetric, well conditioned
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    def fn(x):
        return qn.FuncEvalMultivariate(float(0.5 * x @ (a @ x) - b @ x), 0.5 * (a + a.T) @ x - b).with_hessian(a)
def step(flags):
    s = qn.Newton(1e-12, x0)
    if synth:
        s.set_option("newton_pivoted_lu", 1)
    for f in flags:
        s.set_option(f, 0)
    s.set_trace(1, with_x=True)
    try:
        s.minimize(qn.BackTracking(1e-4, 0.5), fn, 1, 5)
    except qn.MaxIterReached:
        pass
    global expired
    expired += s.stats()["newton_lu_sync_timeouts"]
    return s.trace()[1][0].copy()
expired = 0
ref = step(("lu_lookahead", "lu_one_launch_panel"))
bad = 0
t0 = time.time()
for r in range(reps):
    x = step(())
    if not np.array_equal(x, ref):
        bad += 1
        print("rep %d: MISMATCH, max |diff| %.3e" % (r, np.max(np.abs(x - ref))))
print("n=%d: %d repetitions of the default path against the launch-per-step path: %d mismatches, %d expired waits (%.1f s)" % (n, reps, bad, expired, time.time() - t0))
sys.exit(1 if (bad or expired) else 0)  # (round 6: QN_LU_SPLIT_MIN=0 in the environment shares every panel's pivot chain between four workgroups)
