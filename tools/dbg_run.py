"""Diagnostic: a few short runs at the sizes given on the command line, progress on stderr (to locate a crash)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
qn = ge.load_package()
import problems as P
for n in [int(v) for v in sys.argv[1:]] or [1280, 4096]:
    diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
    obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
    for sync in (1, 0):
        for tiling in (None, ("row_slivers", 0), ("folded_accept_reduce", 1)):
            print("n", n, "sync", sync, "tiling", tiling, file=sys.stderr, flush=True)
            s = qn.BFGS(1e-10, x0)
            s.set_trace(12, with_x=False)
            s.set_sync_mode(sync)
            if tiling:
                s.configure(*tiling)
            try:
                s.minimize(qn.MoreThuente(), obj, 12, 20)
            except qn.MaxIterReached:
                pass
            tr, _ = s.trace()
            print("   f", [r["f"] for r in tr[-3:]], "t", [r["t"] for r in tr[-3:]], "path", s.stats()["path"], file=sys.stderr, flush=True)
