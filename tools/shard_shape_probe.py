"""Per-rank kernel durations at the shape one rank of an 8-GPU run sees (n = 32768, 4096 x 32768 shards of H and Q), measured
on ONE GPU: rank 0 of a world-8 partition whose exchange callback duplicates rank 0's slice into every slot.  The numbers the
solver computes are meaningless; launch shapes, bytes and therefore kernel durations are those of a real rank."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as ge
qn = ge.load_package()
import problems as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def dup(send, recv):
    c = send.size
    for r in range(recv.size // c):
        recv[r * c:(r + 1) * c] = send


ctx = qn.Context(0, rank=0, world=world, host_allgather=dup)
diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
obj = qn.Quadratic.synthetic(n, P.SEED, diag, b, ctx=ctx)
for tiling in ((0, 0), (4, 101), (4, 102), (8, 101), (8, 102), (16, 101), (2, 102)):
    s = qn.BFGS(1e-10, x0, ctx=ctx)
    if tiling != (0, 0):
        s.configure(*tiling)
    s.set_sync_mode(1)
    s.set_profiling(True)
    try:
        s.minimize(qn.MoreThuente(), obj, 12, 20)
    except qn.SolverError as e:
        pass
    st = s.stats()
    nh, ne = st["n_hpass_timed"], st["n_eval_timed"]
    if nh and ne:
        th, te = st["t_hpass_ms"] / nh, st["t_eval_ms"] / ne
        print(f"n={n} world={world} tiling={tiling}: h_pass {th*1e3:8.1f} us ({16.0*n*n/world/(th*1e-3)/1e9:6.0f} GB/s, {nh} launches)  "
              f"eval {te*1e3:8.1f} us ({8.0*n*n/world/(te*1e-3)/1e9:6.0f} GB/s, {ne} launches)  k={s.k()}", flush=True)
    else:
        print(f"tiling={tiling}: nothing timed (k={s.k()})")
    del s
