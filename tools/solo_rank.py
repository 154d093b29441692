#!/usr/bin/env python3
"""ONE rank of a row-sharded run, alone on the GPU, with the true data of the other ranks: what a GPU of an N-GPU node would
execute, measured on a one-GPU box.

    python tools/solo_rank.py record  N DIM ITERS FILE      all N ranks as threads of this process (tests/thread_ranks.py), synchronous
                                                           host exchange; every exchange of every rank is written to FILE (.npz)
    python tools/solo_rank.py replay  N DIM ITERS FILE RANK [first]
                                                           rank RANK alone: its exchange is answered from FILE, looked up by the
                                                           bytes it sends (which must be the recorded ones: the run is deterministic)

`replay` runs PIPELINED (stream-ordered host exchange: the launch pattern RCCL runs use), so a launch slot the machine does not
use sends what its buffer held before -- a slice recorded earlier -- and gets that exchange's answer: harmless, nothing reads it.
Run `replay` under `rocprofv3 --kernel-trace --stats -- python3 tools/solo_rank.py replay ...` for the per-kernel durations of
one rank (profiles/r04_*); the exchange itself is host-staged here and its duration means nothing.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED = 0x5EED0001
KAPPA = 1.0e3


def inputs(n):  # bench.py's
    diag = KAPPA ** (np.arange(n, dtype=np.float64) / max(n - 1, 1))
    rng = np.random.Generator(np.random.Philox(key=SEED))
    return diag, rng.standard_normal(n), rng.standard_normal(n)


def key_of(a):
    return hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=16).hexdigest()


def run(qn, ctx, n, iters, first_generation, warm=0):
    diag, b, x0 = inputs(n)
    obj = qn.Quadratic.synthetic(n, SEED, diag, b, ctx=ctx)
    s = qn.BFGS(1e-10, x0, ctx=ctx)
    if first_generation:
        s.set_option("second_generation", 0)
    s.set_trace(iters + warm, with_x=False)
    t = []
    for k in ([warm, iters] if warm else [iters]):  # (a continued call: the warm-up is outside the timed call, the run is one run)
        ctx.synchronize()
        t0 = time.perf_counter()
        try:
            s.minimize(qn.MoreThuente(), obj, k, 20)
        except qn.MaxIterReached:
            pass
        ctx.synchronize()
        t.append(time.perf_counter() - t0)
    st = s.stats()
    x = s.x()
    tr, _ = s.trace()
    return {"x": x, "stats": st, "seconds": t[-1], "f": [r["f"] for r in tr]}, (s, obj)


def main():
    mode, world, n, iters, path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    first_generation = "first" in sys.argv[6:]
    import __graft_entry__ as ge
    qn = ge.load_package()
    if mode == "record":
        from thread_ranks import ThreadGroup, run_ranks
        book = [dict() for _ in range(world)]

        def body(rank, world_, group):
            inner = group.allgather_fn(rank)

            def fn(send, recv):
                k = key_of(send)
                inner(send, recv)
                book[rank][k] = np.array(recv, copy=True)
            ctx = qn.Context(0, rank=rank, world=world_, host_allgather=fn)
            out, keep = run(qn, ctx, n, iters, first_generation)
            group.sync()
            return {"x": out["x"], "f": out["f"], "path": out["stats"]["path"]}
        res = run_ranks(world, body)
        for r in res[1:]:
            assert np.array_equal(r["x"], res[0]["x"])
        flat = {}
        for r in range(world):
            for k, v in book[r].items():
                flat[f"r{r}_{k}"] = v
        np.savez(path, x=res[0]["x"], f=np.array(res[0]["f"]), **flat)
        print(json.dumps({"recorded": {f"rank{r}": len(book[r]) for r in range(world)}, "path": res[0]["path"], "f_last": res[0]["f"][-1]}))
        return
    rank = int(sys.argv[6])
    rec = np.load(path)
    prefix = f"r{rank}_"
    book = {k[len(prefix):]: rec[k] for k in rec.files if k.startswith(prefix)}
    hits = [0, 0]

    def fn(send, recv):
        g = book.get(key_of(send))
        if g is not None and g.size == recv.size:
            recv[:] = g
            hits[0] += 1
        else:  # a slot nothing reads (predicated-off launches past the recorded run)
            recv[:] = np.tile(send, world)
            hits[1] += 1
    ctx = qn.Context(0, rank=rank, world=world, host_allgather=fn)
    ctx.set_host_exchange_async(True)
    out, keep = run(qn, ctx, n, iters, first_generation)
    st = out["stats"]
    ok = bool(np.array_equal(out["x"], rec["x"]))
    it = max(st["iterations"], 1)
    print(json.dumps({"rank": rank, "world": world, "n": n, "iterations": st["iterations"], "same_bits_as_the_recorded_run": ok,
                      "exchanges_answered_from_the_record": hits[0], "exchanges_nobody_reads": hits[1],
                      "path": st["path"], "launches": st["launches"], "launches_per_iteration": st["launches"] / it,
                      "collectives": {"n_vectors": st["total_xchg_vector"], "scalars": st["total_xchg_scalar"]},
                      "evaluations": st["oracle_evals"], "matrix_bytes_per_pass": st["matrix_bytes_per_pass"],
                      "wall_seconds_with_host_staged_exchange": out["seconds"]}))
    assert ok, "the solo rank did not reproduce the recorded run"


if __name__ == "__main__":
    main()
