"""Ranks as THREADS of one process: rehearsal of the row-sharded partitions with more ranks than the GPU box admits processes.

The box lets at most 6 processes use its GPU, so BASELINE config 3's 8-way partition cannot be rehearsed with one process per
rank.  The library's host-staged exchange is a callback, so each rank can just as well be a thread with its own qn_context
(rank r of P, same device): ctypes releases the GIL around every library call, the callback meets the other ranks at a
threading.Barrier.  Only the synchronous exchange is used here -- stream-ordered host nodes of eight streams would all run on
the HIP runtime's callback thread and wait for each other; the process-based tests (2 and 3 ranks) cover that mode."""
import threading

import numpy as np


class ThreadGroup:
    def __init__(self, world, timeout=600.0):
        self.world = world
        self.timeout = timeout
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def allgather_fn(self, rank):
        def fn(send, recv):
            self.slots[rank] = np.array(send, copy=True)
            self.barrier.wait(self.timeout)
            count = self.slots[rank].size
            for r in range(self.world):
                recv[r * count:(r + 1) * count] = self.slots[r]
            self.barrier.wait(self.timeout)  # nobody overwrites its slot before everybody has read it
        return fn

    def sync(self):
        self.barrier.wait(self.timeout)


def run_ranks(world, body, timeout=600.0):
    """body(rank, world, group) on `world` threads; returns the list of results in rank order, re-raises the first failure."""
    group = ThreadGroup(world, timeout)
    out, err = [None] * world, [None] * world

    def run(r):
        try:
            out[r] = body(r, world, group)
        except BaseException as e:  # noqa: BLE001
            err[r] = e
            group.barrier.abort()  # the others must not wait for a rank that is gone

    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout + 60.0)
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    if any(t.is_alive() for t in th):
        raise TimeoutError("a rank thread did not finish")
    return out
