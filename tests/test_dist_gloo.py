"""CPU tests of the N > 1 path with world_size 2 over gloo: the partition, the all-gather plumbing and the
sharding semantics (row-sharded mat-vec + all-gather + replicated scalar work == unsharded, bit for bit)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(mode, tmp_path, nproc=2, timeout=300, extra_env=None):
    out = os.path.join(str(tmp_path), f"dist_{mode}.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), mode, out]
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    with open(out) as fh:
        return json.load(fh)


def test_partition_function(qn):
    for n, world in [(1, 1), (2, 1), (16, 1), (17, 1), (4096, 1), (4096, 8), (32768, 8), (700, 2), (203, 2), (1000, 4)]:
        rpr, n_pad = qn.partition(n, world)
        assert rpr % 16 == 0 and n_pad == rpr * world and n_pad >= n
        assert rpr * (world - 1) < n + 16 * world  # no rank is left without rows except through rounding
        los = [qn.dist.row_range(n, r, world, rpr) for r in range(world)]
        assert los[0][0] == 0 and los[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(los, los[1:]))
    assert qn.partition(4096, 1) == (4096, 4096) and qn.partition(32768, 8) == (4096, 32768)


def test_world2_gloo_sharding_semantics(tmp_path):
    res = launch("cpu", tmp_path)
    assert len(res) == 2
    for r in res:
        assert r["allgather_ok"]
        assert r["sharded_equals_unsharded_bitwise"]
    assert res[0]["x_hex"] == res[1]["x_hex"]  # replicated scalar work: both ranks end on the same bits
    assert res[0]["partition"][:2] == res[1]["partition"][:2] == [112, 224]
    assert res[0]["partition"][2:] == [0, 112] and res[1]["partition"][2:] == [112, 203]
