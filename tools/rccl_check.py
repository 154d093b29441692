"""Which librccl does libqn_hip pick up when torch is already imported (as in bench.py), and does the self-test pass?"""
import sys
sys.path.insert(0, ".")
import torch  # noqa: F401  (loads torch's bundled librccl)
import torch.distributed  # noqa: F401
import __graft_entry__ as ge
qn = ge.load_package()
ctx = qn.Context(0)
ctx.comm_selftest()
uid = qn.Context.unique_id()
print("selftest ok; unique id bytes:", len(uid))
libs = sorted({line.split()[-1] for line in open("/proc/self/maps") if "rccl" in line})
print("loaded:", libs)
