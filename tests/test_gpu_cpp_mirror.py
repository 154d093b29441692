"""GPU test: the C++ host mirror (include/qn_solver.hpp) running the reference's examples/quadratic.rs,
the bfgs_backtracking unit test and the error paths (examples/quadratic.cpp, built by __graft_entry__.build())."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_examples_quadratic_cpp():
    exe = os.path.join(ROOT, "examples", "quadratic.bin")
    if not os.path.exists(exe):
        import __graft_entry__ as ge
        ge.build()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "f(x): 0" in p.stdout and "k: 2" in p.stdout  # examples/quadratic.rs:43 assert_eq!(eval.f(), &0.0)
    assert "error path: Max iter reached" in p.stdout and p.stdout.strip().endswith("ok")
