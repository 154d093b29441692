"""GPU test: the C ABI driven from plain C99 (examples/ffi_consumer.c) the way the Rust shim of INTEGRATION.md drives it --
`void* user` closure and callback trampolines, status codes as Result<(), SolverError>, getters -- on the reference's
bfgs.rs:141-188 unit test and examples/bfgs_example.rs, with the oracle's call counts as the expected values."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ffi_consumer_c99(qo):
    exe = os.path.join(ROOT, "examples", "ffi_consumer.bin")
    if not os.path.exists(exe):
        import __graft_entry__ as ge
        ge.build()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.strip().endswith("ok") and "error path: Error in input parameters" in p.stdout
    got = {m.group(1): (int(m.group(2)), int(m.group(3)))
           for m in re.finditer(r"(\w+): f = \S+, k = (\d+), oracle calls = (\d+)", p.stdout)}

    def ref(fn, x0, tol, mi, ml):
        calls = [0]

        def oracle(x):
            calls[0] += 1
            return fn(x)
        s = qo.Solver(qo.BFGS, tol, np.array(x0, dtype=float))
        assert s.minimize(qo.morethuente(), oracle, mi, ml) == 0
        return s.k, calls[0]
    a = ref(lambda x: (0.5 * ((x[0] + 1.0) ** 2 + (x[1] - 1.0) ** 2), np.array([x[0] + 1.0, x[1] - 1.0])), [180.0, 152.0], 1e-12, 1000, 100000)
    b = ref(lambda x: (x[0] ** 2 + 2 * x[1] ** 2 + 3 * x[2] ** 2 + x[0] * x[1] + x[1] * x[2],
                       np.array([2 * x[0] + x[1], 4 * x[1] + x[0] + x[2], 6 * x[2] + x[1]])), [1.0, 1.0, 1.0], 1e-8, 50, 20)
    assert got["bfgs_morethuente"] == a and got["bfgs_example"] == b  # same iteration count, same number of closure calls
