"""GPU test of `qn_oracle.kind = QN_ORACLE_DEVICE_FN` (include/qn_hip.h): the caller's own HIP objective
(examples/device_closure.hip, built by __graft_entry__.build()) is called on the solver's stream with device pointers, where the
reference takes `impl FnMut(&DVector<f64>) -> FuncEvalMultivariate` (ls_solver.rs:69).  The objective is a chain of double
wells -- non-quadratic and non-convex, so More-Thuente brackets and interpolates (cases 2-4 occur) and the curvature condition
fails now and then -- and the run is compared with the CPU oracle driven by the same formula in numpy: same iteration count,
same line-search cases, same number of closure calls, iterates to the parity sweep's tolerances."""
import ctypes as C
import os

import numpy as np
import pytest

import mt_workloads as W
from test_gpu_parity import _compare, _ls

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "examples", "libdevice_closure.so")


def _numpy_chain(a, c):
    def fn(x):
        w = x * x - a
        d = np.diff(x)
        g = x * w
        g[:-1] -= c * d
        g[1:] += c * d
        return 0.25 * np.sum(w * w) + 0.5 * c * np.sum(d * d), g
    return fn


class _Chain:
    def __init__(self, qn, a, c):
        if not os.path.exists(LIB):
            import __graft_entry__ as ge
            ge.build()
        self.dll = C.CDLL(LIB)
        self.dll.double_well_chain_create.restype = C.c_void_p
        self.dll.double_well_chain_create.argtypes = [C.c_size_t, C.POINTER(C.c_double), C.c_double]
        self.dll.double_well_chain_destroy.argtypes = [C.c_void_p]
        self.dll.double_well_chain_calls.restype = C.c_ulonglong
        self.dll.double_well_chain_calls.argtypes = [C.c_void_p]
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.user = self.dll.double_well_chain_create(a.size, a.ctypes.data_as(C.POINTER(C.c_double)), float(c))
        assert self.user
        self.closure = qn.DeviceClosure(self.dll.double_well_chain_eval, self.user, keep=self)

    def calls(self):
        return self.dll.double_well_chain_calls(self.user)

    def close(self):
        self.dll.double_well_chain_destroy(self.user)
        self.user = None


def _problem(n, seed=5):
    rng = np.random.default_rng(seed)
    a = rng.uniform(0.5, 2.0, n)
    x0 = rng.uniform(-2.0, 2.0, n)
    return a, 0.3, x0


@pytest.mark.parametrize("memoize", [0, 1])
@pytest.mark.parametrize("method,lsname,n", [("bfgs", "mt", 200), ("dfp", "mt", 96), ("bfgs", "bt", 200), ("gd", "mt", 1500)])
def test_device_closure_vs_oracle(qn, qo, method, lsname, n, memoize):
    a, c, x0 = _problem(n)
    iters = 40
    calls_ref = [0]
    fn = _numpy_chain(a, c)

    def counted(x):
        calls_ref[0] += 1
        return fn(x)
    kind = {"bfgs": qo.BFGS, "dfp": qo.DFP, "gd": qo.GRADIENT_DESCENT}[method]
    ref = qo.Solver(kind, 1e-9, x0, qo.UPDATE_AS_WRITTEN)
    st_ref = ref.minimize(_ls(qo, lsname), counted, iters, 30, trace_cap=iters, trace_x=True)

    ch = _Chain(qn, a, c)
    try:
        s = {"bfgs": qn.BFGS, "dfp": qn.DFP, "gd": qn.GradientDescent}[method](1e-9, x0)
        s.memoize = memoize
        s.set_trace(iters, with_x=True)
        st = 0
        try:
            s.minimize(_ls(qn, lsname), ch.closure, iters, 30)
        except qn.MaxIterReached:
            st = 1
        tr, xs = s.trace()
        w = _compare(tr, xs, ref.trace, ref.trace_x)
        if w == len(ref.trace):
            assert st == st_ref and len(tr) == len(ref.trace)
            if memoize == 0:  # closure calls: the reference's sequence, call for call
                assert ch.calls() == calls_ref[0] == s.stats()["oracle_calls"]
            else:
                assert ch.calls() == s.stats()["oracle_evals"] < calls_ref[0]
        stats = s.stats()
        assert stats["path"] & 1 == 0  # a closure runs on the generic path (the fused kernels evaluate the built-in quadratic)
        if lsname == "mt" and method != "gd":
            digits = [d for r in ref.trace[:w] for d in W.case_digits(r["ls_cases"])]
            assert any(d != 1 for d in digits), digits  # the non-convex objective leaves case 1
    finally:
        ch.close()


def test_device_closure_as_a_line_search_oracle(qn, qo):
    """LineSearch::compute_step_len (line_search/mod.rs:14-23) with the device closure"""
    n = 300
    a, c, x0 = _problem(n, seed=9)
    fn = _numpy_chain(a, c)
    f0, g0 = fn(x0)
    d = -g0
    t_ref = qo.compute_step_len(qo.morethuente(), x0, f0, g0, d, fn, 30)
    ch = _Chain(qn, a, c)
    try:
        t = qn.MoreThuente().compute_step_len(x0, (f0, g0), d, ch.closure, 30)
        assert abs(t - t_ref) <= 1e-9 * abs(t_ref) and ch.calls() >= 2
    finally:
        ch.close()


def test_device_closure_failure_aborts_the_run(qn):
    """a closure that returns non-zero (here: a dimension it was not built for) ends the run with AbnormalTermination"""
    a, c, x0 = _problem(64)
    ch = _Chain(qn, a, c)
    try:
        s = qn.BFGS(1e-9, np.zeros(65))
        with pytest.raises(qn.AbnormalTermination):
            s.minimize(qn.MoreThuente(), ch.closure, 5, 5)
    finally:
        ch.close()
