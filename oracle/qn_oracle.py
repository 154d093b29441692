"""ctypes binding of the CPU oracle (oracle/qn_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.  Parity status: unpinned beyond the
reference's own known-answer tests (see qn_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# QN_ORACLE_LIB names another build of the same source: `make -C oracle asan` -> libqn_oracle_asan.so (gcc -fsanitize=address,undefined; the
# process then needs the sanitizer runtime preloaded -- tests/test_oracle_sanitizers.py starts one that has it).  CPU only.
_LIB_PATH = os.environ.get("QN_ORACLE_LIB") or os.path.join(_HERE, "libqn_oracle.so")

OK, MAX_ITER_REACHED, OUT_OF_DOMAIN, ERROR_INPUT_PARAMS, ABNORMAL_TERMINATION = range(5)
BFGS, DFP, GRADIENT_DESCENT, NEWTON, SR1 = 0, 1, 2, 3, 4
UPDATE_AS_WRITTEN, UPDATE_RANK2 = 0, 1
LS_MORETHUENTE, LS_BACKTRACKING, LS_MORETHUENTE_B, LS_BACKTRACKING_B = 0, 1, 2, 3


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "qn_oracle.c")
    hdr = os.path.join(_HERE, "qn_oracle.h")
    if os.environ.get("QN_ORACLE_LIB"):  # (a named build: whoever named it has built it)
        return _LIB_PATH
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libqn_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class LineSearch(C.Structure):
    _fields_ = [("kind", C.c_int),
                ("c1", C.c_double), ("c2", C.c_double), ("t_min", C.c_double), ("t_max", C.c_double),
                ("delta_min", C.c_double), ("delta", C.c_double), ("delta_max", C.c_double),
                ("bt_c1", C.c_double), ("bt_beta", C.c_double),
                ("lower_bound", C.c_void_p), ("upper_bound", C.c_void_p)]


class TraceRec(C.Structure):
    _fields_ = [("f", C.c_double), ("gnorm", C.c_double), ("t", C.c_double),
                ("s_norm", C.c_double), ("y_norm", C.c_double),
                ("n_evals", C.c_int32), ("ls_iters", C.c_int32), ("ls_cases", C.c_int32), ("updated", C.c_int32)]


class Trace(C.Structure):
    _fields_ = [("rec", C.POINTER(TraceRec)), ("cap", C.c_size_t), ("len", C.c_size_t),
                ("xs", C.POINTER(C.c_double)), ("n_oracle_calls", C.c_size_t)]


class Quadratic(C.Structure):
    _fields_ = [("n", C.c_size_t), ("q", C.c_void_p), ("b", C.c_void_p), ("nthreads", C.c_int), ("calls", C.c_size_t)]


class LogSumExp(C.Structure):
    _fields_ = [("m", C.c_size_t), ("n", C.c_size_t), ("a", C.c_void_p), ("c", C.c_void_p),
                ("mu", C.c_double), ("nthreads", C.c_int), ("calls", C.c_size_t)]


ORACLE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double))
CALLBACK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
HESSIAN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_double))

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    L.qo_solver_create.restype = C.c_void_p
    L.qo_solver_create.argtypes = [C.c_int, C.c_double, dp, C.c_size_t, C.c_int, C.c_int]
    L.qo_solver_destroy.argtypes = [C.c_void_p]
    L.qo_minimize.restype = C.c_int
    L.qo_minimize.argtypes = [C.c_void_p, C.POINTER(LineSearch), C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                              C.c_void_p, C.c_void_p, C.POINTER(Trace)]
    L.qo_compute_step_len.restype = C.c_double
    L.qo_compute_step_len.argtypes = [C.POINTER(LineSearch), dp, C.c_double, dp, dp, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
    L.qo_solver_n.restype = C.c_size_t
    L.qo_solver_n.argtypes = [C.c_void_p]
    L.qo_solver_k.restype = C.c_size_t
    L.qo_solver_k.argtypes = [C.c_void_p]
    L.qo_solver_x.restype = dp
    L.qo_solver_x.argtypes = [C.c_void_p]
    L.qo_solver_inv_hessian.restype = dp
    L.qo_solver_bytes_streamed.restype = C.c_double
    L.qo_solver_bytes_streamed.argtypes = [C.c_void_p]
    L.qo_solver_inv_hessian.argtypes = [C.c_void_p]
    L.qo_solver_s_norm.restype = C.c_int
    L.qo_solver_s_norm.argtypes = [C.c_void_p, dp]
    L.qo_solver_y_norm.restype = C.c_int
    L.qo_solver_y_norm.argtypes = [C.c_void_p, dp]
    L.qo_solver_set_inv_hessian.argtypes = [C.c_void_p, dp]
    L.qo_solver_set_bounds.argtypes = [C.c_void_p, dp, dp]
    L.qo_solver_set_hessian_fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.qo_solver_decrement_squared.restype = C.c_int
    L.qo_solver_decrement_squared.argtypes = [C.c_void_p, dp]
    L.qo_quadratic_hessian.restype = C.c_int
    L.qo_dot.restype = C.c_double
    L.qo_dot.argtypes = [dp, dp, C.c_size_t]
    L.qo_norm.restype = C.c_double
    L.qo_norm.argtypes = [dp, C.c_size_t]
    L.qo_gemv_colsweep.argtypes = [dp, dp, dp, C.c_size_t]
    L.qo_axpy_new.argtypes = [dp, C.c_double, dp, dp, C.c_size_t]
    L.qo_quadratic_eval.restype = C.c_int
    L.qo_quadratic_eval.argtypes = [C.c_void_p, dp, C.c_size_t, dp, dp]
    L.qo_logsumexp_eval.restype = C.c_int
    L.qo_logsumexp_eval.argtypes = [C.c_void_p, dp, C.c_size_t, dp, dp]
    L.qo_synth_u.restype = C.c_double
    L.qo_synth_u.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    L.qo_synth_fill_rows.argtypes = [dp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64, dp, C.c_int]
    L.qo_morethuente_default.argtypes = [C.POINTER(LineSearch)]
    L.qo_backtracking_new.argtypes = [C.POINTER(LineSearch), C.c_double, C.c_double]
    L.qo_max_threads.restype = C.c_int
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def morethuente(**kw):
    ls = LineSearch()
    lib().qo_morethuente_default(C.byref(ls))
    for k, v in kw.items():
        setattr(ls, k, v)
    return ls


def backtracking(c1, beta):
    ls = LineSearch()
    lib().qo_backtracking_new(C.byref(ls), c1, beta)
    return ls


def _with_bounds(ls, n, lb, ub):
    ls._lb = _f64(np.full(n, -np.inf) if lb is None else lb)
    ls._ub = _f64(np.full(n, np.inf) if ub is None else ub)
    ls.lower_bound, ls.upper_bound = ls._lb.ctypes.data, ls._ub.ctypes.data
    return ls


def morethuente_b(n, lb=None, ub=None, **kw):
    """MoreThuenteB::new(n).with_lower_bound(..).with_upper_bound(..) (morethuente_b.rs:18-39)"""
    ls = morethuente(**kw)
    ls.kind = LS_MORETHUENTE_B
    return _with_bounds(ls, n, lb, ub)


def backtracking_b(c1, beta, lb, ub):
    """BackTrackingB::new(c1, beta, lower, upper) (backtracking_b.rs:10-23)"""
    ls = backtracking(c1, beta)
    ls.kind = LS_BACKTRACKING_B
    return _with_bounds(ls, len(lb), lb, ub)


def dot(a, b):
    a, b = _f64(a), _f64(b)
    return lib().qo_dot(_dp(a), _dp(b), a.size)


def norm(a):
    a = _f64(a)
    return lib().qo_norm(_dp(a), a.size)


def gemv_colsweep(a_colmajor, x):
    """y = A x with nalgebra's column sweep; `a_colmajor` is a 2-D array read as A[i, j]."""
    a = np.asfortranarray(a_colmajor, dtype=np.float64)
    x = _f64(x)
    y = np.empty(x.size)
    lib().qo_gemv_colsweep(a.ctypes.data_as(C.POINTER(C.c_double)), _dp(x), _dp(y), x.size)
    return y


def axpy_new(x, t, d):
    x, d = _f64(x), _f64(d)
    out = np.empty_like(x)
    lib().qo_axpy_new(_dp(x), float(t), _dp(d), _dp(out), x.size)
    return out


def synth_u(seed, i, j):
    return lib().qo_synth_u(seed, i, j)


def synth_rows(n, row0, nrows, seed, diag, nthreads=1):
    diag = _f64(diag)
    out = np.empty((nrows, n))
    lib().qo_synth_fill_rows(_dp(out), n, row0, nrows, seed, _dp(diag), nthreads)
    return out


def max_threads():
    return lib().qo_max_threads()


class PyOracle:
    """Wraps a Python closure x -> (f, g) as a qo_oracle_fn."""

    def __init__(self, fn):
        self.fn = fn
        self.calls = 0
        self.points = []

        def tramp(_user, xp, n, fp, gp):
            x = np.ctypeslib.as_array(xp, shape=(n,)).copy()
            self.calls += 1
            self.points.append(x)
            f, g = self.fn(x)
            fp[0] = float(f)
            g = np.asarray(g, dtype=np.float64)
            for i in range(n):
                gp[i] = g[i]
            return 0

        self.cfn = ORACLE_FN(tramp)

    def ptr(self):
        return C.cast(self.cfn, C.c_void_p), None


class QuadraticOracle:
    """Built-in benchmark objective f = 1/2 x'Qx - b'x (row-major symmetric Q)."""

    def __init__(self, q, b, nthreads=1):
        self.q = _f64(q)
        self.b = _f64(b)
        n = self.b.size
        assert self.q.shape == (n, n)
        self.st = Quadratic(n, self.q.ctypes.data, self.b.ctypes.data, nthreads, 0)

    def ptr(self):
        return C.cast(lib().qo_quadratic_eval, C.c_void_p), C.cast(C.pointer(self.st), C.c_void_p)

    @property
    def calls(self):
        return self.st.calls

    def __call__(self, x):
        x = _f64(x)
        g = np.empty_like(x)
        f = C.c_double()
        lib().qo_quadratic_eval(C.cast(C.pointer(self.st), C.c_void_p), _dp(x), x.size, C.byref(f), _dp(g))
        return f.value, g


class LogSumExpOracle:
    def __init__(self, a, c, mu, nthreads=1):
        self.a = _f64(a)
        self.c = _f64(c)
        m, n = self.a.shape
        self.st = LogSumExp(m, n, self.a.ctypes.data, self.c.ctypes.data, mu, nthreads, 0)

    def ptr(self):
        return C.cast(lib().qo_logsumexp_eval, C.c_void_p), C.cast(C.pointer(self.st), C.c_void_p)

    def __call__(self, x):
        x = _f64(x)
        g = np.empty_like(x)
        f = C.c_double()
        lib().qo_logsumexp_eval(C.cast(C.pointer(self.st), C.c_void_p), _dp(x), x.size, C.byref(f), _dp(g))
        return f.value, g


class Solver:
    """BFGS / DFP / GradientDescent restatement with the reference's getter surface."""

    def __init__(self, method, tol, x0, update_mode=UPDATE_AS_WRITTEN, nthreads=1):
        x0 = _f64(x0)
        self.n = x0.size
        self.h = lib().qo_solver_create(method, tol, _dp(x0), x0.size, update_mode, nthreads)
        self.trace = None
        self.n_oracle_calls = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().qo_solver_destroy(self.h)
            self.h = None

    def minimize(self, ls, oracle, max_iter_solver, max_iter_line_search, callback=None, trace_cap=0, trace_x=False):
        if callable(oracle) and not hasattr(oracle, "ptr"):
            oracle = PyOracle(oracle)
        fn, user = oracle.ptr()
        cb = None
        if callback is not None:
            cb = CALLBACK_FN(lambda _u, _s: callback(self))
        tr = None
        if trace_cap:
            recs = (TraceRec * trace_cap)()
            xs = np.zeros((trace_cap, self.n)) if trace_x else None
            tr = Trace(recs, trace_cap, 0, _dp(xs) if trace_x else None, 0)
        status = lib().qo_minimize(self.h, C.byref(ls), fn, user, max_iter_solver, max_iter_line_search,
                                   C.cast(cb, C.c_void_p) if cb else None, None, C.byref(tr) if tr else None)
        if tr:
            self.trace = [dict(f=r.f, gnorm=r.gnorm, t=r.t, s_norm=r.s_norm, y_norm=r.y_norm, n_evals=r.n_evals,
                               ls_iters=r.ls_iters, ls_cases=r.ls_cases, updated=r.updated)
                          for r in recs[:tr.len]]
            self.trace_x = xs[:tr.len].copy() if trace_x else None
            self.n_oracle_calls = tr.n_oracle_calls
        return status

    def set_bounds(self, lb, ub):
        """BFGSB / DFPB / SR1B::new(tol, x0, lower, upper): projects x and makes the directions P(x - Hg) - x."""
        lb, ub = _f64(lb), _f64(ub)
        lib().qo_solver_set_bounds(self.h, _dp(lb), _dp(ub))

    def set_hessian(self, fn_or_quadratic):
        """Newton: where the Hessian part of the FuncEval comes from (a Python x -> H closure, or a QuadraticOracle)."""
        if isinstance(fn_or_quadratic, QuadraticOracle):
            self._hess_keep = fn_or_quadratic
            lib().qo_solver_set_hessian_fn(self.h, C.cast(lib().qo_quadratic_hessian, C.c_void_p),
                                           C.cast(C.pointer(fn_or_quadratic.st), C.c_void_p))
            return
        n = self.n

        def tramp(_u, xp, nn, hp):
            x = np.ctypeslib.as_array(xp, shape=(nn,)).copy()
            h = np.asfortranarray(fn_or_quadratic(x), dtype=np.float64)
            np.ctypeslib.as_array(hp, shape=(nn * nn,))[:] = h.ravel(order="F")
            return 0
        self._hess_keep = HESSIAN_FN(tramp)
        lib().qo_solver_set_hessian_fn(self.h, C.cast(self._hess_keep, C.c_void_p), None)

    @property
    def decrement_squared(self):
        v = C.c_double()
        return v.value if lib().qo_solver_decrement_squared(self.h, C.byref(v)) else None

    @property
    def x(self):
        return np.ctypeslib.as_array(lib().qo_solver_x(self.h), shape=(self.n,)).copy()

    @property
    def k(self):
        return lib().qo_solver_k(self.h)

    @property
    def bytes_streamed(self):
        """matrix bytes the solver's own sweeps have moved (mat-vecs with H, the rank-2 update); the objective counts its calls"""
        return lib().qo_solver_bytes_streamed(self.h)

    @property
    def approx_inv_hessian(self):
        p = lib().qo_solver_inv_hessian(self.h)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(self.n, self.n)).T.copy()  # column-major -> H[i, j]

    def set_inv_hessian(self, h):
        a = np.asfortranarray(h, dtype=np.float64)
        lib().qo_solver_set_inv_hessian(self.h, a.ctypes.data_as(C.POINTER(C.c_double)))

    @property
    def s_norm(self):
        v = C.c_double()
        return v.value if lib().qo_solver_s_norm(self.h, C.byref(v)) else None

    @property
    def y_norm(self):
        v = C.c_double()
        return v.value if lib().qo_solver_y_norm(self.h, C.byref(v)) else None


def compute_step_len(ls, x, f0, g0, d, oracle, max_iter):
    if callable(oracle) and not hasattr(oracle, "ptr"):
        oracle = PyOracle(oracle)
    fn, user = oracle.ptr()
    x, g0, d = _f64(x), _f64(g0), _f64(d)
    return lib().qo_compute_step_len(C.byref(ls), _dp(x), float(f0), _dp(g0), _dp(d), x.size, fn, user, max_iter)
