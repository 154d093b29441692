"""Host-side mirror of the reference's trait surface over the C ABI (include/qn_hip.h).

Names, argument meaning and error behaviour follow the Rust crate so the parity tests read like the
reference's own tests:

    BFGS::new(tol, x0)                          -> BFGS(tol, x0)                      (bfgs.rs:27-39)
    MoreThuente::default().with_c1(..)          -> MoreThuente().with_c1(..)          (morethuente.rs:16-62)
    BackTracking::new(c1, beta)                 -> BackTracking(c1, beta)             (backtracking.rs:8-10)
    solver.minimize(&mut ls, oracle, a, b, cb)  -> solver.minimize(ls, oracle, a, b, callback)   (ls_solver.rs:66-111)
    Result<(), SolverError>                     -> returns None / raises SolverError  (ls_solver.rs:10-20)
    FuncEvalMultivariate::new(f, g)             -> FuncEvalMultivariate(f, g) or a plain (f, g) tuple (func_eval.rs:4-41)

Everything numeric happens in libqn_hip.so on the GPU; this file only marshals.
"""
import atexit
import ctypes as C
import weakref

import numpy as np

from . import _abi as A

# Every live handle, so that interpreter exit can release them in dependency order (solvers and objectives before their
# context) instead of whatever order module teardown picks -- a solver destroyed after its context would free device memory
# through a dead stream.
_live = {"solver": weakref.WeakSet(), "objective": weakref.WeakSet(), "context": weakref.WeakSet()}


@atexit.register
def _close_all():
    for kind in ("solver", "objective", "context"):
        for obj in list(_live[kind]):
            try:
                obj.close()
            except Exception:  # noqa: BLE001
                pass


# qn_option (include/qn_hip.h, ABI 5)
OPTIONS = {
    "generic_kernels": 1,
    "deferred_update_step": 2,
    "symmetric_storage": 3,
    "second_generation": 4,
    "folded_accept_reduce": 5,
    "row_slivers": 6,
    "eval_pair_instance": 7,
    "eval_mover_multiplier": 8,
    "tail_reduce": 9,
    "bounded_second_generation": 10,
    "newton_pivoted_lu": 11,
    "lu_per_column_panel": 12,
    "lu_lookahead": 13,
    "lu_one_launch_panel": 14,
    "lu_force_wait_expiry": 15,
    "chunks_per_trip": 16,
    "lu_split_role_a": 17,
    "lu_split_min_rows": 18,
    "btb_project_in_eval": 19,
    "eval_zigzag": 20,
    "touch_h_rows": 21,
    "touch_q_rows": 22,
}
OPT_GENERIC_KERNELS = 1
OPT_DEFERRED_UPDATE_STEP = 2
OPT_SYMMETRIC_STORAGE = 3
OPT_SECOND_GENERATION = 4
OPT_FOLDED_ACCEPT_REDUCE = 5
OPT_ROW_SLIVERS = 6
OPT_EVAL_PAIR_INSTANCE = 7
OPT_EVAL_MOVER_MULTIPLIER = 8
OPT_TAIL_REDUCE = 9
OPT_BOUNDED_SECOND_GENERATION = 10
OPT_NEWTON_PIVOTED_LU = 11
OPT_LU_PER_COLUMN_PANEL = 12
OPT_LU_LOOKAHEAD = 13
OPT_LU_ONE_LAUNCH_PANEL = 14
OPT_LU_FORCE_WAIT_EXPIRY = 15
OPT_CHUNKS_PER_TRIP = 16
OPT_LU_SPLIT_ROLE_A = 17
OPT_LU_SPLIT_MIN_ROWS = 18
OPT_BTB_PROJECT_IN_EVAL = 19
OPT_EVAL_ZIGZAG = 20
OPT_TOUCH_H_ROWS = 21
OPT_TOUCH_Q_ROWS = 22


class SolverError(Exception):
    """ls_solver.rs:10-20"""
    code = A.ABNORMAL_TERMINATION

    def __init__(self, msg=None):
        super().__init__(msg or A.lib().qn_status_string(self.code).decode())


class MaxIterReached(SolverError):
    code = A.MAX_ITER_REACHED


class OutOfDomain(SolverError):
    code = A.OUT_OF_DOMAIN


class ErrorInputParams(SolverError):
    code = A.ERROR_INPUT_PARAMS


class AbnormalTermination(SolverError):
    code = A.ABNORMAL_TERMINATION


_ERRORS = {A.MAX_ITER_REACHED: MaxIterReached, A.OUT_OF_DOMAIN: OutOfDomain, A.ERROR_INPUT_PARAMS: ErrorInputParams,
           A.ABNORMAL_TERMINATION: AbnormalTermination}


def _check(status):
    if status == A.OK:
        return
    detail = A.lib().qn_last_error_message().decode()
    cls = _ERRORS.get(status, AbnormalTermination)
    if status in (A.ERROR_INPUT_PARAMS, A.ABNORMAL_TERMINATION) and detail:
        raise cls(f"{A.lib().qn_status_string(status).decode()}: {detail}")
    raise cls()


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(A.dp)


class FuncEvalMultivariate:
    """func_eval.rs:4-41 (f, g; the optional Hessian is not used on this path)."""

    def __init__(self, f, g, hessian=None):
        self._f, self._g = float(f), np.asarray(g, dtype=np.float64)
        self._h = hessian

    @classmethod
    def new(cls, f, g):
        return cls(f, g)

    def with_hessian(self, hessian):  # func_eval.rs:27-30
        self._h = np.asarray(hessian, dtype=np.float64)
        return self

    def hessian(self):
        return self._h

    def f(self):
        return self._f

    def g(self):
        return self._g

    def __iter__(self):
        return iter((self._f, self._g))


class Context:
    """One GPU (+ optionally one rank of a row-sharded group)."""

    def __init__(self, device=0, rank=0, world=1, unique_id=None, host_allgather=None):
        L = A.lib()
        self.h = C.c_void_p()
        self._keep = None
        if world == 1:
            _check(L.qn_context_create(device, C.byref(self.h)))
        elif host_allgather is not None:
            def tramp(_u, send, recv, count):
                s = np.ctypeslib.as_array(send, shape=(count,))
                r = np.ctypeslib.as_array(recv, shape=(count * world,))
                try:
                    host_allgather(s, r)
                    return 0
                except Exception:  # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1
            self._keep = A.HOST_ALLGATHER_FN(tramp)
            _check(L.qn_context_create_sharded_host_exchange(device, rank, world, C.cast(self._keep, C.c_void_p), None, C.byref(self.h)))
        else:
            buf = C.create_string_buffer(bytes(unique_id), A.UNIQUE_ID_BYTES)
            _check(L.qn_context_create_sharded(device, rank, world, buf, C.byref(self.h)))
        self.rank, self.world, self.device = rank, world, device
        _live["context"].add(self)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(A.UNIQUE_ID_BYTES)
        _check(A.lib().qn_comm_unique_id(buf))
        return bytes(buf.raw)

    def synchronize(self):
        _check(A.lib().qn_context_synchronize(self.h))

    def comm_selftest(self):
        _check(A.lib().qn_comm_selftest(self.h))

    def comm_check(self):
        """Collective: one verified rank-tagged all-gather through this context's exchange (RCCL or host-staged)."""
        _check(A.lib().qn_context_comm_check(self.h))

    def exchange_probe(self, count, reps=20):
        """Collective: the latency of one exchange of `count` doubles per rank on this context (qn_context_exchange_probe):
        {"median_us", "min_us", "max_us", "bytes_per_rank"}."""
        out = (C.c_double * 3)()
        _check(A.lib().qn_context_exchange_probe(self.h, int(count), int(reps), out))
        return {"bytes_per_rank": 8 * int(count), "median_us": out[0], "min_us": out[1], "max_us": out[2]}

    def set_allreduce(self, on=True):
        """Sharded symmetric storage: all-reduce the partial n-vectors (RCCL's order) instead of all-gather + rank-order sum."""
        _check(A.lib().qn_context_set_allreduce(self.h, 1 if on else 0))

    def set_trial_vector_exchange(self, on=True):
        """Row-sharded second-generation runs, quadratic objective: every evaluation's collective also carries the rank's partial n-vector of
        the trial point (one grouped all-gather), an accepted evaluation then needs no exchange of its own -- E + 1 collectives per
        iteration instead of E + 2; the same iterates.  Not together with set_allreduce."""
        _check(A.lib().qn_context_set_trial_vector_exchange(self.h, 1 if on else 0))

    def set_host_exchange_async(self, on=True):
        """Host-exchange contexts: run the exchange in stream order (no synchronisation), so sharded runs can be pipelined."""
        _check(A.lib().qn_context_set_host_exchange_async(self.h, 1 if on else 0))

    def event_bracket_overhead_ms(self, reps=200):
        """Mean elapsed time an event / launch / event bracket reports for an empty kernel on the idle stream."""
        out = C.c_double(0.0)
        _check(A.lib().qn_context_event_bracket_overhead(self.h, int(reps), C.byref(out)))
        return out.value

    def close(self):
        if self.h:
            A.lib().qn_context_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def partition(n, world):
    """(rows_per_rank, n_pad) of the row partition used for H and the objective's matrix."""
    rpr, npad = C.c_size_t(), C.c_size_t()
    _check(A.lib().qn_partition(n, world, C.byref(rpr), C.byref(npad)))
    return rpr.value, npad.value


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class _LineSearch:
    """LineSearch (line_search/mod.rs:14-23): `compute_step_len` on its own, as the reference's line-search tests call it."""

    def compute_step_len(self, x_k, eval_x_k, direction_k, oracle, max_iter, ctx=None, memoize=None):
        ctx = ctx or (oracle.ctx if isinstance(oracle, Objective) else default_context())
        x, d = _f64(x_k), _f64(direction_k)
        f0, g0 = (eval_x_k.f(), _f64(eval_x_k.g())) if isinstance(eval_x_k, FuncEvalMultivariate) else (float(eval_x_k[0]), _f64(eval_x_k[1]))
        o = A.OracleStruct()
        keep, err = [], []
        if isinstance(oracle, Objective):
            o.kind = A.ORACLE_OBJECTIVE
            o.objective = oracle.h
            o.memoize = 1 if memoize is None else int(memoize)
        elif isinstance(oracle, DeviceClosure):
            o.kind, o.device_fn, o.device_user = A.ORACLE_DEVICE_FN, oracle.fn, oracle.user
            o.memoize = 0 if memoize is None else int(memoize)
        else:
            def tramp(_u, xp, nn, fp, gp):
                try:
                    res = oracle(np.ctypeslib.as_array(xp, shape=(nn,)).copy())
                    f, g = (res.f(), res.g()) if isinstance(res, FuncEvalMultivariate) else res[:2]
                    fp[0] = float(f)
                    np.ctypeslib.as_array(gp, shape=(nn,))[:] = np.asarray(g, dtype=np.float64)
                    return 0
                except Exception as e:  # noqa: BLE001 -- a panicking closure aborts the run
                    err.append(e)
                    return 1
            cfn = A.HOST_ORACLE_FN(tramp)
            keep.append(cfn)
            o.kind = A.ORACLE_HOST
            o.host_fn = C.cast(cfn, C.c_void_p)
            o.memoize = 0 if memoize is None else int(memoize)
        t = C.c_double(0.0)
        status = A.lib().qn_compute_step_len(ctx.h, C.byref(self.s), x.ctypes.data_as(A.dp), float(f0), g0.ctypes.data_as(A.dp),
                                             d.ctypes.data_as(A.dp), x.size, C.byref(o), int(max_iter), C.byref(t))
        if err:
            raise err[0]
        _check(status)
        return t.value


def _device_dot(a, b, ctx=None):
    """a.b with the library's dot kernel (nalgebra's accumulation order, mod.rs:35,47,55), operands uploaded for the call"""
    ctx = ctx or default_context()
    a, b = _f64(a), _f64(b)
    da, db = DeviceBuffer(ctx, a), DeviceBuffer(ctx, b)
    try:
        return dot(ctx, a.size, da, db)
    finally:
        da.free()
        db.free()


class _WolfeConditions:
    """SufficientDecreaseCondition / CurvatureCondition / WolfeConditions (line_search/mod.rs:25-83) as the reference's line-search
    structs expose them; the dots run on the device.  (The line searches themselves evaluate these tests inside the device state
    machine; these methods exist so that caller code written against the traits keeps working.)"""

    def c1(self):
        return self.s.c1 if self.s.kind in (A.LS_MORETHUENTE, A.LS_MORETHUENTE_B) else self.s.bt_c1

    def c2(self):
        return self.s.c2

    def sufficient_decrease(self, f_k, f_kp1, grad_k, t, direction_k):  # mod.rs:27-36
        return f_kp1 - f_k <= self.c1() * t * _device_dot(grad_k, direction_k)

    def curvature_condition(self, grad_k, grad_kp1, direction_k):  # mod.rs:41-48
        return _device_dot(grad_kp1, direction_k) >= self.c2() * _device_dot(grad_k, direction_k)

    def strong_curvature_condition(self, grad_k, grad_kp1, direction_k):  # mod.rs:49-56
        return abs(_device_dot(grad_kp1, direction_k)) <= self.c2() * abs(_device_dot(grad_k, direction_k))

    def wolfe_conditions_with_directional_derivative(self, f_k, f_kp1, grad_k, grad_kp1, t, direction_k):  # mod.rs:60-71
        return self.sufficient_decrease(f_k, f_kp1, grad_k, t, direction_k) and self.curvature_condition(grad_k, grad_kp1, direction_k)

    def strong_wolfe_conditions_with_directional_derivative(self, f_k, f_kp1, grad_k, grad_kp1, t, direction_k):  # mod.rs:72-83
        return (self.sufficient_decrease(f_k, f_kp1, grad_k, t, direction_k)
                and self.strong_curvature_condition(grad_k, grad_kp1, direction_k))


class MoreThuente(_LineSearch, _WolfeConditions):
    """morethuente.rs:6-62"""

    def __init__(self):
        self.s = A.LineSearchStruct()
        A.lib().qn_morethuente_default(C.byref(self.s))

    @classmethod
    def default(cls):
        return cls()

    def with_deltas(self, delta_min, delta, delta_max):
        _check(A.lib().qn_morethuente_with_deltas(C.byref(self.s), delta_min, delta, delta_max))
        return self

    def with_t_min(self, t_min):
        _check(A.lib().qn_morethuente_with_t_min(C.byref(self.s), t_min))
        return self

    def with_t_max(self, t_max):
        _check(A.lib().qn_morethuente_with_t_max(C.byref(self.s), t_max))
        return self

    def with_c1(self, c1):
        _check(A.lib().qn_morethuente_with_c1(C.byref(self.s), c1))
        return self

    def with_c2(self, c2):
        _check(A.lib().qn_morethuente_with_c2(C.byref(self.s), c2))
        return self


class BackTracking(_LineSearch, _WolfeConditions):
    """backtracking.rs:3-11 (implements SufficientDecreaseCondition only, backtracking.rs:13-18: `c2` is not meaningful here)"""

    def __init__(self, c1, beta):
        self.s = A.LineSearchStruct()
        A.lib().qn_backtracking_new(C.byref(self.s), c1, beta)

    @classmethod
    def new(cls, c1, beta):
        return cls(c1, beta)


class MoreThuenteB(MoreThuente):
    """morethuente_b.rs: More-Thuente whose t_max is clipped to the box (and stays clipped)."""

    def __init__(self, n):
        self.s = A.LineSearchStruct()
        A.lib().qn_morethuente_b_new(C.byref(self.s))
        self.n = n
        self._lb = self._ub = None

    @classmethod
    def new(cls, n):
        return cls(n)

    def with_lower_bound(self, lb):
        self._lb = _f64(lb)
        A.lib().qn_linesearch_with_lower_bound(C.byref(self.s), self._lb.ctypes.data)
        return self

    def with_upper_bound(self, ub):
        self._ub = _f64(ub)
        A.lib().qn_linesearch_with_upper_bound(C.byref(self.s), self._ub.ctypes.data)
        return self

    def t_max(self):
        return self.s.t_max


class BackTrackingB(_LineSearch):
    """backtracking_b.rs: projected trial points, modified Armijo rule."""

    def __init__(self, c1, beta, lower_bound, upper_bound):
        self.s = A.LineSearchStruct()
        self._lb, self._ub = _f64(lower_bound), _f64(upper_bound)
        A.lib().qn_backtracking_b_new(C.byref(self.s), c1, beta, self._lb.ctypes.data, self._ub.ctypes.data)

    @classmethod
    def new(cls, c1, beta, lower_bound, upper_bound):
        return cls(c1, beta, lower_bound, upper_bound)


class Objective:
    """A device-resident objective owned by the library."""

    def __init__(self, ctx, handle, n):
        self.ctx, self.h, self.n = ctx, handle, n
        _live["objective"].add(self)

    def __call__(self, x):
        x = _f64(x)
        g = np.empty(self.n)
        f = C.c_double()
        _check(A.lib().qn_objective_eval(self.h, _dp(x), C.byref(f), _dp(g)))
        return FuncEvalMultivariate(f.value, g)

    def rows(self, row0, nrows):
        out = np.empty((nrows, self.n))
        _check(A.lib().qn_objective_get_rows(self.h, row0, nrows, _dp(out)))
        return out

    def close(self):
        if self.h:
            A.lib().qn_objective_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class DeviceClosure:
    """`qn_oracle.kind = QN_ORACLE_DEVICE_FN` (include/qn_hip.h): the caller's own objective, already on the GPU.  `fn` is the
    address of a `qn_device_oracle_fn` -- int fn(void* user, void* stream, const double* x_dev, size_t n, double* f_dev,
    double* g_dev) -- that ENQUEUES its work on `stream`; the solver never copies x, f or g to the host on this path.  It stands
    where the reference takes `impl FnMut(&DVector<f64>) -> FuncEvalMultivariate` (ls_solver.rs:69); like a host closure it is
    called in the reference's order unless `solver.memoize = 1` declares it a pure function of x."""

    def __init__(self, fn, user=None, keep=None):
        self.fn = C.cast(fn, C.c_void_p)
        self.user = C.c_void_p(user) if not isinstance(user, C.c_void_p) else user
        self.keep = keep  # whatever owns `fn` / `user` (a ctypes.CDLL, a create/destroy pair): kept alive with the closure

    @classmethod
    def from_library(cls, path, symbol, user=None):
        dll = C.CDLL(path)
        return cls(getattr(dll, symbol), user, keep=dll)


class Quadratic(Objective):
    """f = 1/2 x'Qx - b'x on the device."""

    def __init__(self, q, b, ctx=None):
        ctx = ctx or default_context()
        q, b = _f64(q), _f64(b)
        h = C.c_void_p()
        _check(A.lib().qn_quadratic_create(ctx.h, b.size, _dp(q), _dp(b), C.byref(h)))
        super().__init__(ctx, h, b.size)

    @classmethod
    def synthetic(cls, n, seed, diag, b, ctx=None):
        ctx = ctx or default_context()
        diag, b = _f64(diag), _f64(b)
        h = C.c_void_p()
        _check(A.lib().qn_quadratic_create_synthetic(ctx.h, n, seed, _dp(diag), _dp(b), C.byref(h)))
        self = cls.__new__(cls)
        Objective.__init__(self, ctx, h, n)
        return self


class LogSumExp(Objective):
    """f = log sum_i exp(a_i'x + c_i) + mu/2 ||x||^2 on the device (A is m x n row-major, rows sharded over ranks)."""

    def __init__(self, a, c, mu, ctx=None):
        ctx = ctx or default_context()
        a, c = _f64(a), _f64(c)
        m, n = a.shape
        h = C.c_void_p()
        _check(A.lib().qn_logsumexp_create(ctx.h, m, n, _dp(a), _dp(c), float(mu), C.byref(h)))
        super().__init__(ctx, h, n)
        self.m = m


class _SolverBase:
    METHOD = None

    def __init__(self, tol, x0, ctx=None):
        self.ctx = ctx or default_context()
        x0 = _f64(x0)
        self.n = x0.size
        self.h = C.c_void_p()
        _check(A.lib().qn_solver_create(self.ctx.h, self.METHOD, tol, _dp(x0), x0.size, C.byref(self.h)))
        self.memoize = None  # None: 1 for device objectives, 0 for host closures
        self._trace_cap = 0
        _live["solver"].add(self)

    @classmethod
    def new(cls, tol, x0, ctx=None):
        return cls(tol, x0, ctx)

    def reset(self, x0):
        """Back to the state right after `new(tol, x0)`."""
        x0 = _f64(x0)
        _check(A.lib().qn_solver_reset(self.h, _dp(x0)))

    def close(self):
        if getattr(self, "h", None):
            A.lib().qn_solver_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ---- LineSearchSolver::minimize (ls_solver.rs:66-111) ----
    def minimize(self, line_search, oracle, max_iter_solver, max_iter_line_search, callback=None):
        L = A.lib()
        o = A.OracleStruct()
        keep = []
        if isinstance(oracle, Objective):
            o.kind = A.ORACLE_OBJECTIVE
            o.objective = oracle.h
            o.memoize = 1 if self.memoize is None else int(self.memoize)
        elif isinstance(oracle, DeviceClosure):
            o.kind, o.device_fn, o.device_user = A.ORACLE_DEVICE_FN, oracle.fn, oracle.user
            o.memoize = 0 if self.memoize is None else int(self.memoize)  # a closure: the reference's call sequence by default
        else:
            n = self.n
            err = []

            def tramp(_u, xp, nn, fp, gp):
                try:
                    x = np.ctypeslib.as_array(xp, shape=(nn,)).copy()
                    res = oracle(x)
                    f, g = (res.f(), res.g()) if isinstance(res, FuncEvalMultivariate) else res[:2]
                    fp[0] = float(f)
                    np.ctypeslib.as_array(gp, shape=(nn,))[:] = np.asarray(g, dtype=np.float64)
                    return 0
                except Exception as e:  # noqa: BLE001 -- a panicking closure aborts the run
                    err.append(e)
                    return 1
            def htramp(_u, xp, nn, hp):  # Newton: the Hessian part of the closure's FuncEval (newton/mod.rs:31-35)
                try:
                    x = np.ctypeslib.as_array(xp, shape=(nn,)).copy()
                    res = oracle(x)
                    hm = res.hessian() if isinstance(res, FuncEvalMultivariate) else (res[2] if len(res) > 2 else None)
                    if hm is None:
                        raise RuntimeError("Hessian not available in the oracle")
                    np.ctypeslib.as_array(hp, shape=(nn * nn,))[:] = np.asfortranarray(hm, dtype=np.float64).ravel(order="F")
                    return 0
                except Exception as e:  # noqa: BLE001
                    err.append(e)
                    return 1
            cfn = A.HOST_ORACLE_FN(tramp)
            keep.append(cfn)
            o.kind = A.ORACLE_HOST
            o.host_fn = C.cast(cfn, C.c_void_p)
            if self.METHOD == A.NEWTON:
                hfn = A.HOST_HESSIAN_FN(htramp)
                keep.append(hfn)
                o.host_hessian_fn = C.cast(hfn, C.c_void_p)
            o.memoize = 0 if self.memoize is None else int(self.memoize)
        cb = None
        if callback is not None:
            cb = A.CALLBACK_FN(lambda _u, _s: callback(self))
            keep.append(cb)
        status = L.qn_minimize(self.h, C.byref(line_search.s), C.byref(o), max_iter_solver, max_iter_line_search,
                               C.cast(cb, C.c_void_p) if cb else None, None)
        if not isinstance(oracle, (Objective, DeviceClosure)) and err:
            raise err[0]
        _check(status)

    # ---- getters (derive_getters, bfgs.rs:3-12 ; LineSearchSolver::xk/k, bfgs.rs:52-63) ----
    def x(self):
        out = np.empty(self.n)
        _check(A.lib().qn_solver_get_x(self.h, _dp(out)))
        return out

    xk = x

    def set_x(self, x):
        x = _f64(x)
        _check(A.lib().qn_solver_set_x(self.h, _dp(x)))

    def k(self):
        return A.lib().qn_solver_k(self.h)

    def set_k(self, k):  # k_mut(), bfgs.rs:58-63
        _check(A.lib().qn_solver_set_k(self.h, int(k)))

    def identity(self):  # derive_getters on bfgs.rs:9 `identity: DMatrix<Floating>`; the GPU solver keeps no copy of it
        return np.eye(self.n)

    def tol(self):
        return A.lib().qn_solver_tol(self.h)

    def _opt(self, fn):
        v, some = C.c_double(), C.c_int()
        _check(fn(self.h, C.byref(v), C.byref(some)))
        return v.value if some.value else None

    def s_norm(self):
        return self._opt(A.lib().qn_solver_s_norm)

    def y_norm(self):
        return self._opt(A.lib().qn_solver_y_norm)

    def next_iterate_too_close(self):
        v = C.c_int()
        _check(A.lib().qn_solver_next_iterate_too_close(self.h, C.byref(v)))
        return bool(v.value)

    def gradient_next_iterate_too_close(self):
        v = C.c_int()
        _check(A.lib().qn_solver_gradient_next_iterate_too_close(self.h, C.byref(v)))
        return bool(v.value)

    def has_converged(self, eval_x_k):
        """bfgs.rs:64-76 evaluated on a host FuncEval (used by the reference's tests after minimize)."""
        if self.next_iterate_too_close() or self.gradient_next_iterate_too_close():
            return True
        g = eval_x_k.g() if isinstance(eval_x_k, FuncEvalMultivariate) else eval_x_k[1]
        return float(np.sqrt(np.dot(g, g))) < self.tol()

    def approx_inv_hessian(self, all_ranks=True):
        out = np.zeros((self.n, self.n), order="F")
        _check(A.lib().qn_solver_get_inv_hessian(self.h, out.ctypes.data_as(A.dp), 1 if all_ranks else 0))
        return np.ascontiguousarray(out)

    def compute_direction(self, eval_x_k):
        """ComputeDirection::compute_direction (ls_solver.rs:3-8; bfgs.rs:42-49, dfp.rs:42-49, gradient_descent.rs:24-30)"""
        g = _f64(eval_x_k.g() if isinstance(eval_x_k, FuncEvalMultivariate) else eval_x_k[1])
        if g.size != self.n:
            raise ErrorInputParams("gradient has the wrong dimension")
        d = np.empty(self.n)
        _check(A.lib().qn_solver_compute_direction(self.h, g.ctypes.data_as(A.dp), d.ctypes.data_as(A.dp)))
        return d

    def secant_update(self, s, y):
        """the inverse-Hessian half of `update_next_iterate` (bfgs.rs:92-130, dfp.rs:92-118) on its own"""
        s, y = _f64(s), _f64(y)
        if s.size != self.n or y.size != self.n:
            raise ErrorInputParams("s / y have the wrong dimension")
        _check(A.lib().qn_solver_secant_update(self.h, s.ctypes.data_as(A.dp), y.ctypes.data_as(A.dp)))

    def set_approx_inv_hessian(self, h):
        a = np.asfortranarray(h, dtype=np.float64)
        _check(A.lib().qn_solver_set_inv_hessian(self.h, a.ctypes.data_as(A.dp)))

    # ---- instrumentation ----
    def set_trace(self, cap, with_x=False):
        self._trace_cap = cap
        self._trace_x = with_x
        _check(A.lib().qn_solver_set_trace(self.h, cap, 1 if with_x else 0))

    def trace(self):
        cap = self._trace_cap
        recs = (A.TraceRec * max(cap, 1))()
        ln = C.c_size_t()
        xs = np.zeros((max(cap, 1), self.n)) if getattr(self, "_trace_x", False) else None
        _check(A.lib().qn_solver_get_trace(self.h, recs, cap, C.byref(ln), _dp(xs) if xs is not None else None))
        out = [dict(f=r.f, gnorm=r.gnorm, t=r.t, s_norm=r.s_norm, y_norm=r.y_norm, n_evals=r.n_evals, ls_iters=r.ls_iters,
                    ls_cases=r.ls_cases, updated=r.updated) for r in recs[:ln.value]]
        return out, (xs[:ln.value] if xs is not None else None)

    def stats(self):
        st = A.Stats()
        _check(A.lib().qn_solver_get_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in A.Stats._fields_}

    def set_profiling(self, on):
        _check(A.lib().qn_solver_set_profiling(self.h, 1 if on else 0))

    def set_sync_mode(self, sync):
        _check(A.lib().qn_solver_set_sync_mode(self.h, int(sync)))

    def set_tiling(self, rows_per_block=0, col_splits=0):
        """tuning of the fused row kernels (rows per tile 2 / 4 / 8 / 16, column splits); diagnostics are named options: set_option"""
        _check(A.lib().qn_solver_set_tiling(self.h, rows_per_block, col_splits))

    def set_option(self, option, value=1):
        """qn_solver_set_option: `option` is a qn_option value (the OPT_* constants of this module) or its lower-case name without the prefix
        ("second_generation", "newton_pivoted_lu", ...); value != 0 switches the named thing on, 0 off."""
        if isinstance(option, str):
            option = OPTIONS[option]
        _check(A.lib().qn_solver_set_option(self.h, int(option), int(value)))

    def configure(self, what, value=0):
        """one entry point for parameter lists in tests and tools: configure(rows_per_block, col_splits) tunes, configure("option_name", v) sets a
        named option"""
        if isinstance(what, str):
            self.set_option(what, value)
        else:
            self.set_tiling(what, value)


class BFGS(_SolverBase):
    """quasi_newton/bfgs.rs"""
    METHOD = A.BFGS


class DFP(_SolverBase):
    """quasi_newton/dfp.rs"""
    METHOD = A.DFP


class GradientDescent(_SolverBase):
    """steepest_descent/gradient_descent.rs (config-1 plumbing)"""
    METHOD = A.GRADIENT_DESCENT

    def has_converged(self, eval_x_k):
        g = eval_x_k.g() if isinstance(eval_x_k, FuncEvalMultivariate) else eval_x_k[1]
        return float(np.max(np.abs(g))) < self.tol()


class _BoundedBase(_SolverBase):
    """bfgs_b.rs / dfp_b.rs / sr1_b.rs: `new(tol, x0, lower_bound, upper_bound)`; x0 is projected, directions are P(x - Hg) - x."""

    def __init__(self, tol, x0, lower_bound, upper_bound, ctx=None):
        super().__init__(tol, x0, ctx)
        self._lb, self._ub = _f64(lower_bound), _f64(upper_bound)
        _check(A.lib().qn_solver_set_bounds(self.h, _dp(self._lb), _dp(self._ub)))

    @classmethod
    def new(cls, tol, x0, lower_bound, upper_bound, ctx=None):
        return cls(tol, x0, lower_bound, upper_bound, ctx)

    def lower_bound(self):
        return self._lb

    def upper_bound(self):
        return self._ub

    def projected_gradient(self, eval_x_k):  # ls_solver.rs:121-133
        g = np.array(eval_x_k.g() if isinstance(eval_x_k, FuncEvalMultivariate) else eval_x_k[1], dtype=np.float64)
        x = self.x()
        g[((x == self._lb) & (g > 0)) | ((x == self._ub) & (g < 0))] = 0.0
        return g


class BFGSB(_BoundedBase):
    METHOD = A.BFGS


class DFPB(_BoundedBase):
    METHOD = A.DFP


class SR1B(_BoundedBase):
    METHOD = A.SR1


class Newton(_SolverBase):
    """newton/mod.rs: direction from a dense factorisation of the Hessian on the GPU."""
    METHOD = A.NEWTON

    def decrement_squared(self):
        return self._opt(A.lib().qn_solver_decrement_squared)

    def has_converged(self, eval_x_k=None):  # newton/mod.rs:64-69
        d = self.decrement_squared()
        return d is not None and d * 0.5 < self.tol()


# ---- kernel-level FFI helpers (tests of the individual primitives) ----
class DeviceBuffer:
    def __init__(self, ctx, host_array):
        a = _f64(host_array)
        self.ctx, self.shape, self.nbytes = ctx, a.shape, a.nbytes
        self.p = C.c_void_p()
        _check(A.lib().qn_dev_alloc(ctx.h, max(a.nbytes, 8), C.byref(self.p)))
        if a.nbytes:
            _check(A.lib().qn_h2d(ctx.h, self.p, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def get(self):
        out = np.empty(self.shape)
        if self.nbytes:
            _check(A.lib().qn_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.p, self.nbytes))
        return out

    def free(self):
        if self.p:
            A.lib().qn_dev_free(self.ctx.h, self.p)
            self.p = None


def gemv(ctx, a_dev, ld, nrows, ncols, x_dev, y_dev):
    _check(A.lib().qn_gemv(ctx.h, a_dev.p, ld, nrows, ncols, x_dev.p, y_dev.p))


def rank2_update(ctx, h_dev, ld, row0, nrows, n, s_dev, u_dev, c_ss, c_su, c_uu):
    _check(A.lib().qn_rank2_update(ctx.h, h_dev.p, ld, row0, nrows, n, s_dev.p, u_dev.p, c_ss, c_su, c_uu))


def axpy(ctx, n, x_dev, t, d_dev, out_dev):
    _check(A.lib().qn_axpy(ctx.h, n, x_dev.p, t, d_dev.p, out_dev.p))


def dot(ctx, n, a_dev, b_dev):
    v = C.c_double()
    _check(A.lib().qn_dot(ctx.h, n, a_dev.p, b_dev.p, C.byref(v)))
    return v.value


def nrm2(ctx, n, a_dev):
    v = C.c_double()
    _check(A.lib().qn_nrm2(ctx.h, n, a_dev.p, C.byref(v)))
    return v.value
