"""ctypes declarations for libqn_hip.so (include/qn_hip.h).  Loading fails loudly if the HIP library has not
been built: there is no Python or CPU fallback for any compute on this path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QN_HIP_LIB") or os.path.join(_HERE, "lib", "libqn_hip.so")  # QN_HIP_LIB: diagnostic builds

OK, MAX_ITER_REACHED, OUT_OF_DOMAIN, ERROR_INPUT_PARAMS, ABNORMAL_TERMINATION = range(5)
LS_MORETHUENTE, LS_BACKTRACKING, LS_MORETHUENTE_B, LS_BACKTRACKING_B = 0, 1, 2, 3
ORACLE_HOST, ORACLE_DEVICE_FN, ORACLE_OBJECTIVE = 0, 1, 2
BFGS, DFP, GRADIENT_DESCENT, NEWTON, SR1 = 0, 1, 2, 3, 4
UNIQUE_ID_BYTES = 128

dp = C.POINTER(C.c_double)


class LineSearchStruct(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32),
                ("c1", C.c_double), ("c2", C.c_double), ("t_min", C.c_double), ("t_max", C.c_double),
                ("delta_min", C.c_double), ("delta", C.c_double), ("delta_max", C.c_double),
                ("bt_c1", C.c_double), ("bt_beta", C.c_double),
                ("lower_bound_host", C.c_void_p), ("upper_bound_host", C.c_void_p)]


HOST_ORACLE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, C.c_size_t, dp, dp)
DEVICE_ORACLE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)
CALLBACK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
HOST_HESSIAN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, C.c_size_t, dp)
HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, dp, C.c_size_t)


class OracleStruct(C.Structure):
    _fields_ = [("kind", C.c_int32), ("memoize", C.c_int32),
                ("host_fn", C.c_void_p), ("host_user", C.c_void_p),
                ("device_fn", C.c_void_p), ("device_user", C.c_void_p),
                ("objective", C.c_void_p), ("host_hessian_fn", C.c_void_p)]


class TraceRec(C.Structure):
    _fields_ = [("f", C.c_double), ("gnorm", C.c_double), ("t", C.c_double), ("s_norm", C.c_double), ("y_norm", C.c_double),
                ("n_evals", C.c_int32), ("ls_iters", C.c_int32), ("ls_cases", C.c_int32), ("updated", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("oracle_calls", C.c_uint64), ("oracle_evals", C.c_uint64),
                ("h_passes", C.c_uint64), ("h_bytes", C.c_uint64), ("obj_bytes", C.c_uint64),
                ("launches", C.c_uint64), ("host_syncs", C.c_uint64),
                ("t_hpass_ms", C.c_double), ("t_eval_ms", C.c_double), ("t_ctl_ms", C.c_double), ("t_comm_ms", C.c_double),
                ("n_hpass_timed", C.c_uint64), ("n_eval_timed", C.c_uint64), ("n_ctl_timed", C.c_uint64), ("n_comm_timed", C.c_uint64),
                ("matrix_bytes_per_pass", C.c_uint64),
                ("total_minimize_calls", C.c_uint64), ("total_iterations", C.c_uint64), ("total_oracle_calls", C.c_uint64),
                ("total_oracle_evals", C.c_uint64), ("total_h_passes", C.c_uint64), ("total_h_bytes", C.c_uint64),
                ("total_obj_bytes", C.c_uint64), ("path", C.c_uint32), ("_pad", C.c_uint32),
                ("t_hreduce_ms", C.c_double), ("t_ereduce_ms", C.c_double), ("n_hreduce_timed", C.c_uint64), ("n_ereduce_timed", C.c_uint64),
                ("total_xchg_vector", C.c_uint64), ("total_xchg_scalar", C.c_uint64),
                ("t_newton_ms", C.c_double), ("n_newton_timed", C.c_uint64), ("newton_lu_sync_timeouts", C.c_uint64)]


PATH_FUSED, PATH_SYM, PATH_SYM_GENERIC, PATH_PIPELINED, PATH_SYM2, PATH_TILES1 = 1, 2, 4, 8, 16, 32


# every symbol include/qn_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("qn_status_string", C.c_char_p, [C.c_int]),
    ("qn_last_error_message", C.c_char_p, []),
    ("qn_abi_version", C.c_int, []),
    ("qn_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("qn_context_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("qn_comm_unique_id", C.c_int, [C.c_void_p]),
    ("qn_context_create_sharded", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("qn_context_create_sharded_host_exchange", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("qn_context_destroy", None, [C.c_void_p]),
    ("qn_partition", C.c_int, [C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("qn_comm_selftest", C.c_int, [C.c_void_p]),
    ("qn_context_comm_check", C.c_int, [C.c_void_p]),
    ("qn_context_exchange_probe", C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    ("qn_context_event_bracket_overhead", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    ("qn_context_synchronize", C.c_int, [C.c_void_p]),
    ("qn_context_rank", C.c_int, [C.c_void_p]),
    ("qn_context_world", C.c_int, [C.c_void_p]),
    ("qn_context_stream", C.c_void_p, [C.c_void_p]),
    ("qn_context_set_host_exchange_async", C.c_int, [C.c_void_p, C.c_int]),
    ("qn_context_set_trial_vector_exchange", C.c_int, [C.c_void_p, C.c_int]),
    ("qn_context_set_allreduce", C.c_int, [C.c_void_p, C.c_int]),
    ("qn_morethuente_default", None, [C.POINTER(LineSearchStruct)]),
    ("qn_morethuente_with_deltas", C.c_int, [C.POINTER(LineSearchStruct), C.c_double, C.c_double, C.c_double]),
    ("qn_morethuente_with_t_min", C.c_int, [C.POINTER(LineSearchStruct), C.c_double]),
    ("qn_morethuente_with_t_max", C.c_int, [C.POINTER(LineSearchStruct), C.c_double]),
    ("qn_morethuente_with_c1", C.c_int, [C.POINTER(LineSearchStruct), C.c_double]),
    ("qn_morethuente_with_c2", C.c_int, [C.POINTER(LineSearchStruct), C.c_double]),
    ("qn_backtracking_new", None, [C.POINTER(LineSearchStruct), C.c_double, C.c_double]),
    ("qn_morethuente_b_new", None, [C.POINTER(LineSearchStruct)]),
    ("qn_backtracking_b_new", None, [C.POINTER(LineSearchStruct), C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    ("qn_linesearch_with_lower_bound", None, [C.POINTER(LineSearchStruct), C.c_void_p]),
    ("qn_linesearch_with_upper_bound", None, [C.POINTER(LineSearchStruct), C.c_void_p]),
    ("qn_quadratic_create", C.c_int, [C.c_void_p, C.c_size_t, dp, dp, C.POINTER(C.c_void_p)]),
    ("qn_quadratic_create_synthetic", C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64, dp, dp, C.POINTER(C.c_void_p)]),
    ("qn_logsumexp_create", C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, dp, dp, C.c_double, C.POINTER(C.c_void_p)]),
    ("qn_objective_destroy", None, [C.c_void_p]),
    ("qn_objective_eval", C.c_int, [C.c_void_p, dp, dp, dp]),
    ("qn_objective_get_rows", C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, dp]),
    ("qn_solver_create", C.c_int, [C.c_void_p, C.c_int, C.c_double, dp, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("qn_solver_destroy", None, [C.c_void_p]),
    ("qn_solver_reset", C.c_int, [C.c_void_p, dp]),
    ("qn_solver_set_bounds", C.c_int, [C.c_void_p, dp, dp]),
    ("qn_minimize", C.c_int, [C.c_void_p, C.POINTER(LineSearchStruct), C.POINTER(OracleStruct), C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    ("qn_compute_step_len", C.c_int, [C.c_void_p, C.POINTER(LineSearchStruct), dp, C.c_double, dp, dp, C.c_size_t, C.POINTER(OracleStruct), C.c_size_t,
                             C.POINTER(C.c_double)]),
    ("qn_solver_n", C.c_size_t, [C.c_void_p]),
    ("qn_solver_k", C.c_size_t, [C.c_void_p]),
    ("qn_solver_set_k", C.c_int, [C.c_void_p, C.c_size_t]),
    ("qn_solver_tol", C.c_double, [C.c_void_p]),
    ("qn_solver_get_x", C.c_int, [C.c_void_p, dp]),
    ("qn_solver_set_x", C.c_int, [C.c_void_p, dp]),
    ("qn_solver_s_norm", C.c_int, [C.c_void_p, dp, C.POINTER(C.c_int)]),
    ("qn_solver_y_norm", C.c_int, [C.c_void_p, dp, C.POINTER(C.c_int)]),
    ("qn_solver_next_iterate_too_close", C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    ("qn_solver_gradient_next_iterate_too_close", C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    ("qn_solver_decrement_squared", C.c_int, [C.c_void_p, dp, C.POINTER(C.c_int)]),
    ("qn_solver_get_inv_hessian", C.c_int, [C.c_void_p, dp, C.c_int]),
    ("qn_solver_set_inv_hessian", C.c_int, [C.c_void_p, dp]),
    ("qn_solver_compute_direction", C.c_int, [C.c_void_p, dp, dp]),
    ("qn_solver_secant_update", C.c_int, [C.c_void_p, dp, dp]),
    ("qn_solver_set_trace", C.c_int, [C.c_void_p, C.c_size_t, C.c_int]),
    ("qn_solver_get_trace", C.c_int, [C.c_void_p, C.POINTER(TraceRec), C.c_size_t, C.POINTER(C.c_size_t), dp]),
    ("qn_solver_get_stats", C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    ("qn_solver_set_profiling", C.c_int, [C.c_void_p, C.c_int]),
    ("qn_solver_set_sync_mode", C.c_int, [C.c_void_p, C.c_int]),
    ("qn_solver_set_tiling", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("qn_solver_set_option", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("qn_dev_alloc", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("qn_dev_free", C.c_int, [C.c_void_p, C.c_void_p]),
    ("qn_h2d", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("qn_d2h", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("qn_gemv", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    ("qn_rank2_update", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                  C.c_double, C.c_double, C.c_double]),
    ("qn_axpy", C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    ("qn_dot", C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, dp]),
    ("qn_nrm2", C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, dp]),
]

_lib = None


def lib():
    """Load libqn_hip.so and bind every declared symbol.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for this path.")
    L = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
