"""optimization-solvers_amd: MI355X-native quasi-Newton / line-search inner loop behind the reference's
Solver::minimize() surface.  The directory name is not a Python identifier; load it with
`__graft_entry__.load_package()` (registers it as `optimization_solvers_amd`)."""
from . import _abi, dist  # noqa: F401
from .facade import OptimizationResult, OptimizationSolver  # noqa: F401
from .solver import (BFGS, BFGSB, DFP, DFPB, SR1B, BackTrackingB, MoreThuenteB, AbnormalTermination, BackTracking, Context, DeviceBuffer, DeviceClosure, ErrorInputParams,  # noqa: F401
                     FuncEvalMultivariate, GradientDescent, LogSumExp, MaxIterReached, MoreThuente, Newton, Objective, OutOfDomain, Quadratic,
                     SolverError, axpy, default_context, dot, gemv, nrm2, partition, rank2_update)
