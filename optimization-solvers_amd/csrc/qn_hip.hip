// qn_hip.hip -- host side of libqn_hip.so: the C ABI of include/qn_hip.h, the request pump that drives the
// device-resident state machine (qn_ctl.h), objectives, and the RCCL exchange for row-sharded runs.
//
// No CPU compute path exists here: every flop of the hot path runs in the kernels of qn_kernels.hip.h.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/qn_hip.h"
#include "qn_kernels.hip.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int status, const std::string& msg) {
    g_err = msg;
    return status;
}
#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess)                                                                              \
            return fail(QN_ABNORMAL_TERMINATION, std::string(#expr) + ": " + hipGetErrorString(_e));       \
    } while (0)
#define QNCHK(expr)                        \
    do {                                   \
        int _s = (expr);                   \
        if (_s != QN_OK) return _s;        \
    } while (0)

extern "C" const char* qn_last_error_message(void) { return g_err.c_str(); }
extern "C" int qn_abi_version(void) { return QN_ABI_VERSION; }
extern "C" const char* qn_status_string(int status) {
    switch (status) { // Display strings of ls_solver.rs:12-19
    case QN_OK: return "Ok";
    case QN_MAX_ITER_REACHED: return "Max iter reached";
    case QN_OUT_OF_DOMAIN: return "Out of domain";
    case QN_ERROR_INPUT_PARAMS: return "Error in input parameters";
    case QN_ABNORMAL_TERMINATION: return "Abnormal termination";
    default: return "Unknown status";
    }
}

// The host side in parts (round 6): textual includes of this one translation unit, in dependency order.
#include "qn_host_context.hip.h"
#include "qn_host_linesearch.hip.h"
#include "qn_host_objective.hip.h"
#include "qn_host_solver.hip.h"
#include "qn_host_launch.hip.h"
#include "qn_host_newton.hip.h"
#include "qn_host_minimize.hip.h"
#include "qn_host_blas.hip.h"
