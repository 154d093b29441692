// device_closure.hip -- a user-side DEVICE closure for `qn_oracle.kind = QN_ORACLE_DEVICE_FN` (include/qn_hip.h): what a caller
// whose objective already lives on the GPU hands to qn_minimize instead of the reference's host closure
// `impl FnMut(&DVector<f64>) -> FuncEvalMultivariate` (ls_solver.rs:69).  The library never copies x, f or g to the host on this
// path: it calls `double_well_chain_eval` on ITS stream with device pointers and the closure enqueues its own kernel there.
//
// Objective (non-quadratic, non-convex -- a chain of double wells):
//     f(x) = sum_i 1/4 (x_i^2 - a_i)^2  +  c/2 sum_{i<n-1} (x_{i+1} - x_i)^2
//     g_i  = x_i (x_i^2 - a_i) + c (2 x_i - x_{i-1} - x_{i+1})        (one-sided at the two ends)
// Built into examples/libdevice_closure.so by __graft_entry__.build(); driven by tests/test_gpu_device_closure.py, which
// compares the run against the CPU oracle driven by the same formula in numpy.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct double_well_chain {
    double* a_dev;
    double c;
    size_t n;
    unsigned long long calls; // host-side count of closure invocations (what the boxed Rust closure would count)
};

// One workgroup: a fixed-order reduction, so f does not depend on the launch geometry or on scheduling.
__global__ __launch_bounds__(1024) void double_well_chain_kernel(const double* __restrict__ x, const double* __restrict__ a,
                                                                 double c, size_t n, double* __restrict__ f,
                                                                 double* __restrict__ g) {
    __shared__ double red[1024];
    double acc = 0.0;
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        const double xi = x[i];
        const double w = xi * xi - a[i];
        double gi = xi * w;
        acc += 0.25 * w * w;
        if (i + 1 < n) {
            const double d = x[i + 1] - xi;
            acc += 0.5 * c * d * d;
            gi -= c * d;
        }
        if (i > 0) gi += c * (xi - x[i - 1]);
        g[i] = gi;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *f = red[0];
}

extern "C" {

void* double_well_chain_create(size_t n, const double* a_host, double c) {
    double_well_chain* o = new double_well_chain{nullptr, c, n, 0};
    if (hipMalloc(&o->a_dev, n * sizeof(double)) != hipSuccess ||
        hipMemcpy(o->a_dev, a_host, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        delete o;
        return nullptr;
    }
    return o;
}

void double_well_chain_destroy(void* user) {
    double_well_chain* o = (double_well_chain*)user;
    if (!o) return;
    (void)hipFree(o->a_dev);
    delete o;
}

unsigned long long double_well_chain_calls(void* user) { return ((double_well_chain*)user)->calls; }

// qn_device_oracle_fn (include/qn_hip.h): enqueue on `stream`, read x_dev[0..n), write *f_dev and g_dev[0..n); no host sync.
int double_well_chain_eval(void* user, void* stream, const double* x_dev, size_t n, double* f_dev, double* g_dev) {
    double_well_chain* o = (double_well_chain*)user;
    if (!o || n != o->n) return 1;
    o->calls++;
    hipLaunchKernelGGL(double_well_chain_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x_dev, o->a_dev, o->c, n, f_dev, g_dev);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

} // extern "C"
