// chol_diag_probe.hip -- where does chol_diag_inv_kernel spend its time?  (s_memrealtime stamps, 100 MHz)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DQN_DIAG_STAMPS -o chol_diag_probe chol_diag_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define QN_SMALL_N 5
__device__ long long qn_diag_stamps[16];
typedef double v2d __attribute__((ext_vector_type(2)));
#include "../optimization-solvers_amd/csrc/qn_newton.hip.h"
int main() {
    const int n = 64;
    std::vector<double> h(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) h[i * n + j] = (i == j ? 70.0 : 0.0) + 1.0 / (1.0 + abs(i - j));
    double *W, *inv; int* fail;
    hipMalloc(&W, n * n * 8); hipMalloc(&inv, n * n * 8); hipMalloc(&fail, 4);
    hipMemset(fail, 0, 4);
    for (int rep = 0; rep < 5; ++rep) {
        hipMemcpy(W, h.data(), n * n * 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(chol_diag_inv_kernel, dim3(1), dim3(256), 0, 0, W, (size_t)n, 0, inv, fail);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long st[16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(qn_diag_stamps), sizeof(st));
        printf("rep %d: event %.1f us; stamps (us since start):", rep, ms * 1e3);
        for (int i = 1; i < 6; ++i) printf(" %.2f", (st[i] - st[0]) / 100.0);
        printf("\n");
    }
    std::vector<double> L(n * n), X(n * n);
    hipMemcpy(L.data(), W, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(X.data(), inv, n * n * 8, hipMemcpyDeviceToHost);
    double err = 0, err2 = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double s = 0; for (int k = 0; k <= j; ++k) s += L[i * n + k] * L[j * n + k];
        err = fmax(err, fabs(s - h[i * n + j]));
        double t = 0; for (int k = j; k <= i; ++k) t += L[i * n + k] * X[k * n + j];
        err2 = fmax(err2, fabs(t - (i == j ? 1.0 : 0.0)));
    }
    printf("max |LL' - A| = %.3e, max |L invL - I| = %.3e\n", err, err2);
    return 0;
}
