// mfma_f64_probe.hip -- sustained v_mfma_f64_16x16x4_f64 rate of the device (calibration for the Newton roofline).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe.bin mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void probe(double* out, int iters, double a0, double b0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
static void run(int blocks, int iters, const char* label) {
    double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /*waves*/ * iters * NACC * 2048.0;
    printf("%-40s %8.3f ms  %7.2f TFLOP/s\n", label, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<4>(256 * 1, 20000, "1 WG/CU (1 wave/SIMD), 4 acc chains");
    run<4>(256 * 2, 20000, "2 WG/CU (2 waves/SIMD), 4 acc chains");
    run<4>(256 * 4, 10000, "4 WG/CU (4 waves/SIMD), 4 acc chains");
    run<8>(256 * 4, 5000, "4 WG/CU (4 waves/SIMD), 8 acc chains");
    run<1>(256 * 4, 40000, "4 WG/CU (4 waves/SIMD), 1 acc chain");
    return 0;
}
