// Probe: shader clock under FP64 load.  256 workgroups x 512 threads, each wave issues N independent-enough v_fma_f64 (8 chains);
// wall time (100 MHz counter) and s_memtime ticks are recorded: cycles per FMA per wave pair and the effective clock.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/clock_probe.bin tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(512, 2) void fma_loop(double* out, unsigned long long* t, int n) {
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 1e-3 + k;
    const double m = 1.0000001, c = 1e-9;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], m, c);
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    double s = 0; for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 1.2345) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = w1 - w0; t[1] = c1 - c0; }
}
int main() {
    double* out; unsigned long long* t; CHECK(hipMalloc((void**)&out, 8)); CHECK(hipMalloc((void**)&t, 16));
    for (int wgs : {1, 256}) {
        const int n = 20000;
        hipLaunchKernelGGL(fma_loop, dim3(wgs), dim3(512), 0, 0, out, t, n);
        hipLaunchKernelGGL(fma_loop, dim3(wgs), dim3(512), 0, 0, out, t, n);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[2]; CHECK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
        const double ns = h[0] * 10.0, fmas = 8.0 * n;
        printf("%3d workgroup(s): %.1f us wall, %llu s_memtime ticks; %.2f ns per FMA per wave with 2 waves per SIMD (%.2f ns of SIMD time per wave-FMA)\n",
               wgs, ns / 1e3, h[1], ns / fmas, ns / fmas / 2.0);
    }
    return 0;
}
