// issue_probe.hip -- how fast does a SIMD issue the evaluation kernel's row-loop mix with 1, 2, 4 waves on it?
// One workgroup per CU of W waves (64 W threads: W / 4 waves per SIMD); every wave runs the same loop of independent groups of the row loop's
// instructions -- a ds_read_b128 from LDS, two v_readlane into SGPRs, four f64 FMAs with those SGPRs -- ITER times, no global memory in the loop.
// Reported: cycles (s_memtime at 100 MHz x the shader clock is not known here, so: nanoseconds) per group per wave, and per group per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o issue_probe.bin issue_probe.hip && ./issue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int W>
__global__ __launch_bounds__(64 * W) void mix_kernel(unsigned long long* out, double* sink, int iters) {
    __shared__ v2d rows[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < 16; r += W) rows[r][lane] = (v2d){1.0 + lane, 2.0 + r};
    __syncthreads();
    double xr = 1.0 + lane, cx = 0.0, cy = 0.0, t0 = 0.0, t1 = 0.0;
    const v2d xt = {0.5, 0.25};
    const unsigned long long a = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const v2d hv = rows[r][lane];
            const int lo = __builtin_amdgcn_readlane(__double2loint(xr), r), hi = __builtin_amdgcn_readlane(__double2hiint(xr), r);
            const double xi = __hiloint2double(hi, lo);
            t0 = __builtin_fma(hv.x, xt.x, t0);
            t1 = __builtin_fma(hv.y, xt.y, t1);
            cx = __builtin_fma(hv.x, xi, cx);
            cy = __builtin_fma(hv.y, xi, cy);
        }
        asm volatile("" : "+v"(xr));
    }
    const unsigned long long b = wall_clock64();
    if (lane == 0) out[blockIdx.x * W + wave] = b - a;
    if (cx + cy + t0 + t1 == 12345.678) sink[0] = cx;
}

template <int W>
static int run(unsigned long long* d, double* sink, int iters) {
    const int G = 256;
    std::vector<unsigned long long> h(G * W);
    hipLaunchKernelGGL(mix_kernel<W>, dim3(G), dim3(64 * W), 0, 0, d, sink, iters);
    hipLaunchKernelGGL(mix_kernel<W>, dim3(G), dim3(64 * W), 0, 0, d, sink, iters);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h.data(), d, G * W * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double ns = 10.0 * (double)h[h.size() / 2];
    const double groups = 16.0 * iters;
    printf("%2d waves per workgroup (%d per SIMD): %8.0f ns per wave for %d row groups = %6.2f ns per group per wave, %6.2f ns per group per SIMD (7 instructions per group)\n",
           W, W / 4 ? W / 4 : 1, ns, (int)groups, ns / groups, ns / groups / (W >= 4 ? W / 4.0 : 1.0));
    return 0;
}
int main() {
    unsigned long long* d = nullptr; double* sink = nullptr;
    CHK(hipMalloc(&d, 256 * 16 * 8)); CHK(hipMalloc(&sink, 64));
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) {
        if (run<4>(d, sink, iters)) return 1;
        if (run<8>(d, sink, iters)) return 1;
        if (run<12>(d, sink, iters)) return 1;
        if (run<16>(d, sink, iters)) return 1;
    }
    return 0;
}
