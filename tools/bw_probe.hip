// bw_probe.hip -- calibration: what a plain streaming kernel reaches on this GPU at the solver's sizes.
// Build: hipcc --offload-arch=gfx950 -O3 -o bw_probe bw_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int R>
__global__ __launch_bounds__(256) void read_rows(const double* __restrict__ A, int n, double* out) {
    const int rb = blockIdx.x * R, tid = threadIdx.x;
    double acc = 0.0;
    for (int c = 0; c < n / 512; ++c) {
        const int j = c * 512 + 2 * tid;
        v2d h[R];
#pragma unroll
        for (int r = 0; r < R; ++r) h[r] = *reinterpret_cast<const v2d*>(A + (size_t)(rb + r) * n + j);
#pragma unroll
        for (int r = 0; r < R; ++r) acc += h[r].x + h[r].y;
    }
    if (acc == 12345.678) out[blockIdx.x] = acc;
}
template <int R>
__global__ __launch_bounds__(256) void rw_rows(double* __restrict__ A, int n, double s) {
    const int rb = blockIdx.x * R, tid = threadIdx.x;
    for (int c = 0; c < n / 512; ++c) {
        const int j = c * 512 + 2 * tid;
        v2d h[R];
#pragma unroll
        for (int r = 0; r < R; ++r) h[r] = *reinterpret_cast<const v2d*>(A + (size_t)(rb + r) * n + j);
#pragma unroll
        for (int r = 0; r < R; ++r) { h[r].x += s; h[r].y += s; *reinterpret_cast<v2d*>(A + (size_t)(rb + r) * n + j) = h[r]; }
    }
}
__global__ void noop(int* p) { if (p[0] == 12345) p[1] = 1; }

int main() {
    const int n = 4096;
    const size_t elems = (size_t)n * n;
    const int NB = 8; // 8 x 128 MiB = 1 GiB cycled: defeats the 256 MiB Infinity Cache
    double* bufs[NB];
    for (int i = 0; i < NB; ++i) { CHK(hipMalloc(&bufs[i], elems * 8)); CHK(hipMemset(bufs[i], 0, elems * 8)); }
    double* out; CHK(hipMalloc(&out, 1 << 20));
    int* flag; CHK(hipMalloc(&flag, 64)); CHK(hipMemset(flag, 0, 64));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipStream_t st; CHK(hipStreamCreate(&st));
    auto time = [&](auto launch, int reps, double bytes, const char* name) {
        for (int i = 0; i < 3; ++i) launch(i);
        hipStreamSynchronize(st);
        // per-launch timing with events around each launch
        double tot = 0; float best = 1e9;
        for (int i = 0; i < reps; ++i) {
            hipEventRecord(a, st); launch(i); hipEventRecord(b, st); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); tot += ms; if (ms < best) best = ms;
        }
        printf("%-28s avg %.2f us  best %.2f us  -> %.0f GB/s avg, %.0f GB/s best\n", name, 1e3 * tot / reps, 1e3 * best,
               bytes / (tot / reps * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9);
    };
    time([&](int i) { hipLaunchKernelGGL(noop, dim3(1), dim3(64), 0, st, flag); }, 50, 0, "noop 1 block");
    time([&](int i) { hipLaunchKernelGGL(noop, dim3(1024), dim3(256), 0, st, flag); }, 50, 0, "noop 1024 blocks");
    time([&](int i) { hipLaunchKernelGGL(read_rows<4>, dim3(n / 4), dim3(256), 0, st, bufs[i % NB], n, out); }, 64, elems * 8.0, "read R=4 (cycled 1GiB)");
    time([&](int i) { hipLaunchKernelGGL(read_rows<8>, dim3(n / 8), dim3(256), 0, st, bufs[i % NB], n, out); }, 64, elems * 8.0, "read R=8 (cycled 1GiB)");
    time([&](int i) { hipLaunchKernelGGL(read_rows<4>, dim3(n / 4), dim3(256), 0, st, bufs[0], n, out); }, 64, elems * 8.0, "read R=4 (same 128MiB)");
    time([&](int i) { hipLaunchKernelGGL(rw_rows<4>, dim3(n / 4), dim3(256), 0, st, bufs[i % NB], n, 1.0); }, 64, elems * 16.0, "r+w R=4 (cycled 1GiB)");
    time([&](int i) { hipLaunchKernelGGL(rw_rows<8>, dim3(n / 8), dim3(256), 0, st, bufs[i % NB], n, 1.0); }, 64, elems * 16.0, "r+w R=8 (cycled 1GiB)");
    time([&](int i) { hipLaunchKernelGGL(rw_rows<4>, dim3(n / 4), dim3(256), 0, st, bufs[0], n, 1.0); }, 64, elems * 16.0, "r+w R=4 (same 128MiB)");
    // the solver's working set: Q read twice, H read+write, 256 MiB total
    time([&](int i) { hipLaunchKernelGGL(read_rows<4>, dim3(n / 4), dim3(256), 0, st, bufs[1], n, out);
                      hipLaunchKernelGGL(read_rows<4>, dim3(n / 4), dim3(256), 0, st, bufs[1], n, out);
                      hipLaunchKernelGGL(rw_rows<4>, dim3(n / 4), dim3(256), 0, st, bufs[0], n, 1.0); }, 64, elems * 32.0, "iteration pattern Q,Q,H");
    return 0;
}
