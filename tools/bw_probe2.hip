// bw_probe2.hip -- in-place vs out-of-place read+write streaming at 2 GiB (calibration for the H pass)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
template <int R, bool NT>
__global__ __launch_bounds__(256) void rw_rows(const double* __restrict__ A, double* __restrict__ B, size_t n, double s) {
    const size_t rb = (size_t)blockIdx.x * R; const int tid = threadIdx.x;
    for (size_t c = 0; c < n / 512; ++c) {
        const size_t j = c * 512 + 2 * tid;
        v2d h[R];
#pragma unroll
        for (int r = 0; r < R; ++r) h[r] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(A + (rb + r) * n + j)) : *reinterpret_cast<const v2d*>(A + (rb + r) * n + j);
#pragma unroll
        for (int r = 0; r < R; ++r) { h[r].x += s; h[r].y += s;
            if (NT) __builtin_nontemporal_store(h[r], reinterpret_cast<v2d*>(B + (rb + r) * n + j)); else *reinterpret_cast<v2d*>(B + (rb + r) * n + j) = h[r]; }
    }
}
// flat copy: each block a contiguous 1 MiB span
template <bool NT>
__global__ __launch_bounds__(256) void rw_flat(const double* __restrict__ A, double* __restrict__ B, size_t elems, double s) {
    const size_t per = 131072; // doubles per block (1 MiB)
    const size_t base = (size_t)blockIdx.x * per;
    for (size_t o = 0; o < per; o += 512 * 4) {
        v2d h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = *reinterpret_cast<const v2d*>(A + base + o + r * 512 + 2 * threadIdx.x);
#pragma unroll
        for (int r = 0; r < 4; ++r) { h[r].x += s; h[r].y += s;
            if (NT) __builtin_nontemporal_store(h[r], reinterpret_cast<v2d*>(B + base + o + r * 512 + 2 * threadIdx.x)); else *reinterpret_cast<v2d*>(B + base + o + r * 512 + 2 * threadIdx.x) = h[r]; }
    }
}
int main() {
    const size_t n = 16384; const size_t elems = n * n; // 2 GiB
    double *A, *B; hipMalloc(&A, elems * 8); hipMalloc(&B, elems * 8); hipMemset(A, 0, elems * 8); hipMemset(B, 0, elems * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto launch, const char* name) {
        launch(); hipDeviceSynchronize();
        float best = 1e9, tot = 0;
        for (int i = 0; i < 8; ++i) { hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); tot += ms; if (ms < best) best = ms; }
        printf("%-40s avg %.3f ms -> %.0f GB/s (best %.0f)\n", name, tot / 8, elems * 16.0 / (tot / 8 * 1e-3) / 1e9, elems * 16.0 / (best * 1e-3) / 1e9);
    };
    run([&] { hipLaunchKernelGGL((rw_rows<4, false>), dim3(n / 4), dim3(256), 0, 0, A, A, n, 1.0); }, "rows R=4 in place");
    run([&] { hipLaunchKernelGGL((rw_rows<4, false>), dim3(n / 4), dim3(256), 0, 0, A, B, n, 1.0); }, "rows R=4 out of place");
    run([&] { hipLaunchKernelGGL((rw_rows<8, false>), dim3(n / 8), dim3(256), 0, 0, A, A, n, 1.0); }, "rows R=8 in place");
    run([&] { hipLaunchKernelGGL((rw_rows<8, false>), dim3(n / 8), dim3(256), 0, 0, A, B, n, 1.0); }, "rows R=8 out of place");
    run([&] { hipLaunchKernelGGL((rw_rows<8, true>), dim3(n / 8), dim3(256), 0, 0, A, A, n, 1.0); }, "rows R=8 in place NT");
    run([&] { hipLaunchKernelGGL((rw_rows<8, true>), dim3(n / 8), dim3(256), 0, 0, A, B, n, 1.0); }, "rows R=8 out of place NT");
    run([&] { hipLaunchKernelGGL((rw_flat<false>), dim3(elems / 131072), dim3(256), 0, 0, A, A, elems, 1.0); }, "flat 1MiB/block in place");
    run([&] { hipLaunchKernelGGL((rw_flat<false>), dim3(elems / 131072), dim3(256), 0, 0, A, B, elems, 1.0); }, "flat 1MiB/block out of place");
    run([&] { hipLaunchKernelGGL((rw_flat<true>), dim3(elems / 131072), dim3(256), 0, 0, A, B, elems, 1.0); }, "flat 1MiB/block out of place NT");
    run([&] { hipMemcpyAsync(B, A, elems * 8, hipMemcpyDeviceToDevice, 0); }, "hipMemcpy D2D");
    return 0;
}
