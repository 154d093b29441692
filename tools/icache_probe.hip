// icache_probe.hip -- is the instruction cache warm when a kernel that ran a moment ago is launched again?
// A kernel whose one wave per workgroup walks N KB of straight-line code (8-byte v_mov with a literal: nothing to wait for but the
// instruction stream) and records how long the walk took (s_memtime); launched (a) cold, (b) again right behind itself, (c) again
// behind ANOTHER kernel of the same size (different code), (d) with a 64 KB memset kernel in between.  If (b) is much faster than
// (a) and (c), code survives a kernel boundary in the instruction cache and a prologue that is the same code in every kernel of an
// iteration could run warm.   hipcc --offload-arch=gfx950 -O3 -o icache_probe.bin icache_probe.hip && ./icache_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define WALK(TAG, NINSTR) \
    asm volatile(".rept " #NINSTR "\n v_mov_b32 %0, 0x12345678\n .endr" : "=v"(v)); \
    asm volatile("" :: "v"(v));

template <int ID>
__global__ __launch_bounds__(64) void walk_kernel(unsigned long long* out) {
    unsigned v = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (ID == 0) { WALK(a, 3072) } else { WALK(b, 3073) }  // 24 KB of code each (different kernels: different addresses)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v == 1) out[0] = 0;
}
__global__ void fill_kernel(double* p, size_t n) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0; }

int main() {
    unsigned long long* d = nullptr; double* big = nullptr;
    const int G = 256;
    CHK(hipMalloc(&d, G * 8)); CHK(hipMalloc(&big, (size_t)64 << 20));
    std::vector<unsigned long long> h(G);
    auto med = [&]() { CHK(hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost)); std::sort(h.begin(), h.end()); printf("median %6llu  min %6llu  max %6llu ticks (100 MHz: x10 ns)\n", h[G / 2], h[0], h[G - 1]); return 0; };
    for (int rep = 0; rep < 3; ++rep) {
        printf("-- repetition %d\n", rep);
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, big, (size_t)8 << 20); CHK(hipDeviceSynchronize());
        printf("A after a 64 MB fill (cold?)        : "); hipLaunchKernelGGL(walk_kernel<0>, dim3(G), dim3(64), 0, 0, d); CHK(hipDeviceSynchronize()); med();
        printf("A again, right behind itself        : "); hipLaunchKernelGGL(walk_kernel<0>, dim3(G), dim3(64), 0, 0, d); hipLaunchKernelGGL(walk_kernel<0>, dim3(G), dim3(64), 0, 0, d); CHK(hipDeviceSynchronize()); med();
        printf("A behind B (other code, same size)  : "); hipLaunchKernelGGL(walk_kernel<1>, dim3(G), dim3(64), 0, 0, d); hipLaunchKernelGGL(walk_kernel<0>, dim3(G), dim3(64), 0, 0, d); CHK(hipDeviceSynchronize()); med();
        printf("A behind A behind a fill, one stream: "); hipLaunchKernelGGL(walk_kernel<0>, dim3(G), dim3(64), 0, 0, d); hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, big, (size_t)8 << 20); hipLaunchKernelGGL(walk_kernel<0>, dim3(G), dim3(64), 0, 0, d); CHK(hipDeviceSynchronize()); med();
    }
    printf("(24 KB = 3072 eight-byte instructions; issue-bound floor of one wave: 3072 x 4 cycles = 12 288 cycles ~ 5-6 us = 500-600 ticks)\n");
    return 0;
}
