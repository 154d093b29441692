// Probe: how fast do 256 workgroups read the upper block triangle of a 4096 x 4096 f64 matrix (528 tiles of 128 x 128, two per
// workgroup + 16 slivers' worth ignored) when a tile is (a) 128 row segments of 1 KB at a 32 KB stride (the row-major matrix the
// solver streams today) or (b) one contiguous 128 KB block (tiles stored one after the other)?  Same bytes, same launch shape,
// same register window (16 rows x 16 B per lane in flight per wave); launch-to-launch time over back-to-back launches.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/tile_layout_probe.bin tools/tile_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool CONTIG>
__global__ __launch_bounds__(512, 2) void read_tiles(const double* __restrict__ M, double* __restrict__ out, int nb, int items_per_wg) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t np = (size_t)nb * 128;
    double acc = 0.0;
    for (int k = 0; k < items_per_wg; ++k) {
        const int t = k * gridDim.x + blockIdx.x; // tile index in row-major order over the upper triangle (approximation: t -> (I, J) by rows of nb)
        const int I = t / nb, J = t % nb;
        v2d h[16];
        if (CONTIG) {
            const double* base = M + (size_t)t * 128 * 128 + (size_t)(wave * 16) * 128 + 2 * lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) h[r] = *reinterpret_cast<const v2d*>(base + (size_t)r * 128);
        } else {
            const double* base = M + (size_t)(I * 128 + wave * 16) * np + (size_t)J * 128 + 2 * lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) h[r] = *reinterpret_cast<const v2d*>(base + (size_t)r * np);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += h[r].x + h[r].y;
    }
    if (acc == 12345.678) out[blockIdx.x * 512 + tid] = acc; // (keeps the loads alive)
}

int main() {
    const int nb = 32, G = 256, items = 2; // 512 tiles of 128 KB = 67 MB per launch
    const size_t n = (size_t)nb * 128;
    double *M[4], *out;
    for (int i = 0; i < 4; ++i) { CHECK(hipMalloc((void**)&M[i], n * n * 8)); CHECK(hipMemset(M[i], 0, n * n * 8)); }
    CHECK(hipMalloc((void**)&out, (size_t)G * 512 * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int reps = 200;
    for (int nbuf = 1; nbuf <= 4; nbuf *= 2) { // 1 buffer: re-read (Infinity Cache may hold it); 2, 4: rotating footprints of 134 / 268 MB
        for (int mode = 0; mode < 2; ++mode) {
            for (int w = 0; w < 20; ++w) {
                if (mode) hipLaunchKernelGGL(read_tiles<true>, dim3(G), dim3(512), 0, 0, M[w % nbuf], out, nb, items);
                else hipLaunchKernelGGL(read_tiles<false>, dim3(G), dim3(512), 0, 0, M[w % nbuf], out, nb, items);
            }
            CHECK(hipEventRecord(a, 0));
            for (int w = 0; w < reps; ++w) {
                if (mode) hipLaunchKernelGGL(read_tiles<true>, dim3(G), dim3(512), 0, 0, M[w % nbuf], out, nb, items);
                else hipLaunchKernelGGL(read_tiles<false>, dim3(G), dim3(512), 0, 0, M[w % nbuf], out, nb, items);
            }
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            const double us = 1e3 * ms / reps, bytes = (double)G * items * 128 * 128 * 8;
            printf("%s, %d buffer(s) in rotation: %.2f us per launch (launch to launch), %.2f TB/s on %.1f MB\n",
                   mode ? "contiguous 128 KB tiles      " : "1 KB segments, 32 KB stride  ", nbuf, us, bytes / us / 1e6, bytes / 1e6);
        }
    }
    return 0;
}
