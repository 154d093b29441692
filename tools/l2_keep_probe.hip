// Probe (round 6): does an XCD's 4 MB L2 keep a workgroup's tile from one launch to the next?  The evaluation kernel at n = 4096 streams the
// same two 128 KB tiles and 8 KB sliver per workgroup in every launch, two launches back to back per iteration (More-Thuente: E = 2), and the
// counters say every byte crosses the fabric each time (FETCH_SIZE = 1.07 x the algorithmic bytes).  32 workgroups per XCD x 264 KB = 8.25 MB go
// through a 4 MB L2 in the same order each launch, so a least-recently-used L2 never hits even if it survives the boundary.  Questions:
//   (1) footprints of 2 MB and 4 MB per XCD re-read by the next launch (same workgroup -> same bytes): L2 rate, or the fabric's again?
//       control: the same launches over four rotating matrices (nothing to re-use);
//   (2) the kernel's real shape -- two tiles per workgroup -- with the SECOND tile as non-temporal loads: does the first stay?
// 256 workgroups of 512 threads, a wave holds 16 rows x 16 B per lane in flight (the tile kernels' register window); tile (I, J) = 128 row
// segments of 1 KB at a 32 KB stride.  Launch-to-launch time over back-to-back launches.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/l2_keep_probe.bin tools/l2_keep_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

// item k of workgroup b = tile k * G + b; rows_per_wave rows of it per wave (16: the whole tile, 8: its even half); bit k of ntmask: non-temporal
template <int ROWS, int NTMASK>
__global__ __launch_bounds__(512, 1) void read_items(const double* __restrict__ M, double* __restrict__ out, int nb, int items) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t np = (size_t)nb * 128;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k >= items) break;
        const int t = k * gridDim.x + blockIdx.x;
        const int I = t / nb, J = t % nb;
        const double* base = M + (size_t)(I * 128 + wave * 16) * np + (size_t)J * 128 + 2 * lane;
        v2d h[ROWS];
        if ((NTMASK >> k) & 1) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) h[r] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(base + (size_t)r * np));
        } else {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) h[r] = *reinterpret_cast<const v2d*>(base + (size_t)r * np);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc += h[r].x + h[r].y;
    }
    if (acc == 12345.678) out[blockIdx.x * 512 + tid] = acc; // (keeps the loads alive)
}

int main() {
    const int nb = 32, G = 256;
    const size_t n = (size_t)nb * 128;
    double *M[4], *out;
    for (int i = 0; i < 4; ++i) { CHECK(hipMalloc((void**)&M[i], n * n * 8)); CHECK(hipMemset(M[i], 0, n * n * 8)); }
    CHECK(hipMalloc((void**)&out, (size_t)G * 512 * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int reps = 400;
    struct Case { const char* name; int rows, items, ntmask, nbuf; };
    const Case cases[] = {
        {"half tile  (2 MB per XCD), one matrix  ", 8, 1, 0, 1},
        {"half tile  (2 MB per XCD), four rotating", 8, 1, 0, 4},
        {"one tile   (4 MB per XCD), one matrix  ", 16, 1, 0, 1},
        {"one tile   (4 MB per XCD), four rotating", 16, 1, 0, 4},
        {"two tiles  (8 MB per XCD), one matrix  ", 16, 2, 0, 1},
        {"two tiles, second non-temporal, one matrix", 16, 2, 2, 1},
        {"two tiles, first non-temporal, one matrix ", 16, 2, 1, 1},
        {"two tiles, both non-temporal, one matrix  ", 16, 2, 3, 1},
        {"two tiles  (8 MB per XCD), four rotating", 16, 2, 0, 4},
        {"two tiles, second non-temporal, four rotating", 16, 2, 2, 4},
        {"half + half tile, second non-temporal, one matrix", 8, 2, 2, 1},
        {"half + half tile, one matrix", 8, 2, 0, 1},
    };
    for (const Case& c : cases) {
        for (int pass = 0; pass < 2; ++pass) { // pass 0: warm-up
            const int cnt = pass ? reps : 20;
            CHECK(hipEventRecord(a, 0));
            for (int w = 0; w < cnt; ++w) {
#define QN_L(R, N) hipLaunchKernelGGL((read_items<R, N>), dim3(G), dim3(512), 0, 0, M[w % c.nbuf], out, nb, c.items)
                if (c.rows == 16) { if (c.ntmask == 0) QN_L(16, 0); else if (c.ntmask == 1) QN_L(16, 1); else if (c.ntmask == 2) QN_L(16, 2); else QN_L(16, 3); }
                else { if (c.ntmask == 0) QN_L(8, 0); else if (c.ntmask == 1) QN_L(8, 1); else if (c.ntmask == 2) QN_L(8, 2); else QN_L(8, 3); }
            }
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
            if (!pass) continue;
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            const double us = 1e3 * ms / cnt, bytes = (double)G * c.items * c.rows * 8 * 1024;
            printf("%-52s %6.1f MB  %6.2f us launch to launch  %5.2f TB/s\n", c.name, bytes / 1e6, us, bytes / us / 1e6);
        }
    }
    return 0;
}
