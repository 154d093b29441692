// Probe (round 4): what can the tile kernels' access pattern reach in the HBM regime?  256 (or 512) workgroups stream `items` tiles of
// 128 x 128 f64 each out of a matrix that does not fit the Infinity Cache, as the solver's second-generation kernels do (one
// 512-thread workgroup per CU, a 16-row register window per wave refilled row by row), in four variants:
//   layout   : a tile is 128 row segments of 1 KB at the matrix's row stride (today's row-major storage) | one contiguous 128 KB block
//   traffic  : read only (evaluation)                                                                   | read + write back (update pass)
//   occupancy: 1 workgroup per CU, 16 rows in flight per wave                                           | 2 per CU, 8 rows each
// Launch-to-launch time over back-to-back launches on rotating regions (footprint 4 launches' worth).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/stream_shape_probe.bin tools/stream_shape_probe.hip ; run: ./tools/stream_shape_probe.bin [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ v2d ldv(const double* p) {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p)) : *reinterpret_cast<const v2d*>(p);
}
template <bool NT> __device__ __forceinline__ void stv(double* p, v2d v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(p)); else *reinterpret_cast<v2d*>(p) = v;
}

// tile t of the launch's region: row-major: block-row I = t / nbc, block-column J = t % nbc of a matrix with row stride np
template <bool CONTIG, bool WRITE, int ROWS, bool NT>
__global__ __launch_bounds__(512, 2) void stream_tiles(double* __restrict__ M, double* __restrict__ out, size_t np, int nbc, int tile0, int items) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PASSES = 16 / ROWS; // a wave owns 16 rows of a tile, ROWS of them in flight at a time
    double acc = 0.0;
    auto rowptr = [&](int t, int r) -> double* {
        if (CONTIG) return M + (size_t)t * 128 * 128 + (size_t)(wave * 16 + r) * 128 + 2 * lane;
        const int I = t / nbc, J = t % nbc;
        return M + (size_t)(I * 128 + wave * 16 + r) * np + (size_t)J * 128 + 2 * lane;
    };
    v2d h[ROWS];
    int t = tile0 + blockIdx.x;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) h[r] = ldv<NT>(rowptr(t, r));
    for (int k = 0; k < items; ++k) {
        const int tn = (k + 1 < items) ? t + (int)gridDim.x : t;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                v2d v = h[r];
                // refill with the row that is ROWS rows further on (next pass of this tile, or the first pass of the next tile)
                const int rr = p * ROWS + r + ROWS;
                h[r] = ldv<NT>(rr < 16 ? rowptr(t, rr) : rowptr(tn, rr - 16));
                acc += v.x + v.y;
                if (WRITE) { v.x += 1.0; v.y += 1.0; stv<NT>(rowptr(t, p * ROWS + r), v); }
            }
        }
        t = tn;
    }
    if (acc == 12345.678) out[blockIdx.x * 512 + tid] = acc;
}

template <bool CONTIG, bool WRITE, int ROWS, bool NT>
static int run(const char* name, double* M, double* out, size_t np, int nbc, int G, int items, size_t region_tiles, int nregions) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int reps = 40;
    for (int w = 0; w < 8; ++w)
        hipLaunchKernelGGL((stream_tiles<CONTIG, WRITE, ROWS, NT>), dim3(G), dim3(512), 0, 0, M, out, np, nbc, (int)((w % nregions) * region_tiles), items);
    CHECK(hipEventRecord(a, 0));
    for (int w = 0; w < reps; ++w)
        hipLaunchKernelGGL((stream_tiles<CONTIG, WRITE, ROWS, NT>), dim3(G), dim3(512), 0, 0, M, out, np, nbc, (int)((w % nregions) * region_tiles), items);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = 1e3 * ms / reps, bytes = (double)G * items * 128 * 128 * 8 * (WRITE ? 2 : 1);
    printf("%-78s %8.1f us per launch  %5.2f TB/s on %7.1f MB\n", name, us, bytes / us / 1e6, bytes / 1e6);
    return 0;
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 32768;
    const int nbc = (int)(n / 128);
    const int tiles_per_launch = 4096; // = one rank's share at P = 8, n = 32768 (537 MB)
    const int nregions = 4;
    // rows needed: 4 regions x 4096 tiles / nbc block-columns
    const size_t block_rows = ((size_t)nregions * tiles_per_launch + nbc - 1) / nbc + 1;
    const size_t bytes = block_rows * 128 * n * 8;
    double *M, *out;
    CHECK(hipMalloc((void**)&M, bytes));
    CHECK(hipMemset(M, 0, bytes));
    CHECK(hipMalloc((void**)&out, (size_t)512 * 512 * 8));
    printf("matrix row stride %zu B, %zu block-rows allocated (%.2f GB), %d tiles (%.0f MB) per launch, %d regions in rotation\n", n * 8, block_rows,
           bytes / 1e9, tiles_per_launch, tiles_per_launch * 131072.0 / 1e6, nregions);
#define RUN(C, W, R, NTF, G, name) if (run<C, W, R, NTF>(name, M, out, n, nbc, G, tiles_per_launch / G, tiles_per_launch, nregions)) return 1;
    RUN(false, false, 16, false, 256, "read : row segments, 1 WG/CU x 16 rows");
    RUN(true,  false, 16, false, 256, "read : contiguous tiles, 1 WG/CU x 16 rows");
    RUN(false, false, 8,  false, 512, "read : row segments, 2 WG/CU x 8 rows");
    RUN(true,  false, 8,  false, 512, "read : contiguous tiles, 2 WG/CU x 8 rows");
    RUN(false, false, 16, true,  256, "read : row segments, 1 WG/CU x 16 rows, non-temporal");
    RUN(true,  false, 16, true,  256, "read : contiguous tiles, 1 WG/CU x 16 rows, non-temporal");
    RUN(false, true,  16, false, 256, "r + w: row segments, 1 WG/CU x 16 rows");
    RUN(true,  true,  16, false, 256, "r + w: contiguous tiles, 1 WG/CU x 16 rows");
    RUN(false, true,  16, true,  256, "r + w: row segments, 1 WG/CU x 16 rows, non-temporal");
    RUN(true,  true,  16, true,  256, "r + w: contiguous tiles, 1 WG/CU x 16 rows, non-temporal");
    RUN(false, true,  8,  true,  512, "r + w: row segments, 2 WG/CU x 8 rows, non-temporal");
    RUN(true,  true,  8,  true,  512, "r + w: contiguous tiles, 2 WG/CU x 8 rows, non-temporal");
    return 0;
}
