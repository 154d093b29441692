// Probe (round 6), third part: the iteration's two SMALL launches (accept-reduce: 32 workgroups, update-reduce: 64; ~4-5.5 us each, the chip's fabric
// nearly idle) carrying 256 extra workgroups that only TOUCH the first tile the next tile launch's workgroup of the same index will stream -- an XCD's
// L2 keeps bytes across a kernel boundary (tools/l2_keep_probe.hip), workgroup g of a launch lands on XCD g mod 8, and 32 tiles of 128 KB are the L2's
// 4 MB.  Pattern: ev(A, B), ev(B, A), small [+ touch H's A tiles], up (reads and writes H: A, B), small [+ touch Q's A tiles].
// build: hipcc --offload-arch=gfx950 -O3 -o tools/l2_prefetch_probe.bin tools/l2_prefetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool FLIP, int TAG>
__global__ __launch_bounds__(512, 1) void ev(const double* __restrict__ M, double* __restrict__ out, int nb) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t np = (size_t)nb * 128;
    double acc = 0.0;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = FLIP ? 1 - kk : kk;
        const int t = k * gridDim.x + blockIdx.x;
        const int I = t / nb, J = t % nb;
        const double* base = M + (size_t)(I * 128 + wave * 16) * np + (size_t)J * 128 + 2 * lane;
        v2d h[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = *reinterpret_cast<const v2d*>(base + (size_t)r * np);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += h[r].x + h[r].y;
    }
    if (acc == 12345.678) out[blockIdx.x * 512 + tid] = acc;
}

template <int TAG>
__global__ __launch_bounds__(512, 1) void up(double* __restrict__ H, int nb) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t np = (size_t)nb * 128;
    for (int k = 0; k < 2; ++k) {
        const int t = k * gridDim.x + blockIdx.x;
        const int I = t / nb, J = t % nb;
        double* base = H + (size_t)(I * 128 + wave * 16) * np + (size_t)J * 128 + 2 * lane;
        v2d h[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = *reinterpret_cast<const v2d*>(base + (size_t)r * np);
#pragma unroll
        for (int r = 0; r < 16; ++r) { h[r].x += 1.0; *reinterpret_cast<v2d*>(base + (size_t)r * np) = h[r]; }
    }
}

// the small launch: workgroups 0 .. nsmall - 1 are busy for `ticks` x 10 ns (dependent work on one wave, as the state machine is); workgroups
// nsmall .. nsmall + 255 touch rows [0, 16 * ROWS / 16 ...) of tile (g - nsmall) of M -- ROWS rows per wave
template <int ROWS, int TAG>
__global__ __launch_bounds__(512, 1) void small(const double* __restrict__ M, double* __restrict__ out, int nb, int nsmall, int ticks) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x < nsmall) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
        return;
    }
    const size_t np = (size_t)nb * 128;
    const int t = (int)blockIdx.x - nsmall;
    const int I = t / nb, J = t % nb;
    const double* base = M + (size_t)(I * 128 + wave * 16) * np + (size_t)J * 128 + 2 * lane;
    v2d h[ROWS > 0 ? ROWS : 1];
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) h[r] = *reinterpret_cast<const v2d*>(base + (size_t)r * np);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc += h[r].x + h[r].y;
    if (acc == 12345.678) out[blockIdx.x * 512 + tid] = acc;
}

template <int RH, int RQ, int TAG>
static int run_case(const char* name, const double* Q, double* H, double* out, hipEvent_t a, hipEvent_t b) {
    const int nb = 32, G = 256, reps = 300;
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int cnt = pass ? reps : 20;
        CHECK(hipEventRecord(a, 0));
        for (int w = 0; w < cnt; ++w) {
            hipLaunchKernelGGL((ev<false, 2 * TAG>), dim3(G), dim3(512), 0, 0, Q, out, nb);
            hipLaunchKernelGGL((ev<true, 2 * TAG + 1>), dim3(G), dim3(512), 0, 0, Q, out, nb);
            hipLaunchKernelGGL((small<RH, 2 * TAG>), dim3(32 + (RH ? G : 0)), dim3(512), 0, 0, H, out, nb, 32, 350);
            hipLaunchKernelGGL((up<TAG>), dim3(G), dim3(512), 0, 0, H, nb);
            hipLaunchKernelGGL((small<RQ, 2 * TAG + 1>), dim3(64 + (RQ ? G : 0)), dim3(512), 0, 0, Q, out, nb, 64, 300);
        }
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        if (pass) CHECK(hipEventElapsedTime(&ms, a, b));
    }
    printf("%-64s  %6.2f us per iteration\n", name, 1e3 * ms / reps);
    return 0;
}

int main() {
    const int nb = 32;
    const size_t n = (size_t)nb * 128;
    double *Q, *H, *out;
    CHECK(hipMalloc((void**)&Q, n * n * 8)); CHECK(hipMemset(Q, 0, n * n * 8));
    CHECK(hipMalloc((void**)&H, n * n * 8)); CHECK(hipMemset(H, 0, n * n * 8));
    CHECK(hipMalloc((void**)&out, (size_t)512 * 512 * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int round = 0; round < 2; ++round) {
        printf("round %d\n", round);
        if (run_case<0, 0, 0>("no touching", Q, H, out, a, b)) return 1;
        if (run_case<16, 0, 1>("accept-reduce touches H's first tiles (16 rows per wave)", Q, H, out, a, b)) return 1;
        if (run_case<8, 0, 2>("accept-reduce touches half of them (8 rows per wave)", Q, H, out, a, b)) return 1;
        if (run_case<0, 16, 3>("update-reduce touches Q's first tiles", Q, H, out, a, b)) return 1;
        if (run_case<0, 8, 4>("update-reduce touches half of them", Q, H, out, a, b)) return 1;
        if (run_case<16, 16, 5>("both, whole tiles", Q, H, out, a, b)) return 1;
        if (run_case<12, 12, 6>("both, 12 rows per wave", Q, H, out, a, b)) return 1;
        if (run_case<8, 8, 7>("both, 8 rows per wave", Q, H, out, a, b)) return 1;
    }
    return 0;
}
