// Probe (round 6), fourth part: the update pass writes every tile it reads; its 264 KB per CU go into the XCD's L2 as dirty lines and leave it by
// eviction -- and, for what is still dirty when the kernel ends, by the write-back at the kernel boundary.  Does it matter HOW the stores are issued?
// The iteration's pattern of tools/l2_prefetch_probe.hip (ev, ev flipped, small + touch H, up, small + touch Q) with the update launch's stores as
//   0 plain, 1 non-temporal, 2 written through at system scope (sc0 sc1), 3 sc1 alone, 4 sc0 alone      and its loads plain or non-temporal.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/l2_store_probe.bin tools/l2_store_probe.hip
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

template <int ST>
__device__ __forceinline__ void store_pol(double* p, v2d v) {
    if (ST == 0) *reinterpret_cast<v2d*>(p) = v;
    else if (ST == 1) __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(p));
    else if (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if (ST == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}

template <int ST, int LD, int TAG>
__global__ __launch_bounds__(512, 1) void up(double* __restrict__ H, int nb) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t np = (size_t)nb * 128;
    for (int k = 0; k < 2; ++k) {
        const int t = k * gridDim.x + blockIdx.x;
        const int I = t / nb, J = t % nb;
        double* base = H + (size_t)(I * 128 + wave * 16) * np + (size_t)J * 128 + 2 * lane;
        v2d h[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = LD ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(base + (size_t)r * np)) : *reinterpret_cast<const v2d*>(base + (size_t)r * np);
#pragma unroll
        for (int r = 0; r < 16; ++r) { h[r].x += 1.0; store_pol<ST>(base + (size_t)r * np, h[r]); }
    }
}

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

template <int ST, int LD, int TAG>
static int run_case(const char* name, const double* Q, double* H, double* out, hipEvent_t a, hipEvent_t b) {
    const int nb = 32, G = 256, reps = 300;
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int cnt = pass ? reps : 20;
        CHECK(hipEventRecord(a, 0));
        for (int w = 0; w < cnt; ++w) {
            hipLaunchKernelGGL((ev<false, 2 * TAG>), dim3(G), dim3(512), 0, 0, Q, out, nb);
            hipLaunchKernelGGL((ev<true, 2 * TAG + 1>), dim3(G), dim3(512), 0, 0, Q, out, nb);
            hipLaunchKernelGGL((small<8, 2 * TAG>), dim3(32 + G), dim3(512), 0, 0, H, out, nb, 32, 350);
            hipLaunchKernelGGL((up<ST, LD, TAG>), dim3(G), dim3(512), 0, 0, H, nb);
            hipLaunchKernelGGL((small<8, 2 * TAG + 1>), dim3(64 + G), dim3(512), 0, 0, Q, out, nb, 64, 300);
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
        if (run_case<0, 0, 0>("stores plain, loads plain (today)", Q, H, out, a, b)) return 1;
        if (run_case<1, 0, 1>("stores non-temporal", Q, H, out, a, b)) return 1;
        if (run_case<2, 0, 2>("stores sc0 sc1 (written through, system scope)", Q, H, out, a, b)) return 1;
        if (run_case<3, 0, 3>("stores sc1", Q, H, out, a, b)) return 1;
        if (run_case<4, 0, 4>("stores sc0", Q, H, out, a, b)) return 1;
        if (run_case<0, 1, 5>("stores plain, loads non-temporal", Q, H, out, a, b)) return 1;
        if (run_case<1, 1, 6>("stores and loads non-temporal", Q, H, out, a, b)) return 1;
        if (run_case<3, 1, 7>("stores sc1, loads non-temporal", Q, H, out, a, b)) return 1;
    }
    return 0;
}
