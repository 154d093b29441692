// Probe (round 6), second part: tools/l2_keep_probe.hip showed that an XCD's L2 DOES keep bytes across a kernel boundary (one 128 KB tile per
// workgroup = 4 MB per XCD re-read by the next launch: 3.1 us against 6.5 us from the Infinity Cache) and that two tiles per workgroup in the same
// order every launch never hit (8 MB through a 4 MB least-recently-used cache: 11.1 us either way).  Here: the iteration's pattern -- two
// "evaluation" launches over Q back to back, then an "update" launch that reads and writes H -- with the two evaluations' cache policy varied:
//   KEEPA / KEEPB: the first KEEP rows (of a wave's 16) of tile A / B are plain loads, the rest non-temporal; FLIP: tile B is requested first.
// Every launch of a case is an instantiation of its own (TAG), so `rocprofv3 --kernel-trace --stats` lists each position's duration by name.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/l2_keep_probe2.bin tools/l2_keep_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int KEEP>
__device__ __forceinline__ void load_tile(v2d (&h)[16], const double* base, const size_t np) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r < KEEP) h[r] = *reinterpret_cast<const v2d*>(base + (size_t)r * np);
        else h[r] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(base + (size_t)r * np));
    }
}

template <int KEEPA, int KEEPB, bool FLIP, int TAG>
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
        if (k == 0) load_tile<KEEPA>(h, base, np); else load_tile<KEEPB>(h, base, np);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += h[r].x + h[r].y;
    }
    if (acc == 12345.678) out[blockIdx.x * 512 + tid] = acc;
}

// the update pass's traffic: two tiles per workgroup read and written back
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

template <int KA1, int KB1, bool F1, int KA2, int KB2, bool F2, int TAG>
static int run_case(const char* name, const double* Q, double* H, double* out, hipEvent_t a, hipEvent_t b) {
    const int nb = 32, G = 256, reps = 300;
    float ms[2] = {0, 0};
    for (int with_up = 0; with_up < 2; ++with_up) {
        for (int pass = 0; pass < 2; ++pass) {
            const int cnt = pass ? reps : 20;
            CHECK(hipEventRecord(a, 0));
            for (int w = 0; w < cnt; ++w) {
                hipLaunchKernelGGL((ev<KA1, KB1, F1, 2 * TAG>), dim3(G), dim3(512), 0, 0, Q, out, nb);
                hipLaunchKernelGGL((ev<KA2, KB2, F2, 2 * TAG + 1>), dim3(G), dim3(512), 0, 0, Q, out, nb);
                if (with_up) hipLaunchKernelGGL((up<TAG>), dim3(G), dim3(512), 0, 0, H, nb);
            }
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
            if (pass) CHECK(hipEventElapsedTime(&ms[with_up], a, b));
        }
    }
    printf("%-70s  E1 + E2 alone %6.2f us;  E1 + E2 + update %6.2f us\n", name, 1e3 * ms[0] / reps, 1e3 * ms[1] / reps);
    return 0;
}

int main() {
    const int nb = 32;
    const size_t n = (size_t)nb * 128;
    double *Q, *H, *out;
    CHECK(hipMalloc((void**)&Q, n * n * 8)); CHECK(hipMemset(Q, 0, n * n * 8));
    CHECK(hipMalloc((void**)&H, n * n * 8)); CHECK(hipMemset(H, 0, n * n * 8));
    CHECK(hipMalloc((void**)&out, (size_t)256 * 512 * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int round = 0; round < 2; ++round) { // (twice: the second round starts from a settled Infinity Cache)
        printf("round %d\n", round);
        if (run_case<16, 16, false, 16, 16, false, 0>("plain loads, A then B in both (today)", Q, H, out, a, b)) return 1;
        if (run_case<16, 16, false, 16, 16, true, 1>("plain loads, second launch B then A", Q, H, out, a, b)) return 1;
        if (run_case<0, 0, false, 0, 0, false, 2>("all non-temporal", Q, H, out, a, b)) return 1;
        if (run_case<16, 16, false, 0, 0, false, 3>("first launch plain, second all non-temporal", Q, H, out, a, b)) return 1;
        if (run_case<16, 0, false, 16, 0, false, 4>("A plain, B non-temporal, both launches", Q, H, out, a, b)) return 1;
        if (run_case<12, 0, false, 12, 0, false, 5>("12 rows of A plain, rest non-temporal, both launches", Q, H, out, a, b)) return 1;
        if (run_case<8, 0, false, 8, 0, false, 6>("8 rows of A plain, rest non-temporal, both launches", Q, H, out, a, b)) return 1;
        if (run_case<0, 16, false, 0, 16, true, 7>("A non-temporal, B plain; second launch B first", Q, H, out, a, b)) return 1;
        if (run_case<0, 12, false, 0, 12, true, 8>("A non-temporal, 12 rows of B plain; second launch B first", Q, H, out, a, b)) return 1;
        if (run_case<16, 16, false, 0, 16, true, 9>("first plain A, B; second B plain first, then A non-temporal", Q, H, out, a, b)) return 1;
        if (run_case<0, 16, false, 0, 16, false, 10>("A non-temporal, B plain, same order both launches", Q, H, out, a, b)) return 1;
        if (run_case<0, 12, false, 0, 12, false, 11>("A non-temporal, 12 rows of B plain, same order both launches", Q, H, out, a, b)) return 1;
    }
    return 0;
}
