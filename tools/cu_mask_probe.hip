// Probe (round 4): does hipExtStreamCreateWithCUMask restrict a stream's kernels on this box, and which CUs does a mask bit name?
// A compute-bound kernel of 2048 workgroups on streams with different masks: its duration tells how many CUs it got; every workgroup
// also records the XCC and CU it ran on (s_getreg HW_ID / XCC_ID), so the set of CUs behind a mask is printed.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/cu_mask_probe.bin tools/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void spin(double* out, unsigned* where, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.000001;
    for (int i = 0; i < iters; ++i) a = a * b + 1e-9;
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); // XCC_ID
        where[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
    }
    if (a == 1234.5) out[blockIdx.x] = a;
}
static int run(const char* name, hipStream_t st, double* out, unsigned* where, unsigned* hwhere) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int G = 2048;
    hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, st, out, where, 20000);
    CHECK(hipEventRecord(a, st));
    hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, st, out, where, 20000);
    CHECK(hipEventRecord(b, st));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipMemcpy(hwhere, where, G * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> cus; int per_xcc[8] = {0};
    for (int i = 0; i < G; ++i) {
        const unsigned hw = hwhere[i] & 0xffff, xcc = hwhere[i] >> 16;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned id = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        if (cus.insert(id).second) per_xcc[xcc & 7]++;
    }
    printf("%-40s %8.3f ms  distinct CUs %3zu  per XCC:", name, ms, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    return 0;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    double* out; unsigned* where; unsigned hwhere[2048];
    CHECK(hipMalloc((void**)&out, 2048 * 8));
    CHECK(hipMalloc((void**)&where, 2048 * 4));
    hipStream_t s0;
    CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    if (run("no mask", s0, out, where, hwhere)) return 1;
    struct { const char* name; int lo, hi, step; } cases[] = {
        {"bits 0..191", 0, 192, 1}, {"bits 0..63", 0, 64, 1}, {"bits 0..31", 0, 32, 1}, {"bits 192..255", 192, 256, 1}, {"every 4th bit off", 0, 256, -4}, {"even bits", 0, 256, 2}};
    for (auto& cs : cases) {
        uint32_t mask[8]; memset(mask, 0, sizeof(mask));
        for (int b = cs.lo; b < cs.hi; ++b) {
            bool on = cs.step > 0 ? ((b - cs.lo) % cs.step == 0) : (b % (-cs.step) != (-cs.step) - 1);
            if (on) mask[b >> 5] |= 1u << (b & 31);
        }
        hipStream_t sm;
        printf("creating a stream with mask '%s'\n", cs.name);
        hipError_t e = hipExtStreamCreateWithCUMask(&sm, 8, mask);
        if (e != hipSuccess) { printf("%-40s hipExtStreamCreateWithCUMask: %s\n", cs.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        if (run(cs.name, sm, out, where, hwhere)) return 1;
        CHECK(hipStreamDestroy(sm));
    }
    return 0;
}
