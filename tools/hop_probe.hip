// hop_probe.hip -- how long does one workgroup take to see a word another workgroup has just stored?  (round 6: the go / no-go for
// splitting the pivoted LU's role A over the workgroups of one XCD -- every pivot step of the chain would carry one such hop.)
// Two workgroups of a 16-workgroup grid bounce a counter REPS times; the partner sits on the same XCD (workgroup 8: workgroups are
// dealt to the eight XCDs round-robin) or on another one (workgroup 1).  Loads / stores are relaxed atomics at agent scope (sc1) or
// at workgroup scope (sc0: past the CU's vector cache only -- coherent inside one XCD's L2, nothing more).  Every wait is bounded.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hop_probe.bin tools/hop_probe.hip && tools/hop_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REPS 2000
#define SPIN_MAX (1 << 16)

template <int SCOPE> __device__ __forceinline__ int ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE> __device__ __forceinline__ void st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }

// payload: the consumer also needs PAY doubles that travel with the flag (data stored first, s_waitcnt vmcnt(0), then the flag -- or,
// SENT, no flag at all: the words start out as a sentinel and the consumer polls the last word of the record)
template <int SCOPE, int PAY, bool SENT>
__global__ __launch_bounds__(64) void hop_kernel(int* flag, double* rec, int partner, unsigned long long* out, unsigned* where) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b != 0 && b != partner) return;
    const int me = b == 0 ? 0 : 1;
    if (lane == 0) where[me] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7u;
    unsigned long long t0 = 0;
    int bad = 0;
    double acc = 0.0;
    for (int r = 0; r < REPS && !bad; ++r) {
        if (r == 8 && me == 0) t0 = wall_clock64();
        // turn 2 r belongs to workgroup 0, turn 2 r + 1 to the partner
        const int myturn = 2 * r + me;
        // wait for the other side's previous turn (myturn - 1)
        if (myturn > 0) {
            if (SENT) {
                const double* src = rec + (size_t)((myturn - 1) & 1023) * 16;
                int spin = 0;
                double v;
                for (;;) {
                    v = __hip_atomic_load(src + (lane < PAY ? lane : 0), __ATOMIC_RELAXED, SCOPE);
                    const bool ok = v == (double)(myturn - 1);
                    if (__all(ok)) break;
                    if (++spin > SPIN_MAX) { bad = 1; break; }
                }
                acc += v;
            } else {
                int spin = 0;
                while (ld<SCOPE>(flag) < myturn) { if (++spin > SPIN_MAX) { bad = 1; break; } }
                if (PAY) acc += __hip_atomic_load(rec + (size_t)((myturn - 1) & 1023) * 16 + (lane < PAY ? lane : 0), __ATOMIC_RELAXED, SCOPE);
            }
        }
        if (bad) break;
        // my turn: publish
        if (PAY) {
            if (lane < PAY) __hip_atomic_store(rec + (size_t)(myturn & 1023) * 16 + lane, (double)myturn, __ATOMIC_RELAXED, SCOPE);
            if (!SENT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!SENT && lane == 0) st<SCOPE>(flag, myturn + 1);
    }
    if (lane == 0) {
        if (me == 0) { out[0] = wall_clock64() - t0; out[1] = bad; }
        else out[2] = bad;
        out[3 + me] = (unsigned long long)acc;
    }
}

template <int SCOPE, int PAY, bool SENT> static void run(const char* what, int partner) {
    int* flag; double* rec; unsigned long long* out; unsigned* where;
    CHECK(hipMalloc(&flag, 256)); CHECK(hipMalloc(&rec, 1024 * 16 * 8)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&where, 8));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(flag, 0, 256)); CHECK(hipMemset(rec, 0xff, 1024 * 16 * 8)); CHECK(hipMemset(out, 0, 64));
        hop_kernel<SCOPE, PAY, SENT><<<16, 64>>>(flag, rec, partner, out, where);
        CHECK(hipDeviceSynchronize());
    }
    unsigned long long h[5]; unsigned w[2];
    CHECK(hipMemcpy(h, out, 40, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(w, where, 8, hipMemcpyDeviceToHost));
    // wall_clock64: 100 MHz
    printf("%-46s partner %d (XCC %u -> %u): %7.1f ns per hop%s\n", what, partner, w[0], w[1], h[0] * 10.0 / (2.0 * (REPS - 8)), (h[1] || h[2]) ? "  ** a wait expired **" : "");
    CHECK(hipFree(flag)); CHECK(hipFree(rec)); CHECK(hipFree(out)); CHECK(hipFree(where));
}

int main() {
    for (int partner : {8, 1}) {
        run<__HIP_MEMORY_SCOPE_AGENT, 0, false>("agent scope, flag only", partner);
        run<__HIP_MEMORY_SCOPE_AGENT, 8, false>("agent scope, 8 doubles + wait + flag", partner);
        run<__HIP_MEMORY_SCOPE_AGENT, 8, true>("agent scope, 8 doubles, the datum is the flag", partner);
        run<__HIP_MEMORY_SCOPE_AGENT, 40, true>("agent scope, 40 doubles, the datum is the flag", partner);
        if (partner == 8) {
            run<__HIP_MEMORY_SCOPE_WORKGROUP, 0, false>("workgroup scope (sc0), flag only", partner);
            run<__HIP_MEMORY_SCOPE_WORKGROUP, 8, true>("workgroup scope (sc0), 8 doubles, datum = flag", partner);
        }
    }
    return 0;
}
