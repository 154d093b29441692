// seam_probe.hip -- what does an all-to-all seam between two streaming phases cost on this chip, at the solver's n = 4096 shape?
//
// The benchmark iteration is three streaming phases (two evaluations over the symmetric half of Q, one read+write pass over
// the half of H), each followed by a grid-wide dependency (a handful of sums every workgroup needs before it can go on).
// Today every seam is a kernel boundary.  This probe times the same phase body -- 256 workgroups of 512 threads, one per CU,
// each streaming two 128 x 128 f64 tiles through a 16-row register window, a per-workgroup scalar out, all 256 scalars summed
// by every workgroup of the next phase -- in four arrangements:
//   0  one launch per phase (the prologue sums the scalars)                        [what qn_sym2.hip.h does]
//   1  ONE launch, an XCD-hierarchical grid barrier per seam                        [MI355X_MICROARCH.md, barrier-xcd]
//   2  ... and the next phase's first tile requested BEFORE the barrier (the matrix does not depend on the seam)
//   3  as 2, phases that stream nothing (the bare seam)
// `--rw k` makes every k-th phase write its tiles back (the H pass leaves dirty lines for the seam's release to write back).
// Every arrangement computes the same chain of scalars (each phase scales its vector by the previous phase's sum), so equal
// final values = the hand-off was seen whole by every workgroup; the program prints them.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o seam_probe.bin seam_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <algorithm>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int TPB = 512, WAVES = 8, RPW = 16, TB = 128, G = 256, NT = 32; // NT tiles per side
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct GridBar { // every word that is polled or added to sits on a 128-byte line of its own
    unsigned census_total[32];
    unsigned census_xcc[8][32];
    unsigned cnt_xcc[8][32];
    unsigned top[32];
    unsigned gen[8][32];
    unsigned abort_flag[32];
};

struct Args {
    const double* Q;
    double* H;
    const double* x;   // n
    double* wgS;       // [2][G] per-workgroup scalars, double-buffered on the phase parity
    double* slots;     // [G][2][TB] one row slot per item (what a tile leaves behind for a later reduce)
    GridBar* bar;
    double* result;    // [G] the scale every workgroup ended with
    int n, nphases, rw_every, stream;
};

__device__ __forceinline__ v2d ld2(const double* p) { return *reinterpret_cast<const v2d*>(p); }
__device__ __forceinline__ void st2(double* p, v2d v) { *reinterpret_cast<v2d*>(p) = v; }
__device__ __forceinline__ unsigned ald(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned aadd(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7u; } // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// the item list of workgroup g: tiles g and g + 256 of the first 512 (row-major over a 32 x 32 tile grid)
__device__ __forceinline__ const double* tile_ptr(const double* M, int n, int g, int it, int wave, int lane) {
    const int t = g + it * G, I = t / NT, J = t % NT;
    return M + (size_t)(I * TB + wave * RPW) * n + (size_t)J * TB + 2 * lane;
}

__device__ __forceinline__ double lane_bcast(const double v, const int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

struct BarState { unsigned epoch, my_xcc, n_in_xcc, n_xccs; };

// first meeting of the grid: who is here, and on which XCD.  Returns false when not every workgroup is resident (bounded wait).
__device__ bool grid_census(GridBar* b, BarState& S, unsigned* lds_word) {
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        aadd(&b->census_xcc[x][0], 1u);
        aadd(&b->census_total[0], 1u);
        unsigned ok = 1, spins = 0;
        while (ald(&b->census_total[0]) < (unsigned)gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { ok = 0; break; }
        }
        unsigned nx = 0, mine = 0;
        if (ok) {
            for (unsigned k = 0; k < 8; ++k) { const unsigned c = ald(&b->census_xcc[k][0]); nx += c != 0; if (k == x) mine = c; }
        }
        lds_word[0] = ok; lds_word[1] = x; lds_word[2] = mine; lds_word[3] = nx;
    }
    __syncthreads();
    S.epoch = 0; S.my_xcc = lds_word[1]; S.n_in_xcc = lds_word[2]; S.n_xccs = lds_word[3];
    const bool ok = lds_word[0] != 0;
    __syncthreads();
    return ok;
}

// XCD-hierarchical barrier.  Every storing wave has waited for its stores (the caller's s_waitcnt) before this is entered.
// thread 0: arrive on the XCD's counter; the XCD's last arriver writes the XCD's L2 back (agent release), arrives on the top
// counter, waits for the other XCDs' leaders and opens its XCD's generation word; everybody else polls that word (an L2 hit in
// the own XCD).  Then an agent acquire (L1 invalidate) per workgroup.
template <bool FENCES>
__device__ bool grid_barrier(GridBar* b, BarState& S, unsigned* lds_word) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned e = ++S.epoch;
        unsigned ok = 1, spins = 0;
        const unsigned old = aadd(&b->cnt_xcc[S.my_xcc][0], 1u);
        if (old == e * S.n_in_xcc - 1u) {
            if (FENCES) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            aadd(&b->top[0], 1u);
            while (ald(&b->top[0]) < e * S.n_xccs) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { ok = 0; break; }
            }
            __hip_atomic_store(&b->gen[S.my_xcc][0], ok ? e : 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned g;
            while ((g = ald(&b->gen[S.my_xcc][0])) < e) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { ok = 0; break; }
            }
            if (g == 0xffffffffu) ok = 0;
        }
        if (FENCES) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        lds_word[0] = ok;
    } else {
        S.epoch++;
    }
    __syncthreads();
    const bool ok = lds_word[0] != 0;
    return ok;
}

// sum of the 256 per-workgroup scalars of the previous phase: every workgroup, fixed order (wave 0, 4 per lane, butterfly)
__device__ __forceinline__ double sum_scalars(const double* wgS, double* lds_d) {
    if (threadIdx.x < 64) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < G / 64; ++j) a += wgS[j * 64 + threadIdx.x];
        a = wave_sum(a);
        if (threadIdx.x == 0) lds_d[0] = a;
    }
    __syncthreads();
    const double s = lds_d[0];
    __syncthreads();
    return s;
}

// one streaming phase over this workgroup's two tiles; h[] holds item 0's rows on entry when PRELOADED.
// Returns with h[] holding the NEXT phase's item 0 when PREFETCH (requested behind this phase's stores).
template <bool PRELOADED, bool PREFETCH>
__device__ __forceinline__ void phase_body(const Args& a, const int phase, const double scale, v2d (&h)[RPW], double* lds_d, double (*colred)[TB]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x, n = a.n;
    const bool rw = a.rw_every > 0 && (phase % a.rw_every) == a.rw_every - 1;
    const double* M = a.Q + (rw ? (a.H - a.Q) : (ptrdiff_t)0); // (offset arithmetic keeps the global address space: a select of two pointers became flat loads)
    if (!PRELOADED) {
        const double* p0 = tile_ptr(M, n, g, 0, wave, lane);
#pragma unroll
        for (int r = 0; r < RPW; ++r) h[r] = ld2(p0 + (size_t)r * n);
    }
    double wsum = 0.0;
    for (int it = 0; it < 2; ++it) {
        const int t = g + it * G, I = t / NT, J = t % NT;
        const v2d xj = ld2(a.x + J * TB + 2 * lane);
        const double xr = a.x[I * TB + wave * RPW + (lane & 15)];
        const double* pn = tile_ptr(M, n, g, 1, wave, lane);
        double* pw = const_cast<double*>(tile_ptr(a.H, n, g, it, wave, lane));
        const bool refill = it == 0;
        double cx = 0.0, cy = 0.0, acc = 0.0;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            v2d hv = h[r];
            if (refill) h[r] = ld2(pn + (size_t)r * n);
            const double xi = lane_bcast(xr, r) * scale;
            double t0 = hv.x * (xj.x * scale);
            t0 = __builtin_fma(hv.y, xj.y * scale, t0);
            acc = __builtin_fma(xi, t0, acc);
            cx = __builtin_fma(hv.x, xi, cx);
            cy = __builtin_fma(hv.y, xi, cy);
            if (rw) { hv.x = hv.x + 1e-30 * xi; hv.y = hv.y + 1e-30 * xi; st2(pw + (size_t)r * n, hv); }
        }
        colred[wave][2 * lane] = cx; colred[wave][2 * lane + 1] = cy;
        acc = wave_sum(acc);
        if (lane == 0) lds_d[8 + wave] = acc;
        __syncthreads();
        if (tid < TB) {
            double s = colred[0][tid];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) s += colred[w][tid];
            a.slots[((size_t)g * 2 + it) * TB + tid] = s;
        }
        if (tid == 0) { double s = 0.0; for (int w = 0; w < WAVES; ++w) s += lds_d[8 + w]; wsum += s; }
        __syncthreads();
    }
    if (tid == 0) a.wgS[(size_t)(phase & 1) * G + g] = 1.0 + 1e-3 * wsum / (1.0 + fabs(wsum)); // O(1), depends on every byte streamed
    if (PREFETCH) { // the next phase's first tile, requested behind this phase's stores: vmcnt(16) below then covers the stores only
        asm volatile("" ::: "memory");
        const bool rwn = a.rw_every > 0 && ((phase + 1) % a.rw_every) == a.rw_every - 1;
        const double* p0 = tile_ptr(a.Q + (rwn ? (a.H - a.Q) : (ptrdiff_t)0), n, g, 0, wave, lane);
#pragma unroll
        for (int r = 0; r < RPW; ++r) h[r] = ld2(p0 + (size_t)r * n);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// arrangement 0: one launch per phase
__global__ __launch_bounds__(TPB, 2) void phase_kernel(const Args a, const int phase) {
    __shared__ double lds_d[16];
    __shared__ double colred[WAVES][TB];
    v2d h[RPW];
    { // the window first, then the scalars (the prologue of the real kernels)
        const double* p0 = tile_ptr(a.Q + ((a.rw_every > 0 && (phase % a.rw_every) == a.rw_every - 1) ? (a.H - a.Q) : (ptrdiff_t)0), a.n, blockIdx.x, 0, threadIdx.x >> 6, threadIdx.x & 63);
        if (a.stream) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) h[r] = ld2(p0 + (size_t)r * a.n);
        }
    }
    const double scale = phase == 0 ? 1.0 : sum_scalars(a.wgS + (size_t)((phase - 1) & 1) * G, lds_d) / G;
    if (a.stream) phase_body<true, false>(a, phase, scale, h, lds_d, colred);
    else if (threadIdx.x == 0) a.wgS[(size_t)(phase & 1) * G + blockIdx.x] = scale;
    if (phase == a.nphases - 1 && threadIdx.x == 0) a.result[blockIdx.x] = scale;
}

// arrangements 1-3: one launch
template <bool PREFETCH, bool FENCES>
__global__ __launch_bounds__(TPB, 2) void persistent_kernel(const Args a) {
    __shared__ double lds_d[16];
    __shared__ double colred[WAVES][TB];
    __shared__ unsigned lds_w[4];
    BarState S;
    v2d h[RPW];
    if (!grid_census(a.bar, S, lds_w)) { if (threadIdx.x == 0) a.result[blockIdx.x] = -1.0; return; }
    if (PREFETCH && a.stream) {
        const double* p0 = tile_ptr(a.Q + ((a.rw_every == 1) ? (a.H - a.Q) : (ptrdiff_t)0), a.n, blockIdx.x, 0, threadIdx.x >> 6, threadIdx.x & 63);
#pragma unroll
        for (int r = 0; r < RPW; ++r) h[r] = ld2(p0 + (size_t)r * a.n);
    }
    double scale = 1.0;
    for (int phase = 0; phase < a.nphases; ++phase) {
        if (phase > 0) scale = sum_scalars(a.wgS + (size_t)((phase - 1) & 1) * G, lds_d) / G;
        if (a.stream) phase_body<PREFETCH, PREFETCH>(a, phase, scale, h, lds_d, colred);
        else { if (threadIdx.x == 0) a.wgS[(size_t)(phase & 1) * G + blockIdx.x] = scale; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (!grid_barrier<FENCES>(a.bar, S, lds_w)) { if (threadIdx.x == 0) a.result[blockIdx.x] = -2.0; return; }
    }
    if (threadIdx.x == 0) a.result[blockIdx.x] = scale;
}

int main(int argc, char** argv) {
    int nphases = 300, reps = 5;
    for (int i = 1; i < argc; ++i) { if (!strcmp(argv[i], "--phases")) nphases = atoi(argv[++i]); if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]); }
    const int n = 4096;
    const size_t elems = (size_t)n * n;
    double *Q, *H, *x, *wgS, *slots, *result; GridBar* bar;
    CHK(hipMalloc(&Q, elems * 8)); CHK(hipMalloc(&H, elems * 8)); CHK(hipMalloc(&x, n * 8));
    CHK(hipMalloc(&wgS, 2 * G * 8)); CHK(hipMalloc(&slots, (size_t)G * 2 * TB * 8)); CHK(hipMalloc(&result, G * 8)); CHK(hipMalloc(&bar, sizeof(GridBar)));
    {
        std::vector<double> hq(elems), hx(n);
        unsigned long long s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
        for (size_t i = 0; i < elems; ++i) hq[i] = rnd() / n;
        for (int i = 0; i < n; ++i) hx[i] = rnd();
        CHK(hipMemcpy(Q, hq.data(), elems * 8, hipMemcpyHostToDevice));
        CHK(hipMemcpy(H, hq.data(), elems * 8, hipMemcpyHostToDevice));
        CHK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice));
    }
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    int ncu = 0; { hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0)); ncu = p.multiProcessorCount; }
    printf("device CUs %d, grid %d x %d threads, %d phases per run, median of %d runs\n", ncu, G, TPB, nphases, reps);
    if (ncu < G) { printf("fewer CUs than workgroups: the persistent arrangements would not be co-resident; stopping\n"); return 0; }
    for (int rw_every : {0, 3}) {
        for (int stream : {1, 0}) {
            for (int mode = 0; mode < 4; ++mode) {
                if (!stream && (mode == 2)) continue;
                if (stream && mode == 3) continue;
                Args a{Q, H, x, wgS, slots, bar, result, n, nphases, rw_every, stream};
                std::vector<float> ms(reps);
                double res0 = 0.0; bool same = true;
                for (int rep = 0; rep < reps; ++rep) {
                    CHK(hipMemsetAsync(bar, 0, sizeof(GridBar), st));
                    CHK(hipMemsetAsync(wgS, 0, 2 * G * 8, st));
                    CHK(hipStreamSynchronize(st));
                    CHK(hipEventRecord(e0, st));
                    if (mode == 0) { for (int p = 0; p < nphases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(TPB), 0, st, a, p); }
                    else if (mode == 1) hipLaunchKernelGGL((persistent_kernel<false, true>), dim3(G), dim3(TPB), 0, st, a);
                    else if (mode == 2) hipLaunchKernelGGL((persistent_kernel<true, true>), dim3(G), dim3(TPB), 0, st, a);
                    else hipLaunchKernelGGL((persistent_kernel<false, false>), dim3(G), dim3(TPB), 0, st, a);
                    CHK(hipEventRecord(e1, st));
                    CHK(hipEventSynchronize(e1));
                    CHK(hipGetLastError());
                    CHK(hipEventElapsedTime(&ms[rep], e0, e1));
                    std::vector<double> hr(G);
                    CHK(hipMemcpy(hr.data(), result, G * 8, hipMemcpyDeviceToHost));
                    if (rep == 0) res0 = hr[0];
                    for (int g = 0; g < G; ++g) same = same && hr[g] == res0;
                }
                std::sort(ms.begin(), ms.end());
                const char* names[4] = {"launch per phase", "one launch, barrier-xcd", "one launch, barrier + prefetch", "one launch, barrier without fences (INVALID hand-off, floor)"};
                printf("rw_every=%d stream=%d  %-62s %8.2f us/phase  (%.2f us per 3 phases)  result %.17g %s\n", rw_every, stream, names[mode],
                       1e3 * ms[reps / 2] / nphases, 3e3 * ms[reps / 2] / nphases, res0, same ? "all workgroups equal" : "WORKGROUPS DIFFER");
            }
        }
    }
    return 0;
}
