// sym_probe.hip -- feasibility probe: the H pass on the upper block triangle only (H symmetric), 128 x 128 tiles.
//   tile (I, J), J >= I:  H_IJ += rank-2(s, u);  row part: r_i += sum_j H_ij [y_j, g_j];  column part (J > I): c_j += sum_i H_ij [y_i, g_i]
//   partials P[R][k][rhs][128] (slot k of block-row R), reduced in fixed order by a second kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o sym_probe.bin sym_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
#define TB 128
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int CNT, int OFF>
struct WaveFold {
    template <int V>
    static __device__ __forceinline__ void run(double (&v)[V], int lane) {
        if constexpr (OFF >= 1) {
            if constexpr (CNT > 1) {
                constexpr int HALF = CNT / 2;
                const bool up = (lane & OFF) != 0;
#pragma unroll
                for (int i = 0; i < HALF; ++i) {
                    const double keep = up ? v[i + HALF] : v[i];
                    const double send = up ? v[i] : v[i + HALF];
                    v[i] = keep + __shfl_xor(send, OFF, 64);
                }
                WaveFold<HALF, OFF / 2>::run(v, lane);
            } else {
                v[0] = v[0] + __shfl_xor(v[0], OFF, 64);
                WaveFold<1, OFF / 2>::run(v, lane);
            }
        }
    }
};

__device__ __forceinline__ void tri_tile(int t, int& ti, int& tj, int nb) { // upper triangle, row-major over (I, J >= I)
    // t = I*nb - I(I-1)/2 + (J - I)
    int i = (int)((2.0 * nb + 1.0 - sqrt((2.0 * nb + 1.0) * (2.0 * nb + 1.0) - 8.0 * (double)t)) * 0.5);
    while (i > 0 && i * nb - i * (i - 1) / 2 > t) --i;
    while ((i + 1) * nb - (i + 1) * i / 2 <= t) ++i;
    ti = i; tj = i + (t - (i * nb - i * (i - 1) / 2));
}

// UPDATE: apply the rank-2 update and write back; NRHS right-hand sides
template <bool UPDATE, int NRHS>
__global__ __launch_bounds__(256) void sym_tile_kernel(double* __restrict__ H, int n, int nb, const double* __restrict__ s, const double* __restrict__ u,
                                                       const double* __restrict__ r0, const double* __restrict__ r1, double c_ss, double c_su,
                                                       double* __restrict__ part) {
    __shared__ double rowv[4][TB]; // s, u, rhs0, rhs1 of the tile's rows
    __shared__ double colred[4][NRHS][TB];
    int I, J;
    tri_tile(blockIdx.x, I, J, nb);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = I * TB, j0 = J * TB;
    if (tid < TB) { rowv[0][tid] = s[i0 + tid]; rowv[1][tid] = u[i0 + tid]; rowv[2][tid] = r0[i0 + tid]; rowv[3][tid] = (NRHS == 2) ? r1[i0 + tid] : 0.0; }
    const int jc = j0 + 2 * lane;
    const v2d sj = UPDATE ? *reinterpret_cast<const v2d*>(s + jc) : (v2d){0, 0};
    const v2d uj = UPDATE ? *reinterpret_cast<const v2d*>(u + jc) : (v2d){0, 0};
    const v2d a0 = *reinterpret_cast<const v2d*>(r0 + jc);
    const v2d a1 = (NRHS == 2) ? *reinterpret_cast<const v2d*>(r1 + jc) : (v2d){0, 0};
    __syncthreads();
    double c0x = 0, c0y = 0, c1x = 0, c1y = 0; // column dots of this thread's two columns over the wave's 32 rows
    double* hbase = H + (size_t)(i0 + wave * 32) * n + jc;
    for (int rc = 0; rc < 32; rc += 8) {
        v2d h[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) h[r] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(hbase + (size_t)(rc + r) * n));
        double racc[8 * NRHS];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int ri = wave * 32 + rc + r;
            v2d hn = h[r];
            if (UPDATE) {
                const double si = rowv[0][ri], ui = rowv[1][ri];
                hn.x = hn.x + c_su * (si * uj.x + ui * sj.x);
                hn.y = hn.y + c_su * (si * uj.y + ui * sj.y);
                hn.x = hn.x + c_ss * (si * sj.x);
                hn.y = hn.y + c_ss * (si * sj.y);
                __builtin_nontemporal_store(hn, reinterpret_cast<v2d*>(hbase + (size_t)(rc + r) * n));
            }
            double t0 = hn.x * a0.x; t0 = __builtin_fma(hn.y, a0.y, t0);
            racc[r] = t0;
            const double yi = rowv[2][ri];
            c0x = __builtin_fma(hn.x, yi, c0x); c0y = __builtin_fma(hn.y, yi, c0y);
            if (NRHS == 2) {
                double t1 = hn.x * a1.x; t1 = __builtin_fma(hn.y, a1.y, t1);
                racc[8 + r] = t1;
                const double gi = rowv[3][ri];
                c1x = __builtin_fma(hn.x, gi, c1x); c1y = __builtin_fma(hn.y, gi, c1y);
            }
        }
        WaveFold<8 * NRHS, 32>::run(racc, lane);
        constexpr int SH = (NRHS == 2) ? 2 : 3; // 16 values -> lane>>2 ; 8 values -> lane>>3
        if ((lane & ((1 << SH) - 1)) == 0) {
            const int idx = lane >> SH; // value index: rhs * 8 + r
            const int rhs = idx >> 3, r = idx & 7;
            // row part -> P[I][J][rhs][row]
            part[(((size_t)I * nb + J) * 2 + rhs) * TB + wave * 32 + rc + r] = racc[0];
        }
    }
    if (J > I) { // column part -> P[J][I][rhs][col]
        colred[wave][0][2 * lane] = c0x; colred[wave][0][2 * lane + 1] = c0y;
        if (NRHS == 2) { colred[wave][1][2 * lane] = c1x; colred[wave][1][2 * lane + 1] = c1y; }
        __syncthreads();
        for (int e = tid; e < NRHS * TB; e += 256) {
            const int rhs = e / TB, c = e % TB;
            part[(((size_t)J * nb + I) * 2 + rhs) * TB + c] = ((colred[0][rhs][c] + colred[1][rhs][c]) + colred[2][rhs][c]) + colred[3][rhs][c];
        }
    }
}

template <int NRHS>
__global__ __launch_bounds__(256) void sym_reduce_kernel(const double* __restrict__ part, int nb, double* __restrict__ out0, double* __restrict__ out1) {
    const int R = blockIdx.x, tid = threadIdx.x;
    if (tid >= NRHS * TB) return;
    const int rhs = tid / TB, i = tid % TB;
    double acc = 0.0;
    for (int k0 = 0; k0 < nb; k0 += 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (k0 + q < nb) ? part[(((size_t)R * nb + k0 + q) * 2 + rhs) * TB + i] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = acc + v[q];
    }
    (rhs == 0 ? out0 : out1)[R * TB + i] = acc;
}

// reference: full-matrix row pass (what the product's h_pass does), one wave per row
__global__ void full_ref_kernel(double* H, int n, const double* s, const double* u, const double* r0, const double* r1, double c_ss, double c_su,
                                double* out0, double* out1, int update) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    double a = 0, b = 0;
    for (int j = lane; j < n; j += 64) {
        double h = H[(size_t)row * n + j];
        if (update) { h = h + c_su * (s[row] * u[j] + u[row] * s[j]); h = h + c_ss * (s[row] * s[j]); H[(size_t)row * n + j] = h; }
        a += h * r0[j]; b += h * r1[j];
    }
    for (int off = 32; off; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
    if (lane == 0) { out0[row] = a; out1[row] = b; }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    const int nb = n / TB, ntiles = nb * (nb + 1) / 2;
    const size_t nn = (size_t)n * n;
    std::vector<double> hs(n), hu(n), hy(n), hg(n);
    for (int i = 0; i < n; ++i) { hs[i] = sin(0.1 * i) * 0.01; hu[i] = cos(0.37 * i) * 0.01; hy[i] = sin(0.71 * i + 1); hg[i] = cos(0.13 * i + 2); }
    double *H, *H2, *s, *u, *y, *g, *part, *o0, *o1, *q0, *q1;
    CHK(hipMalloc(&H, nn * 8)); CHK(hipMalloc(&H2, nn * 8));
    CHK(hipMalloc(&s, n * 8)); CHK(hipMalloc(&u, n * 8)); CHK(hipMalloc(&y, n * 8)); CHK(hipMalloc(&g, n * 8));
    CHK(hipMalloc(&part, (size_t)nb * nb * 2 * TB * 8)); CHK(hipMalloc(&o0, n * 8)); CHK(hipMalloc(&o1, n * 8)); CHK(hipMalloc(&q0, n * 8)); CHK(hipMalloc(&q1, n * 8));
    CHK(hipMemcpy(s, hs.data(), n * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(u, hu.data(), n * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(y, hy.data(), n * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(g, hg.data(), n * 8, hipMemcpyHostToDevice));
    { // symmetric start matrix: identity + small symmetric perturbation
        std::vector<double> row(n);
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) { const int a = i < j ? i : j, b = i < j ? j : i; row[j] = (i == j ? 1.0 : 0.0) + 1e-3 * sin(0.001 * a + 0.002 * b); }
            CHK(hipMemcpy(H + (size_t)i * n, row.data(), n * 8, hipMemcpyHostToDevice));
        }
        CHK(hipMemcpy(H2, H, nn * 8, hipMemcpyDeviceToDevice));
    }
    const double c_ss = 0.7, c_su = -0.3;
    // correctness: one update pass with both
    hipLaunchKernelGGL((sym_tile_kernel<true, 2>), dim3(ntiles), dim3(256), 0, 0, H, n, nb, s, u, y, g, c_ss, c_su, part);
    hipLaunchKernelGGL((sym_reduce_kernel<2>), dim3(nb), dim3(256), 0, 0, part, nb, o0, o1);
    hipLaunchKernelGGL(full_ref_kernel, dim3((n + 3) / 4), dim3(256), 0, 0, H2, n, s, u, y, g, c_ss, c_su, q0, q1, 1);
    CHK(hipDeviceSynchronize());
    std::vector<double> a0(n), a1(n), b0(n), b1(n);
    CHK(hipMemcpy(a0.data(), o0, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(a1.data(), o1, n * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(b0.data(), q0, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(b1.data(), q1, n * 8, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, m0 = 0;
    for (int i = 0; i < n; ++i) { e0 = fmax(e0, fabs(a0[i] - b0[i])); e1 = fmax(e1, fabs(a1[i] - b1[i])); m0 = fmax(m0, fabs(b0[i])); }
    printf("n=%d tiles=%d  max |sym - full| rhs0 %.3e rhs1 %.3e (scale %.3e)\n", n, ntiles, e0, e1, m0);
    // timing
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    const int reps = 20;
    float ms;
    for (int variant = 0; variant < 3; ++variant) {
        hipEventRecord(ea);
        for (int r = 0; r < reps; ++r) {
            if (variant == 0) { hipLaunchKernelGGL((sym_tile_kernel<true, 2>), dim3(ntiles), dim3(256), 0, 0, H, n, nb, s, u, y, g, 1e-9, -1e-9, part);
                                hipLaunchKernelGGL((sym_reduce_kernel<2>), dim3(nb), dim3(256), 0, 0, part, nb, o0, o1); }
            if (variant == 1) { hipLaunchKernelGGL((sym_tile_kernel<false, 1>), dim3(ntiles), dim3(256), 0, 0, H2, n, nb, s, u, y, g, 0.0, 0.0, part);
                                hipLaunchKernelGGL((sym_reduce_kernel<1>), dim3(nb), dim3(256), 0, 0, part, nb, o0, o1); }
            if (variant == 2) hipLaunchKernelGGL((sym_tile_kernel<true, 2>), dim3(ntiles), dim3(256), 0, 0, H, n, nb, s, u, y, g, 1e-9, -1e-9, part);
        }
        hipEventRecord(eb); hipEventSynchronize(eb); hipEventElapsedTime(&ms, ea, eb);
        const double half = (double)ntiles * TB * TB * 8.0;
        const double bytes = variant == 1 ? half : 2 * half;
        printf("%-44s %8.2f us per pass  (%6.0f GB/s on the stored half; the full-matrix pass moves %.0f MB)\n",
               variant == 0 ? "update + 2-RHS (tile kernel + reduce)" : variant == 1 ? "read-only 1-RHS (tile kernel + reduce)" : "update + 2-RHS (tile kernel alone)",
               ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9, (variant == 1 ? 1.0 : 2.0) * nn * 8 / 1e6);
    }
    return 0;
}
