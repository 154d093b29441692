// qn_newton.hip.h -- row f2: Newton direction on the GPU (src/newton/mod.rs:26-49).
//
// The reference inverts the Hessian (`try_inverse`, LU) and multiplies: d = -H^-1 g, decrement^2 = (H^-1 d).d.
// Here the (symmetric positive definite) Hessian is factorised once per iteration, H = L L', by a blocked
// right-looking Cholesky in f64 and both quantities come from triangular solves:
//   chol_diag_kernel   64 x 64 diagonal block, one workgroup, in LDS
//   chol_trsm_kernel   panel below the diagonal block: rows * L_kk^-T, 64 rows per workgroup through LDS
//   chol_syrk_kernel   trailing update C -= P P' on the lower triangle with v_mfma_f64_16x16x4_f64
//                      (64 x 64 tile per workgroup, both panels staged k-major in LDS: conflict-free fragment reads)
//   tri_*_kernel       blocked forward / backward substitution for the vector right-hand sides
// A non-positive or non-finite pivot marks the factorisation failed; the solver then falls back to d = -g
// exactly as the reference does for a singular Hessian (newton/mod.rs:43-46).  An indefinite but invertible
// Hessian is treated as singular here (the reference's LU would still invert it): convex problems only.
// n <= 5 uses newton_small_kernel: the reference's arithmetic order (closed forms for n <= 2), one thread.
#pragma once

#define QN_NB 64 // Cholesky block size

typedef double v4d __attribute__((ext_vector_type(4)));

// 64 x 64 tile  global -> LDS (row stride QN_NB + 1), T threads, 8 independent loads in flight per thread
template <int T, bool TRANSPOSE>
__device__ __forceinline__ void qn_tile_to_lds(double (*dst)[QN_NB + 1], const double* __restrict__ src, size_t ld) {
    const int tid = threadIdx.x;
    for (int e0 = 0; e0 < QN_NB * QN_NB; e0 += 8 * T) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + u * T + tid; v[u] = src[(size_t)(e / QN_NB) * ld + (e % QN_NB)]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * T + tid;
            if (TRANSPOSE) dst[e % QN_NB][e / QN_NB] = v[u]; else dst[e / QN_NB][e % QN_NB] = v[u];
        }
    }
}

// ---- diagonal block: W[k0:k0+NB, k0:k0+NB] = L L' (lower), in place; rows/cols >= n_valid are identity padding ----
__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ W, size_t ld, int k0, int* __restrict__ fail) {
    __shared__ double a[QN_NB][QN_NB + 1];
    const int tid = threadIdx.x;
    qn_tile_to_lds<256, false>(a, W + (size_t)k0 * ld + k0, ld);
    __syncthreads();
    for (int j = 0; j < QN_NB; ++j) {
        const double piv = a[j][j];
        if (!(piv > 0.0) || !isfinite(piv)) { // uniform: every thread reads the same LDS word
            if (tid == 0) *fail = 1;
            return;
        }
        const double d = sqrt(piv);
        __syncthreads();
        if (tid == 0) a[j][j] = d;
        for (int i = j + 1 + tid; i < QN_NB; i += 256) a[i][j] = a[i][j] / d;
        __syncthreads();
        // rank-1 update of the remaining lower triangle: a[i][c] -= a[i][j] * a[c][j], j < c <= i
        const int m = QN_NB - j - 1;
        for (int e = tid; e < m * m; e += 256) {
            const int i = j + 1 + e / m, c = j + 1 + e % m;
            if (c <= i) a[i][c] = a[i][c] - a[i][j] * a[c][j];
        }
        __syncthreads();
    }
    for (int e = tid; e < QN_NB * QN_NB; e += 256) {
        const int i = e / QN_NB, j = e % QN_NB;
        if (j <= i) W[(size_t)(k0 + i) * ld + k0 + j] = a[i][j];
    }
}

// ---- panel: rows r >= k0+NB: W[r, k0:k0+NB] <- W[r, k0:k0+NB] * L_kk^-T ; 64 rows per workgroup (64 threads) ----
__global__ __launch_bounds__(64) void chol_trsm_kernel(double* __restrict__ W, size_t ld, int k0, int nrows_total, const int* __restrict__ fail) {
    if (*fail) return;
    __shared__ double L[QN_NB][QN_NB + 1];
    __shared__ double xt[QN_NB][QN_NB + 1]; // xt[j][r]: column j of the panel tile, row r
    const int tid = threadIdx.x;
    const int r0 = k0 + QN_NB + blockIdx.x * QN_NB;
    qn_tile_to_lds<64, false>(L, W + (size_t)k0 * ld + k0, ld);
    qn_tile_to_lds<64, true>(xt, W + (size_t)r0 * ld + k0, ld); // r0 + 63 < nrows_total: the padded dimension is a multiple of 64
    __syncthreads();
    // thread = row: forward substitution against L' (x L' = a  <=>  x_j = (a_j - sum_{p<j} x_p L[j][p]) / L[j][j])
    for (int j = 0; j < QN_NB; ++j) {
        double acc = xt[j][tid];
        for (int p = 0; p < j; ++p) acc = acc - xt[p][tid] * L[j][p];
        xt[j][tid] = acc / L[j][j];
    }
    __syncthreads();
    for (int e = tid; e < QN_NB * QN_NB; e += 64) {
        const int i = e / QN_NB, j = e % QN_NB;
        const int r = r0 + i;
        if (r < nrows_total) W[(size_t)r * ld + k0 + j] = xt[j][i];
    }
}

// ---- trailing update on the lower triangle: C[ti, tj] -= P_ti * P_tj'  (tiles of 64, ti >= tj), f64 MFMA ----
// grid (nt, nt); 256 threads = 4 waves, wave w owns the 32 x 32 quadrant (w >> 1, w & 1): 2 x 2 MFMA tiles of 16 x 16.
__global__ __launch_bounds__(256) void chol_syrk_kernel(double* __restrict__ W, size_t ld, int k0, const int* __restrict__ fail) {
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    if (*fail) return;
    __shared__ double PI[QN_NB][QN_NB + 1]; // PI[k][i] = P[i0 + i][k0 + k]  (k-major: lanes read consecutive i; +1 pad for the transposing stores)
    __shared__ double PJ[QN_NB][QN_NB + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = k0 + QN_NB + ti * QN_NB, j0 = k0 + QN_NB + tj * QN_NB;
    qn_tile_to_lds<256, true>(PI, W + (size_t)i0 * ld + k0, ld); // coalesced along k in global memory
    qn_tile_to_lds<256, true>(PJ, W + (size_t)j0 * ld + k0, ld);
    __syncthreads();
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll 4
    for (int kk = 0; kk < QN_NB; kk += 4) {
        // A operand: A[i = lane&15][k = lane>>4] ; B operand: B[k = lane>>4][j = lane&15] = P_j[j][k]
        const double a0 = PI[kk + l4][wi + l15], a1 = PI[kk + l4][wi + 16 + l15];
        const double b0 = PJ[kk + l4][wj + l15], b1 = PJ[kk + l4][wj + 16 + l15];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = i0 + wi + a * 16 + l4 + 4 * reg;
                const int col = j0 + wj + b * 16 + l15;
                double* p = W + (size_t)row * ld + col;
                *p = *p - acc[a][b][reg];
            }
}

// ---- triangular solves with vector right-hand sides (in place on x, length n_pad) ----
// forward, block k: x_k <- L_kk^-1 x_k (one wave), then x_i -= L[i, k-block] x_k for the rows below (one wave per row)
__global__ __launch_bounds__(64) void tri_fwd_diag_kernel(const double* __restrict__ W, size_t ld, int k0, double* __restrict__ x) {
    __shared__ double L[QN_NB][QN_NB + 1];
    __shared__ double xs[QN_NB];
    const int tid = threadIdx.x;
    qn_tile_to_lds<64, false>(L, W + (size_t)k0 * ld + k0, ld);
    xs[tid] = x[k0 + tid];
    __syncthreads();
    for (int j = 0; j < QN_NB; ++j) {
        if (tid == j) xs[j] = xs[j] / L[j][j];
        __syncthreads();
        if (tid > j) xs[tid] = xs[tid] - L[tid][j] * xs[j];
        __syncthreads();
    }
    x[k0 + tid] = xs[tid];
}
__global__ __launch_bounds__(256) void tri_fwd_update_kernel(const double* __restrict__ W, size_t ld, int k0, int n_pad, double* __restrict__ x) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double xk = x[k0 + lane];
    for (int r = k0 + QN_NB + blockIdx.x * 4 + wave; r < n_pad; r += gridDim.x * 4) {
        double p = W[(size_t)r * ld + k0 + lane] * xk;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
        if (lane == 0) x[r] = x[r] - p;
    }
}
// backward (L' z = x), block k from the end: z_k <- L_kk^-T x_k, then x_j -= sum_{i in block k} L[i][j] z_i for j < k0
__global__ __launch_bounds__(64) void tri_bwd_diag_kernel(const double* __restrict__ W, size_t ld, int k0, double* __restrict__ x) {
    __shared__ double L[QN_NB][QN_NB + 1];
    __shared__ double xs[QN_NB];
    const int tid = threadIdx.x;
    qn_tile_to_lds<64, false>(L, W + (size_t)k0 * ld + k0, ld);
    xs[tid] = x[k0 + tid];
    __syncthreads();
    for (int j = QN_NB - 1; j >= 0; --j) {
        if (tid == j) xs[j] = xs[j] / L[j][j];
        __syncthreads();
        if (tid < j) xs[tid] = xs[tid] - L[j][tid] * xs[j]; // (L')[tid][j] = L[j][tid]
        __syncthreads();
    }
    x[k0 + tid] = xs[tid];
}
__global__ __launch_bounds__(256) void tri_bwd_update_kernel(const double* __restrict__ W, size_t ld, int k0, double* __restrict__ x) {
    __shared__ double zk[QN_NB];
    if (threadIdx.x < QN_NB) zk[threadIdx.x] = x[k0 + threadIdx.x];
    __syncthreads();
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < k0; j += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int i = 0; i < QN_NB; ++i) acc = __builtin_fma(W[(size_t)(k0 + i) * ld + j], zk[i], acc);
        x[j] = x[j] - acc;
    }
}

// Hessian staging: W (row-major, ld = n_pad64) <- src rows (ld_src), identity on the padding diagonal
__global__ void newton_stage_kernel(double* __restrict__ W, size_t ld, int n, int n_pad64, const double* __restrict__ src, size_t ld_src) {
    const size_t total = (size_t)n_pad64 * ld;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / ld, j = e % ld;
        double v = 0.0;
        if (i < (size_t)n && j < (size_t)n) v = src[i * ld_src + j];
        else if (i == j) v = 1.0;
        W[e] = v;
    }
}
// rhs staging and result: x64 <- sign * src (zero padded); after the solves dst <- sign2 * x64
__global__ void newton_vec_kernel(double* __restrict__ dst, const double* __restrict__ src, int n_src, int n_dst, double sign) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_dst; i += gridDim.x * blockDim.x) dst[i] = (i < n_src) ? sign * src[i] : 0.0;
}

// ---- n <= 5: the reference's arithmetic order, one thread ----
__global__ void newton_small_kernel(const double* __restrict__ Hrow, size_t ld, int n, const double* __restrict__ g, double* __restrict__ d,
                                    double* __restrict__ z, int* __restrict__ fail) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double a[QN_SMALL_N * QN_SMALL_N], inv[QN_SMALL_N * QN_SMALL_N]; // column-major
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) a[i + j * n] = Hrow[(size_t)i * ld + j];
    int ok = 1;
    if (n == 1) {
        if (a[0] == 0.0) ok = 0; else inv[0] = 1.0 / a[0];
    } else if (n == 2) { // [nalgebra] try_inverse 2x2: by the determinant
        const double m11 = a[0], m21 = a[1], m12 = a[2], m22 = a[3];
        const double det = m11 * m22 - m21 * m12;
        if (det == 0.0) ok = 0;
        else { inv[0] = m22 / det; inv[2] = -m12 / det; inv[1] = -m21 / det; inv[3] = m11 / det; }
    } else { // LU with partial pivoting, then the identity columns
        double lu[QN_SMALL_N * QN_SMALL_N];
        int piv[QN_SMALL_N];
        for (int e = 0; e < n * n; ++e) lu[e] = a[e];
        for (int k = 0; k < n && ok; ++k) {
            int p = k;
            double best = fabs(lu[k + k * n]);
            for (int i = k + 1; i < n; ++i) if (fabs(lu[i + k * n]) > best) { best = fabs(lu[i + k * n]); p = i; }
            piv[k] = p;
            if (best == 0.0) { ok = 0; break; }
            if (p != k) for (int j = 0; j < n; ++j) { const double t = lu[k + j * n]; lu[k + j * n] = lu[p + j * n]; lu[p + j * n] = t; }
            const double dd = lu[k + k * n];
            for (int i = k + 1; i < n; ++i) lu[i + k * n] /= dd;
            for (int j = k + 1; j < n; ++j) { const double ukj = lu[k + j * n]; for (int i = k + 1; i < n; ++i) lu[i + j * n] -= lu[i + k * n] * ukj; }
        }
        if (ok) for (int c = 0; c < n; ++c) {
            double* x = inv + c * n;
            for (int i = 0; i < n; ++i) x[i] = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < n; ++k) if (piv[k] != k) { const double t = x[k]; x[k] = x[piv[k]]; x[piv[k]] = t; }
            for (int k = 0; k < n; ++k) for (int i = k + 1; i < n; ++i) x[i] -= lu[i + k * n] * x[k];
            for (int kk = n - 1; kk >= 0; --kk) { x[kk] /= lu[kk + kk * n]; for (int i = 0; i < kk; ++i) x[i] -= lu[i + kk * n] * x[kk]; }
        }
    }
    if (!ok) { *fail = 1; return; }
    // d = -(inv g), z = inv d : column sweeps (newton/mod.rs:38-40)
    double y[QN_SMALL_N];
    for (int i = 0; i < n; ++i) y[i] = inv[i] * g[0];
    for (int j = 1; j < n; ++j) for (int i = 0; i < n; ++i) y[i] = inv[i + j * n] * g[j] + y[i];
    for (int i = 0; i < n; ++i) d[i] = -y[i];
    for (int i = 0; i < n; ++i) y[i] = inv[i] * d[0];
    for (int j = 1; j < n; ++j) for (int i = 0; i < n; ++i) y[i] = inv[i + j * n] * d[j] + y[i];
    for (int i = 0; i < n; ++i) z[i] = y[i];
}
