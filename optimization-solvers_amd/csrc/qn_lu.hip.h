// qn_lu.hip.h -- Newton direction for Hessians that are NOT symmetric positive definite (src/newton/mod.rs:26-49).
//
// The reference inverts whatever matrix the oracle hands it: `hessian.try_inverse()` is [nalgebra 0.33.2] LU with partial
// (row) pivoting for n > 4, and the direction falls back to -g only when that LU meets an exactly zero pivot
// (`is_invertible`: every diagonal entry of U non-zero).  An indefinite or non-symmetric Hessian is therefore inverted, not
// rejected.  The Cholesky path of qn_newton.hip.h covers the convex case; when it reports a non-positive pivot, or when the
// matrix is not symmetric bit for bit, the solver re-stages the matrix and runs this blocked right-looking LU instead:
//
//   per column k of a 64-column panel:
//     lu_pivot_kernel      one workgroup: first row of maximal |W[i][k]|, i >= k (nalgebra's icamax: strict >, NaNs skipped);
//                          records piv[k], swaps rows k and piv[k] inside the panel, flags an exactly zero pivot column
//     lu_col_step_kernel   rows below k: multiplier l = W[i][k] / W[k][k], W[i][j] -= l W[k][j] for the panel's remaining columns
//   per panel:
//     lu_swap_rows_kernel  replays the panel's 64 row swaps on the columns left and right of it
//     lu_trsm_kernel       U12 = L11^-1 A12 (unit lower triangle from the panel; one thread per column, 64 values in registers)
//     lu_gemm_kernel       A22 -= L21 U12 on the f64 matrix cores (v_mfma_f64_16x16x4_f64, 64 x 64 tile per workgroup, depth 64)
//   solves (vector right-hand sides; d = -(H^-1 g), then z = H^-1 d for the decrement):
//     lu_vec_perm_kernel   x = sign * b[perm]   (perm: the swaps replayed on the identity, built on the host from piv)
//     lu_fwd_step_kernel   unit-lower block substitution: every workgroup solves the 64 x 64 diagonal block itself (one wave,
//                          v_readlane broadcasts), then updates its rows of the running right-hand side
//     lu_bwd_step_kernel   the same from the bottom with U (both sweeps read W by rows: contiguous)
//
// This path is a fallback, sized for correctness first: two launches per column keep the panel factorisation simple (no grid-wide
// synchronisation inside a kernel); at n = 8192 it is ~10^4 small launches plus 128 MFMA updates.
// Differences from nalgebra that stay at tolerance level: nalgebra scales the column by the reciprocal of the pivot, this divides;
// nalgebra forms the explicit inverse and multiplies, this solves with the factors.
#pragma once

__global__ __launch_bounds__(1024) void lu_pivot_kernel(double* __restrict__ W, size_t ld, int k, int p0, int nrows, int* __restrict__ piv,
                                                        int* __restrict__ fail) {
    if (*fail) return;
    __shared__ double bv[16];
    __shared__ int bi[16];
    __shared__ int ps;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double best = -1.0;
    int idx = 0x7fffffff;
    for (int i = k + tid; i < nrows; i += 1024) {
        const double v = fabs(W[(size_t)i * ld + k]);
        if (v > best) { best = v; idx = i; } // ascending i per thread: the first maximum of this thread's rows
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        if (!(best > 0.0) || idx >= nrows) { *fail = 1; idx = k; } // no non-zero entry in this column: singular (newton/mod.rs:43-46)
        piv[k] = idx;
        ps = idx;
    }
    __syncthreads();
    const int p = ps;
    if (p != k && tid < 64) {
        double* a = W + (size_t)k * ld + p0 + tid;
        double* b = W + (size_t)p * ld + p0 + tid;
        const double va = *a, vb = *b;
        *a = vb; *b = va;
    }
}

__global__ __launch_bounds__(256) void lu_col_step_kernel(double* __restrict__ W, size_t ld, int k, int p0, int nrows, const int* __restrict__ fail) {
    if (*fail) return;
    __shared__ double urow[64];
    const int tid = threadIdx.x;
    const int jc = p0 + 64 - k; // columns k .. p0 + 63 of the pivot row
    if (tid < jc) urow[tid] = W[(size_t)k * ld + k + tid];
    __syncthreads();
    const double ukk = urow[0];
    const int r = tid >> 3, c8 = tid & 7; // 32 rows per workgroup, 8 lanes (64 contiguous bytes) per row
    for (int i = k + 1 + blockIdx.x * 32 + r; i < nrows; i += gridDim.x * 32) {
        double* row = W + (size_t)i * ld + k;
        const double l = row[0] / ukk; // (all 8 lanes of the row read it in the same wave instruction, before lane 0's store below)
        for (int c = 1 + c8; c < jc; c += 8) row[c] = row[c] - l * urow[c];
        if (c8 == 0) row[0] = l;
    }
}

__global__ __launch_bounds__(256) void lu_swap_rows_kernel(double* __restrict__ W, size_t ld, int p0, int ncols, const int* __restrict__ piv,
                                                           const int* __restrict__ fail) {
    if (*fail) return;
    __shared__ int pv[64];
    if (threadIdx.x < 64) pv[threadIdx.x] = piv[p0 + threadIdx.x];
    __syncthreads();
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ncols; j += gridDim.x * blockDim.x) {
        if (j >= p0 && j < p0 + 64) continue; // the panel's own columns were swapped step by step
        for (int q = 0; q < 64; ++q) {
            const int p = pv[q];
            if (p != p0 + q) {
                double* a = W + (size_t)(p0 + q) * ld + j;
                double* b = W + (size_t)p * ld + j;
                const double va = *a, vb = *b;
                *a = vb; *b = va;
            }
        }
    }
}

__global__ __launch_bounds__(256) void lu_trsm_kernel(double* __restrict__ W, size_t ld, int p0, int ncols, const int* __restrict__ fail) {
    if (*fail) return;
    __shared__ double L[QN_NB][QN_NB + 1];
    qn_tile_to_lds<256, false>(L, W + (size_t)p0 * ld + p0, ld);
    __syncthreads();
    const int j = p0 + 64 + blockIdx.x * 256 + threadIdx.x;
    if (j >= ncols) return;
    double x[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) x[r] = W[(size_t)(p0 + r) * ld + j];
#pragma unroll
    for (int c = 0; c < 63; ++c)
#pragma unroll
        for (int r = c + 1; r < 64; ++r) x[r] = x[r] - L[r][c] * x[c];
#pragma unroll
    for (int r = 1; r < 64; ++r) W[(size_t)(p0 + r) * ld + j] = x[r];
}

__global__ __launch_bounds__(256) void lu_gemm_kernel(double* __restrict__ W, size_t ld, int p0, const int* __restrict__ fail) {
    if (*fail) return;
    __shared__ double PI[QN_NB][QN_NB + 1]; // PI[k][i] = L21[i0 + i][p0 + k]
    __shared__ double PJ[QN_NB][QN_NB + 1]; // PJ[k][j] = U12[p0 + k][j0 + j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = p0 + 64 + blockIdx.y * 64, j0 = p0 + 64 + blockIdx.x * 64;
    qn_tile_to_lds<256, true>(PI, W + (size_t)i0 * ld + p0, ld);
    qn_tile_to_lds<256, false>(PJ, W + (size_t)p0 * ld + j0, ld);
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
                acc[a][b][reg] = -W[(size_t)(i0 + wi + a * 16 + l4 + 4 * reg) * ld + j0 + wj + b * 16 + l15];
    __syncthreads();
    qn_mfma_64(PI, PJ, QN_NB, wi, wj, lane, acc);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                W[(size_t)(i0 + wi + a * 16 + l4 + 4 * reg) * ld + j0 + wj + b * 16 + l15] = -acc[a][b][reg];
}

// x[i] = sign * b[perm[i]] (zero past n_src): the row permutation of the factorisation applied to a right-hand side
__global__ void lu_vec_perm_kernel(double* __restrict__ dst, const double* __restrict__ src, const int* __restrict__ perm, int n_src, int n_dst,
                                   double sign) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_dst; i += gridDim.x * blockDim.x) {
        const int p = perm[i];
        dst[i] = (p < n_src) ? sign * src[p] : 0.0;
    }
}

// L y = b (unit lower), block k0: y_k by substitution inside the 64 x 64 diagonal block, then rhs_i -= L[i, k-block] y_k below it
__global__ __launch_bounds__(256) void lu_fwd_step_kernel(const double* __restrict__ W, size_t ld, int k0, int nrows, double* __restrict__ rhs,
                                                          double* __restrict__ sol) {
    __shared__ double D[QN_NB][QN_NB + 1];
    __shared__ double xk[QN_NB];
    qn_tile_to_lds<256, false>(D, W + (size_t)k0 * ld + k0, ld);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        double v = rhs[k0 + lane];
#pragma unroll
        for (int c = 0; c < 63; ++c) {
            const double xc = qn_readlane_d(v, c);
            if (lane > c) v = v - D[lane][c] * xc;
        }
        xk[lane] = v;
        if (blockIdx.x == 0) sol[k0 + lane] = v;
    }
    __syncthreads();
    const double xl = xk[lane];
    for (int r = k0 + QN_NB + blockIdx.x * 4 + wave; r < nrows; r += gridDim.x * 4) {
        double p = W[(size_t)r * ld + k0 + lane] * xl;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
        if (lane == 0) rhs[r] = rhs[r] - p;
    }
}

// U z = y, block k0 from the bottom: z_k by back substitution inside the diagonal block, then rhs_i -= U[i, k-block] z_k above it
__global__ __launch_bounds__(256) void lu_bwd_step_kernel(const double* __restrict__ W, size_t ld, int k0, double* __restrict__ rhs,
                                                          double* __restrict__ sol) {
    __shared__ double D[QN_NB][QN_NB + 1];
    __shared__ double zk[QN_NB];
    qn_tile_to_lds<256, false>(D, W + (size_t)k0 * ld + k0, ld);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        double v = rhs[k0 + lane];
#pragma unroll
        for (int c = 63; c >= 0; --c) {
            const double zc = qn_readlane_d(v, c) / D[c][c];
            if (lane == c) v = zc;
            if (lane < c) v = v - D[lane][c] * zc;
        }
        zk[lane] = v;
        if (blockIdx.x == 0) sol[k0 + lane] = v;
    }
    __syncthreads();
    const double zl = zk[lane];
    for (int r = blockIdx.x * 4 + wave; r < k0; r += gridDim.x * 4) {
        double p = W[(size_t)r * ld + k0 + lane] * zl;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
        if (lane == 0) rhs[r] = rhs[r] - p;
    }
}

// symmetric bit for bit? (a device-resident Hessian whose symmetry the host has not seen): flags *nonsym
__global__ void lu_symmetry_check_kernel(const double* __restrict__ src, size_t ld_src, int n, int* __restrict__ nonsym) {
    const size_t total = (size_t)n * n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / n, j = e % n;
        if (j > i && src[i * ld_src + j] != src[j * ld_src + i]) *nonsym = 1;
    }
}
