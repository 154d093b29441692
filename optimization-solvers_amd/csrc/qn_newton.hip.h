// qn_newton.hip.h -- row f2: Newton direction on the GPU (src/newton/mod.rs:26-49).
//
// The reference inverts the Hessian (`try_inverse`, LU) and multiplies: d = -H^-1 g, decrement^2 = (H^-1 d).d.
// Here the (symmetric positive definite) Hessian is factorised once per iteration, H = L L', by a blocked
// right-looking Cholesky in f64 and both quantities come from triangular solves:
//   chol_diag_inv_kernel  64 x 64 diagonal block L_kk and its inverse, one wave, rows in registers
//   chol_panel_kernel     panel below the diagonal block: rows * L_kk^-T as a product with the stored inverse (f64 MFMA)
//   chol_syrk_kernel      C -= P P' on the lower triangle with v_mfma_f64_16x16x4_f64: after every 64-column panel only
//                         the rest of its 256-column outer block, then ONE depth-256 update of the trailing matrix
//                         (64 x 64 tile per workgroup, both panels staged k-major in LDS: conflict-free fragment reads)
//   tri_*_step_kernel     blocked forward / backward substitution for the vector right-hand sides, one launch per
//                         block: the block solve is a product with the stored inverse, the rest a row update
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

__device__ __forceinline__ double qn_readlane_d(double v, int l) { // wave-uniform broadcast of lane l's value (l uniform)
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- diagonal block: W[k0:k0+NB, k0:k0+NB] = L L' (lower, in place) and invL = L^-1 (dense 64 x 64, zeros above the
// diagonal) for the panel product and the triangular solves.  4 waves; for L: lane = row, wave w keeps columns
// 16w..16w+15 of that row in registers; for X = L^-1 (forward elimination of the identity, sharing the sweep): lane =
// column, wave w keeps rows 16w..16w+15.  Step j: the owning wave broadcasts the pivot with v_readlane and publishes
// column j of L (zeroed up to the diagonal) and the finished row j of X through double-buffered LDS vectors, ONE barrier,
// then every wave applies both rank-1 updates with plain FMAs.  Measured on the way here: a fully unrolled one-wave
// version 86 us (121 KB of cold instruction fetch), a predicated update 96 us (16 branches per step, serialised LDS
// waits), selects instead of zeroed multipliers 57 us, separate L and X sweeps 36 us.  Rows/cols past n are identity padding.
// Round 2, also measured and dropped: 16 columns at a time inside the owning wave (multipliers broadcast with v_readlane, four
// barriers instead of 64, rank-16 updates by the other waves): 24.5 us -- the 15 dependent v_readlane pairs per column cost more
// than the barrier they replace; and one fused launch per 64-column step (diagonal tile factorised redundantly in every panel
// workgroup, left-looking in-block updates): 38.7 us per step against ~37 us for the three launches.
#ifdef QN_DIAG_STAMPS
#define QN_DSTAMP(i) do { if (threadIdx.x == 0) qn_diag_stamps[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define QN_DSTAMP(i)
#endif
// (the body: the tile is in `a`, *bad_s == 0, the workgroup has met at a barrier.  chol_diag_inv_kernel runs it on a tile it loads;
// chol_syrk_kernel's first workgroup runs it on the diagonal tile it has just updated -- one launch and one kernel boundary less in
// every link of the chain)
__device__ __forceinline__ void chol_diag_inv_body(double (*a)[QN_NB + 1], double (*col)[QN_NB], double (*xrow)[QN_NB], int* bad_sp, double* __restrict__ W,
                                                   const size_t ld, const int k0, double* __restrict__ invL, int* __restrict__ fail) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int& bad_s = *bad_sp;
    double r[16], x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { r[c] = a[lane][16 * wave + c]; x[c] = (16 * wave + c == lane) ? 1.0 : 0.0; }
    for (int w0 = 0; w0 < 4; ++w0) {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * w0 + jj;
            if (wave == w0) {
                const double piv = qn_readlane_d(r[jj], j);
                if (!(piv > 0.0) || !isfinite(piv)) { if (lane == 0) bad_s = 1; }
                const double di = rsqrt(piv); // one reciprocal square root instead of sqrt -> divide on the sweep's critical path
                double l = r[jj] * di;
                if (lane == j) l = piv * di;
                r[jj] = l;
                const double xv = x[jj] * di; // row j of X is complete: X[j][:] = (e_j - sum_{p<j} L[j][p] X[p][:]) / L[j][j]
                x[jj] = xv;
                col[j & 1][lane] = (lane > j) ? l : 0.0; // zeros up to the diagonal: rows <= j and columns <= j drop out of the updates
                xrow[j & 1][lane] = xv;
            }
            __syncthreads();
            const double l = col[j & 1][lane], xp = xrow[j & 1][lane];
            double lc[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) lc[c] = col[j & 1][16 * wave + c]; // broadcast reads, all in flight together
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                r[c] = __builtin_fma(-l, lc[c], r[c]);  // A[lane][16w+c] -= L[lane][j] L[16w+c][j]
                x[c] = __builtin_fma(-lc[c], xp, x[c]); // X[16w+c][lane] -= L[16w+c][j] X[j][lane]
            }
        }
    }
    __syncthreads();
    QN_DSTAMP(2);
    if (bad_s) { // a non-positive pivot poisons everything after it: nothing is stored
        if (tid == 0) *fail = 1;
        return;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) { const int cc = 16 * wave + c; a[lane][cc] = (cc <= lane) ? r[c] : 0.0; }
    __syncthreads();
    for (int e = tid; e < QN_NB * QN_NB; e += 256) {
        const int i = e / QN_NB, j = e % QN_NB;
        if (j <= i) W[(size_t)(k0 + i) * ld + k0 + j] = a[i][j];
    }
    QN_DSTAMP(3);
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) { const int i = 16 * wave + ii; invL[i * QN_NB + lane] = (lane <= i) ? x[ii] : 0.0; }
    QN_DSTAMP(4);
    QN_DSTAMP(5);
}
__global__ __launch_bounds__(256) void chol_diag_inv_kernel(double* __restrict__ W, size_t ld, int k0, double* __restrict__ invL,
                                                            int* __restrict__ fail) {
    __shared__ double a[QN_NB][QN_NB + 1];
    __shared__ double col[2][QN_NB];
    __shared__ double xrow[2][QN_NB];
    __shared__ int bad_s;
    QN_DSTAMP(0);
    __builtin_amdgcn_s_setprio(3); // (a link of the dependent chain: it may run beside the bulk of a trailing update, see enqueue_newton)
    if (*fail) return;
    const int tid = threadIdx.x;
    qn_tile_to_lds<256, false>(a, W + (size_t)k0 * ld + k0, ld);
    if (tid == 0) bad_s = 0;
    __syncthreads();
    QN_DSTAMP(1);
    chol_diag_inv_body(a, col, xrow, &bad_s, W, ld, k0, invL, fail);
}

// 64 x 64 x 64 product on the f64 matrix cores from two k-major LDS panels: acc[a][b] += PI[:, wi+16a ..]' PJ[:, wj+16b ..]
__device__ __forceinline__ void qn_mfma_64(const double (*PI)[QN_NB + 1], const double (*PJ)[QN_NB + 1], int kdepth, int wi, int wj, int lane,
                                           v4d (&acc)[2][2]) {
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll 4
    for (int kk = 0; kk < kdepth; kk += 4) {
        // A operand: A[i = lane&15][k = lane>>4] ; B operand: B[k = lane>>4][j = lane&15]
        const double a0 = PI[kk + l4][wi + l15], a1 = PI[kk + l4][wi + 16 + l15];
        const double b0 = PJ[kk + l4][wj + l15], b1 = PJ[kk + l4][wj + 16 + l15];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// ---- panel: rows r >= k0+NB: W[r, k0:k0+NB] <- W[r, k0:k0+NB] * invL'   (64 rows per workgroup, f64 MFMA) ----
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ W, size_t ld, int k0, const double* __restrict__ invL,
                                                         const int* __restrict__ fail) {
    __builtin_amdgcn_s_setprio(3); // (a link of the dependent chain, see chol_syrk_kernel)
    if (*fail) return;
    __shared__ double PI[QN_NB][QN_NB + 1]; // PI[k][i] = W[r0 + i][k0 + k]
    __shared__ double PJ[QN_NB][QN_NB + 1]; // PJ[k][j] = invL[j][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = k0 + QN_NB + blockIdx.x * QN_NB;
    qn_tile_to_lds<256, true>(PI, W + (size_t)r0 * ld + k0, ld);
    qn_tile_to_lds<256, true>(PJ, invL, QN_NB);
    __syncthreads();
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    qn_mfma_64(PI, PJ, QN_NB, wi, wj, lane, acc);
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
                W[(size_t)(r0 + wi + a * 16 + l4 + 4 * reg) * ld + k0 + wj + b * 16 + l15] = acc[a][b][reg];
}

// ---- symmetric update on the lower triangle: C[ti, tj] -= P_ti P_tj' over the panel columns [kb, kb + klen) ----
// Tiles of 64 with origin c0 (rows c0 + 64 ti, columns c0 + 64 tj, tj <= ti); a linear grid over exactly those tiles.  klen = 64
// inside a 256-column outer block (only the block's remaining columns), klen = 256 for the trailing matrix, which
// quarters the read-modify-write traffic of the big update.  256 threads = 4 waves, wave w owns the 32 x 32 quadrant
// (w >> 1, w & 1): 2 x 2 MFMA tiles of 16 x 16; the panels are staged k-major in LDS 32 columns at a time.  Measured at
// n = 8192: 39.5 TFLOP/s on the depth-256 updates (50 % of the 78.6 TFLOP/s f64 MFMA peak); a 128 x 128-tile variant
// (4 x 4 MFMA tiles per wave, 2 waves per SIMD) was slower (5.7 ms vs 4.3 ms per factorisation) and was dropped.
#define QN_KC 32
// tile (ti, tj) of launch-linear index t in a grid of `ncols` column tiles (tj < ncols, tj <= ti): the first `ncols` tile
// rows are triangular, the rest are full -- no workgroup is launched for the upper triangle
__device__ __forceinline__ void qn_tri_tile(int t, int ncols, int& ti, int& tj) {
    const int tri = ncols * (ncols + 1) / 2;
    if (t >= tri) { const int r = t - tri; ti = ncols + r / ncols; tj = r % ncols; return; }
    int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    while (i * (i + 1) / 2 > t) --i;
    ti = i; tj = t - i * (i + 1) / 2;
}
static inline int qn_tri_tiles(int nrows_t, int ncols_t) { // number of tiles with tj < ncols_t, tj <= ti < nrows_t (ncols_t <= nrows_t)
    return ncols_t * (ncols_t + 1) / 2 + (nrows_t - ncols_t) * ncols_t;
}
// (`chain`: the launch is a link of the factorisation's dependent chain and runs beside the bulk of a trailing update on another
// stream: its waves ask the SIMD's arbiter for priority -- see the look-ahead in enqueue_newton)
// (`invL_next`: when given, the workgroup of the first tile -- the diagonal tile at c0, the next link's -- goes on to factorise and
// invert it (chol_diag_inv_body) instead of handing it to a launch of its own: the tile goes from the accumulators to LDS, not through
// memory, and the chain loses a kernel boundary per 64 columns.)
__global__ __launch_bounds__(256) void chol_syrk_kernel(double* __restrict__ W, size_t ld, int kb, int klen, int c0, int ncols, int* __restrict__ fail,
                                                        int chain = 0, double* __restrict__ invL_next = nullptr) {
    if (chain) __builtin_amdgcn_s_setprio(3);
    int ti, tj;
    qn_tri_tile(blockIdx.x, ncols, ti, tj);
    if (*fail) return;
    __shared__ double PP[2][QN_KC][QN_NB + 1]; // the two panels' chunks; afterwards (first workgroup) the 64 x 65 tile of the diagonal block
    __shared__ double dcol[2][QN_NB];
    __shared__ double dxrow[2][QN_NB];
    __shared__ int dbad;
    double (*PI)[QN_NB + 1] = PP[0];
    double (*PJ)[QN_NB + 1] = PP[1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = c0 + ti * QN_NB, j0 = c0 + tj * QN_NB;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    // thread t stages rows (t >> 5) + 8 u, u = 0..7, column (t & 31) of each 64 x 32 panel chunk (coalesced along k)
    const int sr = tid >> 5, sk = tid & 31;
    const double* pi = W + (size_t)(i0 + sr) * ld + kb + sk;
    const double* pj = W + (size_t)(j0 + sr) * ld + kb + sk;
    double vi[8], vj[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { vi[u] = pi[(size_t)(8 * u) * ld]; vj[u] = pj[(size_t)(8 * u) * ld]; }
    // the accumulators start at -C (its loads overlap the first panel chunk): the result is -(acc) = C - P_i P_j'
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
                acc[a][b][reg] = -W[(size_t)(i0 + wi + a * 16 + l4 + 4 * reg) * ld + j0 + wj + b * 16 + l15];
    for (int kc = 0; kc < klen; kc += QN_KC) {
        if (kc) __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) { PI[sk][sr + 8 * u] = vi[u]; PJ[sk][sr + 8 * u] = vj[u]; }
        __syncthreads();
        if (kc + QN_KC < klen) { // next chunk's loads fly while this one is multiplied
#pragma unroll
            for (int u = 0; u < 8; ++u) { vi[u] = pi[(size_t)(8 * u) * ld + kc + QN_KC]; vj[u] = pj[(size_t)(8 * u) * ld + kc + QN_KC]; }
        }
#pragma unroll
        for (int kk = 0; kk < QN_KC; kk += 4) {
            const double a0 = PI[kk + l4][wi + l15], a1 = PI[kk + l4][wi + 16 + l15];
            const double b0 = PJ[kk + l4][wj + l15], b1 = PJ[kk + l4][wj + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                W[(size_t)(i0 + wi + a * 16 + l4 + 4 * reg) * ld + j0 + wj + b * 16 + l15] = -acc[a][b][reg];
    if (invL_next != nullptr && blockIdx.x == 0) { // (uniform per workgroup; tile (0, 0): i0 = j0 = c0)
        double (*T)[QN_NB + 1] = reinterpret_cast<double (*)[QN_NB + 1]>(&PP[0][0][0]);
        __syncthreads(); // (the last chunk's reads of the panels)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) T[wi + a * 16 + l4 + 4 * reg][wj + b * 16 + l15] = -acc[a][b][reg];
        if (tid == 0) dbad = 0;
        __syncthreads();
        chol_diag_inv_body(T, dcol, dxrow, &dbad, W, ld, c0, invL_next, fail);
    }
}

// Round 4, built, measured and dropped: ONE launch per 64-column step, left-looking inside the outer block (`chol_step_kernel`:
// workgroup b gives tile (k + b, k) what it owes to the outer block's earlier panels -- the tile in the accumulators throughout --,
// workgroup 0 then factorises and inverts the diagonal tile and raises a counter, the others take the inverse as write-through stores /
// sc1 loads and multiply; no in-block update launches).  Correct (32 Newton tests), and 9.49 ms against 9.20 at n = 8192 (3.52 / 3.34
// at 4096): a step took 40 us alone and 62-70 us beside the bulk, against 27-47 + 7 with the two launches -- the left-looking update
// in front of the diagonal block (depth up to 192: six 32-deep chunks, each a global-load round trip of ONE workgroup that nothing
// hides) and the inverse's trip through memory behind it are both on the chain, where the right-looking in-block update is a 6 us
// launch of many workgroups and the panel a 7 us one.
// ---- triangular solves with a vector right-hand side, one launch per 64-block ----
// Every workgroup first forms the block's solution from the stored inverse (a 64 x 64 product: cheaper than waiting for
// a separate launch), workgroup 0 stores it into `sol`, then all update their rows of the running right-hand side `rhs`.
__device__ __forceinline__ void qn_block_solve(const double* __restrict__ invL, const double* __restrict__ rk, bool transpose, double* xk /*LDS[64]*/,
                                               double* rs /*LDS[64]*/) {
    const int tid = threadIdx.x; // 256 threads: thread = (row tid >> 2, quarter tid & 3)
    if (tid < QN_NB) rs[tid] = rk[tid];
    __syncthreads();
    const int row = tid >> 2, q = tid & 3;
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int k = q * 16 + c;
        const double m = transpose ? invL[k * QN_NB + row] : invL[row * QN_NB + k];
        acc = __builtin_fma(m, rs[k], acc);
    }
    acc = acc + __shfl_xor(acc, 1, 64);
    acc = acc + __shfl_xor(acc, 2, 64);
    if (q == 0) xk[row] = acc;
    __syncthreads();
}
// forward (L y = b), block k: y_k = invL_k rhs_k ; rhs_i -= L[i, k-block] y_k for the rows below (one wave per row)
__global__ __launch_bounds__(256) void tri_fwd_step_kernel(const double* __restrict__ W, size_t ld, int k0, int n_pad, const double* __restrict__ invL,
                                                           double* __restrict__ rhs, double* __restrict__ sol) {
    __shared__ double xk[QN_NB], rs[QN_NB];
    qn_block_solve(invL, rhs + k0, false, xk, rs);
    if (blockIdx.x == 0 && threadIdx.x < QN_NB) sol[k0 + threadIdx.x] = xk[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double xl = xk[lane];
    for (int r = k0 + QN_NB + blockIdx.x * 4 + wave; r < n_pad; r += gridDim.x * 4) {
        double p = W[(size_t)r * ld + k0 + lane] * xl;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
        if (lane == 0) rhs[r] = rhs[r] - p;
    }
}
// backward (L' z = y), block k from the end: z_k = invL_k' rhs_k ; rhs_j -= sum_{i in block k} L[i][j] z_i for j < k0
__global__ __launch_bounds__(256) void tri_bwd_step_kernel(const double* __restrict__ W, size_t ld, int k0, const double* __restrict__ invL,
                                                           double* __restrict__ rhs, double* __restrict__ sol) {
    __shared__ double zk[QN_NB], rs[QN_NB];
    qn_block_solve(invL, rhs + k0, true, zk, rs);
    if (blockIdx.x == 0 && threadIdx.x < QN_NB) sol[k0 + threadIdx.x] = zk[threadIdx.x];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < k0; j += gridDim.x * blockDim.x) {
        double acc = 0.0;
#pragma unroll 16
        for (int i = 0; i < QN_NB; ++i) acc = __builtin_fma(W[(size_t)(k0 + i) * ld + j], zk[i], acc);
        rhs[j] = rhs[j] - acc;
    }
}

// ---- triangular solves for large n: 512-wide blocks whose inverses are assembled from the 64-wide ones ----
// inv([A 0; B C]) = [A^-1 0; -C^-1 B A^-1, C^-1]: three doubling levels (64 -> 128 -> 256 -> 512), each two batched MFMA
// products and a copy; a sweep is then 16 steps of (block product, panel update) instead of 128 dependent launches.
#define QN_TS 512

// batched 64 x 64 tile of  OUT = sign * A B  (row-major operands, depth `depth`, a multiple of 64); z = batch index
struct QnBatchGemm {
    const double* A; size_t lda; size_t a_batch; // A + z * a_batch : rows x depth
    const double* B; size_t ldb; size_t b_batch; // B + z * b_batch : depth x cols
    double* C; size_t ldc; size_t c_batch;
    int depth;
    double sign;
};
__global__ __launch_bounds__(256) void tri_batch_gemm_kernel(const QnBatchGemm g) {
    __shared__ double PI[QN_NB][QN_NB + 1]; // PI[k][i] = A[i0 + i][k]
    __shared__ double PJ[QN_NB][QN_NB + 1]; // PJ[k][j] = B[k][j0 + j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.y * QN_NB, j0 = blockIdx.x * QN_NB;
    const double* A = g.A + (size_t)blockIdx.z * g.a_batch + (size_t)i0 * g.lda;
    const double* B = g.B + (size_t)blockIdx.z * g.b_batch + j0;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int kc = 0; kc < g.depth; kc += QN_NB) {
        if (kc) __syncthreads();
        qn_tile_to_lds<256, true>(PI, A + kc, g.lda);
        qn_tile_to_lds<256, false>(PJ, B + (size_t)kc * g.ldb, g.ldb);
        __syncthreads();
        qn_mfma_64(PI, PJ, QN_NB, wi, wj, lane, acc);
    }
    const int l15 = lane & 15, l4 = lane >> 4;
    double* C = g.C + (size_t)blockIdx.z * g.c_batch;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                C[(size_t)(i0 + wi + a * 16 + l4 + 4 * reg) * g.ldc + j0 + wj + b * 16 + l15] = g.sign * acc[a][b][reg];
}
// out block p (2s x 2s) <- [in[2p] 0; (left to the product) in[2p+1]]
__global__ void tri_inv_assemble_kernel(const double* __restrict__ in, double* __restrict__ out, int s, int npairs) {
    const size_t per = (size_t)4 * s * s, total = per * npairs;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t p = e / per, w = e % per;
        const int i = (int)(w / (2 * s)), j = (int)(w % (2 * s));
        if (i < s && j < s) out[e] = in[(2 * p) * (size_t)s * s + (size_t)i * s + j];
        else if (i >= s && j >= s) out[e] = in[(2 * p + 1) * (size_t)s * s + (size_t)(i - s) * s + (j - s)];
        else if (i < s) out[e] = 0.0;
    }
}
__global__ void tri_transpose_blocks_kernel(const double* __restrict__ in, double* __restrict__ out, int s, int nblocks) {
    __shared__ double t[32][33];
    const int tiles = s / 32;
    const int blk = blockIdx.z, ti = blockIdx.y, tj = blockIdx.x;
    if (ti >= tiles || tj >= tiles || blk >= nblocks) return;
    const double* src = in + (size_t)blk * s * s;
    double* dst = out + (size_t)blk * s * s;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 256 threads: 32 x 8
    for (int r = ty; r < 32; r += 8) t[r][tx] = src[(size_t)(ti * 32 + r) * s + tj * 32 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) dst[(size_t)(tj * 32 + r) * s + ti * 32 + tx] = t[tx][r];
}

// y[i] (= or -=) M[i][0:512] . v[0:512], one wave per row, 4 rows in flight per wave for latency overlap
__global__ __launch_bounds__(256) void tri_rowdot512_kernel(const double* __restrict__ M, size_t ldm, int nrows, const double* __restrict__ v,
                                                            double* __restrict__ y, int subtract) {
    __shared__ double vs[QN_TS];
    for (int k = threadIdx.x; k < QN_TS; k += 256) vs[k] = v[k];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double vl[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) { vl[2 * u] = vs[128 * u + 2 * lane]; vl[2 * u + 1] = vs[128 * u + 2 * lane + 1]; }
    const int stride = gridDim.x * 4;
    for (int r0 = blockIdx.x * 4 + wave; r0 < nrows; r0 += 4 * stride) {
        double p[4];
        v2d m[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + q * stride;
            const double* row = M + (size_t)(r < nrows ? r : r0) * ldm + 2 * lane;
#pragma unroll
            for (int u = 0; u < 4; ++u) m[q][u] = *reinterpret_cast<const v2d*>(row + 128 * u);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double a = 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) { a = __builtin_fma(m[q][u].x, vl[2 * u], a); a = __builtin_fma(m[q][u].y, vl[2 * u + 1], a); }
            p[q] = a;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) p[q] = p[q] + __shfl_xor(p[q], off, 64);
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + q * stride;
                if (r < nrows) y[r] = subtract ? (y[r] - p[q]) : p[q];
            }
        }
    }
}
// backward panel update: y[j] -= sum_{i < 512} M[i][j] z[i] for j < ncols; workgroup = 64 columns x 4 waves of 128 rows each
__global__ __launch_bounds__(256) void tri_coldot512_kernel(const double* __restrict__ M, size_t ldm, int ncols, const double* __restrict__ z,
                                                            double* __restrict__ y) {
    __shared__ double zs[QN_TS];
    __shared__ double part[4][64];
    for (int k = threadIdx.x; k < QN_TS; k += 256) zs[k] = z[k];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const double* col = M + (size_t)(wave * 128) * ldm + (j < ncols ? j : 0);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = 0; i0 < 128; i0 += 16) {
        double m[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) m[u] = col[(size_t)(i0 + u) * ldm];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_fma(m[u], zs[wave * 128 + i0 + u], acc[u & 3]);
    }
    part[wave][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (wave == 0 && j < ncols) y[j] = y[j] - (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]);
}

// Hessian staging: W (row-major, ld = n_pad64) <- src rows (ld_src), identity on the padding diagonal
// (lower_only: the Cholesky path reads the lower block triangle only -- the 64 x 64 tiles on and below the diagonal: the rest of the
// copy, half of its 1 GB at n = 8192, is left out)
__global__ void newton_stage_kernel(double* __restrict__ W, size_t ld, int n, int n_pad64, const double* __restrict__ src, size_t ld_src, int lower_only) {
    const size_t total = (size_t)n_pad64 * ld;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / ld, j = e % ld;
        if (lower_only && j > (i | 63)) continue;
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
