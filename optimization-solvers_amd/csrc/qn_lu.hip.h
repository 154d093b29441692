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
// Round 4 (n = 8192: 65 -> 45 ms per Newton iteration, every path the same bits):
//     lu_panel_persist_kernel   the panel's 17 launches as ONE grid of 58 workgroups that wait for each other on counters in memory
//     lu_swap_rows2 / lu_trsm2 / lu_gemm2_kernel   the per-panel steps on a RANGE of columns: the look-ahead (qn_hip.hip) keeps the next
//                               panel's 64 columns on the solver's stream and sends everything else to a CU-masked second stream
//     lu_la_swap_trsm / lu_la_gemm_kernel   the look-ahead columns in two launches that read the panel from its column-major buffer and
//                               leave the next panel in the second buffer
//     lu_sweep_kernel           a whole substitution sweep in one launch (the solution entry is its own flag: sentinel NaN)
//
// Rounds 1-2 ran the panel with these two launches per column (~16 000 small launches at n = 8192: 146-196 ms per Newton
// iteration, rocprofv3 r03_a); they remain the path for panels of more than 8192 rows and the reference the panel kernels below
// are pinned against bit for bit (tests/test_gpu_newton.py).  Round 3: lu_panel_step_kernel -- 19 launches per panel, 65 ms.
// As nalgebra's gauss_step, the column is scaled by the reciprocal of the pivot.  Difference that stays at tolerance level:
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
    const double inv_ukk = 1.0 / urow[0]; // nalgebra's gauss_step scales the column by the reciprocal of the pivot (`coeffs *= inv_diag`)
    const int r = tid >> 3, c8 = tid & 7; // 32 rows per workgroup, 8 lanes (64 contiguous bytes) per row
    for (int i = k + 1 + blockIdx.x * 32 + r; i < nrows; i += gridDim.x * 32) {
        double* row = W + (size_t)i * ld + k;
        const double l = row[0] * inv_ukk; // (all 8 lanes of the row read it in the same wave instruction, before lane 0's store below)
        for (int c = 1 + c8; c < jc; c += 8) row[c] = row[c] - l * urow[c];
        if (c8 == 0) row[0] = l;
    }
}

// ------------------------------------------------------------------------------------------------
// PANEL FACTORISATION WITHOUT ONE LAUNCH PER COLUMN (round 3).  Partial pivoting is a chain of n dependent steps -- search a
// column, swap, eliminate -- and two launches per step made the chain 2 n kernel boundaries (16 384 at n = 8192: 120 of the
// path's 146 ms).  Now the 64-column panel is copied into a column-major buffer P (lu_panel_load_kernel: every column
// contiguous) and worked on four columns at a time by lu_panel_step_kernel, launched 17 times per panel:
//   workgroup 0 (role A), launch s < 16: sub-panel s = columns 4 s .. 4 s + 3.  ONE workgroup of 512 threads holds all its rows
//     in registers (16 rows x 4 columns per thread: panels of up to 8192 rows) and runs the four pivot steps with two workgroup
//     barriers each -- search (wave butterflies, 8 candidates through LDS), swap and broadcast of the pivot row through LDS,
//     elimination in registers;
//   workgroups 2 .. (role B), launch s >= 1: one column right of sub-panel s each, bringing it up to date with sub-panel s - 1:
//     its four row swaps, the 4 x 4 unit-lower solve for the column's U entries, the rank-4 update of the rows below (role A does
//     the same for its own four columns before it factors them).  A and B only read sub-panel s - 1, which nobody writes in
//     launch s: no synchronisation inside the launch;
//   workgroup 1 (role C), launch s >= 2: the swaps of sub-panel s - 1 on the columns LEFT of it (finished multipliers) -- one
//     launch late, because in launch s - 1 role B was still reading those of sub-panel s - 2 in the old row order.
// The arithmetic is the per-column kernels' (the column scaled by the reciprocal of the pivot as nalgebra's gauss_step does, a - l u
// with separate rounding, updates applied in column order): the factors are the same bits, and the old kernels stay as the path
// for panels of more than 8192 rows.
// Where the time goes (rocprofv3 + in-kernel stamps, n = 8192, one factorisation = 67 ms with its two solves): the 2176 step
// launches 39.8 ms -- role A is ONE CU working through a dependent chain: ~2.5 us per pivot step at 16 rows per thread (search
// 1.3, exchange 0.3, elimination 0.8: instruction issue, not memory), 4.6 us of launch and flag read, and its 768 KB of column
// traffic at a single CU's rate; the rank-64 MFMA updates 12.4 ms; triangular solve of U12 4.5 ms; the two vector solves 6 ms.
// Round 4, measured and dropped: TWO-LEVEL BLOCKING -- panels applied only inside an outer block of 256 columns, the columns right
// of it given the block row by block substitution and ONE update of depth 256 (a quarter of the 46 GB the rank-64 updates move,
// which round 3 took for their bound).  Correct (30 Newton tests green), and 66.0 ms against 65-67: the updates went from 12.4 to
// 10.2 ms only -- they are MFMA-bound (3.7e11 flop per factorisation at the 47.8 TFLOP/s a pure f64 MFMA loop sustains here: 7.7
// ms), not HBM-bound -- while the block substitution added 92 launches of the U12 solve, whose one thread per column takes 35 us
// whatever the width (4.5 -> 7.6 ms).  What would pay is that solve on the matrix cores (the 64 x 64 unit-lower inverse once per
// panel, then a product) and the vector solves in fewer, wider steps; the chain of step launches (39 ms) is untouched by either.
// ------------------------------------------------------------------------------------------------
#define QN_LU_SUB 4
#define QN_LU_PT 512
#define QN_LU_RPT 16 // rows per thread: 8192 rows per panel at most (512 x 16: 1024 x 8 left the four pivot steps 128 registers -- three spilled words, each reload a memory round trip inside the dependent chain)

// P[c][i] = W[p0 + i][p0 + c], i < m (m a multiple of 64): 64 x 64 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void lu_panel_load_kernel(const double* __restrict__ W, size_t ld, int p0, double* __restrict__ P, size_t pld,
                                                            const int* __restrict__ fail) {
    if (*fail) return;
    __shared__ double T[QN_NB][QN_NB + 1];
    const int i0 = blockIdx.x * QN_NB;
    for (int e = threadIdx.x; e < QN_NB * QN_NB; e += 256) T[e >> 6][e & 63] = W[(size_t)(p0 + i0 + (e >> 6)) * ld + p0 + (e & 63)];
    __syncthreads();
    for (int e = threadIdx.x; e < QN_NB * QN_NB; e += 256) P[(size_t)(e >> 6) * pld + i0 + (e & 63)] = T[e & 63][e >> 6];
}
__global__ __launch_bounds__(256) void lu_panel_store_kernel(double* __restrict__ W, size_t ld, int p0, const double* __restrict__ P, size_t pld,
                                                             const int* __restrict__ fail) {
    if (*fail) return;
    __shared__ double T[QN_NB][QN_NB + 1];
    const int i0 = blockIdx.x * QN_NB;
    for (int e = threadIdx.x; e < QN_NB * QN_NB; e += 256) T[e & 63][e >> 6] = P[(size_t)(e >> 6) * pld + i0 + (e & 63)];
    __syncthreads();
    for (int e = threadIdx.x; e < QN_NB * QN_NB; e += 256) W[(size_t)(p0 + i0 + (e >> 6)) * ld + p0 + (e & 63)] = T[e >> 6][e & 63];
}

#ifdef QN_LU_STAMPS
__device__ unsigned long long qn_lu_dbg[64 * 16];
#ifndef QN_LU_STAMP_P0
#define QN_LU_STAMP_P0 640 // the panel whose role A is stamped (p0: its first column)
#endif
#define QN_LU_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && s < 64 && p0 == QN_LU_STAMP_P0) qn_lu_dbg[s * 16 + (k)] = wall_clock64(); } while (0)
#else
#define QN_LU_STAMP(k) do { } while (0)
#endif
// RPT: rows per thread -- 16 for the first panels of an 8192-row matrix, fewer as the panels get shorter (the loops over a thread's
// rows are unrolled, registers: a short panel must not pay for sixteen predicated copies of every step)
// The three roles as functions of (sub-panel, column): lu_panel_step_kernel runs them one launch per sub-panel, lu_panel_persist_kernel
// (below) in ONE launch per panel, the workgroups waiting for each other on counters in memory.
// COH: every access to the panel buffer and the pivots as a relaxed atomic at agent scope (loads and stores that go past the
// non-coherent cache levels: what the one-launch panel needs between workgroups on different XCDs -- see lu_panel_persist_kernel)
template <bool COH> __device__ __forceinline__ double lu_ld(const double* p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH> __device__ __forceinline__ int lu_ldi(const int* p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH> __device__ __forceinline__ void lu_st(double* p, const double v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool COH> __device__ __forceinline__ void lu_sti(int* p, const int v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
// (best, idx) of every lane -> the first maximum among the WIDTH lanes of its group: the larger value, between equal values the smaller row index
// (nalgebra's icamax takes the FIRST row of maximal magnitude).  Round 6: the value alone goes through the xor butterfly, then the smallest index
// among the lanes that hold it -- half the instructions of carrying (value, index) through every level with the full comparison (the same
// result in every lane: a maximum and a minimum do not depend on the order they are taken in).  `best` is never a NaN (lu_sub_factor).
template <int WIDTH>
__device__ __forceinline__ void lu_argmax_lanes(double& best, int& idx) {
    double wm = best;
#define QN_LU_LVL(OFF) if constexpr (WIDTH > OFF) { const double o = qn_xor_lanes<OFF>(wm); wm = o > wm ? o : wm; }
    QN_LU_LVL(32) QN_LU_LVL(16) QN_LU_LVL(8) QN_LU_LVL(4) QN_LU_LVL(2) QN_LU_LVL(1)
#undef QN_LU_LVL
    int c = best == wm ? idx : 0x7fffffff;
#define QN_LU_LVL(OFF) if constexpr (WIDTH > OFF) { const int o = qn_xor_lanes_i<OFF>(c); c = o < c ? o : c; }
    QN_LU_LVL(32) QN_LU_LVL(16) QN_LU_LVL(8) QN_LU_LVL(4) QN_LU_LVL(2) QN_LU_LVL(1)
#undef QN_LU_LVL
    best = wm;
    idx = c;
}
__device__ __forceinline__ double lu_uniform_d(const double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
struct QnLuLds {
    double U[QN_LU_SUB][QN_LU_SUB]; // U[q][j]: row r0 + q of column c0 + j after the solve
    double bv[QN_LU_SUB][16];
    int bi[QN_LU_SUB][16];
    double rowk[QN_LU_SUB][QN_LU_SUB], rowp[QN_LU_SUB][QN_LU_SUB];
    double pat[QN_LU_SUB][2 * QN_LU_SUB]; // pat[j][e]: column c0 + j's entry in slot e after the swaps and the 4 x 4 solve (lu_cols_update_pre)
    double l11[QN_LU_SUB][QN_LU_SUB]; // the unit-lower 4 x 4 block of the sub-panel just factorised (l11[r][c], r > c), for lu_cols_update_keep
};
// role C: the swaps of the sub-panel at r0 on the columns left of it (thread = column)
template <bool COH>
__device__ __forceinline__ void lu_role_c(double* __restrict__ P, const size_t pld, const int r0, const int p0, const int* __restrict__ piv, const int tid) {
    if (tid < r0) {
        double* col = P + (size_t)tid * pld;
        int ix[2 * QN_LU_SUB];
#pragma unroll
        for (int q = 0; q < QN_LU_SUB; ++q) { ix[q] = r0 + q; ix[QN_LU_SUB + q] = lu_ldi<COH>(piv + p0 + r0 + q) - p0; }
        double v[2 * QN_LU_SUB];
#pragma unroll
        for (int e = 0; e < 2 * QN_LU_SUB; ++e) v[e] = lu_ld<COH>(col + ix[e]);
        int canon[2 * QN_LU_SUB];
#pragma unroll
        for (int e = 0; e < 2 * QN_LU_SUB; ++e) {
            canon[e] = e;
#pragma unroll
            for (int f = 2 * QN_LU_SUB - 1; f >= 0; --f)
                if (f < e && ix[f] == ix[e]) canon[e] = f;
        }
#pragma unroll
        for (int q = 0; q < QN_LU_SUB; ++q) {
            const int sa = canon[q], sb = canon[QN_LU_SUB + q];
            double va = 0.0, vb = 0.0;
#pragma unroll
            for (int e = 0; e < 2 * QN_LU_SUB; ++e) { if (e == sa) va = v[e]; if (e == sb) vb = v[e]; }
#pragma unroll
            for (int e = 0; e < 2 * QN_LU_SUB; ++e) { if (e == sa) v[e] = vb; else if (e == sb) v[e] = va; }
        }
#pragma unroll
        for (int e = 0; e < 2 * QN_LU_SUB; ++e)
            if (canon[e] == e) lu_st<COH>(col + ix[e], v[e]);
    }
}
// roles A and B, first half: the nc columns from c0 on into registers, brought up to date with the sub-panel at r0 (prev: there is one)
template <int RPT, bool COH, int LCH>
__device__ __forceinline__ void lu_cols_update(double* __restrict__ P, const size_t pld, const int m, const int r0, const bool prev, const int c0, const int nc,
                                               const int p0, const int* __restrict__ piv, QnLuLds& L, double (&a)[QN_LU_SUB][RPT], const int tid) {
    if (prev) {
        if (tid < nc) { // thread j, column c0 + j: the four swaps, then the unit-lower 4 x 4 solve
            // Two memory round trips, not fourteen: the pivots first, then the eight entries the swaps can touch and the six
            // multipliers, all at once; the swaps are replayed on those eight in registers (two slots may name the same row: the
            // first slot of a row holds its value), and only then is anything stored.  (First version, in-order loads and stores on
            // the column: every launch of this kernel took 16 us even on a 64-row panel.)
            double* col = P + (size_t)(c0 + tid) * pld;
            int ix[2 * QN_LU_SUB];
#pragma unroll
            for (int q = 0; q < QN_LU_SUB; ++q) { ix[q] = r0 + q; ix[QN_LU_SUB + q] = lu_ldi<COH>(piv + p0 + r0 + q) - p0; }
            double v[2 * QN_LU_SUB], l11[QN_LU_SUB][QN_LU_SUB];
#pragma unroll
            for (int e = 0; e < 2 * QN_LU_SUB; ++e) v[e] = lu_ld<COH>(col + ix[e]);
#pragma unroll
            for (int c = 0; c < QN_LU_SUB - 1; ++c)
#pragma unroll
                for (int r = c + 1; r < QN_LU_SUB; ++r) l11[r][c] = lu_ld<COH>(P + (size_t)(r0 + c) * pld + r0 + r);
            int canon[2 * QN_LU_SUB]; // the first slot that names the same row
#pragma unroll
            for (int e = 0; e < 2 * QN_LU_SUB; ++e) {
                canon[e] = e;
#pragma unroll
                for (int f = 2 * QN_LU_SUB - 1; f >= 0; --f)
                    if (f < e && ix[f] == ix[e]) canon[e] = f;
            }
#pragma unroll
            for (int q = 0; q < QN_LU_SUB; ++q) { // swap rows r0 + q and piv[r0 + q] (slots q and 4 + q), in pivot order
                const int sa = canon[q], sb = canon[QN_LU_SUB + q];
                double va = 0.0, vb = 0.0;
#pragma unroll
                for (int e = 0; e < 2 * QN_LU_SUB; ++e) { if (e == sa) va = v[e]; if (e == sb) vb = v[e]; }
#pragma unroll
                for (int e = 0; e < 2 * QN_LU_SUB; ++e) { if (e == sa) v[e] = vb; else if (e == sb) v[e] = va; }
            }
            // slots 0..3 are canonical for rows r0 .. r0 + 3 (distinct rows, first in the list): the solve works on them
#pragma unroll
            for (int c = 0; c < QN_LU_SUB - 1; ++c)
#pragma unroll
                for (int r = c + 1; r < QN_LU_SUB; ++r) v[r] = v[r] - l11[r][c] * v[c];
#pragma unroll
            for (int e = 0; e < 2 * QN_LU_SUB; ++e)
                if (canon[e] == e) lu_st<COH>(col + ix[e], v[e]);
#pragma unroll
            for (int q = 0; q < QN_LU_SUB; ++q) L.U[q][tid] = v[q];
        }
        __syncthreads(); // (the swapped entries are read below by other threads of this workgroup: same CU, same L1)
    }
#pragma unroll
    for (int jr = 0; jr < RPT; ++jr) {
        const int i = tid + QN_LU_PT * jr;
#pragma unroll
        for (int j = 0; j < QN_LU_SUB; ++j) a[j][jr] = (i < m && j < nc) ? lu_ld<COH>(P + (size_t)(c0 + j) * pld + i) : 0.0;
    }
    if (prev) {
        // the multipliers CH rows-per-thread at a time, requested together and OUTSIDE any branch (first version: four loads inside
        // `if (row below the sub-panel)` per row of the thread: the compiler waits for memory at every such block).  Measured at
        // n = 8192 (whole Newton iteration): role A with CH = 4 at up to 8 rows per thread and CH = 1 at 16 (at 16, CH = 2 and 4 were
        // 2-3 ms SLOWER: 251-256 registers, and the update's arithmetic no longer interleaves with the requests), role B 16 / 4.
        constexpr int CH = RPT < LCH ? RPT : LCH;
#pragma unroll
        for (int jc = 0; jc < RPT; jc += CH) {
            double l[CH][QN_LU_SUB];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int i = tid + QN_LU_PT * (jc + u);
                const bool on = i >= r0 + QN_LU_SUB && i < m;
#pragma unroll
                for (int q = 0; q < QN_LU_SUB; ++q) l[u][q] = on ? lu_ld<COH>(P + (size_t)(r0 + q) * pld + i) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int i = tid + QN_LU_PT * (jc + u);
                if (i >= r0 + QN_LU_SUB && i < m) {
#pragma unroll
                    for (int j = 0; j < QN_LU_SUB; ++j)
                        if (j < nc) {
#pragma unroll
                            for (int q = 0; q < QN_LU_SUB; ++q) a[j][jc + u] = a[j][jc + u] - l[u][q] * L.U[q][j]; // (column order: the per-column kernels' rounding)
                        }
                }
            }
        }
    }
}
// role A of the one-launch panel.  Two things differ from lu_cols_update:
//   * KEEP (up to 8 rows per thread): the update takes the PREVIOUS sub-panel's multipliers from where they still are -- this thread's
//     own registers (lprev: the rows a thread holds are the same in every sub-panel) -- instead of reading back what the workgroup has
//     just stored; the 4 x 4 block comes from LDS (L.l11) and the pivots from registers (pvprev) in either case.  With KEEP nothing
//     here depends on those stores having completed: the wait for them moves behind this function (lu_panel_persist_kernel);
//   * LOAD FIRST, PATCH AFTER: the new columns are requested at once -- together with the eight entries per column the swaps can touch
//     -- and the four threads that replay the swaps and solve the 4 x 4 system publish the (at most eight) changed entries per column
//     in LDS, from where the rows' owners take them into their registers.  Before: those threads loaded, stored, waited, and only
//     behind the barrier did the workgroup ask for its columns -- two memory round trips in front of every sub-panel's update.
//     (The changed entries need not go back to memory here: the sub-panel's store at the end writes every row.)
// Same products subtracted in the same order: the same bits.
template <int RPT, bool KEEP>
__device__ __forceinline__ void lu_cols_update_pre(double* __restrict__ P, const size_t pld, const int m, const int r0, const bool prev, const int c0, const int p0,
                                                   const int (&pvprev)[QN_LU_SUB], QnLuLds& L, const double (&lprev)[QN_LU_SUB][KEEP ? RPT : 1],
                                                   double (&a)[QN_LU_SUB][RPT], const int tid) {
    int ix[2 * QN_LU_SUB];
    double v[2 * QN_LU_SUB];
#pragma unroll
    for (int q = 0; q < QN_LU_SUB; ++q) { ix[q] = r0 + q; ix[QN_LU_SUB + q] = pvprev[q] - p0; }
    const bool solver = prev && tid < QN_LU_SUB; // thread j: column c0 + j
    {
        const double* col = P + (size_t)(c0 + (tid & (QN_LU_SUB - 1))) * pld;
#pragma unroll
        for (int e = 0; e < 2 * QN_LU_SUB; ++e) v[e] = solver ? lu_ld<true>(col + ix[e]) : 0.0; // (requested in front of the columns: a wave's loads return in order)
    }
#pragma unroll
    for (int jr = 0; jr < RPT; ++jr) {
        const int i = tid + QN_LU_PT * jr;
#pragma unroll
        for (int j = 0; j < QN_LU_SUB; ++j) a[j][jr] = (i < m) ? lu_ld<true>(P + (size_t)(c0 + j) * pld + i) : 0.0;
    }
    if (!prev) return; // (uniform)
    int canon[2 * QN_LU_SUB]; // the first slot that names the same row (the same for every column: it follows from the pivots alone)
#pragma unroll
    for (int e = 0; e < 2 * QN_LU_SUB; ++e) {
        canon[e] = e;
#pragma unroll
        for (int f = 2 * QN_LU_SUB - 1; f >= 0; --f)
            if (f < e && ix[f] == ix[e]) canon[e] = f;
    }
    if (solver) {
#pragma unroll
        for (int q = 0; q < QN_LU_SUB; ++q) { // swap rows r0 + q and piv[r0 + q] (slots q and 4 + q), in pivot order
            const int sa = canon[q], sb = canon[QN_LU_SUB + q];
            double va = 0.0, vb = 0.0;
#pragma unroll
            for (int e = 0; e < 2 * QN_LU_SUB; ++e) { if (e == sa) va = v[e]; if (e == sb) vb = v[e]; }
#pragma unroll
            for (int e = 0; e < 2 * QN_LU_SUB; ++e) { if (e == sa) v[e] = vb; else if (e == sb) v[e] = va; }
        }
#pragma unroll
        for (int c = 0; c < QN_LU_SUB - 1; ++c)
#pragma unroll
            for (int r = c + 1; r < QN_LU_SUB; ++r) v[r] = v[r] - L.l11[r][c] * v[c];
#pragma unroll
        for (int e = 0; e < 2 * QN_LU_SUB; ++e) L.pat[tid][e] = v[e];
#pragma unroll
        for (int q = 0; q < QN_LU_SUB; ++q) L.U[q][tid] = v[q];
    }
    __syncthreads();
    // the changed entries into their rows' registers: slots 0..3 are rows r0 .. r0 + 3 (first rows of threads r0 .. r0 + 3), a slot
    // 4 + q that is the first to name its row is row piv[r0 + q] -- thread (row mod 512), its (row / 512)-th row: a uniform index
#pragma unroll
    for (int q = 0; q < QN_LU_SUB; ++q)
        if (tid == r0 + q) {
#pragma unroll
            for (int j = 0; j < QN_LU_SUB; ++j) a[j][0] = L.pat[j][q];
        }
#pragma unroll
    for (int q = 0; q < QN_LU_SUB; ++q) {
        const int e = QN_LU_SUB + q;
        if (canon[e] == e) { // (uniform)
            const int row = ix[e], tp = row & (QN_LU_PT - 1), jp = __builtin_amdgcn_readfirstlane(row / QN_LU_PT);
            const bool mine = tid == tp;
#pragma unroll
            for (int jr = 0; jr < RPT; ++jr)
                if (jr == jp) {
#pragma unroll
                    for (int j = 0; j < QN_LU_SUB; ++j) a[j][jr] = mine ? L.pat[j][e] : a[j][jr];
                }
        }
    }
    if constexpr (KEEP) {
#pragma unroll
        for (int jr = 0; jr < RPT; ++jr) {
            const int i = tid + QN_LU_PT * jr;
            if (i >= r0 + QN_LU_SUB && i < m) {
#pragma unroll
                for (int j = 0; j < QN_LU_SUB; ++j)
#pragma unroll
                    for (int q = 0; q < QN_LU_SUB; ++q) a[j][jr] = a[j][jr] - lprev[q][jr] * L.U[q][j]; // (column order: the per-column kernels' rounding)
            }
        }
    } else { // the multipliers from the panel buffer, one row-per-thread at a time (lu_cols_update: larger chunks were slower at 16 rows per thread)
#pragma unroll
        for (int jr = 0; jr < RPT; ++jr) {
            const int i = tid + QN_LU_PT * jr;
            const bool on = i >= r0 + QN_LU_SUB && i < m;
            double l[QN_LU_SUB];
#pragma unroll
            for (int q = 0; q < QN_LU_SUB; ++q) l[q] = on ? lu_ld<true>(P + (size_t)(r0 + q) * pld + i) : 0.0;
            if (on) {
#pragma unroll
                for (int j = 0; j < QN_LU_SUB; ++j)
#pragma unroll
                    for (int q = 0; q < QN_LU_SUB; ++q) a[j][jr] = a[j][jr] - l[q] * L.U[q][j];
            }
        }
    }
}
// role B, second half: the column back to the panel buffer
template <int RPT, bool COH>
__device__ __forceinline__ void lu_col_store(double* __restrict__ P, const size_t pld, const int m, const int r0, const int c0, const double (&a)[QN_LU_SUB][RPT],
                                             const int tid) {
    // Every row of the thread, no condition: rows the update did not touch get back the value that was loaded (nobody else writes
    // this column meanwhile), rows past m lie in the buffer's unused tail (pld = 512 x 16 rows).  Stores under a condition each
    // came with a wait for all memory operations before them: sixteen write-through round trips in a row.
    (void)m; (void)r0;
#pragma unroll
    for (int jr = 0; jr < RPT; ++jr) lu_st<COH>(P + (size_t)c0 * pld + tid + QN_LU_PT * jr, a[0][jr]);
}
// role A, second half: the four pivot steps of sub-panel s on the rows in registers, and the columns back to the buffer.
// Returns true when a pivot column holds no non-zero entry (then *fail = 1).
template <int RPT, bool COH>
__device__ __forceinline__ bool lu_sub_factor(double* __restrict__ P, const size_t pld, const int m, const int s, const int p0, int* __restrict__ piv,
                                              int* __restrict__ fail, QnLuLds& L, double (&a)[QN_LU_SUB][RPT], const int tid, int (&pv)[QN_LU_SUB]) {
    const int lane = tid & 63, wave = tid >> 6;
    const int c0 = QN_LU_SUB * s;
    bool failed = false;
#pragma unroll
    for (int j = 0; j < QN_LU_SUB; ++j) pv[j] = p0 + QN_LU_SUB * s + j;
#pragma unroll
    for (int j = 0; j < QN_LU_SUB; ++j) {
        const int k = QN_LU_SUB * s + j; // pivot position (panel row = panel column)
        double best = -1.0;
        int idx = 0x7fffffff;
#pragma unroll
        for (int jr = 0; jr < RPT; ++jr) {
            const int i = tid + QN_LU_PT * jr;
            const double v = fabs(a[j][jr]);
            if (i >= k && i < m && v > best) { best = v; idx = i; } // ascending i per thread: the first maximum (nalgebra's icamax)
        }
        // (xor butterfly on the VALU data path -- DPP / v_permlane*_swap, qn_xor_lanes -- instead of __shfl_xor: that is three
        // ds_bpermute round trips per level, ~2000 cycles of LDS-pipeline latency inside every pivot step of the chain)
        lu_argmax_lanes<64>(best, idx);
        if (lane == 0) { L.bv[j][wave] = best; L.bi[j][wave] = idx; }
        if (j == 0) QN_LU_STAMP(3);
        __syncthreads();
        if (j == 0) QN_LU_STAMP(4);
        // the waves' candidates: lane w takes wave (w mod 8)'s, three more levels -- ONE trip through LDS (before round 6 a loop over the eight
        // candidates: the compiler, seeing uniform values, read and compared them one after the other, a dependent LDS round trip each)
        static_assert(QN_LU_PT / 64 == 8, "eight waves' candidates");
        best = L.bv[j][lane & 7]; idx = L.bi[j][lane & 7];
        lu_argmax_lanes<8>(best, idx);
        best = lu_uniform_d(best); idx = __builtin_amdgcn_readfirstlane(idx);
        if (!(best > 0.0) || idx >= m) { failed = true; idx = k; } // no non-zero entry in this column: singular (newton/mod.rs:43-46)
        const int p = idx;
        pv[j] = p0 + p; // (stored after the four steps: a store here put a wait for the PREVIOUS step's store -- a write-through round trip -- into every step of the chain)
        if (failed) break; // (uniform: every thread reduced the same sixteen candidates)
        // (Round 4, measured and dropped: ONE barrier per pivot step -- every wave's candidate publishes its row and the reciprocal of
        // its pivot entry together with the candidate, behind the barrier everybody picks the winner and reads the winner's row.  Same
        // bits; 48.2-50.0 ms per Newton iteration at n = 8192 against 44.9: pulling the candidate's row out of sixteen rows per thread
        // in EVERY wave, the dependent LDS read of the winner's row and eight more live registers cost more than the barrier saved.)
        // Rows k and p meet in LDS: their owners publish them, take each other's, and everybody reads the pivot row.  Row k
        // (k < 64) is thread k's first row; row p is found by its one owner under a branch the other waves skip (first version:
        // every thread compared every one of its rows with k and p, twice -- a third of the step).
        const int tp = p & (QN_LU_PT - 1), jp = p / QN_LU_PT; // (uniform)
        double cp[QN_LU_SUB]; // this thread's row in slot jp (selects on a uniform condition: no per-lane compares, no divergent branch)
#pragma unroll
        for (int jj = 0; jj < QN_LU_SUB; ++jj) cp[jj] = a[jj][0];
#pragma unroll
        for (int jr = 1; jr < RPT; ++jr)
            if (jr == jp) {
#pragma unroll
                for (int jj = 0; jj < QN_LU_SUB; ++jj) cp[jj] = a[jj][jr];
            }
        if (tid == k) {
#pragma unroll
            for (int jj = 0; jj < QN_LU_SUB; ++jj) L.rowk[j][jj] = a[jj][0];
        }
        if (tid == tp) {
#pragma unroll
            for (int jj = 0; jj < QN_LU_SUB; ++jj) L.rowp[j][jj] = cp[jj];
        }
        __syncthreads();
        {
            const bool own_p = tid == tp && p != k, own_k = tid == k && p != k;
#pragma unroll
            for (int jr = 0; jr < RPT; ++jr) {
                const bool hit = own_p && jr == jp;
#pragma unroll
                for (int jj = 0; jj < QN_LU_SUB; ++jj) a[jj][jr] = hit ? L.rowk[j][jj] : a[jj][jr];
            }
            // (after the owner of p: when both rows are this thread's, row k ends up with the pivot row)
#pragma unroll
            for (int jj = 0; jj < QN_LU_SUB; ++jj) a[jj][0] = own_k ? L.rowp[j][jj] : a[jj][0];
        }
        if (j == 0) QN_LU_STAMP(5);
        const double inv_ukk = 1.0 / L.rowp[j][j]; // (one division per step and thread: eight of them per thread made a step 3 us on this one CU)
#pragma unroll
        for (int jr = 0; jr < RPT; ++jr) {
            const int i = tid + QN_LU_PT * jr;
            if (i > k && i < m) {
                const double l = a[j][jr] * inv_ukk;
                a[j][jr] = l;
#pragma unroll
                for (int jj = 0; jj < QN_LU_SUB; ++jj)
                    if (jj > j) a[jj][jr] = a[jj][jr] - l * L.rowp[j][jj];
            }
        }
        if (j == 0) QN_LU_STAMP(6);
        if (j == 3) QN_LU_STAMP(7);
    }
    if (tid == 0) {
#pragma unroll
        for (int j = 0; j < QN_LU_SUB; ++j) lu_sti<COH>(piv + p0 + QN_LU_SUB * s + j, pv[j]);
        if (failed) lu_sti<COH>(fail, 1);
    }
    if (tid >= c0 && tid < c0 + QN_LU_SUB) { // rows c0 .. c0 + 3 are these threads' first rows: the sub-panel's own 4 x 4 block
#pragma unroll
        for (int c = 0; c < QN_LU_SUB; ++c) L.l11[tid - c0][c] = a[c][0];
    }
    // (every row of the thread, no condition: see lu_col_store)
#pragma unroll
    for (int jr = 0; jr < RPT; ++jr) {
#pragma unroll
        for (int j = 0; j < QN_LU_SUB; ++j) lu_st<COH>(P + (size_t)(c0 + j) * pld + tid + QN_LU_PT * jr, a[j][jr]);
    }
    return failed;
}

template <int RPT>
__global__ __launch_bounds__(QN_LU_PT) void lu_panel_step_kernel(double* __restrict__ P, size_t pld, int m, int s, int p0, int* __restrict__ piv,
                                                                 int* __restrict__ fail) {
    QN_LU_STAMP(0);
    if (*fail) return;
    QN_LU_STAMP(1);
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    if (b == 1) { // role C: the swaps of sub-panel s - 1 on the columns left of it
        if (s >= 2) lu_role_c<false>(P, pld, QN_LU_SUB * (s - 1), p0, piv, tid);
        return;
    }
    const bool isA = b == 0;
    if (isA && s >= QN_NB / QN_LU_SUB) return;
    const int c0 = isA ? QN_LU_SUB * s : QN_LU_SUB * (s + 1) + (b - 2); // first column of this workgroup
    const int nc = isA ? QN_LU_SUB : 1;
    __shared__ QnLuLds L;
    const int r0 = QN_LU_SUB * (s - 1);
    double a[QN_LU_SUB][RPT];
    lu_cols_update<RPT, false, (RPT >= 16 ? 1 : 4)>(P, pld, m, r0, s >= 1, c0, nc, p0, piv, L, a, tid);
    if (!isA) { lu_col_store<RPT, false>(P, pld, m, r0, c0, a, tid); return; }
    QN_LU_STAMP(2);
    int pv_unused[QN_LU_SUB];
    (void)lu_sub_factor<RPT, false>(P, pld, m, s, p0, piv, fail, L, a, tid, pv_unused);
    QN_LU_STAMP(8);
}

// ---- the panel in ONE launch (round 4) ----
// 17 launches per panel are 16 kernel boundaries inside a dependent chain: ~4.6 us each between the last store of one and the first
// load of the next, 2176 of them per factorisation at n = 8192 (10 ms of the chain's 41).  Here the same roles on the same columns
// in the same order -- workgroup 0 factors the sub-panels one after another, workgroup 2 + j keeps column 8 + j up to date until it
// becomes part of a sub-panel, workgroup 1 replays the swaps on the finished columns -- but as ONE grid of 58 workgroups that wait
// for each other on counters in memory instead of on kernel boundaries:
//     sync[0]      = base + s + 1  once sub-panel s and its pivots are stored (workgroup 0)
//     sync[1 + c]  = base + s      once column c carries the update of sub-panel s - 1 (its workgroup)
// (base = 32 x panel index: the counters only grow, nothing is reset between panels).  The workgroups sit on different XCDs -- different
// L2s -- so inside this kernel EVERY access to the panel buffer, the pivots and the counters is a relaxed atomic at agent scope (sc1
// loads and write-through stores, coherent per location by the memory model), a producer waits for its stores (s_waitcnt vmcnt(0))
// and the workgroup barrier before it raises its counter, and no cache is flushed or invalidated.  (First version: plain accesses
// between agent-scope release / acquire fences -- correct, and 565 us per panel against 458 with 17 launches: every release writes
// the whole L2 of its XCD back, the trailing update's dirty lines included, 870 times per panel.)  Who waits for whom: a column for workgroup 0 only; workgroup 0,
// at sub-panel s, for the four columns it is about to take -- the lowest-numbered columns still alive; workgroup 1 for both, and
// nobody for workgroup 1.  Workgroups are placed in index order, so whoever is waited for is resident or next in line: the grid needs
// three free CUs to finish, 58 to run as intended (the look-ahead leaves the panel chain 64, qn_hip.hip).  Every wait is BOUNDED: after
// QN_LU_SPIN_MAX polls it gives up with *fail = 2, everybody else leaves at the next poll, and the host runs the factorisation again
// with one launch per sub-panel (two solvers' panels sharing the free CUs could otherwise wait for each other's unplaced workgroups).
#define QN_LU_SPIN_MAX (1 << 20)
__device__ __forceinline__ bool lu_wait_ge(const int* flag, const int target, int* fail, const int spin_max) {
    for (int spin = 0; spin < spin_max; ++spin) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}
// release: this thread's stores have left for the coherence point; acquire: nothing to do -- every load that follows goes there itself
__device__ __forceinline__ void lu_release_all() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
template <int RPT>
__global__ __launch_bounds__(QN_LU_PT) void lu_panel_persist_kernel(double* __restrict__ P, size_t pld, int m, int p0, int* __restrict__ piv, int* __restrict__ fail,
                                                                    int* __restrict__ sync, int base, int spin_max) { // (spin_max: QN_LU_SPIN_MAX; 0 in the test of the fallback)
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    constexpr int NSUB = QN_NB / QN_LU_SUB;
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    __shared__ QnLuLds L;
    if (b == 0) { // role A: sub-panels 0 .. 15
        constexpr bool KEEP = RPT <= 8; // the previous sub-panel's multipliers stay in registers (at 16 rows per thread there is no room)
        double lprev[QN_LU_SUB][KEEP ? RPT : 1];
        int pvprev[QN_LU_SUB] = {0, 0, 0, 0};
        for (int s = 0; s < NSUB; ++s) {
            QN_LU_STAMP(0);
            if (s >= 2) { // its four columns carry sub-panel s - 2 (their workgroups' last step)
                const bool ok = tid < QN_LU_SUB ? lu_wait_ge(sync + 1 + QN_LU_SUB * s + tid, base + s - 1, fail, spin_max) : true;
                if (!__syncthreads_and(ok)) return;
            } else __syncthreads(); // (L.l11 of the previous sub-panel is complete)
            QN_LU_STAMP(1);
            double a[QN_LU_SUB][RPT];
            lu_cols_update_pre<RPT, KEEP>(P, pld, m, QN_LU_SUB * (s - 1), s >= 1, QN_LU_SUB * s, p0, pvprev, L, lprev, a, tid);
            if constexpr (KEEP) {
                if (s >= 1) { // sub-panel s - 1 is announced only now: its stores completed while the new columns came in
                    lu_release_all();
                    if (tid == 0) __hip_atomic_store(sync, base + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            QN_LU_STAMP(2);
            const bool failed = lu_sub_factor<RPT, true>(P, pld, m, s, p0, piv, fail, L, a, tid, pvprev);
            if (failed) return; // (*fail = 1: the others leave at their next poll)
            QN_LU_STAMP(8);
            if constexpr (KEEP) {
#pragma unroll
                for (int q = 0; q < QN_LU_SUB; ++q)
#pragma unroll
                    for (int jr = 0; jr < RPT; ++jr) lprev[q][jr] = a[q][jr];
                if (s == NSUB - 1) {
                    lu_release_all();
                    if (tid == 0) __hip_atomic_store(sync, base + s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                lu_release_all(); // (also: the next sub-panel's update reads what other threads of this workgroup stored)
                QN_LU_STAMP(9);
                if (tid == 0) __hip_atomic_store(sync, base + s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (b == 1) { // role C: sub-panel sg - 1's swaps on the columns left of it, once nobody reads those in the old row order
        for (int sg = 2; sg <= NSUB; ++sg) {
            bool ok = true;
            if (tid == 0) ok = lu_wait_ge(sync, base + sg, fail, spin_max);
            else if (tid < 1 + QN_NB && tid - 1 >= QN_LU_SUB * sg && tid - 1 >= 2 * QN_LU_SUB) ok = lu_wait_ge(sync + tid, base + sg - 1, fail, spin_max);
            if (!__syncthreads_and(ok)) return;
            lu_role_c<true>(P, pld, QN_LU_SUB * (sg - 1), p0, piv, tid);
        }
        return;
    }
    // role B: column c, sub-panels 0 .. c / 4 - 2
    const int c = 2 * QN_LU_SUB + (b - 2);
    const int last = c / QN_LU_SUB - 1;
    for (int sg = 1; sg <= last; ++sg) {
        const bool ok = tid == 0 ? lu_wait_ge(sync, base + sg, fail, spin_max) : true;
        if (!__syncthreads_and(ok)) return;
        double a[QN_LU_SUB][RPT];
        const int r0 = QN_LU_SUB * (sg - 1);
        lu_cols_update<RPT, true, (RPT >= 16 ? 4 : 16)>(P, pld, m, r0, true, c, 1, p0, piv, L, a, tid);
        lu_col_store<RPT, true>(P, pld, m, r0, c, a, tid);
        lu_release_all();
        if (tid == 0) __hip_atomic_store(sync + 1 + c, base + sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// ---- the same three steps on a RANGE of columns (round 4: look-ahead, enqueue_newton_lu) ----
// After a panel is factorised only the next panel's 64 columns are needed at once; everything right of them -- the bulk of the
// trailing update, MFMA-bound -- and the swaps on the finished columns left of the panel can run on a second stream beside the
// next panel's chain of step launches.  Same arithmetic per entry as lu_swap_rows / lu_trsm / lu_gemm: the factors are the same bits.
__global__ __launch_bounds__(256) void lu_swap_rows2_kernel(double* __restrict__ W, size_t ld, int p0, int col_lo, int col_hi, const int* __restrict__ piv,
                                                            const int* __restrict__ fail) {
    if (*fail) return;
    __shared__ int pv[64];
    if (threadIdx.x < 64) pv[threadIdx.x] = piv[p0 + threadIdx.x];
    __syncthreads();
    for (int j = col_lo + blockIdx.x * blockDim.x + threadIdx.x; j < col_hi; j += gridDim.x * blockDim.x) {
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
// U[p0 .. p0 + 63][j] = L11^-1 A[p0 ..][j] for the columns col_lo <= j < col_hi (unit lower L11 = the diagonal block at p0).
// lu_trsm_kernel keeps a column in ONE thread's registers and walks L11 out of LDS: 412 registers, one wave per SIMD, 2016
// multipliers each an LDS round trip nothing hides -- 35 us whatever the number of columns, and with the look-ahead that is 35 us
// of the CHAIN per panel.  Here a column lives ACROSS a wave, lane r holding x[r]: step c broadcasts x[c] (v_readlane) and every lane
// r > c subtracts L[r][c] x[c] -- per entry the same products subtracted in the same order, so the bits are lu_trsm_kernel's
// (tests/test_gpu_newton.py) -- 63 short steps instead of 2016 dependent ones.  L11 sits transposed in LDS (lane r reads LT[c][r]:
// contiguous), CPW columns per wave share each read.  (Also tried: a column per thread again but row by row with the multipliers
// by scalar loads -- 60-75 us, the scalar loads' address arithmetic and latency in front of every row.)
template <int CPW>
__global__ __launch_bounds__(256) void lu_trsm2_kernel(double* __restrict__ W, size_t ld, int p0, int col_lo, int col_hi, const int* __restrict__ fail) {
    if (*fail) return;
    __shared__ double LT[QN_NB][QN_NB + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = col_lo + (blockIdx.x * 4 + wave) * CPW;
    double x[CPW];
#pragma unroll
    for (int q = 0; q < CPW; ++q) x[q] = W[(size_t)(p0 + lane) * ld + min(j0 + q, col_hi - 1)];
    qn_tile_to_lds<256, true>(LT, W + (size_t)p0 * ld + p0, ld); // LT[c][r] = L11[r][c]
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 63; ++c) {
        const double l = LT[c][lane];
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            const double t = x[q] - l * qn_readlane_d(x[q], c);
            x[q] = (lane > c) ? t : x[q];
        }
    }
    if (lane >= 1) {
#pragma unroll
        for (int q = 0; q < CPW; ++q)
            if (j0 + q < col_hi) W[(size_t)(p0 + lane) * ld + j0 + q] = x[q];
    }
}
// C[p0 + 64 + 64 by ..][col_lo + 64 bx ..] -= L21 U12 of the panel at p0 (depth 64) on the f64 matrix cores: lu_gemm_kernel's tile, for
// the tiles t = blockIdx.x, blockIdx.x + gridDim.x, ... < ntiles of an ncb-wide grid of tiles (t = by ncb + bx) from column col_lo on.
// (The loop: a grid that is resident at once -- two workgroups per CU of the stream's mask -- instead of 16 k workgroups was tried for
// the bulk, on the suspicion that a queue still placing workgroups holds up the other stream's launches.  With the CU mask the chain's
// launches find their CUs either way: 54.1 ms looping, 53.5 ms as a plain grid, which is the default -- QN_LU_BULK_PERSIST.)
__global__ __launch_bounds__(256) void lu_gemm2_kernel(double* __restrict__ W, size_t ld, int p0, int col_lo, int ncb, int ntiles, const int* __restrict__ fail,
                                                       int chain) {
    if (chain) __builtin_amdgcn_s_setprio(3); // (a link of the panel chain, beside the bulk on the other stream: see chol_syrk_kernel)
    if (*fail) return;
    __shared__ double PI[QN_NB][QN_NB + 1]; // PI[k][i] = L21[i0 + i][p0 + k]
    __shared__ double PJ[QN_NB][QN_NB + 1]; // PJ[k][j] = U12[p0 + k][j0 + j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int by = t / ncb, bx = t - by * ncb;
        const int i0 = p0 + 64 + by * 64, j0 = col_lo + bx * 64;
        qn_tile_to_lds<256, true>(PI, W + (size_t)i0 * ld + p0, ld);
        qn_tile_to_lds<256, false>(PJ, W + (size_t)p0 * ld + j0, ld);
        v4d acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
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
        __syncthreads(); // (the panels in LDS are refilled by the next tile)
    }
}

// ---- the look-ahead columns in TWO launches that read the panel from its column-major buffer (round 4) ----
// Between two panels the chain ran: store the panel back (P -> W), swaps / U12 solve / update on the next panel's 64 columns, load
// them into the buffer -- five launches, 46 us.  Now the factorised panel STAYS in its buffer for the chain's purposes (the copy back
// into W moves to the second stream, in front of the bulk that needs it there):
//   lu_la_swap_trsm_kernel  one wave per look-ahead column: the panel's 64 row swaps replayed on the column in registers (the block
//                           row in the wave's lanes, the rows further down in one slot per pivot), then lu_trsm2_kernel's 63 steps
//                           with L11 taken from the buffer;
//   lu_la_gemm_kernel       lu_gemm2_kernel's tile with L21 taken from the buffer, the result written to W AND, transposed through
//                           LDS, into the OTHER buffer as the next panel's columns (lu_panel_load_kernel's job).
// Same operations in the same order on every entry: the bits of the five launches.
__global__ __launch_bounds__(256) void lu_la_swap_trsm_kernel(double* __restrict__ W, size_t ld, int p0, int col_lo, int col_hi, const double* __restrict__ P, size_t pld,
                                                              const int* __restrict__ piv, const int* __restrict__ fail) {
    __builtin_amdgcn_s_setprio(3);
    if (*fail) return;
    __shared__ double LT[QN_NB][QN_NB + 1]; // LT[c][r] = L11[r][c] = P[c][r]
    __shared__ int pv[QN_NB], canon[QN_NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < QN_NB * QN_NB; e += 256) LT[e >> 6][e & 63] = P[(size_t)(e >> 6) * pld + (e & 63)];
    if (tid < QN_NB) pv[tid] = piv[p0 + tid];
    __syncthreads();
    if (tid < QN_NB) { // the first pivot step that names the same row (outside the block row: that step's slot holds the row's value)
        int cq = tid;
        for (int q = tid - 1; q >= 0; --q)
            if (pv[q] == pv[tid]) cq = q;
        canon[tid] = cq;
    }
    __syncthreads();
    const int j = col_lo + blockIdx.x * 4 + wave;
    if (j >= col_hi) return; // (uniform per wave; no barrier below)
    const int myp = pv[lane], mycanon = canon[lane];
    const bool outside = myp >= p0 + QN_NB;
    double x = W[(size_t)(p0 + lane) * ld + j];                 // row p0 + lane
    double y = outside ? W[(size_t)myp * ld + j] : 0.0;          // slot `lane`: row piv[lane] when it lies below the block row
#pragma unroll
    for (int q = 0; q < QN_NB; ++q) {                            // (uniform: step q's pivot and slot by v_readlane from the lanes that hold them)
        const int p = __builtin_amdgcn_readlane(myp, q);
        if (p != p0 + q) {
            const double a = qn_readlane_d(x, q);
            if (p < p0 + QN_NB) {
                const double b = qn_readlane_d(x, p - p0);
                x = (lane == q) ? b : ((lane == p - p0) ? a : x);
            } else {
                const int cq = __builtin_amdgcn_readlane(mycanon, q);
                const double b = qn_readlane_d(y, cq);
                x = (lane == q) ? b : x;
                y = (lane == cq) ? a : y;
            }
        }
    }
    if (outside && mycanon == lane) W[(size_t)myp * ld + j] = y;
#pragma unroll
    for (int c = 0; c < 63; ++c) {
        const double l = LT[c][lane];
        const double t = x - l * qn_readlane_d(x, c);
        x = (lane > c) ? t : x;
    }
    W[(size_t)(p0 + lane) * ld + j] = x;
}
__global__ __launch_bounds__(256) void lu_la_gemm_kernel(double* __restrict__ W, size_t ld, int p0, int col_lo, const double* __restrict__ P, size_t pld,
                                                         double* __restrict__ Pn, const int* __restrict__ fail) {
    __builtin_amdgcn_s_setprio(3);
    if (*fail) return;
    __shared__ double PI[QN_NB][QN_NB + 1]; // PI[k][i] = L21[i0 + i][p0 + k] = P[k][i0 - p0 + i]
    __shared__ double PJ[QN_NB][QN_NB + 1]; // PJ[k][j] = U12[p0 + k][col_lo + j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i0 = p0 + 64 + blockIdx.x * 64, j0 = col_lo;
    for (int e = tid; e < QN_NB * QN_NB; e += 256) PI[e >> 6][e & 63] = P[(size_t)(e >> 6) * pld + (i0 - p0) + (e & 63)];
    qn_tile_to_lds<256, false>(PJ, W + (size_t)p0 * ld + j0, ld);
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                acc[a][b][reg] = -W[(size_t)(i0 + wi + a * 16 + l4 + 4 * reg) * ld + j0 + wj + b * 16 + l15];
    __syncthreads();
    qn_mfma_64(PI, PJ, QN_NB, wi, wj, lane, acc);
    __syncthreads(); // (PI is reused for the transposed result)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const double v = -acc[a][b][reg];
                W[(size_t)(i0 + wi + a * 16 + l4 + 4 * reg) * ld + j0 + wj + b * 16 + l15] = v;
                PI[wj + b * 16 + l15][wi + a * 16 + l4 + 4 * reg] = v; // [column][row]
            }
    __syncthreads();
    // the next panel starts at row and column p0 + 64: its buffer row of global row i is i - (p0 + 64)
    for (int e = tid; e < QN_NB * QN_NB; e += 256) Pn[(size_t)(e >> 6) * pld + (size_t)(i0 - p0 - 64) + (e & 63)] = PI[e >> 6][e & 63];
}

// x[i] = sign * b[perm[i]] (zero past n_src): the row permutation of the factorisation applied to a right-hand side
// (`fill`: the vector the one-launch sweep that follows will write -- set to the sentinel its consumers wait on, lu_sweep_kernel)
__global__ void lu_vec_perm_kernel(double* __restrict__ dst, const double* __restrict__ src, const int* __restrict__ perm, int n_src, int n_dst,
                                   double sign, double* __restrict__ fill) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_dst; i += gridDim.x * blockDim.x) {
        const int p = perm[i];
        dst[i] = (p < n_src) ? sign * src[p] : 0.0;
        if (fill) fill[i] = __longlong_as_double((long long)0x7ff8c0de5eed1e55ull);
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
        p = qn_wave_sum(p); // (xor 32, 16, ... 1 on the VALU data path: the same order as the __shfl_xor loop it replaces)
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
        p = qn_wave_sum(p); // (xor 32, 16, ... 1 on the VALU data path: the same order as the __shfl_xor loop it replaces)
        if (lane == 0) rhs[r] = rhs[r] - p;
    }
}

// ---- a whole substitution sweep in ONE launch (round 4) ----
// The 128 step launches of a sweep are a dependent chain of 10-13 us links (launch, the diagonal block into LDS, 63 readlane steps,
// the update of the rows below), 512 of them per Newton iteration: 6 ms.  Here workgroup j owns block row w (forward: w = j, backward:
// w = nb - 1 - j), keeps its 64 running right-hand-side entries in registers, takes the solution blocks x_k of its predecessors as
// they appear, subtracts W[w, k] x_k in the SAME order and with the same wave sums as the step kernels (row r: blocks k in sweep
// order, each a product per lane and qn_wave_sum), then solves its own diagonal block as they do: the same bits.
// How a block "appears": the solution vector is filled with a SENTINEL (a quiet NaN with a payload no arithmetic produces) before
// the sweep, a solution entry is ONE 8-byte store, and a consumer's lane simply re-reads its entry (sc1: past the non-coherent cache
// levels) until it is not the sentinel -- the datum is its own flag: no counter, no fence, one memory round trip per link instead of
// two (first version: a counter raised after the block was stored -- 7.8 us per link, 4 ms per Newton iteration for the four sweeps).
// The waves of a workgroup poll independently (no barrier inside the loop); the matrix rows of the next predecessor are requested
// before the poll.  Who fills: the kernel that writes the right-hand side fills the sweep's output with sentinels
// (lu_vec_perm_kernel), and a sweep workgroup that has taken its 64 right-hand-side entries puts sentinels in their place -- the
// vector it read is the NEXT sweep's output.  A workgroup waits for lower-numbered workgroups only (placed before it); every wait
// is bounded as in the panel kernel (*fail = 2, the host runs the factorisation again with step launches).
#define QN_LU_SENTINEL 0x7ff8c0de5eed1e55ull
__device__ __forceinline__ double lu_sentinel() { return __longlong_as_double((long long)QN_LU_SENTINEL); }
__device__ __forceinline__ bool lu_is_sentinel(const double v) { return (unsigned long long)__double_as_longlong(v) == QN_LU_SENTINEL; }
template <bool BWD>
__global__ __launch_bounds__(256) void lu_sweep_kernel(const double* __restrict__ W, size_t ld, int nb, double* __restrict__ rhs, double* __restrict__ sol,
                                                       int* __restrict__ fail, int spin_max) {
    __shared__ double D[QN_NB][QN_NB + 1];
    __shared__ double accs[QN_NB];
    __shared__ int bail_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.x, w = BWD ? nb - 1 - j : j;
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    if (tid == 0) bail_s = 0;
    qn_tile_to_lds<256, false>(D, W + (size_t)w * QN_NB * ld + (size_t)w * QN_NB, ld);
    double acc[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) acc[jj] = rhs[w * QN_NB + wave + 4 * jj];
    const double* __restrict__ rowp = W + (size_t)(w * QN_NB + wave) * ld + lane; // row wave + 4 jj: + 4 jj ld
    double m[16];
    if (j > 0) {
        const int k = BWD ? nb - 1 : 0;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) m[jj] = rowp[(size_t)(4 * jj) * ld + (size_t)k * QN_NB];
    }
    __syncthreads(); // (every wave has its right-hand-side entries: their places can take the sentinel; bail_s is set; D is complete)
    if (tid < QN_NB) lu_st<true>(rhs + w * QN_NB + tid, lu_sentinel());
    // wave 0's row of the diagonal block into registers NOW, while the predecessors are still at work: the 63 substitution steps at
    // the end of the link then run on registers (an LDS round trip per dependent step before: 5.6 us per forward link, 3.x after)
    double dr[QN_NB];
#pragma unroll
    for (int c = 0; c < QN_NB; ++c) dr[c] = (wave == 0) ? D[lane][c] : 0.0;
    double dd[BWD ? QN_NB : 1]; // (backward: the divisors D[c][c], the same for every lane)
    if (BWD) {
#pragma unroll
        for (int c = 0; c < QN_NB; ++c) dd[c] = D[c][c];
    }
    bool bail = false;
    for (int q = 0; q < j; ++q) {
        const int k = BWD ? nb - 1 - q : q;
        double xl = lu_ld<true>(sol + k * QN_NB + lane);
        for (int spin = 0; __any(lu_is_sentinel(xl)); ++spin) {
            if (spin >= spin_max) { if (lane == 0) __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bail = true; break; }
            if ((spin & 15) == 15 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { bail = true; break; }
            __builtin_amdgcn_s_sleep(1);
            xl = lu_ld<true>(sol + k * QN_NB + lane);
        }
        if (bail) break;
        double mc[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) mc[jj] = m[jj];
        if (q + 1 < j) { // the next predecessor's part of this wave's rows: in flight during the products and the next poll
            const int kn = BWD ? k - 1 : k + 1;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) m[jj] = rowp[(size_t)(4 * jj) * ld + (size_t)kn * QN_NB];
        }
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            double p = mc[jj] * xl;
            p = qn_wave_sum(p);
            acc[jj] = acc[jj] - p;
        }
    }
    if (bail && lane == 0) bail_s = 1;
    if (lane == 0) {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) accs[wave + 4 * jj] = acc[jj];
    }
    __syncthreads();
    if (bail_s) return; // (nothing is published: the successors give up at their own bound or see *fail)
    if (wave == 0) {
        double v = accs[lane];
        if (!BWD) {
#pragma unroll
            for (int c = 0; c < 63; ++c) {
                const double xc = qn_readlane_d(v, c);
                if (lane > c) v = v - dr[c] * xc;
            }
        } else {
#pragma unroll
            for (int c = 63; c >= 0; --c) {
                const double zc = qn_readlane_d(v, c) / dd[BWD ? c : 0];
                if (lane == c) v = zc;
                if (lane < c) v = v - dr[c] * zc;
            }
        }
        if (lu_is_sentinel(v)) v = __longlong_as_double(0x7ff8000000000000ll); // (cannot come out of arithmetic; if it ever did, it must not read as "not yet")
        lu_st<true>(sol + w * QN_NB + lane, v);
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
