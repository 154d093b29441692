// qn_lu_split.hip.h -- the pivoted LU's panel chain (qn_lu.hip.h, newton/mod.rs:31-40) with ROLE A SPLIT OVER THE WORKGROUPS OF ONE XCD (round 6).
//
// lu_panel_persist_kernel's role A is ONE CU working through the panel's 64 dependent pivot steps with every row of the panel in
// its registers: at 7552 rows a sub-panel of four columns costs it 10 us of column traffic (a single CU's memory pipeline), 4 x 2.8 us
// of pivot steps (sixteen rows per thread to search and to eliminate) and 4.5 us until its stores are through -- 26 us, 16 times
// per panel (profiles/r04_k_*).  Here G workgroups share role A: the panel's rows are dealt to them in chunks of 512 (row i belongs
// to part (i / 512) mod G, thread i mod 512, slot (i / 512) / G), so a part holds at most four rows per thread, loads a G-th of the
// columns, and keeps the previous sub-panel's multipliers in registers at every panel height.  What the parts owe each other in a
// pivot step travels as ONE record per part and step through memory:
//     word 0, 1     the part's candidate: |value| and row index of the first maximal entry among its rows
//     word 2..5     the candidate's row (the sub-panel's four columns)
//     word 6..9     part 0 only: row k, the row the pivot row changes places with
//     word 10..12   part 0, fourth step only: the finished 4 x 4 block's multipliers below the diagonal (rows 1 and 2; row 3 is the pivot row's)
// Every part reads all G records, picks the winner by the rule the one-workgroup search applies between its waves (larger value, then
// smaller index: the first row of maximal magnitude, nalgebra's icamax) and has the pivot row and row k at once -- one hop through
// memory per pivot step (tools/hop_probe.hip: 0.38-0.41 us between workgroups of one XCD, 0.54-0.57 us between XCDs, relaxed
// atomics at agent scope), no flag: the words start out as a sentinel (all ones) and the readers' lanes poll until their word is
// something else -- the datum is its own flag, as in lu_sweep_kernel.  The records of a panel are written once; the area of the NEXT
// panel is reset by role C's workgroup of this launch (two areas, panel parity).  A matrix that holds the sentinel's bit pattern
// makes a wait expire: like every bounded wait here that ends in *fail = 2 and the host runs the factorisation launch by launch.
// The parts are workgroups 0, 8, 16, 24 of the grid -- dealt to the same XCD -- and wait for each other: all G must be resident
// (the look-ahead's CU mask leaves the chain 64 CUs; otherwise a wait expires and the fallback runs).
// Per entry the same operations in the same order as lu_sub_factor / lu_cols_update_pre: the factors are the same bits
// (tests/test_gpu_newton.py: test_panel_lu_equals_the_per_column_lu_bit_for_bit).
#pragma once

#define QN_LUS_W 16 // words per record
#define QN_LUS_MAXG 4
#define QN_LUS_SENT 0xffffffffffffffffull
#define QN_LUS_REC_WORDS (QN_NB * QN_LUS_MAXG * QN_LUS_W) // one panel's records

struct QnLusLds {
    double U[QN_LU_SUB][QN_LU_SUB];
    double bv[QN_LU_SUB][QN_LU_PT / 64];
    int bi[QN_LU_SUB][QN_LU_PT / 64];
    double pat[QN_LU_SUB][2 * QN_LU_SUB];
    double l11[QN_LU_SUB][QN_LU_SUB];
    unsigned long long rec[QN_LU_SUB][QN_LUS_MAXG * QN_LUS_W]; // a step's records as wave 0 saw them
    int dead[QN_LU_SUB];
};

__device__ __forceinline__ void lus_st(unsigned long long* p, const double v) {
    __hip_atomic_store(p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lus_sti(unsigned long long* p, const int v) {
    __hip_atomic_store(p, (unsigned long long)(unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double lus_d(const unsigned long long v) { return __longlong_as_double((long long)v); }

// the new sub-panel's columns into registers, brought up to date with the sub-panel at r0 (lu_cols_update_pre<RPT, KEEP = true> on this part's rows)
template <int RPTL, int G>
__device__ __forceinline__ void lus_cols_update_pre(double* __restrict__ P, const size_t pld, const int m, const int r0, const bool prev, const int c0, const int p0,
                                                    const int (&pvprev)[QN_LU_SUB], QnLusLds& L, const double (&lprev)[QN_LU_SUB][RPTL],
                                                    double (&a)[QN_LU_SUB][RPTL], const int tid, const int g) {
    int ix[2 * QN_LU_SUB];
    double v[2 * QN_LU_SUB];
#pragma unroll
    for (int q = 0; q < QN_LU_SUB; ++q) { ix[q] = r0 + q; ix[QN_LU_SUB + q] = pvprev[q] - p0; }
    const bool solver = prev && tid < QN_LU_SUB; // thread j: column c0 + j (every part solves the 4 x 4 system for itself: the entries come from memory)
    {
        const double* col = P + (size_t)(c0 + (tid & (QN_LU_SUB - 1))) * pld;
#pragma unroll
        for (int e = 0; e < 2 * QN_LU_SUB; ++e) v[e] = solver ? lu_ld<true>(col + ix[e]) : 0.0;
    }
#pragma unroll
    for (int jr = 0; jr < RPTL; ++jr) {
        const int i = tid + QN_LU_PT * (jr * G + g);
#pragma unroll
        for (int j = 0; j < QN_LU_SUB; ++j) a[j][jr] = (i < m) ? lu_ld<true>(P + (size_t)(c0 + j) * pld + i) : 0.0;
    }
    if (!prev) return; // (uniform)
    int canon[2 * QN_LU_SUB];
#pragma unroll
    for (int e = 0; e < 2 * QN_LU_SUB; ++e) {
        canon[e] = e;
#pragma unroll
        for (int f = 2 * QN_LU_SUB - 1; f >= 0; --f)
            if (f < e && ix[f] == ix[e]) canon[e] = f;
    }
    if (solver) {
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
        for (int c = 0; c < QN_LU_SUB - 1; ++c)
#pragma unroll
            for (int r = c + 1; r < QN_LU_SUB; ++r) v[r] = v[r] - L.l11[r][c] * v[c];
#pragma unroll
        for (int e = 0; e < 2 * QN_LU_SUB; ++e) L.pat[tid][e] = v[e];
#pragma unroll
        for (int q = 0; q < QN_LU_SUB; ++q) L.U[q][tid] = v[q];
    }
    __syncthreads();
    // the changed entries into their rows' registers: rows r0 .. r0 + 3 are part 0's (chunk 0, first slot of threads r0 .. r0 + 3), a
    // pivot row belongs to part (row / 512) mod G
    if (g == 0) {
#pragma unroll
        for (int q = 0; q < QN_LU_SUB; ++q)
            if (tid == r0 + q) {
#pragma unroll
                for (int j = 0; j < QN_LU_SUB; ++j) a[j][0] = L.pat[j][q];
            }
    }
#pragma unroll
    for (int q = 0; q < QN_LU_SUB; ++q) {
        const int e = QN_LU_SUB + q;
        if (canon[e] == e) { // (uniform)
            const int row = ix[e], chunk = __builtin_amdgcn_readfirstlane(row / QN_LU_PT), tp = row & (QN_LU_PT - 1);
            if (chunk % G == g) { // (uniform)
                const int jp = chunk / G;
                const bool mine = tid == tp;
#pragma unroll
                for (int jr = 0; jr < RPTL; ++jr)
                    if (jr == jp) {
#pragma unroll
                        for (int j = 0; j < QN_LU_SUB; ++j) a[j][jr] = mine ? L.pat[j][e] : a[j][jr];
                    }
            }
        }
    }
#pragma unroll
    for (int jr = 0; jr < RPTL; ++jr) {
        const int i = tid + QN_LU_PT * (jr * G + g);
        if (i >= r0 + QN_LU_SUB && i < m) {
#pragma unroll
            for (int j = 0; j < QN_LU_SUB; ++j)
#pragma unroll
                for (int q = 0; q < QN_LU_SUB; ++q) a[j][jr] = a[j][jr] - lprev[q][jr] * L.U[q][j]; // (column order: the per-column kernels' rounding)
        }
    }
}

// the four pivot steps of sub-panel s on this part's rows, and its rows of the four columns back to the buffer.
// Returns 0, 1 (a pivot column without a non-zero entry: *fail = 1) or 2 (a wait expired: *fail = 2).
template <int RPTL, int G>
__device__ __forceinline__ int lus_sub_factor(double* __restrict__ P, const size_t pld, const int m, const int s, const int p0, int* __restrict__ piv,
                                              int* __restrict__ fail, unsigned long long* __restrict__ rec, QnLusLds& L, double (&a)[QN_LU_SUB][RPTL], const int tid,
                                              const int g, int (&pv)[QN_LU_SUB], const int spin_max) {
    const int lane = tid & 63, wave = tid >> 6;
    const int c0 = QN_LU_SUB * s;
    int status = 0;
    double rowp3[QN_LU_SUB] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < QN_LU_SUB; ++j) pv[j] = p0 + QN_LU_SUB * s + j;
#pragma unroll
    for (int j = 0; j < QN_LU_SUB; ++j) {
        const int k = QN_LU_SUB * s + j; // pivot position (panel row = panel column)
        double best = -1.0;
        int idx = 0x7fffffff;
#pragma unroll
        for (int jr = 0; jr < RPTL; ++jr) {
            const int i = tid + QN_LU_PT * (jr * G + g);
            const double v = fabs(a[j][jr]);
            if (i >= k && i < m && v > best) { best = v; idx = i; } // ascending i per thread: the first maximum (nalgebra's icamax)
        }
        lu_argmax_lanes<64>(best, idx);
        if (lane == 0) { L.bv[j][wave] = best; L.bi[j][wave] = idx; }
        if (j == 0) QN_LU_STAMP(3);
        __syncthreads();
        if (j == 0) QN_LU_STAMP(4);
        best = L.bv[j][lane & 7]; idx = L.bi[j][lane & 7]; // (lane-parallel: lu_sub_factor)
        lu_argmax_lanes<8>(best, idx);
        best = lu_uniform_d(best); idx = __builtin_amdgcn_readfirstlane(idx);
        // this part's candidate, its row, and (part 0) row k and the finished block's multipliers: the record
        unsigned long long* const r = rec + ((size_t)k * G + g) * QN_LUS_W;
        {
            const bool have = idx != 0x7fffffff;
            const int to = have ? (idx & (QN_LU_PT - 1)) : 0, jl = have ? (idx / QN_LU_PT) / G : 0; // (uniform)
            double cp[QN_LU_SUB];
#pragma unroll
            for (int jj = 0; jj < QN_LU_SUB; ++jj) cp[jj] = a[jj][0];
#pragma unroll
            for (int jr = 1; jr < RPTL; ++jr)
                if (jr == jl) {
#pragma unroll
                    for (int jj = 0; jj < QN_LU_SUB; ++jj) cp[jj] = a[jj][jr];
                }
            if (tid == to) {
                lus_st(r + 0, best);
                lus_sti(r + 1, idx);
#pragma unroll
                for (int jj = 0; jj < QN_LU_SUB; ++jj) lus_st(r + 2 + jj, cp[jj]);
            }
            if (g == 0) {
                if (tid == k) {
#pragma unroll
                    for (int jj = 0; jj < QN_LU_SUB; ++jj) lus_st(r + 6 + jj, a[jj][0]);
                }
                if (j == QN_LU_SUB - 1) { // rows c0 + 1, c0 + 2 are final since steps 1 and 2
                    if (tid == c0 + 1) lus_st(r + 10, a[0][0]);
                    if (tid == c0 + 2) { lus_st(r + 11, a[0][0]); lus_st(r + 12, a[1][0]); }
                }
            }
        }
        if (j == 0) QN_LU_STAMP(10);
        if (wave == 0) { // all G records of this step, lane = word
            constexpr int NW = G * QN_LUS_W;
            const int gl = lane / QN_LUS_W, w = lane % QN_LUS_W;
            const bool need = lane < NW && (w < 6 || (gl == 0 && w < 10) || (gl == 0 && j == QN_LU_SUB - 1 && w < 13));
            const unsigned long long* src = rec + (size_t)k * G * QN_LUS_W + (lane < NW ? lane : 0);
            unsigned long long v;
            int dead = 0;
            for (int spin = 0;; ++spin) {
                v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!need || v != QN_LUS_SENT)) break;
                if (spin >= spin_max) { dead = 2; break; }
                if ((spin & 31) == 31 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { dead = 3; break; }
            }
            if (j == 0) QN_LU_STAMP(11);
            if (lane < NW) L.rec[j][lane] = v;
            if (lane == 0) {
                L.dead[j] = dead;
                if (dead == 2) __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        if (j == 0) QN_LU_STAMP(12);
        if (L.dead[j]) return 2; // (uniform)
        // the parts' candidates, lane g2 = lane mod G taking part g2's (one trip through LDS); the winner's part follows from its row
        double gb = lus_d(L.rec[j][(lane & (G - 1)) * QN_LUS_W]);
        int gi = (int)L.rec[j][(lane & (G - 1)) * QN_LUS_W + 1];
        lu_argmax_lanes<G>(gb, gi);
        gb = lu_uniform_d(gb); gi = __builtin_amdgcn_readfirstlane(gi);
        const int gw = (gi / QN_LU_PT) % G; // (no candidate anywhere: 0x7fffffff -- the singular exit below)
        if (!(gb > 0.0) || gi >= m) { status = 1; gi = k; } // no non-zero entry in this column: singular (newton/mod.rs:43-46)
        const int p = gi;
        pv[j] = p0 + p;
        if (status) break; // (uniform)
        double rowp[QN_LU_SUB], rowk[QN_LU_SUB];
#pragma unroll
        for (int jj = 0; jj < QN_LU_SUB; ++jj) { rowp[jj] = lus_d(L.rec[j][gw * QN_LUS_W + 2 + jj]); rowk[jj] = lus_d(L.rec[j][6 + jj]); }
        {
            const int chunk = p / QN_LU_PT, tp = p & (QN_LU_PT - 1); // (uniform)
            const bool own_p = chunk % G == g && tid == tp && p != k, own_k = g == 0 && tid == k && p != k;
            const int jp = chunk / G;
#pragma unroll
            for (int jr = 0; jr < RPTL; ++jr) {
                const bool hit = own_p && jr == jp;
#pragma unroll
                for (int jj = 0; jj < QN_LU_SUB; ++jj) a[jj][jr] = hit ? rowk[jj] : a[jj][jr];
            }
            // (after the owner of p: when both rows are this thread's, row k ends up with the pivot row)
#pragma unroll
            for (int jj = 0; jj < QN_LU_SUB; ++jj) a[jj][0] = own_k ? rowp[jj] : a[jj][0];
        }
        if (j == 0) QN_LU_STAMP(5);
        const double inv_ukk = 1.0 / rowp[j];
#pragma unroll
        for (int jr = 0; jr < RPTL; ++jr) {
            const int i = tid + QN_LU_PT * (jr * G + g);
            if (i > k && i < m) {
                const double l = a[j][jr] * inv_ukk;
                a[j][jr] = l;
#pragma unroll
                for (int jj = 0; jj < QN_LU_SUB; ++jj)
                    if (jj > j) a[jj][jr] = a[jj][jr] - l * rowp[jj];
            }
        }
        if (j == QN_LU_SUB - 1) {
#pragma unroll
            for (int jj = 0; jj < QN_LU_SUB; ++jj) rowp3[jj] = rowp[jj];
        }
        if (j == 0) QN_LU_STAMP(6);
        if (j == 3) QN_LU_STAMP(7);
    }
    if (tid == 0) {
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < QN_LU_SUB; ++j) lu_sti<true>(piv + p0 + QN_LU_SUB * s + j, pv[j]);
            if (status) lu_sti<true>(fail, 1);
        }
        if (!status) { // the finished 4 x 4 block's multipliers, for the next sub-panel's 4 x 4 solve (read behind the next barrier)
            L.l11[1][0] = lus_d(L.rec[QN_LU_SUB - 1][10]);
            L.l11[2][0] = lus_d(L.rec[QN_LU_SUB - 1][11]);
            L.l11[2][1] = lus_d(L.rec[QN_LU_SUB - 1][12]);
#pragma unroll
            for (int c = 0; c < QN_LU_SUB - 1; ++c) L.l11[3][c] = rowp3[c];
        }
    }
    // (every row of the thread, no condition: see lu_col_store)
#pragma unroll
    for (int jr = 0; jr < RPTL; ++jr) {
#pragma unroll
        for (int j = 0; j < QN_LU_SUB; ++j) lu_st<true>(P + (size_t)(c0 + j) * pld + tid + QN_LU_PT * (jr * G + g), a[j][jr]);
    }
    return status;
}

// sync[0]: part 0's sub-panel counter (as lu_panel_persist_kernel's); sync[1 + c]: column c's; sync[65 + g]: part g's, g >= 1
__device__ __forceinline__ int* lus_part_flag(int* sync, const int g) { return g == 0 ? sync : sync + 1 + QN_NB + g; }

template <int RPTL, int G>
__global__ __launch_bounds__(QN_LU_PT) void lu_panel_split_kernel(double* __restrict__ P, size_t pld, int m, int p0, int* __restrict__ piv, int* __restrict__ fail,
                                                                  int* __restrict__ sync, int base, int spin_max, unsigned long long* __restrict__ rec,
                                                                  unsigned long long* __restrict__ rec_next) {
    static_assert(G >= 2 && G <= QN_LUS_MAXG && G * QN_LUS_W <= 64, "one wave reads a step's records");
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    constexpr int NSUB = QN_NB / QN_LU_SUB;
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    if ((b & 7) == 0 && (b >> 3) < G) { // role A, part g: workgroups 0, 8, .. -- one XCD
        __shared__ QnLusLds L;
        const int g = b >> 3;
        double lprev[QN_LU_SUB][RPTL];
        int pvprev[QN_LU_SUB] = {0, 0, 0, 0};
        int* const myflag = lus_part_flag(sync, g);
        for (int s = 0; s < NSUB; ++s) {
            QN_LU_STAMP(0);
            if (s >= 2) { // its four columns carry sub-panel s - 2 (their workgroups' last step)
                const bool ok = tid < QN_LU_SUB ? lu_wait_ge(sync + 1 + QN_LU_SUB * s + tid, base + s - 1, fail, spin_max) : true;
                if (!__syncthreads_and(ok)) return;
            } else __syncthreads(); // (L.l11 of the previous sub-panel is complete)
            QN_LU_STAMP(1);
            double a[QN_LU_SUB][RPTL];
            lus_cols_update_pre<RPTL, G>(P, pld, m, QN_LU_SUB * (s - 1), s >= 1, QN_LU_SUB * s, p0, pvprev, L, lprev, a, tid, g);
            if (s >= 1) { // sub-panel s - 1 is announced only now: its stores completed while the new columns came in
                lu_release_all();
                if (tid == 0) __hip_atomic_store(myflag, base + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            QN_LU_STAMP(2);
            const int st = lus_sub_factor<RPTL, G>(P, pld, m, s, p0, piv, fail, rec, L, a, tid, g, pvprev, spin_max);
            if (st) return; // (*fail is set: the others leave at their next poll)
            QN_LU_STAMP(8);
#pragma unroll
            for (int q = 0; q < QN_LU_SUB; ++q)
#pragma unroll
                for (int jr = 0; jr < RPTL; ++jr) lprev[q][jr] = a[q][jr];
            if (s == NSUB - 1) {
                lu_release_all();
                if (tid == 0) __hip_atomic_store(myflag, base + s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    __shared__ QnLuLds LB;
    if (b == 1) { // role C: first the next panel's records back to the sentinel (nobody reads that area in this launch), then as in lu_panel_persist_kernel
        for (int e = tid; e < QN_LUS_REC_WORDS; e += QN_LU_PT) rec_next[e] = QN_LUS_SENT;
        for (int sg = 2; sg <= NSUB; ++sg) {
            bool ok = true;
            if (tid == 0) ok = lu_wait_ge(sync, base + sg, fail, spin_max);
            else if (tid < 1 + QN_NB) { if (tid - 1 >= QN_LU_SUB * sg && tid - 1 >= 2 * QN_LU_SUB) ok = lu_wait_ge(sync + tid, base + sg - 1, fail, spin_max); }
            else if (tid > 1 + QN_NB && tid < 1 + QN_NB + G) ok = lu_wait_ge(sync + tid, base + sg, fail, spin_max); // (parts 1 .. G - 1)
            if (!__syncthreads_and(ok)) return;
            lu_role_c<true>(P, pld, QN_LU_SUB * (sg - 1), p0, piv, tid);
        }
        return;
    }
    // role B: column c, sub-panels 0 .. c / 4 - 2 (the workgroups that are neither a part of role A nor role C, in index order)
    const int c = 2 * QN_LU_SUB + (b - 2) - min(G - 1, (b - 1) >> 3);
    const int last = c / QN_LU_SUB - 1;
    for (int sg = 1; sg <= last; ++sg) {
        const bool ok = tid < G ? lu_wait_ge(lus_part_flag(sync, tid), base + sg, fail, spin_max) : true;
        if (!__syncthreads_and(ok)) return;
        // (rows per thread of a whole column: RPTL x G)
        constexpr int RPT = RPTL * G;
        double a[QN_LU_SUB][RPT];
        const int r0 = QN_LU_SUB * (sg - 1);
        lu_cols_update<RPT, true, (RPT >= 16 ? 4 : 16)>(P, pld, m, r0, true, c, 1, p0, piv, LB, a, tid);
        lu_col_store<RPT, true>(P, pld, m, r0, c, a, tid);
        lu_release_all();
        if (tid == 0) __hip_atomic_store(sync + 1 + c, base + sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
