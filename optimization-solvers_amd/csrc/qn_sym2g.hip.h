// qn_sym2g.hip.h -- a device-resident objective that is NOT the quadratic, in the structure of the second-generation path
// (round 5, VERDICT r4 item 3: BASELINE.json config 5 -- DFP + More-Thuente on the log-sum-exp objective -- ran on the generic path:
// synchronous, a host round trip per request, a one-workgroup control kernel doing the O(n) vector work, ~9 launches per iteration).
//
// What carries over from qn_sym2.hip.h: the solver's state machine runs on the device in the PROLOGUE of launches (nothing goes
// back to the host during a batch: the pattern is enqueued hundreds of iterations ahead, every kernel predicated on the control
// block); an evaluation hands the line search two scalars, f and g(x + t d)'d; the trial point is formed on the fly from the lazy
// direction d = -(v + al s + be u); accepting a point is a toggle of the x / s double buffers; the update pass keeps the
// symmetric half of H.
// What differs, and why: at the sizes this objective is for (n = m = 16384: A is 2.1 GB, H's half 1.1 GB) a launch is 0.5 % of an
// iteration and the streaming kernels want MANY workgroups (the first-generation tile kernel, two workgroups per CU, streams H's
// half at 6.2 TB/s where the one-workgroup-per-CU kernel of qn_sym2.hip.h reaches 5.4 past the Infinity Cache).  So the machine
// does not ride in the streaming kernels' prologues: it runs in a ONE-WORKGROUP launch in front of them (s2_advance_kernel<.., GOBJ,
// KIND>), which marks the request `serviced = 1`, and the kernels that do the work only READ the control block:
//
//     advance(eval) | s2g_onepass (one pass over A: running-maximum softmax, per-workgroup (m, S, G)) | s2g_combine |
//     advance(tiles) | sym_hpass_tile_kernel (qn_sym.hip.h) | s2_hreduce_kernel
//
// s2g_combine folds the workgroups' (m, S, G) -- and, because it then holds g(x + t d) entry by entry, does at once what the
// line search AND an acceptance need: g+'d for the Wolfe test, and -- staged, in case the machine accepts this point -- g+, y = g+ - g,
// x+ = x + t d, s = x+ - x and the five sums of bfgs.rs:94-102 / dfp.rs:94-102.  The next prologue finds them in the table (columns
// the second table) and goes from "accepted" to "update pass requested" without a launch (qn_s2_prologue_w0<.., GOBJ>).  A rejected
// trial point costs its staging writes (5 n doubles), never a launch.
// Reference lines served: dfp.rs:78-123, bfgs.rs:78-127, morethuente.rs:165-297, backtracking.rs:20-58, ls_solver.rs:66-111.
#pragma once

struct QnS2GArgs {
    QnLseArgs L;             // A, c, mu, the sizes (x, f_out, g_out are not used: the point comes from the control block, the results go to the table)
    double* wgms;            // [G][2]  per workgroup of the pass: running maximum m, S = sum exp(z - m)
    double* wgg;             // [G][n_pad] ... and G = sum exp(z - m) a_i
    int G;                   // workgroups of the pass over A
    const QnCtl* ctl;        // the control block the advance launch in front has written (read only)
    QnFused F;               // X0[2], S0[2], G, GT, Y, UN, VV
    double* wgS;             // the table half this evaluation's combine launch writes: [QN_S2_ROW][trows]
    double* wgV;             // ... and the same half of the second table (the staged accepted-point sums)
    int trows;
    // row-sharded runs (rows of A and of H sharded, vectors replicated):
    double* gall;            // [world][n_pad]: the ranks' G_r = sum_w exp(m_w - m_r) G_w, rank r's slice written by its combine launch;
                             // all-gathered only for the point the line search accepts
    double* ev_slice;        // this rank's slice of evS for this launch: [QN_S2SH_NEC][QN_S2_MAXG] -- column 0: G_r'd per workgroup, column 1: m_r, S_r
};

// the request as the two kernels decode it (qn_s2_eval_req without the scalar-register pinning: these kernels have registers to spare)
__device__ __forceinline__ bool qn_s2g_mine(const QnCtl* __restrict__ ctl) { return ctl->phase == QN_PH_REQ_EVAL && ctl->serviced == 1; }

// ONE pass over this rank's rows of A at the trial point (lse_onepass_kernel, qn_kernels.hip.h, with the point formed here).
// The first row is requested before the control block is read: its address does not depend on it.
template <int KCH, bool NTA>
__global__ __launch_bounds__(512) void s2g_onepass_kernel(const QnS2GArgs g) {
    extern __shared__ __attribute__((aligned(16))) double lse_x[]; // KCH * 1024 entries of x + t d, zero past n_pad
    __shared__ double red[2][8];
    const QnLseArgs& a = g.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int np = a.n_pad;
#define QN_LSE_JC(k) min(2 * tid + 1024 * (k), np - 2)
    const int G = gridDim.x, per = (a.mrpr + G - 1) / G;
    const int r_lo = blockIdx.x * per;
    const int r_hi = min(min(a.mrpr, r_lo + per), a.m - a.rank * a.mrpr); // (rows past m are padding)
    v2d gacc[KCH], cur[KCH], nxt[KCH];
#pragma unroll
    for (int k = 0; k < KCH; ++k) { gacc[k] = (v2d){0.0, 0.0}; cur[k] = (v2d){0.0, 0.0}; }
    if (r_lo < r_hi) {
#pragma unroll
        for (int k = 0; k < KCH; ++k) cur[k] = NTA ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(a.A + (size_t)r_lo * np + QN_LSE_JC(k))) : ld2(a.A + (size_t)r_lo * np + QN_LSE_JC(k));
    }
    if (!qn_s2g_mine(g.ctl)) return; // (uniform: every thread reads the same two words)
    const QnEvalReq q = qn_s2_eval_req<false>(*g.ctl, false);
    {
        const double* __restrict__ x = g.F.X0 + (size_t)q.xc * (size_t)np;
        const double* __restrict__ sp = g.F.S0 + (size_t)q.sc * (size_t)np;
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int j = 2 * tid + 1024 * k;
            v2d xt = {0.0, 0.0};
            if (j < np) {
                const v2d xv = ld2(x + j), vv = ld2(g.F.VV + j), sv = ld2(sp + j), uv = ld2(g.F.UN + j);
                double d0, d1;
                xt.x = qn_s2_trial(q, xv.x, vv.x, sv.x, uv.x, d0);
                xt.y = qn_s2_trial(q, xv.y, vv.y, sv.y, uv.y, d1);
            }
            lse_x[j] = xt.x; lse_x[j + 1] = xt.y;
        }
    }
    __syncthreads();
    double m_run = -INFINITY, s_run = 0.0;
    for (int r = r_lo; r < r_hi; ++r) {
        const double* nrow = a.A + (size_t)((r + 1 < r_hi) ? r + 1 : r) * np; // (last row: a harmless re-read, no branch)
#pragma unroll
        for (int k = 0; k < KCH; ++k) nxt[k] = NTA ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(nrow + QN_LSE_JC(k))) : ld2(nrow + QN_LSE_JC(k));
        double p = 0.0;
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int j = 2 * tid + 1024 * k;
            const v2d xv = *reinterpret_cast<const v2d*>(&lse_x[j]);
            p = __builtin_fma(cur[k].x, xv.x, p);
            p = __builtin_fma(cur[k].y, xv.y, p);
        }
        p = qn_wave_sum(p);
        if (lane == 0) red[r & 1][wave] = p;
        __syncthreads();
        double z = red[r & 1][0];
#pragma unroll
        for (int w = 1; w < 8; ++w) z = z + red[r & 1][w];
        z = z + a.c[a.rank * a.mrpr + r];
        const double m_new = fmax(m_run, z);
        const double scale = exp(m_run - m_new), e = exp(z - m_new); // (first row: exp(-inf) = 0)
        s_run = __builtin_fma(s_run, scale, e);
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            gacc[k].x = __builtin_fma(e, cur[k].x, gacc[k].x * scale);
            gacc[k].y = __builtin_fma(e, cur[k].y, gacc[k].y * scale);
            cur[k] = nxt[k];
        }
        m_run = m_new;
    }
    if (tid == 0) { g.wgms[2 * blockIdx.x] = m_run; g.wgms[2 * blockIdx.x + 1] = s_run; }
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
        const int j = 2 * tid + 1024 * k;
        if (j < np) st2(g.wgg + (size_t)blockIdx.x * np + j, gacc[k]);
    }
#undef QN_LSE_JC
}

// Workgroup b: columns 64 b .. 64 b + 63.  The workgroups' (m, S, G) folded in a fixed order (each quarter of the workgroups in
// workgroup order, sixteen loads in flight, the four quarters added in order: lse_combine_kernel's order), then for its columns
//     g+ = G / S + mu xt,  y = g+ - g,  x+ = xt,  s = xt - x          (bfgs.rs:94-99; s is x+ - x, not t d)
// staged in the buffers the update pass reads, and this workgroup's row of the table:
//     columns 0..5 (what qn_s2_advance reads for an evaluation: f = 1/2 tot0 - tot1, g(xt)'d = tot2 - tot3, g'd = tot4, #non-finite d):
//         mu sum xt^2 (+ 2 (M + log S) in workgroup 0's row), 0, sum g+ d, 0, sum g d, #non-finite d
//     columns the second table + 4 (what it reads for an accepted point): y'y, y's, g+'g+, s's, s'g+
// The launch's prologue passes the control block on (serviced 1 -> 2): nothing is decided between the pass over A and this.
// SHARD (row-sharded runs): the rank folds ITS workgroups -- rows of A it owns -- into (m_r, S_r, G_r) and hands the line search what it
// can: G_r'd per workgroup and (m_r, S_r) into its slice of evS (one 8 KB exchange follows; the next prologue weighs the ranks with
// exp(m_r - M) in rank order), mu xt'xt, xt'd, g'd and the non-finite count -- the same on every rank, the vectors are replicated --
// into the table.  G_r stays in the rank's slice of `gall`: it crosses the links only if the point is accepted (s2g_vec_kernel).  A
// trial point that is rejected never moves an n-vector between GPUs.
template <bool SHARD>
__global__ __launch_bounds__(256) void s2g_combine_kernel(const QnS2Args a, const QnS2GArgs g) {
    __shared__ QnS2Lds L;
    __shared__ double fac[256];
    __shared__ double lds[32];
    __shared__ double part[3][64];
    const int tid = threadIdx.x;
    const QnLseArgs& la = g.L;
    const int G = g.G;
    // (the workgroups' maxima and sums, and the first sixteen partial gradients of this thread's quarter, go out before the control
    // block is known: their addresses do not depend on it)
    const double mw = tid < G ? g.wgms[2 * tid] : -INFINITY;
    const double sw = tid < G ? g.wgms[2 * tid + 1] : 0.0;
    if (tid < 64) qn_s2_prologue_w0<QN_S2_GCOMB>(a, L);
    __syncthreads();
    qn_s2_ctl_out(a, L);
    if (!L.mine) return;
    const double mr = ctl_block_fmax(mw, lds);
    // (a workgroup without rows: exp(-inf) = 0; no row at all -- every maximum -inf -- would make exp(-inf - (-inf)) = NaN)
    const double fw = (tid < G && mw != -INFINITY) ? exp(mw - mr) : 0.0;
    fac[tid] = fw;
    double sp1[1] = {sw * fw};
    ctl_block_sum<1>(sp1, lds); // S = sum_w S_w exp(m_w - M): fixed tree, every workgroup the same bits
    const double S = sp1[0];
    __syncthreads();
    const int c = tid & 63, qd = tid >> 6;
    const int j = blockIdx.x * 64 + c; // (n = n_pad on this path, a multiple of 64)
    const int per = (G + 3) / 4, w_lo = qd * per, w_hi = min(G, w_lo + per);
    double acc = 0.0;
    for (int w0 = w_lo; w0 < w_hi; w0 += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (w0 + u < w_hi) ? g.wgg[(size_t)(w0 + u) * la.n_pad + j] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_fma(v[u], (w0 + u < w_hi) ? fac[w0 + u] : 0.0, acc);
    }
    if (qd > 0) part[qd - 1][c] = acc;
    __syncthreads();
    if (qd != 0) return; // (wave 0 holds the 64 columns)
    const double gs = ((acc + part[0][c]) + part[1][c]) + part[2][c];
    const QnEvalReq q = qn_s2_eval_req<false>(L.c, false);
    const size_t np = (size_t)la.n_pad;
    const double* __restrict__ x = g.F.X0 + (size_t)q.xc * np;
    double* __restrict__ xstage = g.F.X0 + (size_t)(1 - q.xc) * np;
    const double* __restrict__ spv = g.F.S0 + (size_t)q.sc * np;
    double* __restrict__ sstage = g.F.S0 + (size_t)(1 - q.sc) * np;
    double di;
    const double xi = x[j];
    const double xti = qn_s2_trial(q, xi, g.F.VV[j], spv[j], g.F.UN[j], di);
    if (SHARD) {
        const double go = g.F.G[j];
        g.gall[(size_t)la.rank * np + j] = gs;
        double p[8] = {la.mu * (xti * xti), xti * di, go * di, isfinite(di) ? 0.0 : 1.0, gs * di, 0.0, 0.0, 0.0};
        QnWaveFold<8, 32>::run(p, c);
        double* T = g.wgS;
        const size_t tr = (size_t)g.trows;
        if ((c & 7) == 0) {
            const int k = c >> 3;
            if (k < 4) { const int col = k == 0 ? 0 : (k == 1 ? 1 : (k == 2 ? 4 : 5)); T[(size_t)col * tr + blockIdx.x] = p[0]; }
            else if (k == 4) g.ev_slice[blockIdx.x] = p[0];                       // column 0 of the slice: G_r'd of this workgroup's columns
            else if (k == 5) { T[(size_t)2 * tr + blockIdx.x] = 0.0; T[(size_t)3 * tr + blockIdx.x] = 0.0; }
            if (k == 6 && blockIdx.x == 0) { g.ev_slice[QN_S2_MAXG] = mr; g.ev_slice[QN_S2_MAXG + 1] = S; } // column 1, rows 0 and 1
        }
        return;
    }
    const double gti = gs / S + la.mu * xti;
    const double go = g.F.G[j];
    const double yi = gti - go;
    const double si = xti - xi; // s = x+ - x, not t d (bfgs.rs:96)
    g.F.GT[j] = gti;
    g.F.Y[j] = yi;
    xstage[j] = xti;
    sstage[j] = si;
    double p[8] = {la.mu * (xti * xti), gti * di, go * di, isfinite(di) ? 0.0 : 1.0, 0.0, 0.0, 0.0, 0.0};
    double pv[8] = {yi * yi, yi * si, gti * gti, si * si, si * gti, 0.0, 0.0, 0.0};
    QnWaveFold<8, 32>::run(p, c);  // lane 8 k holds the total of value k
    QnWaveFold<8, 32>::run(pv, c);
    double* T = g.wgS;
    const size_t tr = (size_t)g.trows;
    if ((c & 7) == 0) {
        const int k = c >> 3;
        double v = p[0];
        if (k == 0 && blockIdx.x == 0) v = v + 2.0 * (mr + log(S)); // f = M + log S + mu/2 ||xt||^2 = 1/2 tot0
        if (k < 4) { const int col = k == 0 ? 0 : (k == 1 ? 2 : (k == 2 ? 4 : 5)); T[(size_t)col * tr + blockIdx.x] = v; }
        else if (k == 4) T[(size_t)1 * tr + blockIdx.x] = 0.0;
        else if (k == 5) T[(size_t)3 * tr + blockIdx.x] = 0.0;
        if (k < QN_S2_NR) g.wgV[(size_t)k * tr + blockIdx.x] = pv[0]; // (the second table, the same half)
    }
}

// Row-sharded runs, the point the line search accepted: block-row R of g+ = (sum_r w_r G_r) / S + mu x+ (the ranks' G_r gathered by the
// exchange in front, the weights and S as the prologue that consumed the evaluation left them), y = g+ - g, x+, s = x+ - x and the five
// sums of bfgs.rs:94-102 -- on every rank, from the same inputs in the same order: the same bits.  What s2_vec_kernel<true> is for the
// quadratic; the launch's prologue passes the control block on (serviced 1 -> 2).
__global__ __launch_bounds__(QN_TB) void s2g_vec_kernel(const QnS2Args a, const QnS2GArgs g) {
    __shared__ QnS2Lds L;
    __shared__ double bred[2][8];
    const int R = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) qn_s2_prologue_w0<QN_S2_VEC, true>(a, L);
    __syncthreads();
    qn_s2_ctl_out(a, L);
    if (!L.mine) return;
    const QnLseArgs& la = g.L;
    const QnEvalReq q = qn_s2_eval_req<false>(L.c, true);
    const size_t np = (size_t)la.n_pad;
    const int j = R * QN_TB + tid;
    const double* __restrict__ x = g.F.X0 + (size_t)q.xc * np;
    double* __restrict__ xstage = g.F.X0 + (size_t)(1 - q.xc) * np;
    const double* __restrict__ spv = g.F.S0 + (size_t)q.sc * np;
    double* __restrict__ sstage = g.F.S0 + (size_t)(1 - q.sc) * np;
    double gs = 0.0;
    for (int r = 0; r < la.world; ++r) gs = __builtin_fma(g.gall[(size_t)r * np + j], a.gws[r], gs); // ranks in rank order (lse_finish1_kernel)
    const double S = a.gws[la.world];
    double di;
    const double xi = x[j];
    const double xti = qn_s2_trial(q, xi, g.F.VV[j], spv[j], g.F.UN[j], di);
    const double gti = gs / S + la.mu * xti;
    const double yi = gti - g.F.G[j];
    const double si = xti - xi; // s = x+ - x, not t d (bfgs.rs:96)
    g.F.GT[j] = gti;
    g.F.Y[j] = yi;
    xstage[j] = xti;
    sstage[j] = si;
    double p[8] = {yi * yi, yi * si, gti * gti, si * si, si * gti, 0.0, 0.0, 0.0};
    QnWaveFold<8, 32>::run(p, lane);
    if ((lane & 7) == 0) bred[wave][lane >> 3] = p[0];
    __syncthreads();
    if (tid < QN_S2_NR) a.wgS[((size_t)a.parity * QN_S2_ROW + tid) * a.trows + R] = bred[0][tid] + bred[1][tid]; // (as s2_vec_kernel)
}
