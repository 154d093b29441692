// qn_fused.hip.h -- the fused fast path (device objective, memoised oracle, BFGS / DFP).
//
// Measured on MI355X (profiles/r01_a_*): a single workgroup moves ~50 GB/s, so sweeping n-vectors in the
// control workgroup cost 10-18 us per step at n = 4096 and would cost ~100 us at n = 32768.  In the fused
// path NO kernel other than the two streaming kernels ever touches an n-vector:
//
//   quad_eval_fused_kernel  q = Q_rows (x + t d) with the direction formed on the fly
//                           d_j = -(v_j + c_su (s_j (u.g) + u_j (s.g)) + c_ss s_j (s.g) + c_uu u_j (u.g)),
//       epilogue per row i: gt_i = q_i - b_i, y_i = gt_i - g_i, and nine per-workgroup partial sums
//                           (x+.q, b.x+, gt.d, g.d, y.y, y.s, gt.gt, s.s, #non-finite d);
//       row-block 0 also stores the trial point x+ (into the other half of the x double buffer), the staged
//       step s = x+ - x, and refreshes the pending-u copy.
//   h_pass_fused_kernel     the H pass of qn_kernels.hip.h with an epilogue: u_i, v_i stored as vectors,
//                           three partial sums (y.u, u.g+, s.g+); row-block 0 commits g <- g+.
//   ctl_step (fused states) sums P*nblk partials per quantity and runs the scalar state machine.
//
// Accepting a line-search point is a pointer toggle (x double buffer, s double buffer) -- no copies.
// Buffers whose slices are all-gathered across ranks (gt, y, u, v, partials) are fixed, so the host can issue
// the collectives without knowing any device-side decision.
#pragma once

// d_j on the fly; `mode` 0: d = -v (direction pass), 1: lazy H+ g+ formula
__device__ __forceinline__ double qn_dir1(int mode, double v, double s, double u, double c_ss, double c_su, double c_uu, double ug,
                                          double sg) {
    double w = v;
    if (mode) {
        if (c_su != 0.0) w = w + c_su * (s * ug + u * sg);
        w = w + c_ss * (s * sg);
        if (c_uu != 0.0) w = w + c_uu * (u * ug);
    }
    return -w;
}

// Sum of column k of a per-workgroup partial buffer [world][NP][nblk] by ONE wave: lanes stride the tiles in global
// order (rank-major = row order) with 16 loads in flight, then a shuffle tree.  Every lane returns the total.  Used by
// the control step and, for the deferred update, by every workgroup of the evaluation kernel: the same code, so the
// same bits everywhere.
__device__ __forceinline__ double qn_partial_col_sum(const double* __restrict__ part, int world, int nblk, int NP, int k, int lane) {
    const int E = world * nblk;
    double acc = 0.0;
    for (int e0 = 0; e0 < E; e0 += 16 * 64) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = e0 + u * 64 + lane;
            const bool ok = e < E;
            const int ee = ok ? e : 0;
            const int r = ee / nblk, b = ee - r * nblk;
            const double x = part[((size_t)r * NP + k) * nblk + b];
            v[u] = ok ? x : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = acc + v[u];
    }
    return qn_wave_sum(acc);
}

// coefficients of the symmetric rank-2 form of bfgs.rs:115-124 (method 0) / dfp.rs:115-120 (method 1)
// (method 4, SR1 -- second-generation path only: (s - u)(s - u)'/((s - u).y) = c (s s' - (s u' + u s') + u u'), c = 1 / den, sr1_b.rs:143-146)
__device__ __forceinline__ void qn_update_coeffs(int method, double ys, double yu, double& c_ss, double& c_su, double& c_uu, double den = 0.0) {
    if (method == 0) { const double rho = 1.0 / ys; c_su = -rho; c_ss = rho * rho * yu + rho; c_uu = 0.0; }
    else if (method == 4) { c_ss = 1.0 / den; c_su = -c_ss; c_uu = c_ss; }
    else { c_ss = 1.0 / ys; c_su = 0.0; c_uu = -1.0 / yu; }
}

// reduce NP values over the first R lanes of wave 0 (R a power of two <= 16); lane 0 gets the totals
template <int R, int NP>
__device__ __forceinline__ void qn_rows_reduce(double (&p)[NP]) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int off = R / 2; off >= 1; off >>= 1) p[k] = p[k] + __shfl_xor(p[k], off, 64);
    }
}

struct QnEvalFusedArgs {
    const double* Q;
    QnTile T;
    QnFused F;
    const QnCtl* ctl;
    int expect_phase;
    int after_h; // this launch directly follows an h_pass launch: it may service QN_PH_REQ_HPASS_EVAL
    int world;
};

// experiment switches (cache policy of the streamed matrices)
#ifdef QN_NT_H
#define QN_ST2_STREAM(p, v) __builtin_nontemporal_store((v), reinterpret_cast<v2d*>(p))
#define QN_LD2_H(p) __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p))
#else
#define QN_ST2_STREAM(p, v) st2((p), (v))
#define QN_LD2_H(p) ld2(p)
#endif
#ifdef QN_NT_Q
#define QN_LD2_Q(p) __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p))
#else
#define QN_LD2_Q(p) ld2(p)
#endif

template <int R, int U>
__global__ __launch_bounds__(QN_TPB) void quad_eval_fused_kernel(const QnEvalFusedArgs a) {
    __shared__ double red[4 * R];
    const QnTile T = a.T;
    const size_t np = (size_t)T.n_pad;
    const int tid = threadIdx.x;
    const int rb = blockIdx.x * R;
    const int nchunks = (T.n_pad + QN_CHUNK - 1) / QN_CHUNK;
    // The matrix addresses do not depend on the control block: issue the first tile's loads before reading it, so
    // the (remote) control-block read and the first HBM round trip overlap.  A predicated-off launch wastes them.
    v2d h[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = u * QN_CHUNK + 2 * tid;
        const double* qbase = a.Q + (size_t)rb * np + (j < T.n_pad ? j : 0);
#pragma unroll
        for (int r = 0; r < R; ++r) h[u][r] = QN_LD2_Q(qbase + (size_t)r * np);
    }
    const QnCtl* __restrict__ ctl = a.ctl;
    const int phase = ctl->phase;
    const bool post_h = a.after_h && phase == QN_PH_REQ_HPASS_EVAL;
    if (phase != a.expect_phase && !post_h) return;
    const int kind = ctl->req_kind;
    const double t = ctl->req_t;
    int mode = ctl->dir_mode;
    double c_ss = ctl->c_ss, c_su = ctl->c_su, c_uu = ctl->c_uu, ug = ctl->dir_ug, sg = ctl->dir_sg;
    const int xc = ctl->xc;
    int sc = ctl->sc;
    if (post_h) { // the update that h_pass just completed is not committed yet: derive its coefficients here (the control
                  // step after this launch commits the very same values: qn_partial_col_sum is order-identical)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (wave < QN_NHPP) { const double v = qn_partial_col_sum(a.F.hpp, a.world, a.F.nblk, QN_NHPP, wave, lane); if (lane == 0) red[wave] = v; }
        __syncthreads();
        const double yu = red[0];
        ug = red[1]; sg = red[2];
        __syncthreads();
        qn_update_coeffs(ctl->method, ctl->ys, yu, c_ss, c_su, c_uu);
        mode = 1;
        sc ^= 1; // the staged s became the pending s
    }
    const double* __restrict__ x = a.F.X0 + (size_t)xc * np;
    double* __restrict__ xt = a.F.X0 + (size_t)(1 - xc) * np;
    const double* __restrict__ sp = a.F.S0 + (size_t)sc * np;
    double* __restrict__ sstage = a.F.S0 + (size_t)(1 - sc) * np;
    const double* __restrict__ un = a.F.UN;
    const double* __restrict__ vv = a.F.VV;
    const bool lead = blockIdx.x == 0;
    const bool is_t = kind == QN_REQ_T;

    // row-side inputs of the epilogue, fetched now so that their latency is hidden behind the streaming loop
    const int gi = T.row_off + rb + (tid < R ? tid : 0);
    const double e_x = x[gi], e_b = a.F.b[gi], e_g = a.F.G[gi];
    const double e_v = is_t ? vv[gi] : 0.0;
    const double e_s = (is_t && mode) ? sp[gi] : 0.0, e_u = (is_t && mode) ? un[gi] : 0.0;

    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    for (int c = 0; c < nchunks; c += U) {
        if (c > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = (c + u) * QN_CHUNK + 2 * tid;
                const double* qbase = a.Q + (size_t)rb * np + (j < T.n_pad ? j : 0);
#pragma unroll
                for (int r = 0; r < R; ++r) h[u][r] = QN_LD2_Q(qbase + (size_t)r * np);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = (c + u) * QN_CHUNK + 2 * tid;
            if (j < T.n_pad) {
                const v2d xj = ld2(x + j);
                v2d xtj = xj;
                v2d uj = {0.0, 0.0};
                if (is_t) {
                    const v2d vj = ld2(vv + j);
                    v2d sj = {0.0, 0.0};
                    if (mode) { sj = ld2(sp + j); uj = ld2(un + j); }
                    const double d0 = qn_dir1(mode, vj.x, sj.x, uj.x, c_ss, c_su, c_uu, ug, sg);
                    const double d1 = qn_dir1(mode, vj.y, sj.y, uj.y, c_ss, c_su, c_uu, ug, sg);
                    const double td0 = t * d0, td1 = t * d1; // `step * direction` rounds first (bfgs.rs:94)
                    xtj.x = xj.x + td0;
                    xtj.y = xj.y + td1;
                }
                if (lead) {
                    st2(xt + j, xtj);
                    v2d sj2;
                    sj2.x = xtj.x - xj.x; // s = x+ - x (bfgs.rs:96)
                    sj2.y = xtj.y - xj.y;
                    st2(sstage + j, sj2);
                    st2(a.F.UP + j, (is_t && mode) ? uj : ld2(un + j)); // refresh the pending-u copy read by the next H pass
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r] = __builtin_fma(h[u][r].x, xtj.x, acc[r]);
                    acc[r] = __builtin_fma(h[u][r].y, xtj.y, acc[r]);
                }
            }
        }
    }
    const double qi = qn_block_fold<R>(acc, red);
    if (tid < R) {
        const double xi = e_x;
        const double di = is_t ? qn_dir1(mode, e_v, e_s, e_u, c_ss, c_su, c_uu, ug, sg) : 0.0;
        double xti = xi;
        if (is_t) { const double td = t * di; xti = xi + td; }
        const double bi = e_b, go = e_g;
        const double gti = qi - bi;
        const double yi = gti - go;
        const double si = xti - xi;
        a.F.GT[gi] = gti;
        a.F.Y[gi] = yi;
        double p[QN_NEVP];
        p[0] = xti * qi;
        p[1] = bi * xti;
        p[2] = gti * di;
        p[3] = go * di;
        p[4] = yi * yi;
        p[5] = yi * si;
        p[6] = gti * gti;
        p[7] = si * si;
        p[8] = isfinite(di) ? 0.0 : 1.0;
        qn_rows_reduce<R, QN_NEVP>(p);
        if (tid == 0) {
            double* out = a.F.evp + (size_t)T.rank * QN_NEVP * a.F.nblk + blockIdx.x;
#pragma unroll
            for (int k = 0; k < QN_NEVP; ++k) out[(size_t)k * a.F.nblk] = p[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// H pass with the fused epilogue.  Column work is h_pass_body's; this kernel adds the vector outputs.
// ------------------------------------------------------------------------------------------------
struct QnHPassFusedArgs {
    double* H;
    QnTile T;
    QnFused F;
    const QnCtl* ctl;
    int expect_phase;
};

template <int R, int U, int NRHS, bool PENDING>
__device__ __forceinline__ void h_pass_fused_body(double* __restrict__ H, const QnTile T, const QnFused& F, const double* __restrict__ sp,
                                                  const double* __restrict__ sstage, const double c_ss, const double c_su,
                                                  const double c_uu, v2d (&h)[U][R], double* red) {
    const int tid = threadIdx.x;
    const int rb = blockIdx.x * R;
    const size_t np = (size_t)T.n_pad;
    const int nchunks = (T.n_pad + QN_CHUNK - 1) / QN_CHUNK;
    const bool use_su = c_su != 0.0, use_uu = c_uu != 0.0;
    const bool lead = blockIdx.x == 0;
    const double* __restrict__ up = F.UP;
    const double* __restrict__ gt = F.GT;
    const double* __restrict__ yv = F.Y;

    double si[R], ui[R];
    bool rowok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int gi = T.row_off + rb + r;
        rowok[r] = gi < T.n;
        si[r] = PENDING ? sp[gi] : 0.0;
        ui[r] = PENDING ? up[gi] : 0.0;
    }
    // row-side inputs of the epilogue (fetched early)
    const int egi = T.row_off + rb + (tid < R ? tid : 0);
    const double e_gp = gt[egi];
    const double e_y = (NRHS == 2) ? yv[egi] : 0.0, e_s = (NRHS == 2) ? sstage[egi] : 0.0;

    constexpr int NV = NRHS * R;
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;

    for (int c = 0; c < nchunks; c += U) {
        if (c > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = (c + u) * QN_CHUNK + 2 * tid;
                const double* hbase = H + (size_t)rb * np + (j < T.n_pad ? j : 0);
#pragma unroll
                for (int r = 0; r < R; ++r) h[u][r] = QN_LD2_H(hbase + (size_t)r * np);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = (c + u) * QN_CHUNK + 2 * tid;
            if (j < T.n_pad) {
                double* hbase = H + (size_t)rb * np + j;
                v2d sj = {0.0, 0.0}, uj = {0.0, 0.0};
                if (PENDING) { sj = ld2(sp + j); uj = ld2(up + j); }
                const v2d gj = ld2(gt + j);                    // g at the accepted point (g+), or g_0 for a direction pass
                const v2d r0 = (NRHS == 2) ? ld2(yv + j) : gj; // update pass: rhs0 = y, rhs1 = g+ ; direction pass: rhs0 = g
                if (lead) st2(F.G + j, gj);                    // commit g <- g+
                const bool c0ok = j < T.n, c1ok = (j + 1) < T.n;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    v2d hn = h[u][r];
                    if (PENDING) {
                        if (use_su) {
                            hn.x = hn.x + c_su * (si[r] * uj.x + ui[r] * sj.x);
                            hn.y = hn.y + c_su * (si[r] * uj.y + ui[r] * sj.y);
                        }
                        hn.x = hn.x + c_ss * (si[r] * sj.x);
                        hn.y = hn.y + c_ss * (si[r] * sj.y);
                        if (use_uu) {
                            hn.x = hn.x + c_uu * (ui[r] * uj.x);
                            hn.y = hn.y + c_uu * (ui[r] * uj.y);
                        }
                        hn.x = (rowok[r] && c0ok) ? hn.x : 0.0;
                        hn.y = (rowok[r] && c1ok) ? hn.y : 0.0;
                        QN_ST2_STREAM(hbase + (size_t)r * np, hn);
                    }
                    acc[r] = __builtin_fma(hn.x, r0.x, acc[r]);
                    acc[r] = __builtin_fma(hn.y, r0.y, acc[r]);
                    if (NRHS == 2) {
                        acc[R + r] = __builtin_fma(hn.x, gj.x, acc[R + r]);
                        acc[R + r] = __builtin_fma(hn.y, gj.y, acc[R + r]);
                    }
                }
            }
        }
    }
    const double tot = qn_block_fold<NV>(acc, red);
    if (tid < NV) {
        const int rhs = tid / R, r = tid % R;
        const int gi = T.row_off + rb + r;
        if (NRHS == 2) {
            if (rhs == 0) F.UN[gi] = tot; else F.VV[gi] = tot;
        } else {
            F.VV[gi] = tot; // direction pass: v = H g
        }
    }
    if (NRHS == 2 && tid < R) { // y.u, u.g+, s.g+ over this tile's rows (tid < R holds u_i)
        double p[QN_NHPP];
        p[0] = e_y * tot;
        p[1] = tot * e_gp;
        p[2] = e_s * e_gp;
        qn_rows_reduce<R, QN_NHPP>(p);
        if (tid == 0) {
            double* out = F.hpp + (size_t)T.rank * QN_NHPP * F.nblk + blockIdx.x;
#pragma unroll
            for (int k = 0; k < QN_NHPP; ++k) out[(size_t)k * F.nblk] = p[k];
        }
    }
}

template <int R, int U>
__global__ __launch_bounds__(QN_TPB) void h_pass_fused_kernel(const QnHPassFusedArgs a) {
    __shared__ double red[4 * 2 * R];
    const size_t np = (size_t)a.T.n_pad;
    v2d h[U][R]; // first tile: issued before the control block is read (see quad_eval_fused_kernel)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = u * QN_CHUNK + 2 * (int)threadIdx.x;
        const double* hbase = a.H + (size_t)(blockIdx.x * R) * np + (j < a.T.n_pad ? j : 0);
#pragma unroll
        for (int r = 0; r < R; ++r) h[u][r] = QN_LD2_H(hbase + (size_t)r * np);
    }
    const QnCtl* __restrict__ ctl = a.ctl;
    const int phase = ctl->phase;
    if (phase != a.expect_phase && phase != QN_PH_REQ_HPASS_EVAL) return;
    const int nrhs = ctl->hp_nrhs;
    const int pending = ctl->pending;
    const double c_ss = ctl->c_ss, c_su = ctl->c_su, c_uu = ctl->c_uu;
    const int sc = ctl->sc;
    const double* sp = a.F.S0 + (size_t)sc * np;
    const double* sstage = a.F.S0 + (size_t)(1 - sc) * np;
    if (pending) {
        if (nrhs == 2) h_pass_fused_body<R, U, 2, true>(a.H, a.T, a.F, sp, sstage, c_ss, c_su, c_uu, h, red);
        else h_pass_fused_body<R, U, 1, true>(a.H, a.T, a.F, sp, sstage, c_ss, c_su, c_uu, h, red);
    } else {
        if (nrhs == 2) h_pass_fused_body<R, U, 2, false>(a.H, a.T, a.F, sp, sstage, c_ss, c_su, c_uu, h, red);
        else h_pass_fused_body<R, U, 1, false>(a.H, a.T, a.F, sp, sstage, c_ss, c_su, c_uu, h, red);
    }
}
