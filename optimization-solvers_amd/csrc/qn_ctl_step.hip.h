// qn_ctl_step.hip.h -- the solver state machine (included by qn_kernels.hip.h).
//
// One workgroup of 1024 threads.  The kernel is latency-bound, so it is organised around two rules:
//   * every vector phase issues ALL its loads before the first use (4 elements per thread per trip, pointers
//     marked __restrict__, invalid lanes clamped to index 0 instead of branching), and phases that can share a
//     sweep are fused (evaluation epilogue + g.d; s, y, norms, y.s and ||g+||^2 in one sweep);
//   * consecutive scalar states run back to back in thread 0 without workgroup barriers; the workgroup only
//     meets at a barrier when the next state needs all threads.
// Thread t always touches elements t, t+1024, ..., so it only ever re-reads its own writes.
#pragma once

#ifdef QN_CTL_STAMPS
#define QN_STAMP(id) do { if (tid == 0 && V.dbg) V.dbg[stamp_base + (id)] = wall_clock64(); } while (0)
#else
#define QN_STAMP(id) do { } while (0)
#endif

#define QN_TILE_IDX(base)                                        \
    int idx[4];                                                  \
    bool ok[4];                                                  \
    _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {           \
        const int i_ = (base) + u_ * tpb + tid;                  \
        ok[u_] = i_ < n_pad;                                     \
        idx[u_] = ok[u_] ? i_ : 0;                               \
    }

// fused path: sum the per-workgroup partials [world][NP][nblk] in global tile order (rank-major = row order).
// Column k is owned by wave (k mod nwaves): its 64 lanes stride the entries with all loads in flight, one
// shuffle tree per column, lane 0 parks the total in LDS.  After the caller's barrier lds[k] holds column k.
template <int NP>
__device__ __forceinline__ void ctl_sum_partials(const double* __restrict__ part, int world, int nblk, double* lds, int wave0 = 0) {
    const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6) - wave0, nw = (int)(blockDim.x >> 6) - wave0;
    if (wave < 0) return;
    for (int k = wave; k < NP; k += nw) {
        const double acc = qn_partial_col_sum(part, world, nblk, NP, k, lane);
        if (lane == 0) lds[k] = acc;
    }
}

__device__ __forceinline__ void qn_keepalive(double v) { asm volatile("" ::"v"(v)); }

// ------------------------------------------------------------------------------------------------
// scalar states (thread 0 only).  Runs until the machine yields or reaches a state that needs all threads.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool qn_check_is_scalar(const QnCtl& c) {
    return c.method != 2 && (c.small_n || c.gg_valid || c.method == 3);
}

// Every scalar state is a function of its own; ctl_scalar_run dispatches on the state and then runs the successions that are
// certain or near-certain BACK TO BACK in straight-line code (qn_st_* below, QN_RUN_NEXT).  Why: the control block lives in LDS and
// the machine is one lane's dependent chain.  As a `switch` inside a loop every state re-read what the previous one had just
// stored -- the compiler cannot forward a store to a load across the loop's back edge -- and a run of six states cost ~100
// serialised LDS round trips, 3 us of every sym2 prologue (tools/s2_stamps.py).  Inlined one after the other the stores are
// forwarded, the `state == X` tests between them fold at compile time, and what remains is the arithmetic.  The functions are
// the former `case` bodies verbatim: one source for every path, nothing is decided differently.
// A state function returns false when the state needs all threads (the caller leaves the scalar run).
// `side_effects`: false in all but one of the workgroups that run the machine redundantly (qn_sym2.hip.h): no trace stores.

// LEAN: the instantiation for the fused symmetric path (qn_sym2.hip.h), where the host has already excluded n <= 5, Newton,
// gradient descent, bounded solvers / line searches, callbacks and qn_compute_step_len (minimize_impl: `r.fused`): the branches for
// those are compiled out.  Same decisions on that path by construction; about half the code -- and the machine is inlined into
// every kernel of an iteration, where code size is instruction-fetch time behind the kernel's data burst.
// LEAN == 2: the lean machine of the second-generation path's BOUNDED runs (BFGSB / DFPB, MoreThuenteB; QnCtl.s2_dir): a new direction is
// not searched along before it has been through QN_PH_REQ_DIR -- s2_dir_kernel stores it, projected, and reduces the step to the line
// search's box, which clips t_max (morethuente_b.rs:201; from there on MoreThuenteB IS More-Thuente: the host hands this machine ls_kind 0).
template <int LEAN>
__device__ __forceinline__ void qn_st_begin(QnCtl& c) { // ls_solver.rs:74-76: only k is reset
    c.k = 0;
    // A fresh call evaluates the oracle at x_k and forms d = -H g from scratch (ls_solver.rs:79, bfgs.rs:47).  When it
    // continues the previous call -- same immutable device objective, memoised oracle, state untouched -- both are already
    // known: the last accepted evaluation IS f, g at x_k, and the direction is pending in its lazy form.
    if (!c.warm) { c.have_cur_eval = 0; c.have_dir = 0; c.last_valid = 0; c.gg_valid = 0; }
    else c.last_valid = 0;
    c.n_oracle_calls = 0; c.n_oracle_evals = 0; c.n_hpasses = 0; c.n_hpass_rw = 0; c.n_iterations = 0;
    c.status = -1;
    c.state = QN_ST_LOOP_TOP;
}

template <int LEAN>
__device__ __forceinline__ void qn_st_loop_top(QnCtl& c) { // ls_solver.rs:78-79
    if (!(c.max_iter > c.k)) {
        c.status = 1; // MaxIterReached, ls_solver.rs:109-110
        c.phase = QN_PH_DONE;
    } else {
        c.tr_n_evals = 0; c.tr_ls_iters = 0; c.tr_ls_cases = 0; c.tr_ndigits = 0; c.tr_updated = 0;
        c.ls_result = NAN;
        c.n_oracle_calls++;
        c.tr_n_evals++;
        if (c.memoize && c.have_cur_eval) {
            c.state = QN_ST_CHECK;
        } else {
            c.req_kind = QN_REQ_X; c.req_t = 0.0; c.req_need_vectors = 1;
            c.after_state = QN_ST_AFTER_EVALX;
            c.phase = QN_PH_REQ_EVAL;
        }
    }
}

template <int LEAN>
__device__ __forceinline__ bool qn_st_after_evalx(QnCtl& c) {
    if (!LEAN && !c.fused) return false;
    c.f_k = c.f_e; // g <- gt is committed by the direction pass (h_pass row-block 0)
    c.gg = c.st_gg; c.gg_valid = 1;
    c.have_cur_eval = c.memoize;
    c.have_dir = 0;
    c.state = QN_ST_CHECK;
    return true;
}

template <int LEAN>
__device__ __forceinline__ bool qn_st_after_dir(QnCtl& c) {
    if (!LEAN && !c.fused) return false;
    c.n_hpasses++;
    if (c.pending) c.n_hpass_rw++;
    c.pending = 0;
    c.dir_mode = 0; // d = -v
    c.gd0_valid = 0; c.d_finite = 0; c.last_valid = 0;
    c.state = QN_ST_LS_BEGIN;
    if (LEAN == 2 && c.s2_dir) { c.dir_ready = 0; c.after_state = QN_ST_LS_BEGIN; c.phase = QN_PH_REQ_DIR; }
    return true;
}

template <int LEAN>
__device__ __forceinline__ bool qn_st_after_next(QnCtl& c) { // bfgs.rs:94-102 from the sums staged by the accepted evaluation
    if (!LEAN && !c.fused) return false;
    c.s_norm = sqrt(c.st_ss); c.has_s_norm = 1;
    c.y_norm = sqrt(c.st_yy); c.has_y_norm = 1;
    c.ys = c.st_ys;
    c.gg = c.st_gg; c.gg_valid = 1;
    c.f_k = c.f_e;
    c.xc ^= 1; // x <- x+ : the trial half of the double buffer becomes x
    c.have_cur_eval = c.memoize;
    c.have_dir = 0;
    c.last_valid = 0;
    if (c.s_norm < c.tol || c.y_norm < c.tol) { // bfgs.rs:106-112: H is not updated
        c.state = QN_ST_ITER_END;
    } else {
        c.hp_lazy = 1; c.hp_nrhs = 2;
        const double fk = c.f_k;
        const bool will_continue = (c.k + 1 < c.max_iter) && !(isnan(fk) || isinf(fk)) && !(sqrt(c.gg) < c.tol);
        if (!LEAN && will_continue && !c.callback_mode && !c.no_defer && !c.sym2) { // (sym2: every step is a prologue, nothing to save)
            // Deferred update: everything the step after the H pass would decide is already known except the
            // update's coefficients (they need y.u, u.g+, s.g+ from the pass).  Run the rest of the iteration
            // bookkeeping now; the next evaluation request becomes QN_PH_REQ_HPASS_EVAL, its kernel derives the
            // coefficients from the pass's partial sums, and the following step commits them.  One launch less.
            c.defer_u = 1;
            c.tr_updated = 1;
            c.have_dir = 1; c.gd0_valid = 0; c.d_finite = 0;
            c.state = QN_ST_ITER_END;
        } else {
            c.after_state = QN_ST_AFTER_U;
            c.phase = QN_PH_REQ_HPASS;
        }
    }
    return true;
}

template <int LEAN>
__device__ __forceinline__ bool qn_st_after_u(QnCtl& c) { // coefficients of bfgs.rs:115-124 / dfp.rs:115-120 in rank-2 form
    if (!LEAN && !c.fused) return false;
    const double yu = c.hp_yu;
    double c_ss, c_su, c_uu;
    qn_update_coeffs(c.method, c.ys, yu, c_ss, c_su, c_uu, c.hp_den);
    c.c_ss = c_ss; c.c_su = c_su; c.c_uu = c_uu;
    c.n_hpasses++;
    if (c.pending) c.n_hpass_rw++;
    c.pending = 1;
    c.sc ^= 1; // the staged s becomes the pending s; the new u is already in UN
    c.dir_mode = 1; c.dir_ug = c.hp_ug; c.dir_sg = c.hp_sg; // next direction formed on the fly by the evaluations
    c.gd0_valid = 0; c.d_finite = 0;
    c.have_dir = 1;
    if (LEAN == 2) c.dir_ready = 0;
    c.tr_updated = 1;
    c.state = QN_ST_ITER_END;
    return true;
}

template <int LEAN>
__device__ __forceinline__ bool qn_st_check(QnCtl& c, const QnVecs& V, double* small_scratch) { // ls_solver.rs:37-40 (OutOfDomain), has_converged (bfgs.rs:64-76)
    const int n = V.n, n_pad = V.n_pad;
    if (!LEAN && !qn_check_is_scalar(c)) return false; // gradient descent / unknown ||g||: all threads needed
    if (!LEAN && c.method == 3) { // Newton: has_converged is the decrement test (newton/mod.rs:64-69)
        c.gnorm = c.gg_valid ? sqrt(c.gg) : NAN; c.tr_f = c.f_k; c.tr_gnorm = c.gnorm;
        const double f = c.f_k;
        if (isnan(f) || isinf(f)) { c.status = 2; c.phase = QN_PH_DONE; }
        else if (c.has_dec && c.dec * 0.5 < c.tol) { c.status = 0; c.phase = QN_PH_DONE; }
        else { c.after_state = QN_ST_AFTER_NEWTON; c.phase = QN_PH_REQ_NEWTON; } // compute_direction, newton/mod.rs:26-49
        return true;
    }
    double gnorm, gd0 = c.gd0;
    int d_finite = c.d_finite;
    if (!LEAN && c.small_n) { // reference order: norm = sqrt(dot), direction by column sweep (bfgs.rs:47)
        gnorm = sqrt(ref_dot(V.g, V.g, n));
        small_direction(V.H, n_pad, n, V.g, V.d, small_scratch, V.x, c.bounded ? V.lb : nullptr, c.bounded ? V.ub : nullptr);
        if (c.ls_kind == 2) { // morethuente_b.rs:185-198
            double cand = INFINITY;
            for (int i = 0; i < n; ++i) {
                const double di = V.d[i];
                double v = INFINITY;
                if (di > 0.0) v = (V.lub[i] - V.x[i]) / di; else if (di < 0.0) v = (V.llb[i] - V.x[i]) / di;
                cand = fmin(v, cand);
            }
            c.mtb_cand = cand;
        }
        gd0 = ref_dot(V.g, V.d, n);
        d_finite = 1;
        for (int i = 0; i < n; ++i) d_finite &= isfinite(V.d[i]) ? 1 : 0;
    } else {
        gnorm = sqrt(c.gg);
    }
    c.gnorm = gnorm; c.tr_f = c.f_k; c.tr_gnorm = gnorm;
    const double f = c.f_k;
    if (isnan(f) || isinf(f)) {
        c.status = 2; c.phase = QN_PH_DONE; // OutOfDomain
    } else if ((c.has_s_norm && c.s_norm < c.tol) || (c.has_y_norm && c.y_norm < c.tol) || (gnorm < c.tol)) {
        c.status = 0; c.phase = QN_PH_DONE;
    } else if (!LEAN && c.small_n) {
        c.gd0 = gd0; c.d_finite = d_finite; c.last_valid = 0;
        c.state = QN_ST_LS_BEGIN;
    } else if (c.have_dir) {
        c.state = QN_ST_LS_BEGIN;
        if (LEAN == 2 && c.s2_dir) {
            if (!c.dir_ready) { c.after_state = QN_ST_LS_BEGIN; c.phase = QN_PH_REQ_DIR; }
            else if (c.s2_dir & 2) c.mt_tmax = fmin(c.mt_tmax, c.mtb_cand); // (a call that continues behind a served request: the clip is the line search's)
        }
    } else { // bfgs.rs:47 d = -(H g): one pass over H (applies a pending update on the way)
        c.hp_nrhs = 1; c.hp_lazy = 0;
        c.after_state = QN_ST_AFTER_DIR;
        c.phase = QN_PH_REQ_HPASS;
    }
    return true;
}

template <int LEAN>
__device__ __forceinline__ void qn_st_ls_begin(QnCtl& c) {
    c.ls_i = 0;
    if (!LEAN && c.ls_kind == 2) c.mt_tmax = fmin(c.mt_tmax, c.mtb_cand); // morethuente_b.rs:201: self.t_max = self.t_max.min(candidate) -- persists
    if (c.ls_kind == 0 || (!LEAN && c.ls_kind == 2)) { // morethuente.rs:173-178
        c.use_mod = 0; c.conv = 0;
        c.t = fmin(fmax(1.0, c.mt_tmin), c.mt_tmax);
        c.tl = c.mt_tmin; c.tu = c.mt_tmax;
        c.state = QN_ST_MT_LOOP;
    } else { // backtracking.rs:28-29
        c.t = 1.0;
        c.state = QN_ST_BT_LOOP;
    }
}

template <int LEAN>
__device__ __forceinline__ void qn_st_mt_loop(QnCtl& c) { // morethuente.rs:181-182
    if (!(c.ls_i < c.max_iter_ls)) { c.ls_result = c.t; c.state = QN_ST_AFTER_LS; } // :295-296
    else { c.tr_ls_iters++; req_eval_t<LEAN>(c, c.t, QN_ST_MT_AFTER_T, 0); }
}

template <int LEAN>
__device__ __forceinline__ void qn_st_mt_after_t(QnCtl& c) { // morethuente.rs:184-217
    const double f_et = c.f_e, gd_t = c.gd_e, t = c.t;
    const bool wolfe = (f_et - c.f_k <= c.mt_c1 * t * c.gd0) && (fabs(gd_t) <= c.mt_c2 * fabs(c.gd0));
    if (wolfe || c.conv || t == c.tl || t == c.tu) {
        tr_push_case(c, 0);
        c.ls_result = t; c.state = QN_ST_AFTER_LS;
    } else {
        c.phi_t_f = f_et; c.phi_t_g = gd_t;
        c.psi_t_f = f_et - c.f_k - c.mt_c1 * t * c.gd0; // psi, :140-149
        c.psi_t_g = gd_t - c.mt_c1 * c.gd0;
        if (!c.use_mod && c.psi_t_f <= 0. && c.phi_t_g > 0.) { c.use_mod = 1; c.tr_ls_cases |= QN_LS_MODIFIED_BIT; } // :212-215 (sticky)
        req_eval_t<LEAN>(c, c.tl, QN_ST_MT_AFTER_TL, 0); // :217
    }
}

template <int LEAN>
__device__ __forceinline__ void qn_st_mt_after_tl(QnCtl& c) { // morethuente.rs:218-287
    const double phi_tl_f = c.f_e, phi_tl_g = c.gd_e;
    double f_tl, g_tl, f_t, g_t;
    if (c.use_mod) { f_tl = phi_tl_f; g_tl = phi_tl_g; f_t = c.phi_t_f; g_t = c.phi_t_g; }
    else {
        f_tl = phi_tl_f - c.f_k - c.mt_c1 * c.tl * c.gd0;
        g_tl = phi_tl_g - c.mt_c1 * c.gd0;
        f_t = c.psi_t_f; g_t = c.psi_t_g;
    }
    c.sel_f_tl = f_tl; c.sel_g_tl = g_tl; c.sel_f_t = f_t; c.sel_g_t = g_t;
    const double t = c.t, tl = c.tl, tu = c.tu;
    if (f_t > f_tl) { // case 1
        const double tc = mt_cubic(tl, t, f_tl, f_t, g_tl, g_t);
        const double tq = mt_quad1(tl, t, f_tl, f_t, g_tl);
        tr_push_case(c, 1);
        c.t = (fabs(tc - tl) < fabs(tq - tl)) ? tc : 0.5 * (tq + tc);
        c.state = QN_ST_MT_FINISH;
    } else if (g_t * g_tl < 0.) { // case 2
        const double tc = mt_cubic(tl, t, f_tl, f_t, g_tl, g_t);
        const double ts = mt_quad2(tl, t, g_tl, g_t);
        tr_push_case(c, 2);
        c.t = (fabs(tc - t) >= fabs(ts - t)) ? tc : ts;
        c.state = QN_ST_MT_FINISH;
    } else if (fabs(g_t) <= fabs(g_tl)) { // case 3
        const double tc = mt_cubic(tl, t, f_tl, f_t, g_tl, g_t);
        const double ts = mt_quad2(tl, t, g_tl, g_t);
        tr_push_case(c, 3);
        const double t_plus = (fabs(tc - t) < fabs(ts - t)) ? tc : ts;
        if (t > tl) c.t = fmin(t_plus, t + c.mt_delta * (tu - t));
        else c.t = fmax(t_plus, t + c.mt_delta * (tu - t));
        c.state = QN_ST_MT_FINISH;
    } else { // case 4: evaluates at tu (possibly +inf), :274-287
        req_eval_t<LEAN>(c, c.tu, QN_ST_MT_AFTER_TU, 0);
    }
}

template <int LEAN>
__device__ __forceinline__ void qn_st_mt_after_tu(QnCtl& c) {
    double f_tu, g_tu;
    if (c.use_mod) { f_tu = c.f_e; g_tu = c.gd_e; }
    else { f_tu = c.f_e - c.f_k - c.mt_c1 * c.tu * c.gd0; g_tu = c.gd_e - c.mt_c1 * c.gd0; }
    tr_push_case(c, 4);
    c.t = mt_cubic(c.tu, c.t, c.sel_f_t, f_tu, c.sel_g_t, g_tu); // :286, argument order as written
    c.state = QN_ST_MT_FINISH;
}

template <int LEAN>
__device__ __forceinline__ void qn_st_mt_finish(QnCtl& c) { // morethuente.rs:290-293: the NEW t with the OLD trial's f_t, g_t
    c.t = fmin(fmax(c.t, c.mt_tmin), c.mt_tmax);
    double tl = c.tl, tu = c.tu;
    c.conv = mt_update_interval(c.sel_f_tl, c.sel_f_t, c.sel_g_t, tl, c.t, tu);
    c.tl = tl; c.tu = tu;
    c.ls_i++;
    c.state = QN_ST_MT_LOOP;
}

template <int LEAN>
__device__ __forceinline__ void qn_st_bt_loop(QnCtl& c) { // backtracking.rs:31-34
    if (!(c.max_iter_ls > c.ls_i)) { c.ls_result = c.t; c.state = QN_ST_AFTER_LS; } // :54
    else { c.tr_ls_iters++; req_eval_t<LEAN>(c, c.t, QN_ST_BT_AFTER, 0, (LEAN != 1 && c.ls_kind == 3) ? 1 : 0); } // (LEAN == 2: BackTrackingB on the second-generation path, round 6)
}

template <int LEAN>
__device__ __forceinline__ void qn_st_bt_after(QnCtl& c) { // backtracking.rs:37-51
    const double f1 = c.f_e;
    if (isnan(f1) || isinf(f1)) { c.t *= c.bt_beta; c.state = QN_ST_BT_LOOP; } // shrink, iteration not counted
    else if ((LEAN != 1 && c.ls_kind == 3) ? (f1 - c.f_k <= (-c.bt_c1 / c.t) * c.bt_diff2) // backtracking_b.rs:24-34
                            : (f1 - c.f_k <= c.bt_c1 * c.t * c.gd0)) { c.ls_result = c.t; c.state = QN_ST_AFTER_LS; }
    else { c.t *= c.bt_beta; c.ls_i++; c.state = QN_ST_BT_LOOP; }
}

template <int LEAN>
__device__ __forceinline__ bool qn_st_after_ls(QnCtl& c) {
    if (!LEAN && c.ls_only) { c.status = 0; c.phase = QN_PH_DONE; return true; } // compute_step_len returns the step, nothing else
    if (!LEAN && (c.method == 2 || c.method == 3)) return false; // gradient descent / Newton: the default hook x += step*d needs all threads
    req_eval_t<LEAN>(c, c.ls_result, QN_ST_AFTER_NEXT, 1); // bfgs.rs:94,98: oracle(x + step*d)
    return true;
}

template <int LEAN>
__device__ __forceinline__ bool qn_st_iter_end(QnCtl& c, const QnVecs& V, const bool side_effects) { // ls_solver.rs:104-107
    const bool rec = c.k < c.trace_cap;
    if (rec && c.trace_x && !c.xtrace_done) return false; // the iterate has to be copied by all threads first
    if (rec) {
        QnTraceRec r;
        r.f = c.tr_f; r.gnorm = c.tr_gnorm; r.t = c.ls_result;
        r.s_norm = c.has_s_norm ? c.s_norm : NAN;
        r.y_norm = c.has_y_norm ? c.y_norm : NAN;
        r.n_evals = c.tr_n_evals; r.ls_iters = c.tr_ls_iters; r.ls_cases = c.tr_ls_cases; r.updated = c.tr_updated;
        if (side_effects) V.trace[c.k] = r;
    }
    c.xtrace_done = 0;
    c.k += 1;
    c.n_iterations++;
    c.state = QN_ST_LOOP_TOP;
    if (!LEAN && c.callback_mode) c.phase = QN_PH_ITER_DONE;
    return true;
}

// run state function CALL when the machine is still running and in state ST; otherwise back to the dispatcher
#define QN_RUN_NEXT(ST, CALL)                                           \
    if (!(c.phase == QN_PH_RUNNING && c.state == (ST))) break;          \
    CALL;
#define QN_RUN_NEXT_B(ST, CALL)                                         \
    if (!(c.phase == QN_PH_RUNNING && c.state == (ST))) break;          \
    if (!(CALL)) return;

template <int LEAN = 0>
__device__ __forceinline__ void ctl_scalar_run(QnCtl& c, const QnVecs& V, double* small_scratch, const bool side_effects = true) {
    for (int guard = 0; guard < (1 << 22); ++guard) {
        if (c.phase != QN_PH_RUNNING) return;
        switch (c.state) {
        // (only the successions of the steady iteration are chained: every chained copy of a state is code, and the machine is
        // inlined into every sym2 kernel)
        case QN_ST_BEGIN:
            qn_st_begin<LEAN>(c);
            qn_st_loop_top<LEAN>(c);
            break;

        case QN_ST_LOOP_TOP: qn_st_loop_top<LEAN>(c); break;

        case QN_ST_AFTER_EVALX:
            if (!qn_st_after_evalx<LEAN>(c)) return;
            break;

        case QN_ST_AFTER_DIR:
            if (!qn_st_after_dir<LEAN>(c)) return;
            if (LEAN == 2 && c.phase != QN_PH_RUNNING) break; // (the direction goes through QN_PH_REQ_DIR first)
            qn_st_ls_begin<LEAN>(c);
            QN_RUN_NEXT(QN_ST_MT_LOOP, qn_st_mt_loop<LEAN>(c))
            break;

        case QN_ST_AFTER_NEXT:
            if (!qn_st_after_next<LEAN>(c)) return;
            break;

        case QN_ST_AFTER_U: // ... the iteration ends, the next one begins: up to the first trial of its line search
            if (!qn_st_after_u<LEAN>(c)) return;
            QN_RUN_NEXT_B(QN_ST_ITER_END, qn_st_iter_end<LEAN>(c, V, side_effects))
            QN_RUN_NEXT(QN_ST_LOOP_TOP, qn_st_loop_top<LEAN>(c))
            QN_RUN_NEXT_B(QN_ST_CHECK, qn_st_check<LEAN>(c, V, small_scratch))
            QN_RUN_NEXT(QN_ST_LS_BEGIN, qn_st_ls_begin<LEAN>(c))
            QN_RUN_NEXT(QN_ST_MT_LOOP, qn_st_mt_loop<LEAN>(c))
            break;

        case QN_ST_CHECK:
            if (!qn_st_check<LEAN>(c, V, small_scratch)) return;
            QN_RUN_NEXT(QN_ST_LS_BEGIN, qn_st_ls_begin<LEAN>(c))
            QN_RUN_NEXT(QN_ST_MT_LOOP, qn_st_mt_loop<LEAN>(c))
            break;

        case QN_ST_LS_BEGIN:
            qn_st_ls_begin<LEAN>(c);
            QN_RUN_NEXT(QN_ST_MT_LOOP, qn_st_mt_loop<LEAN>(c))
            break;

        case QN_ST_MT_LOOP: qn_st_mt_loop<LEAN>(c); break;

        case QN_ST_MT_AFTER_T: // a trial came back: accepted (the step leaves the search), or the next trial from the memo of phi(tl)
            qn_st_mt_after_t<LEAN>(c);
            if (c.phase == QN_PH_RUNNING && c.state == QN_ST_AFTER_LS) { if (!qn_st_after_ls<LEAN>(c)) return; break; }
            QN_RUN_NEXT(QN_ST_MT_AFTER_TL, qn_st_mt_after_tl<LEAN>(c))
            QN_RUN_NEXT(QN_ST_MT_FINISH, qn_st_mt_finish<LEAN>(c))
            qn_st_mt_loop<LEAN>(c);
            break;

        case QN_ST_MT_AFTER_TL:
            qn_st_mt_after_tl<LEAN>(c);
            QN_RUN_NEXT(QN_ST_MT_FINISH, qn_st_mt_finish<LEAN>(c))
            qn_st_mt_loop<LEAN>(c);
            break;

        case QN_ST_MT_AFTER_TU:
            qn_st_mt_after_tu<LEAN>(c);
            qn_st_mt_finish<LEAN>(c);
            qn_st_mt_loop<LEAN>(c);
            break;

        case QN_ST_MT_FINISH:
            qn_st_mt_finish<LEAN>(c);
            qn_st_mt_loop<LEAN>(c);
            break;

        case QN_ST_BT_LOOP: qn_st_bt_loop<LEAN>(c); break;

        case QN_ST_BT_AFTER:
            qn_st_bt_after<LEAN>(c);
            if (c.phase == QN_PH_RUNNING && c.state == QN_ST_AFTER_LS) { if (!qn_st_after_ls<LEAN>(c)) return; break; }
            QN_RUN_NEXT(QN_ST_BT_LOOP, qn_st_bt_loop<LEAN>(c))
            break;

        case QN_ST_LS_ONLY: return; // g.d needs all threads

        case QN_ST_AFTER_LS:
            if (!qn_st_after_ls<LEAN>(c)) return;
            break;

        case QN_ST_ITER_END:
            if (!qn_st_iter_end<LEAN>(c, V, side_effects)) return;
            break;

        default:
            return; // a vector state
        }
    }
    c.status = 4; // guard tripped
    c.phase = QN_PH_DONE;
}
#undef QN_RUN_NEXT
#undef QN_RUN_NEXT_B

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
template <int ORACLE>
__global__ __launch_bounds__(QN_CTL_TPB) void ctl_step_kernel(QnCtl* __restrict__ gctl, const QnVecs V, const int expect_mask) {
    __shared__ QnCtl c;
    __shared__ double lds[16 * QN_NEVP]; // block sums; fused path: [0, 9) evaluation totals, [9, 12) update-pass totals
    __shared__ double small_scratch[5 * QN_SMALL_N * QN_SMALL_N + QN_SMALL_N];
    const int tid = threadIdx.x;
    const int tpb = blockDim.x;
    const int n = V.n, n_pad = V.n_pad;
    const bool fused = V.fused_hint != 0; // host-known: lets the fused step start its loads before the control block arrives
    if (fused) {
        // the control block and the partial sums are fetched together (one memory round trip instead of two); a
        // predicated-off launch wastes the partial loads, nothing else
        constexpr int NW = (int)(sizeof(QnCtl) / 8);
        uint64_t cw = 0;
        if (tid < NW) cw = reinterpret_cast<const uint64_t*>(gctl)[tid];
        const bool want_ev = (expect_mask & ((1 << QN_PH_REQ_EVAL) | (1 << QN_PH_REQ_HPASS_EVAL))) != 0;
        const bool want_hp = (expect_mask & ((1 << QN_PH_REQ_HPASS) | (1 << QN_PH_REQ_HPASS_EVAL))) != 0;
        if (want_ev) ctl_sum_partials<QN_NEVP>(V.F.evp, V.F.pworld, V.F.nblk, lds);                                   // waves 0..8
        if (want_hp) ctl_sum_partials<QN_NHPP>(V.F.hpp, V.F.pworld, V.F.nblk, lds + QN_NEVP, want_ev ? QN_NEVP : 0); // waves 9..11
        if (tid < NW) reinterpret_cast<uint64_t*>(&c)[tid] = cw;
        __syncthreads();
        if (!((1 << c.phase) & expect_mask)) return;
    } else {
        if (!((1 << gctl->phase) & expect_mask)) return;
    }
    int expect_phase = fused ? c.phase : gctl->phase; // the request actually being consumed
#ifdef QN_CTL_STAMPS
    const long stamp_base = V.dbg ? (long)(V.dbg[0] & 0xffff) * 16 + 16 : 0;
    if (tid == 0 && V.dbg) { V.dbg[0] = V.dbg[0] + 1; V.dbg[stamp_base + 15] = expect_phase; }
#endif
    QN_STAMP(0);
    double* __restrict__ const vx = V.x;
    double* __restrict__ const vg = V.g;
    double* __restrict__ const vd = V.d;
    double* __restrict__ const vxt = V.xt;
    double* __restrict__ const vgt = V.gt;
    double* __restrict__ const vs = V.s;
    double* __restrict__ const vy = V.y;
    double* __restrict__ const vsp = V.sp;
    double* __restrict__ const vup = V.up;
    const double* __restrict__ const vb = V.b;

    // ---- consume the serviced evaluation: f, g at xt, and g.d, in one sweep; warm x and g for AFTER_NEXT ----
    double cons_f = 0.0, cons_gd = 0.0;
    if (!fused && expect_phase == QN_PH_REQ_EVAL) {
        double p[3] = {0.0, 0.0, 0.0};
        for (int base = 0; base < n_pad; base += 4 * tpb) {
            QN_TILE_IDX(base)
            double qv[4], xv[4], bv[4], dv[4], px[4], pg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ORACLE == QN_ORACLE_QUAD) { qv[u] = q_val(V, idx[u]); xv[u] = vxt[idx[u]]; bv[u] = vb[idx[u]]; }
                else { qv[u] = vgt[idx[u]]; xv[u] = 0.0; bv[u] = 0.0; }
                dv[u] = vd[idx[u]];
                px[u] = vx[idx[u]];
                pg[u] = vg[idx[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ok[u]) {
                    double gt = qv[u];
                    if (ORACLE == QN_ORACLE_QUAD) { // f = 1/2 xt'(Q xt) - b'xt ; g = Q xt - b
                        gt = qv[u] - bv[u];
                        vgt[idx[u]] = gt;
                        p[0] = __builtin_fma(xv[u], qv[u], p[0]);
                        p[1] = __builtin_fma(bv[u], xv[u], p[1]);
                    }
                    p[2] = __builtin_fma(gt, dv[u], p[2]);
                }
                qn_keepalive(px[u]);
                qn_keepalive(pg[u]);
            }
        }
        QN_STAMP(1);
        ctl_block_sum<3, true>(p, lds);
        cons_f = (ORACLE == QN_ORACLE_QUAD) ? (0.5 * p[0] - p[1]) : *V.f_dev;
        cons_gd = p[2];
    }
    if (!fused) {
        const uint64_t* src = reinterpret_cast<const uint64_t*>(gctl);
        uint64_t* dst = reinterpret_cast<uint64_t*>(&c);
        for (int i = tid; i < (int)(sizeof(QnCtl) / 8); i += tpb) dst[i] = src[i];
        __syncthreads();
    }
    double cons_diff2 = 0.0;
    if (!fused && expect_phase == QN_PH_REQ_EVAL && c.req_project) { // BackTrackingB: ||P(x + t d) - x||^2
        double p[1] = {0.0};
        for (int i = tid; i < n_pad; i += tpb) { const double df = vxt[i] - vx[i]; p[0] = __builtin_fma(df, df, p[0]); }
        ctl_block_sum<1, true>(p, lds);
        if (c.small_n) {
            __syncthreads();
            if (tid == 0) { double acc8[QN_SMALL_N]; for (int i = 0; i < n; ++i) acc8[i] = V.xt[i] - V.x[i]; p[0] = ref_dot(acc8, acc8, n); }
        }
        cons_diff2 = p[0];
    }
    QN_STAMP(2);
    if (tid == 0) {
        if (expect_phase == QN_PH_REQ_EVAL) { c.bt_diff2 = cons_diff2; c.last_projected = c.req_project; }
        if (fused && expect_phase == QN_PH_REQ_HPASS_EVAL) { // commit the update the evaluation kernel already used
            const double yu = lds[QN_NEVP + 0];
            qn_update_coeffs(c.method, c.ys, yu, c.c_ss, c.c_su, c.c_uu);
            c.n_hpasses++;
            if (c.pending) c.n_hpass_rw++;
            c.pending = 1;
            c.sc ^= 1; // the staged s becomes the pending s; the new u is already in UN
            c.dir_mode = 1; c.dir_ug = lds[QN_NEVP + 1]; c.dir_sg = lds[QN_NEVP + 2];
            c.defer_u = 0;
            expect_phase = QN_PH_REQ_EVAL; // from here on: an ordinary evaluation result
        }
        if (fused && expect_phase == QN_PH_REQ_EVAL) {
            cons_f = 0.5 * lds[0] - lds[1]; // f = 1/2 x+'(Q x+) - b'x+
            cons_gd = lds[2];
            c.st_gd0 = lds[3]; c.st_yy = lds[4]; c.st_ys = lds[5]; c.st_gg = lds[6]; c.st_ss = lds[7]; c.st_dnf = lds[8];
            if (c.req_kind == QN_REQ_T && !c.gd0_valid) { c.gd0 = lds[3]; c.d_finite = lds[8] == 0.0; c.gd0_valid = 1; }
        }
        if (fused && expect_phase == QN_PH_REQ_HPASS) { c.hp_yu = lds[QN_NEVP + 0]; c.hp_ug = lds[QN_NEVP + 1]; c.hp_sg = lds[QN_NEVP + 2]; }
        if (expect_phase == QN_PH_REQ_EVAL) {
            const int kind = c.req_kind;
            if (c.small_n && kind == QN_REQ_T) cons_gd = ref_dot(V.gt, V.d, n);
            c.f_e = cons_f;
            c.gd_e = cons_gd;
            c.n_oracle_evals++;
            if (kind == QN_REQ_T) { c.last_valid = 1; c.last_t = c.req_t; c.f_last = cons_f; c.gd_last = cons_gd; }
            else c.last_valid = 0;
        }
        if (expect_phase == QN_PH_IDLE) c.state = c.ls_only ? QN_ST_LS_ONLY : QN_ST_BEGIN;
        else if (expect_phase == QN_PH_REQ_EVAL || expect_phase == QN_PH_REQ_HPASS || expect_phase == QN_PH_REQ_NEWTON) c.state = c.after_state;
        // (QN_PH_REQ_HPASS_EVAL was turned into QN_PH_REQ_EVAL by the commit above)
        c.phase = QN_PH_RUNNING;
        ctl_scalar_run(c, V, small_scratch);
    }
    QN_STAMP(3);

    for (int guard = 0; guard < (1 << 20); ++guard) {
        __syncthreads();
        if (c.phase != QN_PH_RUNNING) break;
        const int st = c.state;
        QN_STAMP(4 + 3 * (guard < 3 ? guard : 2));
        switch (st) {
        case QN_ST_AFTER_EVALX: { // g <- gt, ||g||^2
            double p[1] = {0.0};
            for (int base = 0; base < n_pad; base += 4 * tpb) {
                QN_TILE_IDX(base)
                double gv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) gv[u] = vgt[idx[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u]) { vg[idx[u]] = gv[u]; p[0] = __builtin_fma(gv[u], gv[u], p[0]); }
            }
            ctl_block_sum<1, true>(p, lds);
            if (tid == 0) {
                c.f_k = c.f_e;
                c.gg = p[0]; c.gg_valid = 1;
                c.have_cur_eval = c.memoize;
                c.have_dir = 0;
                c.state = QN_ST_CHECK;
            }
        } break;

        case QN_ST_CHECK: { // vector variant: gradient descent (inf-norm, d = -g), or ||g|| not known yet
            const bool gd_method = c.method == 2;
            double gnorm, p[2] = {0.0, 0.0};
            if (gd_method) {
                double m = -INFINITY; // fold(NEG_INFINITY, |acc, x| x.abs().max(acc)): NaN entries are ignored
                for (int i = tid; i < n; i += tpb) {
                    const double gi = vg[i];
                    m = fmax(fabs(gi), m);
                    const double di = -gi; // gradient_descent.rs:29
                    vd[i] = di;
                    p[0] = __builtin_fma(gi, di, p[0]);
                    p[1] += isfinite(di) ? 0.0 : 1.0;
                }
                gnorm = ctl_block_fmax<true>(m, lds);
                ctl_block_sum<2, true>(p, lds);
                if (c.small_n) {
                    __syncthreads();
                    if (tid == 0) p[0] = ref_dot(V.g, V.d, n);
                }
            } else {
                for (int i = tid; i < n_pad; i += tpb) { const double gi = vg[i]; p[0] = __builtin_fma(gi, gi, p[0]); }
                ctl_block_sum<2, true>(p, lds);
                gnorm = sqrt(p[0]);
            }
            if (tid == 0) {
                if (!gd_method) { // now ||g||^2 is known: the scalar variant finishes the state
                    c.gg = p[0]; c.gg_valid = 1;
                } else {
                    c.gnorm = gnorm; c.tr_f = c.f_k; c.tr_gnorm = gnorm;
                    const double f = c.f_k;
                    if (isnan(f) || isinf(f)) { c.status = 2; c.phase = QN_PH_DONE; }
                    else if (gnorm < c.tol) { c.status = 0; c.phase = QN_PH_DONE; } // gradient_descent.rs:46-53
                    else { c.gd0 = p[0]; c.d_finite = p[1] == 0.0; c.last_valid = 0; c.state = QN_ST_LS_BEGIN; }
                }
            }
        } break;

        case QN_ST_AFTER_DIR: { // d = -(H g) from the gathered h_pass output (bounded: P(x - H g) - x), g.d
            double p[2] = {0.0, 0.0};
            const bool bounded = c.bounded != 0, want_cand = c.ls_kind == 2;
            double cand = INFINITY;
            for (int base = 0; base < n_pad; base += 4 * tpb) {
                QN_TILE_IDX(base)
                double hv[4], gv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { hv[u] = hp_val(V, 1, 0, idx[u]); gv[u] = vg[idx[u]]; }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u]) {
                        double di = -hv[u];
                        if (bounded) { const double xi = vx[idx[u]]; double t = xi - hv[u]; t = fmin(fmax(t, V.lb[idx[u]]), V.ub[idx[u]]); di = t - xi; } // bfgs_b.rs:72-75
                        vd[idx[u]] = di;
                        p[0] = __builtin_fma(gv[u], di, p[0]);
                        p[1] += isfinite(di) ? 0.0 : 1.0;
                        if (want_cand) { // morethuente_b.rs:185-198
                            const double xi = vx[idx[u]];
                            double v = INFINITY;
                            if (di > 0.0) v = (V.lub[idx[u]] - xi) / di; else if (di < 0.0) v = (V.llb[idx[u]] - xi) / di;
                            cand = fmin(v, cand);
                        }
                    }
            }
            ctl_block_sum<2, true>(p, lds);
            if (want_cand) cand = -ctl_block_fmax<true>(-cand, lds);
            if (tid == 0) {
                c.mtb_cand = cand;
                c.n_hpasses++;
                if (c.pending) c.n_hpass_rw++;
                c.pending = 0;
                c.gd0 = p[0]; c.d_finite = p[1] == 0.0; c.last_valid = 0;
                c.state = QN_ST_LS_BEGIN;
            }
        } break;

        case QN_ST_LS_ONLY: { // x, g = g(x), d uploaded by the host; phi'(0) = g.d (morethuente.rs:137 / backtracking.rs:32)
            double p[2] = {0.0, 0.0};
            const bool want_cand = c.ls_kind == 2;
            double cand = INFINITY;
            for (int i = tid; i < n_pad; i += tpb) {
                const double gi = vg[i], di = vd[i];
                p[0] = __builtin_fma(gi, di, p[0]);
                p[1] += isfinite(di) ? 0.0 : 1.0;
                if (want_cand && i < n) { // morethuente_b.rs:185-198
                    const double xi = vx[i];
                    double v = INFINITY;
                    if (di > 0.0) v = (V.lub[i] - xi) / di; else if (di < 0.0) v = (V.llb[i] - xi) / di;
                    cand = fmin(v, cand);
                }
            }
            ctl_block_sum<2, true>(p, lds);
            if (want_cand) cand = -ctl_block_fmax<true>(-cand, lds);
            if (c.small_n) {
                __threadfence_block();
                __syncthreads();
                if (tid == 0) p[0] = ref_dot(V.g, V.d, n);
            }
            if (tid == 0) {
                c.k = 0; c.status = -1;
                c.n_oracle_calls = 0; c.n_oracle_evals = 0;
                c.tr_n_evals = 0; c.tr_ls_iters = 0; c.tr_ls_cases = 0; c.tr_ndigits = 0; c.tr_updated = 0;
                c.mtb_cand = cand;
                c.gd0 = p[0]; c.d_finite = p[1] == 0.0; c.last_valid = 0;
                c.ls_result = NAN;
                c.state = QN_ST_LS_BEGIN;
            }
        } break;

        case QN_ST_AFTER_NEWTON: { // d and z = H^-1 d were left in V.d / V.s by the factorisation + solves
            const bool failed = *V.nfail != 0;
            double p[3] = {0.0, 0.0, 0.0};
            for (int i = tid; i < n_pad; i += tpb) {
                const double gi = vg[i];
                double di;
                if (failed) { di = -gi; vd[i] = di; } // singular Hessian: gradient-descent direction (newton/mod.rs:43-46)
                else di = vd[i];
                p[0] = __builtin_fma(gi, di, p[0]);
                p[1] += isfinite(di) ? 0.0 : 1.0;
                if (!failed) p[2] = __builtin_fma(vs[i], di, p[2]); // (hessian_inv * direction).dot(direction)
            }
            ctl_block_sum<3, true>(p, lds);
            if (c.small_n) {
                __threadfence_block();
                __syncthreads();
                if (tid == 0) { p[0] = ref_dot(V.g, V.d, n); if (!failed) p[2] = ref_dot(V.s, V.d, n); }
            }
            if (tid == 0) {
                if (!failed) { c.dec = p[2]; c.has_dec = 1; }
                c.gd0 = p[0]; c.d_finite = p[1] == 0.0; c.last_valid = 0;
                c.state = QN_ST_LS_BEGIN;
            }
        } break;

        case QN_ST_AFTER_LS: { // gradient descent / Newton (ls_solver.rs:44-64 / gradient_descent.rs:55-82): x += step*d
            const double step = c.ls_result;
            const bool hit = c.last_valid && c.last_t == step;
            const bool memo = c.memoize != 0;
            double p[1] = {0.0};
            for (int i = tid; i < n_pad; i += tpb) {
                if (hit) {
                    vx[i] = vxt[i];
                    if (memo) { const double gi = vgt[i]; vg[i] = gi; p[0] = __builtin_fma(gi, gi, p[0]); }
                } else {
                    const double td = step * vd[i];
                    vx[i] = vx[i] + td;
                }
            }
            ctl_block_sum<1, true>(p, lds);
            if (tid == 0) {
                if (hit && memo) { c.f_k = c.f_last; c.have_cur_eval = 1; } else c.have_cur_eval = 0;
                c.gg_valid = 0;
                c.last_valid = 0;
                c.state = QN_ST_ITER_END;
            }
        } break;

        case QN_ST_AFTER_NEXT: { // bfgs.rs:94-102 in one sweep: s, y, ||s||, ||y||, y.s, x <- x+, g <- g+, ||g+||^2
            double p[4] = {0.0, 0.0, 0.0, 0.0};
            for (int base = 0; base < n_pad; base += 4 * tpb) {
                QN_TILE_IDX(base)
                double xn[4], gn[4], xo[4], go[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { xn[u] = vxt[idx[u]]; gn[u] = vgt[idx[u]]; xo[u] = vx[idx[u]]; go[u] = vg[idx[u]]; }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u]) {
                        const double si = xn[u] - xo[u]; // s = x+ - x (not t*d), :96
                        const double yi = gn[u] - go[u]; // :98
                        vs[idx[u]] = si; vy[idx[u]] = yi; vx[idx[u]] = xn[u]; vg[idx[u]] = gn[u];
                        p[0] = __builtin_fma(si, si, p[0]);
                        p[1] = __builtin_fma(yi, yi, p[1]);
                        p[2] = __builtin_fma(yi, si, p[2]);
                        p[3] = __builtin_fma(gn[u], gn[u], p[3]);
                    }
            }
            ctl_block_sum<4, true>(p, lds);
            if (c.small_n) {
                __threadfence_block();
                __syncthreads();
                if (tid == 0) { p[0] = ref_dot(V.s, V.s, n); p[1] = ref_dot(V.y, V.y, n); p[2] = ref_dot(V.y, V.s, n); }
            }
            if (tid == 0) {
                c.s_norm = sqrt(p[0]); c.has_s_norm = 1;
                c.y_norm = sqrt(p[1]); c.has_y_norm = 1;
                c.ys = p[2];
                c.gg = p[3]; c.gg_valid = 1;
                c.f_k = c.f_e;
                c.have_cur_eval = c.memoize;
                c.have_dir = 0;
                c.last_valid = 0;
                if (c.s_norm < c.tol || c.y_norm < c.tol) { // bfgs.rs:106-112: H is not updated
                    c.state = QN_ST_ITER_END;
                } else if (c.small_n) {
                    small_update(V.H, n_pad, n, V.s, V.y, c.method, small_scratch);
                    c.tr_updated = 1;
                    c.state = QN_ST_ITER_END;
                } else {
                    c.hp_lazy = c.memoize;
                    c.hp_nrhs = c.memoize ? 2 : 1; // u = H y (and v = H g+ when the next direction may be formed lazily)
                    c.after_state = QN_ST_AFTER_U;
                    c.phase = QN_PH_REQ_HPASS;
                }
            }
        } break;

        case QN_ST_AFTER_U: { // bfgs.rs:115-124 / dfp.rs:115-120 in rank-2 form; the update itself is applied by the next h_pass
            const int nrhs = c.hp_nrhs;
            const bool lazy = c.hp_lazy != 0;
            const int method = c.method;
            const double ys = c.ys;
            double p[4] = {0.0, 0.0, 0.0, 0.0};
            for (int base = 0; base < n_pad; base += 4 * tpb) {
                QN_TILE_IDX(base)
                double uv[4], sv[4], yv[4], gv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uv[u] = hp_val(V, nrhs, 0, idx[u]); sv[u] = vs[idx[u]]; yv[u] = vy[idx[u]]; gv[u] = vg[idx[u]];
                    if (lazy) qn_keepalive(hp_val(V, nrhs, 1, idx[u])); // warm v for the second sweep
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u]) {
                        vup[idx[u]] = uv[u]; vsp[idx[u]] = sv[u];
                        p[0] = __builtin_fma(yv[u], uv[u], p[0]);
                        p[1] = __builtin_fma(uv[u], gv[u], p[1]);
                        p[2] = __builtin_fma(sv[u], gv[u], p[2]);
                        p[3] = __builtin_fma(sv[u] - uv[u], yv[u], p[3]); // SR1: (s - H y).y, sr1_b.rs:145
                    }
            }
            ctl_block_sum<4, true>(p, lds);
            const double yu = p[0], ug = p[1], sg = p[2];
            double c_ss, c_su, c_uu;
            if (method == 0) { const double rho = 1.0 / ys; c_su = -rho; c_ss = rho * rho * yu + rho; c_uu = 0.0; }
            else if (method == 4) { c_ss = 1.0 / p[3]; c_su = -c_ss; c_uu = c_ss; } // (s-u)(s-u)'/((s-u).y) = c (ss' - (su' + us') + uu')
            else { c_ss = 1.0 / ys; c_su = 0.0; c_uu = -1.0 / yu; }
            double q[2] = {0.0, 0.0};
            double lazy_cand = INFINITY;
            if (lazy) { // d+ = -(H+ g+) = -(v + c_su (s (u.g) + u (s.g)) + c_ss s (s.g) + c_uu u (u.g)),  v = H g+
                for (int base = 0; base < n_pad; base += 4 * tpb) {
                    QN_TILE_IDX(base)
                    double uv[4], sv[4], vv[4], gv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { uv[u] = vup[idx[u]]; sv[u] = vsp[idx[u]]; vv[u] = hp_val(V, nrhs, 1, idx[u]); gv[u] = vg[idx[u]]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (ok[u]) {
                            double w = vv[u];
                            if (c_su != 0.0) w = w + c_su * (sv[u] * ug + uv[u] * sg);
                            w = w + c_ss * (sv[u] * sg);
                            if (c_uu != 0.0) w = w + c_uu * (uv[u] * ug);
                            double di = -w;
                            if (c.bounded) { const double xi = vx[idx[u]]; double t = xi - w; t = fmin(fmax(t, V.lb[idx[u]]), V.ub[idx[u]]); di = t - xi; }
                            vd[idx[u]] = di;
                            q[0] = __builtin_fma(gv[u], di, q[0]);
                            q[1] += isfinite(di) ? 0.0 : 1.0;
                            if (c.ls_kind == 2) {
                                const double xi = vx[idx[u]];
                                double v = INFINITY;
                                if (di > 0.0) v = (V.lub[idx[u]] - xi) / di; else if (di < 0.0) v = (V.llb[idx[u]] - xi) / di;
                                lazy_cand = fmin(v, lazy_cand);
                            }
                        }
                }
                ctl_block_sum<2, true>(q, lds);
                if (c.ls_kind == 2) lazy_cand = -ctl_block_fmax<true>(-lazy_cand, lds);
            }
            if (tid == 0) {
                c.n_hpasses++;
                if (c.pending) c.n_hpass_rw++;
                c.c_ss = c_ss; c.c_su = c_su; c.c_uu = c_uu;
                c.pending = 1;
                c.tr_updated = 1;
                if (lazy) { c.gd0 = q[0]; c.d_finite = q[1] == 0.0; c.have_dir = 1; c.mtb_cand = lazy_cand; }
                c.state = QN_ST_ITER_END;
            }
        } break;

        case QN_ST_ITER_END: { // only reached here when the iterate has to be recorded (trace with x)
            double* row = V.xtrace + (size_t)c.k * (size_t)n;
            const double* xs = c.fused ? V.F.X0 + (size_t)c.xc * (size_t)n_pad : vx;
            for (int i = tid; i < n; i += tpb) row[i] = xs[i];
            if (tid == 0) {
                c.xtrace_done = 1;
            }
        } break;

        default: { // a scalar state left over (cannot happen: ctl_scalar_run consumes them) -> abort instead of spinning
            if (tid == 0) { c.status = 4; c.phase = QN_PH_DONE; }
        } break;
        }
        QN_STAMP(5 + 3 * (guard < 3 ? guard : 2));
        if (tid == 0) ctl_scalar_run(c, V, small_scratch); // chain the scalar states that follow without further barriers
        QN_STAMP(6 + 3 * (guard < 3 ? guard : 2));
    }
    QN_STAMP(13);
    __syncthreads();
    if (tid == 0 && c.phase == QN_PH_RUNNING) { c.status = 4; c.phase = QN_PH_DONE; } // guard tripped
    __syncthreads();
    {
        uint64_t* dst = reinterpret_cast<uint64_t*>(gctl);
        const uint64_t* src = reinterpret_cast<const uint64_t*>(&c);
        for (int i = tid; i < (int)(sizeof(QnCtl) / 8); i += tpb) dst[i] = src[i];
    }
    QN_STAMP(14);
}
