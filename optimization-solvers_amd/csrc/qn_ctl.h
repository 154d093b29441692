// qn_ctl.h -- the device-resident control block shared by host pump and kernels.
//
// The whole solver (ls_solver.rs:66-111 driver, bfgs.rs:64-127 / dfp.rs hooks, morethuente.rs:165-297,
// backtracking.rs:20-58) runs as ONE state machine inside the single-workgroup kernel `ctl_step`; it
// yields only when it needs an oracle evaluation, a pass over H, or a host callback.  The host never takes
// a line-search decision: it services requests (sync mode) or enqueues a fixed pattern of predicated
// kernels and polls `phase` now and then (pipelined mode).
#pragma once
#include <stdint.h>

// what the state machine is waiting for
enum QnPhase : int32_t {
    QN_PH_IDLE = 0,      // between qn_minimize calls / ready to start
    QN_PH_REQ_EVAL = 1,  // evaluate the oracle at (req_kind, req_t)
    QN_PH_REQ_HPASS = 2, // run h_pass with hp_nrhs right-hand sides
    QN_PH_ITER_DONE = 3, // host callback wanted (ls_solver.rs:105-107)
    QN_PH_DONE = 4,      // finished: status holds the SolverError code
    QN_PH_REQ_HPASS_EVAL = 7, // fused path: run h_pass, then the evaluation at (req_kind, req_t) whose direction uses the
                              // coefficients of the update this pass completes (derived from the pass's partial sums)
    QN_PH_REQ_VEC = 8,   // sym2: turn the slots of the last evaluation into vectors (g+, y, x+, s) and their sums
    QN_PH_REQ_NEWTON = 6, // Newton: factorise the Hessian at x_k and solve for d and H^-1 d (5 is the in-kernel RUNNING marker)
    QN_PH_REQ_DIR = 9    // sym2, bounded variants: turn the lazy direction into a stored one -- projected onto the solver's box (bfgs_b.rs:72-75)
                         // -- and find the step at which it leaves the line search's box (morethuente_b.rs:185-198): s2_dir_kernel
};

enum QnReqKind : int32_t { QN_REQ_X = 0 /* at x_k itself */, QN_REQ_T = 1 /* at x_k + t d_k */ };

enum QnState : int32_t {
    QN_ST_BEGIN = 0,
    QN_ST_LOOP_TOP,
    QN_ST_AFTER_EVALX,
    QN_ST_CHECK,
    QN_ST_AFTER_DIR,
    QN_ST_LS_BEGIN,
    QN_ST_MT_LOOP,
    QN_ST_MT_AFTER_T,
    QN_ST_MT_AFTER_TL,
    QN_ST_MT_AFTER_TU,
    QN_ST_MT_FINISH,
    QN_ST_BT_LOOP,
    QN_ST_BT_AFTER,
    QN_ST_AFTER_LS,
    QN_ST_AFTER_NEXT,
    QN_ST_AFTER_U,
    QN_ST_ITER_END,
    QN_ST_AFTER_NEWTON,
    QN_ST_LS_ONLY // qn_compute_step_len: g.d for the caller's direction, then the line search alone
};

#define QN_LS_MODIFIED_BIT (1 << 30) // trace: ls_cases bit 30 = the modified-updating switch of morethuente.rs:212-215 was thrown

struct QnTraceRec { // == qn_trace_rec (include/qn_hip.h)
    double f, gnorm, t, s_norm, y_norm;
    int32_t n_evals, ls_iters, ls_cases, updated;
};

struct QnCtl {
    // ---- configuration (written by the host before a run) ----
    double tol;
    int64_t max_iter, max_iter_ls;
    int32_t method, ls_kind, memoize, callback_mode;
    double mt_c1, mt_c2, mt_tmin, mt_tmax, mt_delta;
    double bt_c1, bt_beta;
    int64_t trace_cap;
    int32_t trace_x;
    int32_t small_n; // n <= 5 on one GPU: solver arithmetic in the reference's exact operation order (thread 0)

    // ---- solver state ----
    int32_t phase, state, status, after_state;
    int64_t k;
    int32_t has_s_norm, has_y_norm;
    double s_norm, y_norm;
    double f_k;   // f(x_k)
    double gd0;   // g_k . d_k
    double gnorm; // ||g_k||
    double gg;    // ||g(x_k)||^2 when gg_valid (computed in the sweep that produced g)
    int32_t gg_valid, xtrace_done;
    int32_t have_cur_eval; // g / f_k hold the evaluation at the current x (memoised loop-top call)
    int32_t have_dir;      // d / gd0 already hold the direction for the current x (lazy H+ g+)

    // ---- fused fast path (qn_fused.hip.h): buffer toggles, on-the-fly direction, staged sums of the last evaluation ----
    int32_t fused, xc, sc, dir_mode, gd0_valid;
    int32_t no_defer; // diagnostics: disable the deferred update
    int32_t warm;     // this call continues the previous one on the same objective with nothing touched in between: the memo of the
                      // evaluation at x_k and the lazily formed direction (have_cur_eval, have_dir) carry over (set by the host)
    int32_t defer_u; // the coefficients of the update in flight are not committed yet (QN_PH_REQ_HPASS_EVAL)
    // second-generation symmetric path (qn_sym2.hip.h): the machine runs in the prologue of every kernel
    int32_t sym2;     // this run uses it
    int32_t serviced; // 0: the request in `phase` is pending, 1: its tiles are done and wait for their reduce launch, 2: complete
    int32_t ev_par;   // half of the per-workgroup evaluation scalars the next evaluation writes
    int32_t ev_kind;  // req_kind of the last serviced evaluation ...
    double ev_t;      // ... and its step: what the accept-reduce turns into vectors
    double dir_ug, dir_sg;
    double st_gd0, st_yy, st_ys, st_gg, st_ss, st_dnf;
    double hp_yu, hp_ug, hp_sg;
    double hp_den; // SR1 on the second-generation path: (s - u).y, the update's denominator (sr1_b.rs:143-146), summed by the update-reduce

    // ---- bounded variants (row f4) ----
    int32_t bounded, req_project, last_projected;
    int32_t ls_only; // LineSearch::compute_step_len on its own (line_search/mod.rs:14-23): stop when the line search returns
    double mtb_cand; // min over i of the step to the box along d (morethuente_b.rs:185-198)
    double bt_diff2; // ||P(x + t d) - x||^2 of the last projected trial (backtracking_b.rs:33-34)

    // ---- Newton (newton/mod.rs:8-13) ----
    int32_t has_dec;
    int32_t s2_dir; // sym2, bounded variants (qn_sym2.hip.h, s2_dir_kernel): a new direction goes through QN_PH_REQ_DIR before the line search
                    // starts -- bit 0: it is projected onto the solver's box, bit 1: MoreThuenteB, t_max is clipped by the step to the box
    double dec; // decrement_squared: Option<f64>

    // ---- request ----
    int32_t req_kind, req_need_vectors;
    double req_t;
    int32_t hp_nrhs, hp_lazy;

    // ---- pending symmetric rank-2 update: H_true = H_stored + c_su (sp up' + up sp') + c_ss sp sp' + c_uu up up'
    int32_t pending;
    int32_t spec_tiles; // sym2, folded accept-reduce: the launch that formed the vectors of the accepted point ran the update tiles of the
                        // pass the machine is about to ask for (qn_sym2.hip.h, s2_hpass_kernel); cleared when that request is met
    double c_ss, c_su, c_uu;

    // ---- evaluation results / memo ----
    double f_e, gd_e;            // result handed to the state machine
    int32_t last_valid, d_finite; // (xt, gt, f_last, gd_last) hold the evaluation at x + last_t d
    double last_t, f_last, gd_last;

    // ---- line-search state ----
    int64_t ls_i;
    double t, tl, tu, ls_result;
    int32_t use_mod, conv;
    double phi_t_f, phi_t_g, psi_t_f, psi_t_g;
    double sel_f_tl, sel_g_tl, sel_f_t, sel_g_t; // the (f_tl, g_tl, f_t, g_t) tuple of morethuente.rs:221-226
    double ys;

    // ---- per-iteration trace scratch ----
    double tr_f, tr_gnorm;
    int32_t tr_n_evals, tr_ls_iters, tr_ls_cases, tr_ndigits, tr_updated;
    int32_t dir_ready; // s2_dir: the direction of the current x has been through its QN_PH_REQ_DIR (VV holds -d, mtb_cand its step to the box)

    // ---- counters ----
    uint64_t n_oracle_calls, n_oracle_evals, n_hpasses, n_hpass_rw, n_iterations;
};
