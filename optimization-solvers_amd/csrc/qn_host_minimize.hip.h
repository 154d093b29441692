// qn_host_minimize.hip.h -- host side, part 7 of 7: minimize_impl (LineSearchSolver::minimize, ls_solver.rs:66-111 -- path selection, the control
// block's configuration, the pipelined and the synchronous pump), qn_minimize, qn_compute_step_len.
#pragma once
static int minimize_impl(qn_solver* s, qn_linesearch* ls, const qn_oracle* o, size_t max_iter_solver, size_t max_iter_line_search,
                         qn_callback_fn callback, void* callback_user, int ls_only, double ls_f0);

extern "C" int qn_minimize(qn_solver* s, qn_linesearch* ls, const qn_oracle* o, size_t max_iter_solver,
                           size_t max_iter_line_search, qn_callback_fn callback, void* callback_user) {
    return minimize_impl(s, ls, o, max_iter_solver, max_iter_line_search, callback, callback_user, 0, 0.0);
}

// LineSearch::compute_step_len (line_search/mod.rs:14-23) on its own: the same device state machine entered at the line search
extern "C" int qn_compute_step_len(qn_context* ctx, qn_linesearch* ls, const double* x_k_host, double f_k, const double* g_k_host,
                                   const double* direction_host, size_t n, const qn_oracle* oracle, size_t max_iter, double* step_out) {
    if (!ctx || !ls || !x_k_host || !g_k_host || !direction_host || !oracle || !step_out || n == 0)
        return fail(QN_ERROR_INPUT_PARAMS, "null argument");
    qn_solver* s = nullptr;
    QNCHK(qn_solver_create(ctx, QN_GRADIENT_DESCENT, 0.0, x_k_host, n, &s)); // owns x and the work vectors; no inverse Hessian
    int st = QN_OK;
    hipError_t e = hipMemcpyAsync(s->V.g, g_k_host, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->V.d, direction_host, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) st = fail(QN_ABNORMAL_TERMINATION, std::string("compute_step_len upload: ") + hipGetErrorString(e));
    if (st == QN_OK) st = minimize_impl(s, ls, oracle, 1, max_iter, nullptr, nullptr, 1, f_k);
    if (st == QN_OK) *step_out = s->hctl->ls_result;
    qn_solver_destroy(s);
    return st;
}

static int minimize_impl(qn_solver* s, qn_linesearch* ls, const qn_oracle* o, size_t max_iter_solver, size_t max_iter_line_search,
                         qn_callback_fn callback, void* callback_user, int ls_only, double ls_f0) {
    if (!s || !ls || !o) return fail(QN_ERROR_INPUT_PARAMS, "null argument");
    qn_context* c = s->ctx;
    HIPCHK(hipSetDevice(c->device));
    Run r{s, o, nullptr, QN_ORACLE_GENERIC, false};
    const uint64_t xv0 = c->n_xchg_vector, xs0 = c->n_xchg_scalar; // (collectives of this call, for qn_stats)
    if (o->kind == QN_ORACLE_OBJECTIVE) {
        if (!o->objective) return fail(QN_ERROR_INPUT_PARAMS, "objective is null");
        if (o->objective->ctx != c || o->objective->n != s->n) return fail(QN_ERROR_INPUT_PARAMS, "objective does not match the solver");
        r.obj = o->objective;
        if (r.obj->kind == OBJ_QUADRATIC) { r.oracle_tpl = QN_ORACLE_QUAD; s->V.b = r.obj->b; }
        else if (r.obj->kind == OBJ_LOGSUMEXP) r.oracle_tpl = QN_ORACLE_GENERIC; // evaluated by its own kernels into (f_dev, gt)
        else return fail(QN_ERROR_INPUT_PARAMS, "unsupported objective");
    } else if (o->kind == QN_ORACLE_HOST) {
        if (!o->host_fn) return fail(QN_ERROR_INPUT_PARAMS, "host oracle is null");
    } else if (o->kind == QN_ORACLE_DEVICE_FN) {
        if (!o->device_fn) return fail(QN_ERROR_INPUT_PARAMS, "device oracle is null");
    } else return fail(QN_ERROR_INPUT_PARAMS, "unknown oracle kind");
    if (ls->kind < QN_LS_MORETHUENTE || ls->kind > QN_LS_BACKTRACKING_B) return fail(QN_ERROR_INPUT_PARAMS, "unknown line search");
    const bool ls_bounded = ls->kind == QN_LS_MORETHUENTE_B || ls->kind == QN_LS_BACKTRACKING_B;
    if (ls_bounded || s->bounded) {
        QNCHK(bounds_alloc(s));
        if (ls_bounded) {
            QNCHK(bounds_upload(s, s->bounds_block + 2 * (size_t)s->T.n_pad, ls->lower_bound_host, -INFINITY));
            QNCHK(bounds_upload(s, s->bounds_block + 3 * (size_t)s->T.n_pad, ls->upper_bound_host, INFINITY));
        }
    }
    { // what the stored direction of a bounded second-generation run was formed FOR: the line search's kind and box.  A warm call keeps the direction
      // (and the clip of t_max, QnCtl.mtb_cand) only when both are what they were (the reference recomputes the clip in every
      // compute_step_len, morethuente_b.rs:185-201 -- a call with another box must go through QN_PH_REQ_DIR again).
        std::vector<double> box;
        if (ls_bounded) {
            box.assign(2 * s->n, 0.0);
            for (size_t i = 0; i < s->n; ++i) { box[i] = ls->lower_bound_host ? ls->lower_bound_host[i] : -INFINITY; box[s->n + i] = ls->upper_bound_host ? ls->upper_bound_host[i] : INFINITY; }
        }
        s->ls_box_changed = ls->kind != s->last_ls_kind || box.size() != s->last_ls_box.size() ||
                            (!box.empty() && memcmp(box.data(), s->last_ls_box.data(), box.size() * sizeof(double)) != 0);
        s->last_ls_kind = ls->kind;
        s->last_ls_box.swap(box);
    }
    if (s->method == QN_NEWTON) {
        if (c->world > 1) return fail(QN_ERROR_INPUT_PARAMS, "Newton is single-GPU (SURVEY.md 8(f) row f2)");
        if (!(r.obj && r.obj->kind == OBJ_QUADRATIC) && !(o->kind == QN_ORACLE_HOST && o->host_hessian_fn))
            return fail(QN_ERROR_INPUT_PARAMS, "Hessian not available in the oracle"); // newton/mod.rs:34 .expect(...)
        QNCHK(newton_alloc(s));
    }

    // configuration -> control block (state carried over from earlier runs: x, H, pending update, s_norm, y_norm)
    QnCtl* h = s->hctl;
    h->tol = s->tol;
    h->max_iter = (int64_t)std::min<size_t>(max_iter_solver, (size_t)1 << 62);
    h->max_iter_ls = (int64_t)std::min<size_t>(max_iter_line_search, (size_t)1 << 62);
    h->method = s->method;
    h->ls_kind = ls->kind;
    h->memoize = o->memoize ? 1 : 0;
    h->callback_mode = callback ? 1 : 0;
    h->mt_c1 = ls->c1; h->mt_c2 = ls->c2; h->mt_tmin = ls->t_min; h->mt_tmax = ls->t_max; h->mt_delta = ls->delta;
    h->bt_c1 = ls->bt_c1; h->bt_beta = ls->bt_beta;
    h->trace_cap = (int64_t)s->trace_cap;
    h->trace_x = s->trace_x;
    h->bounded = s->bounded;
    h->req_project = 0; h->last_projected = 0; s->mtb_cand_keep = h->mtb_cand; h->mtb_cand = INFINITY;
    h->ls_only = ls_only;
    if (ls_only) { h->f_k = ls_f0; h->have_cur_eval = 0; h->have_dir = 0; h->last_valid = 0; }
    h->small_n = (s->n <= QN_SMALL_N && c->world == 1) ? 1 : 0;
    if (h->small_n && h->pending) QNCHK(flush_pending(s));
    // fused fast path: device quadratic, memoised, BFGS/DFP, no callback, one column split, n > 5
    // Bounded variants (row f4): BFGSB / DFPB and MoreThuenteB run on the second-generation symmetric path when everything that path needs
    // holds (one rank, whole 128-blocks without padding, a symmetric Q, bitwise symmetric H) -- s2_dir_kernel, qn_sym2.hip.h; BackTrackingB's
    // projected trial points by s2_proj_kernel (round 6).  Everything else bounded keeps the generic path.  QN_S2_BND=0 switches it off (tests: generic path).
    const bool s2b = (s->bounded || ls_bounded) && !ls_only && c->world == 1 && (s->T.n_pad % QN_TB) == 0 &&
                     s->T.n_pad >= 8 * QN_TB && (size_t)s->T.n_pad == s->n && !s->no_sym && !s->no_sym2 && !s->h_nonsym && r.obj && r.obj->q_symmetric &&
                     !s->no_s2bnd && !(getenv("QN_S2_BND") && atoi(getenv("QN_S2_BND")) == 0);
    // SR1 (sr1_b.rs; row f4) has its three-term update only in the second-generation kernels (s2_hpass_kernel<.., SR1>): it takes the fused path
    // exactly when that path will be taken -- the same structural conditions as the bounded variants'.
    const bool s2_struct = !ls_only && c->world == 1 && (s->T.n_pad % QN_TB) == 0 && s->T.n_pad >= 8 * QN_TB && (size_t)s->T.n_pad == s->n && !s->no_sym &&
                           !s->no_sym2 && !s->h_nonsym && r.obj && r.obj->q_symmetric && !s->no_s2bnd && !(getenv("QN_S2_BND") && atoi(getenv("QN_S2_BND")) == 0);
    const bool sr1_s2 = s->method == QN_SR1 && s2_struct;
    r.fused = r.oracle_tpl == QN_ORACLE_QUAD && h->memoize && (s->method == QN_BFGS || s->method == QN_DFP || sr1_s2) && !callback && s->hcs == 1 &&
              s->qcs == 1 && !h->small_n && !s->no_fused && (!(s->bounded || ls_bounded) || s2b);
    // ... and the log-sum-exp objective in the structure of the second-generation path (qn_sym2g.hip.h; round 5): one rank, its one-pass
    // evaluation (n <= 16384), whole 128-blocks without padding, a bitwise symmetric H.  Everything else keeps the generic path.
    r.gobj = r.obj && r.obj->kind == OBJ_LOGSUMEXP && r.obj->lse_kch && !r.obj->lse_two_pass && h->memoize && (s->method == QN_BFGS || s->method == QN_DFP) &&
             !callback && s->hcs == 1 && !h->small_n && !s->no_fused && !s->bounded && !ls_bounded && !ls_only &&
             (c->world == 1 ? (s->T.n_pad % QN_TB) == 0 : ((s->T.rpr % QN_TB) == 0 && c->world <= 64 && (c->comm || c->host_xchg || c->host_async))) &&
             s->T.n_pad >= 8 * QN_TB && (size_t)s->T.n_pad == s->n && !s->no_sym && !s->no_sym2 && !s->h_nonsym &&
             !(getenv("QN_S2G") && atoi(getenv("QN_S2G")) == 0);
    if (r.gobj) r.fused = true;
    h->fused = r.fused ? 1 : 0;
    s->V.fused_hint = h->fused;
    // ... and on the upper block triangle only (half the bytes) when H and Q are whole 128-tiles on one rank
    const bool sym_ok = c->world == 1 && (s->T.n_pad % QN_TB) == 0 && s->T.n_pad >= 8 * QN_TB && !s->no_sym && !s->h_nonsym;
    // ... row-sharded: every rank streams the circulant half of its own block-rows (whole 128-row blocks per rank)
    const bool symsh_ok = c->world > 1 && (s->T.rpr % QN_TB) == 0 && s->T.n_pad >= 8 * QN_TB && !s->no_sym && !s->h_nonsym;
    r.sym = r.fused && (sym_ok || symsh_ok) && r.obj && (r.obj->q_symmetric || r.gobj);
    // the generic path's H pass alone (closures, log-sum-exp objective, SR1, bounded variants): same tiles, sums into V.hp
    r.sym_generic = !r.fused && (sym_ok || symsh_ok) && s->H && s->hcs == 1 && (s->method == QN_BFGS || s->method == QN_DFP || s->method == QN_SR1);
    if ((r.sym || r.sym_generic) && c->world > 1) QNCHK(solver_alloc_symsh_lists(s));
    if (r.sym_generic) {
        if (c->world > 1 && !s->symsh_xg) QNCHK(dev_alloc_zero(&s->symsh_xg, (size_t)c->world * 2 * s->T.n_pad, c->stream));
        const int nb = s->T.n_pad / QN_TB;
        if (s->sym_nb != nb) {
            if (s->sym_part) { HIPCHK(hipFree(s->sym_part)); s->sym_part = nullptr; }
            QNCHK(dev_alloc_zero(&s->sym_part, (size_t)nb * nb * 2 * QN_TB, c->stream));
            s->sym_nb = nb;
        }
    }
    // (the second-generation kernels keep no padding entries at zero; row-sharded: the SHARD instantiations, qn_sym2sh.hip.h)
    r.sym2 = r.sym && !s->no_sym2 && (size_t)s->T.n_pad == s->n && (c->world == 1 || c->world <= 64);
    h->sym2 = r.sym2 ? 1 : 0;
    r.bnd = r.sym2 && (s->bounded || ls_bounded || s->method == QN_SR1); // (SR1: the BND prologues also carry its third update-reduce column)
    if ((s->bounded || ls_bounded) && r.fused && !r.bnd) return fail(QN_ABNORMAL_TERMINATION, "bounded run on a fused path that is not the second-generation one");
    if (s->method == QN_SR1 && r.fused && !r.sym2) return fail(QN_ABNORMAL_TERMINATION, "SR1 on a fused path that is not the second-generation one");
    h->s2_dir = r.bnd ? ((s->bounded ? 1 : 0) | (ls->kind == QN_LS_MORETHUENTE_B ? 2 : 0)) : 0;
    r.dirq = h->s2_dir != 0; // (the stored-direction launch is part of the pattern only where a direction asks for it)
    r.btb = r.bnd && ls->kind == QN_LS_BACKTRACKING_B;
    r.proj = r.btb; // (unless the evaluation kernel projects itself: QnS2Args.projfold, below)
    if (r.bnd && ls->kind == QN_LS_MORETHUENTE_B) h->ls_kind = QN_LS_MORETHUENTE; // (the clip of t_max is applied where the direction's request is consumed: from there on it IS More-Thuente)
    // WHICH KERNEL STREAMS THE UPDATE PASS OF A ROW-SHARDED RUN (round 5).  The one-workgroup-per-CU kernel of
    // qn_sym2.hip.h (16-row register windows, the machine in its prologue) was built for n = 4096, where a launch is a twelfth of the
    // iteration.  On one rank of the P = 8, n = 32768 partition -- 4112 tiles, 16 per workgroup, 1074 MB read and written back -- it takes
    // 222 us (4.85 TB/s); the first-generation tile kernel (one workgroup per tile, two per CU, 8-row windows: nothing to balance, no
    // prologue, no per-workgroup ramp) streams the same tiles in 166 us = 6.47 TB/s = 0.81 of the roofline, and the one-workgroup
    // launch that then runs the machine in front of it costs 7.6 us: 459.6 -> 413.4 us of kernels per iteration on that rank
    // (profiles/r05_l_*; one rank replayed alone with the other ranks' recorded data, the replay reproducing the recorded bits).
    // On ONE GPU the choice does not matter past the cache -- the lists are long, the fixed parts amortised: n = 16384 404 -> 382 us per
    // pass but 1249 -> 1241 it/s with the extra launch, n = 32768 1.575 -> 1.562 ms, 320 -> 324 it/s (profiles/r05_m_*) -- and inside the
    // cache the second-generation kernel is the faster one.  So: a row-sharded run whose share of H's half is beyond 320 MB (the size
    // from which both matrices are streamed non-temporally anyway) -> first-generation tiles; everything else -> the second-
    // generation kernel.  QN_S2_GEN1_TILES=0 / 1 overrides (any rank count).
    {
        const size_t hhalf = c->world > 1 ? (size_t)s->T.rpr * s->T.n_pad * 8 / 2 : (size_t)s->T.n_pad * s->T.n_pad * 8 / 2;
        r.tiles1 = r.sym2 && !r.gobj && c->world > 1 && hhalf > ((size_t)320 << 20);
        if (r.sym2 && !r.gobj && getenv("QN_S2_GEN1_TILES")) r.tiles1 = atoi(getenv("QN_S2_GEN1_TILES")) != 0;
    }
    h->serviced = 0; h->ev_par = 0; h->ev_kind = QN_REQ_X; h->ev_t = 0.0; h->spec_tiles = 0;
    h->defer_u = 0;
    h->no_defer = s->no_defer;
    if (!r.fused) QNCHK(fused_export(s)); // another path takes over: it works on the canonical buffers
    if (!(r.sym || r.sym_generic) || (s->h_diag_stale && (!r.sym2 || (r.gobj && c->world == 1) || r.tiles1))) QNCHK(ensure_full_h(s)); // ... and on whole rows of H (or whole diagonal tiles: the first-generation tile kernel, which the generic-objective path runs too, reads them whole)
    if (r.fused) {
        QNCHK(solver_alloc_fused(s, r.sym));
        s->V.F.pworld = r.sym ? 1 : c->world; // symmetric storage: every rank forms all the per-block partial sums itself
        if (r.sym && c->world > 1 && !s->symsh_xg) QNCHK(dev_alloc_zero(&s->symsh_xg, (size_t)c->world * 2 * s->T.n_pad, c->stream));
        s->V.F.b = r.gobj ? nullptr : r.obj->b; // (the quadratic's linear term; the log-sum-exp path's kernels do not read it)
        if (!s->fused_live) { // import the canonical state (x, pending s and u) into the fused buffers
            const size_t vb = (size_t)s->T.n_pad * sizeof(double);
            HIPCHK(hipMemcpyAsync(s->V.F.X0, s->V.x, vb, hipMemcpyDeviceToDevice, c->stream));
            if (h->pending) {
                HIPCHK(hipMemcpyAsync(s->V.F.S0, s->V.sp, vb, hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(hipMemcpyAsync(s->V.F.UN, s->V.up, vb, hipMemcpyDeviceToDevice, c->stream));
            }
            h->xc = 0; h->sc = 0;
        } // (else: a fused run left them there; xc / sc in the control block say which halves are current)
        h->warm = (s->fused_live && s->warm_obj != 0 && s->warm_obj == r.obj->serial && h->memoize && h->pending && !ls_only) ? 1 : 0;
        if (!h->warm) { h->dir_mode = 0; h->gd0_valid = 0; h->dir_ready = 0; }
        if (!r.bnd || s->ls_box_changed) h->dir_ready = 0;
        // (a warm bounded call whose direction has been through its request keeps mtb_cand: the machine clips this call's t_max with it)
        if (r.bnd && h->dir_ready) h->mtb_cand = s->mtb_cand_keep;
    } else {
        h->warm = 0;
    }
    h->phase = QN_PH_IDLE;
    h->status = -1;
    if (r.sym2) {
        QNCHK(solver_alloc_sym2(s));
        QnS2Args& a = r.s2;
        a.Q = r.obj->Q; a.H = s->H; a.n = (int)s->n; a.np = s->T.n_pad; a.nb = s->s2_nb; a.G = s->s2_G;
        a.item_ij = s->s2_items; a.maxk = s->s2_maxk; a.inorder = s->s2_inorder; a.F = s->V.F; a.part = s->sym_part;
        a.wgS = s->s2_wgS; a.trows = s->s2_trows; a.ctl2 = s->s2_ctl; a.partE = s->s2_partE;
        a.gw = r.gobj ? s->T.n_pad / 64 : 0;
        a.gmu = r.gobj ? r.obj->mu : 0.0;
        // (allocated only for the runs that use them: the tail reduce's counters, the sharded log-sum-exp path's weights)
        if (r.gobj && c->world > 1 && !s->s2_gws) QNCHK(dev_alloc_zero(&s->s2_gws, 80, c->stream)); // the ranks' weights and S, world <= 64
        a.gws = s->s2_gws;
        if (r.gobj && !s->s2_wgV) QNCHK(dev_alloc_zero(&s->s2_wgV, (size_t)2 * s->s2_trows * QN_S2_ROW, c->stream));
        a.wgV = s->s2_wgV;
        if (r.gobj && a.gw > s->s2_trows) return fail(QN_ABNORMAL_TERMINATION, "sym2 (generic objective): more combine workgroups than table rows");
        // folded accept-reduce (s2_hpass_kernel): every workgroup holds at most three items, so the blocks whose slots it sums fit
        // its LDS staging area -- n <= 4096 with 256 workgroups; larger n keeps the accept-reduce launch (7 us of 250+)
        a.fold = (s->s2_maxk <= 3 && s->s2_nb <= 32 && s->fold) ? 1 : 0;
        a.sl_first = s->s2_sl_first; a.sl_per = s->s2_sl_per;
        a.pair = (a.sl_per != 0 && s->s2_maxk == 2 && s->s2_inorder == 2 * s->s2_G && !s->no_pair) ? 1 : 0;
        // sliver rows read the diagonal tiles sl_first .. nb - 1 whole: a run of another kind since the last sliver-mode update
        // pass (or none yet) may have left their lower sub-blocks behind -- restore them once
        if (a.sl_per && !s->h_sliver_whole) { QNCHK(ensure_full_h(s)); s->h_sliver_whole = true; }
        a.trace = s->V.trace; a.xtrace = s->V.xtrace;
        a.sh_world = c->world; a.sh_rank = c->rank; a.sh_ioff = c->rank * (s->T.rpr / QN_TB);
        a.sh_nsum = c->use_allreduce ? 1 : c->world;
        a.evS = s->s2_evS; a.xg = s->symsh_xg; a.sl_off = s->s2_sl_off; a.sl_idx = s->s2_sl_idx;
        if (c->world > 1 || r.gobj) { a.fold = 0; a.pair = 0; }
        { // the pair instance's evaluation as mover + multiplier waves (qn_sym2r.hip.h): the same bits as round 5's kernel, 14.3 us against 15.3 per launch
          // (profiles/r06_a_*).  QN_S2_RING=0 / QN_OPT_EVAL_MOVER_MULTIPLIER 0: round 5's kernel.  It needs every workgroup's FIRST item off the diagonal.
            a.ring = (a.pair && s->ring && s->s2_nb * (s->s2_nb - 1) / 2 >= s->s2_G) ? 1 : 0;
            a.zig = (a.ring && s->zig) ? 1 : 0;
        }
        // BackTrackingB's projection INSIDE the evaluation kernel (round 6, s2_evalr_kernel<true>): no s2_proj_kernel launch per trial -- the trial point is clamped
        // where it is formed, the shares of ||P(x + t d) - x||^2 leave the launch as column 6 of its table.  The same bits as the launch-per-trial flow.
        a.projfold = (r.btb && a.pair && a.ring && !s->no_projfold) ? 1 : 0;
        if (a.projfold) r.proj = false;
        if (r.bnd) a.fold = 0;
        a.method = s->method;
        if (s->method == QN_SR1) a.fold = 0;
        a.lb = (r.bnd && s->bounded) ? s->V.lb : nullptr; a.ub = (r.bnd && s->bounded) ? s->V.ub : nullptr;
        a.llb = (r.bnd && ls_bounded) ? s->V.llb : nullptr; a.lub = (r.bnd && ls_bounded) ? s->V.lub : nullptr;
        if (r.tiles1) a.fold = 0;
        // tail reduce (s2_hpass_kernel<.., TRED>): the update-reduce in the tail of the update-tile launch, 4 launches per iteration
        // instead of 5 -- one rank, lists short enough for one wave to announce (n <= ~15 k).  BUILT, BIT-IDENTICAL, SLOWER, OFF BY
        // DEFAULT (QN_S2_TRED=1 / QN_OPT_TAIL_REDUCE switch it on; the note in front of the kernel has the stamps): the update kernel
        // 23.5 -> 36.7 us for a 5.0 us launch saved.
        a.cnt = s->s2_cnt;
        a.cnt_stride = getenv("QN_S2_CNT_STRIDE") ? std::max(1, std::min(QN_S2_CNT_STRIDE, atoi(getenv("QN_S2_CNT_STRIDE")))) : QN_S2_CNT_STRIDE; // (diagnostics)
        const bool want_tred = getenv("QN_S2_TRED") ? atoi(getenv("QN_S2_TRED")) != 0 : s->tred;
        if (want_tred && !s->s2_cnt) {
            HIPCHK(hipMalloc((void**)&s->s2_cnt, (size_t)a.nb * QN_S2_CNT_STRIDE * sizeof(int)));
            HIPCHK(hipMemsetAsync(s->s2_cnt, 0, (size_t)a.nb * QN_S2_CNT_STRIDE * sizeof(int), c->stream));
        }
        a.cnt = s->s2_cnt;
        a.tred = (c->world == 1 && !a.fold && !r.gobj && !r.tiles1 && s->s2_maxk <= QN_S2_TRED_MAXK && want_tred && s->method != QN_SR1) ? 1 : 0;
        // TOUCH workgroups (qn_s2_touch): where the two small launches are s2_vec_kernel<false> / s2_hreduce_kernel<false> in front of the pair instance's
        // tile launches, and a workgroup's XCD survives the offset (nb a multiple of 8)
        {
            auto rows_ok = [](int v) { return v == 4 || v == 6 || v == 8 || v == 10 || v == 12 || v == 16; };
            const bool can = a.ring && !r.bnd && !a.fold && !a.tred && s->method != QN_SR1 && a.nb == 32 && a.G == 256; // (the instantiations are nb = 32's)
            a.touch = (can && rows_ok(s->touch)) ? s->touch : 0;
            a.touchq = (can && rows_ok(s->touchq)) ? s->touchq : 0;
            a.touch_delay = std::max(0, std::min(512, s->touch_delay));
            a.touchq_delay = std::max(0, std::min(512, s->touchq_delay));
        }
        // WHO GETS THE INFINITY CACHE (256 MB).  Per iteration a rank streams its half of Q twice (read) and its half of H once
        // (read + written); non-temporal accesses pass the cache by.  Measured (round 4, bench.py same box, it/s for the policies
        // H plain / Q plain, H plain / Q non-temporal, H non-temporal / Q plain, both non-temporal):
        //     n =  4096 (2 x  67 MB): both plain (known since round 1)          n =  5120 (2 x 105 MB): 10 101   9 905   9 473   9 188
        //     n =  6144 (2 x 151 MB):  7 624  *8 020*  7 676   7 421            n =  8192 (2 x 268 MB):  4 262  *4 860*  4 756   4 656
        //     n = 10240 (2 x 419 MB):  2 738   2 962   2 941  *3 101*           n = 12288 (2 x 604 MB):  2 017   2 122   2 166  *2 271*
        // While both halves fit, everything stays plain; when they do not, H is the better tenant (its bytes are touched twice per
        // pass) and Q is streamed past it -- until H's half alone is well over the cache's size, where nothing is worth keeping.
        // tools/stream_shape_probe.hip has the ceilings (a plain read 6.1-6.3 TB/s, non-temporal 6.5-6.85; read + write 5.2 / 5.5-5.6).
        const size_t half = c->world > 1 ? (size_t)s->T.rpr * s->T.n_pad * 8 / 2 : (size_t)s->T.n_pad * s->T.n_pad * 8 / 2;
        if (2 * half <= ((size_t)230 << 20)) { a.nt = 0; a.ntq = 0; }
        else if (half <= ((size_t)320 << 20)) { a.nt = 0; a.ntq = 1; }
        else { a.nt = 1; a.ntq = 1; }
        if (getenv("QN_S2_NT")) a.nt = atoi(getenv("QN_S2_NT"));    // (diagnostics: tools/README.md)
        if (getenv("QN_S2_NTQ")) a.ntq = atoi(getenv("QN_S2_NTQ"));
        // (nothing is uploaded here: the FIRST launch of the call reads the control block from the pinned, device-mapped mirror
        // itself -- QnS2Args.ctl_first.  Round 3 went from hipMemcpyAsync (~8 us in front of the first kernel of every call) to a
        // one-workgroup upload launch (~4 us); now there is neither.  The host does not write the mirror again before the batch's
        // last launch has reported, or s2_peek has synchronised.)
    } else {
        QNCHK(poke_ctl(s));
    }

    // only the quadratic objective's kernels are predicated on the control block; everything else is serviced synchronously
    const bool can_pipeline = (r.oracle_tpl == QN_ORACLE_QUAD || r.gobj) && !callback && !(c->world > 1 && !c->comm && !c->host_async);
    const bool sync = s->method == QN_NEWTON || s->sync_mode == 1 || (s->sync_mode == -1 && !(can_pipeline && o->memoize)) || !can_pipeline;

    int status = QN_ABNORMAL_TERMINATION;
    if (r.sym2 && !r.gobj) QNCHK(place_h(r)); // (once per solver: H where the update kernel runs fastest)
    if (r.sym2) {
        if (sync) { // one request at a time: [service launch(es), advance], the host reads the control block in between
            QNCHK(s2_launch(r, QN_S2_ADVANCE));
            for (;;) {
                QNCHK(s2_peek(r));
                const int ph = h->phase;
                if (ph == QN_PH_DONE) { status = h->status; break; }
                const bool tiles_done = ph == QN_PH_REQ_HPASS && h->serviced == 1; // (folded accept-reduce: the tiles ran with the vectors)
                if (h->serviced != 0 && !tiles_done) return fail(QN_ABNORMAL_TERMINATION, "sym2: request in an unexpected service state");
                if (ph == QN_PH_REQ_EVAL) QNCHK(s2_do_eval(r));
                else if (ph == QN_PH_REQ_VEC) QNCHK(s2_do_vec(r));
                else if (ph == QN_PH_REQ_DIR && r.bnd) QNCHK(s2_launch(r, QN_S2_DIR));
                else if (ph == QN_PH_REQ_HPASS) QNCHK(s2_do_hpass(r, !tiles_done));
                else return fail(QN_ABNORMAL_TERMINATION, "sym2: control block in an unexpected phase");
                QNCHK(s2_launch(r, QN_S2_ADVANCE));
            }
        } else { // pipelined: [eval x slots, (accept-reduce,) update tiles, update-reduce] per period, each launch predicated in its prologue
            // Row-sharded: every evaluation launch is followed by a collective whether the machine uses the slot or not (RCCL cannot
            // be predicated from the device), so the pattern is sized to the line search in use: two slots per period to start
            // with (More-Thuente on a quadratic: t = 1, then one interpolation; backtracking near the solution: t = 1), and from
            // the second batch on what the run has needed so far -- the counters are replicated, every rank sizes alike.  An
            // iteration that needs more evaluations than a period holds rolls over into the next one: only time is lost.
            int slots = (ls->kind == QN_LS_MORETHUENTE) ? 2 : 4;
            // (generic objective: an unused evaluation slot is three launches that find nothing to do, and on such objectives More-Thuente
            // takes t = 1 almost every time -- the pattern is sized like the sharded one: one slot to start with, then what the run has needed)
            const bool adaptive = r.s2.sh_world > 1 || r.gobj || r.bnd; // (bounded: MoreThuenteB's clipped first step is often the accepted one)
            if (adaptive) slots = s->s2_slots_hint ? s->s2_slots_hint : (r.gobj ? 1 : 2);
            bool first = true;
            uint64_t ev0 = 0, it0 = 0;
            unsigned long long seq = 0;
            for (;;) {
                if (!first) {
                    QNCHK(s2_wait_report(r, seq));
                    if (adaptive && h->n_iterations > it0) { // evaluations per iteration of the batch just run, rounded up
                        const uint64_t di = h->n_iterations - it0;
                        uint64_t de = h->n_oracle_evals - ev0;
                        if (it0 == 0 && !h->warm && de > 0) de -= 1; // (the evaluation at x0 that opens a run had a period of its own)
                        slots = (int)std::min<uint64_t>(4, std::max<uint64_t>(1, (de + di - 1) / di)); // (measured, round 6: up to 8 slots for BackTrackingB's ~8 evaluations per iteration -- 283 us against 250: an unused slot costs more than a short period's four unused launches)
                        s->s2_slots_hint = slots; // (the next call starts from it)
                    }
                    ev0 = h->n_oracle_evals; it0 = h->n_iterations;
                    if (h->phase == QN_PH_DONE) { status = h->status; break; }
                }
                int64_t remaining = h->max_iter - (first ? 0 : h->k);
                if (remaining < 1) remaining = 1;
                // one period per iteration, one more for a run that has no direction yet (evaluation at x, direction pass), and
                // a last evaluation launch whose prologue finds the iteration cap reached and writes DONE
                const int64_t periods = std::min<int64_t>(remaining + ((first && !h->warm) ? 1 : 0), 256);
                first = false;
                auto one_period = [&]() -> int {
                    if (r.dirq) QNCHK(s2_launch(r, QN_S2_DIR)); // (the direction the period's evaluations search along: stored, projected)
                    for (int e = 0; e < slots; ++e) QNCHK(s2_do_eval(r));
                    // (BackTrackingB: the accept-reduce, the update tiles and the update-reduce run the UNBOUNDED machine in their prologues -- they are the
                    // benchmark's kernels, unchanged -- and must not be the ones that consume a projected trial: its Armijo rule and its memo are the
                    // bounded machine's.  Every evaluation slot but the last is followed by the next slot's s2_proj_kernel, whose prologue is the bounded
                    // one; behind the last a one-workgroup machine launch consumes.)
                    if (r.btb) QNCHK(s2_launch(r, QN_S2_ADVANCE));
                    if (!r.s2.fold && !(r.gobj && r.s2.sh_world == 1)) QNCHK(s2_do_vec(r)); // (folded into the update tiles otherwise; generic objective: staged by every evaluation's combine launch)
                    QNCHK(s2_do_hpass(r, true));
                    return QN_OK;
                };
                int64_t p = 0;
                // MEASUREMENT (QN_S2_GRAPH=1; tools/README.md): the periods behind the first one as launches of ONE captured hipGraph of two
                // periods (an even number of launches, so the control block's parity repeats; the first period carries ctl_first, the
                // reporting launch stays outside).  Single rank, quadratic objective, profiling off.
                static const bool want_graph = getenv("QN_S2_GRAPH") && atoi(getenv("QN_S2_GRAPH")) != 0;
                if (want_graph && c->world == 1 && !r.gobj && !s->profiling && periods >= 3) {
                    QNCHK(one_period()); ++p;
                    const uint64_t l0 = r.s2_launches;
                    if ((l0 & 1) != 0) { QNCHK(one_period()); ++p; } // (start the captured pair on an even launch count)
                    if (periods - p >= 2) {
                        const uint64_t lbase = r.s2_launches, stat0 = s->stats.launches;
                        const bool reuse = s->s2_graph_exec && memcmp(&s->s2_graph_args, &r.s2, sizeof(QnS2Args)) == 0 && s->s2_graph_slots == slots && s->s2_graph_bnd == (int)r.bnd;
                        if (!reuse) {
                            if (s->s2_graph_exec) { (void)hipGraphExecDestroy(s->s2_graph_exec); s->s2_graph_exec = nullptr; }
                            hipGraph_t g = nullptr;
                            HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                            int rc = one_period(); if (rc == QN_OK) rc = one_period();
                            const hipError_t e = hipStreamEndCapture(c->stream, &g);
                            if (rc != QN_OK) return rc;
                            HIPCHK(e);
                            HIPCHK(hipGraphInstantiate(&s->s2_graph_exec, g, nullptr, nullptr, 0));
                            (void)hipGraphDestroy(g);
                            s->s2_graph_args = r.s2; s->s2_graph_slots = slots; s->s2_graph_bnd = (int)r.bnd;
                            s->s2_graph_len = r.s2_launches - lbase; s->s2_graph_stat = s->stats.launches - stat0;
                            r.s2_launches = lbase; s->stats.launches = stat0; // (captured, not run)
                        }
                        if ((s->s2_graph_len & 1) == 0) {
                            for (; periods - p >= 2; p += 2) {
                                HIPCHK(hipGraphLaunch(s->s2_graph_exec, c->stream));
                                r.s2_launches += s->s2_graph_len; s->stats.launches += s->s2_graph_stat;
                            }
                        }
                    }
                }
                for (; p < periods; ++p) QNCHK(one_period());
                seq = ++s->rep_seq; // (the batch's last launch reports)
                // One rank, quadratic objective: the reporting launch is the ONE-WORKGROUP machine launch, not an evaluation launch.  The
                // evaluation kernel requests its first item before it knows whether there is anything to evaluate -- at the end of a call
                // there is not: the machine finds the iteration cap and writes DONE -- which made the last launch of every call 10.7 us
                // (kernel trace of the driver's 20-step call); the machine launch is 3-4.  Whatever request the machine leaves pending is
                // served by the next batch's first launches.
                if (r.s2.sh_world == 1 && !r.gobj) { r.report_seq = seq; QNCHK(s2_launch(r, QN_S2_ADVANCE)); }
                else QNCHK(s2_do_eval(r, seq));
            }
        }
    } else {
    QNCHK(launch_ctl(r, QN_PH_IDLE));
    if (sync) {
        for (;;) {
            QNCHK(peek_ctl(s));
            const int ph = h->phase;
            if (ph == QN_PH_DONE) { status = h->status; break; }
            if (ph == QN_PH_REQ_EVAL) { QNCHK(enqueue_eval(r)); QNCHK(launch_ctl(r, QN_PH_REQ_EVAL)); }
            else if (ph == QN_PH_REQ_HPASS) { QNCHK(enqueue_hpass_req(r)); QNCHK(launch_ctl(r, QN_PH_REQ_HPASS)); }
            else if (ph == QN_PH_REQ_HPASS_EVAL) { // fused path: update pass, then the evaluation that derives the update's coefficients itself
                QNCHK(enqueue_hpass_req(r)); QNCHK(enqueue_eval(r, 1)); QNCHK(launch_ctl(r, QN_PH_REQ_HPASS_EVAL));
            }
            else if (ph == QN_PH_REQ_NEWTON) { { ProfScope ps(s, KC_NEWTON); QNCHK(enqueue_newton(s, o, r.obj)); } QNCHK(launch_ctl(r, QN_PH_REQ_NEWTON)); }
            else if (ph == QN_PH_ITER_DONE) { callback(callback_user, s); QNCHK(launch_ctl(r, QN_PH_ITER_DONE)); }
            else return fail(QN_ABNORMAL_TERMINATION, "control block in an unexpected phase");
        }
    } else {
        // pipelined: every kernel is predicated on the control block, so a fixed pattern can be enqueued ahead
        // of the decisions; one period = [eval, step] x slots, [h_pass, step], and advances at most one iteration.
        const int slots_max = (ls->kind == QN_LS_MORETHUENTE || ls->kind == QN_LS_MORETHUENTE_B) ? 2 : 4;
        int slots = slots_max;
        // The generic path (closures excluded: they are synchronous) sizes its periods like the sharded second-generation path does:
        // an evaluation slot the line search does not use is two launches that find nothing to do (5.2 + 4.8 us at n = 4096 -- a tenth of a
        // bounded iteration, where MoreThuenteB accepts t = 1: profiles/r05_q_*); from the second batch on a period carries what the run
        // has needed per iteration so far, rounded up, and an iteration that needs more rolls over into the next period (every launch
        // is predicated: only time is lost).  The counters are the control block's: the same on every rank.
        const bool adaptive = !r.fused;
        if (adaptive && s->gen_slots_hint) slots = std::min(slots_max, s->gen_slots_hint);
        uint64_t ev0 = 0, it0 = 0;
        bool first_batch = true;
        const int gd = s->method == QN_GRADIENT_DESCENT;
        for (;;) {
            QNCHK(peek_ctl(s));
            if (adaptive && !first_batch && h->n_iterations > it0) {
                const uint64_t di = h->n_iterations - it0;
                uint64_t de = h->n_oracle_evals - ev0;
                if (it0 == 0 && de > 0) de -= 1; // (the evaluation at x0 that opens a run)
                slots = (int)std::min<uint64_t>((uint64_t)slots_max, std::max<uint64_t>(1, (de + di - 1) / di));
                s->gen_slots_hint = slots;
            }
            ev0 = h->n_oracle_evals; it0 = h->n_iterations;
            first_batch = false;
            if (h->phase == QN_PH_DONE) { status = h->status; break; }
            int64_t remaining = h->max_iter - h->k;
            if (remaining < 1) remaining = 1;
            const int64_t periods = std::min<int64_t>(remaining, 256);
            const int first_mask = (1 << QN_PH_REQ_EVAL) | (1 << QN_PH_REQ_HPASS) | (1 << QN_PH_REQ_HPASS_EVAL);
            for (int64_t p = 0; p < periods; ++p) {
                if (r.fused) {
                    // [eval*, step*] [eval, step] x (slots-1) [h_pass]: the first evaluation of a period directly follows the
                    // previous period's h_pass and may service QN_PH_REQ_HPASS_EVAL; its step also consumes a plain h_pass
                    QNCHK(enqueue_eval(r, 1)); QNCHK(launch_ctl_mask(r, first_mask));
                    for (int e = 1; e < slots; ++e) { QNCHK(enqueue_eval(r, 0)); QNCHK(launch_ctl(r, QN_PH_REQ_EVAL)); }
                    QNCHK(enqueue_hpass_req(r));
                } else {
                    for (int e = 0; e < slots; ++e) { QNCHK(enqueue_eval(r)); QNCHK(launch_ctl(r, QN_PH_REQ_EVAL)); }
                    if (!gd) { QNCHK(enqueue_hpass_req(r)); QNCHK(launch_ctl(r, QN_PH_REQ_HPASS)); }
                }
            }
        }
    }
    } // (!r.sym2)
    // Nothing is copied back here: the iterate and the pending vectors stay in the fused buffers (fused_export), the lower
    // triangle of H stays stale (ensure_full_h) until a getter, a setter or a run on another path asks for them.  A solve made
    // of several qn_minimize calls (warm restarts, a harness timing short calls) pays for neither.
    if (r.fused) s->fused_live = true;
    s->warm_obj = (r.fused && status == QN_MAX_ITER_REACHED && h->memoize && h->have_cur_eval && h->have_dir) ? r.obj->serial : 0;
    if (ls->kind == QN_LS_MORETHUENTE_B) ls->t_max = h->mt_tmax; // morethuente_b.rs:201: the clipped t_max stays in the line search
    s->stats.iterations = h->n_iterations;
    s->stats.oracle_calls = h->n_oracle_calls;
    s->stats.oracle_evals = h->n_oracle_evals;
    s->stats.h_passes = h->n_hpasses;
    uint64_t shard = (uint64_t)s->T.rpr * (uint64_t)s->T.n_pad * 8ull;
    const uint64_t full_shard = shard;
    if (r.sym || r.sym_generic) shard = (uint64_t)s->sym_nb * (uint64_t)(s->sym_nb + 1) / 2ull * (uint64_t)QN_TB * QN_TB * 8ull; // the streamed tiles
    if ((r.sym || r.sym_generic) && c->world > 1) shard = (uint64_t)qn_symsh_ntiles(s->sym_nb, s->T.rpr / QN_TB, c->rank * (s->T.rpr / QN_TB)) * (uint64_t)QN_TB * QN_TB * 8ull;
    if (r.sym2 && !r.gobj && !r.tiles1) // diagonal tiles: wave w (rows 16 w ...) reads 64 - 8 w lanes of 16 bytes per row = 73 728 of the 131 072 bytes
        shard = (uint64_t)s->sym_nb * (uint64_t)(s->sym_nb - 1) / 2ull * (uint64_t)QN_TB * QN_TB * 8ull + (uint64_t)s->sym_nb * 73728ull;
    if (r.sym2 && c->world > 1 && !r.tiles1) { // this rank's windows: one diagonal tile per local block-row, the rest whole tiles
        const uint64_t nbl = (uint64_t)(s->T.rpr / QN_TB);
        const uint64_t nt = (uint64_t)qn_symsh_ntiles(s->sym_nb, (int)nbl, c->rank * (int)nbl);
        shard = (nt - nbl) * (uint64_t)QN_TB * QN_TB * 8ull + nbl * 73728ull;
    }
    s->stats.h_bytes = (h->n_hpasses + h->n_hpass_rw) * shard;
    if (r.sym2) s->stats.h_bytes = 2 * h->n_hpasses * shard; // (its one branch-free body writes every pass back, pending update or not)
    s->stats.obj_bytes = (r.oracle_tpl == QN_ORACLE_QUAD) ? h->n_oracle_evals * (r.sym ? shard : full_shard) : 0;
    if (r.obj && r.obj->kind == OBJ_LOGSUMEXP) // one pass over this rank's rows of A per evaluation (two for n > 16384)
        s->stats.obj_bytes = h->n_oracle_evals * (uint64_t)r.obj->TA.rpr * (uint64_t)r.obj->T.n_pad * 8ull * ((r.obj->lse_kch && !r.obj->lse_two_pass) ? 1ull : 2ull);
    s->stats.matrix_bytes_per_pass = shard;
    s->stats.total_minimize_calls++;
    s->stats.total_iterations += s->stats.iterations;
    s->stats.total_oracle_calls += s->stats.oracle_calls;
    s->stats.total_oracle_evals += s->stats.oracle_evals;
    s->stats.total_h_passes += s->stats.h_passes;
    s->stats.total_h_bytes += s->stats.h_bytes;
    s->stats.total_obj_bytes += s->stats.obj_bytes;
    s->stats.total_xchg_vector += c->n_xchg_vector - xv0;
    s->stats.total_xchg_scalar += c->n_xchg_scalar - xs0;
    s->stats.path = (r.fused ? QN_PATH_FUSED : 0u) | (r.sym ? QN_PATH_SYM : 0u) | (r.sym_generic ? QN_PATH_SYM_GENERIC : 0u) |
                    (sync ? 0u : QN_PATH_PIPELINED) | (r.sym2 ? QN_PATH_SYM2 : 0u) | ((r.tiles1 || (r.gobj && c->world == 1)) ? QN_PATH_TILES1 : 0u);
    if (c->host_async_failed) { c->host_async_failed = 0; return fail(QN_ABNORMAL_TERMINATION, "host exchange callback failed"); }
    if (status == QN_ABNORMAL_TERMINATION) return fail(status, "solver state machine aborted");
    return status;
}
