// qn_host_launch.hip.h -- host side, part 5 of 7: one qn_minimize call's `Run`, the launches of every path (second-generation machine, generic
// objectives on it, fused rows, generic control step), the placement probe of H.
#pragma once
struct Run;
static int enqueue_newton(qn_solver* s, const qn_oracle* o, qn_objective* obj);

// ---- the pump ----
struct Run {
    qn_solver* s;
    const qn_oracle* o;
    qn_objective* obj;
    int oracle_tpl; // QN_ORACLE_GENERIC / QN_ORACLE_QUAD
    bool fused;
    bool sym = false; // fused path on the upper block triangle of H and Q (qn_sym.hip.h)
    bool sym_generic = false; // generic path: the H pass alone on the upper block triangle
    bool sym2 = false;        // second-generation symmetric path (qn_sym2.hip.h)
    bool gobj = false;        // ... in its form for a device objective that is not the quadratic (qn_sym2g.hip.h: the log-sum-exp objective)
    bool dirq = false;        // ... whose pattern has the stored-direction launch (QnCtl.s2_dir != 0)
    bool proj = false;        // ... whose line search is BackTrackingB: every evaluation slot has s2_proj_kernel in front of it (projected trial points)
    bool btb = false;         // ... BackTrackingB on this path at all (proj: with the projection as a launch of its own; QnS2Args.projfold: inside the evaluation kernel)
    bool bnd = false;         // ... a bounded run on it (BFGSB / DFPB, MoreThuenteB): one more launch per iteration, s2_dir_kernel (qn_sym2.hip.h)
    bool tiles1 = false;      // the update pass's tiles through the first-generation tile kernel (one workgroup per tile, two per CU) behind a
                              // one-workgroup launch that runs the machine: H's share past the Infinity Cache (see minimize_impl)
    QnS2Args s2{};
    uint64_t s2_launches = 0; // parity of the control-block double buffer = launches so far & 1
    unsigned long long report_seq = 0; // != 0: the next launch reports its control block to the host (s2_wait_report)
};

// ---- generic objectives on the second-generation structure (qn_sym2g.hip.h) ----
static QnS2GArgs s2g_args(const Run& r) {
    qn_solver* s = r.s;
    qn_objective* o = r.obj;
    qn_context* c = s->ctx;
    QnS2GArgs g{};
    g.L.A = o->Q; g.L.c = o->b; g.L.mu = o->mu;
    g.L.m = (int)o->m; g.L.m_pad = o->TA.n_pad; g.L.mrpr = o->TA.rpr; g.L.n = (int)o->n; g.L.n_pad = o->T.n_pad;
    g.L.world = c->world; g.L.rank = c->rank; g.L.rs = 1;
    g.wgms = o->lwgms; g.wgg = o->lwgg; g.G = o->lse_G;
    g.ctl = s->s2_ctl + (r.s2_launches & 1); // what the last prologue launch has written
    g.F = s->V.F;
    g.wgS = nullptr; g.trows = s->s2_trows;
    g.gall = o->lgall; g.ev_slice = nullptr;
    return g;
}
template <int KCH, bool NTA>
static int s2g_launch_onepass_nt(hipStream_t st, const QnS2GArgs& g) {
    static std::atomic<bool> attr_set[64]; // per device: hipFuncSetAttribute applies to the current device only (atomic: ranks may be threads)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t lds = (size_t)KCH * 1024 * sizeof(double);
    if ((dev < 0 || dev >= 64 || !attr_set[dev].load()) && lds > 48 * 1024) { // the trial point in LDS: up to 128 KB of the CU's 160 KB
        if (hipFuncSetAttribute((const void*)s2g_onepass_kernel<KCH, NTA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return fail(QN_ABNORMAL_TERMINATION, "log-sum-exp: the device does not grant the evaluation kernel its LDS");
        }
        if (dev >= 0 && dev < 64) attr_set[dev].store(true);
    }
    hipLaunchKernelGGL((s2g_onepass_kernel<KCH, NTA>), dim3(g.G), dim3(512), lds, st, g);
    return QN_OK;
}
template <int KCH>
static int s2g_launch_onepass(hipStream_t st, const QnS2GArgs& g) {
    static const int nt_env = getenv("QN_LSE_NT") ? atoi(getenv("QN_LSE_NT")) : -1; // (as lse_launch_onepass)
    const bool nt = nt_env >= 0 ? nt_env != 0 : (size_t)g.L.mrpr * (size_t)g.L.n_pad * sizeof(double) > ((size_t)230 << 20);
    return nt ? s2g_launch_onepass_nt<KCH, true>(st, g) : s2g_launch_onepass_nt<KCH, false>(st, g);
}
static int s2g_enqueue_onepass(Run& r) {
    qn_solver* s = r.s;
    hipStream_t st = s->ctx->stream;
    const QnS2GArgs g = s2g_args(r);
    ProfScope ps(s, KC_EVAL);
    switch (r.obj->lse_kch) {
    case 1: QNCHK(s2g_launch_onepass<1>(st, g)); break;
    case 2: QNCHK(s2g_launch_onepass<2>(st, g)); break;
    case 4: QNCHK(s2g_launch_onepass<4>(st, g)); break;
    case 8: QNCHK(s2g_launch_onepass<8>(st, g)); break;
    default: QNCHK(s2g_launch_onepass<16>(st, g)); break;
    }
    s->stats.launches++;
    HIPCHK(hipGetLastError());
    return QN_OK;
}
// the update pass's tiles: the first-generation tile kernel (qn_sym.hip.h: one workgroup per tile, two per CU -- 6.2 TB/s on H's half at
// n = 16384 where the one-workgroup-per-CU kernel of qn_sym2.hip.h reaches 5.4), reading the control block the launch in front wrote
static int s2g_enqueue_tiles(Run& r) {
    qn_solver* s = r.s;
    qn_context* c = s->ctx;
    QnSymHPassArgs y{};
    y.H = s->H; y.T = s->T; y.T.cs = 1; y.F = s->V.F; y.F.UP = y.F.UN; // (no kernel of this path writes u while another reads it)
    y.ctl = s->s2_ctl + (r.s2_launches & 1); y.expect_phase = QN_PH_REQ_HPASS; y.need_serviced = 1;
    y.nb = s->sym_nb; y.part = s->sym_part;
    y.nt = s->T.n_pad >= 8192; // past the Infinity Cache (same-box A/B, round 1: +7 % at n = 32768, +3 % at 8192, -1 % at 4096)
    int grid = y.nb * (y.nb + 1) / 2;
    if (c->world > 1) { y.sh = sym_shard(s); grid = qn_symsh_ntiles(y.nb, y.sh.nbl, y.sh.ioff); } // row-sharded: the rank's circulant windows
    {
        ProfScope ps(s, KC_HPASS);
        hipLaunchKernelGGL(sym_hpass_tile_kernel, dim3(grid), dim3(QN_SYM_TPB), 0, c->stream, y);
    }
    s->h_lower_stale = true;
    s->stats.launches++;
    HIPCHK(hipGetLastError());
    return QN_OK;
}

static int s2_launch(Run& r, int kind) {
    qn_solver* s = r.s;
    hipStream_t st = s->ctx->stream;
    QnS2Args a = r.s2;
    a.parity = (int)(r.s2_launches & 1);
    a.ctl_first = r.s2_launches == 0 ? s->hctl : nullptr; // (the first launch of a call takes the control block from the pinned mirror)
    a.rep = s->hrep; a.rep_flag = s->hrep_flag; a.rep_seq = r.report_seq;
    r.report_seq = 0;
#ifdef QN_S2_STAMPS
    a.dbg = s->V.dbg; a.slot = (int)r.s2_launches;
    a.swz = (getenv("QN_S2_SWZ") && a.pair) ? atoi(getenv("QN_S2_SWZ")) : 0; // (only where both items follow from the workgroup index: n = 4096)
#endif
    r.s2_launches++;
    const int cls = kind == QN_S2_EVAL ? KC_EVAL : (kind == QN_S2_VEC || kind == QN_S2_VECD || kind == QN_S2_VSUM || kind == QN_S2_GCOMB || kind == QN_S2_DIR || kind == QN_S2_PROJ) ? KC_EREDUCE : kind == QN_S2_HTILE ? KC_HPASS
                  : (kind == QN_S2_HREDUCE || kind == QN_S2_HSUM) ? KC_HREDUCE : KC_CTL;
    ProfScope ps(s, cls);
    const bool sh = a.sh_world > 1; // row-sharded: the SHARD instantiations (qn_sym2sh.hip.h)
    switch (kind) {
    case QN_S2_EVAL:
        if (sh) {
            if (a.ntq) hipLaunchKernelGGL((s2_eval_kernel<false, true, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            else hipLaunchKernelGGL((s2_eval_kernel<false, true, false>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
        } else if (r.bnd) { // (bounded variants: the same kernels behind the bounded runs' prologue)
            if (a.pair && a.ring) hipLaunchKernelGGL(s2_evalr_kernel<true>, dim3(a.G), dim3(QN_S2R_TPB), 0, st, a);
            else if (a.pair) hipLaunchKernelGGL((s2_eval_kernel<true, false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            else if (a.ntq) hipLaunchKernelGGL((s2_eval_kernel<false, false, true, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            else hipLaunchKernelGGL((s2_eval_kernel<false, false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
        } else if (a.pair && a.ring) hipLaunchKernelGGL(s2_evalr_kernel<false>, dim3(a.G), dim3(QN_S2R_TPB), 0, st, a);
        else if (a.pair) hipLaunchKernelGGL(s2_eval_kernel<true>, dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
        else if (a.ntq) hipLaunchKernelGGL((s2_eval_kernel<false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
        else hipLaunchKernelGGL(s2_eval_kernel<false>, dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
        break;
    case QN_S2_DIR: hipLaunchKernelGGL(s2_dir_kernel, dim3(a.nb), dim3(QN_TB), 0, st, a); break;
    case QN_S2_PROJ: hipLaunchKernelGGL(s2_proj_kernel, dim3(a.nb), dim3(QN_TB), 0, st, a); break;
    case QN_S2_VSUM:
        if (r.gobj) hipLaunchKernelGGL((s2_advance_kernel<true, true, QN_S2_VSUM>), dim3(1), dim3(128), 0, st, a); // (the machine sees the accepted point; the gather follows)
        else hipLaunchKernelGGL(s2sh_vsum_kernel, dim3(a.nb), dim3(QN_S2_TPB), 0, st, a);
        break;
    case QN_S2_HSUM: hipLaunchKernelGGL(s2sh_hsum_kernel, dim3(a.nb), dim3(QN_S2_TPB), 0, st, a); break;
    case QN_S2_VEC:
        if (r.gobj) { const QnS2GArgs g = s2g_args(r); hipLaunchKernelGGL(s2g_vec_kernel, dim3(a.nb), dim3(QN_TB), 0, st, a, g); }
        else if (sh) hipLaunchKernelGGL(s2_vec_kernel<true>, dim3(a.nb), dim3(QN_S2_TPB), 0, st, a);
        else if (a.touch) hipLaunchKernelGGL((s2_vec_kernel<false, false, 32>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a); // (with the TOUCH workgroups, qn_s2_touch: G in all -- one workgroup per CU)
        else hipLaunchKernelGGL(s2_vec_kernel<false>, dim3(a.nb), dim3(QN_S2_TPB), 0, st, a);
        break;
    case QN_S2_VECD: hipLaunchKernelGGL((s2_vec_kernel<true, true>), dim3(a.nb), dim3(QN_S2_TPB), 0, st, a); break; // (row-sharded, trial-vector exchange)
    case QN_S2_HTILE:
        if (s->method == QN_SR1) { // (one rank, no fold, no tail reduce: minimize_impl)
            if (a.nt) hipLaunchKernelGGL((s2_hpass_kernel<true, false, false, false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            else hipLaunchKernelGGL((s2_hpass_kernel<false, false, false, false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
        } else if (sh) {
            if (s->method == QN_BFGS) {
                if (a.nt) hipLaunchKernelGGL((s2_hpass_kernel<true, true, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
                else hipLaunchKernelGGL((s2_hpass_kernel<false, true, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            } else {
                if (a.nt) hipLaunchKernelGGL((s2_hpass_kernel<true, false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
                else hipLaunchKernelGGL((s2_hpass_kernel<false, false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            }
        } else if (a.fold) { // (n <= 4096: H stays in the Infinity Cache, no streaming hints)
            if (s->method == QN_BFGS) hipLaunchKernelGGL((s2_hpass_kernel<false, true, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            else hipLaunchKernelGGL((s2_hpass_kernel<false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
        } else if (a.tred) { // the update-reduce in the launch's tail
            if (s->method == QN_BFGS) {
                if (a.nt) hipLaunchKernelGGL((s2_hpass_kernel<true, true, false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
                else hipLaunchKernelGGL((s2_hpass_kernel<false, true, false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            } else {
                if (a.nt) hipLaunchKernelGGL((s2_hpass_kernel<true, false, false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
                else hipLaunchKernelGGL((s2_hpass_kernel<false, false, false, false, true>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            }
        } else if (s->method == QN_BFGS) {
            if (a.nt) hipLaunchKernelGGL((s2_hpass_kernel<true, true, false>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            else hipLaunchKernelGGL((s2_hpass_kernel<false, true, false>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
        } else {
            if (a.nt) hipLaunchKernelGGL((s2_hpass_kernel<true, false, false>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            else hipLaunchKernelGGL((s2_hpass_kernel<false, false, false>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
        }
        s->h_lower_stale = true; s->h_diag_stale = true;
        s->h_sliver_whole = a.sl_per != 0; // (sliver rows update every entry of their tiles; without them only the upper sub-blocks are kept)
        break;
    case QN_S2_HREDUCE:
        if (sh) hipLaunchKernelGGL(s2_hreduce_kernel<true>, dim3(a.nb), dim3(QN_S2_TPB), 0, st, a);
        else if (s->method == QN_SR1) hipLaunchKernelGGL((s2_hreduce_kernel<false, true>), dim3(2 * a.nb), dim3(QN_S2_TPB), 0, st, a);
        else if (a.touchq) hipLaunchKernelGGL((s2_hreduce_kernel<false, false, 32>), dim3(2 * a.nb + a.G), dim3(QN_S2_TPB), 0, st, a); // (+ the TOUCH workgroups)
        else hipLaunchKernelGGL(s2_hreduce_kernel<false>, dim3(2 * a.nb), dim3(QN_S2_TPB), 0, st, a); // (a workgroup per block-row and right-hand side)
        break;
    case QN_S2_GEVAL_A:
        if (sh) hipLaunchKernelGGL((s2_advance_kernel<true, true, QN_S2_GEVAL_A>), dim3(1), dim3(128), 0, st, a);
        else hipLaunchKernelGGL((s2_advance_kernel<false, true, QN_S2_GEVAL_A>), dim3(1), dim3(128), 0, st, a);
        break;
    case QN_S2_GHT_A:
        if (r.gobj) hipLaunchKernelGGL((s2_advance_kernel<false, true, QN_S2_GHT_A>), dim3(1), dim3(128), 0, st, a);
        else if (sh) hipLaunchKernelGGL((s2_advance_kernel<true, false, QN_S2_GHT_A>), dim3(1), dim3(128), 0, st, a); // (measurement: QN_S2SH_GEN1_TILES)
        else hipLaunchKernelGGL((s2_advance_kernel<false, false, QN_S2_GHT_A>), dim3(1), dim3(128), 0, st, a);
        break;
    case QN_S2_GCOMB: {
        QnS2GArgs g = s2g_args(r);
        g.wgS = a.wgS + (size_t)a.parity * (size_t)a.trows * QN_S2_ROW; // the half this launch writes (the next prologue reads it)
        g.wgV = a.wgV + (size_t)a.parity * (size_t)a.trows * QN_S2_ROW;
        if (sh) {
            g.ev_slice = a.evS + ((size_t)a.parity * (size_t)a.sh_world + (size_t)a.sh_rank) * (QN_S2SH_NEC * QN_S2_MAXG);
            hipLaunchKernelGGL(s2g_combine_kernel<true>, dim3(a.gw), dim3(256), 0, st, a, g);
        } else hipLaunchKernelGGL(s2g_combine_kernel<false>, dim3(a.gw), dim3(256), 0, st, a, g);
        break;
    }
    default:
        if (r.gobj && sh) hipLaunchKernelGGL((s2_advance_kernel<true, true, QN_S2_ADVANCE>), dim3(1), dim3(128), 0, st, a);
        else if (r.gobj) hipLaunchKernelGGL((s2_advance_kernel<false, true, QN_S2_ADVANCE>), dim3(1), dim3(128), 0, st, a);
        else if (sh) hipLaunchKernelGGL(s2_advance_kernel<true>, dim3(1), dim3(128), 0, st, a);
        else if (r.bnd) hipLaunchKernelGGL((s2_advance_kernel<false, false, QN_S2_ADVANCE, true>), dim3(1), dim3(128), 0, st, a);
        else hipLaunchKernelGGL(s2_advance_kernel<false>, dim3(1), dim3(128), 0, st, a);
        break;
    }
    s->stats.launches++;
    HIPCHK(hipGetLastError());
    return QN_OK;
}
static int s2_peek(Run& r) { // the control block the last enqueued launch writes -> host mirror
    qn_solver* s = r.s;
    HIPCHK(hipMemcpyAsync(s->hctl, s->s2_ctl + (r.s2_launches & 1), sizeof(QnCtl), hipMemcpyDeviceToHost, s->ctx->stream));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    s->stats.host_syncs++;
    return QN_OK;
}

// ---- one request of the sym2 machine = its launches and, row-sharded, the collectives between them ----
// An evaluation: the tiles, then (sharded) ONE exchange of the ranks' per-workgroup scalars -- 8 KB per rank, whatever n is.
static int s2_do_eval(Run& r, unsigned long long report_seq = 0) {
    if (r.gobj) { // the machine in a one-workgroup launch, the pass over A, the combine launch (which also stages the vectors of this point)
        QNCHK(s2_launch(r, QN_S2_GEVAL_A));
        QNCHK(s2g_enqueue_onepass(r));
        r.report_seq = report_seq; // (the batch's last launch reports: only the combine launch leaves the request as the host may see it)
        QNCHK(s2_launch(r, QN_S2_GCOMB));
        if (r.s2.sh_world > 1) { // row-sharded: the ranks' (m_r, S_r) and G_r'd per workgroup -- 8 KB per rank; an all-gather whatever the
            qn_solver* s = r.s;  // context's exchange mode is (the ranks are weighed with exp(m_r - M) before they are added)
            qn_context* c = s->ctx;
            ProfScope ps(s, KC_COMM);
            const size_t cnt = (size_t)QN_S2SH_NEC * QN_S2_MAXG;
            c->n_xchg_scalar++;
            QNCHK(exchange(c, s->s2_evS + (size_t)((r.s2_launches - 1) & 1) * (size_t)c->world * cnt, cnt));
        }
        return QN_OK;
    }
    if (r.proj) QNCHK(s2_launch(r, QN_S2_PROJ)); // (BackTrackingB: a projected trial's point is stored first; any other request passes through)
    if (report_seq) r.report_seq = report_seq;
    QNCHK(s2_launch(r, QN_S2_EVAL));
    qn_solver* s = r.s;
    qn_context* c = s->ctx;
    if (r.s2.sh_world > 1) {
        const size_t cnt = (size_t)QN_S2SH_NEC * QN_S2_MAXG;
        double* half = s->s2_evS + (size_t)((r.s2_launches - 1) & 1) * (size_t)c->world * cnt; // the half the launch above wrote
        if (c->trial_vector) { // the trial's partial n-vector rides on the scalar exchange (qn_sym2sh.hip.h: s2sh_vsumt_kernel; not a link of the control block's chain)
            {
                ProfScope ps(s, KC_EREDUCE);
                QnS2Args a = r.s2;
                a.parity = (int)(r.s2_launches & 1); // the block the evaluation launch has just handed on
                hipLaunchKernelGGL(s2sh_vsumt_kernel, dim3(a.nb), dim3(QN_S2_TPB), 0, c->stream, a);
                HIPCHK(hipGetLastError());
                s->stats.launches++;
            }
            ProfScope ps(s, KC_COMM);
            const XchgItem items[2] = {{half, cnt}, {s->symsh_xg, (size_t)s->T.n_pad}};
            c->n_xchg_scalar++;
            QNCHK(exchange_group(c, items, 2));
            return QN_OK;
        }
        ProfScope ps(s, KC_COMM);
        c->n_xchg_scalar++;
        if (c->use_allreduce) QNCHK(exchange_sum(c, half, cnt));
        else QNCHK(exchange(c, half, cnt));
    }
    return QN_OK;
}
// the accepted point's vectors: (sharded) this rank's slot sums, the exchange of ONE n-vector, the epilogue on every rank
static int s2_do_vec(Run& r) {
    qn_solver* s = r.s;
    qn_context* c = s->ctx;
    if (r.s2.sh_world > 1 && c->trial_vector && !r.gobj) return s2_launch(r, QN_S2_VECD); // (the vectors came with the evaluation's scalars)
    if (r.s2.sh_world > 1) {
        QNCHK(s2_launch(r, QN_S2_VSUM));
        {
            ProfScope ps(s, KC_COMM);
            c->n_xchg_vector++;
            if (r.gobj) QNCHK(exchange(c, r.obj->lgall, (size_t)s->T.n_pad)); // the ranks' G_r of the accepted point (weighed and added in rank order by s2g_vec_kernel)
            else if (c->use_allreduce) QNCHK(exchange_sum(c, s->symsh_xg, (size_t)s->T.n_pad));
            else QNCHK(exchange(c, s->symsh_xg, (size_t)s->T.n_pad));
        }
        return s2_launch(r, QN_S2_VEC);
    }
    return s2_launch(r, r.s2.fold ? QN_S2_HTILE : QN_S2_VEC);
}
// the update pass: tiles (unless the folded accept-reduce ran them), (sharded) partial sums and the exchange of [u, v], the reduce
static int s2_do_hpass(Run& r, bool tiles) {
    qn_solver* s = r.s;
    qn_context* c = s->ctx;
    if (r.gobj && r.s2.sh_world == 1) {
        if (!tiles) return fail(QN_ABNORMAL_TERMINATION, "sym2 (generic objective): tiles marked done without their launch");
        QNCHK(s2_launch(r, QN_S2_GHT_A));
        QNCHK(s2g_enqueue_tiles(r));
        return s2_launch(r, QN_S2_HREDUCE);
    }
    if (tiles && r.tiles1) { // H past the Infinity Cache: the machine in a one-workgroup launch, then the first-generation tile kernel
        QNCHK(s2_launch(r, QN_S2_GHT_A));
        QNCHK(s2g_enqueue_tiles(r));
    } else if (tiles) QNCHK(s2_launch(r, QN_S2_HTILE));
    if (r.s2.tred) return QN_OK; // (tail reduce: the tile launch has summed the slots itself)
    if (r.s2.sh_world > 1) {
        QNCHK(s2_launch(r, QN_S2_HSUM));
        ProfScope ps(s, KC_COMM);
        c->n_xchg_vector++;
        if (c->use_allreduce) QNCHK(exchange_sum(c, s->symsh_xg, 2 * (size_t)s->T.n_pad));
        else QNCHK(exchange(c, s->symsh_xg, 2 * (size_t)s->T.n_pad));
    }
    return s2_launch(r, QN_S2_HREDUCE);
}

// The control block of the launch that was told to report (Run.report_seq) -> host mirror, without synchronising the stream: the
// launch stores the block and then the sequence number into pinned memory, the host spins on the number.  (Round 3 copied the block
// back with hipMemcpyAsync + hipStreamSynchronize: a copy-engine transfer and an interrupt-driven wake-up at the end of every call,
// ~25 us of the 64 us a call cost beyond its iterations.)
static int s2_wait_report(Run& r, unsigned long long seq) {
    qn_solver* s = r.s;
    volatile unsigned long long* flag = s->hrep_flag;
    for (uint64_t spins = 0;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
        if ((spins & 0xfff) == 0xfff) { // the stream has drained and nothing reported: a launch failed
            hipError_t e = hipStreamQuery(s->ctx->stream);
            if (e == hipSuccess) { if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break; return fail(QN_ABNORMAL_TERMINATION, "sym2: the batch ended without a report"); }
            if (e != hipErrorNotReady) return fail(QN_ABNORMAL_TERMINATION, std::string("sym2 batch: ") + hipGetErrorString(e));
        }
    }
    memcpy(s->hctl, s->hrep, sizeof(QnCtl));
    s->stats.host_syncs++;
    return QN_OK;
}

// Where the inverse Hessian lives decides how fast the update pass runs on it while it is Infinity-Cache resident (qn_sym2.hip.h,
// PLACEMENT PROBE: one H in eight is 18 % slower for as long as it lives).  Probes that only move H's bytes do not see it, so the
// probe is the update kernel ITSELF: a direction pass with no update pending (H + 0: every tile read, written back with the
// values it had, slots into the scratch buffer) on the solver's H and on a second allocation -- a third one when the two differ,
// to know which was the odd one -- and H moves to the best.  Once per solver, at its first run on the second-generation path;
// ~1 ms, up to three times H's size for that long; QN_H_PLACEMENT=0 switches it off, =2 prints what it measured.
static int place_h(Run& r) {
    qn_solver* s = r.s;
    if (s->h_placed) return QN_OK;
    s->h_placed = true;
    const size_t np = s->T.n_pad, bytes = (size_t)s->T.rpr * np * sizeof(double);
    const char* sw = getenv("QN_H_PLACEMENT");
    if ((sw && atoi(sw) == 0) || s->ctx->world != 1 || bytes > ((size_t)330 << 20) || r.s2.fold) return QN_OK; // (n <= 6144: H's half is an Infinity Cache tenant; larger H is streamed from HBM anyway)
    hipStream_t st = s->ctx->stream;
    // the request: a direction pass (one right-hand side, g in both places), nothing pending; the vectors it multiplies are whatever
    // the fused buffers hold (zeros before the first run) -- only the duration matters, and H comes back as it was
    QnCtl* pc = s->hrep; // (pinned, device-mapped; the report area is free between calls)
    memcpy(pc, s->hctl, sizeof(QnCtl));
    pc->phase = QN_PH_REQ_HPASS; pc->serviced = 0; pc->hp_nrhs = 1; pc->pending = 0; pc->after_state = QN_ST_AFTER_DIR; pc->sym2 = 1; pc->fused = 1;
    pc->spec_tiles = 0; pc->sc = 0; pc->xc = 0;
    // ... and, in front of every timed pass, what an iteration has in front of it: two evaluations (Q's half streamed twice).  Timed
    // alone on H the update kernel showed the same 24.4 us on allocations where, in the run, it then took 29.6-30.3 us (2-3 processes
    // in 20, tools/modes_ab.sh): the slow mode is H sharing the Infinity Cache with Q, not H by itself.
    QnCtl* pe = nullptr;
    if (hipHostMalloc((void**)&pe, sizeof(QnCtl), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); pe = nullptr; }
    if (pe) {
        memcpy(pe, s->hctl, sizeof(QnCtl));
        pe->phase = QN_PH_REQ_EVAL; pe->serviced = 0; pe->sym2 = 1; pe->fused = 1; pe->sc = 0; pe->xc = 0;
        pe->ev_kind = QN_REQ_T; pe->t = 1.0; pe->status = -1;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { // no probe: H stays where it is, nothing is left behind
        (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        if (pe) (void)hipHostFree(pe);
        return QN_OK;
    }
    // (every failure below -- inside time_on too -- comes back as `status` and leaves through the one cleanup path at the end: the
    // candidates that are not kept, the events and the pinned block are freed, H stays the solver's own)
    auto time_on = [&](double* H, float* out_us) -> int {
        QnS2Args a = r.s2;
        a.H = H; a.parity = 0; a.ctl_first = pc; a.rep_seq = 0;
        a.tred = 0; // (the probe times the tiles; the tail reduce would overwrite u, v and g with the probe's sums)
        float t[6];
        for (int rep = 0; rep < 6; ++rep) {
            if (pe) {
                QnS2Args ae = a;
                ae.ctl_first = pe;
                for (int e = 0; e < 2; ++e) {
                    if (ae.pair && ae.ring) hipLaunchKernelGGL(s2_evalr_kernel<false>, dim3(ae.G), dim3(QN_S2R_TPB), 0, st, ae);
                    else if (ae.pair) hipLaunchKernelGGL(s2_eval_kernel<true>, dim3(ae.G), dim3(QN_S2_TPB), 0, st, ae);
                    else if (ae.ntq) hipLaunchKernelGGL((s2_eval_kernel<false, false, true>), dim3(ae.G), dim3(QN_S2_TPB), 0, st, ae);
                    else hipLaunchKernelGGL(s2_eval_kernel<false>, dim3(ae.G), dim3(QN_S2_TPB), 0, st, ae);
                }
            }
            HIPCHK(hipEventRecord(e0, st));
            if (s->method == QN_BFGS) hipLaunchKernelGGL((s2_hpass_kernel<false, true, false>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            else hipLaunchKernelGGL((s2_hpass_kernel<false, false, false>), dim3(a.G), dim3(QN_S2_TPB), 0, st, a);
            HIPCHK(hipEventRecord(e1, st));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            t[rep] = ms * 1e3f;
        }
        std::sort(t + 2, t + 6); // (the first two repetitions bring the tiles in)
        *out_us = t[3];
        return QN_OK;
    };
    double* cand[3] = {s->H, nullptr, nullptr};
    float us[3] = {0.f, 0.f, 0.f};
    int ncand = 1, keep = 0;
    int status = time_on(cand[0], &us[0]);
    for (int k = 1; k < 3 && status == QN_OK; ++k) {
        if (k == 2 && std::fabs(us[0] - us[1]) <= 0.06f * std::min(us[0], us[1])) break; // the two agree: both are the common case
        if (hipMalloc((void**)&cand[k], bytes) != hipSuccess) { (void)hipGetLastError(); cand[k] = nullptr; break; } // (no room: keep what there is)
        ++ncand;
        if (hipMemcpyAsync(cand[k], s->H, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) { status = fail(QN_ABNORMAL_TERMINATION, "H placement: copy failed"); break; }
        status = time_on(cand[k], &us[k]);
    }
    if (status == QN_OK)
        for (int k = 1; k < ncand; ++k)
            if (us[k] < 0.96f * us[keep]) keep = k; // (the one it has, unless another is clearly better)
    if (sw && atoi(sw) == 2) fprintf(stderr, "[qn] H placement: %d candidates, update kernel %.2f %.2f %.2f us, kept %d\n", ncand, us[0], us[1], us[2], keep);
    (void)hipStreamSynchronize(st);
    for (int k = 0; k < ncand; ++k)
        if (k != keep && cand[k]) (void)hipFree(cand[k]);
    s->H = cand[keep];
    s->V.H = s->H;
    r.s2.H = s->H;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (pe) (void)hipHostFree(pe);
    return status;
}

static int launch_ctl_mask(Run& r, int expect_mask) {
    qn_solver* s = r.s;
    ProfScope ps(s, KC_CTL);
    // fused path: the step only sums per-workgroup partials, one wave per column (9 evaluation + 3 update-pass columns);
    // generic path: it sweeps n-vectors with 1024 threads
    const bool hp = (expect_mask & ((1 << QN_PH_REQ_HPASS) | (1 << QN_PH_REQ_HPASS_EVAL))) != 0;
    const bool ev = (expect_mask & ((1 << QN_PH_REQ_EVAL) | (1 << QN_PH_REQ_HPASS_EVAL))) != 0;
    const dim3 blk(r.fused ? ((hp && ev) ? 768 : 576) : QN_CTL_TPB);
    if (r.oracle_tpl == QN_ORACLE_QUAD)
        hipLaunchKernelGGL(ctl_step_kernel<QN_ORACLE_QUAD>, dim3(1), blk, 0, s->ctx->stream, s->ctl, s->V, expect_mask);
    else
        hipLaunchKernelGGL(ctl_step_kernel<QN_ORACLE_GENERIC>, dim3(1), blk, 0, s->ctx->stream, s->ctl, s->V, expect_mask);
    s->stats.launches++;
    HIPCHK(hipGetLastError());
    return QN_OK;
}
static int launch_ctl(Run& r, int expect_phase) { return launch_ctl_mask(r, 1 << expect_phase); }

template <int R, int U>
static void launch_eval_fused(hipStream_t st, const QnEvalFusedArgs& a) {
    hipLaunchKernelGGL((quad_eval_fused_kernel<R, U>), dim3(a.T.rpr / R), dim3(QN_TPB), 0, st, a);
}
template <int R, int U>
static void launch_hpass_fused(hipStream_t st, const QnHPassFusedArgs& a) {
    hipLaunchKernelGGL((h_pass_fused_kernel<R, U>), dim3(a.T.rpr / R), dim3(QN_TPB), 0, st, a);
}
#define QN_DISPATCH_RU(fn, R_, U_, ...)                                             \
    do {                                                                            \
        const int key_ = (R_) * 10 + (U_);                                          \
        switch (key_) {                                                             \
        case 21: fn<2, 1>(__VA_ARGS__); break;                                      \
        case 22: fn<2, 2>(__VA_ARGS__); break;                                      \
        case 24: fn<2, 4>(__VA_ARGS__); break;                                      \
        case 41: fn<4, 1>(__VA_ARGS__); break;                                      \
        case 42: fn<4, 2>(__VA_ARGS__); break;                                      \
        case 44: fn<4, 4>(__VA_ARGS__); break;                                      \
        case 82: fn<8, 2>(__VA_ARGS__); break;                                      \
        case 161: fn<16, 1>(__VA_ARGS__); break;                                    \
        default: fn<8, 1>(__VA_ARGS__); break;                                      \
        }                                                                           \
    } while (0)

static int enqueue_eval_fused(Run& r, int after_h) {
    qn_solver* s = r.s;
    qn_context* c = s->ctx;
    QnEvalFusedArgs a{};
    a.Q = r.obj->Q; a.T = s->T; a.T.cs = 1; a.F = s->V.F; a.ctl = s->ctl; a.expect_phase = QN_PH_REQ_EVAL;
    a.after_h = after_h; a.world = c->world;
    if (r.sym) {
        QnSymEvalArgs y{};
        y.Q = r.obj->Q; y.T = a.T; y.F = a.F; y.ctl = s->ctl; y.expect_phase = QN_PH_REQ_EVAL; y.after_h = after_h; y.nb = s->sym_nb; y.part = s->sym_part;
        y.nt = 0; // Q is only read: non-temporal loads measured no gain at n = 32768 and -4 % at n = 16384
        if (c->world > 1) { // row-sharded: this rank's circulant half, partial sums gathered, epilogue on every rank
            y.sh = sym_shard(s);
            {
                ProfScope ps(s, KC_EVAL);
                hipLaunchKernelGGL(sym_eval_tile_kernel, dim3(qn_symsh_ntiles(y.nb, y.sh.nbl, y.sh.ioff)), dim3(QN_SYM_TPB), 0, c->stream, y);
            }
            {
                ProfScope ps(s, KC_EREDUCE);
                hipLaunchKernelGGL(symsh_eval_sum_kernel, dim3(y.nb), dim3(256), 0, c->stream, y);
            }
            HIPCHK(hipGetLastError());
            {
                ProfScope ps(s, KC_COMM);
                c->n_xchg_vector++;
                if (c->use_allreduce) QNCHK(exchange_sum(c, s->symsh_xg, (size_t)s->T.n_pad));
                else QNCHK(exchange(c, s->symsh_xg, (size_t)s->T.n_pad));
            }
            {
                ProfScope ps(s, KC_EREDUCE);
                hipLaunchKernelGGL(symsh_eval_epi_kernel, dim3(y.nb), dim3(256), 0, c->stream, y);
            }
            s->stats.launches += 3;
            HIPCHK(hipGetLastError());
            return QN_OK;
        }
        {
            ProfScope ps(s, KC_EVAL);
            hipLaunchKernelGGL(sym_eval_tile_kernel, dim3(y.nb * (y.nb + 1) / 2), dim3(QN_SYM_TPB), 0, c->stream, y);
        }
        {
            ProfScope ps(s, KC_EREDUCE);
            hipLaunchKernelGGL(sym_eval_reduce_kernel, dim3(y.nb), dim3(256), 0, c->stream, y);
        }
        s->stats.launches += 2;
        HIPCHK(hipGetLastError());
        return QN_OK;
    }
    {
        ProfScope ps(s, KC_EVAL);
        QN_DISPATCH_RU(launch_eval_fused, s->R, s->U, c->stream, a);
        s->stats.launches++;
        HIPCHK(hipGetLastError());
    }
    if (c->world > 1) {
        ProfScope ps(s, KC_COMM);
        const XchgItem items[3] = {{s->V.F.GT, (size_t)s->T.rpr}, {s->V.F.Y, (size_t)s->T.rpr}, {s->V.F.evp, (size_t)QN_NEVP * s->V.F.nblk}};
        c->n_xchg_vector++;
        QNCHK(exchange_group(c, items, 3));
    }
    return QN_OK;
}

static int enqueue_hpass_fused(Run& r) {
    qn_solver* s = r.s;
    qn_context* c = s->ctx;
    QnHPassFusedArgs a{};
    a.H = s->H; a.T = s->T; a.T.cs = 1; a.F = s->V.F; a.ctl = s->ctl; a.expect_phase = QN_PH_REQ_HPASS;
    if (r.sym) {
        QnSymHPassArgs y{};
        y.H = s->H; y.T = a.T; y.F = a.F; y.ctl = s->ctl; y.expect_phase = QN_PH_REQ_HPASS; y.nb = s->sym_nb; y.part = s->sym_part;
        y.nt = s->T.n_pad >= 8192; // past the Infinity Cache (same-box A/B: +7 % at n = 32768, +3 % at 8192, -1 % at 4096)
        s->h_lower_stale = true;
        if (c->world > 1) {
            y.sh = sym_shard(s);
            {
                ProfScope ps(s, KC_HPASS);
                hipLaunchKernelGGL(sym_hpass_tile_kernel, dim3(qn_symsh_ntiles(y.nb, y.sh.nbl, y.sh.ioff)), dim3(QN_SYM_TPB), 0, c->stream, y);
            }
            {
                ProfScope ps(s, KC_HREDUCE);
                hipLaunchKernelGGL(symsh_hpass_sum_kernel, dim3(y.nb), dim3(256), 0, c->stream, y);
            }
            HIPCHK(hipGetLastError());
            {
                ProfScope ps(s, KC_COMM);
                c->n_xchg_vector++;
                if (c->use_allreduce) QNCHK(exchange_sum(c, s->symsh_xg, 2 * (size_t)s->T.n_pad));
                else QNCHK(exchange(c, s->symsh_xg, 2 * (size_t)s->T.n_pad));
            }
            {
                ProfScope ps(s, KC_HREDUCE);
                hipLaunchKernelGGL(symsh_hpass_epi_kernel, dim3(y.nb), dim3(256), 0, c->stream, y);
            }
            s->stats.launches += 3;
            HIPCHK(hipGetLastError());
            return QN_OK;
        }
        {
            ProfScope ps(s, KC_HPASS);
            hipLaunchKernelGGL(sym_hpass_tile_kernel, dim3(y.nb * (y.nb + 1) / 2), dim3(QN_SYM_TPB), 0, c->stream, y);
        }
        {
            ProfScope ps(s, KC_HREDUCE);
            hipLaunchKernelGGL(sym_hpass_reduce_kernel, dim3(y.nb), dim3(256), 0, c->stream, y);
        }
        s->stats.launches += 2;
        HIPCHK(hipGetLastError());
        return QN_OK;
    }
    {
        ProfScope ps(s, KC_HPASS);
        QN_DISPATCH_RU(launch_hpass_fused, s->R, s->U, c->stream, a);
        s->stats.launches++;
        HIPCHK(hipGetLastError());
    }
    if (c->world > 1) {
        ProfScope ps(s, KC_COMM);
        const XchgItem items[3] = {{s->V.F.UN, (size_t)s->T.rpr}, {s->V.F.VV, (size_t)s->T.rpr}, {s->V.F.hpp, (size_t)QN_NHPP * s->V.F.nblk}};
        c->n_xchg_vector++;
        QNCHK(exchange_group(c, items, 3));
    }
    return QN_OK;
}

// enqueue the evaluation of the oracle at the requested point (predicated on phase == REQ_EVAL)
static int enqueue_eval(Run& r, int after_h = 0) {
    qn_solver* s = r.s;
    qn_context* c = s->ctx;
    if (r.fused) return enqueue_eval_fused(r, after_h);
    if (r.oracle_tpl == QN_ORACLE_QUAD) {
        QnQuadArgs a{};
        a.Q = r.obj->Q; a.T = s->T; a.T.cs = s->qcs;
        a.x = s->V.x; a.d = s->V.d; a.xt = s->V.xt;
        a.llb = s->V.llb; a.lub = s->V.lub;
        a.out = s->V.q + (size_t)c->rank * s->qcs * s->T.rpr;
        a.ctl = s->ctl; a.expect_phase = QN_PH_REQ_EVAL;
        {
            ProfScope ps(s, KC_EVAL);
            QNCHK(launch_quad_R(s->R, c->stream, a));
            s->stats.launches++;
        }
        if (c->world > 1) {
            ProfScope ps(s, KC_COMM);
            c->n_xchg_vector++;
            QNCHK(exchange(c, s->V.q, (size_t)s->qcs * s->T.rpr));
        }
        return QN_OK;
    }
    hipLaunchKernelGGL(trial_point_kernel, dim3(std::min(1024, (s->T.n_pad + 255) / 256)), dim3(256), 0, c->stream, s->V.x, s->V.d,
                       s->V.xt, s->T.n_pad, s->ctl, (int)QN_PH_REQ_EVAL, s->V.llb, s->V.lub);
    s->stats.launches++;
    HIPCHK(hipGetLastError());
    if (r.o->kind == QN_ORACLE_HOST) {
        HIPCHK(hipMemcpyAsync(s->hx, s->V.xt, s->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        s->stats.host_syncs++;
        double f = NAN;
        if (r.o->host_fn(r.o->host_user, s->hx, s->n, &f, s->hg) != 0) return fail(QN_ABNORMAL_TERMINATION, "host oracle returned non-zero");
        s->hg[s->n] = f; // pinned staging: g[0..n) then f
        HIPCHK(hipMemcpyAsync(s->V.gt, s->hg, s->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(s->f_dev, s->hg + s->n, sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return QN_OK;
    }
    if (r.o->kind == QN_ORACLE_OBJECTIVE) return lse_enqueue_eval(r.obj, s->V.xt, s->f_dev, s->V.gt); // log-sum-exp
    // device closure
    if (r.o->device_fn(r.o->device_user, (void*)c->stream, s->V.xt, s->n, s->f_dev, s->V.gt) != 0)
        return fail(QN_ABNORMAL_TERMINATION, "device oracle returned non-zero");
    return QN_OK;
}

static int enqueue_hpass_req(Run& r) {
    qn_solver* s = r.s;
    qn_context* c = s->ctx;
    if (r.fused) return enqueue_hpass_fused(r);
    if (r.sym_generic) {
        QnSymHPassArgs y{};
        y.H = s->H; y.T = s->T; y.T.cs = 1; y.ctl = s->ctl; y.expect_phase = QN_PH_REQ_HPASS; y.nb = s->sym_nb; y.part = s->sym_part;
        y.generic = 1; y.gsp = s->V.sp; y.gup = s->V.up; y.gvy = s->V.y; y.gvg = s->V.g; y.ghp = s->V.hp;
        y.nt = s->T.n_pad >= 8192; // past the Infinity Cache (same-box A/B: +7 % at n = 32768, +3 % at 8192, -1 % at 4096)
        s->h_lower_stale = true;
        if (c->world > 1) { // row-sharded: the circulant half of this rank's block-rows; partial sums gathered, totals on every rank
            y.sh = sym_shard(s);
            {
                ProfScope ps(s, KC_HPASS);
                hipLaunchKernelGGL(sym_hpass_tile_kernel, dim3(qn_symsh_ntiles(y.nb, y.sh.nbl, y.sh.ioff)), dim3(QN_SYM_TPB), 0, c->stream, y);
            }
            {
                ProfScope ps(s, KC_HREDUCE);
                hipLaunchKernelGGL(symsh_hpass_sum_kernel, dim3(y.nb), dim3(256), 0, c->stream, y);
            }
            HIPCHK(hipGetLastError());
            {
                ProfScope ps(s, KC_COMM);
                c->n_xchg_vector++;
                if (c->use_allreduce) QNCHK(exchange_sum(c, s->symsh_xg, 2 * (size_t)s->T.n_pad));
                else QNCHK(exchange(c, s->symsh_xg, 2 * (size_t)s->T.n_pad));
            }
            {
                ProfScope ps(s, KC_HREDUCE);
                hipLaunchKernelGGL(symsh_hpass_epi_generic_kernel, dim3(y.nb), dim3(256), 0, c->stream, y);
            }
            s->stats.launches += 3;
            HIPCHK(hipGetLastError());
            return QN_OK;
        }
        {
            ProfScope ps(s, KC_HPASS);
            hipLaunchKernelGGL(sym_hpass_tile_kernel, dim3(y.nb * (y.nb + 1) / 2), dim3(QN_SYM_TPB), 0, c->stream, y);
        }
        {
            ProfScope ps(s, KC_HREDUCE);
            hipLaunchKernelGGL(sym_hpass_reduce_kernel, dim3(y.nb), dim3(256), 0, c->stream, y);
        }
        s->stats.launches += 2;
        HIPCHK(hipGetLastError());
        return QN_OK;
    }
    QnHPassArgs a = hpass_args(s, QN_PH_REQ_HPASS);
    {
        ProfScope ps(s, KC_HPASS);
        QNCHK(launch_hpass_R(s, a));
    }
    if (c->world > 1) {
        ProfScope ps(s, KC_COMM);
        c->n_xchg_vector++;
        QNCHK(exchange(c, s->V.hp, (size_t)s->hcs * 2 * s->T.rpr));
    }
    return QN_OK;
}
