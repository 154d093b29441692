// qn_host_newton.hip.h -- host side, part 6 of 7: Newton's direction (row f2): blocked Cholesky and pivoted LU drivers, triangular sweeps.
#pragma once
// ---- Newton direction (newton/mod.rs:26-49): Cholesky factorisation + four triangular solves, or the n <= 5 kernel ----
static int newton_alloc(qn_solver* s) {
    if (s->newton_w) return QN_OK;
    size_t n64 = (s->n + QN_NB - 1) / QN_NB * QN_NB;
    s->newton_big = n64 > QN_TS; // 512-wide triangular blocks (inverses doubled up from the 64-wide ones)
    if (s->newton_big) n64 = (s->n + QN_TS - 1) / QN_TS * QN_TS;
    s->newton_n64 = n64;
    hipStream_t st = s->ctx->stream;
    QNCHK(dev_alloc_zero(&s->newton_w, n64 * n64, st));
    QNCHK(dev_alloc_zero(&s->newton_x, 2 * n64, st));
    QNCHK(dev_alloc_zero(&s->newton_invl, n64 * QN_NB, st));
    if (s->newton_big) { // inverse blocks of width 128, 256, 512, the transposed 512 ones, and the product scratch
        QNCHK(dev_alloc_zero(&s->newton_inv2, n64 * (128 + 256 + 512 + 512 + 256), st));
    }
    HIPCHK(hipMalloc((void**)&s->newton_fail, 2 * sizeof(int)));
    HIPCHK(hipMemsetAsync(s->newton_fail, 0, 2 * sizeof(int), st));
    HIPCHK(hipMalloc((void**)&s->newton_piv, n64 * sizeof(int)));
    HIPCHK(hipMalloc((void**)&s->newton_perm, n64 * sizeof(int)));
    s->newton_piv_host.resize(n64);
    s->V.nfail = s->newton_fail;
    return QN_OK;
}

static inline int rowdot_grid(int nrows) { return std::max(1, std::min(1024, nrows <= 4096 ? (nrows + 3) / 4 : (nrows + 15) / 16)); }

// inverses of the 512-wide diagonal blocks of L from the 64-wide ones: inv([A 0; B C]) = [A^-1 0; -C^-1 B A^-1, C^-1]
static int newton_build_block_inverses(qn_solver* s) {
    hipStream_t st = s->ctx->stream;
    const size_t n64 = s->newton_n64, ld = n64;
    double* lvl[4] = {s->newton_invl, s->newton_inv2, s->newton_inv2 + n64 * 128, s->newton_inv2 + n64 * (128 + 256)};
    double* invT = s->newton_inv2 + n64 * (128 + 256 + 512);
    double* T = s->newton_inv2 + n64 * (128 + 256 + 512 + 512);
    for (int l = 0; l < 3; ++l) {
        const int sz = QN_NB << l;
        const int npairs = (int)(n64 / (2 * (size_t)sz));
        const size_t ss = (size_t)sz * sz;
        const dim3 grid(sz / QN_NB, sz / QN_NB, npairs);
        QnBatchGemm g1{s->newton_w + (size_t)sz * ld, ld, 2 * (size_t)sz * ld + 2 * (size_t)sz, lvl[l], (size_t)sz, 2 * ss, T, (size_t)sz, ss, sz, 1.0};
        hipLaunchKernelGGL(tri_batch_gemm_kernel, grid, dim3(256), 0, st, g1); // T = B A^-1
        QnBatchGemm g2{lvl[l] + ss, (size_t)sz, 2 * ss, T, (size_t)sz, ss, lvl[l + 1] + (size_t)sz * 2 * sz, 2 * (size_t)sz, 4 * ss, sz, -1.0};
        hipLaunchKernelGGL(tri_batch_gemm_kernel, grid, dim3(256), 0, st, g2); // lower-left = -C^-1 T
        hipLaunchKernelGGL(tri_inv_assemble_kernel, dim3(1024), dim3(256), 0, st, lvl[l], lvl[l + 1], sz, npairs);
    }
    const int nb = (int)(n64 / QN_TS);
    hipLaunchKernelGGL(tri_transpose_blocks_kernel, dim3(QN_TS / 32, QN_TS / 32, nb), dim3(256), 0, st, lvl[3], invT, (int)QN_TS, nb);
    s->stats.launches += 10;
    HIPCHK(hipGetLastError());
    return QN_OK;
}

static int newton_tri_solve(qn_solver* s, double* x, double* tmp) { // x <- (L L')^-1 x ; tmp: n64 scratch
    hipStream_t st = s->ctx->stream;
    const int n64 = (int)s->newton_n64;
    const size_t ld = s->newton_n64;
    if (s->newton_big) {
        const double* inv = s->newton_inv2 + (size_t)n64 * (128 + 256);
        const double* invT = s->newton_inv2 + (size_t)n64 * (128 + 256 + 512);
        const size_t bb = (size_t)QN_TS * QN_TS;
        const int nb = n64 / QN_TS;
        for (int K = 0; K < nb; ++K) { // L y = x : rhs x (consumed), solution tmp
            const int K0 = K * QN_TS, below = n64 - K0 - QN_TS;
            hipLaunchKernelGGL(tri_rowdot512_kernel, dim3(rowdot_grid(QN_TS)), dim3(256), 0, st, inv + K * bb, (size_t)QN_TS, (int)QN_TS, x + K0, tmp + K0, 0);
            if (below > 0)
                hipLaunchKernelGGL(tri_rowdot512_kernel, dim3(rowdot_grid(below)), dim3(256), 0, st, s->newton_w + (size_t)(K0 + QN_TS) * ld + K0, ld,
                                   below, tmp + K0, x + K0 + QN_TS, 1);
        }
        for (int K = nb - 1; K >= 0; --K) { // L' z = y : rhs tmp (consumed), solution x
            const int K0 = K * QN_TS;
            hipLaunchKernelGGL(tri_rowdot512_kernel, dim3(rowdot_grid(QN_TS)), dim3(256), 0, st, invT + K * bb, (size_t)QN_TS, (int)QN_TS, tmp + K0, x + K0, 0);
            if (K0 > 0)
                hipLaunchKernelGGL(tri_coldot512_kernel, dim3((K0 + 63) / 64), dim3(256), 0, st, s->newton_w + (size_t)K0 * ld, ld, K0, x + K0, tmp);
        }
        s->stats.launches += 4 * (uint64_t)nb;
        HIPCHK(hipGetLastError());
        return QN_OK;
    }
    for (int k0 = 0; k0 < n64; k0 += QN_NB) { // L y = x : rhs x (consumed), solution tmp
        const int below = n64 - k0 - QN_NB;
        const int grid = std::max(1, std::min(256, (below + 3) / 4));
        hipLaunchKernelGGL(tri_fwd_step_kernel, dim3(grid), dim3(256), 0, st, s->newton_w, ld, k0, n64,
                           s->newton_invl + (size_t)(k0 / QN_NB) * QN_NB * QN_NB, x, tmp);
    }
    for (int k0 = n64 - QN_NB; k0 >= 0; k0 -= QN_NB) { // L' z = y : rhs tmp (consumed), solution x
        const int grid = std::max(1, std::min(256, (k0 + 255) / 256));
        hipLaunchKernelGGL(tri_bwd_step_kernel, dim3(grid), dim3(256), 0, st, s->newton_w, ld, k0,
                           s->newton_invl + (size_t)(k0 / QN_NB) * QN_NB * QN_NB, tmp, x);
    }
    HIPCHK(hipGetLastError());
    return QN_OK;
}


// The second stream of Newton's look-ahead (LU: the bulk of a trailing update beside the next panel's chain).  A panel step's
// workgroups hold 8 waves of 232 registers -- a CU running even ONE workgroup of the update (4 waves of 196) has no room for them, and
// a grid of 16 k update workgroups never leaves a CU empty: without a mask the chain waits for the bulk to drain and nothing overlaps
// (measured: 65.6 ms with the second stream, 65.1 without).  So the bulk's stream may use only `QN_LU_BULK_CUS` of the 256 CUs
// (default 192: the low mask bits -- 24 CUs of every XCD, tools/cu_mask_probe.hip; the panel's grid is 58 workgroups).
static int ensure_masked_stream(qn_context* c) {
    if (c->stream_lu) return QN_OK;
    static const int bulk_cus = getenv("QN_LU_BULK_CUS") ? atoi(getenv("QN_LU_BULK_CUS")) : 192;
    uint32_t mask[8];
    for (int w = 0; w < 8; ++w) mask[w] = 0;
    const int keep = std::max(32, std::min(256, bulk_cus));
    for (int b = 0; b < keep; ++b) mask[b >> 5] |= 1u << (b & 31);
    c->lu_bulk_cus = keep;
    if (keep >= 256 || hipExtStreamCreateWithCUMask(&c->stream_lu, 8, mask) != hipSuccess) {
        (void)hipGetLastError();
        c->lu_bulk_cus = 256;
        HIPCHK(hipStreamCreateWithFlags(&c->stream_lu, hipStreamNonBlocking));
    }
    return QN_OK;
}

// Pivoted LU of the staged Hessian and the two solves (qn_lu.hip.h); leaves d in V.d, z = H^-1 d in V.s, and newton_fail[0] = 1
// when a pivot column is exactly zero (then QN_ST_AFTER_NEWTON takes -g, newton/mod.rs:43-46).
// A bounded wait of the one-launch LU kernels expired (the bound is a number of polls: a co-tenant on the GPU, or workgroups that were
// not resident together, can do that).  The factorisation is run again launch by launch -- same bits -- and so are the next
// QN_LU_RETRY_AFTER ones; then the one-launch kernels get another chance (one transient used to cost the solver 54
// instead of 47 ms per iteration for the rest of its life, invisibly).  Counted in qn_stats.newton_lu_sync_timeouts, said once on stderr.
#define QN_LU_RETRY_AFTER 8
static void lu_note_timeout(qn_solver* s) {
    s->newton_lu_no_persist = 1;
    s->newton_lu_timeout_fallback = 1;
    s->newton_lu_sync_timeouts++;
    s->stats.newton_lu_sync_timeouts = s->newton_lu_sync_timeouts;
    s->newton_lu_runs--;
    static std::atomic<int> said{0};
    if (said.exchange(1) == 0)
        fprintf(stderr, "[qn] Newton / LU: a bounded wait of the one-launch kernels expired; this factorisation and the next %d run launch by launch "
                        "(same result, slower; qn_stats.newton_lu_sync_timeouts counts these)\n", QN_LU_RETRY_AFTER);
}
static int enqueue_newton_lu(qn_solver* s, const double* hsrc, size_t ld_src) {
    hipStream_t st = s->ctx->stream;
    if (s->newton_lu_timeout_fallback && ++s->newton_lu_timeout_fallback > 1 + QN_LU_RETRY_AFTER) { // (the re-run itself is the first)
        s->newton_lu_timeout_fallback = 0;
        s->newton_lu_no_persist = 0;
    }
    const int n = (int)s->n, n64 = (int)s->newton_n64;
    const int nlu = (n + QN_NB - 1) / QN_NB * QN_NB; // the factorisation works on whole 64-blocks; identity padding
    const size_t ld = s->newton_n64;
    double* W = s->newton_w;
    int* flag = s->newton_fail;
    s->newton_lu_runs++;
    HIPCHK(hipMemsetAsync(flag, 0, 2 * sizeof(int), st));
    hipLaunchKernelGGL(newton_stage_kernel, dim3(2048), dim3(256), 0, st, W, ld, n, n64, hsrc, ld_src, 0); // both triangles
    uint64_t launches = 1;
    const size_t panel_doubles = (size_t)QN_NB * QN_LU_PT * QN_LU_RPT; // (two buffers: the look-ahead writes the next panel's while this one's is still read)
    if (!s->newton_panel) HIPCHK(hipMalloc((void**)&s->newton_panel, 2 * panel_doubles * sizeof(double)));
    if (!s->newton_sync) HIPCHK(hipMalloc((void**)&s->newton_sync, 128 * sizeof(int)));
    HIPCHK(hipMemsetAsync(s->newton_sync, 0, 128 * sizeof(int), st));
    // role A split over the workgroups of one XCD (qn_lu_split.hip.h): the parts' records, two panels' worth, all words the sentinel
    // (measured at n = 8192, ms per Newton iteration: every panel split 44.2 = no split 44.2 -- a split pivot step costs 2.6-3.2 us at any height, one
    // workgroup's 1.6 (1792 rows) ... 2.6 (7552 rows) us and what the split saves is column traffic --; from 2560 rows 42.1, 3136: 42.0, 4160: 41.4-41.7,
    // 5184: 42.5.  The first ~16 panels gain nothing either way: there the previous panel's bulk update, not the chain, sets the period)
    const int lu_split_min = s->newton_lu_split_min; // panels of fewer rows: one workgroup (QN_OPT_LU_SPLIT_MIN_ROWS, default 4160)
    if (s->newton_lu_split > 1) {
        if (!s->newton_rec) HIPCHK(hipMalloc((void**)&s->newton_rec, 2 * (size_t)QN_LUS_REC_WORDS * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(s->newton_rec, 0xff, 2 * (size_t)QN_LUS_REC_WORDS * sizeof(unsigned long long), st));
    }
    static const int lu_persist_on = getenv("QN_LU_PERSIST") ? atoi(getenv("QN_LU_PERSIST")) : 1;
    const bool persist = lu_persist_on && !s->newton_lu_no_persist;
    const int spin_max = s->newton_lu_force_timeout ? 0 : QN_LU_SPIN_MAX; // (diagnostics: every wait that is not satisfied at once gives up -> the fallback below)
    // LOOK-AHEAD (round 4, as in the Cholesky path: enqueue_newton).  A panel's factorisation is a chain of 17 small launches (one CU
    // working through 64 pivot steps: 150-400 us); what it needs from the previous panel is its own 64 columns brought up to date.
    // So after panel p: its swaps, U12 solve and update on the NEXT panel's columns on this stream, and everything else -- the swaps
    // on the finished columns left of it, swaps / solve / MFMA update on the columns right of the next panel -- on the context's
    // second stream beside panel p + 1's chain (events E_p: panel p and its pivots are final; F_p: the bulk of panel p is done, awaited
    // before the same columns are touched again).  The bulk update caps its occupancy as the Cholesky one does.
    qn_context* c = s->ctx;
    static const int lu_la_on = getenv("QN_LU_LOOKAHEAD") ? atoi(getenv("QN_LU_LOOKAHEAD")) : 1;
    const int npanels = nlu / QN_NB;
    const bool la = lu_la_on && !s->newton_lu_no_la && npanels >= 8;
    size_t bulk_lds = 0;
    if (la) {
        QNCHK(ensure_masked_stream(c));
        while ((int)c->la_events.size() < 3 * npanels) { hipEvent_t e = nullptr; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->la_events.push_back(e); }
        static std::atomic<int> attr_state[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        bulk_lds = (size_t)(80 * 1024 - 2 * QN_NB * (QN_NB + 1) * 8 - 1024); // (two workgroups per CU: room for the chain's on every CU)
        if (dev >= 0 && dev < 64 && attr_state[dev].load() == 0) {
            const bool ok = hipFuncSetAttribute((const void*)lu_gemm2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bulk_lds) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            attr_state[dev].store(ok ? 1 : 2);
        }
        if (dev < 0 || dev >= 64 || attr_state[dev].load() != 1) bulk_lds = 0;
    }
    int last_f = -1;
    bool last_strip = false; // the update of panel last_f ran its first tile column as a launch of its own (event 2 npanels + last_f behind it)
    static const int lu_strip = getenv("QN_LU_STRIP") ? atoi(getenv("QN_LU_STRIP")) : 1;
    static const int la_fused = getenv("QN_LU_LA_FUSED") ? atoi(getenv("QN_LU_LA_FUSED")) : 1;
    bool p_ready = false; // the look-ahead update of the previous panel has already written this panel's buffer
    for (int p0 = 0, pi = 0; p0 < nlu; p0 += QN_NB, ++pi) {
        const int m = nlu - p0;
        bool in_p = false; // this panel was factorised in its column-major buffer (and, with the fused look-ahead, is not yet back in W)
        if (m <= QN_LU_PT * QN_LU_RPT && !s->newton_lu_percol) {
            // the panel in a column-major buffer, four columns at a time (qn_lu.hip.h: 19 launches instead of 128)
            double* P = s->newton_panel + (size_t)(pi & 1) * panel_doubles;
            const size_t pld = (size_t)QN_LU_PT * QN_LU_RPT;
            if (!p_ready) { hipLaunchKernelGGL(lu_panel_load_kernel, dim3(m / QN_NB), dim3(256), 0, st, W, ld, p0, P, pld, flag); launches++; }
            p_ready = false;
            in_p = true;
            const int rpt_p = (m + QN_LU_PT - 1) / QN_LU_PT;
            const int G = (persist && m >= lu_split_min) ? s->newton_lu_split : 1;
            if (G > 1) { // ... with role A split over G workgroups of one XCD (qn_lu_split.hip.h)
                const dim3 pg(2 + QN_NB - 2 * QN_LU_SUB + (G - 1)), pb(QN_LU_PT);
                const int base = 32 * pi;
                unsigned long long* rec = s->newton_rec + (size_t)(pi & 1) * QN_LUS_REC_WORDS;
                unsigned long long* rec_next = s->newton_rec + (size_t)((pi + 1) & 1) * QN_LUS_REC_WORDS;
                const int rptl = ((m + QN_LU_PT - 1) / QN_LU_PT + G - 1) / G; // rows per thread of a part
#define QN_LUS_LAUNCH(R, GG) hipLaunchKernelGGL((lu_panel_split_kernel<R, GG>), pg, pb, 0, st, P, pld, m, p0, s->newton_piv, flag, s->newton_sync, base, spin_max, rec, rec_next)
                if (G == 2) {
                    if (rptl <= 1) QN_LUS_LAUNCH(1, 2); else if (rptl <= 2) QN_LUS_LAUNCH(2, 2); else if (rptl <= 4) QN_LUS_LAUNCH(4, 2); else QN_LUS_LAUNCH(8, 2);
                } else {
                    if (rptl <= 1) QN_LUS_LAUNCH(1, 4); else if (rptl <= 2) QN_LUS_LAUNCH(2, 4); else QN_LUS_LAUNCH(4, 4);
                }
#undef QN_LUS_LAUNCH
                launches += 1;
            } else if (persist) { // the panel in one launch: 58 workgroups waiting for each other on counters (qn_lu.hip.h)
                const dim3 pg(2 + QN_NB - 2 * QN_LU_SUB), pb(QN_LU_PT);
                const int base = 32 * pi;
                if (rpt_p <= 1) hipLaunchKernelGGL(lu_panel_persist_kernel<1>, pg, pb, 0, st, P, pld, m, p0, s->newton_piv, flag, s->newton_sync, base, spin_max);
                else if (rpt_p <= 2) hipLaunchKernelGGL(lu_panel_persist_kernel<2>, pg, pb, 0, st, P, pld, m, p0, s->newton_piv, flag, s->newton_sync, base, spin_max);
                else if (rpt_p <= 4) hipLaunchKernelGGL(lu_panel_persist_kernel<4>, pg, pb, 0, st, P, pld, m, p0, s->newton_piv, flag, s->newton_sync, base, spin_max);
                else if (rpt_p <= 8) hipLaunchKernelGGL(lu_panel_persist_kernel<8>, pg, pb, 0, st, P, pld, m, p0, s->newton_piv, flag, s->newton_sync, base, spin_max);
                else hipLaunchKernelGGL(lu_panel_persist_kernel<QN_LU_RPT>, pg, pb, 0, st, P, pld, m, p0, s->newton_piv, flag, s->newton_sync, base, spin_max);
                launches += 1;
            } else
            for (int sp = 0; sp <= QN_NB / QN_LU_SUB; ++sp) {
                const int ncol_b = sp >= 1 ? std::max(0, QN_NB - QN_LU_SUB * (sp + 1)) : 0; // role B: the columns right of sub-panel sp
                const int grid = sp == 0 ? 1 : 2 + ncol_b;                                    // (workgroup 1: role C)
                const int rpt = (m + QN_LU_PT - 1) / QN_LU_PT; // rows per thread: the smallest instantiation that holds the panel
                if (rpt <= 1) hipLaunchKernelGGL(lu_panel_step_kernel<1>, dim3(grid), dim3(QN_LU_PT), 0, st, P, pld, m, sp, p0, s->newton_piv, flag);
                else if (rpt <= 2) hipLaunchKernelGGL(lu_panel_step_kernel<2>, dim3(grid), dim3(QN_LU_PT), 0, st, P, pld, m, sp, p0, s->newton_piv, flag);
                else if (rpt <= 4) hipLaunchKernelGGL(lu_panel_step_kernel<4>, dim3(grid), dim3(QN_LU_PT), 0, st, P, pld, m, sp, p0, s->newton_piv, flag);
                else if (rpt <= 8) hipLaunchKernelGGL(lu_panel_step_kernel<8>, dim3(grid), dim3(QN_LU_PT), 0, st, P, pld, m, sp, p0, s->newton_piv, flag);
                else hipLaunchKernelGGL(lu_panel_step_kernel<QN_LU_RPT>, dim3(grid), dim3(QN_LU_PT), 0, st, P, pld, m, sp, p0, s->newton_piv, flag);
            }
            if (!(la && la_fused)) hipLaunchKernelGGL(lu_panel_store_kernel, dim3(m / QN_NB), dim3(256), 0, st, W, ld, p0, P, pld, flag); // (else: on the second stream, below)
            launches += 1 + (persist ? 0 : 1 + QN_NB / QN_LU_SUB);
        } else {
            for (int k = p0; k < p0 + QN_NB; ++k) {
                hipLaunchKernelGGL(lu_pivot_kernel, dim3(1), dim3(1024), 0, st, W, ld, k, p0, nlu, s->newton_piv, flag);
                const int below = nlu - k - 1;
                if (below > 0)
                    hipLaunchKernelGGL(lu_col_step_kernel, dim3(std::min(1024, (below + 31) / 32)), dim3(256), 0, st, W, ld, k, p0, nlu, flag);
                launches += 2;
            }
        }
        const int right = nlu - p0 - QN_NB;
        if (!la) {
            if (nlu > QN_NB)
                hipLaunchKernelGGL(lu_swap_rows_kernel, dim3(std::min(256, (nlu + 255) / 256)), dim3(256), 0, st, W, ld, p0, nlu, s->newton_piv, flag);
            if (right > 0) {
                hipLaunchKernelGGL(lu_trsm_kernel, dim3((right + 255) / 256), dim3(256), 0, st, W, ld, p0, nlu, flag);
                hipLaunchKernelGGL(lu_gemm_kernel, dim3(right / QN_NB, right / QN_NB), dim3(256), 0, st, W, ld, p0, flag);
                launches += 2;
            }
            launches++;
            continue;
        }
        const int la_lo = p0 + QN_NB, la_hi = std::min(la_lo + QN_NB, nlu); // the next panel's columns
        const int below = nlu - la_lo;                                       // rows (and columns) right of / below this panel
        const bool fused = la_fused && in_p; // the chain's part reads the panel from its buffer; the copy back into W goes to the second stream
        double* Pc = s->newton_panel + (size_t)(pi & 1) * panel_doubles;
        double* Pn = s->newton_panel + (size_t)((pi + 1) & 1) * panel_doubles;
        const size_t pld_c = (size_t)QN_LU_PT * QN_LU_RPT;
        // the next panel's columns on this stream -- once the previous panel's bulk, which wrote them too, is through
        if (la_hi > la_lo) {
            if (last_f >= 0) HIPCHK(hipStreamWaitEvent(st, last_strip ? c->la_events[2 * npanels + last_f] : c->la_events[2 * last_f + 1], 0));
            if (fused) { // two launches, the second one leaving the next panel in its buffer (qn_lu.hip.h; the next panel is shorter: it fits)
                hipLaunchKernelGGL(lu_la_swap_trsm_kernel, dim3((la_hi - la_lo + 3) / 4), dim3(256), 0, st, W, ld, p0, la_lo, la_hi, Pc, pld_c, s->newton_piv, flag);
                hipLaunchKernelGGL(lu_la_gemm_kernel, dim3(below / QN_NB), dim3(256), 0, st, W, ld, p0, la_lo, Pc, pld_c, Pn, flag);
                p_ready = true;
                launches += 2;
            } else {
                hipLaunchKernelGGL(lu_swap_rows2_kernel, dim3(1), dim3(64), 0, st, W, ld, p0, la_lo, la_hi, s->newton_piv, flag);
                hipLaunchKernelGGL(lu_trsm2_kernel<1>, dim3((la_hi - la_lo + 3) / 4), dim3(256), 0, st, W, ld, p0, la_lo, la_hi, flag);
                hipLaunchKernelGGL(lu_gemm2_kernel, dim3(below / QN_NB), dim3(256), 0, st, W, ld, p0, la_lo, 1, below / QN_NB, flag, 1);
                launches += 3;
            }
        }
        // everything else beside the next panel's chain (the event behind the look-ahead launches, not in front of them: its packet and
        // the wait's were 13 us between the panel and the first look-ahead kernel.  For the tall panels, whose bulk is as long as the
        // next panel's chain, in front measured the same: 45.5 against 45.2 ms)
        HIPCHK(hipEventRecord(c->la_events[2 * pi], st));
        HIPCHK(hipStreamWaitEvent(c->stream_lu, c->la_events[2 * pi], 0));
        if (fused) { // the panel back into W, in front of the bulk that reads it there
            hipLaunchKernelGGL(lu_panel_store_kernel, dim3(m / QN_NB), dim3(256), 0, c->stream_lu, W, ld, p0, Pc, pld_c, flag);
            launches++;
        }
        if (p0 > 0) hipLaunchKernelGGL(lu_swap_rows2_kernel, dim3(std::min(64, (p0 + 255) / 256)), dim3(256), 0, c->stream_lu, W, ld, p0, 0, p0, s->newton_piv, flag);
        const int rest = nlu - la_hi;
        if (rest > 0) {
            hipLaunchKernelGGL(lu_swap_rows2_kernel, dim3(std::min(64, (rest + 255) / 256)), dim3(256), 0, c->stream_lu, W, ld, p0, la_hi, nlu, s->newton_piv, flag);
            hipLaunchKernelGGL(lu_trsm2_kernel<2>, dim3((rest + 7) / 8), dim3(256), 0, c->stream_lu, W, ld, p0, la_hi, nlu, flag);
            const int ncb = rest / QN_NB, ntiles = ncb * (below / QN_NB);
            static const int persist = getenv("QN_LU_BULK_PERSIST") ? atoi(getenv("QN_LU_BULK_PERSIST")) : 0; // (a resident grid that loops: 54.1 ms against 53.5)
            // (round 6) THE NEXT LOOK-AHEAD'S COLUMNS FIRST, in a launch of their own with an event behind it: the look-ahead kernels of panel pi + 1
            // touch the 64 columns right of la_hi and nothing else of what this update writes.  While the bulk sets the period (the first ~28 panels
            // at n = 8192 once the chain is split, qn_lu_split.hip.h) they no longer wait for the whole update -- the cycle was update, look-ahead
            // kernels (30 us), this stream's swaps and U12 solve (44 us), update -- and the look-ahead leaves the cycle.
            last_strip = false;
            if (lu_strip && ncb >= 2 && !persist) {
                const int nrt = below / QN_NB;
                hipLaunchKernelGGL(lu_gemm2_kernel, dim3(nrt), dim3(256), bulk_lds, c->stream_lu, W, ld, p0, la_hi, 1, nrt, flag, 0);
                HIPCHK(hipEventRecord(c->la_events[2 * npanels + pi], c->stream_lu));
                hipLaunchKernelGGL(lu_gemm2_kernel, dim3((ncb - 1) * nrt), dim3(256), bulk_lds, c->stream_lu, W, ld, p0, la_hi + QN_NB, ncb - 1, (ncb - 1) * nrt, flag, 0);
                last_strip = true;
                launches++;
            } else
            hipLaunchKernelGGL(lu_gemm2_kernel, dim3(persist ? std::min(ntiles, 2 * c->lu_bulk_cus) : ntiles), dim3(256), bulk_lds, c->stream_lu, W, ld, p0, la_hi, ncb,
                               ntiles, flag, 0);
            launches += 3;
        } else last_strip = false;
        HIPCHK(hipEventRecord(c->la_events[2 * pi + 1], c->stream_lu));
        last_f = pi;
        launches++;
    }
    if (last_f >= 0) HIPCHK(hipStreamWaitEvent(st, c->la_events[2 * last_f + 1], 0));
    HIPCHK(hipGetLastError());
    // row permutation: the swaps replayed on the identity (host; this path synchronises per Newton iteration anyway)
    int lu_failed = 0;
    HIPCHK(hipMemcpyAsync(s->newton_piv_host.data(), s->newton_piv, (size_t)nlu * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&lu_failed, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    s->stats.host_syncs++;
    s->stats.launches += launches;
    if (lu_failed == 2) { // a bounded wait of the one-launch panel gave up (its workgroups were not placed together): one launch per sub-panel for a while
        lu_note_timeout(s);
        return enqueue_newton_lu(s, hsrc, ld_src); // (the abandoned attempt's launches stay counted: they ran)
    }
    if (lu_failed) return QN_OK; // singular: the control kernel falls back to -g
    std::vector<int> perm((size_t)nlu);
    for (int i = 0; i < nlu; ++i) perm[i] = i;
    for (int k = 0; k < nlu; ++k) std::swap(perm[k], perm[s->newton_piv_host[k]]);
    HIPCHK(hipMemcpyAsync(s->newton_perm, perm.data(), (size_t)nlu * sizeof(int), hipMemcpyHostToDevice, st));
    double* x1 = s->newton_x;
    double* x2 = s->newton_x + n64;
    const dim3 vg(std::min(1024, (n64 + 255) / 256)), vb(256);
    int sweeps = 0;
    auto solve = [&](double* x, double* tmp) { // x <- U^-1 L^-1 x (x already permuted); tmp: scratch
        if (persist) { // a sweep in one launch: workgroups taking each other's solution blocks as they are published (qn_lu.hip.h)
            const int nb = nlu / QN_NB; // (`tmp` holds sentinels: lu_vec_perm_kernel; the forward sweep leaves them in `x`, the backward one in `tmp`)
            hipLaunchKernelGGL(lu_sweep_kernel<false>, dim3(nb), dim3(256), 0, st, W, ld, nb, x, tmp, flag, spin_max);
            hipLaunchKernelGGL(lu_sweep_kernel<true>, dim3(nb), dim3(256), 0, st, W, ld, nb, tmp, x, flag, spin_max);
            sweeps += 2;
            return;
        }
        for (int k0 = 0; k0 < nlu; k0 += QN_NB) {
            const int below = nlu - k0 - QN_NB;
            hipLaunchKernelGGL(lu_fwd_step_kernel, dim3(std::max(1, std::min(256, (below + 3) / 4))), dim3(256), 0, st, W, ld, k0, nlu, x, tmp);
        }
        for (int k0 = nlu - QN_NB; k0 >= 0; k0 -= QN_NB)
            hipLaunchKernelGGL(lu_bwd_step_kernel, dim3(std::max(1, std::min(256, (k0 + 3) / 4))), dim3(256), 0, st, W, ld, k0, tmp, x);
    };
    hipLaunchKernelGGL(lu_vec_perm_kernel, vg, vb, 0, st, x1, s->V.g, s->newton_perm, n, nlu, -1.0, persist ? x2 : nullptr); // P (-g)
    solve(x1, x2);
    hipLaunchKernelGGL(newton_vec_kernel, vg, vb, 0, st, s->V.d, x1, n, s->T.n_pad, 1.0); // d = -(H^-1 g)
    hipLaunchKernelGGL(lu_vec_perm_kernel, vg, vb, 0, st, x2, s->V.d, s->newton_perm, n, nlu, 1.0, persist ? x1 : nullptr); // P d
    solve(x2, x1);
    hipLaunchKernelGGL(newton_vec_kernel, vg, vb, 0, st, s->V.s, x2, n, s->T.n_pad, 1.0); // z = H^-1 d
    HIPCHK(hipGetLastError());
    s->stats.launches += 4 + (persist ? 4 : 4 * (uint64_t)(nlu / QN_NB));
    if (persist) HIPCHK(hipMemcpyAsync(&lu_failed, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st)); // `perm` is a local
    if (persist && lu_failed == 2) { // a bounded wait of a one-launch sweep gave up: the whole factorisation again, launch by launch
        lu_note_timeout(s);
        return enqueue_newton_lu(s, hsrc, ld_src);
    }
    return QN_OK;
}

static int enqueue_newton(qn_solver* s, const qn_oracle* o, qn_objective* obj) {
    qn_context* c = s->ctx;
    hipStream_t st = c->stream;
    QNCHK(newton_alloc(s));
    const int n = (int)s->n, n64 = (int)s->newton_n64;
    const size_t ld = s->newton_n64;
    // the Hessian at x_k: a device objective's own matrix, or the host closure's (uploaded)
    const double* hsrc = nullptr;
    size_t ld_src = 0;
    bool symmetric = true; // the Cholesky path reads the lower triangle only: it needs H == H' bit for bit
    if (obj) { hsrc = obj->Q; ld_src = (size_t)obj->T.n_pad; symmetric = obj->q_symmetric; }
    else {
        if (!s->newton_hsrc) HIPCHK(hipMalloc((void**)&s->newton_hsrc, (size_t)n * n * sizeof(double)));
        s->newton_hhost.resize((size_t)n * n * 2);
        HIPCHK(hipMemcpyAsync(s->hx, s->V.x, s->n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        double* hc = s->newton_hhost.data();           // column-major from the closure (DMatrix)
        double* hr = s->newton_hhost.data() + (size_t)n * n; // row-major for the device
        if (o->host_hessian_fn(o->host_user, s->hx, s->n, hc) != 0) return fail(QN_ABNORMAL_TERMINATION, "host Hessian callback failed");
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                hr[(size_t)i * n + j] = hc[i + (size_t)j * n];
                if (j > i && hc[i + (size_t)j * n] != hc[j + (size_t)i * n]) symmetric = false;
            }
        HIPCHK(hipMemcpyAsync(s->newton_hsrc, hr, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice, st));
        hsrc = s->newton_hsrc; ld_src = (size_t)n;
    }
    HIPCHK(hipMemsetAsync(s->newton_fail, 0, 2 * sizeof(int), st));
    if (s->hctl->small_n) { // reference-order arithmetic, one thread
        hipLaunchKernelGGL(newton_small_kernel, dim3(1), dim3(64), 0, st, hsrc, ld_src, n, s->V.g, s->V.d, s->V.s, s->newton_fail);
        HIPCHK(hipGetLastError());
        return QN_OK;
    }
    if (!symmetric || s->newton_force_lu) return enqueue_newton_lu(s, hsrc, ld_src);
    s->newton_chol_runs++;
    hipLaunchKernelGGL(newton_stage_kernel, dim3(2048), dim3(256), 0, st, s->newton_w, ld, n, n64, hsrc, ld_src, 1); // (the lower block triangle)
    // blocked right-looking Cholesky, lower triangle in place.  Outer blocks of 256 columns: each 64-column panel is
    // factorised and applied to the REST OF ITS OUTER BLOCK only; the trailing matrix then takes one depth-256 update.
    // (Look-ahead -- the next block's diag/panel chain on this stream beside the bulk update on a second, low-priority
    // stream -- was measured and dropped: the chain's single-workgroup kernels sat behind the bulk kernel's ~10^4
    // workgroups until it drained, 21 us -> 240-280 us each, and the iteration got 7 % slower.)
    const int KB = 4 * QN_NB;
    // LOOK-AHEAD (round 4).  The chain of an outer block -- 4 x (diagonal block, panel, in-block update): ~150 us of small, dependent
    // kernels -- needs only the block's own 256 columns up to date; the rest of the trailing matrix (up to 0.4 ms of MFMA work per
    // block at n = 8192) is needed one block later.  So the trailing update is cut in two: the next block's columns on the solver's
    // stream, the rest on a second stream, ordered by events (E_b: block b's panel columns are final; F_b: the bulk of block b is
    // done, awaited before the look-ahead columns of block b + 1 are touched again).  Round 1 measured this and dropped it: the
    // chain's one-workgroup kernels starved behind the bulk grid's 10^4 workgroups (21 us -> 240-280 us each).  What is different
    // now: the bulk launch asks for so much LDS that only QN_CHOL_BULK_WGS (2) of its workgroups fit a CU, which leaves wave slots,
    // registers and LDS on EVERY CU for the chain's workgroups the moment they are launched, and the chain's kernels raise their
    // waves' priority (s_setprio).  Measured at n = 8192 (tools/newton_time.py, tools/chol_timeline.py): 9.93-9.97 -> 9.34-9.41 ms per
    // Newton iteration.  The bulk keeps its pace (16.3 GFLOP in 360 us = 45 TFLOP/s for the first block), the chain's kernels take
    // twice their solo time beside it (diagonal block 22 -> 29-40 us, panel 7 -> 10-19, in-block update 9 -> 20-26): the first ten
    // blocks are bound by the bulk, the rest by the chain.  Also measured: the bulk stream restricted to 192-240 CUs
    // (hipExtStreamCreateWithCUMask; the chain's workgroups still land on busy CUs: no gain), 1, 3 and 4 bulk workgroups per CU.
    static const int la_on = getenv("QN_CHOL_LOOKAHEAD") ? atoi(getenv("QN_CHOL_LOOKAHEAD")) : 1;
    static const int bulk_wgs = getenv("QN_CHOL_BULK_WGS") ? std::max(1, atoi(getenv("QN_CHOL_BULK_WGS"))) : 2;
    const int nblocks = (n64 + KB - 1) / KB;
    const bool la = la_on && nblocks >= 4;
    static const int chol_masked = getenv("QN_CHOL_BULK_MASKED") ? atoi(getenv("QN_CHOL_BULK_MASKED")) : 0;
    if (la) {
        if (!c->stream2) HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        if (chol_masked) QNCHK(ensure_masked_stream(c));
        while ((int)c->la_events.size() < 2 * nblocks) { hipEvent_t e = nullptr; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->la_events.push_back(e); }
    }
    // (static LDS of chol_syrk_kernel: 2 x 32 x 65 doubles + the diagonal step's vectors = 35 KB; the CU has 160 KB: the dynamic part tops a workgroup up to 160 / bulk_wgs)
    size_t bulk_lds = (size_t)std::max(0, (160 * 1024) / bulk_wgs - 36 * 1024);
    if (la && bulk_lds > 0) { // (more than the default 64 KB per workgroup needs the attribute; refused: run the bulk without the cap)
        static std::atomic<int> attr_state[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && attr_state[dev].load() == 0) {
            const bool ok = hipFuncSetAttribute((const void*)chol_syrk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bulk_lds) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            attr_state[dev].store(ok ? 1 : 2);
        }
        if (dev < 0 || dev >= 64 || attr_state[dev].load() != 1) bulk_lds = 0;
    }
    int last_f = -1;
    static const int chol_fuse = getenv("QN_CHOL_FUSE_DIAG") ? atoi(getenv("QN_CHOL_FUSE_DIAG")) : 1;
    static const int chol_left = getenv("QN_CHOL_LEFT") ? atoi(getenv("QN_CHOL_LEFT")) : 0;
    bool diag_done = false;
    hipStream_t bulk_st = (la && chol_masked) ? c->stream_lu : c->stream2;
    for (int K0 = 0, b = 0; K0 < n64; K0 += KB, ++b) {
        const int Kend = std::min(K0 + KB, n64);
        for (int k0 = K0; k0 < Kend; k0 += QN_NB) {
            double* invl = s->newton_invl + (size_t)(k0 / QN_NB) * QN_NB * QN_NB;
            // (the diagonal block's factor and inverse: a launch of its own for the first block only -- afterwards the update that
            // produced the block went on to factorise it, chol_syrk_kernel's invL_next)
            if (!diag_done) { hipLaunchKernelGGL(chol_diag_inv_kernel, dim3(1), dim3(256), 0, st, s->newton_w, ld, k0, invl, s->newton_fail); s->stats.launches++; }
            diag_done = false;
            const int nrt = (n64 - k0 - QN_NB) / QN_NB; // row tiles below the diagonal block
            if (nrt > 0) hipLaunchKernelGGL(chol_panel_kernel, dim3(nrt), dim3(256), 0, st, s->newton_w, ld, k0, invl, s->newton_fail);
            const int nct = (Kend - k0 - QN_NB) / QN_NB; // column tiles left in this outer block
            if (nrt > 0 && nct > 0) {
                hipLaunchKernelGGL(chol_syrk_kernel, dim3(qn_tri_tiles(nrt, nct)), dim3(256), 0, st, s->newton_w, ld, k0, QN_NB, k0 + QN_NB, nct, s->newton_fail, 1,
                                   chol_fuse ? invl + QN_NB * QN_NB : nullptr);
                diag_done = chol_fuse;
            }
            s->stats.launches += 2;
        }
        const int nt = (n64 - Kend) / QN_NB;
        if (nt > 0 && la && chol_left) {
            // LEFT-LOOKING second stream (round 4; measured, NOT the default: QN_CHOL_LEFT=1).  With the right-looking bulk -- block b's panel
            // applied to everything right of the next block, beside chain b + 1 -- the first ten outer blocks are bound by the bulk (345 us
            // of MFMA work against a 240 us chain) and the last twenty by the chain with the second stream nearly idle: 9.1 ms where the
            // chain alone is ~5.5.  The same flops in another order: what block column b + 2 owes to ALL the panels so far (0 .. b) as ONE
            // update of depth 256 (b + 1), launched on the second stream as soon as chain b is through and awaited a whole chain period
            // later, before this stream adds panel b + 1's part.  Every tile is then read and written once, and the chain never waits for
            // more than four tile columns.  Measured: 10.7 ms against 9.1 -- the deep, narrow updates of the last third (10-150 tiles of
            // depth 5000-7700: one workgroup per CU, each 32-deep chunk a global-load round trip nothing hides: 1.2 us against 0.43 us of
            // MFMA work) take 220-380 us where a chain period is 170, and in the middle third a panel kernel of the chain was seen
            // waiting 150 us for CUs beside them (profiles/r04_j_*).  What it needs is an update kernel that is efficient at one workgroup
            // per CU (deeper prefetch, 64 x 128 tiles: a 64 x 64 tile at full MFMA rate asks a CU for 77 KB/us, more than it takes in).
            const int nla = std::min(KB / QN_NB, nt);
            if (nt > nla) { // J_{b+2}: block column b + 2 (tile columns nla .. 2 nla - 1 right of Kend) -= panels [0, Kend) ...
                const int ncj = std::min(KB / QN_NB, nt - nla);
                HIPCHK(hipEventRecord(c->la_events[2 * b], st));
                HIPCHK(hipStreamWaitEvent(bulk_st, c->la_events[2 * b], 0));
                hipLaunchKernelGGL(chol_syrk_kernel, dim3(qn_tri_tiles(nt - nla, ncj)), dim3(256), bulk_lds, bulk_st, s->newton_w, ld, 0, Kend, Kend + nla * QN_NB, ncj,
                                   s->newton_fail, 0);
                HIPCHK(hipEventRecord(c->la_events[2 * b + 1], bulk_st));
                s->stats.launches++;
            }
            // ... and block column b + 1 -= panel b on this stream, once J_{b+1} (launched a chain period ago) has brought it up to panel b - 1
            if (last_f >= 0) HIPCHK(hipStreamWaitEvent(st, c->la_events[2 * last_f + 1], 0));
            last_f = nt > nla ? b : -1;
            hipLaunchKernelGGL(chol_syrk_kernel, dim3(qn_tri_tiles(nt, nla)), dim3(256), 0, st, s->newton_w, ld, K0, Kend - K0, Kend, nla, s->newton_fail, 1,
                               chol_fuse ? s->newton_invl + (size_t)(Kend / QN_NB) * QN_NB * QN_NB : nullptr);
            diag_done = chol_fuse;
            s->stats.launches++;
        } else if (nt > 0) {
            const int nla = la ? std::min(KB / QN_NB, nt) : nt; // tile columns of the next outer block
            // Round 5: in the FIRST THIRD of the outer blocks -- where the bulk, not the chain, sets the pace (tools/chol_timeline.py: periods of
            // 470 ... 250 us against a chain of 240) -- the bulk may start as soon as this block's panels are final, BESIDE the look-ahead
            // columns' update instead of behind it: the two touch different columns.  The second stream then never idles there.  Not later:
            // where the chain is the pace, the look-ahead update is a link of it and runs slower beside a bulk.  Measured at n = 8192, alternating:
            // 9.00-9.14 -> 8.92-8.98 ms per Newton iteration (QN_CHOL_EARLY_BULK = 0 / 6 / 10 / 14 / 32 blocks: 9.05 / 8.96 / 8.95 / 8.96 / 9.05).
            // The gain is small because the first third does MFMA work back to back either way: what would shorten it is moving flops into the last
            // two thirds, where the second stream is mostly idle -- the left-looking order, which needs an update kernel that is efficient on
            // deep, narrow updates (see above).
            // Also measured and dropped in round 5 (same tool, alternating): the look-ahead columns in TWO launches -- the first 64-column strip,
            // which the chain's next link needs, on this stream, strips 1..3 on a third stream beside it, awaited in front of the next block's first
            // in-block update: the strip-0 launch is 42 us instead of 60, but two more event pairs sit in the chain (7-8 us each) and the first
            // in-block update still runs beside the freshly started bulk at twice its solo time: 8.9-9.0 -> 9.2-9.4 ms; and the bulk at ONE
            // workgroup per CU in the chain-paced blocks: 9.03-9.06 ms either way.
            static const int chol_early_env = getenv("QN_CHOL_EARLY_BULK") ? atoi(getenv("QN_CHOL_EARLY_BULK")) : -1;
            const int chol_early = chol_early_env >= 0 ? chol_early_env : nblocks / 3;
            const bool early = la && nt > nla && b < chol_early;
            if (early) {
                HIPCHK(hipEventRecord(c->la_events[2 * b], st));
                HIPCHK(hipStreamWaitEvent(bulk_st, c->la_events[2 * b], 0));
            }
            // the next block's columns on this stream -- once the PREVIOUS bulk, which wrote them too, is through -- ...
            if (la && last_f >= 0) HIPCHK(hipStreamWaitEvent(st, c->la_events[2 * last_f + 1], 0));
            hipLaunchKernelGGL(chol_syrk_kernel, dim3(qn_tri_tiles(nt, nla)), dim3(256), 0, st, s->newton_w, ld, K0, Kend - K0, Kend, nla, s->newton_fail, la ? 1 : 0,
                               chol_fuse ? s->newton_invl + (size_t)(Kend / QN_NB) * QN_NB * QN_NB : nullptr);
            diag_done = chol_fuse;
            s->stats.launches++;
            // ... then the bulk, beside the next block's chain.  (Launched BEFORE the look-ahead columns -- it needs only this block's
            // panel -- it measured slower: 9.43-9.64 ms per Newton iteration against 9.34-9.41, three alternating runs; the chain of
            // the next block then runs under contention from its first kernel on.)
            if (nt > nla) {
                if (!early) {
                    HIPCHK(hipEventRecord(c->la_events[2 * b], st));
                    HIPCHK(hipStreamWaitEvent(bulk_st, c->la_events[2 * b], 0));
                }
                hipLaunchKernelGGL(chol_syrk_kernel, dim3(qn_tri_tiles(nt - nla, nt - nla)), dim3(256), bulk_lds, bulk_st, s->newton_w, ld, K0, Kend - K0,
                                   Kend + nla * QN_NB, nt - nla, s->newton_fail, 0);
                s->stats.launches++;
            }
            if (nt > nla) { HIPCHK(hipEventRecord(c->la_events[2 * b + 1], bulk_st)); last_f = b; }
        }
    }
    if (last_f >= 0) HIPCHK(hipStreamWaitEvent(st, c->la_events[2 * last_f + 1], 0));
    HIPCHK(hipGetLastError());
    if (s->newton_big) QNCHK(newton_build_block_inverses(s));
    // d = -(H^-1 g) ; z = H^-1 d
    double* x1 = s->newton_x;
    double* x2 = s->newton_x + n64;
    const dim3 vg(std::min(1024, (n64 + 255) / 256)), vb(256);
    hipLaunchKernelGGL(newton_vec_kernel, vg, vb, 0, st, x1, s->V.g, n, n64, -1.0);
    QNCHK(newton_tri_solve(s, x1, x2));
    hipLaunchKernelGGL(newton_vec_kernel, vg, vb, 0, st, s->V.d, x1, n, s->T.n_pad, 1.0);
    QNCHK(newton_tri_solve(s, x1, x2)); // the first solve's result is the second's right-hand side
    hipLaunchKernelGGL(newton_vec_kernel, vg, vb, 0, st, s->V.s, x1, n, s->T.n_pad, 1.0);
    HIPCHK(hipGetLastError());
    s->stats.launches += 4 + (s->newton_big ? 0 : 4 * (uint64_t)(n64 / QN_NB));
    // Not positive definite?  The reference's LU inverts any non-singular matrix (newton/mod.rs:36-41): take the pivoted-LU path.
    // (The flag is read here, after everything was enqueued, so the convex case keeps its launch pipeline; the caller
    // synchronises right after this function anyway.)
    int chol_failed = 0;
    HIPCHK(hipMemcpyAsync(&chol_failed, s->newton_fail, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    s->stats.host_syncs++;
    if (chol_failed) return enqueue_newton_lu(s, hsrc, ld_src);
    return QN_OK;
}
