// qn_host_solver.hip.h -- host side, part 4 of 7: qn_solver -- state, allocations, options, getters / setters, the trait-hook entry points
// (compute_direction, secant_update), profiling brackets.
#pragma once
// ------------------------------------------------------------------------------------------------
// solver
// ------------------------------------------------------------------------------------------------
enum { KC_HPASS = 0, KC_EVAL = 1, KC_CTL = 2, KC_COMM = 3, KC_HREDUCE = 4, KC_EREDUCE = 5, KC_NEWTON = 6 };
struct TimedEvent { hipEvent_t a, b; int cls; };

struct qn_solver {
    qn_context* ctx = nullptr;
    int method = QN_BFGS;
    size_t n = 0;
    double tol = 0.0;
    QnTile T{};
    int R = 4, U = 1, hcs = 1, qcs = 1; // row tile, chunks per trip (fused kernels), column splits
    double* H = nullptr;
    double* vec_block = nullptr; // one allocation holding all n_pad vectors
    QnVecs V{};
    double* f_dev = nullptr;
    // Newton: Hessian work matrix (row-major, ld = nw), right-hand sides, staging for host Hessians, failure flag
    double *newton_w = nullptr, *newton_x = nullptr, *newton_hsrc = nullptr, *newton_invl = nullptr, *newton_inv2 = nullptr;
    bool newton_big = false;
    // symmetric-storage fast path (qn_sym.hip.h): slot buffer, tile count per side, opt-out, "user installed a non-symmetric H"
    double* sym_part = nullptr;
    int sym_nb = 0;
    bool no_sym = false, h_nonsym = false;
    bool h_lower_stale = false; // a symmetric-storage run is (or was) updating the upper block triangle only
    double *symsh_xg = nullptr, *symsh_gath = nullptr; // row-sharded symmetric storage: gathered partial sums [world][2][n_pad]; mirror staging
    bool h_diag_stale = false;  // ... and (second-generation kernels) only the upper triangle of 16 x 16 sub-blocks inside the diagonal tiles
    // second-generation symmetric path (qn_sym2.hip.h): static work lists, per-workgroup scalars, double-buffered control block
    int* s2_items = nullptr;
    int s2_G = 0, s2_nb = 0, s2_maxk = 0, s2_inorder = 0;
    int s2_sl_first = 0, s2_sl_per = 0, s2_sl_cfg = -1; // row slivers (QnS2Args.sl_first / sl_per); the switches the lists were built for
    bool no_sliver = false;    // diagnostics: sym2 without row slivers (round 2's work lists)
    bool no_pair = false;      // diagnostics: the general evaluation kernel where the two-items-and-a-sliver instance would run
    int ring = (getenv("QN_S2_RING") && atoi(getenv("QN_S2_RING")) == 0) ? 0 : 1; // the pair instance's evaluation as mover + multiplier waves (qn_sym2r.hip.h); QN_OPT_EVAL_MOVER_MULTIPLIER
    int zig = (getenv("QN_S2_ZIGZAG") && atoi(getenv("QN_S2_ZIGZAG")) == 0) ? 0 : 1; // ... its two tiles in the other order in launches of odd parity (the L2 across evaluation launches); QN_OPT_EVAL_ZIGZAG
    int touch = getenv("QN_S2_TOUCH") ? atoi(getenv("QN_S2_TOUCH")) : 8;    // TOUCH workgroups in the accept-reduce: rows per wave of H's tiles (0, 4, 6, 8, 10, 12, 16); QN_OPT_TOUCH_H_ROWS
    int touchq = getenv("QN_S2_TOUCHQ") ? atoi(getenv("QN_S2_TOUCHQ")) : 6; // ... in the update-reduce: rows of Q's tiles; QN_OPT_TOUCH_Q_ROWS
    int touch_delay = getenv("QN_S2_TOUCH_DELAY") ? atoi(getenv("QN_S2_TOUCH_DELAY")) : 32; // ... units of 64 clocks the accept-reduce's touching workgroups sleep first
    int touchq_delay = getenv("QN_S2_TOUCHQ_DELAY") ? atoi(getenv("QN_S2_TOUCHQ_DELAY")) : 0; // ... and the update-reduce's
    bool tred = false;         // measurement: the update-reduce in the tail of the update-tile launch (s2_hpass_kernel<.., TRED>: bit-identical, slower)
    int* s2_cnt = nullptr;     // tail reduce: arrival counters of the block-rows
    int gen_slots_hint = 0;    // generic pipelined path: evaluation slots per period the last batch needed (0: none run yet)
    double* s2_gws = nullptr;  // row-sharded log-sum-exp (qn_sym2g.hip.h): the ranks' weights and S of the last evaluation consumed
    double* s2_wgV = nullptr;  // generic objectives: the second table of per-workgroup sums (QnS2Args.wgV)
    hipGraphExec_t s2_graph_exec = nullptr; // measurement (QN_S2_GRAPH): two periods of the pipelined pattern as one graph, and what it was captured for
    QnS2Args s2_graph_args{}; int s2_graph_slots = 0, s2_graph_bnd = 0; uint64_t s2_graph_len = 0, s2_graph_stat = 0;
    int last_ls_kind = -1; std::vector<double> last_ls_box; bool ls_box_changed = false; // the line search (kind, box) of the last qn_minimize call: see minimize_impl
    double mtb_cand_keep = INFINITY; // bounded second-generation runs: the step to the box of the direction a warm call continues with
    bool no_s2bnd = false;     // tests: bounded runs keep the generic path (QN_OPT_BOUNDED_SECOND_GENERATION 0)
    bool h_sliver_whole = false; // the diagonal tiles that sliver rows read are complete (both triangles): kept so by sliver-mode update passes
    double* s2_evS = nullptr;   // row-sharded: [2][world][QN_S2SH_NEC][QN_S2_MAXG] the ranks' evaluation scalars, by launch parity (QnS2Args.evS)
    int *s2_sl_off = nullptr, *s2_sl_idx = nullptr; // row-sharded: per block-row, the slots this rank's tiles write (QnS2Args.sl_off / sl_idx)
    int s2_sl_nb = 0;
    bool h_placed = false;      // H's placement has been measured (place_h)
    int* symsh_tiles = nullptr; // row-sharded, first-generation kernels: this rank's tiles in launch order (QnSymShard.tiles)
    int s2_slots_hint = 0;      // row-sharded, pipelined: evaluation launches (each followed by a collective) enqueued per period
    double* s2_partE = nullptr; // [nb][nb][128]: row / column slots of the last evaluation (QnS2Args.partE)
    double* s2_wgS = nullptr; // [2][s2_trows][QN_S2_ROW]: the sums a servicing launch leaves for the next launch's prologue, by launch parity
    int s2_trows = 0;
    QnCtl* s2_ctl = nullptr;
    bool no_sym2 = false; // diagnostics: the first-generation tile kernels (qn_sym.hip.h)
    bool fold = false; // sym2 with the accept-reduce folded into the update-tile launch (four launches per iteration instead of five).
                       // OFF by default -- measured, rocprofv3 averages, n = 4096, same box: the accept-reduce launch (6.7 us) goes, the
                       // update-tile launch gets 5 us longer (every workgroup sums the slots of its own six blocks: 48 MB of L2 reads
                       // instead of 1 MB, and the prologue there is the long run of the machine) and the update-reduce 2.4 us (its
                       // prologue now runs the machine): 80.3 against 79.4 us per iteration.
    // After a fused run the iterate and the pending update's vectors stay where the fused kernels keep them (X0[xc], S0[sc], UN);
    // they are copied back to the canonical buffers only when something other than the next fused run wants them.
    bool fused_live = false;
    // the previous qn_minimize ended on the iteration cap of a fused, memoised run on the objective with this serial (0: none);
    // nothing has touched the state since.  A serial, not the pointer: destroy A, create B of the same size and the allocator
    // hands the address back -- the run on B would have inherited f, g and the lazy direction of A.
    uint64_t warm_obj = 0;
    int* newton_fail = nullptr; // [0]: the factorisation met a bad pivot, [1]: the staged Hessian is not symmetric bit for bit
    int *newton_piv = nullptr, *newton_perm = nullptr; // LU fallback (qn_lu.hip.h): pivot rows, row permutation
    double* newton_panel = nullptr; // ... and the column-major copy of the 64-column panel being factorised
    bool newton_lu_percol = false;  // diagnostics: the panel factorisation with two launches per column (rounds 1-2)
    std::vector<int> newton_piv_host;
    uint64_t newton_lu_runs = 0, newton_chol_runs = 0;
    int newton_lu_force_timeout = 0; // diagnostics (QN_OPT_LU_FORCE_WAIT_EXPIRY): the one-launch kernels' waits give up at once (exercises the fallback)
    int newton_lu_no_persist = 0; // diagnostics (QN_OPT_LU_ONE_LAUNCH_PANEL 0), or set after a bounded wait of the one-launch panel gave up: one launch per sub-panel
    int* newton_sync = nullptr;   // the one-launch panel's counters (qn_lu.hip.h, lu_panel_persist_kernel)
    unsigned long long* newton_rec = nullptr; // ... and, with role A split over workgroups, the parts' records (qn_lu_split.hip.h)
    bool no_projfold = getenv("QN_S2_PROJ_FOLD") && atoi(getenv("QN_S2_PROJ_FOLD")) == 0; // QN_OPT_BTB_PROJECT_IN_EVAL 0: BackTrackingB's projection as a launch per trial
    int newton_lu_split = getenv("QN_LU_SPLIT") ? atoi(getenv("QN_LU_SPLIT")) : 4; // parts of role A: 1 (one workgroup, rounds 4-5), 2 or 4 (QN_OPT_LU_SPLIT_ROLE_A)
    int newton_lu_split_min = getenv("QN_LU_SPLIT_MIN") ? atoi(getenv("QN_LU_SPLIT_MIN")) : 4160; // ... for panels of at least this many rows (QN_OPT_LU_SPLIT_MIN_ROWS)
    uint64_t newton_lu_sync_timeouts = 0;
    int newton_lu_timeout_fallback = 0; // newton_lu_no_persist was set by an expired wait (not by the diagnostics switch): how many factorisations have run launch by launch since
    int newton_lu_no_la = 0; // diagnostics (QN_OPT_LU_LOOKAHEAD 0): the LU without the look-ahead on a second stream
    int newton_force_lu = 0; // diagnostics (QN_OPT_NEWTON_PIVOTED_LU): skip the Cholesky attempt
    size_t newton_n64 = 0;
    std::vector<double> newton_hhost;
    double* bounds_block = nullptr; // lb, ub (solver), llb, lub (bounded line search): 4 n_pad vectors
    int bounded = 0;
    double* fused_block = nullptr; // X0[2], S0[2], G, GT, Y, UN, UP, VV (10 n_pad vectors)
    double *fused_evp = nullptr, *fused_hpp = nullptr;
    int fused_nblk = 0;
    QnCtl* ctl = nullptr;  // device
    QnCtl* hctl = nullptr; // pinned host mirror
    QnCtl* hrep = nullptr; // pinned: the control block as the last launch of a sym2 batch left it (written by the device)
    unsigned long long* hrep_flag = nullptr;
    unsigned long long rep_seq = 0;
    double *hx = nullptr, *hg = nullptr; // pinned staging for host oracles
    size_t trace_cap = 0;
    int trace_x = 0;
    int sync_mode = -1; // -1 auto
    int no_fused = 0;   // diagnostics: force the generic (non-fused) path
    int no_defer = 0;   // diagnostics: fused path without the deferred update step
    int profiling = 0;
    std::vector<TimedEvent> events;
    std::vector<hipEvent_t> event_pool;
    qn_stats stats{};
};

static hipEvent_t ev_get(qn_solver* s) {
    if (!s->event_pool.empty()) { hipEvent_t e = s->event_pool.back(); s->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope { // brackets one launch (or one exchange) with events when profiling is on
    qn_solver* s; int cls; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(qn_solver* s_, int cls_) : s(s_), cls(cls_) {
        if (s->profiling && s->events.size() < 200000) { a = ev_get(s); b = ev_get(s); (void)hipEventRecord(a, s->ctx->stream); }
    }
    ~ProfScope() { if (a) { (void)hipEventRecord(b, s->ctx->stream); s->events.push_back({a, b, cls}); } }
};
static void prof_collect(qn_solver* s) {
    if (s->events.empty()) return;
    (void)hipStreamSynchronize(s->ctx->stream);
    // In pipelined mode the launch pattern runs ahead of the decisions, so some bracketed launches found their request not
    // pending and returned from the prologue (a few microseconds).  They are not work: a class's sums take only the launches
    // that lasted more than half of one of its LONGEST launches (in synchronous mode every launch is real and passes; a median-based
    // floor failed when most of a class's launches were skipped -- backtracking's four slots per period, a run that ends early in a
    // batch).  "One of the longest" = the (n / 50 + 1)-th longest: a single outlier twice the typical duration (a cold first launch,
    // a co-tenant's preemption) would otherwise set a floor that discards every genuine launch.
    std::vector<std::vector<float>> dur(8);
    std::vector<std::pair<int, float>> all;
    all.reserve(s->events.size());
    for (auto& e : s->events) {
        float ms = 0.f;
        const bool ok = hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess;
        const int cls = (e.cls >= 0 && e.cls < 7) ? e.cls : 7;
        if (ok) { dur[cls].push_back(ms); all.push_back({e.cls, ms}); }
        s->event_pool.push_back(e.a);
        s->event_pool.push_back(e.b);
    }
    float floor_ms[8];
    for (int c = 0; c < 8; ++c) {
        floor_ms[c] = 0.f;
        if (dur[c].size() >= 8) {
            const size_t kth = dur[c].size() / 50; // 0: the maximum
            std::nth_element(dur[c].begin(), dur[c].begin() + kth, dur[c].end(), std::greater<float>());
            floor_ms[c] = 0.5f * dur[c][kth];
        }
    }
    for (auto& e : all) {
        const int cls = (e.first >= 0 && e.first < 7) ? e.first : 7;
        const float ms = e.second;
        if (ms < floor_ms[cls]) continue;
        switch (e.first) {
        case KC_HPASS: s->stats.t_hpass_ms += ms; s->stats.n_hpass_timed++; break;
        case KC_EVAL: s->stats.t_eval_ms += ms; s->stats.n_eval_timed++; break;
        case KC_CTL: s->stats.t_ctl_ms += ms; s->stats.n_ctl_timed++; break;
        case KC_HREDUCE: s->stats.t_hreduce_ms += ms; s->stats.n_hreduce_timed++; break;
        case KC_EREDUCE: s->stats.t_ereduce_ms += ms; s->stats.n_ereduce_timed++; break;
        case KC_NEWTON: s->stats.t_newton_ms += ms; s->stats.n_newton_timed++; break;
        default: s->stats.t_comm_ms += ms; s->stats.n_comm_timed++; break;
        }
    }
    s->events.clear();
}

// buffers of the fused fast path (allocated on first use; partial buffers depend on the row tile R)
static int solver_alloc_fused(qn_solver* s, bool sym) {
    const size_t np = s->T.n_pad;
    hipStream_t st = s->ctx->stream;
    if (!s->fused_block) QNCHK(dev_alloc_zero(&s->fused_block, 10 * np, st));
    const int nblk = sym ? (int)(np / QN_TB) : s->T.rpr / s->R; // partial-sum rows: 128-row blocks or R-row workgroups
    if (sym && s->sym_nb != nblk) {
        if (s->sym_part) { HIPCHK(hipFree(s->sym_part)); s->sym_part = nullptr; }
        QNCHK(dev_alloc_zero(&s->sym_part, (size_t)nblk * nblk * 2 * QN_TB, st));
        s->sym_nb = nblk;
    }
    if (nblk != s->fused_nblk) {
        if (s->fused_evp) { HIPCHK(hipFree(s->fused_evp)); s->fused_evp = nullptr; }
        if (s->fused_hpp) { HIPCHK(hipFree(s->fused_hpp)); s->fused_hpp = nullptr; }
        QNCHK(dev_alloc_zero(&s->fused_evp, (size_t)s->ctx->world * QN_NEVP * nblk, st));
        QNCHK(dev_alloc_zero(&s->fused_hpp, (size_t)s->ctx->world * QN_NHPP * nblk, st));
        s->fused_nblk = nblk;
    }
    double* p = s->fused_block;
    QnFused& F = s->V.F;
    F.X0 = p; F.S0 = p + 2 * np; F.G = p + 4 * np; F.GT = p + 5 * np; F.Y = p + 6 * np;
    F.UN = p + 7 * np; F.UP = p + 8 * np; F.VV = p + 9 * np;
    F.evp = s->fused_evp; F.hpp = s->fused_hpp;
    F.nblk = nblk;
    return QN_OK;
}

// Row-sharded symmetric storage (both generations of kernels): per block-row R, the slots this rank's tiles write -- R's own
// window (row parts; a diagonal tile's single slot) and the column parts of the local block-rows whose windows contain R.  Every
// unordered pair of block-rows is owned once, so no slot appears twice; ascending order = the summation order of the rank's share.
static int solver_alloc_symsh_lists(qn_solver* s) {
    const int nb = s->T.n_pad / QN_TB;
    if (s->s2_sl_off && s->s2_sl_nb == nb) return QN_OK;
    (void)hipFree(s->s2_sl_off); (void)hipFree(s->s2_sl_idx); (void)hipFree(s->symsh_tiles);
    s->s2_sl_off = nullptr; s->s2_sl_idx = nullptr; s->symsh_tiles = nullptr;
    const int nbl = s->T.rpr / QN_TB, ioff = s->ctx->rank * nbl;
    {
        std::vector<int> tiles;
        for (int il = 0; il < nbl; ++il)
            for (int k = 0, I = ioff + il; k < qn_symsh_cnt(I, nb); ++k) tiles.push_back((I << 16) | ((I + k) % nb));
        if (tiles.empty()) tiles.push_back(0);
        HIPCHK(hipMalloc((void**)&s->symsh_tiles, tiles.size() * sizeof(int)));
        HIPCHK(hipMemcpy(s->symsh_tiles, tiles.data(), tiles.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    std::vector<int> off(nb + 1, 0), idx;
    for (int R = 0; R < nb; ++R) {
        const bool r_local = R >= ioff && R < ioff + nbl;
        for (int t = 0; t < nb; ++t) {
            const bool t_local = t >= ioff && t < ioff + nbl;
            if ((r_local && qn_symsh_owns(R, t, nb)) || (t_local && t != R && qn_symsh_owns(t, R, nb))) idx.push_back(t);
        }
        off[R + 1] = (int)idx.size();
    }
    if (idx.empty()) idx.push_back(0);
    HIPCHK(hipMalloc((void**)&s->s2_sl_off, off.size() * sizeof(int)));
    HIPCHK(hipMalloc((void**)&s->s2_sl_idx, idx.size() * sizeof(int)));
    HIPCHK(hipMemcpy(s->s2_sl_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->s2_sl_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    s->s2_sl_nb = nb;
    return QN_OK;
}

// sym2: work items (off-diagonal tiles cost 1, diagonal tiles -- upper triangle only -- 0.5625), assigned to min(items, 256)
// workgroups by longest-processing-time-first so that every workgroup streams the same number of bytes to within one tile.
// Row-sharded runs: the items are the tiles of the rank's circulant windows (qn_sym.hip.h), the grid is the same on every rank
// (the ranks' per-workgroup scalars are exchanged as rows of one table), and the rank gets the list of slots its tiles write.
static int solver_alloc_sym2(qn_solver* s) {
    const int nb = s->T.n_pad / QN_TB;
    hipStream_t st = s->ctx->stream;
    const int world = s->ctx->world, rank = s->ctx->rank;
    const bool sharded = world > 1;
    const int cfg = (s->fold ? 1 : 0) | (s->no_sliver ? 2 : 0) | (sharded ? 4 : 0);
    if (s->s2_nb != nb || s->s2_sl_cfg != cfg) {
        (void)hipFree(s->s2_items); (void)hipFree(s->s2_wgS); (void)hipFree(s->s2_partE);
        if (s->s2_graph_exec) { (void)hipGraphExecDestroy(s->s2_graph_exec); s->s2_graph_exec = nullptr; } // (the pointer dangled -- a matching memcmp launched it again)
    (void)hipFree(s->s2_evS); (void)hipFree(s->s2_cnt);
        s->s2_items = nullptr; s->s2_wgS = nullptr; s->s2_partE = nullptr;
        s->s2_evS = nullptr; s->s2_cnt = nullptr;
        s->s2_nb = 0;
        const int nbl = s->T.rpr / QN_TB, ioff = rank * nbl; // (sharded: this rank's block-rows)
        int nitems = nb * (nb + 1) / 2;
        int G = std::min(nitems, QN_S2_MAXG);
        if (sharded) {
            // every rank launches the same grid: the smallest share of tiles bounds it (each workgroup has at least one item)
            nitems = qn_symsh_ntiles(nb, nbl, ioff);
            int least = nitems;
            for (int r = 0; r < world; ++r) least = std::min(least, qn_symsh_ntiles(nb, nbl, r * nbl));
            G = std::min(least, QN_S2_MAXG);
            if (G < 1) return fail(QN_ABNORMAL_TERMINATION, "sym2: a rank without tiles");
        }
        std::vector<std::vector<int>> lists;
        int inorder = 0;
        // deals the items to G lists; L: the last L diagonal tiles stay off the lists (row slivers)
        auto deal = [&](int L) {
            lists.assign(G, std::vector<int>());
            std::vector<std::pair<double, int>> heap; // (-load, workgroup): max-heap on the least loaded
            for (int g = 0; g < G; ++g) heap.push_back({0.0, -g});
            std::make_heap(heap.begin(), heap.end());
            auto give = [&](int I, int J, double cost) {
                std::pop_heap(heap.begin(), heap.end());
                auto e = heap.back();
                lists[-e.second].push_back((I << 16) | J);
                e.first -= cost;
                heap.back() = e;
                std::push_heap(heap.begin(), heap.end());
            };
            if (sharded) { // the windows' off-diagonal tiles in window order, then the diagonal ones (the cheap items last)
                inorder = 0; // (the kernels read every item from the list: qn_s2_first_item_of)
                for (int il = 0; il < nbl; ++il) {
                    const int I = ioff + il, cnt = qn_symsh_cnt(I, nb);
                    for (int k = 1; k < cnt; ++k) give(I, (I + k) % nb, 1.0);
                }
                for (int il = 0; il < nbl; ++il) give(ioff + il, ioff + il, 0.5625);
                return;
            }
            // the first min(2 G, items) items go out in order -- item t to workgroup t mod G -- so the kernels compute a workgroup's
            // first two items from its index (qn_s2_item_of_index); the rest to whoever has streamed least so far
            inorder = std::min(nitems - L, 2 * G);
            std::vector<double> load0(G, 0.0);
            int handed = 0;
            auto hand = [&](int I, int J, double cost) {
                if (handed < inorder) {
                    lists[handed % G].push_back((I << 16) | J);
                    load0[handed % G] += cost;
                    ++handed;
                    if (handed == inorder) {
                        for (auto& e : heap) e.first = -load0[-e.second];
                        std::make_heap(heap.begin(), heap.end());
                    }
                } else {
                    give(I, J, cost);
                }
            };
            for (int I = 0; I < nb; ++I)
                for (int J = I + 1; J < nb; ++J) hand(I, J, 1.0);
            for (int I = 0; I < nb - L; ++I) hand(I, I, 0.5625);
        };
        // Row slivers (qn_sym2.hip.h, qn_s2_eval_sliver): when the tiles do not deal out evenly and the L left over can be cut into
        // one 8-row sliver per workgroup (16 L = G: n = 4096 on 256 workgroups), the last L diagonal tiles leave the work lists.
        // The kernels take the sliver in the place of a last, odd item: EVERY list must then have the same, even length -- true
        // when all items go out in order (n = 4096), not in general once the heap deals a tail of mixed costs (nb = 991
        // passed the arithmetic test with lists of different, odd lengths -- the sliver of such a workgroup was never evaluated).
        // So the lists are checked after the deal, and dealt again without slivers if they are not uniform.
        int L = (!sharded && nitems > G) ? nitems % G : 0;
        if (!(L > 0 && 16 * L == G && L <= nb && ((nitems - L) / G) % 2 == 0 && !s->fold && !s->no_sliver)) L = 0;
        deal(L);
        if (L) {
            bool uniform = true;
            for (int g = 0; g < G; ++g) uniform = uniform && lists[g].size() == lists[0].size() && lists[g].size() % 2 == 0;
            if (!uniform) { L = 0; deal(0); }
        }
        s->s2_sl_first = nb - L;
        s->s2_sl_per = L ? G / L : 0;
        if (s->s2_sl_per != 0 && (G % L != 0 || s->s2_sl_per != 16)) return fail(QN_ABNORMAL_TERMINATION, "sym2: row slivers do not tile the grid");
        s->s2_sl_cfg = cfg;
        size_t maxk = 0;
        for (int g = 0; g < G; ++g) maxk = std::max(maxk, lists[g].size());
        for (int g = 0; g < G; ++g)
            if (lists[g].empty()) return fail(QN_ABNORMAL_TERMINATION, "sym2: a workgroup without items");
        std::vector<int> items(maxk * (size_t)G, -1); // [k][g]: the workgroup's k-th item; -1 ends its list
        for (int g = 0; g < G; ++g)
            for (size_t k = 0; k < lists[g].size(); ++k) items[k * (size_t)G + g] = lists[g][k];
        HIPCHK(hipMalloc((void**)&s->s2_items, items.size() * sizeof(int)));
        HIPCHK(hipMemcpy(s->s2_items, items.data(), items.size() * sizeof(int), hipMemcpyHostToDevice));
        s->s2_maxk = (int)maxk;
        s->s2_inorder = inorder;
        s->s2_trows = std::max(QN_S2_MAXG, (nb + 63) / 64 * 64);
        QNCHK(dev_alloc_zero(&s->s2_wgS, (size_t)2 * s->s2_trows * QN_S2_ROW, st));
        (void)hipFree(s->s2_wgV); s->s2_wgV = nullptr; // (the generic objectives' second table: allocated by the runs that use it)
        QNCHK(dev_alloc_zero(&s->s2_partE, (size_t)nb * nb * QN_TB, st));
        if (sharded) {
            QNCHK(dev_alloc_zero(&s->s2_evS, (size_t)2 * world * QN_S2SH_NEC * QN_S2_MAXG, st));
        }

        s->s2_G = G;
        s->s2_nb = nb;
    }
    if (!s->s2_ctl) {
        HIPCHK(hipMalloc((void**)&s->s2_ctl, 2 * sizeof(QnCtl)));
        HIPCHK(hipMemsetAsync(s->s2_ctl, 0, 2 * sizeof(QnCtl), st));
    }
    return QN_OK;
}

static int solver_alloc_hp(qn_solver* s) {
    if (s->V.hp) { HIPCHK(hipFree(s->V.hp)); s->V.hp = nullptr; }
    if (s->V.q) { HIPCHK(hipFree(s->V.q)); s->V.q = nullptr; }
    QNCHK(dev_alloc_zero(&s->V.hp, (size_t)s->ctx->world * s->hcs * 2 * s->T.rpr, s->ctx->stream));
    QNCHK(dev_alloc_zero(&s->V.q, (size_t)s->ctx->world * s->qcs * s->T.rpr, s->ctx->stream));
    s->V.hcs = s->hcs;
    s->V.qcs = s->qcs;
    return QN_OK;
}

extern "C" int qn_solver_create(qn_context* ctx, int method, double tol, const double* x0_host, size_t n, qn_solver** out) {
    if (!ctx || !x0_host || !out || n == 0) return fail(QN_ERROR_INPUT_PARAMS, "null argument or n == 0");
    if (method != QN_BFGS && method != QN_DFP && method != QN_GRADIENT_DESCENT && method != QN_NEWTON && method != QN_SR1)
        return fail(QN_ERROR_INPUT_PARAMS, "unknown method");
    if (n > (size_t)1 << 30) return fail(QN_ERROR_INPUT_PARAMS, "n too large");
    HIPCHK(hipSetDevice(ctx->device));
    qn_solver* s = new qn_solver();
    *out = s;
    s->ctx = ctx; s->method = method; s->n = n; s->tol = tol;
    s->T = make_tile(n, ctx, 1);
    // row tile: 4 rows per workgroup keeps 4 workgroups per CU busy at n = 4096; at large n the per-workgroup partial
    // sums read by the control step dominate its latency, so use 8 (measured: profiles/r01_c_tiling_sweep.txt)
    s->R = (s->T.rpr >= 16384 / ctx->world && n >= 16384) ? 8 : 4;
    s->U = (n >= 16384) ? 2 : 1; // column chunks per loop trip of the fused kernels
    const size_t np = s->T.n_pad;
    hipStream_t st = ctx->stream;
    if (method == QN_BFGS || method == QN_DFP || method == QN_SR1) {
        QNCHK(dev_alloc_zero(&s->H, (size_t)s->T.rpr * np, st));
        hipLaunchKernelGGL(identity_fill_kernel, dim3(2048), dim3(256), 0, st, s->H, s->T); // bfgs.rs:27-39: H = I
        HIPCHK(hipGetLastError());
    }
    QNCHK(dev_alloc_zero(&s->vec_block, 9 * np, st));
    double* p = s->vec_block;
    s->V.x = p; s->V.g = p + np; s->V.d = p + 2 * np; s->V.xt = p + 3 * np; s->V.gt = p + 4 * np;
    s->V.s = p + 5 * np; s->V.y = p + 6 * np; s->V.sp = p + 7 * np; s->V.up = p + 8 * np;
    s->V.n = (int)n; s->V.n_pad = (int)np; s->V.rpr = s->T.rpr; s->V.world = ctx->world;
    s->V.H = s->H;
    QNCHK(solver_alloc_hp(s));
    QNCHK(dev_alloc_zero(&s->f_dev, 2, st));
    s->V.f_dev = s->f_dev;
    HIPCHK(hipMalloc((void**)&s->ctl, sizeof(QnCtl)));
    HIPCHK(hipMemsetAsync(s->ctl, 0, sizeof(QnCtl), st));
    // (mapped + coherent, explicitly: s2_ctl_upload_kernel reads the mirror from the device, call after call -- with a non-coherent
    // mapping the second call could be served a stale line from L2)
    HIPCHK(hipHostMalloc((void**)&s->hctl, 2 * sizeof(QnCtl) + 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset(s->hctl, 0, 2 * sizeof(QnCtl) + 64);
    s->hrep = s->hctl + 1;                                                    // what a batch's last launch reports (QnS2Args.rep)
    s->hrep_flag = reinterpret_cast<unsigned long long*>(s->hctl + 2);        // ... and the sequence number it stores behind it
    HIPCHK(hipHostMalloc((void**)&s->hx, n * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&s->hg, (n + 1) * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipMemcpyAsync(s->V.x, x0_host, n * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return QN_OK;
}

extern "C" void qn_solver_destroy(qn_solver* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    for (auto& e : s->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : s->event_pool) (void)hipEventDestroy(e);
    if (s->s2_graph_exec) (void)hipGraphExecDestroy(s->s2_graph_exec);
    (void)hipFree(s->H); (void)hipFree(s->vec_block); (void)hipFree(s->V.hp); (void)hipFree(s->V.q);
    (void)hipFree(s->newton_w); (void)hipFree(s->newton_x); (void)hipFree(s->newton_invl); (void)hipFree(s->newton_inv2); (void)hipFree(s->sym_part); (void)hipFree(s->symsh_xg); (void)hipFree(s->symsh_gath); (void)hipFree(s->newton_hsrc); (void)hipFree(s->newton_fail); (void)hipFree(s->newton_piv); (void)hipFree(s->newton_perm); (void)hipFree(s->newton_panel); (void)hipFree(s->newton_sync); (void)hipFree(s->newton_rec);
    (void)hipFree(s->bounds_block);
    (void)hipFree(s->fused_block); (void)hipFree(s->fused_evp); (void)hipFree(s->fused_hpp);
    (void)hipFree(s->s2_items); (void)hipFree(s->s2_wgS); (void)hipFree(s->s2_partE); (void)hipFree(s->s2_ctl);
    (void)hipFree(s->s2_evS); (void)hipFree(s->s2_cnt); (void)hipFree(s->s2_gws); (void)hipFree(s->s2_wgV); (void)hipFree(s->s2_sl_off); (void)hipFree(s->s2_sl_idx); (void)hipFree(s->symsh_tiles);
    (void)hipFree(s->f_dev); (void)hipFree(s->ctl); (void)hipFree(s->V.trace); (void)hipFree(s->V.xtrace);
    (void)hipHostFree(s->hctl); (void)hipHostFree(s->hx); (void)hipHostFree(s->hg);
    delete s;
}

extern "C" int qn_solver_set_trace(qn_solver* s, size_t cap, int with_x) {
    HIPCHK(hipSetDevice(s->ctx->device));
    if (s->V.trace) { HIPCHK(hipFree(s->V.trace)); s->V.trace = nullptr; }
    if (s->V.xtrace) { HIPCHK(hipFree(s->V.xtrace)); s->V.xtrace = nullptr; }
    s->trace_cap = cap;
    s->trace_x = with_x && cap;
    if (cap) {
        HIPCHK(hipMalloc((void**)&s->V.trace, cap * sizeof(QnTraceRec)));
        HIPCHK(hipMemset(s->V.trace, 0, cap * sizeof(QnTraceRec)));
        if (with_x) QNCHK(dev_alloc_zero(&s->V.xtrace, cap * s->n, s->ctx->stream));
    }
    return QN_OK;
}

extern "C" int qn_solver_get_trace(qn_solver* s, qn_trace_rec* out_host, size_t cap, size_t* len, double* x_trace_host) {
    HIPCHK(hipSetDevice(s->ctx->device));
    size_t m = std::min<size_t>(std::min(cap, s->trace_cap), (size_t)s->hctl->n_iterations);
    if (len) *len = m;
    static_assert(sizeof(qn_trace_rec) == sizeof(QnTraceRec), "trace record layout");
    if (m && out_host) HIPCHK(hipMemcpy(out_host, s->V.trace, m * sizeof(QnTraceRec), hipMemcpyDeviceToHost));
    if (m && x_trace_host && s->V.xtrace) HIPCHK(hipMemcpy(x_trace_host, s->V.xtrace, m * s->n * sizeof(double), hipMemcpyDeviceToHost));
    return QN_OK;
}

#if defined(QN_CTL_STAMPS) || defined(QN_S2_STAMPS)
extern "C" int qn_debug_stamps(qn_solver* s, unsigned long long* out, size_t count) { // diagnostic build only
    HIPCHK(hipSetDevice(s->ctx->device));
    if (!s->V.dbg) {
        HIPCHK(hipMalloc((void**)&s->V.dbg, (1 << 20) * 8));
        HIPCHK(hipMemset(s->V.dbg, 0, (1 << 20) * 8));
        return QN_OK;
    }
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    HIPCHK(hipMemcpy(out, s->V.dbg, count * 8, hipMemcpyDeviceToHost));
    return QN_OK;
}
#endif

#ifdef QN_LU_STAMPS
extern "C" int qn_debug_lu_stamps(unsigned long long* out) { // diagnostic build: the stamps of the LAST panel's step launches
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(qn_lu_dbg), sizeof(unsigned long long) * 64 * 16));
    return QN_OK;
}
#endif
extern "C" int qn_solver_set_profiling(qn_solver* s, int on) { s->profiling = on; return QN_OK; }
extern "C" int qn_solver_set_sync_mode(qn_solver* s, int sync) { s->sync_mode = sync; return QN_OK; }
// Named options (ABI 5): what rounds 1-5 selected through negative `rows_per_block` codes of qn_solver_set_tiling.  Every option
// SETS a state (value != 0: on), none toggles; the defaults are in include/qn_hip.h.
extern "C" int qn_solver_set_option(qn_solver* s, int option, int value) {
    if (!s) return fail(QN_ERROR_INPUT_PARAMS, "null argument");
    const bool on = value != 0;
    switch (option) {
    case QN_OPT_GENERIC_KERNELS: s->no_fused = on ? 1 : 0; return QN_OK;
    case QN_OPT_DEFERRED_UPDATE_STEP: s->no_defer = on ? 0 : 1; return QN_OK;
    case QN_OPT_SYMMETRIC_STORAGE: s->no_sym = on ? 0 : 1; return QN_OK;
    case QN_OPT_SECOND_GENERATION: s->no_sym2 = on ? 0 : 1; return QN_OK;
    case QN_OPT_FOLDED_ACCEPT_REDUCE: s->fold = on ? 1 : 0; return QN_OK;
    case QN_OPT_ROW_SLIVERS: s->no_sliver = !on; return QN_OK;
    case QN_OPT_EVAL_PAIR_INSTANCE: s->no_pair = !on; return QN_OK;
    case QN_OPT_EVAL_MOVER_MULTIPLIER: s->ring = on ? 1 : 0; return QN_OK;
    case QN_OPT_TAIL_REDUCE: s->tred = on; return QN_OK;
    case QN_OPT_BOUNDED_SECOND_GENERATION: s->no_s2bnd = !on; return QN_OK;
    case QN_OPT_NEWTON_PIVOTED_LU: s->newton_force_lu = on ? 1 : 0; return QN_OK;
    case QN_OPT_LU_PER_COLUMN_PANEL: s->newton_lu_percol = on ? 1 : 0; return QN_OK;
    case QN_OPT_LU_LOOKAHEAD: s->newton_lu_no_la = on ? 0 : 1; return QN_OK;
    case QN_OPT_LU_ONE_LAUNCH_PANEL: s->newton_lu_no_persist = on ? 0 : 1; return QN_OK;
    case QN_OPT_LU_FORCE_WAIT_EXPIRY: s->newton_lu_force_timeout = on ? 1 : 0; return QN_OK;
    case QN_OPT_BTB_PROJECT_IN_EVAL: s->no_projfold = !on; return QN_OK;
    case QN_OPT_EVAL_ZIGZAG: s->zig = on ? 1 : 0; return QN_OK;
    case QN_OPT_TOUCH_H_ROWS:
    case QN_OPT_TOUCH_Q_ROWS:
        if (value != 0 && value != 4 && value != 6 && value != 8 && value != 10 && value != 12 && value != 16) return fail(QN_ERROR_INPUT_PARAMS, "rows per wave to touch: 0, 4, 6, 8, 10, 12 or 16");
        (option == QN_OPT_TOUCH_H_ROWS ? s->touch : s->touchq) = value;
        return QN_OK;
    case QN_OPT_LU_SPLIT_ROLE_A:
        if (value != 0 && value != 1 && value != 2 && value != 4) return fail(QN_ERROR_INPUT_PARAMS, "role A runs as 1, 2 or 4 workgroups");
        s->newton_lu_split = value == 0 ? 1 : value;
        return QN_OK;
    case QN_OPT_LU_SPLIT_MIN_ROWS:
        if (value < 0) return fail(QN_ERROR_INPUT_PARAMS, "a number of rows");
        s->newton_lu_split_min = value;
        return QN_OK;
    case QN_OPT_CHUNKS_PER_TRIP:
        if (value != 1 && value != 2 && value != 4) return fail(QN_ERROR_INPUT_PARAMS, "chunks per trip must be 1, 2 or 4");
        s->U = value;
        return QN_OK;
    default: return fail(QN_ERROR_INPUT_PARAMS, "unknown option");
    }
}

// tuning only: rows per workgroup tile of the fused ROW kernels (2, 4, 8 or 16) and column splits (1 .. 64); 0 keeps what is set
extern "C" int qn_solver_set_tiling(qn_solver* s, int rows_per_block, int col_splits) {
    if (!s) return fail(QN_ERROR_INPUT_PARAMS, "null argument");
    if (rows_per_block != 0 && rows_per_block != 2 && rows_per_block != 4 && rows_per_block != 8 && rows_per_block != 16)
        return fail(QN_ERROR_INPUT_PARAMS, "rows_per_block must be 2, 4, 8 or 16 (diagnostics are named options: qn_solver_set_option)");
    if (col_splits < 0 || col_splits > 64) return fail(QN_ERROR_INPUT_PARAMS, "col_splits out of range");
    HIPCHK(hipSetDevice(s->ctx->device));
    if (rows_per_block) s->R = rows_per_block;
    if (col_splits) { s->hcs = col_splits; s->qcs = col_splits; }
    return solver_alloc_hp(s);
}

extern "C" size_t qn_solver_n(const qn_solver* s) { return s->n; }
extern "C" size_t qn_solver_k(const qn_solver* s) { return (size_t)s->hctl->k; }
extern "C" double qn_solver_tol(const qn_solver* s) { return s->tol; }

// canonical buffers <- fused buffers (the lazy half of qn_minimize's export)
static int fused_export(qn_solver* s) {
    s->warm_obj = 0; // whoever asks for the canonical buffers may change them: the next call starts from scratch
    if (!s->fused_live) return QN_OK;
    qn_context* c = s->ctx;
    const QnCtl* h = s->hctl;
    const size_t np = s->T.n_pad, vb = np * sizeof(double);
    HIPCHK(hipMemcpyAsync(s->V.x, s->V.F.X0 + (size_t)h->xc * np, vb, hipMemcpyDeviceToDevice, c->stream));
    if (h->pending) {
        HIPCHK(hipMemcpyAsync(s->V.sp, s->V.F.S0 + (size_t)h->sc * np, vb, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(s->V.up, s->V.F.UN, vb, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    s->fused_live = false;
    return QN_OK;
}

extern "C" int qn_solver_get_x(qn_solver* s, double* out) {
    HIPCHK(hipSetDevice(s->ctx->device));
    // (a getter: read the iterate where it lives, do not disturb a run that may be continued)
    const double* src = s->fused_live ? s->V.F.X0 + (size_t)s->hctl->xc * (size_t)s->T.n_pad : s->V.x;
    HIPCHK(hipMemcpyAsync(out, src, s->n * sizeof(double), hipMemcpyDeviceToHost, s->ctx->stream));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    return QN_OK;
}

static int poke_ctl(qn_solver* s) { // host mirror -> device
    HIPCHK(hipMemcpyAsync(s->ctl, s->hctl, sizeof(QnCtl), hipMemcpyHostToDevice, s->ctx->stream));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    return QN_OK;
}
static int peek_ctl(qn_solver* s) { // device -> host mirror
    HIPCHK(hipMemcpyAsync(s->hctl, s->ctl, sizeof(QnCtl), hipMemcpyDeviceToHost, s->ctx->stream));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    s->stats.host_syncs++;
    return QN_OK;
}

// box vectors live in one allocation: [lb | ub | llb | lub], padding -inf / +inf so padded entries never move
static int bounds_alloc(qn_solver* s) {
    if (s->bounds_block) return QN_OK;
    const size_t np = s->T.n_pad;
    HIPCHK(hipMalloc((void**)&s->bounds_block, 4 * np * sizeof(double)));
    std::vector<double> init(4 * np);
    for (size_t i = 0; i < np; ++i) { init[i] = -INFINITY; init[np + i] = INFINITY; init[2 * np + i] = -INFINITY; init[3 * np + i] = INFINITY; }
    HIPCHK(hipMemcpy(s->bounds_block, init.data(), init.size() * sizeof(double), hipMemcpyHostToDevice));
    s->V.lb = s->bounds_block; s->V.ub = s->bounds_block + np; s->V.llb = s->bounds_block + 2 * np; s->V.lub = s->bounds_block + 3 * np;
    return QN_OK;
}
static int bounds_upload(qn_solver* s, double* dst, const double* src_host, double fill) {
    std::vector<double> v(s->T.n_pad, fill);
    if (src_host) memcpy(v.data(), src_host, s->n * sizeof(double));
    HIPCHK(hipMemcpy(dst, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    return QN_OK;
}

extern "C" int qn_solver_set_bounds(qn_solver* s, const double* lb_host, const double* ub_host) { // BFGSB::new, bfgs_b.rs:43-63
    if (!s || !lb_host || !ub_host) return fail(QN_ERROR_INPUT_PARAMS, "null argument");
    if (s->method != QN_BFGS && s->method != QN_DFP && s->method != QN_SR1) return fail(QN_ERROR_INPUT_PARAMS, "bounds need a BFGS / DFP / SR1 solver");
    HIPCHK(hipSetDevice(s->ctx->device));
    QNCHK(bounds_alloc(s));
    QNCHK(bounds_upload(s, s->bounds_block, lb_host, -INFINITY));
    QNCHK(bounds_upload(s, s->bounds_block + s->T.n_pad, ub_host, INFINITY));
    std::vector<double> x(s->n);
    QNCHK(qn_solver_get_x(s, x.data()));
    for (size_t i = 0; i < s->n; ++i) x[i] = std::fmin(std::fmax(x[i], lb_host[i]), ub_host[i]); // x0.box_projection(&lower, &upper), :49
    s->bounded = 1;
    return qn_solver_set_x(s, x.data());
}

extern "C" int qn_solver_reset(qn_solver* s, const double* x0_host) {
    if (!s || !x0_host) return fail(QN_ERROR_INPUT_PARAMS, "null argument");
    HIPCHK(hipSetDevice(s->ctx->device));
    hipStream_t st = s->ctx->stream;
    if (s->H) {
        hipLaunchKernelGGL(identity_fill_kernel, dim3(2048), dim3(256), 0, st, s->H, s->T);
        HIPCHK(hipGetLastError());
        s->h_lower_stale = false; s->h_diag_stale = false;
        s->h_nonsym = false;
    }
    s->fused_live = false; s->warm_obj = 0; // (whatever the fused buffers hold is dropped with the rest of the state)
    HIPCHK(hipMemsetAsync(s->vec_block, 0, 9 * (size_t)s->T.n_pad * sizeof(double), st));
    HIPCHK(hipMemcpyAsync(s->V.x, x0_host, s->n * sizeof(double), hipMemcpyHostToDevice, st));
    memset(s->hctl, 0, sizeof(QnCtl));
    return poke_ctl(s);
}

extern "C" int qn_solver_set_k(qn_solver* s, size_t k) { // k_mut() (bfgs.rs:58-63); minimize() resets it to 0 itself (ls_solver.rs:74-76)
    if (!s) return fail(QN_ERROR_INPUT_PARAMS, "null argument");
    s->hctl->k = (int64_t)k;
    return poke_ctl(s);
}
extern "C" int qn_solver_set_x(qn_solver* s, const double* x_host) {
    HIPCHK(hipSetDevice(s->ctx->device));
    QNCHK(fused_export(s));
    HIPCHK(hipMemcpyAsync(s->V.x, x_host, s->n * sizeof(double), hipMemcpyHostToDevice, s->ctx->stream));
    s->hctl->have_cur_eval = 0; s->hctl->have_dir = 0; s->hctl->last_valid = 0;
    return poke_ctl(s);
}
extern "C" int qn_solver_s_norm(qn_solver* s, double* out, int* is_some) {
    if (is_some) *is_some = s->hctl->has_s_norm;
    if (out) *out = s->hctl->s_norm;
    return QN_OK;
}
extern "C" int qn_solver_y_norm(qn_solver* s, double* out, int* is_some) {
    if (is_some) *is_some = s->hctl->has_y_norm;
    if (out) *out = s->hctl->y_norm;
    return QN_OK;
}
extern "C" int qn_solver_decrement_squared(qn_solver* s, double* out, int* is_some) { // newton/mod.rs:10
    if (is_some) *is_some = s->hctl->has_dec;
    if (out) *out = s->hctl->dec;
    return QN_OK;
}
extern "C" int qn_solver_next_iterate_too_close(qn_solver* s, int* out) { // bfgs.rs:15-20
    *out = s->hctl->has_s_norm && s->hctl->s_norm < s->tol;
    return QN_OK;
}
extern "C" int qn_solver_gradient_next_iterate_too_close(qn_solver* s, int* out) { // bfgs.rs:21-26
    *out = s->hctl->has_y_norm && s->hctl->y_norm < s->tol;
    return QN_OK;
}

// ---- launches ----
template <int R>
static void launch_hpass(hipStream_t st, const QnHPassArgs& a) {
    dim3 grid(a.T.rpr / R, a.T.cs);
    hipLaunchKernelGGL(h_pass_kernel<R>, grid, dim3(QN_TPB), 0, st, a);
}
static int launch_hpass_R(qn_solver* s, const QnHPassArgs& a) {
    ProfScope ps(s, KC_HPASS);
    switch (s->R) {
    case 4: launch_hpass<4>(s->ctx->stream, a); break;
    case 16: launch_hpass<16>(s->ctx->stream, a); break;
    default: launch_hpass<8>(s->ctx->stream, a); break;
    }
    s->stats.launches++;
    HIPCHK(hipGetLastError());
    return QN_OK;
}

static QnHPassArgs hpass_args(qn_solver* s, int expect_phase) {
    QnHPassArgs a{};
    a.H = s->H;
    a.T = s->T; a.T.cs = s->hcs;
    a.sp = s->V.sp; a.up = s->V.up;
    a.vy = s->V.y; a.vg = s->V.g;
    a.r0 = nullptr; a.r1 = nullptr;
    a.hp = s->V.hp;
    a.ctl = s->ctl;
    a.expect_phase = expect_phase;
    return a;
}

static QnSymShard sym_shard(const qn_solver* s) {
    QnSymShard sh{};
    sh.world = s->ctx->world; sh.rank = s->ctx->rank;
    sh.nbl = s->T.rpr / QN_TB; sh.ioff = sh.rank * sh.nbl;
    sh.xg = s->symsh_xg;
    sh.sl_off = s->s2_sl_off; sh.sl_idx = s->s2_sl_idx; sh.tiles = s->symsh_tiles;
    sh.nsum = s->ctx->use_allreduce ? 1 : sh.world; // all-reduce mode: the exchange already left the total in slice 0
    return sh;
}

// the symmetric-storage paths maintain the upper block triangle only: restore the lower one before anything reads whole rows
static int ensure_full_h(qn_solver* s) {
    if (!s->H || !s->h_lower_stale) return QN_OK;
    qn_context* c = s->ctx;
    if (c->world > 1) { // row-sharded: the stale half of a block-row is maintained by other ranks (circulant windows, qn_sym.hip.h)
        const size_t np = (size_t)s->T.n_pad, blk = (size_t)QN_TB * np;
        const QnSymShard sh = sym_shard(s);
        if (!s->symsh_gath) HIPCHK(hipMalloc((void**)&s->symsh_gath, (size_t)c->world * blk * sizeof(double)));
        if (s->h_diag_stale) // (second-generation tiles: inside the local diagonal tiles only the upper 16 x 16 sub-blocks are current)
            hipLaunchKernelGGL(s2sh_diag_mirror_kernel, dim3(QN_TB / 32, QN_TB / 32, sh.nbl), dim3(256), 0, c->stream, s->H, s->T.n_pad, sh.ioff);
        for (int il = 0; il < sh.nbl; ++il) {
            HIPCHK(hipMemcpyAsync(s->symsh_gath + (size_t)c->rank * blk, s->H + (size_t)il * blk, blk * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            QNCHK(exchange(c, s->symsh_gath, blk));
            hipLaunchKernelGGL(symsh_mirror_kernel, dim3(sh.nbl, c->world), dim3(256), 0, c->stream, s->H, s->symsh_gath, s->T.n_pad,
                               s->T.n_pad / QN_TB, il, sh);
        }
        HIPCHK(hipGetLastError());
        s->h_lower_stale = false;
        s->h_diag_stale = false;
        return QN_OK;
    }
    const int b32 = s->T.n_pad / 32;
    hipLaunchKernelGGL(sym2_mirror_kernel, dim3(b32, b32), dim3(256), 0, s->ctx->stream, s->H, s->T.n_pad); // (also inside the diagonal tiles)
    HIPCHK(hipGetLastError());
    s->h_lower_stale = false;
    s->h_diag_stale = false;
    return QN_OK;
}

static int flush_pending(qn_solver* s) { // H_stored <- H_true
    QNCHK(fused_export(s));  // the pending update's vectors
    QNCHK(ensure_full_h(s)); // (also from a callback in the middle of a symmetric-storage run)
    if (!s->H || !s->hctl->pending) return QN_OK;
    QnHPassArgs a = hpass_args(s, -1);
    a.force_nrhs = 0; a.force_pending = 1;
    a.c_ss = s->hctl->c_ss; a.c_su = s->hctl->c_su; a.c_uu = s->hctl->c_uu;
    QNCHK(launch_hpass_R(s, a));
    s->hctl->pending = 0;
    return poke_ctl(s);
}

extern "C" int qn_solver_get_inv_hessian(qn_solver* s, double* out, int all_ranks) {
    if (!s->H) return fail(QN_ERROR_INPUT_PARAMS, "gradient descent keeps no inverse Hessian");
    qn_context* c = s->ctx;
    HIPCHK(hipSetDevice(c->device));
    QNCHK(flush_pending(s));
    const size_t n = s->n, np = s->T.n_pad, rpr = s->T.rpr;
    const size_t chunk = 16;
    std::vector<double> rows(chunk * np * (all_ranks ? c->world : 1));
    double* tmp = nullptr;
    if (all_ranks && c->world > 1) HIPCHK(hipMalloc((void**)&tmp, (size_t)c->world * chunk * np * sizeof(double)));
    for (size_t r0 = 0; r0 < rpr; r0 += chunk) {
        if (all_ranks && c->world > 1) {
            HIPCHK(hipMemcpyAsync(tmp + (size_t)c->rank * chunk * np, s->H + r0 * np, chunk * np * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            QNCHK(exchange(c, tmp, chunk * np));
            HIPCHK(hipMemcpyAsync(rows.data(), tmp, (size_t)c->world * chunk * np * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (int p = 0; p < c->world; ++p)
                for (size_t r = 0; r < chunk; ++r) {
                    const size_t i = (size_t)p * rpr + r0 + r;
                    if (i >= n) continue;
                    const double* row = rows.data() + ((size_t)p * chunk + r) * np;
                    for (size_t j = 0; j < n; ++j) out[i + j * n] = row[j];
                }
        } else {
            HIPCHK(hipMemcpyAsync(rows.data(), s->H + r0 * np, chunk * np * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (size_t r = 0; r < chunk; ++r) {
                const size_t i = (size_t)s->T.row_off + r0 + r;
                if (i >= n) continue;
                const double* row = rows.data() + r * np;
                for (size_t j = 0; j < n; ++j) out[i + j * n] = row[j];
            }
        }
    }
    if (tmp) HIPCHK(hipFree(tmp));
    return QN_OK;
}

// ComputeDirection::compute_direction (bfgs.rs:42-49, dfp.rs:42-49: `-&self.approx_inv_hessian * eval.g()`;
// gradient_descent.rs:24-30: `-eval.g()`) on its own, for a binding that implements the trait: g goes up, d comes down.
extern "C" int qn_solver_compute_direction(qn_solver* s, const double* g_host, double* d_host) {
    if (!s || !g_host || !d_host) return fail(QN_ERROR_INPUT_PARAMS, "null argument");
    if (s->method == QN_NEWTON) return fail(QN_ERROR_INPUT_PARAMS, "the Newton direction needs the oracle's Hessian: use qn_minimize");
    const size_t n = s->n;
    if (!s->H) { // gradient descent
        for (size_t i = 0; i < n; ++i) d_host[i] = -g_host[i];
        return QN_OK;
    }
    qn_context* c = s->ctx;
    HIPCHK(hipSetDevice(c->device));
    QNCHK(flush_pending(s)); // the lazy update of the last iteration, and whole rows of H
    const size_t np = s->T.n_pad, rpr = s->T.rpr;
    double* buf = nullptr; // [g (np) | H g (np)]
    HIPCHK(hipMalloc((void**)&buf, 2 * np * sizeof(double)));
    HIPCHK(hipMemsetAsync(buf, 0, 2 * np * sizeof(double), c->stream));
    HIPCHK(hipMemcpyAsync(buf, g_host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int st = qn_gemv(c, s->H, np, rpr, n, buf, buf + np + (size_t)s->T.row_off);
    if (st == QN_OK) st = exchange(c, buf + np, rpr);
    if (st == QN_OK) {
        hipError_t e = hipMemcpyAsync(d_host, buf + np, n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) st = fail(QN_ABNORMAL_TERMINATION, hipGetErrorString(e));
    }
    (void)hipFree(buf);
    if (st != QN_OK) return st;
    if (s->bounded) { // BFGSB / DFPB / SR1B: P(x - H g) - x with the solver's box (bfgs_b.rs:66-77), not -H g
        std::vector<double> x(n), lb(n), ub(n);
        QNCHK(qn_solver_get_x(s, x.data()));
        HIPCHK(hipMemcpy(lb.data(), s->bounds_block, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ub.data(), s->bounds_block + np, n * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) {
            double t = x[i] - d_host[i];
            t = std::fmin(std::fmax(t, lb[i]), ub[i]);
            d_host[i] = t - x[i];
        }
        return QN_OK;
    }
    for (size_t i = 0; i < n; ++i) d_host[i] = -d_host[i];
    return QN_OK;
}

// The second half of the `update_next_iterate` hook (bfgs.rs:92-130, dfp.rs:92-118) on its own, for a binding that implements
// LineSearchSolver hook by hook: records ||s|| and ||y|| (s_norm / y_norm), returns early when either is below tol
// (bfgs.rs:103-109), otherwise applies the secant update to the device-resident inverse Hessian as the rank-2 form of DESIGN.md 4
// (u = H y; BFGS: rho = 1/y's, H += -rho (su' + us') + (rho^2 y'u + rho) ss'; DFP: H += ss'/y's - uu'/y'u).
extern "C" int qn_solver_secant_update(qn_solver* s, const double* s_host, const double* y_host) {
    if (!s || !s_host || !y_host) return fail(QN_ERROR_INPUT_PARAMS, "null argument");
    if ((s->method != QN_BFGS && s->method != QN_DFP) || s->bounded)
        return fail(QN_ERROR_INPUT_PARAMS, "secant update: BFGS and DFP only");
    qn_context* c = s->ctx;
    HIPCHK(hipSetDevice(c->device));
    QNCHK(flush_pending(s));
    const size_t n = s->n, np = s->T.n_pad, rpr = s->T.rpr;
    double ss = 0.0, yy = 0.0, ys = 0.0;
    for (size_t i = 0; i < n; ++i) { ss += s_host[i] * s_host[i]; yy += y_host[i] * y_host[i]; ys += y_host[i] * s_host[i]; }
    QnCtl* h = s->hctl;
    h->has_s_norm = 1; h->s_norm = std::sqrt(ss);
    h->has_y_norm = 1; h->y_norm = std::sqrt(yy);
    h->have_dir = 0; h->have_cur_eval = 0;
    QNCHK(poke_ctl(s));
    if (h->s_norm < s->tol || h->y_norm < s->tol) return QN_OK;
    double* buf = nullptr; // [s | y | u = H y], n_pad each
    HIPCHK(hipMalloc((void**)&buf, 3 * np * sizeof(double)));
    std::vector<double> u(n);
    int st = QN_OK;
    auto hip_ok = [&](hipError_t e) { if (e != hipSuccess && st == QN_OK) st = fail(QN_ABNORMAL_TERMINATION, hipGetErrorString(e)); return e == hipSuccess; };
    hip_ok(hipMemsetAsync(buf, 0, 3 * np * sizeof(double), c->stream));
    hip_ok(hipMemcpyAsync(buf, s_host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    hip_ok(hipMemcpyAsync(buf + np, y_host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (st == QN_OK) st = qn_gemv(c, s->H, np, rpr, n, buf + np, buf + 2 * np + (size_t)s->T.row_off);
    if (st == QN_OK) st = exchange(c, buf + 2 * np, rpr);
    if (st == QN_OK) {
        hip_ok(hipMemcpyAsync(u.data(), buf + 2 * np, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        hip_ok(hipStreamSynchronize(c->stream));
    }
    if (st == QN_OK) {
        double yu = 0.0;
        for (size_t i = 0; i < n; ++i) yu += y_host[i] * u[i];
        double c_ss, c_su, c_uu;
        if (s->method == QN_BFGS) { const double rho = 1.0 / ys; c_su = -rho; c_ss = rho * rho * yu + rho; c_uu = 0.0; }
        else { c_ss = 1.0 / ys; c_su = 0.0; c_uu = -1.0 / yu; }
        st = qn_rank2_update(c, s->H, np, (size_t)s->T.row_off, rpr, n, buf, buf + 2 * np, c_ss, c_su, c_uu);
        if (st == QN_OK) hip_ok(hipStreamSynchronize(c->stream));
    }
    (void)hipFree(buf);
    return st;
}

extern "C" int qn_solver_set_inv_hessian(qn_solver* s, const double* h) {
    if (!s->H) return fail(QN_ERROR_INPUT_PARAMS, "gradient descent keeps no inverse Hessian");
    HIPCHK(hipSetDevice(s->ctx->device));
    const size_t n = s->n, np = s->T.n_pad;
    std::vector<double> rows((size_t)s->T.rpr * np, 0.0);
    for (size_t r = 0; r < (size_t)s->T.rpr; ++r) {
        const size_t i = (size_t)s->T.row_off + r;
        if (i >= n) break;
        for (size_t j = 0; j < n; ++j) rows[r * np + j] = h[i + j * n];
    }
    QNCHK(fused_export(s));
    HIPCHK(hipMemcpy(s->H, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
    s->h_lower_stale = false; s->h_diag_stale = false; // every entry was just replaced
    s->h_nonsym = false; // the symmetric-storage path needs H == H' bit for bit (BFGS / DFP keep it so from a symmetric start)
    for (size_t i = 0; i < n && !s->h_nonsym; ++i)
        for (size_t j = i + 1; j < n; ++j)
            if (h[i + j * n] != h[j + i * n]) { s->h_nonsym = true; break; }
    s->hctl->pending = 0; s->hctl->have_dir = 0;
    return poke_ctl(s);
}

extern "C" int qn_solver_get_stats(qn_solver* s, qn_stats* out) {
    HIPCHK(hipSetDevice(s->ctx->device));
    prof_collect(s);
    *out = s->stats;
    return QN_OK;
}
