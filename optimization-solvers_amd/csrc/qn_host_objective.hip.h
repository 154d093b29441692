// qn_host_objective.hip.h -- host side, part 3 of 7: device-resident objectives (the synthetic quadratic, the log-sum-exp) and their evaluation launches.
#pragma once
// ------------------------------------------------------------------------------------------------
// objectives
// ------------------------------------------------------------------------------------------------
enum { OBJ_QUADRATIC = 1, OBJ_LOGSUMEXP = 2 };
static std::atomic<uint64_t> g_objective_serial{0}; // objectives are numbered at creation: an address can come back after a destroy, a serial cannot
struct qn_objective {
    uint64_t serial = ++g_objective_serial;
    qn_context* ctx = nullptr;
    int kind = 0;
    size_t n = 0;
    QnTile T{};
    double* Q = nullptr; // this rank's rows, [rpr][n_pad]
    double* b = nullptr; // n_pad
    bool q_symmetric = true; // quadratic: Q == Q' bit for bit (checked at creation); the symmetric-storage evaluation needs it
    // scratch for qn_objective_eval
    double *ex = nullptr, *eq = nullptr, *eg = nullptr, *ef = nullptr;
    // log-sum-exp: A rows are in Q ([mrpr][n_pad]), c in b (m_pad)
    size_t m = 0;
    QnTile TA{}; // row partition of A (m rows)
    double mu = 0.0;
    int lse_rs = 1;
    int lse_two_pass = 0; // diagnostics (QN_LSE_TWO_PASS=1 in the environment at creation): the round-1 two-pass evaluation
    double *lz = nullptr, *lw = nullptr, *lgpart = nullptr, *lgall = nullptr, *lscal = nullptr;
    double *lwgms = nullptr, *lwgg = nullptr, *lms = nullptr; // one-pass evaluation: per-workgroup (m, S), G vectors; gathered per-rank (m, S)
    int lse_G = 0, lse_kch = 0;                                 // its grid and column chunks per thread (0: two-pass evaluation)
};

static int objective_base(qn_context* ctx, size_t n, const double* b_host, qn_objective** out) {
    if (!ctx || !out || !b_host || n == 0) return fail(QN_ERROR_INPUT_PARAMS, "null argument or n == 0");
    if (n > (size_t)1 << 30) return fail(QN_ERROR_INPUT_PARAMS, "n too large");
    HIPCHK(hipSetDevice(ctx->device));
    qn_objective* o = new qn_objective();
    o->ctx = ctx;
    o->kind = OBJ_QUADRATIC;
    o->n = n;
    o->T = make_tile(n, ctx, 1);
    *out = o;
    QNCHK(dev_alloc_zero(&o->Q, (size_t)o->T.rpr * o->T.n_pad, ctx->stream));
    QNCHK(dev_alloc_zero(&o->b, o->T.n_pad, ctx->stream));
    HIPCHK(hipMemcpyAsync(o->b, b_host, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return QN_OK;
}

extern "C" int qn_quadratic_create(qn_context* ctx, size_t n, const double* q_host, const double* b_host, qn_objective** out) {
    if (!q_host) return fail(QN_ERROR_INPUT_PARAMS, "q is null");
    QNCHK(objective_base(ctx, n, b_host, out));
    qn_objective* o = *out;
    const size_t r0 = (size_t)o->T.row_off;
    if (r0 < n) {
        const size_t nr = std::min((size_t)o->T.rpr, n - r0);
        HIPCHK(hipMemcpy2D(o->Q, (size_t)o->T.n_pad * sizeof(double), q_host + r0 * n, n * sizeof(double), n * sizeof(double), nr,
                           hipMemcpyHostToDevice));
    }
    // g = Q x - b is the gradient of f only for a symmetric Q, but nothing stops a caller from passing another matrix: the row
    // kernels multiply by what was given, so the symmetric-storage evaluation is used only when Q is symmetric bit for bit
    for (size_t i = 0; i < n && o->q_symmetric; ++i)
        for (size_t j = i + 1; j < n; ++j)
            if (q_host[i * n + j] != q_host[j * n + i]) { o->q_symmetric = false; break; }
    return QN_OK;
}

extern "C" int qn_quadratic_create_synthetic(qn_context* ctx, size_t n, uint64_t seed, const double* diag_host,
                                             const double* b_host, qn_objective** out) {
    if (!diag_host) return fail(QN_ERROR_INPUT_PARAMS, "diag is null");
    QNCHK(objective_base(ctx, n, b_host, out));
    qn_objective* o = *out;
    double* diag = nullptr;
    HIPCHK(hipMalloc((void**)&diag, n * sizeof(double)));
    HIPCHK(hipMemcpyAsync(diag, diag_host, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(synth_fill_kernel, dim3(2048), dim3(256), 0, ctx->stream, o->Q, o->T, seed, diag, 1.0 / (double)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(diag));
    return QN_OK;
}

extern "C" int qn_logsumexp_create(qn_context* ctx, size_t m, size_t n, const double* a_host, const double* c_host, double mu,
                                   qn_objective** out) {
    if (!ctx || !out || !a_host || !c_host || n == 0 || m == 0) return fail(QN_ERROR_INPUT_PARAMS, "null argument or empty shape");
    if (n > (size_t)1 << 30 || m > (size_t)1 << 30) return fail(QN_ERROR_INPUT_PARAMS, "shape too large");
    HIPCHK(hipSetDevice(ctx->device));
    qn_objective* o = new qn_objective();
    *out = o;
    o->ctx = ctx; o->kind = OBJ_LOGSUMEXP; o->n = n; o->m = m; o->mu = mu;
    o->T = make_tile(n, ctx, 1);  // column padding follows the solver's vectors
    o->TA = make_tile(m, ctx, 1); // rows of A are sharded
    const size_t np = o->T.n_pad, mp = o->TA.n_pad, mrpr = o->TA.rpr;
    hipStream_t st = ctx->stream;
    QNCHK(dev_alloc_zero(&o->Q, mrpr * np, st));
    QNCHK(dev_alloc_zero(&o->b, mp, st));
    HIPCHK(hipMemcpyAsync(o->b, c_host, m * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st)); // the zero-fill above must land before the (null-stream) 2-D upload below
    const size_t r0 = (size_t)o->TA.row_off;
    if (r0 < m) {
        const size_t nr = std::min(mrpr, m - r0);
        HIPCHK(hipMemcpy2D(o->Q, np * sizeof(double), a_host + r0 * n, n * sizeof(double), n * sizeof(double), nr, hipMemcpyHostToDevice));
    }
    o->lse_rs = (int)std::max<size_t>(1, std::min<size_t>(64, mrpr / 64));
    QNCHK(dev_alloc_zero(&o->lz, (size_t)ctx->world * 2 * mrpr, st));
    QNCHK(dev_alloc_zero(&o->lw, mp, st));
    QNCHK(dev_alloc_zero(&o->lgpart, (size_t)o->lse_rs * np, st));
    QNCHK(dev_alloc_zero(&o->lgall, (size_t)ctx->world * np, st));
    QNCHK(dev_alloc_zero(&o->lscal, 2, st));
    o->lse_two_pass = getenv("QN_LSE_TWO_PASS") && atoi(getenv("QN_LSE_TWO_PASS")) != 0;
    if (np <= 16384 && np >= 2) { // the one-pass evaluation keeps a whole row per workgroup in registers
        int kch = 1;
        while ((size_t)kch * 1024 < np) kch *= 2;
        o->lse_kch = kch;
        o->lse_G = (int)std::max<size_t>(1, std::min<size_t>(256, (mrpr + 3) / 4));
        QNCHK(dev_alloc_zero(&o->lwgms, 2 * (size_t)o->lse_G, st));
        QNCHK(dev_alloc_zero(&o->lwgg, (size_t)o->lse_G * np, st));
        QNCHK(dev_alloc_zero(&o->lms, 2 * (size_t)ctx->world, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return QN_OK;
}

template <int R>
static void launch_hpass(hipStream_t st, const QnHPassArgs& a);

// enqueue one evaluation of the log-sum-exp objective at x_dev (n_pad entries): f -> f_dev, g -> g_dev
template <int KCH, bool NTA>
static int lse_launch_onepass_nt(hipStream_t st, int G, const QnLseArgs& a, double* wgms, double* wgg) {
    static std::atomic<bool> attr_set[64]; // per device: hipFuncSetAttribute applies to the current device only (atomic: ranks may be threads)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t lds = (size_t)KCH * 1024 * sizeof(double);
    if ((dev < 0 || dev >= 64 || !attr_set[dev].load()) && lds > 48 * 1024) { // x in LDS: up to 128 KB of the CU's 160 KB
        if (hipFuncSetAttribute((const void*)lse_onepass_kernel<KCH, NTA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return -1; // this device does not grant the LDS: the caller keeps the two-pass evaluation
        }
        if (dev >= 0 && dev < 64) attr_set[dev].store(true);
    }
    hipLaunchKernelGGL((lse_onepass_kernel<KCH, NTA>), dim3(G), dim3(512), lds, st, a, wgms, wgg);
    return QN_OK;
}
template <int KCH>
static int lse_launch_onepass(hipStream_t st, int G, const QnLseArgs& a, double* wgms, double* wgg) {
    // rows of A that cannot stay in the 256 MB Infinity Cache between evaluations are requested as non-temporal (QN_LSE_NT = 0 / 1 overrides)
    static const int nt_env = getenv("QN_LSE_NT") ? atoi(getenv("QN_LSE_NT")) : -1;
    const bool nt = nt_env >= 0 ? nt_env != 0 : (size_t)a.mrpr * (size_t)a.n_pad * sizeof(double) > ((size_t)230 << 20);
    return nt ? lse_launch_onepass_nt<KCH, true>(st, G, a, wgms, wgg) : lse_launch_onepass_nt<KCH, false>(st, G, a, wgms, wgg);
}

static int lse_enqueue_eval(qn_objective* o, const double* x_dev, double* f_dev, double* g_dev) {
    qn_context* c = o->ctx;
    hipStream_t st = c->stream;
    if (o->lse_kch && !o->lse_two_pass) { // one pass over A (qn_kernels.hip.h): 8 m n / P bytes per evaluation instead of 16 m n / P
        QnLseArgs a{};
        a.A = o->Q; a.c = o->b; a.gall = o->lgall; a.x = x_dev; a.f_out = f_dev; a.g_out = g_dev; a.mu = o->mu;
        a.m = (int)o->m; a.m_pad = o->TA.n_pad; a.mrpr = o->TA.rpr; a.n = (int)o->n; a.n_pad = o->T.n_pad;
        a.world = c->world; a.rank = c->rank; a.rs = 1;
        int lst;
        switch (o->lse_kch) {
        case 1: lst = lse_launch_onepass<1>(st, o->lse_G, a, o->lwgms, o->lwgg); break;
        case 2: lst = lse_launch_onepass<2>(st, o->lse_G, a, o->lwgms, o->lwgg); break;
        case 4: lst = lse_launch_onepass<4>(st, o->lse_G, a, o->lwgms, o->lwgg); break;
        case 8: lst = lse_launch_onepass<8>(st, o->lse_G, a, o->lwgms, o->lwgg); break;
        default: lst = lse_launch_onepass<16>(st, o->lse_G, a, o->lwgms, o->lwgg); break;
        }
        if (lst < 0) { o->lse_two_pass = 1; return lse_enqueue_eval(o, x_dev, f_dev, g_dev); }
        hipLaunchKernelGGL(lse_combine_kernel, dim3((a.n_pad + 63) / 64), dim3(256), 0, st, a, o->lse_G, o->lwgms, o->lwgg, o->lms);
        HIPCHK(hipGetLastError());
        const XchgItem items[2] = {{o->lgall, (size_t)a.n_pad}, {o->lms, 2}};
        c->n_xchg_vector++;
        QNCHK(exchange_group(c, items, 2));
        hipLaunchKernelGGL(lse_finish1_kernel, dim3(std::min(1024, (a.n_pad + 255) / 256)), dim3(256), 0, st, a, o->lms);
        HIPCHK(hipGetLastError());
        return QN_OK;
    }
    // pass 1: z = A_rows x (the H-pass kernel in plain mat-vec mode)
    QnHPassArgs h{};
    h.H = o->Q; h.T = o->TA; h.T.n_pad = o->T.n_pad; h.T.n = (int)o->n; h.T.cs = 1;
    h.T.rpr = o->TA.rpr; h.T.row_off = 0; // row guards are not needed for a read-only pass
    h.hp = o->lz; h.expect_phase = -1; h.force_nrhs = 1; h.force_pending = 0; h.r0 = x_dev; h.r1 = x_dev;
    h.sp = x_dev; h.up = x_dev;
    launch_hpass<8>(st, h);
    HIPCHK(hipGetLastError());
    c->n_xchg_vector++;
    QNCHK(exchange(c, o->lz, 2 * (size_t)o->TA.rpr));
    QnLseArgs a{};
    a.A = o->Q; a.c = o->b; a.z = o->lz; a.w = o->lw; a.gpart = o->lgpart; a.gall = o->lgall; a.x = x_dev;
    a.f_out = f_dev; a.g_out = g_dev; a.scal = o->lscal; a.mu = o->mu;
    a.m = (int)o->m; a.m_pad = o->TA.n_pad; a.mrpr = o->TA.rpr; a.n = (int)o->n; a.n_pad = o->T.n_pad;
    a.world = c->world; a.rank = c->rank; a.rs = o->lse_rs;
    hipLaunchKernelGGL(lse_softmax_kernel, dim3(1), dim3(1024), 0, st, a);
    // pass 2: column sums A'w over this rank's rows
    hipLaunchKernelGGL(lse_colsum_kernel, dim3((a.n_pad + QN_CHUNK - 1) / QN_CHUNK, a.rs), dim3(QN_TPB), 0, st, a);
    hipLaunchKernelGGL(lse_reduce_splits_kernel, dim3(std::min(1024, (a.n_pad + 255) / 256)), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    c->n_xchg_vector++;
    QNCHK(exchange(c, o->lgall, (size_t)a.n_pad));
    hipLaunchKernelGGL(lse_finish_kernel, dim3(std::min(1024, (a.n_pad + 255) / 256)), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return QN_OK;
}

extern "C" void qn_objective_destroy(qn_objective* o) {
    if (!o) return;
    (void)hipSetDevice(o->ctx->device);
    (void)hipFree(o->Q); (void)hipFree(o->b);
    (void)hipFree(o->ex); (void)hipFree(o->eq); (void)hipFree(o->eg); (void)hipFree(o->ef);
    (void)hipFree(o->lz); (void)hipFree(o->lw); (void)hipFree(o->lgpart); (void)hipFree(o->lgall); (void)hipFree(o->lscal);
    (void)hipFree(o->lwgms); (void)hipFree(o->lwgg); (void)hipFree(o->lms);
    delete o;
}

extern "C" int qn_objective_get_rows(qn_objective* o, size_t row0, size_t nrows, double* out_host) {
    const bool lse = o->kind == OBJ_LOGSUMEXP;
    const size_t lo = lse ? (size_t)o->TA.row_off : (size_t)o->T.row_off;
    const size_t hi = lse ? std::min(o->m, lo + (size_t)o->TA.rpr) : std::min(o->n, lo + (size_t)o->T.rpr);
    if (row0 < lo || row0 + nrows > hi) return fail(QN_ERROR_INPUT_PARAMS, "rows outside this rank's shard");
    HIPCHK(hipSetDevice(o->ctx->device));
    HIPCHK(hipMemcpy2D(out_host, o->n * sizeof(double), o->Q + (row0 - lo) * (size_t)o->T.n_pad, (size_t)o->T.n_pad * sizeof(double),
                       o->n * sizeof(double), nrows, hipMemcpyDeviceToHost));
    return QN_OK;
}

template <int R>
static void launch_quad(hipStream_t st, const QnQuadArgs& a) {
    dim3 grid(a.T.rpr / R, a.T.cs);
    hipLaunchKernelGGL(quad_matvec_kernel<R>, grid, dim3(QN_TPB), 0, st, a);
}
static int launch_quad_R(int R, hipStream_t st, const QnQuadArgs& a) {
    switch (R) {
    case 4: launch_quad<4>(st, a); break;
    case 16: launch_quad<16>(st, a); break;
    default: launch_quad<8>(st, a); break;
    }
    HIPCHK(hipGetLastError());
    return QN_OK;
}

__global__ __launch_bounds__(QN_CTL_TPB) void quad_finish_kernel(const QnVecs V, double* f_out, double* g_out) {
    __shared__ double lds[32];
    double p[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < V.n_pad; i += QN_CTL_TPB) {
        const double qi = q_val(V, i), xi = V.xt[i], bi = V.b[i];
        p[0] = __builtin_fma(xi, qi, p[0]);
        p[1] = __builtin_fma(bi, xi, p[1]);
        g_out[i] = qi - bi;
    }
    ctl_block_sum<2>(p, lds);
    if (threadIdx.x == 0) *f_out = 0.5 * p[0] - p[1];
}

extern "C" int qn_objective_eval(qn_objective* o, const double* x_host, double* f, double* g_host) {
    qn_context* c = o->ctx;
    HIPCHK(hipSetDevice(c->device));
    const size_t np = o->T.n_pad;
    if (o->kind == OBJ_LOGSUMEXP) {
        if (!o->ex) {
            QNCHK(dev_alloc_zero(&o->ex, 2 * np, c->stream));
            QNCHK(dev_alloc_zero(&o->eg, np, c->stream));
            QNCHK(dev_alloc_zero(&o->ef, 2, c->stream));
        }
        HIPCHK(hipMemcpyAsync(o->ex, x_host, o->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        QNCHK(lse_enqueue_eval(o, o->ex, o->ef, o->eg));
        HIPCHK(hipMemcpyAsync(f, o->ef, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(g_host, o->eg, o->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return QN_OK;
    }
    if (o->kind != OBJ_QUADRATIC) return fail(QN_ERROR_INPUT_PARAMS, "unsupported objective");
    if (!o->ex) {
        QNCHK(dev_alloc_zero(&o->ex, 2 * np, c->stream)); // x and xt
        QNCHK(dev_alloc_zero(&o->eq, np, c->stream));
        QNCHK(dev_alloc_zero(&o->eg, np, c->stream));
        QNCHK(dev_alloc_zero(&o->ef, 2, c->stream));
    }
    HIPCHK(hipMemcpyAsync(o->ex, x_host, o->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    QnQuadArgs a{};
    a.Q = o->Q; a.T = o->T; a.x = o->ex; a.d = o->ex; a.xt = o->ex + np;
    a.out = o->eq + (size_t)c->rank * o->T.rpr;
    a.ctl = nullptr; a.expect_phase = -1; a.force_kind = QN_REQ_X; a.force_t = 0.0;
    QNCHK(launch_quad_R(8, c->stream, a));
    QNCHK(exchange(c, o->eq, (size_t)o->T.rpr));
    QnVecs V{};
    V.q = o->eq; V.xt = o->ex + np; V.b = o->b; V.n = (int)o->n; V.n_pad = (int)np; V.rpr = o->T.rpr; V.world = c->world; V.qcs = 1; V.hcs = 1;
    hipLaunchKernelGGL(quad_finish_kernel, dim3(1), dim3(QN_CTL_TPB), 0, c->stream, V, o->ef, o->eg);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(f, o->ef, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(g_host, o->eg, o->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return QN_OK;
}
