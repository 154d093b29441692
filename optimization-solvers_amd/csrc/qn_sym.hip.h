// qn_sym.hip.h -- the fused fast path on the UPPER BLOCK TRIANGLE of the two symmetric matrices (single rank).
//
// The inverse Hessian approximation is symmetric bit for bit (the rank-2 update's inner sums commute) and so is the
// benchmark objective's Q, so only the 128 x 128 tiles (I, J >= I) are streamed: half the bytes of the row kernels of
// qn_fused.hip.h.  Tile (I, J) contributes
//     row part     r_i += sum_{j in J} M_ij v_j      (i in I)        -> slot J of block-row I
//     column part  r_j += sum_{i in I} M_ij v_i      (j in J, J > I) -> slot I of block-row J     (M_ji = M_ij)
// into part[R][k][rhs][128]: block-row R receives exactly nb slots, each written by exactly one tile.  A second, small
// kernel per pass sums the slots of its block-row in slot order (fixed order: bitwise reproducible) and runs the very
// epilogue the row kernels run (g+, y, u, v, the per-block partial sums for the control step, the vector commits).
// The lower block triangle of H is NOT maintained during a run; qn_minimize mirrors it back before it returns.
// Measured (tools/sym_probe.hip): update pass 27.6 us + 5 us reduce at n = 4096 (row kernel 47.6 us); 1.9x at n = 16384.
// Measured and dropped: summing the slots inside the tile kernel (the last tile to arrive at a block-row, told by an
// agent-scope arrival counter, reads the slots with sc1 loads and runs the epilogue) instead of the second launch.  Bitwise
// identical results, but every tile then ends in `s_waitcnt vmcnt(0)` + barrier + one or two returning atomics, and the
// grid at n = 4096 is a single wave of workgroups, so all of it is exposed: tile kernels +8-9 us against ~6 us for the
// launch saved (9.1 k vs 9.5 k it/s at n = 4096; 211 vs 286 it/s at n = 32768).  Likewise dropped: the evaluation's slot
// reduction and the control step in one launch (last of the nb reduce workgroups runs the step): 9.1 k vs 9.9 k it/s.
#pragma once

#define QN_TB 128
#define QN_SYM_WAVES 8                       // waves per tile workgroup (512 threads)
#define QN_SYM_RPW (QN_TB / QN_SYM_WAVES)     // rows of the tile per wave
#define QN_SYM_TPB (64 * QN_SYM_WAVES)
// Cache policy of the streamed tiles, chosen per run (`nt`): at n = 4096 the two half matrices (2 x 64 MiB) fit the 256 MiB
// Infinity Cache and plain accesses win (+2 %); past it (n >= 8192) every byte is touched once per pass and non-temporal
// loads / stores win (n = 32768: 285 -> 313 it/s).
template <bool NT> __device__ __forceinline__ v2d qn_sym_ld(const double* p) {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p)) : ld2(p);
}
template <bool NT> __device__ __forceinline__ void qn_sym_st(double* p, v2d v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(p)); else st2(p, v);
}

// Row-sharded runs (one process per GPU): rank p stores whole rows of its `nbl` block-rows [ioff, ioff + nbl) and streams the
// CIRCULANT half of them -- block-row I owns the tiles (I, (I + k) mod nb), k = 0 .. cnt(I) - 1, with cnt = (nb + 1) / 2 for odd
// nb; for even nb the pair {I, I + nb / 2} has two candidates, and it goes to I when I mod (nb / 2) is even, to I + nb / 2 when it
// is odd (cnt = nb / 2 + 1 or nb / 2): every unordered pair {I, J} exactly once, the same number of tiles per block-row to within
// one -- and, because the long windows ALTERNATE along the block-rows, the same number of tiles per RANK to within one (rounds 2-3
// gave all the long windows to I < nb / 2: at P = 8, n = 32768 ranks 0-3 streamed 4128 tiles and ranks 4-7 4096, and an
// iteration takes as long as its slowest rank).  A tile's column part lands in a block-row that another rank may own, so each rank sums what ITS
// tiles contributed to every block-row (symsh_sum_kernel), the per-rank partial n-vectors are all-gathered (the context's
// exchange: RCCL or host-staged), and every rank adds them in rank order and runs the epilogue on the full vectors
// (symsh_*_epi_kernel) -- replicated work on n-vectors, identical bits on every rank.  world == 1: the fields are unused.
struct QnSymShard {
    int world, rank, nbl, ioff;
    int nsum;   // slices of xg to add up after the exchange: world (all-gather, rank order), or 1 (an all-reduce left the total in slice 0)
    double* xg; // gathered partial sums: [world][nrhs][n_pad], rank r's slice written by rank r
    const int* tiles;           // this rank's tiles in launch order, (I << 16) | J: the local block-rows in order, each with its window (built by the host)
    const int *sl_off, *sl_idx; // the slots of block-row R that THIS rank's tiles write: sl_idx[sl_off[R] .. sl_off[R + 1]), ascending (built by the host)
};
__device__ __host__ __forceinline__ int qn_symsh_cnt(int I, int nb) {
    if (nb & 1) return (nb + 1) / 2;
    const int h = nb / 2;
    const bool longw = (I < h) ? ((I & 1) == 0) : (((I - h) & 1) == 1); // of the pair {I0, I0 + h}: I0 when I0 is even, I0 + h when it is odd
    return longw ? h + 1 : h;
}
__device__ __host__ __forceinline__ bool qn_symsh_owns(int I, int J, int nb) { // is tile (I, J) in block-row I's circulant window?
    int k = J - I;
    if (k < 0) k += nb;
    return k < qn_symsh_cnt(I, nb);
}
static inline int qn_symsh_ntiles(int nb, int nbl, int ioff) {
    int t = 0;
    for (int il = 0; il < nbl; ++il) t += qn_symsh_cnt(ioff + il, nb);
    return t;
}

struct QnSymEvalArgs {
    const double* Q;
    QnTile T;
    QnFused F;
    const QnCtl* ctl;
    int expect_phase;
    int after_h;
    int nb;
    double* part; // [nb][nb][2][QN_TB]
    int nt;       // non-temporal tile loads
    QnSymShard sh;
};
struct QnSymHPassArgs {
    double* H;
    QnTile T;
    QnFused F;
    const QnCtl* ctl;
    int expect_phase;
    int nb;
    double* part;
    // generic path (host / device closures, log-sum-exp objective, SR1, bounded variants): fixed vector buffers, the sums go to the
    // gathered h_pass output the control step reads (qn_kernels.hip.h h_pass_kernel), no epilogue
    int generic;
    const double *gsp, *gup, *gvy, *gvg;
    double* ghp;
    int nt; // non-temporal tile loads and stores
    QnSymShard sh;
    int need_serviced; // != 0 (generic objectives on the second-generation structure, qn_sym2g.hip.h): run only if the control block says
                       // `serviced == need_serviced` -- the one-workgroup launch in front has marked the request as this kernel's
};

// launch-linear index t -> upper-triangle tile (I, J >= I), row-major over I
__device__ __forceinline__ void qn_sym_tile(int t, int nb, int& I, int& J) {
    const double b = 2.0 * nb + 1.0;
    int i = (int)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    if (i < 0) i = 0;
    if (i > nb - 1) i = nb - 1;
    while (i > 0 && i * nb - i * (i - 1) / 2 > t) --i;
    while ((i + 1) * nb - (i + 1) * i / 2 <= t) ++i;
    I = i;
    J = i + (t - (i * nb - i * (i - 1) / 2));
}

// the evaluation request, decoded exactly as quad_eval_fused_kernel decodes it (incl. the deferred update's coefficients)
struct QnEvalReq {
    double t, c_ss, c_su, c_uu, ug, sg;
    double al, be; // second-generation path only: d = -(v + al s + be u), al = c_su (u.g) + c_ss (s.g), be = c_su (s.g) + c_uu (u.g)
    int mode, xc, sc;
    bool is_t;
};
// returns false when the launch is predicated off; contains barriers: call from uniform control flow, 256 threads
__device__ __forceinline__ bool qn_eval_request(const QnCtl* __restrict__ ctl, const QnFused& F, int expect_phase, int after_h, double* red4,
                                                QnEvalReq& q) {
    const int phase = ctl->phase;
    const bool post_h = after_h && phase == QN_PH_REQ_HPASS_EVAL;
    if (phase != expect_phase && !post_h) return false;
    q.is_t = ctl->req_kind == QN_REQ_T;
    q.t = ctl->req_t;
    q.mode = ctl->dir_mode;
    q.c_ss = ctl->c_ss; q.c_su = ctl->c_su; q.c_uu = ctl->c_uu; q.ug = ctl->dir_ug; q.sg = ctl->dir_sg;
    q.xc = ctl->xc;
    q.sc = ctl->sc;
    if (post_h) { // same derivation as quad_eval_fused_kernel: the control step commits the very same values afterwards
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (wave < QN_NHPP) { const double v = qn_partial_col_sum(F.hpp, 1, F.nblk, QN_NHPP, wave, lane); if (lane == 0) red4[wave] = v; }
        __syncthreads();
        const double yu = red4[0];
        q.ug = red4[1]; q.sg = red4[2];
        __syncthreads();
        qn_update_coeffs(ctl->method, ctl->ys, yu, q.c_ss, q.c_su, q.c_uu);
        q.mode = 1;
        q.sc ^= 1;
    }
    return true;
}
__device__ __forceinline__ double qn_trial_entry(const QnEvalReq& q, const double* __restrict__ x, const double* __restrict__ vv,
                                                 const double* __restrict__ sp, const double* __restrict__ un, int idx, double* d_out = nullptr) {
    const double xi = x[idx];
    if (!q.is_t) { if (d_out) *d_out = 0.0; return xi; }
    const double d = qn_dir1(q.mode, vv[idx], q.mode ? sp[idx] : 0.0, q.mode ? un[idx] : 0.0, q.c_ss, q.c_su, q.c_uu, q.ug, q.sg);
    if (d_out) *d_out = d;
    const double td = q.t * d; // `step * direction` rounds first (bfgs.rs:94)
    return xi + td;
}

// Sum of the nb slots of (block-row R, rhs) for row `i` of the block: the two halves of the workgroup (threads 0..127 and
// 128..255) each add one half of the slot range in slot order, 16 loads in flight, and the halves are combined through LDS
// -- a fixed order.  All 256 threads call it; threads 0..127 get the total of their row (i = tid & 127).
__device__ __forceinline__ double qn_sym_slot_sum(const double* __restrict__ part, int nb, int R, int rhs, double* halfbuf /* LDS[128] */) {
    const int i = threadIdx.x & (QN_TB - 1), half = threadIdx.x >> 7;
    const int nh = (nb + 1) / 2;
    const int k_lo = half * nh, k_hi = (half == 0) ? nh : nb;
    const double* p = part + (((size_t)R * nb) * 2 + rhs) * QN_TB + i;
    double acc = 0.0;
    for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (k0 + q < k_hi) ? p[(size_t)(k0 + q) * 2 * QN_TB] : 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = acc + v[q];
    }
    if (half == 1) halfbuf[i] = acc;
    __syncthreads();
    const double tot = (half == 0) ? acc + halfbuf[i] : 0.0;
    __syncthreads(); // halfbuf is reused by the next call
    return tot;
}
// per-block totals of NP values held by threads 0..127 (threads >= 128 pass zeros): fixed order; thread k < NP gets total k
template <int NP>
__device__ __forceinline__ double qn_sym_block_totals(double (&p)[16], double (*red)[16]) {
    static_assert(NP <= 16, "at most 16 partial sums");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    QnWaveFold<16, 32>::run(p, lane); // lanes with (lane & 3) == 0 hold value index lane >> 2
    if ((lane & 3) == 0) red[wave][lane >> 2] = p[0];
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < NP) tot = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    return tot;
}

// ------------------------------------------------------------------------------------------------
// evaluation: q = Q (x + t d) from the upper block triangle
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(QN_SYM_TPB) void sym_eval_tile_kernel(const QnSymEvalArgs a) {
    __shared__ double red4[4];
    __shared__ double rowx[QN_TB];
    __shared__ double colred[QN_SYM_WAVES][QN_TB];
    int I, J;
    const bool sharded = a.sh.world > 1;
    if (sharded) { const int ij = a.sh.tiles[blockIdx.x]; I = ij >> 16; J = ij & 0xffff; } else qn_sym_tile(blockIdx.x, a.nb, I, J);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t np = (size_t)a.T.n_pad;
    const int i0 = I * QN_TB, j0 = J * QN_TB, jc = j0 + 2 * lane;
    const int i0l = i0 - (sharded ? a.sh.ioff * QN_TB : 0); // row of the tile in this rank's storage
    const double* qbase = a.Q + (size_t)(i0l + wave * QN_SYM_RPW) * np + jc;
    v2d h[QN_SYM_RPW]; // the wave's whole share of the tile is requested before the control block is read: the grid is a single
                       // wave of workgroups at n = 4096, so nothing else would hide the prologue's two dependent round trips
#pragma unroll
    for (int r = 0; r < QN_SYM_RPW; ++r) h[r] = a.nt ? qn_sym_ld<true>(qbase + (size_t)r * np) : qn_sym_ld<false>(qbase + (size_t)r * np);
    QnEvalReq q;
    if (!qn_eval_request(a.ctl, a.F, a.expect_phase, a.after_h, red4, q)) return;
    const double* __restrict__ x = a.F.X0 + (size_t)q.xc * np;
    const double* __restrict__ sp = a.F.S0 + (size_t)q.sc * np;
    if (tid < QN_TB) rowx[tid] = qn_trial_entry(q, x, a.F.VV, sp, a.F.UN, i0 + tid);
    v2d xtj;
    xtj.x = qn_trial_entry(q, x, a.F.VV, sp, a.F.UN, jc);
    xtj.y = qn_trial_entry(q, x, a.F.VV, sp, a.F.UN, jc + 1);
    __syncthreads();
    double cx = 0.0, cy = 0.0;
#pragma unroll
    for (int rc = 0; rc < QN_SYM_RPW; rc += 8) {
        double racc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const v2d hv = h[rc + r];
            double t0 = hv.x * xtj.x;
            t0 = __builtin_fma(hv.y, xtj.y, t0);
            racc[r] = t0;
            const double xi = rowx[wave * QN_SYM_RPW + rc + r];
            cx = __builtin_fma(hv.x, xi, cx);
            cy = __builtin_fma(hv.y, xi, cy);
        }
        QnWaveFold<8, 32>::run(racc, lane); // lanes with (lane & 7) == 0 hold row lane >> 3
        if ((lane & 7) == 0) a.part[(((size_t)I * a.nb + J) * 2 + 0) * QN_TB + wave * QN_SYM_RPW + rc + (lane >> 3)] = racc[0];
    }
    if (J != I) { // (single rank: J > I; sharded: the window wraps around)
        colred[wave][2 * lane] = cx;
        colred[wave][2 * lane + 1] = cy;
        __syncthreads();
        if (tid < QN_TB) {
            double acc = colred[0][tid];
#pragma unroll
            for (int w = 1; w < QN_SYM_WAVES; ++w) acc = acc + colred[w][tid];
            a.part[(((size_t)J * a.nb + I) * 2 + 0) * QN_TB + tid] = acc;
        }
    }
}

// the epilogue of an evaluation for block-row R given q_i (threads 0..127; all 256 threads call it)
__device__ __forceinline__ void qn_sym_eval_epilogue(const QnSymEvalArgs& a, const QnEvalReq& q, int R, double qi, double (*red)[16]) {
    const int tid = threadIdx.x;
    const size_t np = (size_t)a.T.n_pad;
    const double* __restrict__ x = a.F.X0 + (size_t)q.xc * np;
    double* __restrict__ xt = a.F.X0 + (size_t)(1 - q.xc) * np;
    const double* __restrict__ sp = a.F.S0 + (size_t)q.sc * np;
    double* __restrict__ sstage = a.F.S0 + (size_t)(1 - q.sc) * np;
    double p[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] = 0.0;
    if (tid < QN_TB) {
        const int gi = R * QN_TB + tid;
        double di;
        const double xi = x[gi];
        const double xti = qn_trial_entry(q, x, a.F.VV, sp, a.F.UN, gi, &di);
        const double bi = a.F.b[gi], go = a.F.G[gi];
        const double gti = qi - bi;
        const double yi = gti - go;
        const double si = xti - xi; // s = x+ - x (bfgs.rs:96)
        a.F.GT[gi] = gti;
        a.F.Y[gi] = yi;
        xt[gi] = xti;
        sstage[gi] = si;
        a.F.UP[gi] = a.F.UN[gi]; // refresh the pending-u copy read by the next H pass
        p[0] = xti * qi;
        p[1] = bi * xti;
        p[2] = gti * di;
        p[3] = go * di;
        p[4] = yi * yi;
        p[5] = yi * si;
        p[6] = gti * gti;
        p[7] = si * si;
        p[8] = isfinite(di) ? 0.0 : 1.0;
    }
    const double tot = qn_sym_block_totals<QN_NEVP>(p, red);
    if (tid < QN_NEVP) a.F.evp[(size_t)tid * a.F.nblk + R] = tot;
}

// block-row R: q_i = sum of its slots, then the epilogue of quad_eval_fused_kernel for these 128 rows
__global__ __launch_bounds__(256) void sym_eval_reduce_kernel(const QnSymEvalArgs a) {
    __shared__ double red4[4];
    __shared__ double red[4][16];
    __shared__ double halfbuf[QN_TB];
    const int R = blockIdx.x;
    QnEvalReq q;
    if (!qn_eval_request(a.ctl, a.F, a.expect_phase, a.after_h, red4, q)) return;
    const double qi = qn_sym_slot_sum(a.part, a.nb, R, 0, halfbuf);
    qn_sym_eval_epilogue(a, q, R, qi, red);
}

// ---- row-sharded runs: what THIS rank's tiles contributed to block-row R, summed in a fixed order ----
// The slots this rank wrote for block-row R: its own window (R local) and the column parts of the local block-rows whose
// window contains R -- an ascending list per block-row that the host builds once (QnSymShard.sl_off / sl_idx; rounds 2-3
// compacted it in 512 words of LDS at every launch, which capped nb at 512).  The two halves of the workgroup add one half of the
// list each, 16 loads in flight, exactly as qn_sym_slot_sum does.  All 256 threads call it; threads 0..127 get the total of row
// tid (zero when the rank contributed nothing).
__device__ __forceinline__ double qn_symsh_slot_sum(const double* __restrict__ part, int nb, int R, int rhs, const int* list, int nlist,
                                                    double* halfbuf /* LDS[128] */) {
    const int i = threadIdx.x & (QN_TB - 1), half = threadIdx.x >> 7;
    const int nh = (nlist + 1) / 2;
    const int k_lo = half * nh, k_hi = (half == 0) ? nh : nlist;
    const double* p = part + (((size_t)R * nb) * 2 + rhs) * QN_TB + i;
    double acc = 0.0;
    for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (k0 + q < k_hi) ? p[(size_t)list[k0 + q] * 2 * QN_TB] : 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = acc + v[q];
    }
    if (half == 1) halfbuf[i] = acc;
    __syncthreads();
    const double tot = (half == 0) ? acc + halfbuf[i] : 0.0;
    __syncthreads();
    return tot;
}
// sum over the ranks, in rank order, of the gathered partial sums for row gi of right-hand side `rhs` (nrhs per rank)
__device__ __forceinline__ double qn_symsh_rank_sum(const QnSymShard& sh, int nrhs, int rhs, size_t np, int gi) {
    const double* p = sh.xg + (size_t)rhs * np + gi;
    double acc = p[0];
    for (int r = 1; r < sh.nsum; ++r) acc = acc + p[(size_t)r * nrhs * np];
    return acc;
}

// evaluation, block-row R: this rank's partial q for rows R*128 .. into its slice of the gather buffer
__global__ __launch_bounds__(256) void symsh_eval_sum_kernel(const QnSymEvalArgs a) {
    __shared__ double halfbuf[QN_TB];
    const int phase = a.ctl->phase;
    if (phase != a.expect_phase && !(a.after_h && phase == QN_PH_REQ_HPASS_EVAL)) return;
    const int R = blockIdx.x;
    const int* list = a.sh.sl_idx + a.sh.sl_off[R];
    const int nlist = a.sh.sl_off[R + 1] - a.sh.sl_off[R];
    const double qi = qn_symsh_slot_sum(a.part, a.nb, R, 0, list, nlist, halfbuf);
    if (threadIdx.x < QN_TB) a.sh.xg[(size_t)a.sh.rank * a.T.n_pad + (size_t)R * QN_TB + threadIdx.x] = qi;
}
// evaluation, block-row R, after the exchange: q_i = sum over ranks, then the epilogue (every rank, all block-rows)
__global__ __launch_bounds__(256) void symsh_eval_epi_kernel(const QnSymEvalArgs a) {
    __shared__ double red4[4];
    __shared__ double red[4][16];
    const int R = blockIdx.x;
    QnEvalReq q;
    if (!qn_eval_request(a.ctl, a.F, a.expect_phase, a.after_h, red4, q)) return;
    const double qi = threadIdx.x < QN_TB ? qn_symsh_rank_sum(a.sh, 1, 0, (size_t)a.T.n_pad, R * QN_TB + (int)threadIdx.x) : 0.0;
    qn_sym_eval_epilogue(a, q, R, qi, red);
}

// ------------------------------------------------------------------------------------------------
// H pass: pending rank-2 update of the tile, row and column dots with [y, g+] (update pass) or [g] (direction pass)
// ------------------------------------------------------------------------------------------------
template <int NRHS, bool PENDING, bool NT>
__device__ __forceinline__ void sym_hpass_tile_body(const QnSymHPassArgs& a, int I, int J, const double* __restrict__ sp, const double* __restrict__ up,
                                                    const double* __restrict__ r0v, const double* __restrict__ gt, double c_ss, double c_su,
                                                    double c_uu, v2d (&h)[8], double (*rowv)[QN_TB], double (*colred)[2][QN_TB]) { // colred[QN_SYM_WAVES][2][QN_TB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t np = (size_t)a.T.n_pad;
    const int n = a.T.n;
    const int i0 = I * QN_TB, j0 = J * QN_TB, jc = j0 + 2 * lane;
    const int i0l = i0 - (a.sh.world > 1 ? a.sh.ioff * QN_TB : 0); // row of the tile in this rank's storage
    // r0v: update pass rhs0 = y (rhs1 = gt = g+); direction pass rhs0 = g
    if (tid < QN_TB) {
        rowv[0][tid] = PENDING ? sp[i0 + tid] : 0.0;
        rowv[1][tid] = PENDING ? up[i0 + tid] : 0.0;
        rowv[2][tid] = r0v[i0 + tid];
        rowv[3][tid] = (NRHS == 2) ? gt[i0 + tid] : 0.0;
    }
    v2d sj = {0.0, 0.0}, uj = {0.0, 0.0};
    if (PENDING) { sj = ld2(sp + jc); uj = ld2(up + jc); }
    const v2d a0 = ld2(r0v + jc);
    const v2d a1 = (NRHS == 2) ? ld2(gt + jc) : (v2d){0.0, 0.0};
    const bool use_su = c_su != 0.0, use_uu = c_uu != 0.0;
    const bool c0ok = jc < n, c1ok = (jc + 1) < n;
    __syncthreads();
    double c0x = 0.0, c0y = 0.0, c1x = 0.0, c1y = 0.0;
    double* hbase = a.H + (size_t)(i0l + wave * QN_SYM_RPW) * np + jc;
    for (int rc = 0; rc < QN_SYM_RPW; rc += 8) {
        if (rc) {
#pragma unroll
            for (int r = 0; r < 8; ++r) h[r] = qn_sym_ld<NT>(hbase + (size_t)(rc + r) * np);
        }
        double racc[8 * NRHS];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int ri = wave * QN_SYM_RPW + rc + r;
            v2d hn = h[r];
            if (PENDING) {
                const double si = rowv[0][ri], ui = rowv[1][ri];
                if (use_su) {
                    hn.x = hn.x + c_su * (si * uj.x + ui * sj.x);
                    hn.y = hn.y + c_su * (si * uj.y + ui * sj.y);
                }
                hn.x = hn.x + c_ss * (si * sj.x);
                hn.y = hn.y + c_ss * (si * sj.y);
                if (use_uu) {
                    hn.x = hn.x + c_uu * (ui * uj.x);
                    hn.y = hn.y + c_uu * (ui * uj.y);
                }
                const bool rowok = (i0 + ri) < n;
                hn.x = (rowok && c0ok) ? hn.x : 0.0;
                hn.y = (rowok && c1ok) ? hn.y : 0.0;
                qn_sym_st<NT>(hbase + (size_t)(rc + r) * np, hn);
            }
            double t0 = hn.x * a0.x;
            t0 = __builtin_fma(hn.y, a0.y, t0);
            racc[r] = t0;
            const double y0 = rowv[2][ri];
            c0x = __builtin_fma(hn.x, y0, c0x);
            c0y = __builtin_fma(hn.y, y0, c0y);
            if (NRHS == 2) {
                double t1 = hn.x * a1.x;
                t1 = __builtin_fma(hn.y, a1.y, t1);
                racc[8 + r] = t1;
                const double y1 = rowv[3][ri];
                c1x = __builtin_fma(hn.x, y1, c1x);
                c1y = __builtin_fma(hn.y, y1, c1y);
            }
        }
        QnWaveFold<8 * NRHS, 32>::run(racc, lane);
        constexpr int SH = (NRHS == 2) ? 2 : 3; // 16 values -> index lane >> 2 ; 8 values -> index lane >> 3
        if ((lane & ((1 << SH) - 1)) == 0) {
            const int idx = lane >> SH, rhs = idx >> 3, r = idx & 7;
            a.part[(((size_t)I * a.nb + J) * 2 + rhs) * QN_TB + wave * QN_SYM_RPW + rc + r] = racc[0];
        }
    }
    if (J != I) { // (single rank: J > I; sharded: the window wraps around)
        colred[wave][0][2 * lane] = c0x;
        colred[wave][0][2 * lane + 1] = c0y;
        if (NRHS == 2) { colred[wave][1][2 * lane] = c1x; colred[wave][1][2 * lane + 1] = c1y; }
        __syncthreads();
        if (tid < NRHS * QN_TB) {
            const int rhs = tid / QN_TB, c = tid % QN_TB;
            double acc = colred[0][rhs][c];
#pragma unroll
            for (int w = 1; w < QN_SYM_WAVES; ++w) acc = acc + colred[w][rhs][c];
            a.part[(((size_t)J * a.nb + I) * 2 + rhs) * QN_TB + c] = acc;
        }
    }
}

__global__ __launch_bounds__(QN_SYM_TPB) void sym_hpass_tile_kernel(const QnSymHPassArgs a) {
    __shared__ double rowv[4][QN_TB];
    __shared__ double colred[QN_SYM_WAVES][2][QN_TB];
    int I, J;
    if (a.sh.world > 1) { const int ij = a.sh.tiles[blockIdx.x]; I = ij >> 16; J = ij & 0xffff; } else qn_sym_tile(blockIdx.x, a.nb, I, J);
    const size_t np = (size_t)a.T.n_pad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2d h[8]; // the first rows are requested before the control block is read (all 32, as in the evaluation kernel, costs the
              // update kernel its second workgroup per CU: 256 VGPRs, 42 us instead of 34 at n = 4096)
    {
        const int il = I - (a.sh.world > 1 ? a.sh.ioff : 0);
        const double* hbase = a.H + (size_t)(il * QN_TB + wave * QN_SYM_RPW) * np + J * QN_TB + 2 * lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) h[r] = a.nt ? qn_sym_ld<true>(hbase + (size_t)r * np) : qn_sym_ld<false>(hbase + (size_t)r * np);
    }
    const QnCtl* __restrict__ ctl = a.ctl;
    const int phase = ctl->phase;
    if (phase != a.expect_phase && !(phase == QN_PH_REQ_HPASS_EVAL && !a.generic)) return;
    if (a.need_serviced && ctl->serviced != a.need_serviced) return; // (uniform)
    const int nrhs = ctl->hp_nrhs; // generic path: 0 = apply the pending update only (the sums are then not used)
    const int pending = ctl->pending;
    const double c_ss = ctl->c_ss, c_su = ctl->c_su, c_uu = ctl->c_uu;
    const double *sp, *up, *r0v, *gt;
    if (a.generic) {
        sp = a.gsp; up = a.gup; gt = a.gvg;
        r0v = (ctl->after_state == QN_ST_AFTER_DIR) ? a.gvg : a.gvy; // as h_pass_kernel: rhs0 = g for a direction pass, y for an update pass (1 or 2 rhs)
    } else {
        sp = a.F.S0 + (size_t)ctl->sc * np; up = a.F.UP; gt = a.F.GT;
        r0v = (nrhs == 2) ? a.F.Y : a.F.GT;
    }
#define QN_SYM_BODY(NR, PE, NTF) sym_hpass_tile_body<NR, PE, NTF>(a, I, J, sp, up, r0v, gt, c_ss, c_su, c_uu, h, rowv, colred)
    if (a.nt) {
        if (pending) { if (nrhs == 2) QN_SYM_BODY(2, true, true); else QN_SYM_BODY(1, true, true); }
        else { if (nrhs == 2) QN_SYM_BODY(2, false, true); else QN_SYM_BODY(1, false, true); }
    } else {
        if (pending) { if (nrhs == 2) QN_SYM_BODY(2, true, false); else QN_SYM_BODY(1, true, false); }
        else { if (nrhs == 2) QN_SYM_BODY(2, false, false); else QN_SYM_BODY(1, false, false); }
    }
#undef QN_SYM_BODY
}

// the epilogue of an H pass for block-row R given the sums of its two right-hand sides (threads 0..127; all 256 threads call it)
__device__ __forceinline__ void qn_sym_hpass_epilogue(const QnSymHPassArgs& a, int sc, int nrhs, int R, double tot0, double tot1, double (*red)[16]) {
    const int tid = threadIdx.x;
    const size_t np = (size_t)a.T.n_pad;
    const double* sstage = a.F.S0 + (size_t)(1 - sc) * np;
    double p[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] = 0.0;
    if (tid < QN_TB) {
        const int gi = R * QN_TB + tid;
        const double gp = a.F.GT[gi];
        if (nrhs == 2) {
            a.F.UN[gi] = tot0;
            a.F.VV[gi] = tot1;
            p[0] = a.F.Y[gi] * tot0; // y.u
            p[1] = tot0 * gp;        // u.g+
            p[2] = sstage[gi] * gp;  // s.g+
        } else {
            a.F.VV[gi] = tot0; // direction pass: v = H g
        }
        a.F.G[gi] = gp; // commit g <- g+
    }
    if (nrhs == 2) {
        const double tot = qn_sym_block_totals<QN_NHPP>(p, red);
        if (tid < QN_NHPP) a.F.hpp[(size_t)tid * a.F.nblk + R] = tot;
    }
}

// block-row R: u_i, v_i = sums of the slots; the epilogue of h_pass_fused_kernel for these 128 rows
__global__ __launch_bounds__(256) void sym_hpass_reduce_kernel(const QnSymHPassArgs a) {
    __shared__ double red[4][16];
    __shared__ double halfbuf[QN_TB];
    const int R = blockIdx.x, tid = threadIdx.x;
    const QnCtl* __restrict__ ctl = a.ctl;
    const int phase = ctl->phase;
    if (phase != a.expect_phase && !(phase == QN_PH_REQ_HPASS_EVAL && !a.generic)) return;
    const int nrhs = ctl->hp_nrhs;
    if (a.generic) { // the control step does the rest (qn_ctl_step.hip.h, states AFTER_DIR / AFTER_U)
        if (nrhs == 0) return;
        const double t0 = qn_sym_slot_sum(a.part, a.nb, R, 0, halfbuf);
        const double t1 = (nrhs == 2) ? qn_sym_slot_sum(a.part, a.nb, R, 1, halfbuf) : 0.0;
        if (tid < QN_TB) {
            a.ghp[(size_t)R * QN_TB + tid] = t0;
            if (nrhs == 2) a.ghp[(size_t)a.T.rpr + (size_t)R * QN_TB + tid] = t1;
        }
        return;
    }
    const double tot0 = qn_sym_slot_sum(a.part, a.nb, R, 0, halfbuf);
    const double tot1 = (nrhs == 2) ? qn_sym_slot_sum(a.part, a.nb, R, 1, halfbuf) : 0.0; // nrhs is uniform
    qn_sym_hpass_epilogue(a, ctl->sc, nrhs, R, tot0, tot1, red);
}
// row-sharded runs, H pass, block-row R: this rank's partial sums into its slice of the gather buffer [world][2][n_pad]
__global__ __launch_bounds__(256) void symsh_hpass_sum_kernel(const QnSymHPassArgs a) {
    __shared__ double halfbuf[QN_TB];
    const QnCtl* __restrict__ ctl = a.ctl;
    const int phase = ctl->phase;
    if (phase != a.expect_phase && phase != QN_PH_REQ_HPASS_EVAL) return;
    const int nrhs = ctl->hp_nrhs;
    const int R = blockIdx.x;
    const size_t np = (size_t)a.T.n_pad;
    const int* list = a.sh.sl_idx + a.sh.sl_off[R];
    const int nlist = a.sh.sl_off[R + 1] - a.sh.sl_off[R];
    const double t0 = qn_symsh_slot_sum(a.part, a.nb, R, 0, list, nlist, halfbuf);
    const double t1 = (nrhs == 2) ? qn_symsh_slot_sum(a.part, a.nb, R, 1, list, nlist, halfbuf) : 0.0;
    if (threadIdx.x < QN_TB) {
        double* out = a.sh.xg + (size_t)a.sh.rank * 2 * np + (size_t)R * QN_TB + threadIdx.x;
        out[0] = t0;
        out[np] = t1;
    }
}
// ... after the exchange: u_i, v_i = sums over the ranks in rank order, then the epilogue (every rank, all block-rows)
__global__ __launch_bounds__(256) void symsh_hpass_epi_kernel(const QnSymHPassArgs a) {
    __shared__ double red[4][16];
    const QnCtl* __restrict__ ctl = a.ctl;
    const int phase = ctl->phase;
    if (phase != a.expect_phase && phase != QN_PH_REQ_HPASS_EVAL) return;
    const int nrhs = ctl->hp_nrhs;
    const int R = blockIdx.x, gi = R * QN_TB + (int)threadIdx.x;
    const size_t np = (size_t)a.T.n_pad;
    double t0 = 0.0, t1 = 0.0;
    if (threadIdx.x < QN_TB) {
        t0 = qn_symsh_rank_sum(a.sh, 2, 0, np, gi);
        if (nrhs == 2) t1 = qn_symsh_rank_sum(a.sh, 2, 1, np, gi);
    }
    qn_sym_hpass_epilogue(a, ctl->sc, nrhs, R, t0, t1, red);
}

// ... the generic path's variant: the totals go to the gathered h_pass output the control step reads ([P][2][rpr], rank-major)
__global__ __launch_bounds__(256) void symsh_hpass_epi_generic_kernel(const QnSymHPassArgs a) {
    const QnCtl* __restrict__ ctl = a.ctl;
    if (ctl->phase != a.expect_phase) return;
    const int nrhs = ctl->hp_nrhs;
    if (nrhs == 0 || threadIdx.x >= QN_TB) return; // (0: the pass only applied the pending update)
    const int gi = blockIdx.x * QN_TB + (int)threadIdx.x;
    const size_t np = (size_t)a.T.n_pad, rpr = (size_t)a.T.rpr;
    double* out = a.ghp + ((size_t)gi / rpr) * 2 * rpr + (size_t)gi % rpr;
    out[0] = qn_symsh_rank_sum(a.sh, 2, 0, np, gi);
    if (nrhs == 2) out[rpr] = qn_symsh_rank_sum(a.sh, 2, 1, np, gi);
}

// row-sharded runs, after a run: restore the block-rows' stale halves from the ranks that maintain them.  Step `il` of nbl: every
// rank has contributed its local block-row il (128 whole rows) to `gath` [world][128][n_pad]; workgroup (jl, p) copies the
// transpose of tile (I_p, J) -- I_p = p*nbl + il, J = ioff + jl -- into the local tile (J, I_p) when block-row I_p owns the pair.
__global__ __launch_bounds__(256) void symsh_mirror_kernel(double* __restrict__ H, const double* __restrict__ gath, int n_pad, int nb, int il,
                                                           const QnSymShard sh) {
    __shared__ double t[32][33];
    const int p = blockIdx.y, jl = blockIdx.x;
    const int I = p * sh.nbl + il, J = sh.ioff + jl;
    if (I == J || !qn_symsh_owns(I, J, nb)) return;
    const size_t np = (size_t)n_pad;
    const double* src = gath + (size_t)p * QN_TB * np + (size_t)J * QN_TB; // rows of block-row I_p, columns of block J
    double* dst = H + (size_t)jl * QN_TB * np + (size_t)I * QN_TB;         // local rows of block J, columns of block I_p
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int sb = 0; sb < 16; ++sb) { // 4 x 4 sub-tiles of 32 x 32
        const int bi = sb >> 2, bj = sb & 3;
        __syncthreads();
        for (int r = ty; r < 32; r += 8) t[r][tx] = src[(size_t)(bi * 32 + r) * np + bj * 32 + tx];
        __syncthreads();
        for (int r = ty; r < 32; r += 8) dst[(size_t)(bj * 32 + r) * np + bi * 32 + tx] = t[tx][r];
    }
}

// after a run: lower tile (J, I) <- transpose of the maintained upper tile (I, J), I < J; 32 x 32 sub-tiles through LDS
__global__ __launch_bounds__(256) void sym_mirror_kernel(double* __restrict__ H, int n_pad) {
    __shared__ double t[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x; // 32-blocks
    if (bj <= bi) return;
    if ((bi * 32) / QN_TB == (bj * 32) / QN_TB) return; // inside a diagonal 128-tile: both halves are maintained
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t np = (size_t)n_pad;
    for (int r = ty; r < 32; r += 8) t[r][tx] = H[(size_t)(bi * 32 + r) * np + bj * 32 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) H[(size_t)(bj * 32 + r) * np + bi * 32 + tx] = t[tx][r];
}
