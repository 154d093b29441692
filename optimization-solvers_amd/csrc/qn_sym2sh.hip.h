// qn_sym2sh.hip.h -- the second-generation symmetric path, ROW-SHARDED (one process per GPU; BASELINE config 3).
//
// What is sharded is the reference's `-&self.approx_inv_hessian * eval.g()` (bfgs.rs:47) and the update of bfgs.rs:115-124: rank p
// stores whole rows of its block-rows of H and Q and streams the CIRCULANT half of them (qn_sym.hip.h: every unordered pair of
// block-rows exactly once across the ranks).  Rounds 2-3 ran this partition on the first-generation kernels: per evaluation or
// pass `tile, sum, exchange, epilogue, control step` -- 12.2 launches per iteration and an n-vector collective for EVERY evaluation,
// rejected trials included.  Here the partition runs in the structure the single-GPU path proved (qn_sym2.hip.h):
//
//   * the tile kernels are the SAME kernels (SHARD instantiations of s2_eval_kernel / s2_hpass_kernel: work lists over the rank's
//     windows, 16-row register windows, parking, one exchange per group of items), the state machine runs in wave 0's prologue of
//     every launch on every rank -- from the same inputs, hence to the same decisions: no control launch, no scalar broadcast;
//   * an EVALUATION is exchanged as scalars: a rank's workgroups leave QN_S2SH_NEC sums each in the rank's slice of evS (8 KB), ONE
//     all-gather (or all-reduce) of those slices follows the launch, and the next launch's prologue adds the ranks up in rank
//     order.  The line search reads f and g'd, not g: trial points that are rejected never move an n-vector between GPUs;
//   * n-vectors cross the links twice per iteration: q = Q x+ of the ACCEPTED point (s2sh_vsum_kernel: this rank's slot sums ->
//     exchange -> s2_vec_kernel<true>: rank-order sum, g+, y, x+, s on every rank) and [u, v] = H+ [y, g+] of the update pass
//     (s2_hpass_kernel<.., true> -> s2sh_hsum_kernel -> exchange -> s2_hreduce_kernel<true>).
//
// One iteration with More-Thuente on the quadratic (two evaluations):
//     eval | x | eval | x | vsum | X | vec | update tiles | hsum | XX | hreduce         (x: 8 KB of scalars, X: n doubles, XX: 2 n)
// = 7 launches and 2 n-vector + E scalar collectives (first generation: 12.2 launches, E + 1 n-vector collectives).
// (Round 6, qn_context_set_trial_vector_exchange: eval | vsumt | xX | eval | vsumt | xX | vec | update tiles | hsum | XX | hreduce -- the trial's partial
// vector with its scalars in one grouped collective, E + 1 collectives per iteration: s2sh_vsumt_kernel below.)
// Every sum has a fixed order (slots in list order, ranks in rank order): the ranks hold the same bits, and the host-staged
// exchange of the tests gives the bits of the RCCL all-gather.
#pragma once

// Slot sums of block-row R over the slots THIS rank's tiles wrote (host-built ascending list: the rank's own window when R is
// local, and the column parts of the local block-rows whose windows contain R; no limit on nb -- the first generation compacted
// the list in 512 LDS words).  Nothing here depends on the control block, so the sums are formed WHILE wave 0 runs the prologue:
// waves 1..7 take a contiguous seventh of the list each, a whole slot (128 rows) per 16-byte load instruction -- lane l holds rows
// 2 l, 2 l + 1 -- a dozen slots in flight, added in list order; the waves' partial sums meet in LDS and are added in wave order
// behind the workgroup barrier the prologue needs anyway.  (First version, measured on one rank of the P = 8, n = 32768
// partition: the slots requested after the barrier, 8 bytes per lane, sixteen at a time through a dependent index load --
// 27 us per launch, and two of them per iteration.)
#define QN_S2SH_SB 12 // slots in flight per wave
template <int NRHS>
__device__ __forceinline__ void qn_s2sh_wave_sums(const double* __restrict__ part, const int nb, const int R, const int* __restrict__ list, const int nlist,
                                                  const int wave, const int lane, double (*red)[NRHS][QN_TB]) { // red[QN_S2_WAVES - 1][NRHS][128]
    constexpr int NW = QN_S2_WAVES - 1;
    const int per = (nlist + NW - 1) / NW;
    const int wv = __builtin_amdgcn_readfirstlane(wave); // (uniform: the list entries below are then scalar loads)
    const int k_lo = (wv - 1) * per, k_hi = min(nlist, k_lo + per);
    const double* p = part + ((size_t)R * nb) * NRHS * QN_TB + 2 * lane;
    v2d acc[NRHS];
#pragma unroll
    for (int h = 0; h < NRHS; ++h) acc[h] = (v2d){0.0, 0.0};
    for (int k0 = k_lo; k0 < k_hi; k0 += QN_S2SH_SB) {
        v2d v[QN_S2SH_SB][NRHS];
#pragma unroll
        for (int u = 0; u < QN_S2SH_SB; ++u) {
            const int slot = (k0 + u < k_hi) ? list[k0 + u] : -1; // (wave-uniform: scalar loads)
#pragma unroll
            for (int h = 0; h < NRHS; ++h) v[u][h] = (slot >= 0) ? ld2(p + ((size_t)slot * NRHS + h) * QN_TB) : (v2d){0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < QN_S2SH_SB; ++u)
#pragma unroll
            for (int h = 0; h < NRHS; ++h) acc[h] = acc[h] + v[u][h];
    }
#pragma unroll
    for (int h = 0; h < NRHS; ++h) { red[wave - 1][h][2 * lane] = acc[h].x; red[wave - 1][h][2 * lane + 1] = acc[h].y; }
}
// ... behind the barrier: thread (h, i) adds the waves' shares of row i, right-hand side h, in wave order
template <int NRHS>
__device__ __forceinline__ double qn_s2sh_wave_total(const double (*red)[NRHS][QN_TB], const int h, const int i) {
    double t = red[0][h][i];
#pragma unroll
    for (int w = 1; w < QN_S2_WAVES - 1; ++w) t = t + red[w][h][i];
    return t;
}

// accepted evaluation, block-row R: this rank's share of q = Q (x + t d) into its slice of xg ([rank][np]); the exchange and
// s2_vec_kernel<true> follow.  Its prologue is where the machine sees the evaluation that gets accepted (QN_PH_REQ_VEC).
__global__ __launch_bounds__(QN_S2_TPB) void s2sh_vsum_kernel(const QnS2Args a) {
    __shared__ QnS2Lds L;
    __shared__ double red[QN_S2_WAVES - 1][1][QN_TB];
    const int R = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) qn_s2_prologue_w0<QN_S2_VSUM, true>(a, L);
    else {
        const int lo = a.sl_off[R];
        qn_s2sh_wave_sums<1>(a.partE, a.nb, R, a.sl_idx + lo, a.sl_off[R + 1] - lo, wave, lane, red);
    }
    __syncthreads();
    qn_s2_ctl_out(a, L);
    if (!L.mine) return;
    if (tid < QN_TB) a.xg[(size_t)a.sh_rank * (size_t)a.np + (size_t)R * QN_TB + tid] = qn_s2sh_wave_total<1>(red, 0, tid);
}

// THE TRIAL'S PARTIAL VECTOR WITH ITS SCALARS (round 6; qn_context_set_trial_vector_exchange, DESIGN 9.1's fallback): behind EVERY evaluation
// launch that evaluated, block-row R of this rank's share of q = Q (x + t d) goes into its slice of xg at once, and ONE grouped collective
// carries the 8 KB of scalars and the n-vector; the acceptance then needs neither s2sh_vsum_kernel nor an exchange of its own
// (s2_vec_kernel<true, true>): one collective per accepted iteration fewer, n doubles per rank more per REJECTED trial.
// Not a link of the control block's chain: no prologue, no machine -- it reads the block the evaluation launch has just handed on
// (a.parity: the NEXT launch's) and acts when that launch evaluated (REQ_EVAL, serviced 2: nobody has consumed it yet); an evaluation
// slot the machine did not use leaves this rank's slice as it is, and the collective behind it re-sends the same values.
__global__ __launch_bounds__(QN_S2_TPB) void s2sh_vsumt_kernel(const QnS2Args a) {
    __shared__ double red[QN_S2_WAVES - 1][1][QN_TB];
    __shared__ int live;
    const int R = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) {
        if (lane == 0) { const QnCtl* c = a.ctl2 + a.parity; live = c->phase == QN_PH_REQ_EVAL && c->serviced == 2; }
    } else {
        const int lo = a.sl_off[R];
        qn_s2sh_wave_sums<1>(a.partE, a.nb, R, a.sl_idx + lo, a.sl_off[R + 1] - lo, wave, lane, red);
    }
    __syncthreads();
    if (!live) return;
    if (tid < QN_TB) a.xg[(size_t)a.sh_rank * (size_t)a.np + (size_t)R * QN_TB + tid] = qn_s2sh_wave_total<1>(red, 0, tid);
}

// update pass, block-row R: this rank's share of [u, v] = H+ [y, g+] (direction pass: [H g, -]) into its slice of xg ([rank][2][np])
__global__ __launch_bounds__(QN_S2_TPB) void s2sh_hsum_kernel(const QnS2Args a) {
    __shared__ QnS2Lds L;
    __shared__ double red[QN_S2_WAVES - 1][2][QN_TB];
    const int R = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) qn_s2_prologue_w0<QN_S2_HSUM, true>(a, L);
    else {
        const int lo = a.sl_off[R];
        qn_s2sh_wave_sums<2>(a.part, a.nb, R, a.sl_idx + lo, a.sl_off[R + 1] - lo, wave, lane, red);
    }
    __syncthreads();
    qn_s2_ctl_out(a, L);
    if (!L.mine) return;
    if (tid < 2 * QN_TB) {
        const int h = tid >> 7, i = tid & (QN_TB - 1);
        a.xg[((size_t)a.sh_rank * 2 + h) * (size_t)a.np + (size_t)R * QN_TB + i] = qn_s2sh_wave_total<2>(red, h, i);
    }
}

// After a row-sharded run of these kernels: inside the DIAGONAL tiles of the local block-rows only the upper triangle of 16 x 16
// sub-blocks is up to date (qn_sym2.hip.h, qn_s2_col) -- restore the rest from it.  symsh_mirror_kernel (qn_sym.hip.h) then
// restores the stale tiles from the ranks that own the pairs.  Grid (4, 4, nbl): 32 x 32 blocks of local diagonal tile z.
__global__ __launch_bounds__(256) void s2sh_diag_mirror_kernel(double* __restrict__ H, const int n_pad, const int ioff) {
    __shared__ double t[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x, il = blockIdx.z;
    if (bj < bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t np = (size_t)n_pad;
    double* D = H + (size_t)il * QN_TB * np + (size_t)(ioff + il) * QN_TB; // tile (ioff + il, ioff + il) in the rank's storage
    for (int r = ty; r < 32; r += 8) t[r][tx] = D[(size_t)(bi * 32 + r) * np + bj * 32 + tx];
    __syncthreads();
    if (bj > bi) {
        for (int r = ty; r < 32; r += 8) D[(size_t)(bj * 32 + r) * np + bi * 32 + tx] = t[tx][r];
    } else {
        for (int r = ty; r < 32; r += 8)
            if (tx < r) D[(size_t)(bi * 32 + r) * np + bi * 32 + tx] = t[tx][r];
    }
}
