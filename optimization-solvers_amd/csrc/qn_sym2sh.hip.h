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
// Every sum has a fixed order (slots in list order, ranks in rank order): the ranks hold the same bits, and the host-staged
// exchange of the tests gives the bits of the RCCL all-gather.
#pragma once

// Slot sums of block-row R over the slots THIS rank's tiles wrote (host-built ascending list: the rank's own window when R is
// local, and the column parts of the local block-rows whose windows contain R).  4 x 128 threads: each quarter adds a quarter
// of the list in list order, 16 loads in flight, the quarters are combined in order through LDS (qn_s2_slot_sum's scheme; no
// limit on nb -- the first generation compacted the list in 512 LDS words).  Threads 0..127 return row (tid & 127)'s total.
template <int NRHS>
__device__ __forceinline__ double qn_s2sh_list_sum(const double* __restrict__ part, const int nb, const int R, const int rhs, const int* __restrict__ list,
                                                   const int nlist, double (*qbuf)[QN_TB]) {
    const int i = threadIdx.x & (QN_TB - 1), qd = threadIdx.x >> 7;
    const int per = (nlist + 3) / 4;
    const int k_lo = qd * per, k_hi = min(nlist, k_lo + per);
    const double* p = part + (((size_t)R * nb) * NRHS + rhs) * QN_TB + i;
    double acc = 0.0;
    for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (k0 + u < k_hi) ? p[(size_t)list[k0 + u] * NRHS * QN_TB] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = acc + v[u];
    }
    if (qd > 0) qbuf[qd - 1][i] = acc;
    __syncthreads();
    const double tot = (qd == 0) ? ((acc + qbuf[0][i]) + qbuf[1][i]) + qbuf[2][i] : 0.0;
    __syncthreads(); // qbuf is reused by the next call
    return tot;
}

// accepted evaluation, block-row R: this rank's share of q = Q (x + t d) into its slice of xg ([rank][np]); the exchange and
// s2_vec_kernel<true> follow.  Its prologue is where the machine sees the evaluation that gets accepted (QN_PH_REQ_VEC).
__global__ __launch_bounds__(QN_S2_TPB) void s2sh_vsum_kernel(const QnS2Args a) {
    __shared__ QnS2Lds L;
    __shared__ double qbuf[3][QN_TB];
    const int R = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    if (wave == 0) qn_s2_prologue_w0<QN_S2_VSUM, true>(a, L);
    __syncthreads();
    qn_s2_ctl_out(a, L);
    if (!L.mine) return;
    const int lo = a.sl_off[R], nlist = a.sl_off[R + 1] - lo;
    const double qi = qn_s2sh_list_sum<1>(a.partE, a.nb, R, 0, a.sl_idx + lo, nlist, qbuf);
    if (tid < QN_TB) a.xg[(size_t)a.sh_rank * (size_t)a.np + (size_t)R * QN_TB + tid] = qi;
}

// update pass, block-row R: this rank's share of [u, v] = H+ [y, g+] (direction pass: [H g, -]) into its slice of xg ([rank][2][np])
__global__ __launch_bounds__(QN_S2_TPB) void s2sh_hsum_kernel(const QnS2Args a) {
    __shared__ QnS2Lds L;
    __shared__ double qbuf[3][QN_TB];
    const int R = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    if (wave == 0) qn_s2_prologue_w0<QN_S2_HSUM, true>(a, L);
    __syncthreads();
    qn_s2_ctl_out(a, L);
    if (!L.mine) return;
    const int lo = a.sl_off[R], nlist = a.sl_off[R + 1] - lo;
    const double t0 = qn_s2sh_list_sum<2>(a.part, a.nb, R, 0, a.sl_idx + lo, nlist, qbuf);
    const double t1 = qn_s2sh_list_sum<2>(a.part, a.nb, R, 1, a.sl_idx + lo, nlist, qbuf);
    if (tid < QN_TB) {
        double* out = a.xg + (size_t)a.sh_rank * 2 * (size_t)a.np + (size_t)R * QN_TB + tid;
        out[0] = t0;
        out[a.np] = t1;
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
