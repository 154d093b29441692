// qn_kernels.hip.h -- gfx950 kernels of the quasi-Newton inner loop (included once by qn_hip.hip).
//
// Kernel inventory, all f64, all HBM-bound (BLAS-2 / rank-2: MFMA not applicable):
//   h_pass_kernel      K1+K2 fused: apply the pending symmetric rank-2 update to this rank's rows of H
//                      (read + write once) and dot the UPDATED rows with up to two right-hand sides
//                      (u = H y, v = H g+).  Replaces bfgs.rs:47 and bfgs.rs:115-124 / dfp.rs:115-120.
//   quad_matvec_kernel K5: q = Q_rows (x + t d) with the trial point formed on the fly
//                      (morethuente.rs:182,217,276 / backtracking.rs:32 / bfgs.rs:94 + the oracle's GEMV).
//   ctl_step_kernel    K4+K6: every O(n) vector op and every scalar decision of the driver and the
//                      line searches, as one single-workgroup state machine (qn_ctl.h).
//   trial_point_kernel K3 for oracles that are not fused (host / device closures, log-sum-exp).
//   + plain primitives for the kernel-level FFI (gemv, rank-2, axpy, dot).
//
// Tiling of the two streaming kernels: a 256-thread workgroup owns R consecutive rows; per 512-column chunk
// every thread owns two adjacent columns (one 16-byte load per row: 1 KiB per wave instruction, fully
// coalesced), keeps the chunk's vector entries in registers and walks the R rows with all R loads in flight.
// Row sums live in registers, are folded across the wave with a halving xor-shuffle butterfly (R*NRHS values
// cost R*NRHS shuffles, not 6x that) and across the 4 waves through LDS.  No float atomics: every sum has a
// fixed order, so all ranks of a sharded run take bit-identical decisions.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "qn_ctl.h"

typedef double v2d __attribute__((ext_vector_type(2)));

#define QN_TPB 256        // threads per workgroup of the streaming kernels
#define QN_CHUNK 512      // columns per chunk (2 per thread)
#define QN_CTL_TPB 1024   // threads of the control workgroup

struct QnTile {
    int n;       // logical dimension
    int n_pad;   // padded dimension = leading dimension = world * rpr
    int rpr;     // rows per rank (multiple of 16)
    int row_off; // global index of this rank's first row
    int cs;      // column splits (gridDim.y)
    int rank;
};

struct QnVecs {
    double *x, *g, *d, *xt, *gt, *s, *y, *sp, *up;
    double* q;        // quadratic objective: Q (x + t d), gathered layout [world][qcs][rpr]
    const double* b;  // quadratic objective: b (n_pad)
    double* hp;       // h_pass output, gathered: rank block p at hp + p*hcs*2*rpr, inside it [hcs][nrhs][rpr]
    double* f_dev;    // generic oracles: f at xt
    QnTraceRec* trace;
    double* xtrace;
    double* H; // this rank's rows (used directly only by the n <= 5 reference-order path)
    int n, n_pad, rpr, world, hcs, qcs;
};

__device__ __forceinline__ v2d ld2(const double* p) { return *reinterpret_cast<const v2d*>(p); }
__device__ __forceinline__ void st2(double* p, v2d v) { *reinterpret_cast<v2d*>(p) = v; }

// ---- halving butterfly: V values per lane -> lane l ends with the wave total of value (l >> (6 - log2 V)) in v[0]
template <int CNT, int OFF>
struct QnWaveFold {
    template <int V>
    static __device__ __forceinline__ void run(double (&v)[V], int lane) {
        if constexpr (OFF >= 1) {
            if constexpr (CNT > 1) {
                constexpr int HALF = CNT / 2;
                const bool up = (lane & OFF) != 0;
#pragma unroll
                for (int i = 0; i < HALF; ++i) {
                    const double keep = up ? v[i + HALF] : v[i];
                    const double send = up ? v[i] : v[i + HALF];
                    const double recv = __shfl_xor(send, OFF, 64);
                    v[i] = keep + recv;
                }
                QnWaveFold<HALF, OFF / 2>::run(v, lane);
            } else {
                v[0] = v[0] + __shfl_xor(v[0], OFF, 64);
                QnWaveFold<1, OFF / 2>::run(v, lane);
            }
        }
    }
};

template <int V>
__device__ __forceinline__ constexpr int qn_log2() {
    return V <= 1 ? 0 : 1 + qn_log2<V / 2>();
}

// Fold V per-thread partial sums over the whole 256-thread workgroup; thread `idx` (< V) returns value idx.
template <int V>
__device__ __forceinline__ double qn_block_fold(double (&v)[V], double* red /* LDS, 4*V doubles */) {
    static_assert(V >= 1 && V <= 64 && (V & (V - 1)) == 0, "V must be a power of two <= 64");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    QnWaveFold<V, 32>::run(v, lane);
    constexpr int SH = 6 - qn_log2<V>();
    if ((lane & ((1 << SH) - 1)) == 0) red[wave * V + (lane >> SH)] = v[0];
    __syncthreads();
    double tot = 0.0;
    if (tid < V) tot = ((red[tid] + red[V + tid]) + red[2 * V + tid]) + red[3 * V + tid];
    return tot;
}

// ------------------------------------------------------------------------------------------------
// h_pass: H_rows <- H_rows + c_su (sp up' + up sp') + c_ss sp sp' + c_uu up up'   (if PENDING)
//         out[rhs][i] = sum_j H_new[i][j] * rhs[j]
// The update is evaluated with commutative inner sums so H stays bitwise symmetric; the operation order
// ((H + c_su*t1) + c_ss*(s_i s_j)) + c_uu*(u_i u_j) is also the one the CPU checker's rank-2 mode uses (tests only).
// ------------------------------------------------------------------------------------------------
template <int R, int NRHS, bool PENDING>
__device__ __forceinline__ void h_pass_body(double* __restrict__ H, const QnTile T, const double* __restrict__ sp,
                                            const double* __restrict__ up, const double* __restrict__ r0,
                                            const double* __restrict__ r1, const double c_ss, const double c_su,
                                            const double c_uu, double* __restrict__ out, double* red) {
    const int tid = threadIdx.x;
    const int rb = blockIdx.x * R;
    const int cs = blockIdx.y;
    const int nchunks = (T.n_pad + QN_CHUNK - 1) / QN_CHUNK;
    const int cps = (nchunks + T.cs - 1) / T.cs;
    const int c_begin = cs * cps;
    const int c_end = min(nchunks, c_begin + cps);
    const bool use_su = c_su != 0.0, use_uu = c_uu != 0.0;

    double si[R], ui[R];
    bool rowok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int gi = T.row_off + rb + r;
        rowok[r] = gi < T.n;
        si[r] = PENDING ? sp[gi] : 0.0;
        ui[r] = PENDING ? up[gi] : 0.0;
    }
    constexpr int NV = (NRHS > 0 ? NRHS : 1) * R;
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;

    for (int c = c_begin; c < c_end; ++c) {
        const int j = c * QN_CHUNK + 2 * tid;
        if (j < T.n_pad) {
            v2d sj = {0.0, 0.0}, uj = {0.0, 0.0}, y0 = {0.0, 0.0}, y1 = {0.0, 0.0};
            if (PENDING) {
                sj = ld2(sp + j);
                uj = ld2(up + j);
            }
            if (NRHS >= 1) y0 = ld2(r0 + j);
            if (NRHS >= 2) y1 = ld2(r1 + j);
            const bool c0ok = j < T.n, c1ok = (j + 1) < T.n;
            double* hbase = H + (size_t)rb * (size_t)T.n_pad + j;
            v2d h[R];
#pragma unroll
            for (int r = 0; r < R; ++r) h[r] = ld2(hbase + (size_t)r * (size_t)T.n_pad);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                v2d hn = h[r];
                if (PENDING) {
                    if (use_su) {
                        hn.x = hn.x + c_su * (si[r] * uj.x + ui[r] * sj.x);
                        hn.y = hn.y + c_su * (si[r] * uj.y + ui[r] * sj.y);
                    }
                    hn.x = hn.x + c_ss * (si[r] * sj.x);
                    hn.y = hn.y + c_ss * (si[r] * sj.y);
                    if (use_uu) {
                        hn.x = hn.x + c_uu * (ui[r] * uj.x);
                        hn.y = hn.y + c_uu * (ui[r] * uj.y);
                    }
                    hn.x = (rowok[r] && c0ok) ? hn.x : 0.0; // padding stays exactly zero
                    hn.y = (rowok[r] && c1ok) ? hn.y : 0.0;
                    st2(hbase + (size_t)r * (size_t)T.n_pad, hn);
                }
                if (NRHS >= 1) {
                    acc[r] = __builtin_fma(hn.x, y0.x, acc[r]);
                    acc[r] = __builtin_fma(hn.y, y0.y, acc[r]);
                }
                if (NRHS >= 2) {
                    acc[R + r] = __builtin_fma(hn.x, y1.x, acc[R + r]);
                    acc[R + r] = __builtin_fma(hn.y, y1.y, acc[R + r]);
                }
            }
        }
    }
    if (NRHS >= 1) {
        const double tot = qn_block_fold<NV>(acc, red);
        if (tid < NV) {
            const int rhs = tid / R, r = tid % R;
            out[((size_t)cs * NRHS + rhs) * (size_t)T.rpr + rb + r] = tot;
        }
    }
}

struct QnHPassArgs {
    double* H;
    QnTile T;
    const double *sp, *up;
    const double *vy, *vg; // predicated mode: rhs0 = g for a direction pass (bfgs.rs:47), y for an update pass; rhs1 = g
    const double *r0, *r1; // unconditional mode: explicit right-hand sides
    double* hp;            // gathered output buffer (all ranks); this rank's block is at hp + rank*cs*2*rpr
    const QnCtl* ctl;
    int expect_phase; // predicate (pipelined mode); < 0: unconditional, use force_* below
    int force_nrhs, force_pending;
    double c_ss, c_su, c_uu; // used when expect_phase < 0
};

template <int R>
__global__ __launch_bounds__(QN_TPB) void h_pass_kernel(const QnHPassArgs a) {
    __shared__ double red[4 * 2 * R];
    int nrhs, pending;
    double c_ss, c_su, c_uu;
    const double *r0, *r1;
    if (a.expect_phase >= 0) {
        if (a.ctl->phase != a.expect_phase) return;
        nrhs = a.ctl->hp_nrhs;
        pending = a.ctl->pending;
        c_ss = a.ctl->c_ss; c_su = a.ctl->c_su; c_uu = a.ctl->c_uu;
        r0 = (a.ctl->after_state == QN_ST_AFTER_DIR) ? a.vg : a.vy;
        r1 = a.vg;
    } else {
        nrhs = a.force_nrhs;
        pending = a.force_pending;
        c_ss = a.c_ss; c_su = a.c_su; c_uu = a.c_uu;
        r0 = a.r0; r1 = a.r1;
    }
    double* out = a.hp + (size_t)a.T.rank * a.T.cs * 2 * a.T.rpr;
    if (pending) {
        if (nrhs == 2) h_pass_body<R, 2, true>(a.H, a.T, a.sp, a.up, r0, r1, c_ss, c_su, c_uu, out, red);
        else if (nrhs == 1) h_pass_body<R, 1, true>(a.H, a.T, a.sp, a.up, r0, r1, c_ss, c_su, c_uu, out, red);
        else h_pass_body<R, 0, true>(a.H, a.T, a.sp, a.up, r0, r1, c_ss, c_su, c_uu, out, red);
    } else {
        if (nrhs == 2) h_pass_body<R, 2, false>(a.H, a.T, a.sp, a.up, r0, r1, c_ss, c_su, c_uu, out, red);
        else if (nrhs == 1) h_pass_body<R, 1, false>(a.H, a.T, a.sp, a.up, r0, r1, c_ss, c_su, c_uu, out, red);
    }
}

// ------------------------------------------------------------------------------------------------
// quad_matvec: q_rows = Q_rows * xt,  xt = x (QN_REQ_X) or x + t*d (QN_REQ_T: fl(x_j + fl(t*d_j)))
// Row-block 0 of every column split also stores xt.
// ------------------------------------------------------------------------------------------------
struct QnQuadArgs {
    const double* Q;
    QnTile T;
    const double *x, *d;
    double* xt;
    double* out; // this rank's block of the gathered q buffer [qcs][rpr]
    const QnCtl* ctl;
    int expect_phase; // < 0: unconditional with force_kind / force_t
    int force_kind;
    double force_t;
};

template <int R>
__global__ __launch_bounds__(QN_TPB) void quad_matvec_kernel(const QnQuadArgs a) {
    __shared__ double red[4 * R];
    int kind;
    double t;
    if (a.expect_phase >= 0) {
        if (a.ctl->phase != a.expect_phase) return;
        kind = a.ctl->req_kind;
        t = a.ctl->req_t;
    } else {
        kind = a.force_kind;
        t = a.force_t;
    }
    const QnTile T = a.T;
    const int tid = threadIdx.x;
    const int rb = blockIdx.x * R;
    const int cs = blockIdx.y;
    const int nchunks = (T.n_pad + QN_CHUNK - 1) / QN_CHUNK;
    const int cps = (nchunks + T.cs - 1) / T.cs;
    const int c_begin = cs * cps;
    const int c_end = min(nchunks, c_begin + cps);
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    for (int c = c_begin; c < c_end; ++c) {
        const int j = c * QN_CHUNK + 2 * tid;
        if (j < T.n_pad) {
            v2d xj = ld2(a.x + j);
            if (kind == QN_REQ_T) {
                const v2d dj = ld2(a.d + j);
                const double td0 = t * dj.x, td1 = t * dj.y; // `step * direction` rounds first
                xj.x = xj.x + td0;
                xj.y = xj.y + td1;
            }
            if (blockIdx.x == 0) st2(a.xt + j, xj);
            const double* qbase = a.Q + (size_t)rb * (size_t)T.n_pad + j;
            v2d h[R];
#pragma unroll
            for (int r = 0; r < R; ++r) h[r] = ld2(qbase + (size_t)r * (size_t)T.n_pad);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r] = __builtin_fma(h[r].x, xj.x, acc[r]);
                acc[r] = __builtin_fma(h[r].y, xj.y, acc[r]);
            }
        }
    }
    const double tot = qn_block_fold<R>(acc, red);
    if (tid < R) a.out[(size_t)cs * (size_t)T.rpr + rb + tid] = tot;
}

// xt = x or x + t*d for oracles that are not fused with the trial point
__global__ void trial_point_kernel(const double* __restrict__ x, const double* __restrict__ d, double* __restrict__ xt,
                                   int n_pad, const QnCtl* ctl, int expect_phase) {
    if (ctl->phase != expect_phase) return;
    const int kind = ctl->req_kind;
    const double t = ctl->req_t;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) {
        double v = x[i];
        if (kind == QN_REQ_T) {
            const double td = t * d[i];
            v = v + td;
        }
        xt[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// synthetic SPD generator; integer hash + two exact f64 ops => bit-identical to
// the host-side generator used by the tests
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t qn_splitmix64_mix(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

__global__ void synth_fill_kernel(double* __restrict__ Q, const QnTile T, uint64_t seed, const double* __restrict__ diag,
                                  double inv_n) {
    const size_t total = (size_t)T.rpr * (size_t)T.n_pad;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const uint64_t il = e / (size_t)T.n_pad, j = e % (size_t)T.n_pad;
        const uint64_t i = (uint64_t)T.row_off + il;
        double v = 0.0;
        if (i < (uint64_t)T.n && j < (uint64_t)T.n) {
            if (i == j) {
                v = diag[i];
            } else {
                const uint64_t lo = i < j ? i : j, hi = i < j ? j : i;
                const uint64_t z = qn_splitmix64_mix(seed + ((lo << 32) | hi) * 0x9E3779B97F4A7C15ull);
                const double r = (double)(z >> 11) * 0x1.0p-53;
                v = (2.0 * r - 1.0) * inv_n;
            }
        }
        Q[e] = v;
    }
}

__global__ void identity_fill_kernel(double* __restrict__ H, const QnTile T) {
    const size_t total = (size_t)T.rpr * (size_t)T.n_pad;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t il = e / (size_t)T.n_pad, j = e % (size_t)T.n_pad;
        const size_t i = (size_t)T.row_off + il;
        H[e] = (i == j && i < (size_t)T.n) ? 1.0 : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// control workgroup helpers
// ------------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void ctl_block_sum(double (&v)[K], double* lds /* 16*K */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] = v[k] + __shfl_xor(v[k], off, 64);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double t = 0.0;
        for (int w = 0; w < QN_CTL_TPB / 64; ++w) t = t + lds[w * K + k];
        v[k] = t;
    }
}

__device__ __forceinline__ double ctl_block_fmax(double v, double* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double t = -INFINITY;
    for (int w = 0; w < QN_CTL_TPB / 64; ++w) t = fmax(t, lds[w]);
    return t;
}

__device__ __forceinline__ double hp_val(const QnVecs& V, int nrhs, int rhs, int i) {
    const int p = i / V.rpr, il = i - p * V.rpr;
    const double* base = V.hp + (size_t)p * V.hcs * 2 * V.rpr; // rank blocks are sized for two right-hand sides
    double t = base[(size_t)rhs * V.rpr + il];
    for (int cs = 1; cs < V.hcs; ++cs) t = t + base[((size_t)cs * nrhs + rhs) * V.rpr + il];
    return t;
}

__device__ __forceinline__ double q_val(const QnVecs& V, int i) {
    const int p = i / V.rpr, il = i - p * V.rpr;
    const double* base = V.q + (size_t)p * V.qcs * V.rpr;
    double t = base[il];
    for (int cs = 1; cs < V.qcs; ++cs) t = t + base[(size_t)cs * V.rpr + il];
    return t;
}

// ---- scalar pieces of the More-Thuente search (morethuente.rs:64-132), thread 0 only ----
__device__ __forceinline__ int mt_update_interval(double f_tl, double f_t, double g_t, double* tl, double t, double* tu) {
    if (f_t > f_tl) { *tu = t; return 0; }                         // U1
    else if (g_t * (*tl - t) > 0.) { *tl = t; return 0; }          // U2
    else if (g_t * (*tl - t) < 0.) { *tu = *tl; *tl = t; return 0; } // U3
    return 1;                                                      // interval converged to a point
}
__device__ __forceinline__ double mt_cubic(double ta, double tb, double f_ta, double f_tb, double g_ta, double g_tb) {
    const double s = 3. * (f_tb - f_ta) / (tb - ta);
    const double z = s - g_ta - g_tb;
    const double w = sqrt(z * z - g_ta * g_tb);
    return ta + ((tb - ta) * ((w - g_ta - z) / (g_tb - g_ta + 2. * w)));
}
__device__ __forceinline__ double mt_quad1(double ta, double tb, double f_ta, double f_tb, double g_ta) {
    const double lin_int = (f_ta - f_tb) / (ta - tb);
    return ta - 0.5 * ((ta - tb) * g_ta / (g_ta - lin_int));
}
__device__ __forceinline__ double mt_quad2(double ta, double tb, double g_ta, double g_tb) {
    return ta - g_ta * ((ta - tb) / (g_ta - g_tb));
}

__device__ __forceinline__ void tr_push_case(QnCtl& c, int digit) {
    if (c.tr_ndigits < 10) {
        int32_t mul = 1;
        for (int i = 0; i < c.tr_ndigits; ++i) mul *= 8;
        c.tr_ls_cases += mul * digit;
    }
    c.tr_ndigits++;
}

// One oracle call of the reference's sequence at x + t d.  With memoisation a call whose point was already
// evaluated is answered from the memo (the values are identical; see include/qn_hip.h qn_oracle.memoize).
__device__ __forceinline__ void req_eval_t(QnCtl& c, double t, int after_state, int need_vectors) {
    c.n_oracle_calls++;
    c.tr_n_evals++;
    if (c.memoize) {
        if (!need_vectors && t == 0.0 && c.d_finite) { // x + 0*d == x: phi(0) = (f_k, g_k.d)
            c.f_e = c.f_k; c.gd_e = c.gd0; c.state = after_state;
            return;
        }
        if (c.last_valid && c.last_t == t) {
            c.f_e = c.f_last; c.gd_e = c.gd_last; c.state = after_state;
            return;
        }
    }
    c.req_kind = QN_REQ_T;
    c.req_t = t;
    c.req_need_vectors = need_vectors;
    c.after_state = after_state;
    c.phase = QN_PH_REQ_EVAL;
}

// ------------------------------------------------------------------------------------------------
// Reference-order arithmetic for n <= 5 (single thread).  For these sizes nalgebra does not hand products to
// matrixmultiply, so the reference's operation order is fully determined and the HIP
// path reproduces the reference bit for bit -- this is what keeps `assert_eq!(eval.f(), &0.0)` of
// examples/quadratic.rs:43 true.  Larger n use the O(n^2) rank-2 form (tolerance-level parity).
// ------------------------------------------------------------------------------------------------
#define QN_SMALL_N 5

__device__ __forceinline__ double ref_dot(const double* a, const double* b, int n) { // [nalgebra] 8-accumulator dot
    double res = 0.0, a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    int i = 0;
    while (n - i >= 8) {
        a0 += a[i] * b[i]; a1 += a[i + 1] * b[i + 1]; a2 += a[i + 2] * b[i + 2]; a3 += a[i + 3] * b[i + 3];
        a4 += a[i + 4] * b[i + 4]; a5 += a[i + 5] * b[i + 5]; a6 += a[i + 6] * b[i + 6]; a7 += a[i + 7] * b[i + 7];
        i += 8;
    }
    res += a0 + a4; res += a1 + a5; res += a2 + a6; res += a3 + a7;
    for (; i < n; ++i) res += a[i] * b[i];
    return res;
}

// d = -(H g), column sweep (bfgs.rs:47); H is row-major with leading dimension ld
__device__ __forceinline__ void small_direction(const double* H, int ld, int n, const double* g, double* d) {
    double y[QN_SMALL_N];
    for (int i = 0; i < n; ++i) y[i] = H[i * ld] * g[0];
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < n; ++i) y[i] = H[i * ld + j] * g[j] + y[i];
    for (int i = 0; i < n; ++i) d[i] = -y[i];
}

// C = A*B through per-column gemv, all QN_SMALL_N-strided local arrays indexed [i + j*QN_SMALL_N]
__device__ __forceinline__ void small_matmul(const double* a, const double* b, double* c, int n) {
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) c[i + j * QN_SMALL_N] = a[i] * b[j * QN_SMALL_N];
        for (int k = 1; k < n; ++k)
            for (int i = 0; i < n; ++i) c[i + j * QN_SMALL_N] = a[i + k * QN_SMALL_N] * b[k + j * QN_SMALL_N] + c[i + j * QN_SMALL_N];
    }
}

// bfgs.rs:115-124 / dfp.rs:115-120 exactly as written
__device__ __forceinline__ void small_update(double* H, int ld, int n, const double* s, const double* y, int method) {
    double h[QN_SMALL_N * QN_SMALL_N], m0[QN_SMALL_N * QN_SMALL_N], m1[QN_SMALL_N * QN_SMALL_N], m2[QN_SMALL_N * QN_SMALL_N],
        m3[QN_SMALL_N * QN_SMALL_N];
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) h[i + j * QN_SMALL_N] = H[i * ld + j];
    if (method == 0) {
        const double ys = ref_dot(y, s, n);
        const double rho = 1.0 / ys;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                const double w_a = s[i] * y[j], w_b = s[j] * y[i];
                const double id = (i == j) ? 1.0 : 0.0;
                m0[i + j * QN_SMALL_N] = id - (w_a * rho);
                m1[i + j * QN_SMALL_N] = id - (w_b * rho);
            }
        small_matmul(m0, h, m2, n);
        small_matmul(m2, m1, m3, n);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) H[i * ld + j] = m3[i + j * QN_SMALL_N] + (s[i] * s[j]) * rho;
    } else {
        const double sy = ref_dot(s, y, n);
        double u[QN_SMALL_N];
        for (int i = 0; i < n; ++i) u[i] = h[i] * y[0];
        for (int j = 1; j < n; ++j)
            for (int i = 0; i < n; ++i) u[i] = h[i + j * QN_SMALL_N] * y[j] + u[i];
        const double yhy = ref_dot(y, u, n);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) m0[i + j * QN_SMALL_N] = y[i] * y[j];
        small_matmul(h, m0, m1, n);
        small_matmul(m1, h, m2, n);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                const double delta = (s[i] * s[j]) / sy - m2[i + j * QN_SMALL_N] / yhy;
                H[i * ld + j] = h[i + j * QN_SMALL_N] + delta;
            }
    }
}

#define QN_PH_RUNNING 5
#define QN_ORACLE_GENERIC 0
#define QN_ORACLE_QUAD 1

// ------------------------------------------------------------------------------------------------
// ctl_step: the solver state machine.  One workgroup of 1024 threads; vector work is strided over the
// workgroup (thread t always touches elements t, t+1024, ... so it only re-reads its own writes), scalar
// decisions are taken by thread 0 on the LDS copy of the control block.
// ------------------------------------------------------------------------------------------------
template <int ORACLE>
__global__ __launch_bounds__(QN_CTL_TPB) void ctl_step_kernel(QnCtl* __restrict__ gctl, const QnVecs V, const int expect_phase) {
    __shared__ QnCtl c;
    __shared__ double lds[16 * 4];
    if (gctl->phase != expect_phase) return;
    const int tid = threadIdx.x;
    {
        const uint64_t* src = reinterpret_cast<const uint64_t*>(gctl);
        uint64_t* dst = reinterpret_cast<uint64_t*>(&c);
        for (int i = tid; i < (int)(sizeof(QnCtl) / 8); i += QN_CTL_TPB) dst[i] = src[i];
    }
    __syncthreads();
    const int n = V.n, n_pad = V.n_pad;

    // ---- consume the serviced request ----
    if (expect_phase == QN_PH_REQ_EVAL) {
        double f_e;
        if (ORACLE == QN_ORACLE_QUAD) { // f = 1/2 xt'(Q xt) - b'xt ; g = Q xt - b
            double p[2] = {0.0, 0.0};
            for (int i = tid; i < n_pad; i += QN_CTL_TPB) {
                const double qi = q_val(V, i), xi = V.xt[i], bi = V.b[i];
                p[0] = __builtin_fma(xi, qi, p[0]);
                p[1] = __builtin_fma(bi, xi, p[1]);
                V.gt[i] = qi - bi;
            }
            ctl_block_sum<2>(p, lds);
            f_e = 0.5 * p[0] - p[1];
        } else {
            f_e = *V.f_dev;
        }
        double gd[1] = {0.0};
        const int kind = c.req_kind;
        if (kind == QN_REQ_T) {
            for (int i = tid; i < n_pad; i += QN_CTL_TPB) gd[0] = __builtin_fma(V.gt[i], V.d[i], gd[0]);
            ctl_block_sum<1>(gd, lds);
        }
        __syncthreads();
        if (tid == 0) {
            if (c.small_n && kind == QN_REQ_T) gd[0] = ref_dot(V.gt, V.d, n);
            c.f_e = f_e;
            c.gd_e = gd[0];
            c.n_oracle_evals++;
            if (kind == QN_REQ_T) { c.last_valid = 1; c.last_t = c.req_t; c.f_last = f_e; c.gd_last = gd[0]; }
            else c.last_valid = 0;
        }
    }
    if (tid == 0) {
        if (expect_phase == QN_PH_IDLE) c.state = QN_ST_BEGIN;
        else if (expect_phase == QN_PH_REQ_EVAL || expect_phase == QN_PH_REQ_HPASS) c.state = c.after_state;
        c.phase = QN_PH_RUNNING;
    }

    for (int guard = 0; guard < (1 << 24); ++guard) {
        __syncthreads();
        if (c.phase != QN_PH_RUNNING) break;
        const int st = c.state;
        switch (st) {
        case QN_ST_BEGIN: { // ls_solver.rs:74-76: only k is reset
            if (tid == 0) {
                c.k = 0;
                c.have_cur_eval = 0; c.have_dir = 0; c.last_valid = 0;
                c.n_oracle_calls = 0; c.n_oracle_evals = 0; c.n_hpasses = 0; c.n_hpass_rw = 0; c.n_iterations = 0;
                c.status = -1;
                c.state = QN_ST_LOOP_TOP;
            }
        } break;

        case QN_ST_LOOP_TOP: { // ls_solver.rs:78-79
            if (tid == 0) {
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
        } break;

        case QN_ST_AFTER_EVALX: {
            for (int i = tid; i < n_pad; i += QN_CTL_TPB) V.g[i] = V.gt[i];
            if (tid == 0) {
                c.f_k = c.f_e;
                c.have_cur_eval = c.memoize;
                c.have_dir = 0;
                c.state = QN_ST_CHECK;
            }
        } break;

        case QN_ST_CHECK: { // ls_solver.rs:37-40 (OutOfDomain), has_converged (bfgs.rs:64-76 / gradient_descent.rs:46-53)
            const bool gd_method = c.method == 2;
            double gnorm, p[2] = {0.0, 0.0};
            if (gd_method) {
                double m = -INFINITY; // fold(NEG_INFINITY, |acc, x| x.abs().max(acc)): NaN entries are ignored
                for (int i = tid; i < n; i += QN_CTL_TPB) {
                    const double gi = V.g[i];
                    m = fmax(fabs(gi), m);
                    const double di = -gi; // gradient_descent.rs:29
                    V.d[i] = di;
                    p[0] = __builtin_fma(gi, di, p[0]);
                    p[1] += isfinite(di) ? 0.0 : 1.0;
                }
                gnorm = ctl_block_fmax(m, lds);
                ctl_block_sum<2>(p, lds);
            } else {
                for (int i = tid; i < n_pad; i += QN_CTL_TPB) { const double gi = V.g[i]; p[0] = __builtin_fma(gi, gi, p[0]); }
                ctl_block_sum<2>(p, lds);
                gnorm = sqrt(p[0]);
            }
            if (c.small_n) { // reference order, thread 0 (uniform branch)
                __syncthreads();
                if (tid == 0) {
                    if (!gd_method) {
                        gnorm = sqrt(ref_dot(V.g, V.g, n));
                        small_direction(V.H, n_pad, n, V.g, V.d);
                    }
                    p[0] = ref_dot(V.g, V.d, n);
                    p[1] = 0.0;
                    for (int i = 0; i < n; ++i) p[1] += isfinite(V.d[i]) ? 0.0 : 1.0;
                }
            }
            if (tid == 0) {
                c.gnorm = gnorm; c.tr_f = c.f_k; c.tr_gnorm = gnorm;
                const double f = c.f_k;
                if (isnan(f) || isinf(f)) {
                    c.status = 2; c.phase = QN_PH_DONE; // OutOfDomain
                } else {
                    bool conv;
                    if (gd_method) conv = gnorm < c.tol;
                    else conv = (c.has_s_norm && c.s_norm < c.tol) || (c.has_y_norm && c.y_norm < c.tol) || (gnorm < c.tol);
                    if (conv) {
                        c.status = 0; c.phase = QN_PH_DONE;
                    } else if (gd_method) {
                        c.gd0 = p[0]; c.d_finite = p[1] == 0.0; c.last_valid = 0;
                        c.state = QN_ST_LS_BEGIN;
                    } else if (c.small_n) {
                        c.gd0 = p[0]; c.d_finite = p[1] == 0.0; c.last_valid = 0;
                        c.state = QN_ST_LS_BEGIN;
                    } else if (c.have_dir) {
                        c.state = QN_ST_LS_BEGIN;
                    } else { // bfgs.rs:47 d = -(H g): one pass over H (applies a pending update on the way)
                        c.hp_nrhs = 1; c.hp_lazy = 0;
                        c.after_state = QN_ST_AFTER_DIR;
                        c.phase = QN_PH_REQ_HPASS;
                    }
                }
            }
        } break;

        case QN_ST_AFTER_DIR: {
            double p[2] = {0.0, 0.0};
            for (int i = tid; i < n_pad; i += QN_CTL_TPB) {
                const double di = -hp_val(V, 1, 0, i);
                V.d[i] = di;
                p[0] = __builtin_fma(V.g[i], di, p[0]);
                p[1] += isfinite(di) ? 0.0 : 1.0;
            }
            ctl_block_sum<2>(p, lds);
            if (tid == 0) {
                c.n_hpasses++;
                if (c.pending) c.n_hpass_rw++;
                c.pending = 0;
                c.gd0 = p[0]; c.d_finite = p[1] == 0.0; c.last_valid = 0;
                c.state = QN_ST_LS_BEGIN;
            }
        } break;

        case QN_ST_LS_BEGIN: {
            if (tid == 0) {
                c.ls_i = 0;
                if (c.ls_kind == 0) { // morethuente.rs:173-178
                    c.use_mod = 0; c.conv = 0;
                    c.t = fmin(fmax(1.0, c.mt_tmin), c.mt_tmax);
                    c.tl = c.mt_tmin; c.tu = c.mt_tmax;
                    c.state = QN_ST_MT_LOOP;
                } else { // backtracking.rs:28-29
                    c.t = 1.0;
                    c.state = QN_ST_BT_LOOP;
                }
            }
        } break;

        case QN_ST_MT_LOOP: { // morethuente.rs:181-182
            if (tid == 0) {
                if (!(c.ls_i < c.max_iter_ls)) { c.ls_result = c.t; c.state = QN_ST_AFTER_LS; } // :295-296
                else { c.tr_ls_iters++; req_eval_t(c, c.t, QN_ST_MT_AFTER_T, 0); }
            }
        } break;

        case QN_ST_MT_AFTER_T: { // morethuente.rs:184-217
            if (tid == 0) {
                const double f_et = c.f_e, gd_t = c.gd_e, t = c.t;
                const bool wolfe = (f_et - c.f_k <= c.mt_c1 * t * c.gd0) && (fabs(gd_t) <= c.mt_c2 * fabs(c.gd0));
                if (wolfe || c.conv || t == c.tl || t == c.tu) {
                    tr_push_case(c, 0);
                    c.ls_result = t; c.state = QN_ST_AFTER_LS;
                } else {
                    c.phi_t_f = f_et; c.phi_t_g = gd_t;
                    c.psi_t_f = f_et - c.f_k - c.mt_c1 * t * c.gd0; // psi, :140-149
                    c.psi_t_g = gd_t - c.mt_c1 * c.gd0;
                    if (!c.use_mod && c.psi_t_f <= 0. && c.phi_t_g > 0.) c.use_mod = 1; // :212-215
                    req_eval_t(c, c.tl, QN_ST_MT_AFTER_TL, 0); // :217
                }
            }
        } break;

        case QN_ST_MT_AFTER_TL: { // morethuente.rs:218-287
            if (tid == 0) {
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
                    req_eval_t(c, c.tu, QN_ST_MT_AFTER_TU, 0);
                }
            }
        } break;

        case QN_ST_MT_AFTER_TU: {
            if (tid == 0) {
                double f_tu, g_tu;
                if (c.use_mod) { f_tu = c.f_e; g_tu = c.gd_e; }
                else { f_tu = c.f_e - c.f_k - c.mt_c1 * c.tu * c.gd0; g_tu = c.gd_e - c.mt_c1 * c.gd0; }
                tr_push_case(c, 4);
                c.t = mt_cubic(c.tu, c.t, c.sel_f_t, f_tu, c.sel_g_t, g_tu); // :286, argument order as written
                c.state = QN_ST_MT_FINISH;
            }
        } break;

        case QN_ST_MT_FINISH: { // morethuente.rs:290-293: the NEW t with the OLD trial's f_t, g_t
            if (tid == 0) {
                c.t = fmin(fmax(c.t, c.mt_tmin), c.mt_tmax);
                double tl = c.tl, tu = c.tu;
                c.conv = mt_update_interval(c.sel_f_tl, c.sel_f_t, c.sel_g_t, &tl, c.t, &tu);
                c.tl = tl; c.tu = tu;
                c.ls_i++;
                c.state = QN_ST_MT_LOOP;
            }
        } break;

        case QN_ST_BT_LOOP: { // backtracking.rs:31-34
            if (tid == 0) {
                if (!(c.max_iter_ls > c.ls_i)) { c.ls_result = c.t; c.state = QN_ST_AFTER_LS; } // :54
                else { c.tr_ls_iters++; req_eval_t(c, c.t, QN_ST_BT_AFTER, 0); }
            }
        } break;

        case QN_ST_BT_AFTER: { // backtracking.rs:37-51
            if (tid == 0) {
                const double f1 = c.f_e;
                if (isnan(f1) || isinf(f1)) { c.t *= c.bt_beta; c.state = QN_ST_BT_LOOP; } // shrink, iteration not counted
                else if (f1 - c.f_k <= c.bt_c1 * c.t * c.gd0) { c.ls_result = c.t; c.state = QN_ST_AFTER_LS; }
                else { c.t *= c.bt_beta; c.ls_i++; c.state = QN_ST_BT_LOOP; }
            }
        } break;

        case QN_ST_AFTER_LS: {
            if (c.method == 2) { // default hook ls_solver.rs:44-64 / gradient_descent.rs:55-82: x += step*d, no re-evaluation
                const double step = c.ls_result;
                const bool hit = c.last_valid && c.last_t == step;
                for (int i = tid; i < n_pad; i += QN_CTL_TPB) {
                    if (hit) { V.x[i] = V.xt[i]; if (c.memoize) V.g[i] = V.gt[i]; }
                    else { const double td = step * V.d[i]; V.x[i] = V.x[i] + td; }
                }
                __syncthreads();
                if (tid == 0) {
                    if (hit && c.memoize) { c.f_k = c.f_last; c.have_cur_eval = 1; } else c.have_cur_eval = 0;
                    c.last_valid = 0;
                    c.state = QN_ST_ITER_END;
                }
            } else {
                if (tid == 0) req_eval_t(c, c.ls_result, QN_ST_AFTER_NEXT, 1); // bfgs.rs:94,98: oracle(x + step*d)
            }
        } break;

        case QN_ST_AFTER_NEXT: { // bfgs.rs:94-102
            double p[3] = {0.0, 0.0, 0.0};
            for (int i = tid; i < n_pad; i += QN_CTL_TPB) {
                const double xn = V.xt[i], gn = V.gt[i];
                const double si = xn - V.x[i]; // s = x+ - x (not t*d), :96
                const double yi = gn - V.g[i]; // :98
                V.s[i] = si; V.y[i] = yi; V.x[i] = xn; V.g[i] = gn;
                p[0] = __builtin_fma(si, si, p[0]);
                p[1] = __builtin_fma(yi, yi, p[1]);
                p[2] = __builtin_fma(yi, si, p[2]);
            }
            ctl_block_sum<3>(p, lds);
            if (c.small_n) {
                __threadfence_block();
                __syncthreads();
                if (tid == 0) { p[0] = ref_dot(V.s, V.s, n); p[1] = ref_dot(V.y, V.y, n); p[2] = ref_dot(V.y, V.s, n); }
            }
            if (tid == 0) {
                c.s_norm = sqrt(p[0]); c.has_s_norm = 1;
                c.y_norm = sqrt(p[1]); c.has_y_norm = 1;
                c.ys = p[2];
                c.f_k = c.f_e;
                c.have_cur_eval = c.memoize;
                c.have_dir = 0;
                c.last_valid = 0;
                if (c.s_norm < c.tol || c.y_norm < c.tol) { // bfgs.rs:106-112: H is not updated
                    c.state = QN_ST_ITER_END;
                } else if (c.small_n) {
                    small_update(V.H, n_pad, n, V.s, V.y, c.method);
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
            double p[3] = {0.0, 0.0, 0.0};
            for (int i = tid; i < n_pad; i += QN_CTL_TPB) {
                const double ui = hp_val(V, nrhs, 0, i);
                const double si = V.s[i];
                V.up[i] = ui; V.sp[i] = si;
                p[0] = __builtin_fma(V.y[i], ui, p[0]);
                if (lazy) {
                    const double gi = V.g[i];
                    p[1] = __builtin_fma(ui, gi, p[1]);
                    p[2] = __builtin_fma(si, gi, p[2]);
                }
            }
            ctl_block_sum<3>(p, lds);
            const double yu = p[0], ug = p[1], sg = p[2];
            double c_ss, c_su, c_uu;
            if (c.method == 0) { const double rho = 1.0 / c.ys; c_su = -rho; c_ss = rho * rho * yu + rho; c_uu = 0.0; }
            else { c_ss = 1.0 / c.ys; c_su = 0.0; c_uu = -1.0 / yu; }
            double q[2] = {0.0, 0.0};
            if (lazy) { // d+ = -(H+ g+) = -(v + c_su (s (u.g) + u (s.g)) + c_ss s (s.g) + c_uu u (u.g)),  v = H g+
                for (int i = tid; i < n_pad; i += QN_CTL_TPB) {
                    const double si = V.sp[i], ui = V.up[i];
                    double w = hp_val(V, nrhs, 1, i);
                    if (c_su != 0.0) w = w + c_su * (si * ug + ui * sg);
                    w = w + c_ss * (si * sg);
                    if (c_uu != 0.0) w = w + c_uu * (ui * ug);
                    const double di = -w;
                    V.d[i] = di;
                    q[0] = __builtin_fma(V.g[i], di, q[0]);
                    q[1] += isfinite(di) ? 0.0 : 1.0;
                }
                ctl_block_sum<2>(q, lds);
            }
            if (tid == 0) {
                c.n_hpasses++;
                if (c.pending) c.n_hpass_rw++;
                c.c_ss = c_ss; c.c_su = c_su; c.c_uu = c_uu;
                c.pending = 1;
                c.tr_updated = 1;
                if (lazy) { c.gd0 = q[0]; c.d_finite = q[1] == 0.0; c.have_dir = 1; }
                c.state = QN_ST_ITER_END;
            }
        } break;

        case QN_ST_ITER_END: { // ls_solver.rs:104-107
            const bool rec = c.k < c.trace_cap;
            if (rec && c.trace_x) {
                double* row = V.xtrace + (size_t)c.k * (size_t)n;
                for (int i = tid; i < n; i += QN_CTL_TPB) row[i] = V.x[i];
            }
            if (tid == 0) {
                if (rec) {
                    QnTraceRec r;
                    r.f = c.tr_f; r.gnorm = c.tr_gnorm; r.t = c.ls_result;
                    r.s_norm = c.has_s_norm ? c.s_norm : NAN;
                    r.y_norm = c.has_y_norm ? c.y_norm : NAN;
                    r.n_evals = c.tr_n_evals; r.ls_iters = c.tr_ls_iters; r.ls_cases = c.tr_ls_cases; r.updated = c.tr_updated;
                    V.trace[c.k] = r;
                }
                c.k += 1;
                c.n_iterations++;
                c.state = QN_ST_LOOP_TOP;
                if (c.callback_mode) c.phase = QN_PH_ITER_DONE;
            }
        } break;

        default: {
            if (tid == 0) { c.status = 4; c.phase = QN_PH_DONE; }
        } break;
        }
    }
    __syncthreads();
    if (tid == 0 && c.phase == QN_PH_RUNNING) { c.status = 4; c.phase = QN_PH_DONE; } // guard tripped
    __syncthreads();
    {
        uint64_t* dst = reinterpret_cast<uint64_t*>(gctl);
        const uint64_t* src = reinterpret_cast<const uint64_t*>(&c);
        for (int i = tid; i < (int)(sizeof(QnCtl) / 8); i += QN_CTL_TPB) dst[i] = src[i];
    }
}

// ------------------------------------------------------------------------------------------------
// plain primitives for the kernel-level FFI (include/qn_hip.h, last section).  No layout assumptions.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prim_gemv_kernel(const double* __restrict__ A, size_t ld, int nrows, int ncols,
                                                        const double* __restrict__ x, double* __restrict__ y) {
    // one wave per row, lanes stride the columns
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= nrows) return;
    const double* a = A + (size_t)row * ld;
    double acc = 0.0;
    for (int j = lane; j < ncols; j += 64) acc = __builtin_fma(a[j], x[j], acc);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = acc + __shfl_xor(acc, off, 64);
    if (lane == 0) y[row] = acc;
}

__global__ __launch_bounds__(256) void prim_rank2_kernel(double* __restrict__ H, size_t ld, int row0, int nrows, int n,
                                                         const double* __restrict__ s, const double* __restrict__ u,
                                                         double c_ss, double c_su, double c_uu) {
    const int il = blockIdx.y;
    if (il >= nrows) return;
    const int i = row0 + il;
    const double si = s[i], ui = u[i];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        double hn = H[(size_t)il * ld + j];
        if (c_su != 0.0) hn = hn + c_su * (si * u[j] + ui * s[j]);
        hn = hn + c_ss * (si * s[j]);
        if (c_uu != 0.0) hn = hn + c_uu * (ui * u[j]);
        H[(size_t)il * ld + j] = hn;
    }
}

__global__ void prim_axpy_kernel(int n, const double* __restrict__ x, double t, const double* __restrict__ d,
                                 double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double td = t * d[i];
        out[i] = x[i] + td;
    }
}

__global__ __launch_bounds__(QN_CTL_TPB) void prim_dot_kernel(int n, const double* __restrict__ a, const double* __restrict__ b,
                                                              double* __restrict__ out) {
    __shared__ double lds[16];
    double p[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += QN_CTL_TPB) p[0] = __builtin_fma(a[i], b[i], p[0]);
    ctl_block_sum<1>(p, lds);
    if (threadIdx.x == 0) *out = p[0];
}
