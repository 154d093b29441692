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
// CODE WARM-UP.  A kernel's instructions are fetched through the XCD's L2 like its data, and between two launches of the same kernel the
// iteration streams hundreds of megabytes through that L2 and the Infinity Cache behind it: the instruction fetches of a launch start cold,
// one line after the other as the wave reaches them.  For the kernels whose critical path is ONE wave running the solver's state machine
// (straight-line code, ~700 instructions of a 25 KB kernel) that is latency in front of everything: the 32-workgroup accept-reduce took
// 6.4 us or 7.1 us with the SAME instruction stream depending on where the loader had put the code object (profiles/r05_o_*: 320 KB of
// never-launched code behind the last kernel was enough; neither moving the kernels inside the object, nor the data's addresses,
// nor a smaller object brought it back).  So an otherwise idle wave reads the kernel's own bytes as DATA at entry -- `line` counts 128-byte
// lines from the program counter on, one request per lane, all in flight at once -- and the fetches of the wave that executes them hit:
// 7.1 -> 5.8 us on that kernel wherever the code lies.  The value is consumed by a store that never happens (the loads must not be dropped).
// The read is NOT bounded by the kernel's size: it must stay inside the code object's .text -- tools/check_code_warm.py, run by the Makefile after
// every link, fails the build when a warmed kernel ends less than 32 KB in front of the end of .text.
__device__ __forceinline__ unsigned qn_code_warm_issue(const int line) {
    unsigned long long pc;
    asm volatile("s_getpc_b64 %0" : "=s"(pc));
    return ((const volatile unsigned*)(pc & ~127ull))[(size_t)line * 32];
}
__device__ __forceinline__ void qn_code_warm_done(const unsigned c, const bool never, double* sink) {
    if (c == 0x9e3779b9u && never) sink[0] = 1.0;
}


#define QN_TPB 256        // threads per workgroup of the streaming kernels
#define QN_CHUNK 512      // columns per chunk (2 per thread)
#define QN_CTL_TPB 1024   // max threads of the control workgroup (generic path); the fused path launches 256

struct QnTile {
    int n;       // logical dimension
    int n_pad;   // padded dimension = leading dimension = world * rpr
    int rpr;     // rows per rank (multiple of 16)
    int row_off; // global index of this rank's first row
    int cs;      // column splits (gridDim.y)
    int rank;
};

#define QN_NEVP 9 // partial sums per workgroup of an evaluation (fused path)
#define QN_NHPP 3 // partial sums per workgroup of an update pass (fused path)

// buffers of the fused fast path (qn_fused.hip.h)
struct QnFused {
    double* X0;   // [2][n_pad]: x = X0 + xc*n_pad, trial point in the other half
    double* S0;   // [2][n_pad]: pending s = S0 + sc*n_pad, staged s in the other half
    double* G;    // gradient at x_k (committed by h_pass row-block 0)
    double* GT;   // gradient at the last evaluated point (gathered)
    double* Y;    // gt - g at the last evaluated point (gathered)
    double* UN;   // u = H y of the last update pass (gathered)
    double* UP;   // copy of UN read by the next update pass as the pending u
    double* VV;   // v = H g+ (gathered)
    double* evp;  // [world][QN_NEVP][nblk]
    double* hpp;  // [world][QN_NHPP][nblk]
    const double* b;
    int nblk;     // row tiles per rank
    int pworld;   // ranks whose partial rows evp / hpp hold: the context's world on the row path, 1 on the symmetric-storage paths
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
    unsigned long long* dbg; // diagnostic builds only (QN_CTL_STAMPS): in-kernel time stamps
    QnFused F; // fused fast path buffers (valid when ctl->fused)
    int fused_hint;   // set by the host for runs on the fused path (== ctl->fused)
    const double *lb, *ub;   // solver bounds (BFGSB / DFPB / SR1B), n_pad entries, padding -inf / +inf
    const double *llb, *lub; // the bounded line search's own box
    const int* nfail; // Newton: set by the factorisation when the Hessian is not positive definite
    int n, n_pad, rpr, world, hcs, qcs;
};

__device__ __forceinline__ v2d ld2(const double* p) { return *reinterpret_cast<const v2d*>(p); }
__device__ __forceinline__ void st2(double* p, v2d v) { *reinterpret_cast<v2d*>(p) = v; }

// ---- lane ^ OFF exchange of a double without the LDS crossbar: __shfl_xor compiles to two ds_bpermute_b32 (an LDS-pipeline
// round trip each, ~100+ cycles, plus an address register); on gfx950 every power-of-two xor pattern is a VALU data-path
// operation: quad_perm (1, 2), row_shl / row_shr 4 (4), row_ror 8 (8), v_permlane16_swap (16), v_permlane32_swap (32).
// Checked lane for lane on the device by tools/xor_shuffle_probe.hip.
template <int OFF>
__device__ __forceinline__ int qn_xor_lanes_i(const int v) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "power of two below the wave size");
    if constexpr (OFF == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    else if constexpr (OFF == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false); // quad_perm [2,3,0,1]
    else if constexpr (OFF == 4) {
        const int a = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0xf, false); // row_shl:4 : lane i <- lane i + 4
        const int b = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false); // row_shr:4 : lane i <- lane i - 4
        return (threadIdx.x & 4) ? b : a;
    } else if constexpr (OFF == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false); // row_ror:8 within rows of 16
    else if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
}
template <int OFF>
__device__ __forceinline__ double qn_xor_lanes(const double v) { // (block sizes are multiples of 64: threadIdx.x & OFF == lane & OFF)
    return __hiloint2double(qn_xor_lanes_i<OFF>(__double2hiint(v)), qn_xor_lanes_i<OFF>(__double2loint(v)));
}
// sum over the wave, every lane gets the total (order: xor 32, 16, 8, 4, 2, 1 -- as the `for (off = 32; off; off >>= 1)` loops it replaces)
__device__ __forceinline__ double qn_wave_sum(double v) {
    v = v + qn_xor_lanes<32>(v); v = v + qn_xor_lanes<16>(v); v = v + qn_xor_lanes<8>(v);
    v = v + qn_xor_lanes<4>(v); v = v + qn_xor_lanes<2>(v); v = v + qn_xor_lanes<1>(v);
    return v;
}

// (The swap form below is used for the 16-value folds of the tile kernels' row totals only.  Measured with rocprofv3 on one box:
// there it takes 0.7 us off the evaluation kernel; used in the 8-value folds as well -- the prologue's table sums, the small
// kernels' dot products -- the accept-reduce kernel got 1.2 us SLOWER and the update-reduce 0.4, for reasons the instruction
// counts do not show.)
#ifndef QN_FOLD_SWAP_MIN_V
#define QN_FOLD_SWAP_MIN_V 16
#endif
// ---- halving butterfly: V values per lane -> lane l ends with the wave total of value (l >> (6 - log2 V)) in v[0]
template <int CNT, int OFF, bool SWAP = false> // SWAP: the swap form (below) whatever the number of values
struct QnWaveFold {
    template <int V>
    static __device__ __forceinline__ void run(double (&v)[V], int lane) {
        if constexpr (OFF >= 1) {
            if constexpr (CNT > 1) {
                constexpr int HALF = CNT / 2;
                if constexpr ((OFF == 32 || OFF == 16) && (SWAP || V >= QN_FOLD_SWAP_MIN_V)) {
                    // v_permlane32_swap A, B exchanges the upper half of A with the lower half of B (v_permlane16_swap: the odd
                    // 16-lane rows of A with the even rows of B): afterwards A' + B' IS the folded pair -- value i in the lanes
                    // with (lane & OFF) == 0, value i + HALF in the others, each lane holding its own entry plus its partner's.
                    // No selects and no copies: three instructions per exchange instead of nine (round 3: the phase behind the
                    // workgroup barrier of the tile kernels is instruction issue).  Same pairs, and a + b = b + a: same bits.
#pragma unroll
                    for (int i = 0; i < HALF; ++i) {
                        const int alo = __double2loint(v[i]), ahi = __double2hiint(v[i]);
                        const int blo = __double2loint(v[i + HALF]), bhi = __double2hiint(v[i + HALF]);
                        if constexpr (OFF == 32) {
                            const auto rl = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
                            const auto rh = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
                            v[i] = __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
                        } else {
                            const auto rl = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
                            const auto rh = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
                            v[i] = __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
                        }
                    }
                } else {
                    const bool up = (lane & OFF) != 0;
#pragma unroll
                    for (int i = 0; i < HALF; ++i) {
                        const double keep = up ? v[i + HALF] : v[i];
                        const double send = up ? v[i] : v[i + HALF];
                        const double recv = qn_xor_lanes<OFF>(send);
                        v[i] = keep + recv;
                    }
                }
                QnWaveFold<HALF, OFF / 2, SWAP>::run(v, lane);
            } else {
                v[0] = v[0] + qn_xor_lanes<OFF>(v[0]);
                QnWaveFold<1, OFF / 2, SWAP>::run(v, lane);
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
    const double *llb, *lub; // bounded backtracking: the trial point is projected onto the line search's box
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
    bool project = false;
    if (a.expect_phase >= 0) {
        if (a.ctl->phase != a.expect_phase) return;
        kind = a.ctl->req_kind;
        t = a.ctl->req_t;
        project = a.ctl->req_project != 0;
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
                if (project) {
                    const v2d lo = ld2(a.llb + j), hi = ld2(a.lub + j);
                    xj.x = fmin(fmax(xj.x, lo.x), hi.x);
                    xj.y = fmin(fmax(xj.y, lo.y), hi.y);
                }
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
                                   int n_pad, const QnCtl* ctl, int expect_phase, const double* __restrict__ llb,
                                   const double* __restrict__ lub) {
    if (ctl->phase != expect_phase) return;
    const int kind = ctl->req_kind;
    const double t = ctl->req_t;
    const bool project = ctl->req_project != 0; // BackTrackingB projects its trial points (backtracking_b.rs:67)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) {
        double v = x[i];
        if (kind == QN_REQ_T) {
            const double td = t * d[i];
            v = v + td;
            if (project) v = fmin(fmax(v, llb[i]), lub[i]);
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
// LDS_ONLY (ctl_step_kernel): the two barriers order the workgroup's LDS traffic and nothing else.  __syncthreads() also waits for every
// global store of the wave to be acknowledged -- a memory round trip (1.5-2 us on bytes another XCD's kernel has just written) in front of each
// of a vector state's sums, which the control kernel does not need: inside one launch an entry of an n-vector is only ever touched by thread
// (i mod blockDim) -- program order -- and the states are separated by the loop's own __syncthreads().  In-kernel stamps at n = 4096,
// bounded run: the two-sweep state behind an update pass 15.6 us, the one-sweep state behind an evaluation 7.3 us (tools/ctl_stamps_bounded.py).
// THE INVARIANT this rests on -- an n-vector entry is touched by ONE thread per launch -- is stated, not checked by the compiler: a state that reads a
// neighbour's entry would race silently.  -DQN_CTL_FULL_BARRIERS (csrc/Makefile target `fullbar`, built by __graft_entry__.build()) turns every
// LDS-only barrier back into __syncthreads(); tests/test_gpu_bounded.py::test_lds_only_barriers_equal_full_barriers_bit_for_bit runs the bounded
// generic path through both builds and compares the bits.
#ifdef QN_CTL_FULL_BARRIERS
__device__ __forceinline__ void qn_lds_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void qn_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
template <int K, bool LDS_ONLY = false>
__device__ __forceinline__ void ctl_block_sum(double (&v)[K], double* lds /* 16*K */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = qn_wave_sum(v[k]);
    }
    if (LDS_ONLY) qn_lds_barrier(); else __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
    }
    if (LDS_ONLY) qn_lds_barrier(); else __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = t + lds[w * K + k];
        v[k] = t;
    }
}

template <bool LDS_ONLY = false>
__device__ __forceinline__ double ctl_block_fmax(double v, double* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    if (LDS_ONLY) qn_lds_barrier(); else __syncthreads();
    if (lane == 0) lds[wave] = v;
    if (LDS_ONLY) qn_lds_barrier(); else __syncthreads();
    double t = -INFINITY;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = fmax(t, lds[w]);
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
// (By value and by reference, with selects: as `double* tl, double* tu` the compiler merged the stores `*tl = t` / `*tu = t` into
// ONE store through a selected pointer, which forced both locals into scratch memory -- an indexed scratch store, two scratch
// loads and an s_waitcnt vmcnt(0) in the middle of the state machine, i.e. a trip to memory behind whatever the kernel streams.)
__device__ __forceinline__ int mt_update_interval(double f_tl, double f_t, double g_t, double& tl, double t, double& tu) {
    const double tl0 = tl, tu0 = tu;
    const double w = g_t * (tl0 - t);
    double ntl = tl0, ntu = tu0;
    int conv = 0;
    if (f_t > f_tl) ntu = t;                 // U1
    else if (w > 0.) ntl = t;                // U2
    else if (w < 0.) { ntu = tl0; ntl = t; } // U3
    else conv = 1;                           // interval converged to a point
    tl = ntl; tu = ntu;
    return conv;
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
template <int LEAN = 0>
__device__ __forceinline__ void req_eval_t(QnCtl& c, double t, int after_state, int need_vectors, int project = 0) {
    c.n_oracle_calls++;
    c.tr_n_evals++;
    if ((LEAN && !project) || (!LEAN && c.memoize && !project && !c.last_projected)) { // (a projected trial is not x + t d: no memo; LEAN == 2 forgets the memo when it consumes one)
        if (!need_vectors && t == 0.0 && c.d_finite) { // x + 0*d == x: phi(0) = (f_k, g_k.d)
            c.f_e = c.f_k; c.gd_e = c.gd0; c.state = after_state;
            return;
        }
        if (c.last_valid && c.last_t == t) {
            c.f_e = c.f_last; c.gd_e = c.gd_last;
            if ((LEAN || c.sym2) && need_vectors) { // the scalars are known; the vectors of that point are still slots (qn_sym2.hip.h)
                c.after_state = after_state;
                c.phase = QN_PH_REQ_VEC;
                return;
            }
            c.state = after_state;
            return;
        }
    }
    c.req_kind = QN_REQ_T;
    c.req_t = t;
    c.req_project = project;
    c.req_need_vectors = need_vectors;
    c.after_state = after_state;
    c.phase = (!LEAN && c.defer_u) ? QN_PH_REQ_HPASS_EVAL : QN_PH_REQ_EVAL;
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
__device__ __forceinline__ void small_direction(const double* H, int ld, int n, const double* g, double* d, double* y /* LDS, >= 5 */,
                                                const double* x = nullptr, const double* lb = nullptr, const double* ub = nullptr) {
    for (int i = 0; i < n; ++i) y[i] = H[i * ld] * g[0];
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < n; ++i) y[i] = H[i * ld + j] * g[j] + y[i];
    if (lb) { // bfgs_b.rs:72-75: P(x - H g) - x
        for (int i = 0; i < n; ++i) { double t = x[i] - y[i]; t = fmin(fmax(t, lb[i]), ub[i]); d[i] = t - x[i]; }
    } else {
        for (int i = 0; i < n; ++i) d[i] = -y[i];
    }
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
__device__ __forceinline__ void small_update(double* H, int ld, int n, const double* s, const double* y, int method,
                                             double* scratch /* LDS, >= 5*25 + 5 doubles */) {
    constexpr int MM = QN_SMALL_N * QN_SMALL_N;
    double *h = scratch, *m0 = scratch + MM, *m1 = scratch + 2 * MM, *m2 = scratch + 3 * MM, *m3 = scratch + 4 * MM;
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
    } else if (method == 4) { // sr1_b.rs:143-146: hy = H y ; shy = s - hy ; H += shy shy' / shy.dot(y)
        double* w = scratch + 5 * MM;
        for (int i = 0; i < n; ++i) w[i] = h[i] * y[0];
        for (int j = 1; j < n; ++j)
            for (int i = 0; i < n; ++i) w[i] = h[i + j * QN_SMALL_N] * y[j] + w[i];
        for (int i = 0; i < n; ++i) w[i] = s[i] - w[i];
        const double den = ref_dot(w, y, n);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) H[i * ld + j] = h[i + j * QN_SMALL_N] + (w[i] * w[j]) / den;
    } else {
        const double sy = ref_dot(s, y, n);
        double* u = scratch + 5 * MM;
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
// log-sum-exp objective: f = log sum_i exp(a_i'x + c_i) + mu/2 ||x||^2,
// g = A' softmax(Ax + c) + mu x.  Max-shifted for stability.  n_pad <= 16384: ONE pass over this rank's rows of A per evaluation
// (8*m*n/P bytes, lse_onepass_kernel below).  Wider problems: two passes (16*m*n/P bytes): z = A x (h_pass_kernel in mat-vec
// mode), then the column sums A'w.
// ------------------------------------------------------------------------------------------------
struct QnLseArgs {
    const double* A;   // this rank's rows, [mrpr][n_pad]
    const double* c;   // m_pad
    double* z;         // gathered A x: rank block p at z + p*2*mrpr (h_pass layout, one rhs)
    double* w;         // m_pad softmax weights
    double* gpart;     // [rs][n_pad] column-sum partials of this rank
    double* gall;      // gathered per-rank gradients [world][n_pad]
    const double* x;   // evaluation point (n_pad)
    double* f_out;     // scalar
    double* g_out;     // n_pad
    double* scal;      // [0] = zmax + log(sum), [1] = ||x||^2
    double mu;
    int m, m_pad, mrpr, n, n_pad, world, rank, rs;
};

__device__ __forceinline__ double lse_z(const QnLseArgs& a, int i) { // gathered layout of h_pass output, one rhs, one column split
    const int p = i / a.mrpr, il = i - p * a.mrpr;
    return a.z[(size_t)p * 2 * a.mrpr + il];
}

// single workgroup: softmax weights of all m rows (+ ||x||^2)
__global__ __launch_bounds__(1024) void lse_softmax_kernel(const QnLseArgs a) {
    __shared__ double lds[32];
    const int tid = threadIdx.x, tpb = blockDim.x;
    double mx = -INFINITY;
    for (int i = tid; i < a.m; i += tpb) mx = fmax(mx, lse_z(a, i) + a.c[i]);
    mx = ctl_block_fmax(mx, lds);
    double p[2] = {0.0, 0.0};
    for (int i = tid; i < a.m; i += tpb) {
        const double e = exp(lse_z(a, i) + a.c[i] - mx);
        a.w[i] = e;
        p[0] += e;
    }
    for (int j = tid; j < a.n; j += tpb) p[1] = __builtin_fma(a.x[j], a.x[j], p[1]);
    ctl_block_sum<2>(p, lds);
    for (int i = tid; i < a.m_pad; i += tpb) a.w[i] = (i < a.m) ? a.w[i] / p[0] : 0.0;
    if (tid == 0) { a.scal[0] = mx + log(p[0]); a.scal[1] = p[1]; }
}

// column sums over this rank's rows: gpart[rs][j] = sum_{i in split rs} w_i A[i][j]; grid (n_pad/512, rs)
__global__ __launch_bounds__(QN_TPB) void lse_colsum_kernel(const QnLseArgs a) {
    const int j = blockIdx.x * QN_CHUNK + 2 * threadIdx.x;
    if (j >= a.n_pad) return;
    const int rows_per = (a.mrpr + a.rs - 1) / a.rs;
    const int i0 = blockIdx.y * rows_per, i1 = min(a.mrpr, i0 + rows_per);
    const double* wl = a.w + (size_t)a.rank * a.mrpr;
    v2d acc = {0.0, 0.0};
    int i = i0;
    for (; i + 4 <= i1; i += 4) {
        v2d h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = ld2(a.A + (size_t)(i + r) * a.n_pad + j);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double wi = wl[i + r];
            acc.x = __builtin_fma(wi, h[r].x, acc.x);
            acc.y = __builtin_fma(wi, h[r].y, acc.y);
        }
    }
    for (; i < i1; ++i) {
        const v2d h = ld2(a.A + (size_t)i * a.n_pad + j);
        const double wi = wl[i];
        acc.x = __builtin_fma(wi, h.x, acc.x);
        acc.y = __builtin_fma(wi, h.y, acc.y);
    }
    st2(a.gpart + (size_t)blockIdx.y * a.n_pad + j, acc);
}

// this rank's gradient contribution: sum of the row splits (fixed order)
__global__ void lse_reduce_splits_kernel(const QnLseArgs a) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < a.n_pad; j += gridDim.x * blockDim.x) {
        double t = a.gpart[j];
        for (int s = 1; s < a.rs; ++s) t = t + a.gpart[(size_t)s * a.n_pad + j];
        a.gall[(size_t)a.rank * a.n_pad + j] = t;
    }
}

// g = sum over ranks (fixed order) + mu x ; f = zmax + log(sum) + mu/2 ||x||^2
__global__ void lse_finish_kernel(const QnLseArgs a) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < a.n_pad; j += gridDim.x * blockDim.x) {
        double t = a.gall[j];
        for (int p = 1; p < a.world; ++p) t = t + a.gall[(size_t)p * a.n_pad + j];
        a.g_out[j] = (j < a.n) ? t + a.mu * a.x[j] : 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.f_out = a.scal[0] + 0.5 * a.mu * a.scal[1];
}

// ---- the same evaluation in ONE pass over A (n_pad <= 16384) ----
// A softmax needs max_i z_i before any weight is known -- hence the two passes above.  With the running rescale of a streaming
// softmax one pass is enough: a workgroup walks its rows with a running maximum m, S = sum_i exp(z_i - m) and
// G = sum_i exp(z_i - m) a_i; a new row updates m' = max(m, z), S <- S exp(m - m') + exp(z - m'), G <- G exp(m - m') + exp(z - m') a.
// Each of the 512 threads keeps its KCH x 2 columns of the current row, of the prefetched next row and of G in registers
// (3 x 64 VGPRs at n = 16384); x sits in LDS (128 KB at n = 16384); the row's dot product is the only cross-thread step (wave
// butterfly, eight partials through LDS, ONE barrier per row -- a row is 128 KB, 5 us of HBM time per CU, the barrier hides in it).
// lse_combine_kernel folds the workgroups' (m, S, G) in workgroup order, the ranks' results are exchanged once, lse_finish1_kernel
// folds them in rank order: f = M + log S + mu/2 ||x||^2, g = G / S + mu x.  Every order is fixed: reproducible bit for bit.
// NTA: the rows of A as non-temporal loads -- A is read once per evaluation and, at the sizes this kernel is for, does not fit the
// Infinity Cache (n = m = 16384: 2.1 GB): nothing of it is worth keeping (tools/stream_shape_probe.hip: 6.5-6.85 TB/s against 6.1-6.3)
template <int KCH, bool NTA>
__global__ __launch_bounds__(512) void lse_onepass_kernel(const QnLseArgs a, double* __restrict__ wgms, double* __restrict__ wgg) {
    extern __shared__ __attribute__((aligned(16))) double lse_x[]; // KCH * 1024 entries of x, zero past n_pad
    __shared__ double red[2][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int np = a.n_pad;
    // column pair k of this thread: 2 tid + 1024 k; past the end the last pair is re-read (its x entries are zero, its G entries
    // are never stored)
#define QN_LSE_JC(k) min(2 * tid + 1024 * (k), np - 2)
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
        const int j = 2 * tid + 1024 * k;
        v2d xv = {0.0, 0.0};
        if (j < np) xv = ld2(a.x + j);
        lse_x[j] = xv.x; lse_x[j + 1] = xv.y;
    }
    __syncthreads();
    const int G = gridDim.x, per = (a.mrpr + G - 1) / G;
    const int r_lo = blockIdx.x * per;
    const int r_hi = min(min(a.mrpr, r_lo + per), a.m - a.rank * a.mrpr); // (rows past m are padding)
    v2d g[KCH], cur[KCH], nxt[KCH];
#pragma unroll
    for (int k = 0; k < KCH; ++k) { g[k] = (v2d){0.0, 0.0}; cur[k] = (v2d){0.0, 0.0}; }
    double m_run = -INFINITY, s_run = 0.0;
    if (r_lo < r_hi) {
#pragma unroll
        for (int k = 0; k < KCH; ++k) cur[k] = NTA ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(a.A + (size_t)r_lo * np + QN_LSE_JC(k))) : ld2(a.A + (size_t)r_lo * np + QN_LSE_JC(k));
    }
    for (int r = r_lo; r < r_hi; ++r) {
        const double* nrow = a.A + (size_t)((r + 1 < r_hi) ? r + 1 : r) * np; // (last row: a harmless re-read, no branch)
#pragma unroll
        for (int k = 0; k < KCH; ++k) nxt[k] = NTA ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(nrow + QN_LSE_JC(k))) : ld2(nrow + QN_LSE_JC(k));
        double p = 0.0;
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int j = 2 * tid + 1024 * k;
            const v2d xv = *reinterpret_cast<const v2d*>(&lse_x[j]);
            p = __builtin_fma(cur[k].x, xv.x, p);
            p = __builtin_fma(cur[k].y, xv.y, p);
        }
        p = qn_wave_sum(p);
        if (lane == 0) red[r & 1][wave] = p;
        __syncthreads();
        double z = red[r & 1][0];
#pragma unroll
        for (int w = 1; w < 8; ++w) z = z + red[r & 1][w];
        z = z + a.c[a.rank * a.mrpr + r];
        const double m_new = fmax(m_run, z);
        const double scale = exp(m_run - m_new), e = exp(z - m_new); // (first row: exp(-inf) = 0)
        s_run = __builtin_fma(s_run, scale, e);
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            g[k].x = __builtin_fma(e, cur[k].x, g[k].x * scale);
            g[k].y = __builtin_fma(e, cur[k].y, g[k].y * scale);
            cur[k] = nxt[k];
        }
        m_run = m_new;
    }
    if (tid == 0) { wgms[2 * blockIdx.x] = m_run; wgms[2 * blockIdx.x + 1] = s_run; }
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
        const int j = 2 * tid + 1024 * k;
        if (j < np) st2(wgg + (size_t)blockIdx.x * np + j, g[k]);
    }
#undef QN_LSE_JC
}

// this rank's (m, S, G): the workgroups' results folded in a fixed order -- each quarter of the workgroups in workgroup order, 16
// loads in flight, the four quarters then added in order; G into gall[rank], (m, S) into lms[rank].  64 columns per workgroup.
__global__ __launch_bounds__(256) void lse_combine_kernel(const QnLseArgs a, int G, const double* __restrict__ wgms, const double* __restrict__ wgg,
                                                          double* __restrict__ lms) {
    __shared__ double fac[256];
    __shared__ double lds[32];
    __shared__ double part[3][64];
    const int tid = threadIdx.x;
    const double mw = tid < G ? wgms[2 * tid] : -INFINITY;
    const double mr = ctl_block_fmax(mw, lds);
    // (a workgroup without rows: exp(-inf) = 0; a RANK without rows -- every maximum -inf -- would make exp(-inf - (-inf)) = NaN)
    fac[tid] = (tid < G && mw != -INFINITY) ? exp(mw - mr) : 0.0;
    __syncthreads();
    const int c = tid & 63, q = tid >> 6;
    const int j = min(blockIdx.x * 64 + c, a.n_pad - 1);
    const int per = (G + 3) / 4, w_lo = q * per, w_hi = min(G, w_lo + per);
    double acc = 0.0;
    for (int w0 = w_lo; w0 < w_hi; w0 += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (w0 + u < w_hi) ? wgg[(size_t)(w0 + u) * a.n_pad + j] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_fma(v[u], (w0 + u < w_hi) ? fac[w0 + u] : 0.0, acc);
    }
    if (q > 0) part[q - 1][c] = acc;
    __syncthreads();
    if (q == 0 && blockIdx.x * 64 + c < a.n_pad) a.gall[(size_t)a.rank * a.n_pad + j] = ((acc + part[0][c]) + part[1][c]) + part[2][c];
    if (blockIdx.x == 0 && tid == 0) {
        double s = 0.0;
        for (int w = 0; w < G; ++w) s = __builtin_fma(wgms[2 * w + 1], fac[w], s);
        lms[2 * a.rank] = mr; lms[2 * a.rank + 1] = s;
    }
}

// the ranks' (m, S, G) folded in rank order: g = G / S + mu x ; f = M + log S + mu/2 ||x||^2
__global__ __launch_bounds__(256) void lse_finish1_kernel(const QnLseArgs a, const double* __restrict__ lms) {
    __shared__ double lds[32];
    double M = lms[0];
    for (int p = 1; p < a.world; ++p) M = fmax(M, lms[2 * p]);
    double S = 0.0;
    for (int p = 0; p < a.world; ++p) S = __builtin_fma(lms[2 * p + 1], lms[2 * p] != -INFINITY ? exp(lms[2 * p] - M) : 0.0, S); // (a rank that owns no real row contributes nothing)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < a.n_pad; j += gridDim.x * blockDim.x) {
        double t = 0.0;
        for (int p = 0; p < a.world; ++p) t = __builtin_fma(a.gall[(size_t)p * a.n_pad + j], lms[2 * p] != -INFINITY ? exp(lms[2 * p] - M) : 0.0, t);
        a.g_out[j] = (j < a.n) ? t / S + a.mu * a.x[j] : 0.0;
    }
    if (blockIdx.x == 0) {
        double p[1] = {0.0};
        // (sixteen entries requested at a time, added in the order of before: one thread's 64 entries at n = 16384 were 64 dependent
        // round trips -- 12 of this launch's 17.6 us)
        for (int j0 = threadIdx.x; j0 < a.n; j0 += 16 * blockDim.x) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int j = j0 + u * blockDim.x; v[u] = j < a.n ? a.x[j] : 0.0; }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (j0 + u * (int)blockDim.x < a.n) p[0] = __builtin_fma(v[u], v[u], p[0]);
        }
        ctl_block_sum<1>(p, lds);
        if (threadIdx.x == 0) *a.f_out = M + log(S) + 0.5 * a.mu * p[0];
    }
}

#include "qn_fused.hip.h"
#include "qn_sym.hip.h"
#include "qn_newton.hip.h"
#include "qn_lu.hip.h"
#include "qn_lu_split.hip.h"
#include "qn_ctl_step.hip.h"
#include "qn_sym2.hip.h"
#include "qn_sym2r.hip.h"
#include "qn_sym2sh.hip.h"
#include "qn_sym2g.hip.h"

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
    acc = qn_wave_sum(acc);
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
