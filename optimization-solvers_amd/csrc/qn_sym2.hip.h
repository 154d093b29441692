// qn_sym2.hip.h -- the fused fast path on the symmetric half of H and Q, second generation (single rank).
//
// What round 1's profile showed at n = 4096 (profiles/r01_h_kernel_stats_n4096.csv): 8 dependent launches per BFGS
// iteration, 34 us of the 101 us in three kinds of small serial kernels (slot reduction after EVERY evaluation, the
// single-workgroup control step after every evaluation), an update-tile kernel at 202 VGPRs = ONE workgroup per CU, and a grid
// of 528 tiles on 512 (256) resident slots.  This file removes those four things; the arithmetic of the iteration is unchanged.
//
//  1. The control step is no launch any more.  EVERY workgroup of EVERY kernel runs the solver's state machine
//     (qn_ctl_step.hip.h ctl_scalar_run, the same device function the generic path's control kernel runs) in its prologue, from
//     the same inputs, hence to the same bits; workgroup 0 alone performs the side effects (control block write-back, trace).
//     The control block is double-buffered: kernel i reads ctl2[i & 1] and writes ctl2[(i + 1) & 1], so a late workgroup can
//     never see the state its own launch produced.  The prologue costs ~2-3 us of one thread's dependent arithmetic, hidden
//     behind the first tile's loads, which are issued before it.
//  2. An evaluation hands the line search only what it needs: two scalars.  f(x + t d) = 1/2 xt'Q xt - b'xt and
//     g(x + t d)'d = d'Q xt - b'd are sums of per-tile bilinear forms, so the tile kernel emits them per WORKGROUP (fixed
//     order inside, fixed order across workgroups) and the next kernel's prologue sums <= 512 of them.  The row / column
//     slots are still written, but they are reduced to VECTORS (g+, y, x+, s and their norms) only for the point the line
//     search accepts: once per iteration instead of once per evaluation.  The same holds for the update pass:
//     y'H+y and g+'H+y come out of the tiles as scalars.
//  3. Work items, not tiles, are the launch unit: the grid is min(items, 256) workgroups = one per CU, each walking a static,
//     cost-balanced list (longest-processing-time assignment on the host).  A diagonal tile is processed as its upper
//     triangle only (waves skip the lanes left of their rows: 56 % of a tile's bytes), so n = 4096 is 496 + 32 x 0.56 = 514
//     units, 2.0 per CU, instead of 528 tiles on 256 one-workgroup slots (three rounds).  The strictly lower part of H's
//     diagonal tiles is restored with the rest of the lower triangle when a reader needs whole rows (sym2_mirror_kernel).
//  4. Registers.  Two 8-wave workgroups per CU would need the tile kernels inside 128 VGPRs; with a load window, the column
//     vectors and the accumulators of two right-hand sides the compiler (hipcc 7.2) could not hold the window in registers at
//     that budget -- it parked every freshly loaded row in scratch memory and waited for it, which serialises exactly the loads
//     the window exists to overlap.  So the kernels take the other road to the same bytes in flight: ONE workgroup per CU
//     (__launch_bounds__(512, 2): 256 VGPRs), the wave's whole 16-row share of a tile in registers, and every register refilled
//     with the same row of the workgroup's NEXT item the moment its row is consumed: 128 KB of loads in flight per CU across
//     item boundaries, reductions and barriers.  Row-side inputs (x_i, d_i, s_i, u_i, y_i, g_i) live one row per lane and are
//     broadcast into scalar registers as the unrolled row loop needs them: no LDS staging, no barrier before the loop.
//
// Launch pattern of one iteration (More-Thuente on the quadratic: two evaluations, the second one accepted):
//     eval tiles | eval tiles | accept-reduce (nb workgroups) | update tiles | update-reduce (nb workgroups)
// 5 launches instead of 8, every one of them predicated on the control block, so the host still enqueues a fixed pattern
// hundreds of iterations ahead (pipelined mode) or services one request at a time (synchronous mode: the same kernels plus a
// one-workgroup `advance` launch that runs the prologue alone -- bit-identical by construction).
// Reference lines served: ls_solver.rs:66-111, bfgs.rs:42-127, dfp.rs:115-120, morethuente.rs:165-297, backtracking.rs:20-58.
#pragma once
#include <type_traits>

#define QN_S2_TPB 512
#define QN_S2_WAVES 8
#define QN_S2_RPW 16   // rows of a tile per wave
#define QN_S2_MAXG 256 // workgroups of a tile launch: one 8-wave workgroup per CU (see the note on registers below)
#define QN_S2_NSE 6    // evaluation scalars per workgroup: xt'Q xt, b'xt, d'Q xt, b'd, g'd, #non-finite d
#define QN_S2_NR 5     // accept-reduce partials per block-row: y'y, y's, g+'g+, s's, s'g+
#define QN_S2_ROW 8    // doubles per partial row in memory (64 B: loaded as 16-byte pieces)
                       // (generic objectives, qn_sym2g.hip.h: the accepted-point sums a combine launch stages with every evaluation live in a
                       // SECOND table of the same shape, QnS2Args.wgV, so that this one keeps its layout and its size)
#define QN_S2SH_NEC 4  // row-sharded runs: the evaluation scalars that are exchanged, per workgroup: x'(Q xt - 2 b), d'(Q xt - b), g'd,
                       // #non-finite d (the table's b'xt and b'd columns are zero by construction: see CONDITIONING below)
#define QN_S2SH_EB 4   // ... and the slices of them a prologue requests at a time
#define QN_S2_CNT_STRIDE 1056 // ints between two block-rows' arrival counters: 4 KB + 128 B, so that the counters fall on different memory
                              // channels (at 128 B apart all of them sat behind ONE channel: the tail took 13 us -- profiles/r05_b_*)
#define QN_S2_TRED_MAXK 31  // tail reduce: a workgroup's contributions (two per item and the sliver's) are dealt to the lanes of one wave

enum { QN_S2_ADVANCE = 0, QN_S2_EVAL = 1, QN_S2_VEC = 2, QN_S2_HTILE = 3, QN_S2_HREDUCE = 4,
       QN_S2_VSUM = 5, QN_S2_HSUM = 6, // (row-sharded runs, qn_sym2sh.hip.h: this rank's slot sums, in front of the exchange of an n-vector)
       QN_S2_GEVAL_A = 7, QN_S2_GCOMB = 8, QN_S2_GHT_A = 9, // (generic objectives, qn_sym2g.hip.h: the machine in a one-workgroup launch in
                                                             // front of many-workgroup kernels that only READ the control block)
       QN_S2_DIR = 10,   // (bounded variants: s2_dir_kernel)
       QN_S2_PROJ = 11, // (BackTrackingB on this path, round 6: s2_proj_kernel -- the projected trial point, stored, and ||P(x + t d) - x||^2)
       QN_S2_VECD = 12 }; // (row-sharded, the trial's partial vector riding on the scalar exchange -- qn_context_set_trial_vector_exchange, qn_sym2sh.hip.h:
                          //  the accept-reduce launch itself is where the machine sees the accepted evaluation: no partial-sum launch, no exchange in front of it)

struct QnS2Args {
    const double* Q;
    double* H;
    int n, np, nb, G;
    const int* item_ij;  // [maxk][G]: k-th item of workgroup g = (I << 16) | J, J >= I (J == I: diagonal tile, upper triangle only);
                         // -1 = the list has ended.  Transposed so that a workgroup's first item is ONE coalesced load away.
    int maxk;            // longest list
    int inorder;         // the first `inorder` items were dealt in order: item t is the (t / G)-th of workgroup t mod G
    QnFused F;           // X0, S0, G, GT, Y, UN, VV, b (UP is not used here: no kernel writes u while another reads it)
    double* part;        // [nb][nb][2][128] row / column slots of the update pass ([y, g+])
    double* partE;       // [nb][nb][128] row / column slots of the last evaluation (Q (x + t d)): a buffer of its own, because the launch that
                         // turns them into vectors also writes the update pass's slots (folded accept-reduce)
    int pair;            // every workgroup has two list items and a sliver: s2_eval_kernel<true>
    int fold;            // the accept-reduce runs inside the update-tile launch (workgroups hold <= 3 items: n <= 4096)
    int tred;            // TAIL REDUCE (round 5): the update-reduce runs in the tail of the update-tile launch -- the workgroup whose slot
                         // completes block-row R sums R's slots (s2_hpass_kernel<.., TRED>); no s2_hreduce launch follows
    int method;          // the solver's method (QN_SR1 = 4: its runs use the BND prologues, which read it).  In the four bytes of padding in front of
                         // the pointer: nothing moves.  (SR1's extra sum and column count are compiled into instantiations of their own -- s2_hreduce_kernel
                         // <false, true>, the BND prologues: as run-time branches in the kernels of every run they cost the benchmark iteration
                         // 1-2 us -- update tiles 23.5 -> 24.4, update-reduce 4.9 -> 5.5 us, alternating A/Bs on one box -- for reasons that are not the
                         // argument's cache line, not the code's size and not cold code)
    int* cnt;            // [nb][cnt_stride] arrival counters of the block-rows (zero between launches: the last arriver resets its counter)
    int sl_first, sl_per; // ROW SLIVERS (sl_per != 0): the diagonal tiles sl_first .. nb - 1 are not on any work list; each is cut
                         // into sl_per slivers of 8 rows, one per workgroup (workgroup g: tile sl_first + g / sl_per, sliver g % sl_per,
                         // wave w its row 8 (g % sl_per) + w), taken after the workgroup's last item -- see qn_s2_eval_sliver
    double* wgS;         // [2][QN_S2_ROW][trows] (column-major) what a servicing launch hands the state machine, one row of sums per workgroup:
                         // evaluation tiles (G rows, QN_S2_NSE sums), accept-reduce (nb rows, QN_S2_NR), update-reduce (nb rows: y'u, u'g+).
                         // Launch i writes half i & 1 and launch i + 1 -- whose prologue consumes the request -- reads it: the address
                         // follows from the launch parity alone, so the rows are requested together with the control block.
                         // Column-major: every CU reads the whole table at kernel entry, all at once; as 64-byte rows that was
                         // 768 line requests per CU on the same 16 KB (a chip-wide hot spot: 2 us), as columns it is 96.
    int trows;           // rows per half: max(256, nb rounded up to 64)
    QnCtl* ctl2;         // [2]
    QnTraceRec* trace;
    double* xtrace;
    int parity;          // this launch reads ctl2[parity] and writes ctl2[parity ^ 1]
    // the two ends of a qn_minimize call, without a copy-engine transfer or a launch of their own (round 4: per-call cost):
    const QnCtl* ctl_first; // the FIRST launch of a call reads the control block from the host's pinned, device-mapped mirror (null otherwise)
    QnCtl* rep;             // the LAST launch of an enqueued batch also stores the control block it hands on into pinned host memory, ...
    unsigned long long* rep_flag; // ... then this sequence number behind it (system-scope release): the host spins on the flag instead of
    unsigned long long rep_seq;   //     synchronising the stream and copying the block back (0: this launch does not report)
    int nt;              // non-temporal tile accesses on H (past the Infinity Cache)
    int ntq;             // ... and non-temporal loads of Q's tiles
    // ---- row-sharded runs (SHARD instantiations; qn_sym2sh.hip.h) ----
    int sh_world, sh_rank; // ranks of the run, this rank
    int sh_ioff;           // first block-row this rank stores: tile (I, J) lives at local block-row I - sh_ioff
    int sh_nsum;           // slices to add up after an exchange: sh_world (all-gather, rank order) or 1 (an all-reduce left the total in slice 0)
    double* evS;           // [2][sh_world][QN_S2SH_NEC][QN_S2_MAXG]: the evaluation scalars of every rank's workgroups, by launch parity;
                           // slice r written by rank r's evaluation tiles, the others filled in by the exchange that follows the launch
    double* xg;            // [sh_world][2][np]: the ranks' partial n-vectors (q of the accepted evaluation; [u, v] of the update pass)
    const int* sl_off;     // [nb + 1] the slots of block-row R that THIS rank's tiles write are sl_idx[sl_off[R] .. sl_off[R + 1]), ascending
    const int* sl_idx;
    // ---- round 5 additions (kept behind everything the benchmark path's kernels read: the struct is the kernel argument) ----
    int cnt_stride;      // tail reduce: ints between two block-rows' arrival counters
    double* wgV;         // generic objectives: the SECOND table ([2][QN_S2_ROW][trows], an allocation of its own): the accepted-point sums a combine
                         // launch stages with every evaluation, by launch parity like wgS
    int gw;              // generic objectives (qn_sym2g.hip.h): rows of the table an evaluation's combine launch leaves (its workgroups: n / 64)
    double gmu;          // ... log-sum-exp: mu (row-sharded runs: the prologue puts the ranks' (m, S, G'd) together itself)
    double* gws;         // ... [sh_world + 1]: the ranks' weights w_r = exp(m_r - M) and S of the LAST evaluation the machine consumed (written
                         //     by that prologue, read by s2g_vec_kernel when the point is accepted)
    const double *lb, *ub;   // bounded variants (s2_dir_kernel): the solver's box (BFGSB / DFPB; null: none) ...
    const double *llb, *lub; // ... and the line search's (MoreThuenteB; null: none)
    int ring;                // round 6: the pair instance's evaluation runs as s2_evalr_kernel (qn_sym2r.hip.h: mover waves + multiplier waves)
    int projfold;            // round 6: BackTrackingB's projection inside s2_evalr_kernel<true> (no s2_proj_kernel launch per trial): the trial point is clamped where it is
                             // formed, ||P(x + t d) - x||^2 leaves the launch as column 6 of its table (rows 0 .. nb - 1), the consuming prologue adds it up
    int touch, touchq;       // round 6: TOUCH workgroups in the accept-reduce (rows of H's tiles) and in the update-reduce (rows of Q's): rows per wave, of 16
                             // (qn_s2_touch; 0: none -- the launches then are the instantiations without them; 4, 8, 12 or 16)
    int touch_delay, touchq_delay; // ... units of 64 clocks the accept-reduce's / the update-reduce's touching workgroups sleep before their first load
    int zig;                 // round 6: s2_evalr_kernel streams its two tiles in the other order in launches of odd parity (what the XCD's L2 still holds of the
                             // evaluation launch in front comes first: qn_sym2r.hip.h, ZIG-ZAG) -- the same bits
#ifdef QN_S2_STAMPS
    unsigned long long* dbg; // diagnostic build: dbg[((slot % 64) * 256 + workgroup) * 16 + k] = wall clock (10 ns) at stamp k
    int slot;
    int swz;                 // diagnostic build: workgroup g takes the items of workgroup g ^ swz (which XCD streams which tiles: QN_S2_SWZ)
#endif
};
#ifdef QN_S2_STAMPS
#define QN_S2_WG(a) ((int)blockIdx.x ^ (a).swz)
#else
#define QN_S2_WG(a) ((int)blockIdx.x)
#endif
#ifdef QN_S2_STAMPS
#define QN_S2_STAMP(k) do { if (threadIdx.x == 0 && a.dbg && blockIdx.x < 256) a.dbg[(((size_t)(a.slot & 63)) * 256 + blockIdx.x) * 16 + (k)] = wall_clock64(); } while (0)
#define QN_S2_STAMP_T(k, t) do { if (threadIdx.x == (t) && a.dbg && blockIdx.x < 256) a.dbg[(((size_t)(a.slot & 63)) * 256 + blockIdx.x) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define QN_S2_STAMP(k) do { } while (0)
#define QN_S2_STAMP_T(k, t) do { } while (0)
#endif

struct QnS2Lds {
    QnCtl c;
    double red[QN_S2_WAVES][8];
    int mine; // this launch services the pending request
};

// ---- the state machine's view of a finished request: consume its sums, then run until the next request ----
// (measured and dropped: the machine as ONE out-of-line function shared by the five kernels -- the calling convention costs the
// tile kernels ~600 bytes of scratch per lane around the call: 10.9 k instead of 11.8 k it/s)
// (Measured again and dropped, round 3: the machine as ONE out-of-line function, now called from wave 0's prologue where no tile
// window is live -- one copy of its code for the five kernels instead of five.  The call frame lives in scratch memory (400 bytes
// per lane) and the machine then takes 8.4 us instead of 4-5: 11.0 k it/s against 12.3 k inlined, same box.)
template <bool BND = false> // (BND: the bounded variants' machine -- LEAN == 2, qn_ctl_step.hip.h -- and their QN_PH_REQ_DIR)
__device__ __forceinline__ void qn_s2_advance(QnCtl& c, const double* tot, const QnVecs& V, const bool leader, double* scratch, const bool resume) {
    const int ph = c.phase;
    bool run = resume; // (resume: the machine had stopped for the x-trace copy)
    if (!resume) {
        if (ph == QN_PH_DONE) return;
        if (ph == QN_PH_IDLE) {
            c.serviced = 0;
            c.state = c.ls_only ? QN_ST_LS_ONLY : QN_ST_BEGIN;
            c.phase = QN_PH_RUNNING;
            run = true;
        } else {
            if (c.serviced != 2) return; // the request is still pending, or its tiles wait for their reduce launch
            c.serviced = 0;
            if (ph == QN_PH_REQ_EVAL) {
                const double f = 0.5 * tot[0] - tot[1]; // f = 1/2 xt'(Q xt) - b'xt
                const double gd = tot[2] - tot[3];      // g(xt)'d = d'(Q xt) - b'd
                c.st_gd0 = tot[4]; c.st_dnf = tot[5];
                // (BND: a PROJECTED trial -- BackTrackingB, s2_proj_kernel -- was evaluated at a stored point, with d = 0: it says nothing about
                // g'd, and it is not the point x + t d the memo speaks of)
                const bool proj = BND && c.req_project != 0;
                if (c.req_kind == QN_REQ_T && !c.gd0_valid && !proj) { c.gd0 = tot[4]; c.d_finite = tot[5] == 0.0; c.gd0_valid = 1; }
                c.f_e = f; c.gd_e = gd;
                c.n_oracle_evals++;
                c.ev_par ^= 1; c.ev_kind = c.req_kind; c.ev_t = c.req_t;
                if (c.req_kind == QN_REQ_T && !proj) { c.last_valid = 1; c.last_t = c.req_t; c.f_last = f; c.gd_last = gd; }
                else c.last_valid = 0;
                if (BND) c.last_projected = proj ? 1 : 0;
                if (c.req_need_vectors) { c.phase = QN_PH_REQ_VEC; return; } // the caller wants g+, y, s of this point too
                c.state = c.after_state;
            } else if (ph == QN_PH_REQ_VEC) {
                c.st_yy = tot[0]; c.st_ys = tot[1]; c.st_gg = tot[2]; c.st_ss = tot[3]; c.hp_sg = tot[4];
                c.state = c.after_state;
            } else if (ph == QN_PH_REQ_HPASS) {
                if (c.hp_nrhs == 2) { c.hp_yu = tot[0]; c.hp_ug = tot[1]; if (BND) c.hp_den = tot[2]; } // (hp_den: SR1's denominator, third column of its update-reduce)
                c.state = c.after_state;
            } else if (BND && ph == QN_PH_REQ_DIR) { // VV holds -d (stored, projected): the evaluations form x + t d as they do for a first direction
                c.mtb_cand = tot[0];                // (the MINIMUM over the block-rows: the prologue folds this request's column that way)
                if (c.s2_dir & 2) c.mt_tmax = fmin(c.mt_tmax, c.mtb_cand); // morethuente_b.rs:201: self.t_max = self.t_max.min(candidate) -- persists
                c.dir_mode = 0; c.dir_ready = 1;
                c.state = c.after_state;
            } else {
                return;
            }
            c.phase = QN_PH_RUNNING;
            run = true;
        }
    }
    if (run) ctl_scalar_run<BND ? 2 : 1>(c, V, scratch, leader); // (ONE call site: the machine is inlined, and it is large; LEAN: qn_ctl_step.hip.h)
    // Folded accept-reduce: the launch that formed g+, y, s of the accepted point ran the update tiles as well, BEFORE the machine
    // had seen ||s||, ||y|| (the tiles only need the vectors and the coefficients of the PENDING update, all known then).  Now that
    // the machine has consumed those sums: if it asks for exactly that pass, its tiles are done (the reduce launch is what is left);
    // if it does not (step or gradient change below tol: bfgs.rs:106-112 -- the run then ends at the next loop top), the tiles
    // have still applied the pending update to the stored matrix, which is all `pending` stands for.
    // (spec_tiles = the number of right-hand sides the tiles ran with: 2 = [y, g+] after an accepted step, 1 = g after the evaluation
    // at x that opens a run.)
    if (c.spec_tiles && c.phase != QN_PH_RUNNING && c.phase != QN_PH_REQ_VEC) {
        const int ran = c.spec_tiles;
        c.spec_tiles = 0;
        if (c.phase == QN_PH_REQ_HPASS && c.serviced == 0 && c.hp_nrhs == ran) c.serviced = 1;
        else if (c.pending) { c.pending = 0; c.n_hpasses++; c.n_hpass_rw++; }
    }
}

// a wave-uniform double out of lane `l` (uniform) of a per-lane value: the row-side inputs of a tile (x_i, d_i, s_i, u_i, y_i,
// g_i for the wave's 16 rows) live one row per lane and are broadcast into SGPRs as the row loop needs them -- no LDS staging,
// no barrier, and no vector registers held across the loop for them
__device__ __forceinline__ double qn_lane_bcast(const double v, const int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// Prologue of every sym2 kernel: run by WAVE 0 ALONE, before that wave requests any tile data.
//
// Round 2 had all eight waves request their share of the first tile and only then look at the control block.  In-kernel time
// stamps (tools/s2_stamps.py, n = 4096) showed what that costs: a wave's loads return in order and the first window of the whole
// chip is a 33 MB burst, so the control block -- requested first -- was in LDS 5.4 us after kernel entry, the column sums 0.6 us
// and the state machine on an LDS-resident control block 3.2 us later: the row loop started 9.4 us into a 19.5 us kernel, the
// first window had landed 4 us earlier and nothing was in flight meanwhile.  A plain streaming kernel of the same shape with a
// one-load prologue runs the phase in 13.5 us launch to launch (tools/seam_probe.hip) against 23.2 us.  So now:
//   * waves 1..7 request their rows at once and wait at the workgroup barrier; wave 0 -- with nothing queued in front -- gets the
//     control block and the previous launch's sums in one memory round trip of an idle queue, folds the sums, runs the machine and
//     only then requests its own 16 rows (an eighth of the window, behind a burst that is over by then);
//   * the sums of the previous launch sit in ONE table addressed by the launch parity (QnS2Args.wgS), so their address does not
//     depend on the control block and there is one table to request instead of four;
//   (Measured again and dropped: the machine on a private register copy of the control block.  Wave 0 holds no window while it
//   runs, yet ~150 live words plus the machine's temporaries still do not fit 256 registers: 255 spills in the tile kernels.)
//   The sums keep round 2's order (the lane's rows in row order, then the xor butterfly): same bits.
#define QN_S2_PCH (QN_S2_MAXG / 64) // row chunks of 64 the wave loads from the table
// (Measured and dropped, same-box A/B: waves 1..7 sleeping 1-2 us before their 112 KB burst so that wave 0's control block and
// table come back from an idle memory system -- 12.2 k it/s without, 11.3-11.9 k with: the burst itself is on the critical path.)

// Measured and dropped, round 5 (tools/experiments/r05_machine_on_lane_registers.patch rebuilds it): THE MACHINE ON REGISTERS.  The
// block is in registers when it arrives, spread over the wave -- word W in lane W & 63 of (cw0, cw1); a view of exactly that
// (QnCtlLanes, generated from qn_ctl.h: a field read = two v_readlane, a write = two v_writelane) and the state machine templated
// on the control type, run on all 64 lanes alike, the block back to LDS once at the end.  The same bits in every run
// (tools/trace_cmp.py), no scratch, 50 vector registers fewer -- and SLOWER: the accept-reduce 6.4 -> 7.6 us, the update-reduce
// 5.1 -> 5.45, 66.3 -> 68.0 us per iteration (profiles/r05_j_*), with 10 KB more code per kernel.  tools/icache_probe.hip says what
// the machine's time is: one wave issues an instruction every ~5 cycles, cold or warm, right behind itself or behind another kernel --
// neither instruction fetch nor LDS latency: ~700 instructions on the machine's path are its 2 us, and two readlanes and a move per
// field are more instructions than one ds_read.  Fewer instructions on the path (the LEAN instantiation) is what makes it faster.
// (The experiment lives in a patch, not behind a switch: the launches of 32 workgroups are sensitive to where their code lies -- round 5's
// new kernels grew the code object from 2.25 to 2.54 MB and the accept-reduce went 6.45 -> 7.1 us with an IDENTICAL instruction stream,
// profiles/r05_o_* -- so nothing that is not used stays in the tree.)
struct QnS2NoEarly { __device__ __forceinline__ void operator()() const {} };
// `early`: requests the caller wants in flight while the machine runs (issued right behind the control block and the table, so
// that those two still come back first: a wave's loads return in order)
// SHARD (row-sharded runs, qn_sym2sh.hip.h): the machine is the same and runs on every rank from the same inputs.  What differs
// is where an EVALUATION's sums come from -- every rank's workgroups left theirs in its slice of evS and the exchange behind the
// launch has brought in the other ranks' slices: this prologue adds them all up, ranks in order, so every rank has the same
// bits -- and that a request for vectors takes two launches with an exchange of n-vectors in between (serviced: 0 pending,
// 1 tiles / partial sums done, 3 partial sums of an update pass done, 2 complete).
// GOBJ (generic objectives, qn_sym2g.hip.h): an evaluation's sums come from its combine launch -- a.gw rows -- which has ALSO staged the
// vectors of the evaluated point (g+, y, x+, s) and their five sums (table columns QN_S2_VCOL ..): when the machine accepts the point
// and asks for them (QN_PH_REQ_VEC), this prologue hands them over at once and lets the machine go on -- no launch in between.
template <int KIND, bool SHARD = false, class Early = QnS2NoEarly, bool GOBJ = false, bool BND = false>
__device__ __forceinline__ void qn_s2_prologue_w0(const QnS2Args& a, QnS2Lds& L, Early&& early = Early()) {
    static_assert(!(BND && (SHARD || GOBJ)), "the bounded variants run on one rank, on the quadratic");
    const int lane = threadIdx.x; // (wave 0)
    const bool leader = blockIdx.x == 0;
    constexpr int NW = (int)(sizeof(QnCtl) / 8);
    static_assert(NW <= 128, "control block too large for two words per lane");
    __builtin_amdgcn_s_setprio(3); // the workgroup waits for this wave: its instructions go first on the SIMD it shares
    const uint64_t* cin = reinterpret_cast<const uint64_t*>(a.ctl_first ? a.ctl_first : a.ctl2 + a.parity);
    uint64_t* lc = reinterpret_cast<uint64_t*>(&L.c);
    uint64_t cw0 = 0, cw1 = 0;
    if (lane < NW) cw0 = cin[lane];
    if (64 + lane < NW) cw1 = cin[64 + lane];
    double tr[QN_S2_PCH][QN_S2_NSE];
    const double* T = a.wgS + (size_t)(a.parity ^ 1) * (size_t)a.trows * QN_S2_ROW;
    // launches with nothing to decide in front of them: the second (and third) launch of a request pass the control block on
    constexpr bool kPass = KIND == QN_S2_HSUM || KIND == QN_S2_GCOMB || (SHARD && (KIND == QN_S2_VEC || KIND == QN_S2_HREDUCE));
    const bool no_decision = kPass || (KIND == QN_S2_HREDUCE && !a.fold); // (uniform) nothing to decide between the update tiles and their reduction
    if (!no_decision) {
#pragma unroll
        for (int k = 0; k < QN_S2_NSE; ++k)
#pragma unroll
            for (int j = 0; j < QN_S2_PCH; ++j) tr[j][k] = T[(size_t)k * a.trows + j * 64 + lane]; // (trows >= 256)
    }
    double t6[BND ? QN_S2_PCH : 1]; // (BackTrackingB projected inside the evaluation kernel: the shares of ||P(x + t d) - x||^2, column 6)
    if constexpr (BND) {
#pragma unroll
        for (int j = 0; j < QN_S2_PCH; ++j) t6[j] = (a.projfold && !no_decision) ? T[(size_t)6 * a.trows + j * 64 + lane] : 0.0;
    }
    // (Measured and dropped, round 5: the control block's and the table's addresses as LEADING SCALAR kernel arguments, preloaded into SGPRs by the
    // dispatcher -- -mllvm -amdgpu-kernarg-preload-count=16, .amdhsa_user_sgpr_kernarg_preload_length 4 on the accept-reduce -- so that the first
    // requests need no kernel-argument fetch in front of them: 5.93 -> 6.03 us, nothing.  HIP_FORCE_DEV_KERNARG=0, for scale: +2 us on EVERY kernel.)
    // (Measured and dropped, round 5: the first 64 rows of every column requested FIRST and only those waited for when the table has
    // at most 64 rows -- the two small kernels' tables at n = 4096; in-kernel stamps had shown the table 1.3-1.5 us behind the control
    // block in the update kernel.  rocprofv3 averages, alternating runs on one box (profiles/r05_f_*): evaluation -0.1 us, update tiles
    // -0.05 us, and the update-reduce -- which does not run this code -- +0.4 us: no gain.)
    double tv[GOBJ ? QN_S2_PCH : 1][QN_S2_NR]; // GOBJ: the staged accepted-point sums of the last evaluation
    if constexpr (GOBJ) if (!no_decision) {
#pragma unroll
        for (int k = 0; k < QN_S2_NR; ++k)
#pragma unroll
            for (int j = 0; j < QN_S2_PCH; ++j) tv[j][k] = a.wgV[((size_t)(a.parity ^ 1) * QN_S2_ROW + k) * a.trows + j * 64 + lane]; // (the second table's same half)
    }
    // SHARD: the evaluation scalars of the first QN_S2SH_EB ranks go out now as well (every entry of evS is valid at all times --
    // rows past the grid stay zero -- so nothing about them depends on the control block; lane l takes rows 2 l, 2 l + 1 and
    // 128 + 2 l, 129 + 2 l of each column)
    v2d es[SHARD ? QN_S2SH_EB : 1][QN_S2SH_NEC][2];
    const double* E = SHARD ? a.evS + (size_t)(a.parity ^ 1) * (size_t)a.sh_world * (QN_S2SH_NEC * QN_S2_MAXG) : nullptr;
    auto es_load = [&](const int r0) {
#pragma unroll
        for (int rr = 0; rr < QN_S2SH_EB; ++rr)
#pragma unroll
            for (int cc = 0; cc < QN_S2SH_NEC; ++cc)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
                    es[rr][cc][hh] = (r0 + rr < a.sh_nsum) ? ld2(E + ((size_t)(r0 + rr) * QN_S2SH_NEC + cc) * QN_S2_MAXG + hh * 128 + 2 * lane) : (v2d){0.0, 0.0};
    };
    if (SHARD && !GOBJ && !no_decision) es_load(0);
    early();
    if (lane < NW) lc[lane] = cw0;
    if (64 + lane < NW) lc[64 + lane] = cw1;
    __builtin_amdgcn_wave_barrier(); // (one wave: its LDS accesses execute in program order; this only pins the compiler's order)
    QN_S2_STAMP(9);
    QnCtl& c = L.c; // (in LDS: a private register copy of all ~150 words does not fit beside the machine's own temporaries -- 255 spills)
    int mine = 0;
    if (no_decision) { // pass the control block on (with the folded accept-reduce this prologue is where the machine sees the accepted point)
        constexpr int want_ph = (KIND == QN_S2_VEC) ? QN_PH_REQ_VEC : (KIND == QN_S2_GCOMB ? QN_PH_REQ_EVAL : QN_PH_REQ_HPASS);
        constexpr int from = (SHARD && KIND == QN_S2_HREDUCE) ? 3 : 1; // (sharded update pass: tiles 0 -> 1, partial sums 1 -> 3, reduce 3 -> 2)
        constexpr int to = (KIND == QN_S2_HSUM) ? 3 : 2;
        mine = c.phase == want_ph && c.serviced == from;
        if (lane == 0) { if (mine) c.serviced = to; L.mine = mine; }
        __builtin_amdgcn_s_setprio(0);
        return;
    }
    const int ph = c.phase;
    if (ph == QN_PH_DONE) { // launches enqueued past the end of the run: pass the control block on, nothing else
        if (lane == 0) L.mine = 0;
        __builtin_amdgcn_s_setprio(0);
        return;
    }
    double tot[QN_S2_NSE];
    double totv[GOBJ ? QN_S2_NSE : 1];
#pragma unroll
    for (int k = 0; k < QN_S2_NSE; ++k) tot[k] = 0.0;
#pragma unroll
    for (int k = 0; k < (GOBJ ? QN_S2_NSE : 1); ++k) totv[k] = 0.0;
    if (c.serviced == 2 && (ph == QN_PH_REQ_EVAL || ph == QN_PH_REQ_VEC || ph == QN_PH_REQ_HPASS)) { // (uniform)
        const int nrows = ph == QN_PH_REQ_EVAL ? (GOBJ ? a.gw : a.G) : a.nb;
        const int ncol = ph == QN_PH_REQ_EVAL ? QN_S2_NSE : (ph == QN_PH_REQ_VEC ? QN_S2_NR : ((BND && a.method == 4) ? 3 : 2)); // (SR1 -- its runs use the BND instantiations --: y'u, u'g+ and (s - u)'y)
        // The lane's rows in row order, then ONE halving butterfly over all columns at once (QnWaveFold: 17 exchanges instead of
        // six 6-step butterflies).  It pairs lanes l and l ^ 32, then ^ 16, ... ^ 1 for every column exactly as qn_wave_sum
        // does, and floating-point addition commutes: the totals have round 2's bits.
        double acc[8];
        if (SHARD && !GOBJ && ph == QN_PH_REQ_EVAL) { // (uniform)
            // Entry by entry the ranks in rank order FIRST (what an all-reduce would have left in slice 0: with sh_nsum == 1 this
            // loop is empty, and the host-staged stand-in of the tests then gives the very bits of the all-gather), then the lane's
            // four rows of a column in row order, then the butterfly over the lanes.
            v2d rs[QN_S2SH_NEC][2];
#pragma unroll
            for (int cc = 0; cc < QN_S2SH_NEC; ++cc) { rs[cc][0] = es[0][cc][0]; rs[cc][1] = es[0][cc][1]; }
            for (int r0 = 0; r0 < a.sh_nsum; r0 += QN_S2SH_EB) {
                if (r0) es_load(r0);
#pragma unroll
                for (int rr = 0; rr < QN_S2SH_EB; ++rr)
                    if (r0 + rr > 0 && r0 + rr < a.sh_nsum) { // (uniform)
#pragma unroll
                        for (int cc = 0; cc < QN_S2SH_NEC; ++cc) { rs[cc][0] = rs[cc][0] + es[rr][cc][0]; rs[cc][1] = rs[cc][1] + es[rr][cc][1]; }
                    }
            }
            double ea[QN_S2SH_NEC];
#pragma unroll
            for (int cc = 0; cc < QN_S2SH_NEC; ++cc) ea[cc] = ((rs[cc][0].x + rs[cc][0].y) + rs[cc][1].x) + rs[cc][1].y;
            // the table's columns: x'(Q xt - 2b), (b'xt = 0), d'(Q xt - b), (b'd = 0), g'd, #non-finite d
            acc[0] = ea[0]; acc[1] = 0.0; acc[2] = ea[1]; acc[3] = 0.0; acc[4] = ea[2]; acc[5] = ea[3]; acc[6] = 0.0; acc[7] = 0.0;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = 0.0;
                if (k < QN_S2_NSE) {
#pragma unroll
                    for (int j = 0; j < QN_S2_PCH; ++j) acc[k] = acc[k] + ((j * 64 + lane < nrows && k < ncol) ? tr[j][k] : 0.0);
                    for (int b = QN_S2_MAXG + lane; b < nrows; b += 64) // (block-rows past 256: n > 32768)
                        if (k < ncol) acc[k] = acc[k] + T[(size_t)k * a.trows + b];
                }
            }
        }
        QN_S2_STAMP(1); // (the table's values have arrived)
        QnWaveFold<8, 32>::run(acc, lane); // lane l holds the total of column l >> 3
#pragma unroll
        for (int k = 0; k < QN_S2_NSE; ++k) tot[k] = qn_lane_bcast(acc[0], 8 * k);
        if constexpr (GOBJ && SHARD) if (ph == QN_PH_REQ_EVAL) { // (uniform)
            // Row-sharded log-sum-exp (qn_sym2g.hip.h): rank r's combine launch left in its slice of evS -- exchanged behind it -- the
            // partial sums of G_r'd per workgroup (column 0) and its running maximum m_r and S_r = sum exp(z - m_r) (column 1, rows 0
            // and 1); the table's own columns hold what every rank computes alike from the replicated vectors: mu xt'xt, xt'd, g'd and
            // the non-finite count.  With M = max m_r, w_r = exp(m_r - M), S = sum S_r w_r (ranks in rank order, as lse_finish1_kernel):
            //     f = M + log S + mu/2 xt'xt        g(xt)'d = (sum_r w_r G_r'd) / S + mu xt'd
            const double* E = a.evS + (size_t)(a.parity ^ 1) * (size_t)a.sh_world * (QN_S2SH_NEC * QN_S2_MAXG);
            double M = -INFINITY;
            for (int r = 0; r < a.sh_world; ++r) M = fmax(M, E[((size_t)r * QN_S2SH_NEC + 1) * QN_S2_MAXG]);
            double S = 0.0, gdA = 0.0;
            for (int r = 0; r < a.sh_world; ++r) {
                const double* Er = E + (size_t)r * QN_S2SH_NEC * QN_S2_MAXG;
                const double mr = Er[QN_S2_MAXG], sr = Er[QN_S2_MAXG + 1];
                const double w = mr != -INFINITY ? exp(mr - M) : 0.0; // (a rank that owns no real row contributes nothing)
                S = __builtin_fma(sr, w, S);
                double dp = 0.0;
#pragma unroll
                for (int j = 0; j < QN_S2_PCH; ++j) dp = dp + Er[j * 64 + lane]; // (rows past the combine launch's grid stay zero)
                dp = qn_wave_sum(dp);
                gdA = __builtin_fma(dp, w, gdA);
                if (leader && lane == 0) a.gws[r] = w;
            }
            if (leader && lane == 0) a.gws[a.sh_world] = S;
            tot[0] = 2.0 * (M + log(S)) + tot[0];
            tot[2] = gdA / S + a.gmu * tot[1];
            tot[1] = 0.0; tot[3] = 0.0;
        }
        if constexpr (GOBJ) if (ph == QN_PH_REQ_EVAL) { // (uniform) ... and the staged accepted-point sums, the same way
            double av[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                av[k] = 0.0;
                if (k < QN_S2_NR) {
#pragma unroll
                    for (int j = 0; j < QN_S2_PCH; ++j) av[k] = av[k] + ((j * 64 + lane < nrows) ? tv[j][k] : 0.0);
                }
            }
            QnWaveFold<8, 32>::run(av, lane);
#pragma unroll
            for (int k = 0; k < QN_S2_NR; ++k) totv[k] = qn_lane_bcast(av[0], 8 * k);
        }
    }
    if constexpr (BND) if (c.serviced == 2 && ph == QN_PH_REQ_DIR) { // (uniform) column 0 of s2_dir_kernel's rows: the MINIMUM (morethuente_b.rs:198)
        double m = INFINITY;
#pragma unroll
        for (int j = 0; j < QN_S2_PCH; ++j) m = fmin(m, (j * 64 + lane < a.nb) ? tr[j][0] : INFINITY);
        for (int b = QN_S2_MAXG + lane; b < a.nb; b += 64) m = fmin(m, T[b]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmin(m, __shfl_xor(m, off, 64));
        tot[0] = m;
    }
    if constexpr (BND) if (a.projfold && c.serviced == 2 && ph == QN_PH_REQ_EVAL && c.req_project) { // (uniform) the evaluation kernel projected itself: column 6
        double d2 = 0.0; // of ITS table, rows 0 .. nb - 1 -- the same shares in the same order as the s2_proj_kernel flow below: the same bits
#pragma unroll
        for (int j = 0; j < QN_S2_PCH; ++j) d2 = d2 + ((j * 64 + lane < a.nb) ? t6[j] : 0.0);
        for (int b = QN_S2_MAXG + lane; b < a.nb; b += 64) d2 = d2 + T[(size_t)6 * a.trows + b];
        d2 = qn_wave_sum(d2);
        if (lane == 0) c.bt_diff2 = d2; // (consumed by the machine below: backtracking_b.rs:24-34)
    }
    if constexpr (BND) if (c.serviced == 5 && ph == QN_PH_REQ_EVAL) { // (uniform) column 0 of s2_proj_kernel's rows: ||P(x + t d) - x||^2, block-rows in order
        double d2 = 0.0;
#pragma unroll
        for (int j = 0; j < QN_S2_PCH; ++j) d2 = d2 + ((j * 64 + lane < a.nb) ? tr[j][0] : 0.0);
        for (int b = QN_S2_MAXG + lane; b < a.nb; b += 64) d2 = d2 + T[b];
        d2 = qn_wave_sum(d2);
        if (lane == 0) c.bt_diff2 = d2; // (backtracking_b.rs:33-34; the evaluation this launch may run is AT the stored projected point)
    }
    QN_S2_STAMP(10);
    QnVecs V{};
    V.n = a.n; V.n_pad = a.np; V.trace = a.trace;
    // (two copies of the loop on purpose: the generic-objective one -- a second entry into the machine with the staged sums -- runs in
    // one-workgroup launches only; written as one loop with run-time flags it changed the code of EVERY kernel's prologue, and the two
    // small kernels of the benchmark iteration, whose critical path is the machine, measured 0.3-0.65 us slower: profiles/r05_f_*)
    if constexpr (!GOBJ) {
        for (int guard = 0; guard < 64; ++guard) { // (the n <= 5 reference-order code of the machine is never reached on this path: no scratch)
            int need_x = 0;
            if (lane == 0) {
                qn_s2_advance<BND>(c, tot, V, leader, &L.red[0][0], guard > 0);
                need_x = c.phase == QN_PH_RUNNING && c.state == QN_ST_ITER_END;
            }
            need_x = __builtin_amdgcn_readfirstlane(need_x);
            if (!need_x) break;
            // the machine stopped because the iterate has to be recorded (trace with x): workgroup 0 copies it, all go on
            __builtin_amdgcn_wave_barrier();
            if (leader) {
                double* row = a.xtrace + (size_t)L.c.k * (size_t)a.n;
                const double* xs = a.F.X0 + (size_t)L.c.xc * (size_t)a.np;
                for (int i = lane; i < a.n; i += 64) row[i] = xs[i];
            }
            if (lane == 0) c.xtrace_done = 1;
        }
    } else {
        bool resume = false, use_v = false; // (uniform)
        for (int guard = 0; guard < 64; ++guard) {
            int need_x = 0;
            if (lane == 0) {
                qn_s2_advance(c, use_v ? totv : tot, V, leader, &L.red[0][0], resume);
                need_x = c.phase == QN_PH_RUNNING && c.state == QN_ST_ITER_END;
                if (!SHARD && !need_x && c.phase == QN_PH_REQ_VEC && c.serviced == 0) { c.serviced = 2; need_x = 2; } // the vectors are staged: go on
                // (row-sharded: the gradient of the accepted point has to be gathered first -- s2g_vec_kernel, behind an exchange)
            }
            need_x = __builtin_amdgcn_readfirstlane(need_x);
            if (!need_x) break;
            if (need_x == 2) { use_v = true; resume = false; continue; }
            resume = true;
            __builtin_amdgcn_wave_barrier();
            if (leader) {
                double* row = a.xtrace + (size_t)L.c.k * (size_t)a.n;
                const double* xs = a.F.X0 + (size_t)L.c.xc * (size_t)a.np;
                for (int i = lane; i < a.n; i += 64) row[i] = xs[i];
            }
            if (lane == 0) c.xtrace_done = 1;
        }
    }
    QN_S2_STAMP(11);
    if (lane == 0) {
        if (KIND == QN_S2_EVAL) mine = c.phase == QN_PH_REQ_EVAL && c.serviced == ((BND && c.req_project && !a.projfold) ? 5 : 0); // (a projected trial: behind its s2_proj_kernel, unless the evaluation kernel projects itself)
        if (KIND == QN_S2_PROJ) mine = c.phase == QN_PH_REQ_EVAL && c.req_project && c.serviced == 0;
        if (KIND == QN_S2_VEC || KIND == QN_S2_VSUM || KIND == QN_S2_VECD) mine = c.phase == QN_PH_REQ_VEC && c.serviced == 0;
        if (KIND == QN_S2_HTILE) { // 2: the accepted point's slots -> vectors AND the update tiles (folded accept-reduce); 1: the tiles of a pending pass
            if (a.fold && c.phase == QN_PH_REQ_VEC && c.serviced == 0) mine = 2;
            else if (c.phase == QN_PH_REQ_HPASS && c.serviced == 0) mine = 1;
        }
        if (KIND == QN_S2_HREDUCE) mine = c.phase == QN_PH_REQ_HPASS && c.serviced == 1;
        if (KIND == QN_S2_DIR) mine = c.phase == QN_PH_REQ_DIR && c.serviced == 0;
        if (KIND == QN_S2_GEVAL_A) mine = c.phase == QN_PH_REQ_EVAL && c.serviced == 0;  // (generic objectives: the kernels that follow
        if (KIND == QN_S2_GHT_A) mine = c.phase == QN_PH_REQ_HPASS && c.serviced == 0;   //  read `serviced == 1` as "yours")
        if (c.phase == QN_PH_RUNNING) { c.status = 4; c.phase = QN_PH_DONE; } // a state this path cannot service: abort, never spin
        if (mine) { // as this launch leaves the request
            if (KIND == QN_S2_HTILE) { c.serviced = (mine == 2 || a.tred) ? 2 : 1; if (mine == 2) c.spec_tiles = c.ev_kind == QN_REQ_T ? 2 : 1; } // (tail reduce: the launch leaves the pass complete)
            else if (KIND == QN_S2_VSUM || KIND == QN_S2_GEVAL_A || KIND == QN_S2_GHT_A) c.serviced = 1; // (the exchange / the kernels that do the work follow)
            else if (KIND == QN_S2_PROJ) c.serviced = 5; // (the trial point is stored: its evaluation launch follows)
            else c.serviced = 2;
        }
        L.mine = mine;
    }
    __builtin_amdgcn_s_setprio(0);
}

// after the workgroup barrier that follows the prologue: workgroup 0 hands the control block to the next launch
// (called by every thread of the workgroup, from uniform control flow: the reporting branch holds a barrier)
__device__ __forceinline__ void qn_s2_ctl_out(const QnS2Args& a, const QnS2Lds& L) {
    constexpr int NW = (int)(sizeof(QnCtl) / 8);
    if (blockIdx.x == 0 && threadIdx.x < NW)
        reinterpret_cast<uint64_t*>(a.ctl2 + (a.parity ^ 1))[threadIdx.x] = reinterpret_cast<const uint64_t*>(&L.c)[threadIdx.x];
    if (a.rep_seq != 0 && blockIdx.x == 0) { // (uniform) the batch's last launch: the host is waiting for this block
        if (threadIdx.x < NW) {
            reinterpret_cast<uint64_t*>(a.rep)[threadIdx.x] = reinterpret_cast<const uint64_t*>(&L.c)[threadIdx.x];
            __threadfence_system();
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.rep_flag, a.rep_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the evaluation request as the tile and the accept-reduce kernels decode it
// (the request's scalars are the same in every lane: pinned into scalar registers -- as vector registers the row loops' register
// pressure pushed them out to scratch memory, three 16-byte reloads in front of every item)
__device__ __forceinline__ double qn_uniform(const double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <bool PIN, bool BNDREQ = false> // (BNDREQ: the bounded runs' evaluation kernels -- a request may name a STORED point; in_eval: the kernel projects itself, QnS2Args.projfold)
__device__ __forceinline__ QnEvalReq qn_s2_eval_req(const QnCtl& c, bool last_eval, const bool in_eval = false) {
    QnEvalReq q;
    q.is_t = (last_eval ? c.ev_kind : c.req_kind) == QN_REQ_T;
    q.t = last_eval ? c.ev_t : c.req_t;
    const bool stored = BNDREQ && !last_eval && c.req_project != 0 && !in_eval; // (BackTrackingB: evaluate AT the point s2_proj_kernel stored in the trial half of X0)
    if (stored) q.is_t = false;
    q.mode = c.dir_mode;
    q.c_ss = c.c_ss; q.c_su = c.c_su; q.c_uu = c.c_uu; q.ug = c.dir_ug; q.sg = c.dir_sg;
    // The lazy direction -H+ g+ = -(v + c_su (s (u.g) + u (s.g)) + c_ss s (s.g) + c_uu u (u.g)) with the two vector coefficients
    // formed ONCE per wave: every lane forms d at three indices per item, and at thirteen operations each the trial points were a
    // fifth of the instructions behind the workgroup barrier.  (Same formula in the evaluation tiles and in the accept-reduce:
    // the x+ that is stored is the point that was evaluated.)
    q.al = q.c_su * q.ug + q.c_ss * q.sg;
    q.be = q.c_su * q.sg + q.c_uu * q.ug;
    q.xc = stored ? 1 - c.xc : c.xc; q.sc = c.sc;
    if (PIN) { q.al = qn_uniform(q.al); q.be = qn_uniform(q.be); }
    if (PIN) { // (the tile kernel; the accept-reduce has registers to spare and only pays the readfirstlane latency)
        q.is_t = __builtin_amdgcn_readfirstlane(q.is_t) != 0;
        q.t = qn_uniform(q.t);
        q.mode = __builtin_amdgcn_readfirstlane(q.mode);
        q.c_ss = qn_uniform(q.c_ss); q.c_su = qn_uniform(q.c_su); q.c_uu = qn_uniform(q.c_uu); q.ug = qn_uniform(q.ug); q.sg = qn_uniform(q.sg);
        q.xc = __builtin_amdgcn_readfirstlane(q.xc); q.sc = __builtin_amdgcn_readfirstlane(q.sc);
    }
    return q;
}

// sum of `cnt` values held one per wave in red[w][k] by thread 0 (fixed order)
__device__ __forceinline__ double qn_s2_wave_total(const double (*red)[8], int k) {
    double t = red[0][k];
#pragma unroll
    for (int w = 1; w < QN_S2_WAVES; ++w) t = t + red[w][k];
    return t;
}

// The first items of a workgroup need no load: the host deals the first min(2 G, items) items out in order -- item t goes to
// workgroup t mod G as its (t / G)-th; the off-diagonal tiles row-major over (I, J > I), then the diagonal ones -- so they follow
// from the workgroup's index alone (a dependent load at kernel entry is a memory round trip in front of everything).
// (An integer loop on purpose: the argument is uniform, so this is a dozen SCALAR instructions in one cache line of code.  Round 3
// started with the closed form through a double-precision square root: ~150 instructions per call, twice, in front of the first
// load of every wave -- and a kernel starts with a cold instruction cache behind an L2 that the tile stream has just swept:
// in-kernel stamps showed the last wave of a workgroup issuing its sixteen row loads 6 us after kernel entry.)
__device__ __forceinline__ int qn_s2_item_of_index(int t, int nb) {
    const int noff = nb * (nb - 1) / 2;
    if (t >= noff) return ((t - noff) << 16) | (t - noff);
    int i = 0, len = nb - 1; // block-row i has len tiles right of the diagonal
    while (t >= len) { t -= len; --len; ++i; }
    return (i << 16) | (i + 1 + t);
}
__device__ __forceinline__ int qn_s2_first_item(int g, int nb) { return qn_s2_item_of_index(g, nb); }
// Row-sharded runs (SHARD): the rank's tiles are its block-rows' circulant windows (qn_sym.hip.h: J = (I + k) mod nb, so J < I
// occurs), dealt by the host like any other list -- the first item is read, not computed: one scalar load in front of a launch
// that streams ~16 tiles per workgroup -- and tile (I, J) lives at the rank's local block-row I - sh_ioff.
template <bool SHARD>
__device__ __forceinline__ int qn_s2_first_item_of(const QnS2Args& a) { return SHARD ? a.item_ij[blockIdx.x] : qn_s2_first_item(QN_S2_WG(a), a.nb); }
template <bool SHARD>
__device__ __forceinline__ int qn_s2_lrow(const QnS2Args& a, const int I) { return SHARD ? I - a.sh_ioff : I; }
// the workgroup's second item (-1: none): in closed form when the host dealt it in order (a.inorder), from the list otherwise
__device__ __forceinline__ int qn_s2_second_item(const QnS2Args& a) {
    const int t = a.G + QN_S2_WG(a);
    if (t < a.inorder) return qn_s2_item_of_index(t, a.nb);
    return (a.maxk > 1) ? a.item_ij[(size_t)a.G + blockIdx.x] : -1;
}

// Diagonal items without a single per-element test.  Wave w owns rows 16 w .. 16 w + 15 of the tile; lane l owns columns 2 l, 2 l + 1.
//   * lanes left of the wave's 16 x 16 diagonal sub-block (l < 8 w) have nothing to do: they are CLONES of lane 8 w -- same
//     load address (one request, no extra traffic), same column-side update vectors, hence the same stored values -- with
//     their multiplier entries and column sums forced to zero;
//   * the lanes of the diagonal sub-block (8 w <= l < 8 w + 8) use it WHOLE for the row part -- it is kept symmetric in memory,
//     both triangles, by the symmetric update formula -- and contribute nothing to the column part (that would count it twice);
//   * lanes right of it (l >= 8 w + 8) work as in an off-diagonal tile.
// So a diagonal item differs from an off-diagonal one by three per-lane constants; the row loop is the same straight-line code.
// (The sub-blocks below the diagonal ones are not maintained: sym2_mirror_kernel restores them with the lower block triangle.)
//
// Measured and dropped (round 2, same-box A/B with tools/ab.sh, n = 4096): a second, look-ahead window in LDS -- the item after the
// one in registers fetched by direct global-to-LDS loads (global_load_lds_dwordx4, 128 KB of dynamic LDS), requested while thread 0
// runs the state machine, copied into the register window at the item boundary, no loads left in the row loops.  In-kernel time
// stamps improved as predicted (row loops 2.0 -> 0.8 us and 0.8 -> 0.44 us, median workgroup 19.2 -> 17.5 us) but the launch as the
// stream sees it got 3 us LONGER (event bracket 24.5 -> 27.6 us), and the iteration 88.3 -> 91.5 us.  With the compiler's builtin
// instead of inline assembly every LDS access after a request waits for all outstanding loads (the state machine then sits behind
// the whole tile window: +3 us in the prologue).  Also measured: the work list read one item ahead of its use (no change).
// Why the look-ahead cannot pay at this size: the SLOWEST workgroup ends when the last byte of the 67 MB has arrived, and that is set
// by the device's read rate (~4.9 TB/s here: 13.7 us) plus dispatch, first issue and one tail -- requesting everything earlier makes
// the median workgroup finish sooner, not the last one.  (An unused 128 KB dynamic-LDS allocation by itself costs nothing.)
__device__ __forceinline__ int qn_s2_col(bool diag, int lane, int wave) { return 2 * (diag ? max(lane, 8 * wave) : lane); }
__device__ __forceinline__ bool qn_s2_row_on(bool diag, int lane, int wave) { return !diag || lane >= 8 * wave; }     // multiplier entries kept
__device__ __forceinline__ bool qn_s2_col_on(bool diag, int lane, int wave) { return !diag || lane >= 8 * wave + 8; } // column sums kept

// ------------------------------------------------------------------------------------------------
// evaluation tiles: q-slots of Q (x + t d) and the scalars xt'Q xt, d'Q xt (+ on diagonal items the vector dots of block I)
// ------------------------------------------------------------------------------------------------
// x + t d at one index from loaded values: the arithmetic of qn_trial_entry (qn_sym.hip.h), which the accept-reduce uses
__device__ __forceinline__ double qn_s2_trial(const QnEvalReq& q, const double xi, const double vi, const double si, const double ui, double& d) {
    if (!q.is_t) { d = 0.0; return xi; }
    d = q.mode ? -(vi + __builtin_fma(q.be, ui, q.al * si)) : -vi;
    const double td = q.t * d; // `step * direction` rounds first (bfgs.rs:94)
    return xi + td;
}

// the vector entries one item needs: row side (row 16 w + (lane & 15) of block I) and column side (this lane's two columns of J)
struct QnS2EvalVec {
    double x_r, v_r, s_r, u_r, b_r, g_r;
    v2d x_c, v_c, s_c, u_c;
};
__device__ __forceinline__ void qn_s2_eval_vec_load(const QnS2Args& a, const double* x, const double* sp, const unsigned ir, const unsigned jc, QnS2EvalVec& v) {
    // (unsigned indices: the loads take a scalar base and a 32-bit lane offset -- as 64-bit lane addresses, ten of them hoisted out
    // of the item loop, they pushed the row loops' window into scratch memory)
    v.x_r = x[ir]; v.v_r = a.F.VV[ir]; v.s_r = sp[ir]; v.u_r = a.F.UN[ir]; v.b_r = a.F.b[ir]; v.g_r = a.F.G[ir];
    v.x_c = ld2(x + jc); v.v_c = ld2(a.F.VV + jc); v.s_c = ld2(sp + jc); v.u_c = ld2(a.F.UN + jc);
}

__device__ __forceinline__ void qn_s2_evalvec_touch(const QnS2EvalVec& v) { // (a use of every entry: the compiler waits for their loads here)
    qn_keepalive(v.x_r); qn_keepalive(v.v_r); qn_keepalive(v.s_r); qn_keepalive(v.u_r); qn_keepalive(v.b_r); qn_keepalive(v.g_r);
    qn_keepalive(v.x_c.x); qn_keepalive(v.x_c.y); qn_keepalive(v.v_c.x); qn_keepalive(v.v_c.y);
    qn_keepalive(v.s_c.x); qn_keepalive(v.s_c.y); qn_keepalive(v.u_c.x); qn_keepalive(v.u_c.y);
}

// One item of the evaluation: the wave's 16 rows against the trial point, folded.
// CONDITIONING (round 3).  f = 1/2 xt'Q xt - b'xt and g(xt)'d = d'Q xt - b'd are what the line search reads.  Round 2 summed
// xt'Q xt, b'xt, d'Q xt and b'd separately and subtracted the totals: each pair is a difference of two sums of size ~ ||b|| ||d||
// whose value is ~ ||g|| ||d||, i.e. a loss of ||b|| / ||g|| in relative accuracy that the reference -- which forms g = Q x - b
// entry by entry and only then the dot product (bfgs.rs:98, line_search/mod.rs:35,47) -- does not have.  The 60-digit pin of
// tests/golden/mt_exact_n1024.json measured it: 1.3e-10 from the truth where the f64 restatement is 5.9e-14 (workload
// "case2_mod", ||g_k|| / ||b|| ~ 1e-5).  Now, on a DIAGONAL item, the lane that owns the diagonal entry of row i subtracts b_i
// (2 b_i for f) from its partial row sum before the multiplication by d_i (x_i): for a diagonally dominant Q -- the benchmark
// family, and any well-scaled SPD problem -- that partial is Q_ii xt_i + one neighbour, i.e. ~ b_i, and the cancellation happens
// there, entry by entry, as in the reference.  The workgroup's sums are then x'(Q xt - 2 b) and d'(Q xt - b) outright; the
// columns for b'xt and b'd stay in the table (zero) so that the state machine's formulas are unchanged.
// FROM_PARK: the rows come from the LDS copy parked during the prologue; otherwise from the register window, each register
// refilled with the same row of the item at `refill` the moment it is consumed (rstride 0: nothing follows, every lane re-reads
// one 16-byte word -- a single request per instruction).
// Leaves the column part of the wave's rows in colred_w[128] and the wave's scalar sums in sred_w[8] (LDS, exchanged by the
// caller after ONE workgroup barrier for a group of items) and returns the row total this lane owns (lanes with (lane & 3) == 0:
// row lane >> 2 of the wave's 16).
// (`diag` is a run-time, wave-uniform flag on purpose: a second instantiation of the row loops per call site made the kernel's code
// 30 % larger, and every OTHER kernel of the iteration started ~1 us later -- rocprofv3 averages, same box: instruction fetch.)
template <bool FROM_PARK, bool NTQ = false> // NTQ: non-temporal refills (Q past the Infinity Cache: every byte is read once per evaluation)
__device__ __forceinline__ double qn_s2_eval_item(const QnEvalReq& q, const QnS2EvalVec& v, const bool diag, const int lane, const int wave,
                                                  v2d (&h)[QN_S2_RPW], const v2d* __restrict__ parkw, const double* __restrict__ refill, const size_t rstride,
                                                  double* __restrict__ colred_w, double (&sacc)[4]) {
    // row side: lane l holds row 16 w + (l & 15) of the tile; column side: this lane's two columns
    double dr;
    const double xr = qn_s2_trial(q, v.x_r, v.v_r, v.s_r, v.u_r, dr);
    double p4 = 0.0, p5 = 0.0; // diagonal items: g'd, #non-finite d over block I (lanes 0..15 of every wave)
    if (diag && lane < 16) { p4 = v.g_r * dr; p5 = isfinite(dr) ? 0.0 : 1.0; }
    v2d xtj, dj;
    {
        double d0, d1;
        xtj.x = qn_s2_trial(q, v.x_c.x, v.v_c.x, v.s_c.x, v.u_c.x, d0);
        xtj.y = qn_s2_trial(q, v.x_c.y, v.v_c.y, v.s_c.y, v.u_c.y, d1);
        dj.x = d0; dj.y = d1;
    }
    if (!qn_s2_row_on(diag, lane, wave)) { xtj = (v2d){0.0, 0.0}; dj = (v2d){0.0, 0.0}; }
    double cx = 0.0, cy = 0.0;
    double racc[QN_S2_RPW];
    // The row loop is the bare minimum: one product pair for the row part, one for the column part, one scalar broadcast.  (Until
    // the middle of round 3 it also carried the two scalar sums -- x_i (q_i - 2 b_i) and d_i (q_i - b_i), with b leaving on the
    // lane that holds Q_ii -- at seven more instructions and a second broadcast per row.  In-kernel stamps of the last and the
    // first wave of a SIMD showed the two waves that share it taking 7.5 us between them for two items: the phase behind the
    // workgroup barrier is instruction issue, not memory.  The scalars now come from the sixteen ROW TOTALS after the fold.)
#pragma unroll
    for (int r = 0; r < QN_S2_RPW; ++r) { // row r of the wave's 16
        v2d hv;
        if (FROM_PARK) hv = parkw[r * 64 + lane];
        else { hv = h[r]; h[r] = qn_sym_ld<NTQ>(refill + (size_t)r * rstride); } // the register this row frees takes the same row of the next item at once
        const double xi = qn_lane_bcast(xr, r);
        double t0 = hv.x * xtj.x;
        t0 = __builtin_fma(hv.y, xtj.y, t0);
        racc[r] = t0;
        cx = __builtin_fma(hv.x, xi, cx);
        cy = __builtin_fma(hv.y, xi, cy);
    }
    if (!qn_s2_col_on(diag, lane, wave)) { cx = 0.0; cy = 0.0; }
    colred_w[2 * lane] = cx;
    colred_w[2 * lane + 1] = cy;
    QnWaveFold<QN_S2_RPW, 32>::run(racc, lane); // lanes with (lane & 3) == 0 hold the total of row lane >> 2
    // the wave's scalar sums: the column part on every lane, the row part on the sixteen lanes that hold a row total.  On a
    // diagonal item b_i is subtracted from the row's total over this tile -- Q_ii xt_i and its neighbours, i.e. ~ b_i for a
    // diagonally dominant Q: the cancellation of g = Q x - b happens here, row by row, before the multiplication by d_i (x_i)
    // and before any further summation (see CONDITIONING above; the 60-digit pin bounds it).
    double pf = xtj.x * cx, pg = dj.x * cx;
    pf = __builtin_fma(xtj.y, cy, pf); // xt_J'(column part): the mirrored half of xt'Q xt
    pg = __builtin_fma(dj.y, cy, pg);  // d_J'(column part) = xt_I'Q_IJ d_J
    {
        const int rq = lane >> 2;
        const double xq = __shfl(xr, rq), dq = __shfl(dr, rq);
        const double bq = diag ? __shfl(v.b_r, rq) : 0.0;
        if ((lane & 3) == 0) {
            pf = __builtin_fma(xq, racc[0] - (bq + bq), pf);
            pg = __builtin_fma(dq, racc[0] - bq, pg);
        }
    }
    // (the lane's partial scalars are ADDED to the group's: one fold serves all the items of a group -- s2_eval_kernel)
    sacc[0] = sacc[0] + pf; sacc[1] = sacc[1] + pg; sacc[2] = sacc[2] + p4; sacc[3] = sacc[3] + p5;
    return racc[0];
}
// ROW SLIVERS (round 3).  nb (nb + 1) / 2 tiles do not divide by the 256 workgroups: at n = 4096 it is 528 = 2 x 256 + 16, so
// sixteen workgroups streamed a third tile while 240 CUs idled -- in-kernel time stamps showed every tile kernel's last workgroups
// ending 4-5 us after the median one, a fifth of the launch.  Now the 512 items that deal out evenly are the work lists (the
// off-diagonal tiles and the first diagonal ones; the list order puts the diagonal tiles last), and the sixteen diagonal tiles
// that are left over are cut into 16 slivers of 8 rows each -- 256 slivers, one per workgroup, one ROW per wave.  A row of a
// diagonal tile is used WHOLE (all 128 columns: the tile is symmetric in memory, Q by construction and H because the slivers of
// the update pass maintain every entry of these tiles), so a sliver has a row part only: no column part, no extra slots, the
// reduce kernels are unchanged.  The row is requested through the refill path of the workgroup's last item (stride 0: every
// register of the window receives it), its vector entries with the last item's, and its sums join the last group's exchange.
struct QnS2Sliver {
    int D, row; // diagonal tile, and this WAVE's row in it
};
__device__ __forceinline__ QnS2Sliver qn_s2_sliver(const QnS2Args& a, const int wave) {
    QnS2Sliver s;
    const int per = a.sl_per ? a.sl_per : 1; // (no slivers: the result is not used -- but a division by zero would let the compiler assume sl_per != 0 everywhere)
    s.D = a.sl_first + (int)blockIdx.x / per;
    s.row = 8 * ((int)blockIdx.x % per) + wave;
    return s;
}
// evaluation: row i of Q_DD against the trial point's block D, in two steps.  qn_s2_sliver_prep turns the sliver's vector entries
// into the trial point's entries as soon as they arrive (before the last item's row loop: seven values stay live across it instead
// of the fourteen loaded ones -- the loop has no registers to spare); qn_s2_eval_sliver consumes the row.  It returns (lane 48) the
// row's share of q_i; the wave's scalar sums -- xt_i (Q_DD xt_D - 2 b)_i, d_i (Q_DD xt_D - b)_i with b_i leaving on the lane that
// holds Q_ii (see CONDITIONING above), g_i d_i and the non-finite count -- go to sred_w as an item's do.
struct QnS2SliverVec {
    v2d xtj;                 // the trial point at this lane's two columns of block D
    double xr, dr, b, gd, nf; // the wave's row: xt_i, d_i, b_i, g_i d_i, d_i non-finite (the same in every lane)
};
__device__ __forceinline__ QnS2SliverVec qn_s2_sliver_prep(const QnEvalReq& q, const QnS2EvalVec& v) {
    QnS2SliverVec o;
    double dr, d0, d1;
    o.xr = qn_s2_trial(q, v.x_r, v.v_r, v.s_r, v.u_r, dr);
    o.dr = dr;
    o.xtj.x = qn_s2_trial(q, v.x_c.x, v.v_c.x, v.s_c.x, v.u_c.x, d0);
    o.xtj.y = qn_s2_trial(q, v.x_c.y, v.v_c.y, v.s_c.y, v.u_c.y, d1);
    o.b = v.b_r;
    o.gd = v.g_r * dr;
    o.nf = isfinite(dr) ? 0.0 : 1.0;
    return o;
}
__device__ __forceinline__ double qn_s2_eval_sliver(const QnS2SliverVec& v, const v2d hv, const int row, const int lane, double (&sacc)[4]) {
    double t0 = hv.x * v.xtj.x;
    t0 = __builtin_fma(hv.y, v.xtj.y, t0);
    const double bsel = (lane == (row >> 1)) ? v.b : 0.0;
    sacc[0] = sacc[0] + v.xr * (t0 - (bsel + bsel));
    sacc[1] = sacc[1] + v.dr * (t0 - bsel);
    if (lane == 0) { sacc[2] = sacc[2] + v.gd; sacc[3] = sacc[3] + v.nf; }
    return t0; // this lane's share of the row's total: folded with the group's scalars (value 6 of the group fold)
}

// Measured and dropped, round 4 (n = 4096, rocprofv3 averages of three alternating runs per build on one box): the request
// PREDICTED by the waves that stream.  In the two transitions of a benchmark iteration the request is a short function of scalars
// -- after an update pass: t = 1 and the direction's coefficients from y's, y'u, u'g+, s'g+; after a rejected trial: More-Thuente's
// interpolated step with phi(t_l) from the memo -- so waves 1..7 computed it themselves (the control block by scalar loads, the
// table's totals published in LDS by wave 0 3-4 us in, the machine's own functions), multiplied their rows out of the register
// window as they arrived (each row copied to the park in case the prediction was wrong), and compared with the machine's request
// at the group's exchange: equal field for field in all 256 workgroups of every launch, results bit-identical.  But the kernel got
// SLOWER: 16.2-16.3 us against 15.5 (and 14.4 k against 14.9 k it/s).  In-kernel stamps say why: a wave that has filled its share of
// the CU's memory queue is held at its next request until the queue drains -- the prediction, wherever it is placed behind the
// first twelve row requests, is reached 6.1-7.0 us after entry, when the machine is done anyway -- and rows that are multiplied
// between the requests hold back the refills of the second item, whose last byte then arrives later than with the parking
// loop, which keeps the queue full until 7 us.  What the waves cannot do is request and multiply at the same time; that would
// take producer and consumer waves of their own and a park for both items (LDS holds one).
// PAIR: every workgroup has exactly two list items and one row sliver (n = 4096 on 256 workgroups: QnS2Args.pair).  The general
// body decides at run time whether there is a second item, a third one, a list to read, a sliver, a parked window; with those
// five flags known the compiler drops the variants they select between and the register copies at their joins -- the phase
// behind the workgroup barrier is instruction issue.
// Round 4, built, measured and dropped: THE LAUNCH THAT STAYS for a whole line search (the two-items-and-a-sliver instance).  At
// n = 4096 a workgroup's share of Q -- two tiles and a sliver, 256 KB -- is exactly what its CU holds after the first evaluation (one
// tile parked in LDS, one in the register window), and the second trial point needs the same tiles against other vectors: the
// workgroups published their six sums (write-through), met at a grid barrier, read the table back (sc1), ran the machine again from
// the control block each had kept in LDS and took the next evaluation from the tiles they held -- no matrix traffic.  Correct (smoke,
// bench), and SLOWER: 47.5 us for a launch of two evaluations against 2 x 15.5 (profiles/r04_q_staying_eval_launch_stamps.txt).
// Per seam: the grid barrier 5 us (a counter took 30-40: 256 read-modify-writes of one word; a generation word per table row, polled
// by every workgroup, 5), the table and the machine again 4.5 us -- 9.5 us against the 11 us of streaming it saves -- and ONE SEAM
// MORE than evaluations, because only the machine's next run knows that the line search is over (with a launch per evaluation that
// run sits in the next kernel's prologue, behind its launch latency).  On top, a second inlined copy of the machine and the
// window held across it cost the row loops their registers (256 + scratch: the two evaluations' arithmetic took 10-12 us instead of
// 4.5).  Even with a 2.8 us hierarchical barrier (tools/seam_probe.hip) and no spills the sum comes out level with three launches.
// BND (bounded variants): the same kernel behind the bounded runs' prologue -- their machine and the request that stores the direction
// (s2_dir_kernel); the evaluation itself is the unbounded one's with a stored direction (dir_mode 0).
template <bool PAIR, bool SHARD = false, bool NTQ = false, bool BND = false>
__global__ __launch_bounds__(QN_S2_TPB, 2) void s2_eval_kernel(const QnS2Args a) {
    static_assert(!(PAIR && SHARD), "the two-items-and-a-sliver instance is the single-rank n = 4096 one");
    __shared__ QnS2Lds L;
    __shared__ double colsum[3][QN_TB];
    __shared__ double colred[3][QN_S2_WAVES][QN_TB]; // [item of the group]
    __shared__ double sred[QN_S2_WAVES][8]; // the group's scalar sums per wave (one fold for all its items)
    __shared__ v2d park[QN_S2_WAVES][QN_S2_RPW][64]; // 128 KB: the first item's rows, parked while wave 0 decides (see below)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t np = (size_t)a.np;
    QN_S2_STAMP(0);
    // Wave 0 runs the prologue first (qn_s2_prologue_w0); waves 1..7 request the window at once.  The first two items are
    // functions of blockIdx: no load.
    const int ij0 = qn_s2_first_item_of<SHARD>(a);
    // The wave's 16 rows of the first item are requested before the control block is known (their addresses do not depend on
    // it), and so are the first item's vector entries for BOTH settings of the two buffer toggles the control block holds
    // (x / trial point, pending / staged s); every register of the window is refilled with the next item's row the moment its
    // row is consumed: 128 KB in flight per workgroup across item boundaries, reductions and barriers.
    // WAVE 0 requests no tile rows before the workgroup barrier (round 3, in-kernel time stamps): its sixteen loads, issued
    // after the machine, sat behind the other seven waves' 240 in the CU's memory queue -- the barrier came 0.6 to 2.6 us after
    // the machine had finished, whatever it took -- and nothing it loads is needed before the first item is done: its rows of
    // the first item are parked by the other waves, its window takes the SECOND item after the barrier and fills while the
    // first is consumed from LDS.  Only its vector entries go out early, behind the control block.
    v2d h[QN_S2_RPW];
    QnS2EvalVec va, v1; // va: x = X0[0], s = S0[0]; v1 holds the other halves' x and s entries
    auto vec_spec = [&]() {
        const int I = ij0 >> 16, J = ij0 & 0xffff;
        const int ir = I * QN_TB + wave * QN_S2_RPW + (lane & 15), jc = J * QN_TB + 2 * lane;
        qn_s2_eval_vec_load(a, a.F.X0, a.F.S0, ir, jc, va);
        v1.x_r = a.F.X0[np + ir]; v1.s_r = a.F.S0[np + ir];
        v1.x_c = ld2(a.F.X0 + np + jc); v1.s_c = ld2(a.F.S0 + np + jc);
    };
    auto window_load = [&](const int ijw) {
        const int I = ijw >> 16, J = ijw & 0xffff;
        const double* qb = a.Q + (size_t)(qn_s2_lrow<SHARD>(a, I) * QN_TB + wave * QN_S2_RPW) * np + (size_t)J * QN_TB + qn_s2_col(I == J, lane, wave);
#pragma unroll
        for (int r = 0; r < QN_S2_RPW; ++r) h[r] = qn_sym_ld<NTQ>(qb + (size_t)r * np);
    };
    unsigned cw = 0u;
    int ij1;     // the second item: what the parked window is refilled from (looked up after the first requests: nothing in front of them)
    bool parked; // (uniform) a workgroup with one item parks nothing
    // PARKING.  The prologue of wave 0 -- one memory round trip, the sums of 256 rows, the state machine on one lane -- takes
    // 6-9 us, and the 128 KB register window lands in 5.  Round 2 let the memory system idle until the decision was there.
    // Now waves 1..7 move each row of the first item into LDS the moment it arrives and request the same row of the SECOND item
    // into the register it frees: 256 KB per CU stream back to back from kernel entry whatever the machine takes, the first
    // item is then consumed from LDS and the second from registers.  Wave 0's sixteen rows of the first item are fetched and
    // parked by the other waves too (three rows each, behind their own): a wave that asked for its rows only after the
    // prologue was 3 us behind the rest at the pair's barrier (in-kernel time stamps).
    // Measured and dropped, round 5 (tools/experiments/r05_eval_first_item_by_lds_dma.patch): THE FIRST ITEM STRAIGHT INTO LDS --
    // global_load_lds_dwordx4, lane l's 16 bytes land at row + 16 l, which is the park's layout: no registers, no copy -- and the second item into
    // the register window right behind it, 245 KB per CU requested from entry on.  Bit-identical, and 15.3 us against 15.0-15.1: a wave that has
    // filled its share of the CU's memory queue is held at its next request until the queue drains, so "requested from entry on" is not what
    // happens -- in-kernel stamps: the first item's nineteen requests per wave are out 3.9 us after entry, the second item's sixteen 7.9 us
    // after entry (the CU takes requests at the rate it delivers: 31 KB/us) -- and a wave that is held at a request cannot multiply: the
    // workgroup barrier moved from 8.5 to 9.3 us.  What would overlap the first item's arithmetic with the second item's bytes is a second
    // set of waves that does nothing but request (16 waves per CU at 128 registers each): a different kernel.
    // (The parking lives INSIDE the branch of the waves that do it.  As a second `if (wave != 0)` behind the first one it left the compiler a
    // path that exists in no wave -- this branch's vector-entry loads, then around the parking -- on which those loads are still pending
    // behind the workgroup barrier; s_waitcnt is per wave, so EVERY wave then waited with vmcnt(0) in front of the first item's rows, which
    // are in LDS, for the last row of the SECOND item.  Both branches now leave the vector entries waited for: qn_s2_evalvec_touch.)
    if (wave == 0) {
        qn_s2_prologue_w0<QN_S2_EVAL, SHARD, decltype(vec_spec)&, false, BND>(a, L, vec_spec);
        qn_s2_evalvec_touch(va);
        qn_keepalive(v1.x_r); qn_keepalive(v1.s_r); qn_keepalive(v1.x_c.x); qn_keepalive(v1.x_c.y); qn_keepalive(v1.s_c.x); qn_keepalive(v1.s_c.y);
        ij1 = qn_s2_second_item(a);
        parked = PAIR || ij1 >= 0;
    } else {
        if (wave <= 4) cw = qn_code_warm_issue((wave - 1) * 64 + lane); // (qn_kernels.hip.h, CODE WARM-UP: 32 KB.  All seven waves -- 56 KB, the
        window_load(ij0);                                               // whole two-item instance -- in front of their rows: 15.2 -> 15.55 us;
        vec_spec();                                                     // the last 24 KB behind the rows instead: 15.25 -> 15.35)
        // (Measured and dropped: the vector entries requested IN FRONT of the window.  The compiler then gives the second item a register window of
        // its own and requests it right behind the first -- everything in flight 1.6 us after entry, every parked row waited for exactly -- and the
        // kernel takes 16.05 us instead of 15.25: the second item's bytes delay the first item's, whose arithmetic is what the barrier waits for.)
        ij1 = qn_s2_second_item(a);
        parked = PAIR || ij1 >= 0;
        if (parked) {
            const int I1 = ij1 >> 16, J1 = ij1 & 0xffff;
            const double* q1 = a.Q + (size_t)(qn_s2_lrow<SHARD>(a, I1) * QN_TB + wave * QN_S2_RPW) * np + (size_t)J1 * QN_TB + qn_s2_col(I1 == J1, lane, wave);
            // (wave 0's rows are requested BEFORE the second item's: a wave's loads return in order, and behind the refills these three
            // -- which the barrier below waits for -- arrived with the last byte of the second window, 9.5 us into the kernel: the
            // workgroup then started on the first item when both had landed, however early the machine was done)
            const int I0 = ij0 >> 16, J0 = ij0 & 0xffff;
            const double* q0 = a.Q + (size_t)(qn_s2_lrow<SHARD>(a, I0) * QN_TB) * np + (size_t)J0 * QN_TB + 2 * lane; // wave 0's rows: no clone lanes (qn_s2_col(., ., 0) = 2 lane)
            const int r0 = (wave - 1) * 3;
            v2d t3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) t3[k] = qn_sym_ld<NTQ>(q0 + (size_t)min(r0 + k, QN_S2_RPW - 1) * np);
            // The vector entries are waited for HERE, in front of the second item's requests (they arrive right behind the first item's rows):
            // see the note above.
            qn_s2_evalvec_touch(va);
            qn_keepalive(v1.x_r); qn_keepalive(v1.s_r); qn_keepalive(v1.x_c.x); qn_keepalive(v1.x_c.y); qn_keepalive(v1.s_c.x); qn_keepalive(v1.s_c.y);
            QN_S2_STAMP_T(6, 448); // (wave 7: the first item has landed)
#pragma unroll
            for (int r = 0; r < QN_S2_RPW; ++r) {
                park[wave][r][lane] = h[r];
                h[r] = qn_sym_ld<NTQ>(q1 + (size_t)r * np);
            }
            QN_S2_STAMP_T(7, 448); // (its sixteen rows parked, the second item requested)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (r0 + k < QN_S2_RPW) park[0][r0 + k][lane] = t3[k]; // (they came in right behind the wave's own sixteen)
            QN_S2_STAMP_T(8, 448);
        }
        else { // (nothing is parked: the entries are waited for all the same -- both branches leave them so)
            qn_s2_evalvec_touch(va);
            qn_keepalive(v1.x_r); qn_keepalive(v1.s_r); qn_keepalive(v1.x_c.x); qn_keepalive(v1.x_c.y); qn_keepalive(v1.s_c.x); qn_keepalive(v1.s_c.y);
        }
    }
    QN_S2_STAMP(1);
    __syncthreads();
    qn_s2_ctl_out(a, L);
    qn_code_warm_done(cw, a.n < 0, a.wgS);
    if (!L.mine) return;
    QN_S2_STAMP(2);
    if (wave == 0) window_load(parked ? ij1 : ij0); // (parked: its rows of the first item are in LDS)
    const QnEvalReq q = qn_s2_eval_req<true, BND>(L.c, false);
    const double* __restrict__ x = a.F.X0 + (size_t)q.xc * np;
    const double* __restrict__ sp = a.F.S0 + (size_t)q.sc * np;
    if (q.xc) { va.x_r = v1.x_r; va.x_c = v1.x_c; }
    if (q.sc) { va.s_r = v1.s_r; va.s_c = v1.s_c; }
    // threads 0..5: this workgroup's running total of scalar k (items in list order; waves in order inside an item).  One
    // register pair per thread -- thread 0 holding all six cost the row loops twelve registers -- and six short chains of LDS
    // reads instead of one long one at the end of the launch.
    double wgk = 0.0;
    // Items are taken in GROUPS of two -- three when an odd item is left at the end of the list: the rows of all of them are
    // consumed and folded back to back, then ONE exchange through LDS, one barrier and one set of slot stores serves the group
    // (round 2 did all of that per item: 2-3 us of fold / barrier / store latency each, with every byte already on chip).  The
    // group of three matters at n = 4096: 528 items on 256 CUs leave sixteen workgroups with a third (diagonal) item, and as a
    // round of its own it kept the whole launch waiting 5 us for those sixteen (in-kernel time stamps of the workgroups' ends).
    int ija = ij0, ijb = ij1;
    QnS2EvalVec vb;
    // The FIRST group is peeled out of the loop over groups (the same body, instantiated twice).  Inside the loop the compiler
    // had to assume that loads of the previous trip were still pending when the parked item's LDS reads reuse their registers,
    // and put `s_waitcnt vmcnt(5..0)` in front of them: counted in order, those waits held the parked item back until the SECOND
    // item's rows had all arrived -- the overlap the parking exists for was gone (in-kernel stamps: 3.4 us for an item that is
    // entirely in LDS).
    auto run_group = [&](auto first_tag, const int it) __attribute__((always_inline)) -> bool {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int Ia = ija >> 16, Ja = ija & 0xffff, Ib = ijb >> 16, Jb = ijb & 0xffff;
        const bool has_b = PAIR || ijb >= 0; // (uniform)
        const bool diag_a = Ia == Ja, diag_b = has_b && Ib == Jb;
        // the items after this pair: where the window is refilled from while item b is consumed
        int ijc = -1, ijd = -1;
        if (!PAIR && has_b && it + 2 < a.maxk) ijc = a.item_ij[(size_t)(it + 2) * a.G + blockIdx.x];
        if (!PAIR && ijc >= 0 && it + 3 < a.maxk) ijd = a.item_ij[(size_t)(it + 3) * a.G + blockIdx.x];
        const bool take_c = ijc >= 0 && ijd < 0; // the last, odd item joins this group
        const int Ic = ijc >> 16, Jc = ijc & 0xffff;
        const bool diag_c = take_c && Ic == Jc;
        // (each refill address is formed right in front of the item that uses it: three address pairs held across the items cost
        // registers the row loops need)
        auto tile_ptr = [&](int ij_) {
            const int I_ = ij_ >> 16, J_ = ij_ & 0xffff;
            return a.Q + (size_t)(qn_s2_lrow<SHARD>(a, I_) * QN_TB + wave * QN_S2_RPW) * np + (size_t)J_ * QN_TB + qn_s2_col(I_ == J_, lane, wave);
        };
        // (row slivers: every workgroup has exactly a.maxk items and one sliver, which joins the last group in item c's place)
        const bool sliver = PAIR || (!SHARD && a.sl_per != 0 && ijc < 0); // (uniform; row-sharded runs have no slivers)
        const QnS2Sliver sl = qn_s2_sliver(a, wave);
        const double* slp = a.Q + (size_t)(sl.D * QN_TB + sl.row) * np + (size_t)sl.D * QN_TB + 2 * lane;
        QnS2SliverVec slv{};
        double row_a, row_b = 0.0, row_c = 0.0;
        double sacc[4] = {0.0, 0.0, 0.0, 0.0}; // x'(Q xt - 2b), d'(Q xt - b), g'd, #non-finite d: this lane's share over the group's items
        if (FIRST && parked) row_a = qn_s2_eval_item<true, NTQ>(q, va, diag_a, lane, wave, h, &park[wave][0][0], nullptr, 0, colred[0][wave], sacc);
        else row_a = qn_s2_eval_item<false, NTQ>(q, va, diag_a, lane, wave, h, nullptr, has_b ? tile_ptr(ijb) : (sliver ? slp : tile_ptr(ija)), has_b ? np : 0, colred[0][wave], sacc);
        if (FIRST) { qn_keepalive(row_a); QN_S2_STAMP_T(13, 448); }
        if (has_b) {
            qn_s2_eval_vec_load(a, x, sp, Ib * QN_TB + wave * QN_S2_RPW + (lane & 15), Jb * QN_TB + 2 * lane, vb);
            if (sliver) { // the sliver's entries fly behind item b's
                // (the indices pass through an empty asm: as loop invariants the ten lane addresses were formed once, in front of
                // the item loop, and held in twenty registers across the row loops)
                unsigned sir = sl.D * QN_TB + sl.row, sjc = sl.D * QN_TB + 2 * lane;
                asm volatile("" : "+v"(sir), "+v"(sjc));
                qn_s2_eval_vec_load(a, x, sp, sir, sjc, va);
                slv = qn_s2_sliver_prep(q, va);
            }
            row_b = qn_s2_eval_item<false, NTQ>(q, vb, diag_b, lane, wave, h, nullptr, ijc >= 0 ? tile_ptr(ijc) : (sliver ? slp : tile_ptr(ijb)), ijc >= 0 ? np : 0, colred[1][wave], sacc);
        } else if (sliver) {
            unsigned sir = sl.D * QN_TB + sl.row, sjc = sl.D * QN_TB + 2 * lane;
            asm volatile("" : "+v"(sir), "+v"(sjc));
            qn_s2_eval_vec_load(a, x, sp, sir, sjc, va);
            slv = qn_s2_sliver_prep(q, va);
        }
        // the next item's vector entries go out now: for the item that joins this group, or for the next group while this
        // one's sums are exchanged and stored
        if (ijc >= 0) qn_s2_eval_vec_load(a, x, sp, Ic * QN_TB + wave * QN_S2_RPW + (lane & 15), Jc * QN_TB + 2 * lane, va);
        if (take_c) row_c = qn_s2_eval_item<false, NTQ>(q, va, diag_c, lane, wave, h, nullptr, tile_ptr(ijc), 0, colred[2][wave], sacc);
        if (FIRST) { qn_keepalive(row_b); QN_S2_STAMP_T(14, 448); }
        double t0s = 0.0;
        if (sliver) t0s = qn_s2_eval_sliver(slv, h[0], sl.row, lane, sacc); // (h[0]: the window holds the sliver's row sixteen times)
        { // ONE fold for the group: the four scalars and the sliver's row total (value 6: lane 48 holds it)
            double sv[8] = {sacc[0], sacc[1], 0.0, 0.0, sacc[2], sacc[3], t0s, 0.0};
            QnWaveFold<8, 32, true>::run(sv, lane);
            if ((lane & 7) == 0 && lane < 48) sred[wave][lane >> 3] = sv[0];
            if (sliver) row_c = sv[0];
        }
        if (FIRST) { qn_keepalive(row_a); qn_keepalive(row_b); qn_keepalive(row_c); QN_S2_STAMP(3); QN_S2_STAMP_T(12, 448); }
        __syncthreads();
        if (FIRST) QN_S2_STAMP(4);
        if (tid < 3 * QN_TB) { // threads 0..127: item a's column part, 128..255: item b's, 256..383: item c's
            const int e = tid >> 7, cidx = tid & (QN_TB - 1);
            if (e == 0 || (e == 1 && has_b) || (e == 2 && take_c)) {
                double acc = colred[e][0][cidx];
#pragma unroll
                for (int w = 1; w < QN_S2_WAVES; ++w) acc = acc + colred[e][w][cidx];
                const bool dg = e == 0 ? diag_a : (e == 1 ? diag_b : diag_c);
                const int Ie = e == 0 ? Ia : (e == 1 ? Ib : Ic), Je = e == 0 ? Ja : (e == 1 ? Jb : Jc);
                if (dg) colsum[e][cidx] = acc; // both parts of a diagonal tile belong to block-row I: one slot
                else a.partE[(unsigned)((Je * a.nb + Ie) * QN_TB + cidx)] = acc;
            }
        }
        if (tid < QN_S2_NSE) wgk = wgk + qn_s2_wave_total(sred, tid); // (waves in order; scalars 2..5 are zero off the diagonal items)
        if (diag_a || diag_b || diag_c) __syncthreads(); // (uniform)
        if ((lane & 3) == 0) {
            const int rl = wave * QN_S2_RPW + (lane >> 2);
            double v = row_a;
            if (diag_a) v = v + colsum[0][rl];
            a.partE[(unsigned)((Ia * a.nb + Ja) * QN_TB + rl)] = v;
            if (has_b) {
                v = row_b;
                if (diag_b) v = v + colsum[1][rl];
                a.partE[(unsigned)((Ib * a.nb + Jb) * QN_TB + rl)] = v;
            }
            if (take_c) {
                v = row_c;
                if (diag_c) v = v + colsum[2][rl];
                a.partE[(unsigned)((Ic * a.nb + Jc) * QN_TB + rl)] = v;
            }
        }
        if (sliver && lane == 48) a.partE[(unsigned)((sl.D * a.nb + sl.D) * QN_TB + sl.row)] = row_c;
        if (FIRST) QN_S2_STAMP(5);
        if (ijc < 0 || take_c) return true;
        ija = ijc; ijb = ijd;
        __syncthreads(); // colred / colsum / sred are rewritten by the next group
        return false;
    };
    if (!run_group(std::true_type{}, 0)) {
        if (!PAIR)
            for (int it = 2;; it += 2)
                if (run_group(std::false_type{}, it)) break;
    }
    QN_S2_STAMP(15);
    if (tid < QN_S2_NSE) { // sred column -> table column: xt'(Q xt - 2b), d'(Q xt - b), (b'xt = 0), (b'd = 0), g'd, #non-finite d
        const int col = tid == 1 ? 2 : (tid == 2 ? 1 : tid);
        if (!SHARD) a.wgS[((size_t)a.parity * QN_S2_ROW + col) * a.trows + blockIdx.x] = wgk;
        else if (tid != 2 && tid != 3) { // this rank's slice of the exchanged scalars: the four columns that are not zero by construction
            const int ec = tid < 2 ? tid : tid - 2;
            a.evS[(((size_t)a.parity * a.sh_world + a.sh_rank) * QN_S2SH_NEC + ec) * QN_S2_MAXG + blockIdx.x] = wgk;
        }
    }
}

// Slot sums of block-row R: 4 x 128 threads, each quarter adds a quarter of the row's nb slots in slot order and the quarters are
// combined in order through LDS.  In two steps, so that the first 16 slots of every quarter (all of them up to n = 8192) are in
// flight while the control block is still on its way: the addresses do not depend on it.
struct QnS2Slots { double v[16]; };
// COH (tail reduce, s2_hpass_kernel<.., TRED>): the slots were stored by other workgroups of the SAME launch -- written through
// (sc1 stores) and read past the non-coherent cache levels (sc1 loads): MI355X_MICROARCH.md, hand-offs, first row of the table
template <bool COH> __device__ __forceinline__ void qn_s2_slot_st(double* p, const double v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool COH> __device__ __forceinline__ double qn_s2_slot_ld(const double* p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <int NRHS = 2, bool COH = false>
__device__ __forceinline__ void qn_s2_slot_issue(const double* __restrict__ part, int nb, int R, int rhs, QnS2Slots& S) {
    const int i = threadIdx.x & (QN_TB - 1), qd = threadIdx.x >> 7;
    const int per = (nb + 3) / 4;
    const int k_lo = qd * per, k_hi = min(nb, k_lo + per);
    const double* p = part + (((size_t)R * nb) * NRHS + rhs) * QN_TB + i;
#pragma unroll
    for (int u = 0; u < 16; ++u) S.v[u] = (k_lo + u < k_hi) ? qn_s2_slot_ld<COH>(p + (size_t)(k_lo + u) * NRHS * QN_TB) : 0.0;
}
// threads 0..127 return the total of row (tid & 127); all 512 threads call it
template <int NRHS = 2, bool COH = false>
__device__ __forceinline__ double qn_s2_slot_sum(const double* __restrict__ part, int nb, int R, int rhs, const QnS2Slots& S, double (*qbuf)[QN_TB]) {
    const int i = threadIdx.x & (QN_TB - 1), qd = threadIdx.x >> 7;
    const int per = (nb + 3) / 4;
    const int k_lo = qd * per, k_hi = min(nb, k_lo + per);
    const double* p = part + (((size_t)R * nb) * NRHS + rhs) * QN_TB + i;
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = acc + S.v[u];
    for (int k0 = k_lo + 16; k0 < k_hi; k0 += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (k0 + u < k_hi) ? qn_s2_slot_ld<COH>(p + (size_t)(k0 + u) * NRHS * QN_TB) : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = acc + v[u];
    }
    if (qd > 0) qbuf[qd - 1][i] = acc;
    __syncthreads();
    const double tot = (qd == 0) ? ((acc + qbuf[0][i]) + qbuf[1][i]) + qbuf[2][i] : 0.0;
    __syncthreads(); // qbuf is reused by the next call
    return tot;
}

// TOUCH WORKGROUPS (round 6; n = 4096's two-items-and-a-sliver instance).  The iteration's two small launches -- the accept-reduce (nb workgroups) and
// the update-reduce (2 nb) -- leave the chip's fabric idle for ~10 of the iteration's 61 us, and the tile launch behind each of them starts by streaming,
// in every workgroup, a tile whose address follows from the workgroup's index alone.  An XCD's 4 MB L2 keeps its bytes across a kernel boundary
// (tools/l2_keep_probe.hip) and workgroup g of a launch runs on XCD g mod 8: so the small launch carries G extra workgroups, and extra workgroup b
// TOUCHES the first a.touch rows of every wave's sixteen of the tile that workgroup b of the NEXT launch requests first -- plain loads, nothing kept.
// The tile launch then finds those rows in its XCD's L2 (twice the fabric's rate) and has that much less to bring across.  Half a tile per workgroup
// (8 rows per wave: 64 KB per CU, 2 MB per XCD) fits inside the small launches' own time (tools/l2_prefetch_probe.hip: small launch unchanged, update
// tiles and the first evaluation each ~1 us shorter); whole tiles make the small launches as much longer as the tile launches get shorter.
// What the first measurement of the real kernels added (profiles/r06_e_*): (1) the small launches' own workgroups want their inputs -- control block, table,
// slots -- in the first 1.9 us, and 2 MB per XCD requested at the same moment stand in front of them: the accept-reduce's touching workgroups SLEEP first
// (a.touch_delay), its machine runs on LDS from 1.9 to 3.3 us and that is when the fabric is theirs; the update-reduce's body is 2.5 us in all, it touches less;
// (2) which workgroups touch must follow from the workgroup index and a CONSTANT: the test against a.nb put a scalar load and its wait in front of every
// workgroup's first instruction (accept-reduce 5.65 -> 6.2 us with nobody touching) -- so the kernels with touching workgroups are instantiations of
// their own (NBT = nb = 32: n = 4096, the one size with the two-items-and-a-sliver instance), and every other launch runs the code it ran before.
// Loads only: whatever they bring -- or do not bring, if the placement of workgroups is not the one assumed -- no value in the run depends on it.
template <int ROWS>
__device__ __forceinline__ void qn_s2_touch_rows(const double* __restrict__ base, const size_t np) {
    v2d h[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) h[r] = ld2(base + (size_t)r * np);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) asm volatile("" :: "v"(h[r].x), "v"(h[r].y));
}
__device__ __forceinline__ void qn_s2_touch(const double* __restrict__ M, const QnS2Args& a, const int ij, const int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int I = ij >> 16, J = ij & 0xffff;
    // (an opaque copy of np: as a value both sides of the caller's branch use, the compiler moved its first use -- and with it the wait for the kernel-argument
    // load -- in FRONT of the branch: a scalar memory round trip ahead of every workgroup's first request)
    int npi = a.np;
    asm volatile("" : "+s"(npi));
    const size_t np = (size_t)npi;
    const double* base = M + (size_t)(I * QN_TB + wave * QN_S2_RPW) * np + (size_t)J * QN_TB + 2 * lane;
    switch (rows) { // (uniform)
    case 4: qn_s2_touch_rows<4>(base, np); break;
    case 6: qn_s2_touch_rows<6>(base, np); break;
    case 8: qn_s2_touch_rows<8>(base, np); break;
    case 10: qn_s2_touch_rows<10>(base, np); break;
    case 12: qn_s2_touch_rows<12>(base, np); break;
    case 16: qn_s2_touch_rows<16>(base, np); break;
    default: break;
    }
}
// ... rows j, j + 7, .. (< rows) of every wave's sixteen: a seventh of what qn_s2_touch takes of a tile (seven workgroups share it)
__device__ __forceinline__ void qn_s2_touch_seventh(const double* __restrict__ M, const QnS2Args& a, const int ij, const int rows, const int j) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int I = ij >> 16, J = ij & 0xffff;
    int npi = a.np;
    asm volatile("" : "+s"(npi));
    const size_t np = (size_t)npi;
    const double* base = M + (size_t)(I * QN_TB + wave * QN_S2_RPW) * np + (size_t)J * QN_TB + 2 * lane;
    v2d h0 = {0.0, 0.0}, h1 = {0.0, 0.0}, h2 = {0.0, 0.0};
    if (j < rows) h0 = ld2(base + (size_t)j * np);
    if (j + 7 < rows) h1 = ld2(base + (size_t)(j + 7) * np);
    if (j + 14 < rows) h2 = ld2(base + (size_t)(j + 14) * np);
    asm volatile("" :: "v"(h0.x), "v"(h0.y), "v"(h1.x), "v"(h1.y), "v"(h2.x), "v"(h2.y));
}

// accept-reduce: block-row R of the LAST evaluation becomes vectors: q_i, g+ = q - b, y = g+ - g, x+ and s = x+ - x
// (bfgs.rs:94-99), and the five sums the update needs
// SHARD (row-sharded runs): q_i is the sum over the ranks, in rank order, of the partial vectors the exchange has gathered in xg
// (each rank's share summed by s2sh_vsum_kernel, qn_sym2sh.hip.h); every rank forms every block-row: replicated work, the same bits.
// DECIDE (SHARD only): the partial vectors of the LAST evaluation are already gathered (they rode on its scalar exchange): this launch's
// prologue is the deciding one -- it consumes the evaluation, accepts, and the launch serves the request it has just made.
template <bool SHARD = false, bool DECIDE = false, int NBT = 0> // (NBT: the instantiation with TOUCH workgroups behind its nb = NBT own)
__global__ __launch_bounds__(QN_S2_TPB) void s2_vec_kernel(const QnS2Args a) {
    __shared__ QnS2Lds L;
    __shared__ double qbuf[3][QN_TB];
    __shared__ double bred[2][8];
    const int R = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (NBT > 0 && __builtin_expect((int)blockIdx.x >= NBT, 0)) { // TOUCH workgroup (uniform): the update tiles follow -- H, the first list item
        // (this kernel's 150 registers admit ONE workgroup per CU: the grid stays at G -- nb + G workgroups would leave nb of them waiting for a CU -- and the
        // last nb tiles are shared out, a seventh each, among the G - nb = 7 nb touching workgroups: tile G - nb + b mod nb belongs to the same XCD as b)
        static_assert(NBT == 0 || NBT == 32, "seven touching workgroups share a tile: G = 8 nb");
        const int b = (int)blockIdx.x - NBT;
        for (int k = 0; k < a.touch_delay; k += 8) __builtin_amdgcn_s_sleep(8);
        qn_s2_touch(a.H, a, qn_s2_first_item(b, NBT), a.touch);
        qn_s2_touch_seventh(a.H, a, qn_s2_first_item(a.G - NBT + (b & (NBT - 1)), NBT), a.touch, b / NBT);
        return;
    }
    QnS2Slots S0;
    unsigned cw[4] = {0u, 0u, 0u, 0u};
    QN_S2_STAMP(0);
    // (wave 0 requests its share of the slots behind the control block and the table, BEFORE it runs the machine: requested after
    // it -- 3.5 us into a 6 us kernel -- they were what the slot sums waited for)
    // ... and so do the vector entries of the block's rows (threads 0..127), for both settings of the two buffer toggles the control
    // block holds: requested behind the barrier they were a memory round trip between the machine's answer and the arithmetic
    // (round 4: 0.7 us of a 7.5 us kernel whose critical path is the machine).
    const size_t np = (size_t)a.np;
    double e_x0 = 0.0, e_x1 = 0.0, e_v = 0.0, e_s0 = 0.0, e_s1 = 0.0, e_u = 0.0, e_b = 0.0, e_g = 0.0;
    auto entries = [&]() {
        if (tid < QN_TB) {
            const int gi = R * QN_TB + tid;
            e_x0 = a.F.X0[gi]; e_x1 = a.F.X0[np + gi]; e_v = a.F.VV[gi]; e_s0 = a.F.S0[gi]; e_s1 = a.F.S0[np + gi];
            e_u = a.F.UN[gi]; e_b = a.F.b[gi]; e_g = a.F.G[gi];
        }
    };
    if (wave == 7) { // (qn_kernels.hip.h, CODE WARM-UP: 32 KB, the whole kernel -- 7.1 -> 5.8 us, profiles/r05_o_*)
#pragma unroll
        for (int k = 0; k < 4; ++k) cw[k] = qn_code_warm_issue(k * 64 + lane);
    }
    if (SHARD) { if (wave == 0) qn_s2_prologue_w0<(DECIDE ? QN_S2_VECD : QN_S2_VEC), true>(a, L, entries); else entries(); }
    else if (wave == 0) qn_s2_prologue_w0<QN_S2_VEC>(a, L, [&]() { qn_s2_slot_issue<1>(a.partE, a.nb, R, 0, S0); entries(); });
    else { qn_s2_slot_issue<1>(a.partE, a.nb, R, 0, S0); entries(); }
    __syncthreads();
    QN_S2_STAMP(2);
    qn_s2_ctl_out(a, L);
    qn_code_warm_done(cw[0] ^ cw[1] ^ cw[2] ^ cw[3], a.n < 0, a.wgS);
    if (!L.mine) return;
    const QnEvalReq q = qn_s2_eval_req<false>(L.c, true);
    double* __restrict__ xt = a.F.X0 + (size_t)(1 - q.xc) * np;
    double* __restrict__ sstage = a.F.S0 + (size_t)(1 - q.sc) * np;
    double qi = 0.0;
    if (!SHARD) qi = qn_s2_slot_sum<1>(a.partE, a.nb, R, 0, S0, qbuf);
    else if (tid < QN_TB) {
        const double* xp = a.xg + (size_t)R * QN_TB + tid;
        qi = xp[0];
        for (int r = 1; r < a.sh_nsum; ++r) qi = qi + xp[(size_t)r * np];
    }
    double p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = 0.0;
    if (tid < QN_TB) {
        const int gi = R * QN_TB + tid;
        double di;
        const double xi = q.xc ? e_x1 : e_x0;
        const double xti = qn_s2_trial(q, xi, e_v, q.sc ? e_s1 : e_s0, e_u, di);
        const double bi = e_b, go = e_g;
        const double gti = qi - bi;
        const double yi = gti - go;
        const double si = xti - xi; // s = x+ - x, not t d (bfgs.rs:96)
        a.F.GT[gi] = gti;
        a.F.Y[gi] = yi;
        xt[gi] = xti;
        sstage[gi] = si;
        p[0] = yi * yi; p[1] = yi * si; p[2] = gti * gti; p[3] = si * si; p[4] = si * gti;
    }
    if (wave < 2) { // threads 0..127 hold the rows
        QnWaveFold<8, 32>::run(p, lane);
        if ((lane & 7) == 0) bred[wave][lane >> 3] = p[0];
    }
    __syncthreads();
    if (tid < QN_S2_NR) a.wgS[((size_t)a.parity * QN_S2_ROW + tid) * a.trows + R] = bred[0][tid] + bred[1][tid];
    QN_S2_STAMP(14);
}

// ------------------------------------------------------------------------------------------------
// update tiles: pending rank-2 update in place, row / column slots of [y, g+] (or [g, g]).
// ONE branch-free body serves every request: no update pending = zero coefficients, a direction pass (one right-hand side,
// once per qn_minimize call) = a two-right-hand-side pass with g in both places -- sixteen specialised copies of the row loop
// cost more in registers (the compiler kept the load window in scratch memory across them) than the skipped flops were worth.
// ------------------------------------------------------------------------------------------------
// (the epilogue of a block-row, given its two totals on threads 0..127: called by s2_hreduce_kernel and by the tail reduce of
// s2_hpass_kernel<.., TRED> -- one body, the same sums in the same order, hence the same bits; all 512 threads call it)
// (gp, yv: the block-row's entries of g+ and y on threads 0..127 -- the accept-reduce wrote them a launch earlier, so s2_hreduce_kernel requests them
// at entry, with its slots: behind the totals they were one more memory round trip at the end of a 5 us kernel)
// (half: s2_hreduce_kernel gives a block-row's two right-hand sides to TWO workgroups -- 0: u, the sums and the commit of g; 1: v, passed as tot1,
// and nothing else; -1: both, the tail reduce)
template <bool SR1 = false> // (SR1: also (s - u)'y, the update's denominator, as a third sum -- snew is the staged s)
__device__ __forceinline__ void qn_s2_hreduce_row(const QnS2Args& a, const int R, const int nrhs, const double tot0, const double tot1, double (*bred)[8],
                                                  const double gp, const double yv, const int half = -1, const double snew = 0.0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (half == 1) { // (uniform)
        if (tid < QN_TB && nrhs == 2) a.F.VV[R * QN_TB + tid] = tot1;
        return;
    }
    double p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = 0.0;
    if (tid < QN_TB) {
        const int gi = R * QN_TB + tid;
        if (nrhs == 2) {
            a.F.UN[gi] = tot0;
            if (half < 0) a.F.VV[gi] = tot1;
            p[0] = yv * tot0; // y'u = y'H+y
            if (SR1) p[2] = (snew - tot0) * yv; // (s - u)'y, the update's denominator (sr1_b.rs:145)
            p[1] = tot0 * gp;        // u'g+
        } else {
            a.F.VV[gi] = tot0; // direction pass: v = H g (bfgs.rs:47)
        }
        a.F.G[gi] = gp; // commit g <- g+
    }
    if (wave < 2) { // threads 0..127 hold the rows
        QnWaveFold<8, 32>::run(p, lane);
        if ((lane & 7) == 0) bred[wave][lane >> 3] = p[0];
    }
    __syncthreads();
    if (tid < (SR1 ? 3 : 2)) a.wgS[((size_t)a.parity * QN_S2_ROW + tid) * a.trows + R] = bred[0][tid] + bred[1][tid];
}
struct QnS2HReq { // the update-pass request, decoded once per launch
    const double *sp, *up, *r0v, *gt;
    double c_ss, c_su, c_uu;
    bool pending;
};
// the vector entries one item needs: row side (row 16 w + (lane & 15) of block I) and column side (this lane's two columns of J)
struct QnS2HVec {
    double s_r, u_r, y0_r, y1_r;
    v2d s_c, u_c, a0, a1;
};
template <bool RHS = true> // (false: s and u only -- the right-hand sides come from LDS)
__device__ __forceinline__ void qn_s2_hvec_load(const QnS2HReq& q, const unsigned ir, const unsigned jc, QnS2HVec& v) {
    v.s_r = q.sp[ir]; v.u_r = q.up[ir];
    v.s_c = ld2(q.sp + jc); v.u_c = ld2(q.up + jc);
    if (RHS) { v.y0_r = q.r0v[ir]; v.y1_r = q.gt[ir]; v.a0 = ld2(q.r0v + jc); v.a1 = ld2(q.gt + jc); }
}

// FOLDED ACCEPT-REDUCE (round 3; workgroups with <= 3 items, i.e. n <= 4096: QnS2Args.fold).  Round 2 turned the accepted
// evaluation's slots into vectors in a launch of its own (s2_vec_kernel: 7 us at n = 4096, most of it a kernel boundary, a control
// block round trip and a run of the state machine), and only then started the update tiles.  But the tiles need nothing the
// machine decides after the acceptance: the vectors g+, y of the accepted point, and the PENDING update's s, u and coefficients.
// So this kernel also answers QN_PH_REQ_VEC (L.mine == 2): every workgroup sums the evaluation's slots for the blocks of its OWN
// items (<= 6 blocks x 32 slots x 1 KB, L2-resident; the same order as qn_s2_slot_sum, hence the same bits), forms g+ = q - b and
// y = g+ - g for them in LDS, and goes on into the tiles; the workgroup that holds the diagonal item (R, R) also writes block R
// of the vectors (g+, y, x+, s) and its five sums.  The machine consumes those sums in the NEXT launch's prologue and finds the
// tiles of the pass it then asks for already done (QnCtl.spec_tiles, qn_s2_advance).  An evaluation at x itself (the first of a
// run) is followed by a direction pass: g+ in both right-hand sides, as s2_hreduce_kernel expects it.
#define QN_S2_FOLD_ITEMS 3
#define QN_S2_FOLD_BLOCKS (2 * QN_S2_FOLD_ITEMS)

// BFGS: true -> the update has the (s u' + u s') and s s' terms (bfgs.rs:115-124); false -> DFP: s s' and u u' (dfp.rs:115-120)
// FOLD: the instantiation for a.fold (its right-hand sides always come from LDS: staged from the vectors when the request is a
// plain update pass); the other one is round 2's kernel, vectors from global memory, any number of items.
// TRED (round 5): TAIL REDUCE -- the update-reduce without a launch of its own.  Every slot store of the launch
// is written through (sc1); when a workgroup has stored the slots of all its items, every wave waits for its stores
// (s_waitcnt vmcnt(0)), the workgroup meets at a barrier and one wave adds 1 to the arrival counter of every block-row the
// workgroup contributed to -- once per contribution: an off-diagonal item (I, J) contributes to I (row part) and to J (column
// part), a diagonal item or a row sliver to its own block-row.  Block-row R expects (nb - 1) + 1 contributions ((nb - 1) + sl_per
// where the diagonal tile is streamed as slivers).  The workgroup whose add returns the last count sums R's slots -- sc1 loads,
// IN SLOT ORDER, the code of s2_hreduce_kernel (qn_s2_slot_sum, qn_s2_hreduce_row): the arrival order decides WHO sums, never in
// what order, so the bits are those of the reduce launch -- and resets the counter.  Nobody waits for anybody: no workgroup can
// be held up by one that is not resident.  (The protocol is the first row of the hand-off table in MI355X_MICROARCH.md: one
// lane per contribution adds at agent scope behind every storing wave's wait and the workgroup's barrier; the last arriver's
// other waves load behind a barrier that the adding wave joins; every store and load of the slots is sc1.)
// What the last arriver writes is read by later launches only: u, v and g <- g+ are read by this launch's tiles of block-row R
// alone (row side of I = R, column side of J = R), and those have all stored their slots -- and loaded their vector entries long
// before -- when the counter completes.
// MEASURED (round 5, n = 4096, rocprofv3 averages of alternating runs on one box, profiles/r05_c_*): bit-identical to the reduce
// launch (tests/test_gpu_symmetric.py::test_tail_reduce_is_the_reduce_launch_bit_for_bit) and SLOWER -- the update kernel
// 23.5 -> 36.7 us for the 5.0 us launch it removes (14.9 k -> 13.4 k it/s); OFF by default (QN_OPT_TAIL_REDUCE / QN_S2_TRED=1).
// In-kernel stamps say where it goes: the hand-off itself is cheap -- a workgroup's slot stores are acknowledged 0.4 us after its
// last row, its adds have returned 0.7 us later, and spreading the counters over the memory channels changes nothing -- but the
// last workgroup's tail ends 12.7 us after its adds.  The workgroups end within 2 us of each other, every block-row has ~47
// contributors among the 256, so no block-row completes before the last handful of workgroups arrive: THE LAST WORKGROUP IS THE
// LAST ARRIVER OF ALL ITS BLOCK-ROWS (two items and a sliver: five), the one before it of most of its own, and each of them then
// reads 64 KB per block-row that other XCDs wrote -- at the 50-70 GB/s one workgroup gets on such bytes (MI355X_MICROARCH.md,
// handoff-payload) that is ~2.5 us per block-row, one after the other.  Issuing a workgroup's block-rows together would bring its
// 320 KB to ~5 us: the launch that is saved.  Who sums is decided by arrival, and arrival concentrates the sums on the workgroups
// the launch is already waiting for; a designated, polling reducer would not have that problem and cannot be had without a
// wait that a non-resident workgroup can hold up (two such launches of two solvers sharing the GPU deadlock).  The reduce launch
// stays: 5 launches per iteration.
// MEASURED AND DROPPED (round 6, profiles/r06_e_update_prefetch_ab_n4096.txt): a PRE instantiation for n = 4096's two-item lists -- the first eight rows (of every
// wave's sixteen) of the SECOND item requested in front of the workgroup's barrier, straight into 64 KB of LDS (global_load_lds_dwordx4: one instruction per 1 KB row,
// no register held), the first item's row loop refilling its window's first eight registers from there.  Until the machine has run nothing can be written,
// and behind the barrier 264 KB of stores share the fabric with the second item's loads: 64 KB of those were to move into the quiet part.  Bit-identical
// (the same loads of the same bytes), 252 registers, no scratch -- and 22.0 -> 23.0-23.3 us when requested at entry (in-kernel stamps: the machine done at 6.0 us
// instead of 4.0 -- the requests stand in the CU's queue in front of what wave 0 waits for), 22.7 when requested behind the first item's rows (the barrier then
// waits for them).  Round 2's look-ahead window in LDS went the same way.  The fabric is not idle before the barrier: the first item is still coming in.
template <bool NT, bool BFGS, bool FOLD, bool SHARD = false, bool TRED = false, bool SR1 = false> // (SR1: s u' + u s', s s' AND u u' -- sr1_b.rs)
__global__ __launch_bounds__(QN_S2_TPB, 2) void s2_hpass_kernel(const QnS2Args a) {
    static_assert(!(FOLD && SHARD), "the folded accept-reduce is a single-rank variant");
    static_assert(!(TRED && (FOLD || SHARD)), "the tail reduce is the single-rank kernel without the folded accept-reduce");
    static_assert(!(SR1 && (FOLD || SHARD || TRED || BFGS)), "SR1: the plain single-rank instance");
    __shared__ QnS2Lds L;
    __shared__ double colsum[2][QN_TB]; // [rhs]: row part of a diagonal item, parked until its column part is summed
    __shared__ double colred[QN_S2_WAVES][2][QN_TB];
    __shared__ double fq[FOLD ? QN_S2_FOLD_BLOCKS : 1][4][QN_TB]; // folded accept-reduce: the quarters of the blocks' slot sums, ...
    __shared__ double fbg[FOLD ? QN_S2_FOLD_BLOCKS : 1][2][QN_TB]; // ... b and g of the blocks, ...
    __shared__ double fv[FOLD ? QN_S2_FOLD_BLOCKS : 1][3][QN_TB]; // ... and what they become: [0] first right-hand side (y or g+), [1] g+, [2] y
    __shared__ double fred[2][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t np = (size_t)a.np;
    QN_S2_STAMP(0);
    int ij = qn_s2_first_item_of<SHARD>(a);
    const int ij1 = qn_s2_second_item(a);
    const int ij2 = (FOLD && ij1 >= 0 && a.maxk > 2) ? a.item_ij[(size_t)2 * a.G + blockIdx.x] : -1;
    int I = ij >> 16, J = ij & 0xffff;
    // Wave 0 runs the prologue; it requests nothing but its vector entries before the workgroup barrier (see s2_eval_kernel): its
    // 16 rows of the first item are fetched by waves 1..7 -- three rows each, behind their own -- and handed over in LDS.
    __shared__ v2d park0[QN_S2_RPW][64];
    // the first item's vector entries, for both settings of what the control block decides: which half of the s double buffer
    // is the pending one, and whether the first right-hand side is y (update pass) or g (direction pass)
    QnS2HVec v0;
    double s1_r, y_r;
    v2d s1_c, y_c;
    auto vec_spec = [&]() {
        const int ir = I * QN_TB + wave * QN_S2_RPW + (lane & 15), jc = J * QN_TB + qn_s2_col(I == J, lane, wave);
        QnS2HReq spec;
        spec.sp = a.F.S0; spec.up = a.F.UN; spec.r0v = a.F.GT; spec.gt = a.F.GT;
        qn_s2_hvec_load<!FOLD>(spec, ir, jc, v0);
        s1_r = a.F.S0[np + ir]; s1_c = ld2(a.F.S0 + np + jc);
        if (!FOLD) { y_r = a.F.Y[ir]; y_c = ld2(a.F.Y + jc); }
    };
    if (wave == 0) qn_s2_prologue_w0<QN_S2_HTILE, SHARD>(a, L, vec_spec);
    // Folded accept-reduce: the quarters of the slot sums of the workgroup's blocks (task = entry e of quarter qd, for all six
    // blocks), and b and g of those blocks.  None of the addresses depends on the control block, so this is done by the waves that
    // wait for wave 0; wave 0's own share (entries 0..63 of the first quarter) is dealt to waves 1..6, a block each, so that wave
    // 0 comes out of the prologue with nothing but its rows to request.  ALL of a wave's loads go out before its 16 rows do: a
    // wave's loads return in order, these are cache hits, and behind the rows they would wait for the whole first window (first
    // version, measured: the workgroup barrier moved from 5.5 to 12 us into the kernel).  Block b of the list: I of item b / 2
    // (b even), J (b odd); a missing item repeats the first block.
    int blk[QN_S2_FOLD_BLOCKS];
    {
        const int ijs[QN_S2_FOLD_ITEMS] = {ij, ij1 >= 0 ? ij1 : ij, ij2 >= 0 ? ij2 : ij};
#pragma unroll
        for (int k = 0; k < QN_S2_FOLD_ITEMS; ++k) { blk[2 * k] = ijs[k] >> 16; blk[2 * k + 1] = ijs[k] & 0xffff; }
    }
    double sv[QN_S2_FOLD_BLOCKS][8], sx[8], bg[QN_S2_FOLD_BLOCKS];
    if (FOLD && wave > 0) {
        const int per = (a.nb + 3) / 4; // (fold: nb <= 32, so a quarter is at most 8 slots)
        const int e = tid & (QN_TB - 1), qd = tid >> 7;
        const int k_lo = qd * per, k_hi = min(a.nb, k_lo + per);
#pragma unroll
        for (int b = 0; b < QN_S2_FOLD_BLOCKS; ++b) {
            const double* p = a.partE + ((size_t)blk[b] * a.nb) * QN_TB + e;
#pragma unroll
            for (int u = 0; u < 8; ++u) sv[b][u] = (k_lo + u < k_hi) ? p[(size_t)(k_lo + u) * QN_TB] : 0.0;
        }
        if (wave <= QN_S2_FOLD_BLOCKS) { // wave 0's share of block wave - 1: entries 0..63 of quarter 0
            const double* p = a.partE + ((size_t)blk[wave - 1] * a.nb) * QN_TB + lane;
#pragma unroll
            for (int u = 0; u < 8; ++u) sx[u] = (u < min(a.nb, per)) ? p[(size_t)u * QN_TB] : 0.0;
        }
        if (wave >= 4) { // (waves 4, 5: b; waves 6, 7: g -- committed by the update-reduce of the iteration before)
            const double* src = (wave >= 6) ? a.F.G : a.F.b;
#pragma unroll
            for (int b = 0; b < QN_S2_FOLD_BLOCKS; ++b) bg[b] = src[blk[b] * QN_TB + e];
        }
    }
    // (Measured and dropped, rocprofv3 averages on the same box: parking the first item in LDS as the evaluation does.  This
    // kernel's prologue consumes an evaluation that is accepted -- a short run of the machine -- and every row is written back as
    // well: 26.2 us without parking, 28.0-28.4 with.  Again at the end of round 3, with wave 0's loads behind the barrier and two
    // straight-line copies of the row loop: 22.9-23.0 against 23.1-23.8 us, and the small kernels 0.1-0.2 us slower: dropped.
    // Also without effect on this kernel: the BFGS update as s q' + q s' with q = c_su u + (c_ss / 2) s (eight VALU instructions
    // fewer per row, still bitwise symmetric), and an instance for exactly two items and a sliver.)
    // Measured and dropped, round 4: a SPECULATIVE START -- when the launch follows the accept-reduce, everything this pass needs
    // (g+, y, the pending update's s, u and coefficients) is in the control block as the previous launch left it, so waves 1..7
    // went from their requests straight into the row loop, wave 0 followed when the machine was done, no barrier in between
    // (bookkeeping: the folded accept-reduce's spec_tiles).  Bit-identical, 124 GPU tests green -- and 23.4 us against 23.7: the
    // in-kernel stamps did not move (wave 7 through the first item at 13.1 us with and without the barrier at 5.8).  This kernel is
    // bound by what a CU's memory pipeline takes: 256 KB read + 256 KB written + a sliver at ~29 KB/us is 18.3 us, which is when
    // the median workgroup ends; the barrier was never what the rows waited for.
    v2d h[QN_S2_RPW]; // the wave's 16 rows of the first item go out before the control block is known
    double* hbase = a.H + (size_t)(qn_s2_lrow<SHARD>(a, I) * QN_TB + wave * QN_S2_RPW) * np + (size_t)J * QN_TB + qn_s2_col(I == J, lane, wave);
    if (wave != 0) {
#pragma unroll
        for (int r = 0; r < QN_S2_RPW; ++r) h[r] = qn_sym_ld<NT>(hbase + (size_t)r * np);
        vec_spec();
        const double* q0 = a.H + (size_t)(qn_s2_lrow<SHARD>(a, I) * QN_TB) * np + (size_t)J * QN_TB + 2 * lane; // wave 0's rows: no clone lanes (qn_s2_col(., ., 0) = 2 lane)
        const int r0 = (wave - 1) * 3;
        v2d t3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) t3[k] = qn_sym_ld<NT>(q0 + (size_t)min(r0 + k, QN_S2_RPW - 1) * np);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (r0 + k < QN_S2_RPW) park0[r0 + k][lane] = t3[k];
    }
    if (FOLD && wave > 0) {
        const int e = tid & (QN_TB - 1), qd = tid >> 7;
#pragma unroll
        for (int b = 0; b < QN_S2_FOLD_BLOCKS; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = acc + sv[b][u];
            fq[b][qd][e] = acc;
        }
        if (wave <= QN_S2_FOLD_BLOCKS) {
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = acc + sx[u];
            fq[wave - 1][0][lane] = acc;
        }
        if (wave >= 4) {
#pragma unroll
            for (int b = 0; b < QN_S2_FOLD_BLOCKS; ++b) fbg[b][(wave - 4) >> 1][e] = bg[b];
        }
    }
    __syncthreads();
    qn_s2_ctl_out(a, L);
    const int mine = L.mine;
    if (!mine) return;
    QN_S2_STAMP(2);
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < QN_S2_RPW; ++r) h[r] = park0[r][lane];
    }
    const bool fold = FOLD && mine == 2; // (uniform)
    QnS2HReq q;
    bool nrhs2;
    {
        const QnCtl& c = L.c;
        q.pending = c.pending != 0;
        q.c_ss = c.c_ss; q.c_su = c.c_su; q.c_uu = c.c_uu;
        q.sp = a.F.S0 + (size_t)c.sc * np;
        q.up = a.F.UN;
        q.gt = a.F.GT;
        // a direction pass (bfgs.rs:47) multiplies g twice: slot 0 is what its reduce reads.  Folded: the pass that will be asked
        // for is an update pass after an evaluation at x + t d, a direction pass after one at x itself.
        nrhs2 = fold ? (c.ev_kind == QN_REQ_T) : (c.hp_nrhs == 2);
        q.r0v = nrhs2 ? a.F.Y : a.F.GT;
        if (c.sc) { v0.s_r = s1_r; v0.s_c = s1_c; }
        if (!FOLD && nrhs2) { v0.y0_r = y_r; v0.a0 = y_c; }
    }
    if (FOLD) { // quarters -> totals -> g+, y of the workgroup's blocks, in LDS
        if (tid < QN_TB) {
            if (fold) { // (uniform)
#pragma unroll
                for (int b = 0; b < QN_S2_FOLD_BLOCKS; ++b) {
                    const double qi = ((fq[b][0][tid] + fq[b][1][tid]) + fq[b][2][tid]) + fq[b][3][tid]; // (the order of qn_s2_slot_sum)
                    const double gti = qi - fbg[b][0][tid];
                    const double yi = gti - fbg[b][1][tid];
                    fv[b][0][tid] = nrhs2 ? yi : gti;
                    fv[b][1][tid] = gti;
                    fv[b][2][tid] = yi;
                }
            } else { // a plain pass (the run's machine asked for one the speculation had not covered): the vectors are in memory
#pragma unroll
                for (int b = 0; b < QN_S2_FOLD_BLOCKS; ++b) {
                    const int gi = blk[b] * QN_TB + tid;
                    fv[b][0][tid] = q.r0v[gi];
                    fv[b][1][tid] = q.gt[gi];
                }
            }
        }
        __syncthreads();
        // the first item's right-hand sides come from LDS now (blocks 0 and 1 of the list)
        const int rr = wave * QN_S2_RPW + (lane & 15), cc = qn_s2_col(I == J, lane, wave);
        v0.y0_r = fv[0][0][rr]; v0.y1_r = fv[0][1][rr];
        v0.a0 = (v2d){fv[1][0][cc], fv[1][0][cc + 1]}; v0.a1 = (v2d){fv[1][1][cc], fv[1][1][cc + 1]};
    }
    // The row loop below is branch-free.  No update pending (the first pass of a run) = an update with zero coefficients and
    // zero vectors: H + 0 (0 0) = H.  (The path requires n = n_pad: no padding entries to keep at zero.)
    const bool pend = q.pending;
    const double c_ss = pend ? q.c_ss : 0.0, c_su = pend ? q.c_su : 0.0, c_uu = pend ? q.c_uu : 0.0;
    const bool slv = !FOLD && !SHARD && a.sl_per != 0; // (uniform; the host never combines the folded accept-reduce with row slivers)
    // (the sliver's addresses are formed where they are used, behind an empty asm: as loop invariants they were held in registers
    // across the row loops, which have none to spare)
    auto sliver_ptr = [&]() {
        const QnS2Sliver sl = qn_s2_sliver(a, wave);
        unsigned r = sl.D * QN_TB + sl.row, c = sl.D * QN_TB + 2 * lane;
        asm volatile("" : "+v"(r), "+v"(c));
        return a.H + (size_t)r * np + c;
    };
    for (int it = 0;; ++it) {
        const bool diag = I == J; // (uniform)
        const double sr = pend ? v0.s_r : 0.0, ur = pend ? v0.u_r : 0.0;
        const double y0r = v0.y0_r, y1r = v0.y1_r;
        v2d sj = {0.0, 0.0}, uj = {0.0, 0.0};
        if (pend) { sj = v0.s_c; uj = v0.u_c; } // (a diagonal item's clone lanes hold the clamped column's entries: identical stores)
        v2d a0 = v0.a0, a1 = v0.a1; // update pass: y, g+ ; direction pass: g, g
        if (!qn_s2_row_on(diag, lane, wave)) { a0 = (v2d){0.0, 0.0}; a1 = (v2d){0.0, 0.0}; }
        // the next item: where the window is refilled from while this one is consumed
        int ijn = -1;
        if (it == 0) ijn = ij1;
        else if (FOLD) { if (it == 1) ijn = ij2; }
        else if (it + 1 < a.maxk) ijn = a.item_ij[(size_t)(it + 1) * a.G + blockIdx.x];
        const bool has_next = ijn >= 0; // (uniform)
        const int In = has_next ? (ijn >> 16) : I, Jn = has_next ? (ijn & 0xffff) : J;
        double* hnext = a.H + (size_t)(qn_s2_lrow<SHARD>(a, In) * QN_TB + wave * QN_S2_RPW) * np + (size_t)Jn * QN_TB + qn_s2_col(In == Jn, lane, wave);
        if (!has_next && slv) hnext = sliver_ptr(); // the workgroup's row sliver follows its last item (see qn_s2_eval_sliver)
        const size_t rstride = has_next ? np : 0; // (none left: every lane re-reads one 16-byte word -- of this item, or the sliver's row)
        double c0x = 0.0, c0y = 0.0, c1x = 0.0, c1y = 0.0;
        double racc[QN_S2_RPW];
#pragma unroll
        for (int r = 0; r < QN_S2_RPW; ++r) { // row r of the wave's 16
            v2d hn = h[r];
            h[r] = qn_sym_ld<NT>(hnext + (size_t)r * rstride); // the register this row frees takes the same row of the next item at once
            const double si = qn_lane_bcast(sr, r), ui = qn_lane_bcast(ur, r);
            if (BFGS || SR1) {
                hn.x = hn.x + c_su * (si * uj.x + ui * sj.x);
                hn.y = hn.y + c_su * (si * uj.y + ui * sj.y);
            }
            hn.x = hn.x + c_ss * (si * sj.x);
            hn.y = hn.y + c_ss * (si * sj.y);
            if (!BFGS || SR1) {
                hn.x = hn.x + c_uu * (ui * uj.x);
                hn.y = hn.y + c_uu * (ui * uj.y);
            }
            qn_sym_st<NT>(hbase + (size_t)r * np, hn);
            const double y0 = qn_lane_bcast(y0r, r), y1 = qn_lane_bcast(y1r, r);
            double t0 = hn.x * a0.x;
            t0 = __builtin_fma(hn.y, a0.y, t0);
            double t1 = hn.x * a1.x;
            t1 = __builtin_fma(hn.y, a1.y, t1);
            // first level of the 32-value butterfly (the row's sum for y against its sum for g+, lane against lane ^ 32) at
            // once: two sums become one register; QnWaveFold<16, 16> finishes the same tree after the last row
            // (as one v_permlane32_swap of the pair per half: A' + B' is the folded pair -- no selects: QnWaveFold)
            {
                const auto rl = __builtin_amdgcn_permlane32_swap(__double2loint(t0), __double2loint(t1), false, false);
                const auto rh = __builtin_amdgcn_permlane32_swap(__double2hiint(t0), __double2hiint(t1), false, false);
                racc[r] = __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
            }
            c0x = __builtin_fma(hn.x, y0, c0x);
            c0y = __builtin_fma(hn.y, y0, c0y);
            c1x = __builtin_fma(hn.x, y1, c1x);
            c1y = __builtin_fma(hn.y, y1, c1y);
        }
        if (!qn_s2_col_on(diag, lane, wave)) { c0x = 0.0; c0y = 0.0; c1x = 0.0; c1y = 0.0; }
        // the next item's vector entries go out now: they fly while this item's sums are folded, exchanged and stored
        if (has_next) {
            const int irn = In * QN_TB + wave * QN_S2_RPW + (lane & 15), ccn = qn_s2_col(In == Jn, lane, wave);
            qn_s2_hvec_load<!FOLD>(q, irn, Jn * QN_TB + ccn, v0);
            if (FOLD) { // (blocks 2 (it + 1), 2 (it + 1) + 1 of the list)
                const int rr = wave * QN_S2_RPW + (lane & 15), bI = 2 * (it + 1), bJ = bI + 1;
                v0.y0_r = fv[bI][0][rr]; v0.y1_r = fv[bI][1][rr];
                v0.a0 = (v2d){fv[bJ][0][ccn], fv[bJ][0][ccn + 1]}; v0.a1 = (v2d){fv[bJ][1][ccn], fv[bJ][1][ccn + 1]};
            }
        } else if (slv) { // the sliver's: its one row (the same entry in every lane) and block D's columns
            const QnS2Sliver sl = qn_s2_sliver(a, wave);
            unsigned r = sl.D * QN_TB + sl.row, c = sl.D * QN_TB + 2 * lane;
            asm volatile("" : "+v"(r), "+v"(c));
            qn_s2_hvec_load<true>(q, r, c, v0);
        }
        if (it == 0) { qn_keepalive(racc[0]); QN_S2_STAMP(3); QN_S2_STAMP_T(12, 448); }
        if (it == 1) { qn_keepalive(racc[0]); QN_S2_STAMP(6); QN_S2_STAMP_T(13, 448); }
        QnWaveFold<QN_S2_RPW, 16>::run(racc, lane); // lanes with (lane & 1) == 0: total of row (lane >> 1) & 15 for rhs lane >> 5
        colred[wave][0][2 * lane] = c0x;
        colred[wave][0][2 * lane + 1] = c0y;
        colred[wave][1][2 * lane] = c1x;
        colred[wave][1][2 * lane + 1] = c1y;
        if ((lane & 1) == 0) {
            const int rhs = lane >> 5, rl = wave * QN_S2_RPW + ((lane >> 1) & 15);
            if (diag) colsum[rhs][rl] = racc[0]; // the column part of a diagonal tile lands in the same slot: park the row part
            else qn_s2_slot_st<TRED>(a.part + (unsigned)(((I * a.nb + J) * 2 + rhs) * QN_TB + rl), racc[0]); // (32-bit slot offsets: nb <= 2048)
        }
        __syncthreads();
        if (tid < 2 * QN_TB) {
            const int crhs = tid / QN_TB, c = tid % QN_TB;
            double acc = colred[0][crhs][c];
#pragma unroll
            for (int w = 1; w < QN_S2_WAVES; ++w) acc = acc + colred[w][crhs][c];
            if (diag) qn_s2_slot_st<TRED>(a.part + (unsigned)(((I * a.nb + I) * 2 + crhs) * QN_TB + c), colsum[crhs][c] + acc); // row part + column part
            else qn_s2_slot_st<TRED>(a.part + (unsigned)(((J * a.nb + I) * 2 + crhs) * QN_TB + c), acc);
        }
        if (it == 0) QN_S2_STAMP(5);
        if (!has_next) break;
        I = In; J = Jn; hbase = hnext;
        __syncthreads(); // the LDS staging areas are rewritten by the next item
    }
    if (slv) { // row sl.row of the diagonal tile (D, D), all 128 columns: the update in place, and the row's two sums (no column part)
        v2d hn = h[0]; // (the window holds the row sixteen times)
        const double si = pend ? v0.s_r : 0.0, ui = pend ? v0.u_r : 0.0;
        v2d sj = {0.0, 0.0}, uj = {0.0, 0.0};
        if (pend) { sj = v0.s_c; uj = v0.u_c; }
        if (BFGS || SR1) {
            hn.x = hn.x + c_su * (si * uj.x + ui * sj.x);
            hn.y = hn.y + c_su * (si * uj.y + ui * sj.y);
        }
        hn.x = hn.x + c_ss * (si * sj.x);
        hn.y = hn.y + c_ss * (si * sj.y);
        if (!BFGS || SR1) {
            hn.x = hn.x + c_uu * (ui * uj.x);
            hn.y = hn.y + c_uu * (ui * uj.y);
        }
        qn_sym_st<NT>(sliver_ptr(), hn);
        double t0 = hn.x * v0.a0.x;
        t0 = __builtin_fma(hn.y, v0.a0.y, t0);
        double t1 = hn.x * v0.a1.x;
        t1 = __builtin_fma(hn.y, v0.a1.y, t1);
        double tt[2] = {t0, t1};
        QnWaveFold<2, 32>::run(tt, lane); // lane 32 k: the total of right-hand side k
        const QnS2Sliver sl = qn_s2_sliver(a, wave);
        if ((lane & 31) == 0) qn_s2_slot_st<TRED>(a.part + (unsigned)(((sl.D * a.nb + sl.D) * 2 + (lane >> 5)) * QN_TB + sl.row), tt[0]);
    }
    QN_S2_STAMP(15);
    if (TRED) {
        __shared__ int todo[2 * QN_S2_TRED_MAXK + 2]; // [0]: how many block-rows this workgroup completed; [1 ..]: which
        __shared__ double qbuf[3][QN_TB];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every wave: its slot stores have been acknowledged ...
        __syncthreads();                                  // ... before the workgroup's adds
        QN_S2_STAMP(7);
        if (wave == 0) {
            // contribution `lane`: item lane >> 1, its I (even lanes) or J (odd lanes; a diagonal item has one); lane 2 maxk: the sliver
            const int k = lane >> 1;
            int R = -1;
            if (k < a.maxk) {
                const int ijk = k == 0 ? ij : (k == 1 ? ij1 : a.item_ij[(size_t)k * a.G + blockIdx.x]);
                if (ijk >= 0) {
                    const int Ik = ijk >> 16, Jk = ijk & 0xffff;
                    R = (lane & 1) ? (Jk != Ik ? Jk : -1) : Ik;
                }
            } else if (lane == 2 * a.maxk && slv) R = qn_s2_sliver(a, 0).D;
            bool last = false;
            if (R >= 0) {
                const int expect = (a.nb - 1) + ((a.sl_per != 0 && R >= a.sl_first) ? a.sl_per : 1);
                int* cp = a.cnt + (size_t)R * a.cnt_stride;
                last = __hip_atomic_fetch_add(cp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == expect;
                if (last) __hip_atomic_store(cp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (nobody adds to it again before the next launch)
            }
            const unsigned long long m = __ballot(last);
            if (last) todo[1 + __popcll(m & ((1ull << lane) - 1ull))] = R;
            if (lane == 0) todo[0] = __popcll(m);
        }
        __syncthreads(); // (the adding wave joins: the other waves' loads come behind its returned adds)
        QN_S2_STAMP(8);
        const int ntodo = todo[0];
        const int nrhs = L.c.hp_nrhs;
        for (int t = 0; t < ntodo; ++t) { // (uniform)
            const int R = todo[1 + t];
            QnS2Slots S0, S1;
            qn_s2_slot_issue<2, true>(a.part, a.nb, R, 0, S0);
            qn_s2_slot_issue<2, true>(a.part, a.nb, R, 1, S1);
            const double tot0 = qn_s2_slot_sum<2, true>(a.part, a.nb, R, 0, S0, qbuf);
            const double tot1 = (nrhs == 2) ? qn_s2_slot_sum<2, true>(a.part, a.nb, R, 1, S1, qbuf) : 0.0; // (uniform)
            double gp = 0.0, yv = 0.0;
            if (tid < QN_TB) { gp = a.F.GT[R * QN_TB + tid]; yv = a.F.Y[R * QN_TB + tid]; }
            qn_s2_hreduce_row(a, R, nrhs, tot0, tot1, fred, gp, yv);
            __syncthreads(); // fred is reused
        }
        QN_S2_STAMP(14);
        return;
    }
    if (!fold) return; // (uniform)
    // Folded accept-reduce, the owner's part: the workgroup that holds the diagonal item (R, R) writes block R of the vectors
    // (g+, y, x+, s = x+ - x: bfgs.rs:94-99) and block R's five sums -- after its tiles, because nothing in this launch reads them
    // (every workgroup has formed the entries it needs itself) and the tiles are what the launch is waiting for.
    const QnEvalReq qe = qn_s2_eval_req<false>(L.c, true);
    const double* __restrict__ x = a.F.X0 + (size_t)qe.xc * np;
    double* __restrict__ xt = a.F.X0 + (size_t)(1 - qe.xc) * np;
    const double* __restrict__ spv = a.F.S0 + (size_t)qe.sc * np;
    double* __restrict__ sstage = a.F.S0 + (size_t)(1 - qe.sc) * np;
#pragma unroll 1
    for (int k = 0; k < QN_S2_FOLD_ITEMS; ++k) {
        const int ij_k = k == 0 ? ij : (k == 1 ? ij1 : ij2);
        if (ij_k < 0 || (ij_k >> 16) != (ij_k & 0xffff)) continue; // (uniform)
        const int R = ij_k >> 16;
        __syncthreads(); // fred is reused
        if (tid < QN_TB) {
            const int gi = R * QN_TB + tid;
            const double gti = fv[2 * k][1][tid], yi = fv[2 * k][2][tid];
            double di;
            const double xi = x[gi];
            const double xti = qn_s2_trial(qe, xi, a.F.VV[gi], spv[gi], a.F.UN[gi], di);
            const double si = xti - xi; // s = x+ - x, not t d (bfgs.rs:96)
            a.F.GT[gi] = gti;
            a.F.Y[gi] = yi;
            xt[gi] = xti;
            sstage[gi] = si;
            double p[8] = {yi * yi, yi * si, gti * gti, si * si, si * gti, 0.0, 0.0, 0.0};
            QnWaveFold<8, 32>::run(p, lane);
            if ((lane & 7) == 0) fred[wave][lane >> 3] = p[0]; // (threads 0..127 = waves 0, 1)
        }
        __syncthreads();
        if (tid < QN_S2_NR) a.wgS[((size_t)a.parity * QN_S2_ROW + tid) * a.trows + R] = fred[0][tid] + fred[1][tid]; // (as s2_vec_kernel)
    }
}

// update-reduce: u_i, v_i = sums of block-row R's slots, the partials of y'u and u'g+ (the update's coefficients and the next
// direction need them); commits g <- g+ (the evaluation kernels read g for g'd)
// Single rank: TWO workgroups per block-row, one per right-hand side (grid 2 nb).  A workgroup's slots were 2 x nb KB behind one CU's
// memory pipeline -- in-kernel stamps at n = 4096: control block 1.45 us after entry, barrier 2.0, totals 2.75: the slots, not the machine, were
// what the epilogue waited for -- and nothing in the epilogue needs both totals: u goes with the two sums and the commit of g, v is stored and
// that is all.  The same sums in the same order.
template <bool SHARD = false, bool SR1 = false, int NBT = 0> // (row-sharded: the totals are the rank-order sums of the gathered partial [u, v]: see s2_vec_kernel; NBT: as there)
__global__ __launch_bounds__(QN_S2_TPB) void s2_hreduce_kernel(const QnS2Args a) {
    static_assert(!(SHARD && SR1), "SR1 runs on one rank");
    __shared__ QnS2Lds L;
    __shared__ double qbuf[3][QN_TB];
    __shared__ double bred[2][8];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int R = SHARD ? (int)blockIdx.x : (int)(blockIdx.x >> 1), half = SHARD ? -1 : (int)(blockIdx.x & 1);
    if (NBT > 0 && __builtin_expect((int)blockIdx.x >= 2 * NBT, 0)) { // TOUCH workgroup b (uniform): an evaluation follows -- Q, the tile s2_evalr_kernel's workgroup b
        const int b = (int)blockIdx.x - 2 * NBT;                       // streams FIRST in a launch of the next parity (qn_sym2r.hip.h, ZIG-ZAG)
        const int ij0 = qn_s2_item_of_index(b, NBT), ij1 = qn_s2_item_of_index(a.G + b, NBT);
        const bool flip = a.zig != 0 && ((a.parity ^ 1) & 1) != 0 && (ij1 >> 16) != (ij1 & 0xffff);
        for (int k = 0; k < a.touchq_delay; k += 8) __builtin_amdgcn_s_sleep(8);
        qn_s2_touch(a.Q, a, flip ? ij1 : ij0, a.touchq);
        return;
    }
    QnS2Slots S0;
    QN_S2_STAMP(0);
    double gp = 0.0, yv = 0.0, s0e = 0.0, s1e = 0.0;
    auto entries = [&]() {
        if (half != 1 && tid < QN_TB) {
            gp = a.F.GT[R * QN_TB + tid]; yv = a.F.Y[R * QN_TB + tid];
            if (SR1) { s0e = a.F.S0[R * QN_TB + tid]; s1e = a.F.S0[(size_t)a.np + R * QN_TB + tid]; } // (the staged s, either half)
        }
    };
    if (SHARD) { if (wave == 0) qn_s2_prologue_w0<QN_S2_HREDUCE, true>(a, L, entries); else entries(); }
    else if (wave == 0) qn_s2_prologue_w0<QN_S2_HREDUCE>(a, L, [&]() { qn_s2_slot_issue(a.part, a.nb, R, half, S0); entries(); });
    else { qn_s2_slot_issue(a.part, a.nb, R, half, S0); entries(); }
    __syncthreads();
    QN_S2_STAMP(2);
    qn_s2_ctl_out(a, L);
    if (!L.mine) return;
    const int nrhs = L.c.hp_nrhs;
    double tot0 = 0.0, tot1 = 0.0;
    if (!SHARD) {
        if (half == 1 && nrhs != 2) return; // (uniform) a direction pass has one right-hand side
        const double t = qn_s2_slot_sum(a.part, a.nb, R, half, S0, qbuf);
        if (half == 0) tot0 = t; else tot1 = t;
    } else if (tid < QN_TB) { // xg: [rank][rhs][np]
        const size_t np = (size_t)a.np;
        const double* xp = a.xg + (size_t)R * QN_TB + tid;
        tot0 = xp[0]; tot1 = xp[np];
        for (int r = 1; r < a.sh_nsum; ++r) { tot0 = tot0 + xp[(size_t)r * 2 * np]; tot1 = tot1 + xp[(size_t)r * 2 * np + np]; }
        if (nrhs != 2) tot1 = 0.0;
    }
    QN_S2_STAMP(13); // (the totals)
    qn_s2_hreduce_row<SR1>(a, R, nrhs, tot0, tot1, bred, gp, yv, half, L.c.sc ? s0e : s1e); // (the STAGED s: the half the pending one does not occupy)
    QN_S2_STAMP(12);
}

// PLACEMENT PROBE (round 4).  At n = 4096 about one inverse Hessian in eight runs the update kernel at 29.7 us instead of 25.2 for
// as long as it lives.  What the measurements say:
//   * it is a property of the ALLOCATION, not of the process (tools/slow_mode_probe.py: several solvers in one process, each with
//     an H of its own -- one slow one among fast ones, while the evaluation kernel on the process's one Q never changes);
//   * it is decided behind the L2 (tools/slow_mode_pmc.sh, profiles/r04_c_slow_mode_pmc_per_solver.txt): a slow H has no
//     translation misses (TCP_UTCL1_TRANSLATION_MISS = 0 in both modes), the same requests to the fabric (TCC_EA0_RDREQ / WRREQ
//     equal to 0.02 %) and the same L2 hit count to 2 % -- the same 70 MB of reads and 70 MB of writes are served more slowly by
//     what lies behind the fabric, and no counter of this tool chain looks there;
//   * it is an INTERACTION: a pass over H alone runs equally fast on a slow and on a fast allocation (first version of this
//     probe: 21.8 us on every candidate, and slow solvers all the same).  The Infinity Cache holds H's half and Q's half side by
//     side (2 x 67 MB of 256); where H sits relative to Q decides how much of H the two evaluations of an iteration leave there.
// So the library measures the pattern itself: with the objective in hand (the first run on the fast path) it times the update
// pass's bytes -- every workgroup its first two list items of H, read into the 16-row window and written back -- BEHIND two reads
// of Q's half in the same shape, on the solver's H and on a second allocation, a third one when they differ, and moves H to the
// best (qn_hip.hip, place_h).  (What place_h times today is the real thing: the update kernel itself behind two real evaluation
// launches -- on H alone the kernel measured fast on allocations that were slow in the run; tools/modes_ab.sh, profiles/r04_s_*.)
template <bool WRITE>
__global__ __launch_bounds__(QN_S2_TPB, 2) void s2_place_probe_kernel(double* __restrict__ M, const int nb, const int np_, const int G, double* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t np = (size_t)np_;
    const int nitems = nb * (nb + 1) / 2;
    double acc = 0.0;
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
        const int t = k * G + (int)blockIdx.x;
        if (t >= nitems) break;
        const int ij = qn_s2_item_of_index(t, nb);
        const int I = ij >> 16, J = ij & 0xffff;
        double* hb = M + (size_t)(I * QN_TB + wave * QN_S2_RPW) * np + (size_t)J * QN_TB + 2 * lane;
        v2d h[QN_S2_RPW];
#pragma unroll
        for (int r = 0; r < QN_S2_RPW; ++r) h[r] = ld2(hb + (size_t)r * np);
#pragma unroll
        for (int r = 0; r < QN_S2_RPW; ++r) {
            if (WRITE) {
                asm volatile("" : "+v"(h[r])); // (opaque: a store of the value just loaded from the same address is a store the compiler may drop)
                st2(hb + (size_t)r * np, h[r]);
            } else acc += h[r].x + h[r].y;
        }
    }
    if (!WRITE && acc == 12345.678) sink[blockIdx.x] = acc; // (keeps the loads)
}

// synchronous mode: the prologue alone (one workgroup)
// (GOBJ / KIND: generic objectives -- the prologue-only launch in front of an evaluation's or an update pass's many-workgroup kernels)
template <bool SHARD = false, bool GOBJ = false, int KIND = QN_S2_ADVANCE, bool BND = false>
__global__ __launch_bounds__(128) void s2_advance_kernel(const QnS2Args a) {
    __shared__ QnS2Lds L;
    if (threadIdx.x < 64) qn_s2_prologue_w0<KIND, SHARD, QnS2NoEarly, GOBJ, BND>(a, L);
    __syncthreads();
    qn_s2_ctl_out(a, L);
}

// BOUNDED VARIANTS (round 5): BFGSB / DFPB and MoreThuenteB in this structure.
// What a bounded run does differently from the unbounded one (bfgs_b.rs:66-77, morethuente_b.rs:185-201): its direction is
// P(x - H g) - x, not -H g, and MoreThuenteB clips t_max by the step at which x + t d leaves its box -- a minimum over the entries --
// before it starts; after that it IS More-Thuente.  Both are O(n) work on the direction, which this path otherwise never stores (the
// evaluations form -H+ g+ on the fly from v, s, u and five scalars).  So a bounded run gets ONE more launch per iteration, this one, in front
// of its evaluations: its prologue runs the machine up to the new direction (QN_PH_REQ_DIR, qn_ctl_step.hip.h LEAN == 2), block-row R forms
// the lazy direction at its 128 entries, projects it, stores VV <- -d -- from here on the direction is a STORED one, which is what
// dir_mode 0 means to every other kernel: they run unchanged -- and leaves the block-row's minimal step to the box in the table; the next
// prologue folds that column with fmin and clips t_max.  (The generic path did this in its one-workgroup control kernel, which streams
// 1 MB of n-vectors per iteration through ONE CU: 39 of a bounded iteration's 95 us at n = 4096, tools/ctl_stamps_bounded.py.)
__global__ __launch_bounds__(QN_TB) void s2_dir_kernel(const QnS2Args a) {
    __shared__ QnS2Lds L;
    __shared__ double red[2];
    const int R = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gi = R * QN_TB + tid;
    const size_t np = (size_t)a.np;
    // (requested before the machine runs, for both settings of the two buffer toggles: see s2_vec_kernel)
    double x0 = 0.0, x1 = 0.0, s0 = 0.0, s1 = 0.0, vv = 0.0, un = 0.0, lbi = -INFINITY, ubi = INFINITY, llbi = -INFINITY, lubi = INFINITY;
    auto entries = [&]() {
        x0 = a.F.X0[gi]; x1 = a.F.X0[np + gi]; s0 = a.F.S0[gi]; s1 = a.F.S0[np + gi]; vv = a.F.VV[gi]; un = a.F.UN[gi];
        if (a.lb) { lbi = a.lb[gi]; ubi = a.ub[gi]; }
        if (a.llb) { llbi = a.llb[gi]; lubi = a.lub[gi]; }
    };
    if (wave == 0) qn_s2_prologue_w0<QN_S2_DIR, false, decltype(entries)&, false, true>(a, L, entries);
    else entries();
    __syncthreads();
    qn_s2_ctl_out(a, L);
    if (!L.mine) return;
    const QnCtl& c = L.c;
    const double xi = c.xc ? x1 : x0, si = c.sc ? s1 : s0;
    // the lazy direction as the evaluations would have formed it (qn_s2_trial, qn_s2_eval_req) ...
    const double al = c.c_su * c.dir_ug + c.c_ss * c.dir_sg, be = c.c_su * c.dir_sg + c.c_uu * c.dir_ug;
    const double w = c.dir_mode ? (vv + __builtin_fma(be, un, al * si)) : vv; // H g
    double di = -w;
    if (c.s2_dir & 1) { double t = xi - w; t = fmin(fmax(t, lbi), ubi); di = t - xi; } // bfgs_b.rs:72-75: P(x - H g) - x
    a.F.VV[gi] = -di;
    double cand = INFINITY; // morethuente_b.rs:185-198
    if (c.s2_dir & 2) { if (di > 0.0) cand = (lubi - xi) / di; else if (di < 0.0) cand = (llbi - xi) / di; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cand = fmin(cand, __shfl_xor(cand, off, 64));
    if (lane == 0) red[wave] = cand;
    __syncthreads();
    if (tid == 0) a.wgS[((size_t)a.parity * QN_S2_ROW + 0) * a.trows + R] = fmin(red[0], red[1]);
}

// BACKTRACKINGB ON THIS PATH (round 6; backtracking_b.rs:52-88).  Its trial point is P(x + t d) -- projected onto the LINE SEARCH's box --
// and its Armijo rule reads ||P(x + t d) - x||^2 / t instead of g'd (:24-34).  The evaluation kernels form x + t d on the fly and know nothing of boxes;
// so a projected trial takes ONE more launch of nb workgroups, this one, in front of its evaluation: block-row R forms the trial point at its 128
// entries exactly as the evaluations would (qn_s2_trial on the lazy or the stored direction), projects it, STORES it in the trial half of the x
// double buffer (the half the accept-reduce overwrites with x+ later) and leaves the block-row's share of ||P(x + t d) - x||^2 in the table.  The
// evaluation launch behind it finds the request in service state 5: its prologue adds the shares up (QnCtl.bt_diff2) and the kernel evaluates AT
// the stored point (qn_s2_eval_req<., true>: the other half of X0, no step) -- tiles, slots, scalars as for any point.  The accepted step's
// x + t d is NOT the projected point (bfgs_b.rs:91-98 evaluates the oracle at x + step * direction): it gets the ordinary evaluation, with
// vectors, that every iteration ends with.
__global__ __launch_bounds__(QN_TB) void s2_proj_kernel(const QnS2Args a) {
    __shared__ QnS2Lds L;
    __shared__ double red[2];
    const int R = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gi = R * QN_TB + tid;
    const size_t np = (size_t)a.np;
    double x0 = 0.0, x1 = 0.0, s0 = 0.0, s1 = 0.0, vv = 0.0, un = 0.0, lo = -INFINITY, hi = INFINITY;
    auto entries = [&]() { // (requested before the machine runs, for both settings of the two buffer toggles: see s2_vec_kernel)
        x0 = a.F.X0[gi]; x1 = a.F.X0[np + gi]; s0 = a.F.S0[gi]; s1 = a.F.S0[np + gi]; vv = a.F.VV[gi]; un = a.F.UN[gi];
        if (a.llb) { lo = a.llb[gi]; hi = a.lub[gi]; }
    };
    if (wave == 0) qn_s2_prologue_w0<QN_S2_PROJ, false, decltype(entries)&, false, true>(a, L, entries);
    else entries();
    __syncthreads();
    qn_s2_ctl_out(a, L);
    if (!L.mine) return;
    const QnEvalReq q = qn_s2_eval_req<false>(L.c, false); // (the request as an evaluation would decode it: x + t d)
    const double xi = q.xc ? x1 : x0, si = q.sc ? s1 : s0;
    double d;
    double z = qn_s2_trial(q, xi, vv, si, un, d);
    z = fmin(fmax(z, lo), hi); // x_kp1.box_projection(&self.lower_bound, &self.upper_bound), backtracking_b.rs:67
    a.F.X0[(size_t)(1 - q.xc) * np + gi] = z;
    const double df = z - xi; // diff = x - x0, :31
    double p = qn_wave_sum(df * df);
    if (lane == 0) red[wave] = p;
    __syncthreads();
    if (tid == 0) a.wgS[((size_t)a.parity * QN_S2_ROW + 0) * a.trows + R] = red[0] + red[1];
}

// lower triangle <- transpose of the maintained upper one, 32 x 32 blocks through LDS; inside the diagonal blocks too
__global__ __launch_bounds__(256) void sym2_mirror_kernel(double* __restrict__ H, int n_pad) {
    __shared__ double t[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t np = (size_t)n_pad;
    for (int r = ty; r < 32; r += 8) t[r][tx] = H[(size_t)(bi * 32 + r) * np + bj * 32 + tx];
    __syncthreads();
    if (bj > bi) {
        for (int r = ty; r < 32; r += 8) H[(size_t)(bj * 32 + r) * np + bi * 32 + tx] = t[tx][r];
    } else {
        for (int r = ty; r < 32; r += 8)
            if (tx < r) H[(size_t)(bi * 32 + r) * np + bi * 32 + tx] = t[tx][r];
    }
}
