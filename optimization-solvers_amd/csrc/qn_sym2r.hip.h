// qn_sym2r.hip.h -- the evaluation tiles at n = 4096 with MOVER waves and MULTIPLIER waves (round 6).
//
// What rounds 3-5 measured on s2_eval_kernel<true> (two items and a sliver per workgroup, 256 workgroups; profiles/r05_v_*): the CU takes its
// 264 KB in at ~31 KB/us whatever is requested when, and a wave that is held at a request cannot multiply.  So the eight waves that stream were
// busy requesting until 8.1 us after entry, the workgroup barrier fell at 8.5 us, and ONLY THEN were the two items multiplied, folded and
// exchanged: 4.4 us of arithmetic with no byte in flight, behind 8.5 us of bytes with no arithmetic.  The state machine (done 5.6 us after
// entry) was never the limit.
//
// Here the two jobs belong to different waves of one 16-wave workgroup (1024 threads, <= 128 registers):
//   * waves 0..7 MOVE.  They are the old kernel's waves up to its barrier: wave 0 runs the prologue (the solver's state machine), waves 1..7
//     request the first item into their 16-row register windows, park every row in LDS the moment it lands and refill the register with the
//     same row of the second item.  They never multiply: when the second item's rows (and the sliver's) have landed they go into the park as
//     well, into the slots the multipliers have freed.
//   * waves 8..15 MULTIPLY.  Wave 8 + w does, instruction for instruction, what wave w of the old kernel did behind its barrier
//     (qn_s2_eval_item<FROM_PARK>, the sliver, the group's fold, the slot stores): the same sums in the same order, hence the SAME BITS as
//     s2_eval_kernel<true> (tests/test_gpu_symmetric.py::test_ring_evaluation_is_the_pair_instance_bit_for_bit).  It starts when the machine
//     has decided (a flag in LDS, not a barrier: the movers are still requesting) and takes each row as soon as its mover has parked it.
//   The park is a ring of 16 slots per mover: row k of a mover's sequence (first item 0..15, second item 16..31, sliver row 32) lives in slot
//   k & 15; prod[w] counts the rows mover w has parked, cons[w] the rows multiplier w has read -- a mover writes row k only when
//   cons[w] >= k - 15.  Plain LDS words: a wave's LDS accesses execute in order, the producer waits for its data writes (lgkmcnt(0)) before it
//   publishes the count, the consumer reads the count before the data.  Every spin is bounded; a spin that runs out poisons the workgroup's
//   sums with NaN (the run then ends with QN_ERROR_OUT_OF_DOMAIN instead of hanging).
// The first item's arithmetic now runs UNDER the second item's bytes, and the second item is multiplied by waves that were idle until then.
#pragma once

#define QN_S2R_TPB 1024
#ifndef QN_S2R_HEAD
#define QN_S2R_HEAD 4 // first-item rows a mover requests in front of the entry barrier (behind wave 0's three)
#endif
#define QN_S2R_SPIN_MAX (1 << 18) // polls of an LDS word (~150 cycles each with the s_sleep) before the workgroup gives up

struct QnS2RSync {
    unsigned prod[QN_S2_WAVES]; // rows of its sequence mover w has parked (wave 0's first-item rows are parked by the others: cnt0)
    unsigned adone[QN_S2_WAVES]; // multiplier w has finished the first item: its lanes' shares of the two scalar sums lie in park[w][0]
    unsigned cnt0;              // rows of wave 0's share of the first item parked so far (16 when complete)
    unsigned eready;            // blocks of the trial point staged so far (5 when complete)
    unsigned mdone;             // the machine has run: L.c and L.mine are final
    unsigned bad;               // a bounded spin ran out in some wave of this workgroup
};

__device__ __forceinline__ unsigned qn_s2r_peek(const unsigned* p) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// spin until *p >= want (uniform); returns the value read
template <int SLEEP = 1> // (s_sleep units of 64 clocks between two looks)
__device__ __forceinline__ unsigned qn_s2r_wait_ge(const unsigned* p, const unsigned want, bool& bad) {
    unsigned v = qn_s2r_peek(p);
    for (int spin = 0; v < want && spin < QN_S2R_SPIN_MAX; ++spin) {
        __builtin_amdgcn_s_sleep(SLEEP);
        v = qn_s2r_peek(p);
    }
    if (v < want) { bad = true; v = want; }
    asm volatile("" ::: "memory"); // nothing that follows is read before the count
    return v;
}
// publish: every LDS access this wave has issued is complete before the word changes
template <bool WAIT = true>
__device__ __forceinline__ void qn_s2r_publish(unsigned* p, const unsigned v) {
    if (WAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("" ::: "memory"); // (a wave's LDS instructions execute in the order it issued them: the count lands behind the rows it counts)
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void qn_s2r_publish_add(unsigned* p, const unsigned v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0 && v) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct QnS2RShared {
    QnS2Lds L;
    QnS2RSync Y;
#ifdef QN_S2_STAMPS
    unsigned long long stamps[16]; // diagnostic build: the stamps are kept in LDS and stored behind the exchange (a global store per stamp queues behind
                                   // the movers' requests, and the state machine's next wait for memory then waits for the stamp)
#endif
    double sred[QN_S2_WAVES][8];
    double colsum[QN_TB];                 // (the second item's, when it is a diagonal tile: the first never is)
    double colred[2][QN_S2_WAVES][QN_TB]; // [item of the pair]
    // THE TRIAL POINT at the workgroup's five blocks (qn_s2_trial's xt = x + t d and d, entry by entry), formed ONCE per workgroup by multipliers
    // 0..4 from entries they requested at kernel entry -- 40 load instructions per workgroup.  (First version: every multiplier loaded its own
    // entries, ~28 instructions each: requested at entry they delayed the control block and the movers' first rows by 1.3 us -- a load instruction
    // costs the CU's address pipeline 16 clocks whatever it fetches --, requested later they queued behind the tile stream for 5 us.)
    double xa_r[QN_TB], da_r[QN_TB];                         // first item: rows (block I)
    double xa_c[QN_TB], da_c[QN_TB];                         // ... columns (block J)
    double xb_r[QN_TB], db_r[QN_TB], bb_r[QN_TB], gb_r[QN_TB]; // second item: rows, with b and g (a diagonal tile needs them)
    double xb_c[QN_TB], db_c[QN_TB];
    double xs_c[QN_TB];                                      // the sliver's diagonal block: columns ...
    double ss[8][4];                                         // ... and its eight rows: xt, d, b, g
    double pjred[2];                                         // (BND, a.projfold) block blockIdx.x's share of ||P(x + t d) - x||^2: multipliers 6 and 7
    v2d park[QN_S2_WAVES][QN_S2_RPW][64]; // 128 KB: sixteen ring slots per mover
};

// One item from the park against the STAGED trial point: qn_s2_eval_item<FROM_PARK, ., ., EARLY> (qn_sym2.hip.h) behind its first five lines --
// the same products, sums and exchanges in the same order -- with xt and d read instead of formed.
struct QnS2RTrial {
    double xr, dr, b_r, g_r; // row 16 w + (lane & 15) of block I
    v2d xtj, dj;             // this lane's two columns of block J
};
struct QnS2RNoWait { __device__ __forceinline__ void operator()(int) const {} };
struct QnS2RSums { double pf, pg, p4, p5; }; // an item's share of x'(Q xt - 2b), d'(Q xt - b), g'd, #non-finite d on this lane
// PARK: the rows come from the park (a multiplier; wait_row(r) is called in front of rows r .. r + 3, r = 0, 4, 8, 12); otherwise from the
// register window h (a mover: its own sixteen rows of the second item), whose first register takes the sliver's row once its row is consumed.
template <bool PARK, class WaitRow>
__device__ __forceinline__ double qn_s2r_item(const QnS2RTrial& v, const bool diag, const int lane, const int wave, const v2d* parkw, v2d (&h)[QN_S2_RPW],
                                              const double* __restrict__ refill0, double* __restrict__ colred_w, QnS2RSums& o, WaitRow&& wait_row) {
    const double xr = v.xr, dr = v.dr;
    double p4 = 0.0, p5 = 0.0; // diagonal items: g'd, #non-finite d over block I (lanes 0..15 of every wave)
    if (diag && lane < 16) { p4 = v.g_r * dr; p5 = isfinite(dr) ? 0.0 : 1.0; }
    v2d xtj = v.xtj, dj = v.dj;
    if (!qn_s2_row_on(diag, lane, wave)) { xtj = (v2d){0.0, 0.0}; dj = (v2d){0.0, 0.0}; }
    double cx = 0.0, cy = 0.0;
    double racc[QN_S2_RPW / 2];
    v2d hq[8]; // PARK: a rolling window of eight rows -- while rows r .. r + 3 are multiplied, rows r + 4 .. r + 7 are on their way out of the park
#pragma unroll
    for (int r = 0; r < QN_S2_RPW; ++r) {
        v2d hv;
        if (PARK) {
            wait_row(r); // (at r = 0, 4, 8, 12: the hook has seen rows up to r + 7 parked)
            if ((r & 3) == 0) {
                if (r == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) hq[k] = parkw[k * 64 + lane];
                }
                if (r + 4 < QN_S2_RPW) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) hq[(r + 4 + k) & 7] = parkw[(r + 4 + k) * 64 + lane];
                }
            }
            hv = hq[r & 7];
        } else {
            hv = h[r];
            if (r == 0) h[0] = ld2(refill0);
        }
        const double xi = qn_lane_bcast(xr, r);
        double t0 = hv.x * xtj.x;
        t0 = __builtin_fma(hv.y, xtj.y, t0);
        if (r >= QN_S2_RPW / 2) { // (QnWaveFold<16, 32>'s first level on the pair (r - 8, r): the swap form, as it is used there)
            const double lo = racc[r - QN_S2_RPW / 2];
            const auto rl = __builtin_amdgcn_permlane32_swap(__double2loint(lo), __double2loint(t0), false, false);
            const auto rh = __builtin_amdgcn_permlane32_swap(__double2hiint(lo), __double2hiint(t0), false, false);
            racc[r - QN_S2_RPW / 2] = __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
        } else racc[r] = t0;
        cx = __builtin_fma(hv.x, xi, cx);
        cy = __builtin_fma(hv.y, xi, cy);
        // (a use per row: without it the compiler sinks every row's arithmetic below the last wait -- eight rows held in registers, and nothing
        // multiplied while the next row is on its way)
        if (PARK) { qn_keepalive(cx); qn_keepalive(cy); qn_keepalive(racc[r & (QN_S2_RPW / 2 - 1)]); }
    }
    if (!qn_s2_col_on(diag, lane, wave)) { cx = 0.0; cy = 0.0; }
    colred_w[2 * lane] = cx;
    colred_w[2 * lane + 1] = cy;
    QnWaveFold<QN_S2_RPW / 2, 16, true>::run(racc, lane); // the rest of the sixteen-value tree: lanes with (lane & 3) == 0 hold the total of row lane >> 2
    double pf = xtj.x * cx, pg = dj.x * cx;
    pf = __builtin_fma(xtj.y, cy, pf);
    pg = __builtin_fma(dj.y, cy, pg);
    {
        const int rq = lane >> 2;
        const double xq = __shfl(xr, rq), dq = __shfl(dr, rq);
        const double bq = diag ? __shfl(v.b_r, rq) : 0.0;
        if ((lane & 3) == 0) {
            pf = __builtin_fma(xq, racc[0] - (bq + bq), pf);
            pg = __builtin_fma(dq, racc[0] - bq, pg);
        }
    }
    o.pf = pf; o.pg = pg; o.p4 = p4; o.p5 = p5;
    return racc[0];
}

#ifdef QN_S2_STAMPS
#define QN_S2R_STAMP(k, t) do { if (ltid == (t)) SH.stamps[k] = wall_clock64(); } while (0)
#else
#define QN_S2R_STAMP(k, t) do { } while (0)
#endif

// BND: behind the bounded runs' prologue (their machine and QN_PH_REQ_DIR), as s2_eval_kernel<.., BND>
template <bool BND = false>
__global__ __launch_bounds__(QN_S2R_TPB) void s2_evalr_kernel(const QnS2Args a) {
    // ONE block of LDS with the control block FIRST: the state machine reads and writes ~150 fields of it, and a DS instruction's offset field
    // reaches 64 KB -- behind the 128 KB park every field had its address materialised in a vector register of its own (dozens of them, hoisted
    // out of the machine's loop: the kernel spilled at its 128-register budget)
    __shared__ QnS2RShared SH;
    QnS2Lds& L = SH.L;
    QnS2RSync& Y = SH.Y;
    auto& colsum = SH.colsum;
    auto& colred = SH.colred;
    auto& sred = SH.sred;
    auto& park = SH.park;
    const int tid = threadIdx.x, lane = tid & 63;
    // (the wave index as a SCALAR: with `tid >> 6` the compiler takes every `if (wave ...)` for a divergent branch, lays both sides out in one
    // stream and keeps the movers' 16-row windows allocated across wave 0's run of the state machine -- 180 registers instead of 88 + 80)
    // ROLES BY A CHECKERBOARD over (wave / 4, wave % 4): whichever of the two the hardware takes for a wave's SIMD, every SIMD gets two movers and two
    // multipliers.  (First version: waves 0..7 moved, 8..15 multiplied -- and in-kernel stamps had the multipliers of ONE workgroup finish the same
    // sixteen rows 8.3, 9.9 and 11.3 us after entry: four of them on one SIMD, behind one another.)  `wave` below is the LOGICAL index: 0..7 movers
    // (0 = physical wave 0: the prologue addresses its lanes by threadIdx.x), 8..15 multipliers; 8 + w takes the rows mover w parks.
    const int pwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = (pwave >> 2) * 2 + ((pwave & 3) >> 1) + ((((pwave >> 2) + (pwave & 3)) & 1) ? QN_S2_WAVES : 0);
    const int ltid = wave * 64 + lane; // (the logical thread index: what the diagnostic stamps address)
    (void)ltid;
    const size_t np = (size_t)a.np;
    // (the kernel's own first 32 KB -- the prologue and the movers' code -- are read as DATA by four multipliers at entry: qn_kernels.hip.h, CODE WARM-UP.
    // Round 5's kernel let waves 1..4 do it in front of their rows; a wave's loads return in order, so those four movers' rows -- and the
    // multipliers behind them -- ran 1.5-3 us behind the other three's: in-kernel stamps.)
    unsigned long long pc0;
    asm volatile("s_getpc_b64 %0" : "=s"(pc0));
    QN_S2R_STAMP(0, 0);
    // ONE barrier in front of the exchange: the ring's words are zero (wave 8 clears them first thing) before anybody reads or counts them.
    // Wave 0 and the multipliers execute it BEHIND their first requests (registers only): the control block, the table and the trial point's
    // entries are in flight while the workgroup's last waves arrive (1.4 us after wave 0's first instruction: in-kernel stamps); the movers first.
    auto entry_barrier = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
    const int wg = QN_S2_WG(a);
    const int ij0 = qn_s2_item_of_index(wg, a.nb), ij1 = qn_s2_item_of_index(a.G + wg, a.nb); // (the host deals the pair instance's lists in order)
    bool bad = false;
    const int Ia = ij0 >> 16, Ja = ij0 & 0xffff, Ib = ij1 >> 16, Jb = ij1 & 0xffff;
    constexpr bool diag_a = false; // (the host runs this kernel only where every workgroup's FIRST item is an off-diagonal tile: minimize_impl)
    const bool diag_b = Ib == Jb;
    // ZIG-ZAG (a.zig): launches of odd parity stream the workgroup's two tiles in the OTHER order -- the second list item through the park, the first into
    // the movers' registers.  An XCD's 4 MB L2 keeps its bytes across a kernel boundary (tools/l2_keep_probe.hip: a 4 MB footprint per XCD re-read by the
    // next launch comes at twice the fabric's rate), but its 32 workgroups stream 8.25 MB per launch, in the same order every launch: a least-recently-used
    // cache never hits.  The line search's evaluations are launches next to each other (an iteration is eval, eval, accept-reduce, update tiles,
    // update-reduce: five launches, so two evaluations in a row always differ in parity); with the order turned round the later one STARTS with the tile
    // the earlier one ended with.  Which item a wave multiplies from where does not change a sum: the items' products and folds are the same
    // instructions from the park and from registers (the pair-instance test), every slot and column total is stored under the item's own (I, J), and
    // the lane's scalars are (0 + x) + y with x, y the two items' shares -- commutative.  A workgroup whose second item is a diagonal tile keeps its order
    // (the parked item's code has no diagonal form: sixteen workgroups of 256).
    const bool flip = __builtin_amdgcn_readfirstlane(a.zig != 0 && (a.parity & 1) != 0 && !diag_b) != 0; // (uniform)
    const int ijx = flip ? ij1 : ij0, ijy = flip ? ij0 : ij1; // X: streamed first, through the park to the multipliers; Y: second, multiplied out of the movers' registers
    const int ey = flip ? 0 : 1;                              // Y's place in the pair (colred's first index): X's is 1 - ey
    double row_b = 0.0, row_c = 0.0; // (movers) the second item's row totals, the sliver row's total
    QnS2Sliver sl{};
    if (wave < QN_S2_WAVES) {
        // ------------------------------------------------------------------ movers
        v2d h[QN_S2_RPW];
        // (addresses are formed from an opaque copy of the lane index where they are used: formed once in front of the branch they were held in
        // registers across wave 0's run of the state machine, which then spilled)
        auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
        auto tile_base = [&](const int ij, const int w, const int ln) {
            const int I = ij >> 16, J = ij & 0xffff;
            return a.Q + (size_t)(I * QN_TB + w * QN_S2_RPW) * np + (size_t)J * QN_TB + qn_s2_col(I == J, ln, w);
        };
        if (wave == 0) {
            // (Measured and dropped: the barrier in front of wave 0's publish instead of inside its prologue, the movers behind ALL their first
            // requests -- the stream starts at once, but the movers' nineteenth request is accepted 5-6 us in and the multipliers start that late:
            // 14.3-14.4 us against 13.9-14.1; with eight requests in front 15.8.)
#ifdef QN_S2_STAMPS
            { QnS2Args ap = a; ap.dbg = nullptr; qn_s2_prologue_w0<QN_S2_EVAL, false, decltype(entry_barrier)&, false, BND>(ap, L, entry_barrier); }
#else
            qn_s2_prologue_w0<QN_S2_EVAL, false, decltype(entry_barrier)&, false, BND>(a, L, entry_barrier);
#endif
            QN_S2R_STAMP(11, 0);
            qn_s2r_publish(&Y.mdone, 1u);
            { // workgroup 0 hands the control block on (qn_s2_ctl_out, by one wave)
                constexpr int NW = (int)(sizeof(QnCtl) / 8);
                const uint64_t* src = reinterpret_cast<const uint64_t*>(&L.c);
                if (blockIdx.x == 0) {
                    uint64_t* dst = reinterpret_cast<uint64_t*>(a.ctl2 + (a.parity ^ 1));
                    if (lane < NW) dst[lane] = src[lane];
                    if (64 + lane < NW) dst[64 + lane] = src[64 + lane];
                    if (a.rep_seq != 0) { // the batch's last launch: the host is waiting for this block
                        uint64_t* rp = reinterpret_cast<uint64_t*>(a.rep);
                        if (lane < NW) rp[lane] = src[lane];
                        if (64 + lane < NW) rp[64 + lane] = src[64 + lane];
                        __threadfence_system();
                        if (lane == 0) __hip_atomic_store(a.rep_flag, a.rep_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            if (L.mine) { // its own sixteen rows of the second item (the first item's were parked by the other movers), then its sliver row
                const double* q1 = tile_base(ijy, 0, opaque(lane));
#pragma unroll
                for (int r = 0; r < QN_S2_RPW; ++r) h[r] = ld2(q1 + (size_t)r * np);
            }
        } else {
            // (the movers' barrier comes FIRST: behind their nineteen requests it fell 5-6 us after entry -- a wave is held at a request while the
            // CU's memory queue is full -- and wave 0, waiting in its own, had the state machine done at 8-10 us instead of 4.4)
            // (... but behind the wave's first QN_S2R_HEAD requests: seven movers' worth of them fit the CU's memory queue without holding anybody, and
            // they are in flight while the workgroup's last waves arrive -- the barrier falls ~1.6 us after wave 0's first instruction)
            // wave 0's rows of the first item, three per mover, go out FIRST and are parked first: they come back in front of the wave's own
            const int r0 = (wave - 1) * 3;
            const double* q0 = tile_base(ijx, 0, lane); // (wave 0 has no clone lanes: qn_s2_col(., ., 0) = 2 lane)
            v2d t3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) t3[k] = ld2(q0 + (size_t)min(r0 + k, QN_S2_RPW - 1) * np);
            const double* qa = tile_base(ijx, wave, lane);
#pragma unroll
            for (int r = 0; r < QN_S2R_HEAD; ++r) h[r] = ld2(qa + (size_t)r * np);
            entry_barrier();
#pragma unroll
            for (int r = QN_S2R_HEAD; r < QN_S2_RPW; ++r) h[r] = ld2(qa + (size_t)r * np);
            const double* q1 = tile_base(ijy, wave, lane);
            unsigned n0 = 0u;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (r0 + k < QN_S2_RPW) { park[0][r0 + k][lane] = t3[k]; ++n0; }
            qn_s2r_publish_add(&Y.cnt0, n0);
#pragma unroll
            for (int r = 0; r < QN_S2_RPW; ++r) {
                park[wave][r][lane] = h[r];
                h[r] = ld2(q1 + (size_t)r * np);
                if ((r & 3) == 3) qn_s2r_publish(&Y.prod[wave], (unsigned)r + 1u); // (with the wait for the wave's LDS writes: measured no slower than without -- the in-order variant stays a template switch)
                if (r == 3) QN_S2R_STAMP(2, 448);  // (wave 7: four rows parked and published)
                if (r == 11) QN_S2R_STAMP(5, 448); // (wave 7: twelve)
            }
            QN_S2R_STAMP(7, 448); // (wave 7: its sixteen rows parked, the second item requested)
            QN_S2R_STAMP(6, 64);  // (wave 1)
            QN_S2R_STAMP(8, 256); // (wave 4)
            // (Measured and dropped, round 6, profiles/r06_e_eval_policies_ab_n4096.txt: in the launches that flip, ALL of X parked before the first request for Y --
            // 13.2 -> 14.75 us: even there X's rows land at 5.5-5.8 us, and Y then starts that late; the sliver's row and Y's rows 12..15 as non-temporal loads, so
            // that what the launch leaves in the L2 is three quarters of a tile per workgroup: -0.15 us, inside the noise; all of Y non-temporal in the flipping launches: nothing.)
            qn_s2r_wait_ge<8>(&Y.mdone, 1u, bad); // (long naps: the machine's wave shares its SIMD with three of the waiting ones)
        }
        if (!L.mine) return; // (every wave of the workgroup leaves here, or none does)
        // ---- the SECOND item is multiplied by the wave that holds it, out of its register window (round 5's from-register item): by now the
        // multipliers have been at the first item for 2-3 us, and nothing of the second item has to go through LDS
        const int ln = opaque(lane);
        sl = qn_s2_sliver(a, wave);
        const double* slp = a.Q + (size_t)(sl.D * QN_TB + sl.row) * np + (size_t)sl.D * QN_TB + 2 * ln;
        qn_s2r_wait_ge(&Y.eready, 5u, bad); // the trial point is staged (multipliers 0..4, right behind the machine)
        const int rr = wave * QN_S2_RPW + (ln & 15);
        QnS2RTrial tb;
        tb.xr = (flip ? SH.xa_r : SH.xb_r)[rr]; tb.dr = (flip ? SH.da_r : SH.db_r)[rr]; tb.b_r = SH.bb_r[rr]; tb.g_r = SH.gb_r[rr]; // (b, g: a diagonal Y only, and that is the second item)
        tb.xtj = *reinterpret_cast<const v2d*>((flip ? SH.xa_c : SH.xb_c) + 2 * ln); tb.dj = *reinterpret_cast<const v2d*>((flip ? SH.da_c : SH.db_c) + 2 * ln);
        QnS2SliverVec slv; // (qn_s2_sliver_prep's values; the wave's sliver row is ONE row: its row-side values are the same in every lane -- scalar registers)
        slv.xtj = *reinterpret_cast<const v2d*>(SH.xs_c + 2 * ln);
        slv.xr = qn_uniform(SH.ss[wave][0]); slv.dr = qn_uniform(SH.ss[wave][1]); slv.b = qn_uniform(SH.ss[wave][2]);
        slv.gd = qn_uniform(SH.ss[wave][3] * slv.dr); slv.nf = isfinite(slv.dr) ? 0.0 : 1.0;
        QnS2RSums sb;
        row_b = qn_s2r_item<false>(tb, diag_b, ln, wave, nullptr, h, slp, colred[ey][wave], sb, QnS2RNoWait()); // (diag_b: Y is the second item, or neither tile is diagonal)
        QN_S2R_STAMP(14, 448); // (wave 7: the second item done)
        // the lane's shares of the group's scalars in round 5's order: ((0 + first item) + second item) + sliver -- the first item's from its multiplier
        qn_s2r_wait_ge(&Y.adone[wave], 1u, bad);
        const v2d fa = park[wave][0][ln]; // (pf, pg of the first item on this lane; it is never a diagonal tile: its g'd and non-finite shares are zero)
        double sacc[4] = {(0.0 + fa.x) + sb.pf, (0.0 + fa.y) + sb.pg, (0.0 + 0.0) + sb.p4, (0.0 + 0.0) + sb.p5};
        const double t0s = qn_s2_eval_sliver(slv, h[0], sl.row, ln, sacc);
        { // ONE fold for the pair: the four scalars and the sliver's row total (value 6: lane 48 holds it)
            double sv[8] = {sacc[0], sacc[1], 0.0, 0.0, sacc[2], sacc[3], t0s, 0.0};
            QnWaveFold<8, 32, true>::run(sv, ln);
            if ((ln & 7) == 0 && ln < 48) sred[wave][ln >> 3] = sv[0];
            row_c = sv[0];
        }
        QN_S2R_STAMP(12, 448); // (wave 7: folded)
        QN_S2R_STAMP(3, 0);    // (wave 0: folded)
    }
    // ---------------------------------------------------------------------- multipliers: wave 8 + w takes the FIRST item's rows of wave w out of the park
    const int mw = wave - QN_S2_WAVES; // (movers: negative, and nothing below uses it for them)
    if (wave >= QN_S2_WAVES) {
        if (wave == QN_S2_WAVES && lane < (int)(sizeof(QnS2RSync) / 4)) reinterpret_cast<unsigned*>(&Y)[lane] = 0u; // (wave 8)
#ifdef QN_S2_STAMPS
        if (wave == QN_S2_WAVES && lane < 16 && lane > 0) SH.stamps[lane] = 0ull;
#endif
        const QnS2Sliver slm = qn_s2_sliver(a, mw);
        // Multipliers 0..4 request, for ONE of the workgroup's five blocks each (first item: I, J; second item: I, J; the sliver's diagonal block),
        // the entries the trial point is made of -- x for both settings of the buffer toggle the control block holds, v, s likewise, u, and b, g
        // where a diagonal tile or the sliver needs them: two consecutive entries per lane, eight 1 KB requests per wave, BEFORE the movers' burst.
        const int blk = mw == 0 ? Ia : (mw == 1 ? Ja : (mw == 2 ? Ib : (mw == 3 ? Jb : slm.D))); // (uniform)
        v2d e_x0, e_x1, e_v, e_s0, e_s1, e_u, e_b, e_g;
        v2d e_lo = {-INFINITY, -INFINITY}, e_hi = {INFINITY, INFINITY}; // BND, a.projfold: the line search's box at the block (BackTrackingB projects its trial points)
        // ... and, BND with a.projfold, in workgroups 0 .. nb - 1: multipliers 6 and 7 hold block blockIdx.x's entries, 64 each -- the block's share of
        // ||P(x + t d) - x||^2 (s2_proj_kernel's two waves: the same entries, the same wave sums, added in the same order)
        const bool pj_wave = BND && a.projfold && (mw == 6 || mw == 7) && (int)blockIdx.x < a.nb; // (uniform)
        double pj_x0 = 0.0, pj_x1 = 0.0, pj_s0 = 0.0, pj_s1 = 0.0, pj_v = 0.0, pj_u = 0.0, pj_lo = -INFINITY, pj_hi = INFINITY;
        if (mw < 5) {
            const unsigned ei = (unsigned)blk * QN_TB + 2u * (unsigned)lane;
            e_x0 = ld2(a.F.X0 + ei); e_x1 = ld2(a.F.X0 + np + ei); e_v = ld2(a.F.VV + ei);
            e_s0 = ld2(a.F.S0 + ei); e_s1 = ld2(a.F.S0 + np + ei); e_u = ld2(a.F.UN + ei);
            if (mw == 2 || mw == 4) { e_b = ld2(a.F.b + ei); e_g = ld2(a.F.G + ei); }
            if (BND && a.projfold && a.llb) { e_lo = ld2(a.llb + ei); e_hi = ld2(a.lub + ei); }
        }
        if (pj_wave) {
            const unsigned gi = blockIdx.x * QN_TB + (unsigned)(mw - 6) * 64u + (unsigned)lane;
            pj_x0 = a.F.X0[gi]; pj_x1 = a.F.X0[np + gi]; pj_s0 = a.F.S0[gi]; pj_s1 = a.F.S0[np + gi]; pj_v = a.F.VV[gi]; pj_u = a.F.UN[gi];
            if (a.llb) { pj_lo = a.llb[gi]; pj_hi = a.lub[gi]; }
        }
        unsigned cw = 0u;
        if (mw >= 4) cw = ((const volatile unsigned*)(pc0 & ~127ull))[(size_t)((mw - 4) * 64 + lane) * 32];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (wave 8: the ring's words are cleared)
        entry_barrier();
        qn_s2r_wait_ge<8>(&Y.mdone, 1u, bad); // (long naps: the machine's wave shares its SIMD with three of the waiting ones)
        if (!L.mine) return;
        const QnEvalReq q = qn_s2_eval_req<true, BND>(L.c, false, BND && a.projfold != 0);
        // (BND, a.projfold) a PROJECTED trial: x + t d clamped into the line search's box where it is formed -- the point s2_proj_kernel would have stored,
        // entry by entry the same operations -- and evaluated like a stored point: with d = 0 (qn_s2_trial's `!is_t` branch; qn_s2_advance reads no g'd from it)
        const bool pj = BND && a.projfold && __builtin_amdgcn_readfirstlane(L.c.req_project) != 0; // (uniform)
        if (pj_wave && pj) { // backtracking_b.rs:31-34, :67 -- as s2_proj_kernel
            const double xi = q.xc ? pj_x1 : pj_x0, si = q.sc ? pj_s1 : pj_s0;
            double d;
            double z = qn_s2_trial(q, xi, pj_v, si, pj_u, d);
            z = fmin(fmax(z, pj_lo), pj_hi);
            const double df = z - xi;
            const double p = qn_wave_sum(df * df);
            if (lane == 0) SH.pjred[mw - 6] = p;
        }
        if (mw < 5) { // the trial point at this wave's block: qn_s2_trial, entry by entry -- what every wave of round 5's kernel formed for itself
            const v2d ex = q.xc ? e_x1 : e_x0, es = q.sc ? e_s1 : e_s0;
            v2d xt, dd;
            { double d0, d1; xt.x = qn_s2_trial(q, ex.x, e_v.x, es.x, e_u.x, d0); xt.y = qn_s2_trial(q, ex.y, e_v.y, es.y, e_u.y, d1); dd.x = d0; dd.y = d1; }
            if (pj) {
                xt.x = fmin(fmax(xt.x, e_lo.x), e_hi.x); xt.y = fmin(fmax(xt.y, e_lo.y), e_hi.y);
                dd.x = 0.0; dd.y = 0.0;
            }
            double* xdst = mw == 0 ? SH.xa_r : (mw == 1 ? SH.xa_c : (mw == 2 ? SH.xb_r : (mw == 3 ? SH.xb_c : SH.xs_c)));
            *reinterpret_cast<v2d*>(xdst + 2 * lane) = xt;
            if (mw < 4) {
                double* ddst = mw == 0 ? SH.da_r : (mw == 1 ? SH.da_c : (mw == 2 ? SH.db_r : SH.db_c));
                *reinterpret_cast<v2d*>(ddst + 2 * lane) = dd;
            }
            if (mw == 2) { *reinterpret_cast<v2d*>(SH.bb_r + 2 * lane) = e_b; *reinterpret_cast<v2d*>(SH.gb_r + 2 * lane) = e_g; }
            if (mw == 4) { // the sliver's eight rows of block D: xt, d, b, g per row (rows 8 s .. 8 s + 7: lanes 4 s .. 4 s + 3)
                const int j = 2 * lane - (slm.row - mw); // (slm.row - mw = 8 s: the first of the workgroup's eight rows)
                if (j >= 0 && j < 8) {
                    SH.ss[j][0] = xt.x; SH.ss[j][1] = dd.x; SH.ss[j][2] = e_b.x; SH.ss[j][3] = e_g.x;
                    SH.ss[j + 1][0] = xt.y; SH.ss[j + 1][1] = dd.y; SH.ss[j + 1][2] = e_b.y; SH.ss[j + 1][3] = e_g.y;
                }
            }
            qn_s2r_publish_add(&Y.eready, 1u);
        }
        qn_s2r_wait_ge(&Y.eready, 5u, bad);
        QN_S2R_STAMP(9, 960); // (multiplier 7: the trial point is staged)
        const int rr = mw * QN_S2_RPW + (lane & 15);
        QnS2RTrial ta;
        ta.xr = (flip ? SH.xb_r : SH.xa_r)[rr]; ta.dr = (flip ? SH.db_r : SH.da_r)[rr]; ta.b_r = 0.0; ta.g_r = 0.0;
        ta.xtj = *reinterpret_cast<const v2d*>((flip ? SH.xb_c : SH.xa_c) + 2 * lane); ta.dj = *reinterpret_cast<const v2d*>((flip ? SH.db_c : SH.da_c) + 2 * lane);
        unsigned have = 0u; // (uniform) the last count read from the mover: the word is polled only when it does not cover the rows yet
        if (mw == 0) { qn_s2r_wait_ge(&Y.cnt0, 16u, bad); have = 16u; } // (wave 0's share of the first item: parked by the other movers)
        // in front of rows r .. r + 3: wait for these four and the four that are read ahead, if the last count read does not cover them
        auto row_hook = [&](const int r) __attribute__((always_inline)) {
            if (r & 3) return; // (rows are taken four at a time: one look at the mover's count, four reads in flight together)
            const unsigned want = (unsigned)(r + 8 < QN_S2_RPW ? r + 8 : QN_S2_RPW);
            if (have < want) have = qn_s2r_wait_ge(&Y.prod[mw], want, bad);
        };
        v2d hnone[QN_S2_RPW]; // (the from-park instance never touches a window)
        QnS2RSums sa;
        const double row_a = qn_s2r_item<true>(ta, diag_a, lane, mw, &park[mw][0][0], hnone, nullptr, colred[1 - ey][mw], sa, row_hook);
        if ((lane & 3) == 0) a.partE[(unsigned)(((ijx >> 16) * a.nb + (ijx & 0xffff)) * QN_TB + mw * QN_S2_RPW + (lane >> 2))] = row_a; // (X is off-diagonal: its slot is its own)
        // the lane's shares of the two scalar sums go to the mover that holds the second item, through the first slot of the park (all of it has been read)
        park[mw][0][lane] = (v2d){sa.pf, sa.pg};
        qn_s2r_publish(&Y.adone[mw], 1u);
        qn_code_warm_done(cw, a.n < 0, a.wgS);
        QN_S2R_STAMP(13, 960); // (multiplier 7: the first item done)
        QN_S2R_STAMP(1, 512);  // (multiplier 0)
        QN_S2R_STAMP(10, 768); // (multiplier 4)
    }
    if (bad) __hip_atomic_store(&Y.bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (read behind the exchange barrier)
    __syncthreads();
    QN_S2R_STAMP(4, 0);
    double wgk = 0.0;
    if (tid < 2 * QN_TB) { // threads 0..127: the first item's column part, 128..255: the second item's
        const int e = tid >> 7, cidx = tid & (QN_TB - 1);
        double acc = colred[e][0][cidx];
#pragma unroll
        for (int w = 1; w < QN_S2_WAVES; ++w) acc = acc + colred[e][w][cidx];
        const bool dg = e == 0 ? diag_a : diag_b;
        const int Ie = e == 0 ? Ia : Ib, Je = e == 0 ? Ja : Jb;
        if (dg) colsum[cidx] = acc; // both parts of a diagonal tile belong to block-row I: one slot
        else a.partE[(unsigned)((Je * a.nb + Ie) * QN_TB + cidx)] = acc;
    }
    if (tid < QN_S2_NSE) wgk = wgk + qn_s2_wave_total(sred, tid); // (waves in order; scalars 2..5 are zero off the diagonal items)
    if (diag_a || diag_b) __syncthreads(); // (uniform)
    if (wave < QN_S2_WAVES) {
        if ((lane & 3) == 0) {
            const int rl = wave * QN_S2_RPW + (lane >> 2);
            double v = row_b;
            if (diag_b) v = v + colsum[rl];
            a.partE[(unsigned)(((ijy >> 16) * a.nb + (ijy & 0xffff)) * QN_TB + rl)] = v;
        }
        if (lane == 48) a.partE[(unsigned)((sl.D * a.nb + sl.D) * QN_TB + sl.row)] = row_c;
    }
    // a spin ran out somewhere in this workgroup: its sums are not to be believed -- NaN ends the run (out of domain) instead of a wrong step
    const bool anybad = Y.bad != 0u;
    QN_S2R_STAMP(15, 0);
#ifdef QN_S2_STAMPS
    __syncthreads();
    if (tid < 16 && a.dbg && blockIdx.x < 256) a.dbg[(((size_t)(a.slot & 63)) * 256 + blockIdx.x) * 16 + tid] = SH.stamps[tid];
#endif
    if (BND && a.projfold && tid == 0 && (int)blockIdx.x < a.nb && L.c.req_project != 0) // (behind the exchange barrier: multipliers 6 and 7 have left their sums)
        a.wgS[((size_t)a.parity * QN_S2_ROW + 6) * a.trows + blockIdx.x] = SH.pjred[0] + SH.pjred[1];
    if (tid < QN_S2_NSE) { // sred column -> table column: xt'(Q xt - 2b), d'(Q xt - b), (b'xt = 0), (b'd = 0), g'd, #non-finite d
        const int col = tid == 1 ? 2 : (tid == 2 ? 1 : tid);
        a.wgS[((size_t)a.parity * QN_S2_ROW + col) * a.trows + blockIdx.x] = anybad ? __builtin_nan("") : wgk;
    }
}
