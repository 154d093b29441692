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
#define QN_S2R_SPIN_MAX (1 << 18) // polls of an LDS word (~150 cycles each with the s_sleep) before the workgroup gives up

struct QnS2RSync {
    unsigned prod[QN_S2_WAVES]; // rows of its sequence mover w has parked (wave 0's first-item rows are parked by the others: cnt0)
    unsigned cons[QN_S2_WAVES]; // rows multiplier w has read
    unsigned cnt0;              // rows of wave 0's share of the first item parked so far (16 when complete)
    unsigned eready;            // blocks of the trial point staged so far (5 when complete)
    unsigned mdone;             // the machine has run: L.c and L.mine are final
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
__device__ __forceinline__ void qn_s2r_publish(unsigned* p, const unsigned v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
    v2d park[QN_S2_WAVES][QN_S2_RPW][64]; // 128 KB: sixteen ring slots per mover
};

// One item from the park against the STAGED trial point: qn_s2_eval_item<FROM_PARK, ., ., EARLY> (qn_sym2.hip.h) behind its first five lines --
// the same products, sums and exchanges in the same order -- with xt and d read instead of formed.
struct QnS2RTrial {
    double xr, dr, b_r, g_r; // row 16 w + (lane & 15) of block I
    v2d xtj, dj;             // this lane's two columns of block J
};
template <class WaitRow>
__device__ __forceinline__ double qn_s2r_item(const QnS2RTrial& v, const bool diag, const int lane, const int wave, const v2d* parkw,
                                              double* __restrict__ colred_w, double (&sacc)[4], WaitRow&& wait_row) {
    const double xr = v.xr, dr = v.dr;
    double p4 = 0.0, p5 = 0.0; // diagonal items: g'd, #non-finite d over block I (lanes 0..15 of every wave)
    if (diag && lane < 16) { p4 = v.g_r * dr; p5 = isfinite(dr) ? 0.0 : 1.0; }
    v2d xtj = v.xtj, dj = v.dj;
    if (!qn_s2_row_on(diag, lane, wave)) { xtj = (v2d){0.0, 0.0}; dj = (v2d){0.0, 0.0}; }
    double cx = 0.0, cy = 0.0;
    double racc[QN_S2_RPW / 2];
    v2d hq[8]; // a rolling window of eight rows: while rows r .. r + 3 are multiplied, rows r + 4 .. r + 7 are on their way out of the park
#pragma unroll
    for (int r = 0; r < QN_S2_RPW; ++r) {
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
        const v2d hv = hq[r & 7];
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
        // (a use per row: without it the compiler sinks every row's arithmetic below the last wait -- see qn_s2_eval_item)
        qn_keepalive(cx); qn_keepalive(cy); qn_keepalive(racc[r & (QN_S2_RPW / 2 - 1)]);
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
    sacc[0] = sacc[0] + pf; sacc[1] = sacc[1] + pg; sacc[2] = sacc[2] + p4; sacc[3] = sacc[3] + p5;
    return racc[0];
}

#ifdef QN_S2_STAMPS
#define QN_S2R_STAMP(k, t) do { if (threadIdx.x == (t)) SH.stamps[k] = wall_clock64(); } while (0)
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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t np = (size_t)a.np;
    QN_S2R_STAMP(0, 0);
    // ONE barrier in front of the exchange: the ring's words are zero (wave 8 clears them first thing) before anybody reads or counts them.
    // Wave 0 and the multipliers execute it BEHIND their first requests (registers only): the control block, the table and the trial point's
    // entries are in flight while the workgroup's last waves arrive (1.4 us after wave 0's first instruction: in-kernel stamps); the movers first.
    auto entry_barrier = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
    const int wg = QN_S2_WG(a);
    const int ij0 = qn_s2_item_of_index(wg, a.nb), ij1 = qn_s2_item_of_index(a.G + wg, a.nb); // (the host deals the pair instance's lists in order)
    bool bad = false;
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
                const double* q1 = tile_base(ij1, 0, opaque(lane));
#pragma unroll
                for (int r = 0; r < QN_S2_RPW; ++r) h[r] = ld2(q1 + (size_t)r * np);
            }
        } else {
            // (the movers' barrier comes FIRST: behind their nineteen requests it fell 5-6 us after entry -- a wave is held at a request while the
            // CU's memory queue is full -- and wave 0, waiting in its own, had the state machine done at 8-10 us instead of 4.4)
            entry_barrier();
            QN_S2R_STAMP(5, 64); // (wave 1 behind the entry barrier)
            unsigned cw = 0u;
            if (wave <= 4) cw = qn_code_warm_issue((wave - 1) * 64 + lane); // (qn_kernels.hip.h, CODE WARM-UP)
            // wave 0's rows of the first item, three per mover, go out FIRST and are parked first: they come back in front of the wave's own
            const int r0 = (wave - 1) * 3;
            const double* q0 = tile_base(ij0, 0, lane); // (wave 0 has no clone lanes: qn_s2_col(., ., 0) = 2 lane)
            v2d t3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) t3[k] = ld2(q0 + (size_t)min(r0 + k, QN_S2_RPW - 1) * np);
            const double* qa = tile_base(ij0, wave, lane);
#pragma unroll
            for (int r = 0; r < QN_S2_RPW; ++r) h[r] = ld2(qa + (size_t)r * np);
            const double* q1 = tile_base(ij1, wave, lane);
            unsigned n0 = 0u;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (r0 + k < QN_S2_RPW) { park[0][r0 + k][lane] = t3[k]; ++n0; }
            qn_s2r_publish_add(&Y.cnt0, n0);
            QN_S2R_STAMP(6, 448);
#pragma unroll
            for (int r = 0; r < QN_S2_RPW; ++r) {
                park[wave][r][lane] = h[r];
                h[r] = ld2(q1 + (size_t)r * np);
                if ((r & 3) == 3) qn_s2r_publish(&Y.prod[wave], (unsigned)r + 1u);
            }
            QN_S2R_STAMP(7, 448); // (wave 7: its sixteen rows parked, the second item requested)
            qn_code_warm_done(cw, a.n < 0, a.wgS);
            qn_s2r_wait_ge<8>(&Y.mdone, 1u, bad); // (long naps: the machine's wave shares its SIMD with three of the waiting ones)
        }
        if (!L.mine) return; // (every wave of the workgroup leaves here, or none does)
        // second item: row r into slot r once the multiplier has read the first item's row r; the register of row 0 takes the sliver's row
        const QnS2Sliver sl = qn_s2_sliver(a, opaque(wave));
        const double* slp = a.Q + (size_t)(sl.D * QN_TB + sl.row) * np + (size_t)sl.D * QN_TB + 2 * opaque(lane);
        unsigned freed = 0u;
#pragma unroll
        for (int r = 0; r < QN_S2_RPW; ++r) {
            if (freed < (unsigned)r + 1u) freed = qn_s2r_wait_ge(&Y.cons[wave], (unsigned)r + 1u, bad);
            park[wave][r][lane] = h[r];
            if (r == 0) h[0] = ld2(slp);
            if ((r & 3) == 3) qn_s2r_publish(&Y.prod[wave], 17u + (unsigned)r);
        }
        QN_S2R_STAMP(8, 448); // (wave 7: the second item parked)
        if (freed < 17u) freed = qn_s2r_wait_ge(&Y.cons[wave], 17u, bad);
        park[wave][0][lane] = h[0];
        qn_s2r_publish(&Y.prod[wave], 33u);
        QN_S2R_STAMP(3, 0); // (wave 0: its second item and its sliver row parked)
    }
    // ---------------------------------------------------------------------- multipliers (wave 8 + w is the old kernel's wave w)
    const int mw = wave - QN_S2_WAVES; // (movers: negative, and nothing below the exchange uses it)
    const int Ia = ij0 >> 16, Ja = ij0 & 0xffff, Ib = ij1 >> 16, Jb = ij1 & 0xffff;
    constexpr bool diag_a = false; // (the host runs this kernel only where every workgroup's FIRST item is an off-diagonal tile: minimize_impl)
    const bool diag_b = Ib == Jb;
    double row_b = 0.0, row_c = 0.0;
    QnS2Sliver sl{};
    if (wave >= QN_S2_WAVES) {
        if (tid - QN_S2_WAVES * 64 < (int)(sizeof(QnS2RSync) / 4)) reinterpret_cast<unsigned*>(&Y)[tid - QN_S2_WAVES * 64] = 0u; // (wave 8)
#ifdef QN_S2_STAMPS
        if (tid - QN_S2_WAVES * 64 < 16 && tid - QN_S2_WAVES * 64 > 0) SH.stamps[tid - QN_S2_WAVES * 64] = 0ull;
#endif
        sl = qn_s2_sliver(a, mw);
        // Multipliers 0..4 request, for ONE of the workgroup's five blocks each (first item: I, J; second item: I, J; the sliver's diagonal block),
        // the entries the trial point is made of -- x for both settings of the buffer toggle the control block holds, v, s likewise, u, and b, g
        // where a diagonal tile or the sliver needs them: two consecutive entries per lane, eight 1 KB requests per wave, BEFORE the movers' burst.
        const int blk = mw == 0 ? Ia : (mw == 1 ? Ja : (mw == 2 ? Ib : (mw == 3 ? Jb : sl.D))); // (uniform)
        v2d e_x0, e_x1, e_v, e_s0, e_s1, e_u, e_b, e_g;
        if (mw < 5) {
            const unsigned ei = (unsigned)blk * QN_TB + 2u * (unsigned)lane;
            e_x0 = ld2(a.F.X0 + ei); e_x1 = ld2(a.F.X0 + np + ei); e_v = ld2(a.F.VV + ei);
            e_s0 = ld2(a.F.S0 + ei); e_s1 = ld2(a.F.S0 + np + ei); e_u = ld2(a.F.UN + ei);
            if (mw == 2 || mw == 4) { e_b = ld2(a.F.b + ei); e_g = ld2(a.F.G + ei); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (wave 8: the ring's words are cleared)
        entry_barrier();
        qn_s2r_wait_ge<8>(&Y.mdone, 1u, bad); // (long naps: the machine's wave shares its SIMD with three of the waiting ones)
        if (!L.mine) return;
        QN_S2R_STAMP(2, 512);
        const QnEvalReq q = qn_s2_eval_req<true>(L.c, false);
        if (mw < 5) { // the trial point at this wave's block: qn_s2_trial, entry by entry -- what every wave of round 5's kernel formed for itself
            const v2d ex = q.xc ? e_x1 : e_x0, es = q.sc ? e_s1 : e_s0;
            v2d xt, dd;
            { double d0, d1; xt.x = qn_s2_trial(q, ex.x, e_v.x, es.x, e_u.x, d0); xt.y = qn_s2_trial(q, ex.y, e_v.y, es.y, e_u.y, d1); dd.x = d0; dd.y = d1; }
            double* xdst = mw == 0 ? SH.xa_r : (mw == 1 ? SH.xa_c : (mw == 2 ? SH.xb_r : (mw == 3 ? SH.xb_c : SH.xs_c)));
            *reinterpret_cast<v2d*>(xdst + 2 * lane) = xt;
            if (mw < 4) {
                double* ddst = mw == 0 ? SH.da_r : (mw == 1 ? SH.da_c : (mw == 2 ? SH.db_r : SH.db_c));
                *reinterpret_cast<v2d*>(ddst + 2 * lane) = dd;
            }
            if (mw == 2) { *reinterpret_cast<v2d*>(SH.bb_r + 2 * lane) = e_b; *reinterpret_cast<v2d*>(SH.gb_r + 2 * lane) = e_g; }
            if (mw == 4) { // the sliver's eight rows of block D: xt, d, b, g per row (rows 8 s .. 8 s + 7: lanes 4 s .. 4 s + 3)
                const int j = 2 * lane - (sl.row - mw); // (sl.row - mw = 8 s: the first of the workgroup's eight rows)
                if (j >= 0 && j < 8) {
                    SH.ss[j][0] = xt.x; SH.ss[j][1] = dd.x; SH.ss[j][2] = e_b.x; SH.ss[j][3] = e_g.x;
                    SH.ss[j + 1][0] = xt.y; SH.ss[j + 1][1] = dd.y; SH.ss[j + 1][2] = e_b.y; SH.ss[j + 1][3] = e_g.y;
                }
            }
            qn_s2r_publish_add(&Y.eready, 1u);
        }
        qn_s2r_wait_ge(&Y.eready, 5u, bad);
        QN_S2R_STAMP(9, 960); // (wave 15: the trial point is staged)
        const int rr = mw * QN_S2_RPW + (lane & 15);
        QnS2RTrial ta, tb;
        ta.xr = SH.xa_r[rr]; ta.dr = SH.da_r[rr]; ta.b_r = 0.0; ta.g_r = 0.0;
        ta.xtj = *reinterpret_cast<const v2d*>(SH.xa_c + 2 * lane); ta.dj = *reinterpret_cast<const v2d*>(SH.da_c + 2 * lane);
        tb.xr = SH.xb_r[rr]; tb.dr = SH.db_r[rr]; tb.b_r = SH.bb_r[rr]; tb.g_r = SH.gb_r[rr];
        tb.xtj = *reinterpret_cast<const v2d*>(SH.xb_c + 2 * lane); tb.dj = *reinterpret_cast<const v2d*>(SH.db_c + 2 * lane);
        QnS2SliverVec slv; // (qn_s2_sliver_prep's values; the wave's sliver row is ONE row: its row-side values are the same in every lane -- scalar registers)
        slv.xtj = *reinterpret_cast<const v2d*>(SH.xs_c + 2 * lane);
        slv.xr = qn_uniform(SH.ss[mw][0]); slv.dr = qn_uniform(SH.ss[mw][1]); slv.b = qn_uniform(SH.ss[mw][2]);
        slv.gd = qn_uniform(SH.ss[mw][3] * slv.dr); slv.nf = isfinite(slv.dr) ? 0.0 : 1.0;
        unsigned have = 0u; // (uniform) the last count read from the mover: the word is polled only when it does not cover the row yet
        if (mw == 0) { qn_s2r_wait_ge(&Y.cnt0, 16u, bad); have = 16u; } // (wave 0's share of the first item: parked by the other movers)
        double sacc[4] = {0.0, 0.0, 0.0, 0.0};
        const v2d* pw = &park[mw][0][0];
        // in front of rows r .. r + 3 of an item whose first row is number `base` of the mover's sequence: hand back the slots of the four rows
        // before (they are in registers by now), wait for these four and the next four if the last count read does not cover them
        auto row_hook = [&](const unsigned base, const int r) __attribute__((always_inline)) {
            if (r & 3) return; // (rows are taken four at a time: one look at the mover's count, four reads in flight together)
            if (r) qn_s2r_publish(&Y.cons[mw], base + (unsigned)r);
            const unsigned want = base + (unsigned)(r + 8 < QN_S2_RPW ? r + 8 : QN_S2_RPW); // (the rows about to be multiplied AND the four to be read ahead)
            if (have < want) have = qn_s2r_wait_ge(&Y.prod[mw], want, bad);
        };
        const double row_a = qn_s2r_item(ta, diag_a, lane, mw, pw, colred[0][mw], sacc, [&](const int r) __attribute__((always_inline)) { row_hook(0u, r); });
        qn_s2r_publish(&Y.cons[mw], 16u);
        if ((lane & 3) == 0) a.partE[(unsigned)((Ia * a.nb + Ja) * QN_TB + mw * QN_S2_RPW + (lane >> 2))] = row_a; // (off-diagonal: its slot is its own)
        QN_S2R_STAMP(13, 960);
        row_b = qn_s2r_item(tb, diag_b, lane, mw, pw, colred[1][mw], sacc, [&](const int r) __attribute__((always_inline)) { row_hook(16u, r); });
        qn_s2r_publish(&Y.cons[mw], 32u);
        QN_S2R_STAMP(14, 960);
        if (have < 33u) have = qn_s2r_wait_ge(&Y.prod[mw], 33u, bad);
        const v2d hs = park[mw][0][lane];
        const double t0s = qn_s2_eval_sliver(slv, hs, sl.row, lane, sacc);
        { // ONE fold for the pair: the four scalars and the sliver's row total (value 6: lane 48 holds it)
            double sv[8] = {sacc[0], sacc[1], 0.0, 0.0, sacc[2], sacc[3], t0s, 0.0};
            QnWaveFold<8, 32, true>::run(sv, lane);
            if ((lane & 7) == 0 && lane < 48) sred[mw][lane >> 3] = sv[0];
            row_c = sv[0];
        }
        QN_S2R_STAMP(12, 960);
        QN_S2R_STAMP(1, 512);  // (wave 8: folded)
        QN_S2R_STAMP(10, 768); // (wave 12: folded)
    }
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
    if (wave >= QN_S2_WAVES) {
        if ((lane & 3) == 0) {
            const int rl = mw * QN_S2_RPW + (lane >> 2);
            double v = row_b;
            if (diag_b) v = v + colsum[rl];
            a.partE[(unsigned)((Ib * a.nb + Jb) * QN_TB + rl)] = v;
        }
        if (lane == 48) a.partE[(unsigned)((sl.D * a.nb + sl.D) * QN_TB + sl.row)] = row_c;
    }
    // a spin ran out somewhere in this workgroup: its sums are not to be believed -- NaN ends the run (out of domain) instead of a wrong step
    const bool anybad = __syncthreads_or(bad ? 1 : 0) != 0;
    QN_S2R_STAMP(15, 0);
#ifdef QN_S2_STAMPS
    __syncthreads();
    if (tid < 16 && a.dbg && blockIdx.x < 256) a.dbg[(((size_t)(a.slot & 63)) * 256 + blockIdx.x) * 16 + tid] = SH.stamps[tid];
#endif
    if (tid < QN_S2_NSE) { // sred column -> table column: xt'(Q xt - 2b), d'(Q xt - b), (b'xt = 0), (b'd = 0), g'd, #non-finite d
        const int col = tid == 1 ? 2 : (tid == 2 ? 1 : tid);
        a.wgS[((size_t)a.parity * QN_S2_ROW + col) * a.trows + blockIdx.x] = anybad ? __builtin_nan("") : wgk;
    }
}
