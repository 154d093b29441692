"""Worker for the world_size-2 tests; launched by test_dist_gloo.py / test_gpu_sharded.py through
`python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 ... tests/dist_worker.py MODE OUT`."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import __graft_entry__ as ge
    qn = ge.load_package()
    from oracle import qn_oracle as qo
    import problems as P

    result = {"rank": rank, "world": world}
    if mode == "cpu":
        # (1) the host all-gather helper
        ag = qn.dist.gloo_allgather()
        send = np.arange(5, dtype=np.float64) + 10.0 * rank
        recv = np.zeros(5 * world)
        ag(send, recv)
        result["allgather_ok"] = bool(np.array_equal(recv, np.concatenate([np.arange(5) + 10.0 * r for r in range(world)])))

        # (2) N > 1 semantics on the CPU: row partition + all-gather of mat-vec slices + replicated scalar work
        n = 203
        rpr, n_pad = qn.partition(n, world)
        lo, hi = qn.dist.row_range(n, rank, world, rpr)
        diag = P.synth_diag(n)
        b, x0 = P.synth_vectors(n)
        q_rows = qo.synth_rows(n, lo, hi - lo, P.SEED, diag)  # shard-local generation
        calls = [0]

        def sharded_oracle(x):
            calls[0] += 1
            sl = np.zeros(rpr)
            for r in range(hi - lo):  # row sums left to right, as the oracle's quadratic does
                acc = 0.0
                row = q_rows[r]
                for j in range(n):
                    acc += row[j] * x[j]
                sl[r] = acc
            full = np.zeros(rpr * world)
            ag(sl, full)
            qx = full[:n]
            f = 0.5 * qo.dot(x, qx) - qo.dot(b, x)
            return f, qx - b

        s = qo.Solver(qo.BFGS, 1e-10, x0)
        st = s.minimize(qo.morethuente(), sharded_oracle, 6, 20, trace_cap=6, trace_x=True)
        # unsharded run of the same thing on every rank
        q = qo.synth_rows(n, 0, n, P.SEED, diag)
        ref = qo.Solver(qo.BFGS, 1e-10, x0)
        st_ref = ref.minimize(qo.morethuente(), qo.QuadraticOracle(q, b), 6, 20, trace_cap=6, trace_x=True)
        result["sharded_equals_unsharded_bitwise"] = bool(st == st_ref and np.array_equal(s.trace_x, ref.trace_x)
                                                          and np.array_equal(s.approx_inv_hessian, ref.approx_inv_hessian))
        result["x_hex"] = [float(v).hex() for v in s.x]
        result["partition"] = [rpr, n_pad, lo, hi]
    elif mode == "gpu":
        # two ranks share the one GPU of the dev box; slices are exchanged through host memory with gloo.
        n, iters = 700, 25
        diag = P.synth_diag(n)
        b, x0 = P.synth_vectors(n)
        ctx = qn.dist.sharded_context(0, host_exchange=True)
        ctx.comm_check()  # one verified rank-tagged all-gather through the context's exchange
        obj = qn.Quadratic.synthetic(n, P.SEED, diag, b, ctx=ctx)
        s = qn.BFGS(1e-10, x0, ctx=ctx)
        s.set_trace(iters, with_x=True)
        try:
            s.minimize(qn.MoreThuente(), obj, iters, 20)
        except qn.MaxIterReached:
            pass
        tr, xs = s.trace()
        h = s.approx_inv_hessian(all_ranks=True)
        # the same solve on a private single-rank context
        ctx1 = qn.Context(0)
        obj1 = qn.Quadratic.synthetic(n, P.SEED, diag, b, ctx=ctx1)
        s1 = qn.BFGS(1e-10, x0, ctx=ctx1)
        s1.set_trace(iters, with_x=True)
        s1.set_sync_mode(1)
        try:
            s1.minimize(qn.MoreThuente(), obj1, iters, 20)
        except qn.MaxIterReached:
            pass
        tr1, xs1 = s1.trace()
        h1 = s1.approx_inv_hessian()
        result["iters"] = len(tr)
        result["trace_equal"] = bool(tr == tr1)
        result["x_equal"] = bool(np.array_equal(xs, xs1))
        result["h_equal"] = bool(np.array_equal(h, h1))
        result["objective_rows_ok"] = bool(np.array_equal(
            obj.rows(*[(lambda lo, hi: (lo, hi - lo))(*qn.dist.row_range(n, rank, world, qn.partition(n, world)[0]))][0]),
            qo.synth_rows(n, *[(lambda lo, hi: (lo, hi - lo))(*qn.dist.row_range(n, rank, world, qn.partition(n, world)[0]))][0], P.SEED, diag)))
        ev = obj(x0)
        ev1 = obj1(x0)
        result["eval_equal"] = bool(ev.f() == ev1.f() and np.array_equal(ev.g(), ev1.g()))
        # DFP + backtracking on the sharded context too
        s2 = qn.DFP(1e-10, x0, ctx=ctx)
        s2.set_trace(10, with_x=True)
        try:
            s2.minimize(qn.BackTracking(1e-4, 0.5), obj, 10, 20)
        except qn.MaxIterReached:
            pass
        s3 = qn.DFP(1e-10, x0, ctx=ctx1)
        s3.set_trace(10, with_x=True)
        s3.set_sync_mode(1)
        try:
            s3.minimize(qn.BackTracking(1e-4, 0.5), obj1, 10, 20)
        except qn.MaxIterReached:
            pass
        result["dfp_bt_equal"] = bool(np.array_equal(s2.trace()[1], s3.trace()[1]))
        # log-sum-exp objective (rows of A sharded, gradient summed over ranks in fixed order): sharded vs single rank
        rng = np.random.default_rng(5)
        m2, n2 = 333, 150
        a = rng.standard_normal((m2, n2)) * (3.0 / np.sqrt(n2))
        cvec = rng.standard_normal(m2)
        xs0 = rng.standard_normal(n2)
        lse = qn.LogSumExp(a, cvec, 0.1, ctx=ctx)
        lse1 = qn.LogSumExp(a, cvec, 0.1, ctx=ctx1)
        e, e1 = lse(xs0), lse1(xs0)
        result["lse_eval_close"] = bool(abs(e.f() - e1.f()) <= 1e-13 * max(1.0, abs(e1.f()))
                                        and np.linalg.norm(e.g() - e1.g()) <= 1e-13 * np.linalg.norm(e1.g()))
        s4 = qn.DFP(1e-10, xs0, ctx=ctx)
        s4.set_trace(12, with_x=True)
        s5 = qn.DFP(1e-10, xs0, ctx=ctx1)
        s5.set_trace(12, with_x=True)
        for sv, ob in ((s4, lse), (s5, lse1)):
            try:
                sv.minimize(qn.MoreThuente(), ob, 12, 20)
            except qn.MaxIterReached:
                pass
        (t4, x4), (t5, x5) = s4.trace(), s5.trace()
        result["lse_dfp_close"] = bool(len(t4) == len(t5) and [r["ls_cases"] for r in t4] == [r["ls_cases"] for r in t5]
                                       and np.linalg.norm(x4 - x5) <= 1e-9 * np.linalg.norm(x5))
    elif mode == "gpu_sym":
        # Row-sharded SYMMETRIC storage: the ranks share the one GPU, every rank streams the circulant half of its own block-rows,
        # partial n-vectors are all-gathered (host-staged here) and summed in rank order.  Against the single-rank run of the same
        # problem: same line-search cases and evaluation counts, iterates to the parity tolerance; between ranks: the same bits.
        result["cases"] = []
        for n in ([1024, 2048, 2040] if world == 2 else [128 * 3 * world]):  # nb = 8, 16 (even), 16 with 8 padding rows; world 3: nb = 9 (odd)
            iters = 12
            diag = P.synth_diag(n)
            b, x0 = P.synth_vectors(n)
            rccl = os.environ.get("QN_TEST_EXCHANGE") == "rccl"  # one GPU per rank and the real collective (multi-GPU boxes only)
            dev = rank if rccl else 0
            ctx = qn.dist.sharded_context(dev, host_exchange=not rccl)
            obj = qn.Quadratic.synthetic(n, P.SEED, diag, b, ctx=ctx)
            ctx1 = qn.Context(dev)
            obj1 = qn.Quadratic.synthetic(n, P.SEED, diag, b, ctx=ctx1)
            case = {"n": n, "rccl": rccl}
            for method in ("bfgs", "dfp"):
                mk = qn.BFGS if method == "bfgs" else qn.DFP
                s = mk(1e-10, x0, ctx=ctx)
                s.set_trace(iters, with_x=True)
                s1 = mk(1e-10, x0, ctx=ctx1)
                s1.set_trace(iters, with_x=True)
                for sv, ob in ((s, obj), (s1, obj1)):
                    try:
                        sv.minimize(qn.MoreThuente(), ob, iters, 20)
                    except qn.MaxIterReached:
                        pass
                (tr, xs), (tr1, xs1) = s.trace(), s1.trace()
                st, st1 = s.stats(), s1.stats()
                ok = len(tr) == len(tr1) == iters
                ok = ok and all((a["ls_cases"], a["n_evals"], a["ls_iters"]) == (c["ls_cases"], c["n_evals"], c["ls_iters"]) for a, c in zip(tr, tr1))
                ok = ok and all(abs(a["t"] - c["t"]) <= 1e-9 * abs(c["t"]) for a, c in zip(tr, tr1))
                ok = ok and bool(np.linalg.norm(xs - xs1) <= 1e-9 * np.linalg.norm(xs1))
                case[method + "_close"] = bool(ok)
                case[method + "_path"] = [st["path"], st1["path"]]
                case[method + "_bytes"] = [st["matrix_bytes_per_pass"], st1["matrix_bytes_per_pass"]]
                case[method + "_x_hex"] = [float(v).hex() for v in xs[-1][:64]]
                # the getter restores the stale halves from the ranks that maintain them: the whole matrix, symmetric, = single rank
                h = s.approx_inv_hessian(all_ranks=True)
                h1 = s1.approx_inv_hessian()
                case[method + "_h_symmetric"] = bool(np.array_equal(h, h.T))
                case[method + "_h_close"] = bool(np.linalg.norm(h - h1) <= 1e-8 * np.linalg.norm(h1))
                case[method + "_h_hex"] = [float(v).hex() for v in h[n // 2, ::17]]
                # ... and the run continues from there: on the symmetric path again, then on the row kernels (tiling -3)
                for rows in (False, True):
                    for sv, ob in ((s, obj), (s1, obj1)):
                        if rows:
                            sv.set_option("symmetric_storage", 0)
                        try:
                            sv.minimize(qn.MoreThuente(), ob, 4, 20)
                        except qn.MaxIterReached:
                            pass
                    xa, xb = s.x(), s1.x()
                    case[method + ("_rows" if rows else "_again") + "_close"] = bool(np.linalg.norm(xa - xb) <= 1e-8 * np.linalg.norm(xb))
                    case[method + ("_rows" if rows else "_again") + "_path"] = s.stats()["path"]
            # rounds 1-3's sharded kernels (tile, sum, exchange, epilogue, control step) stay reachable: set_option("second_generation", 0)
            sg1 = qn.BFGS(1e-10, x0, ctx=ctx)
            sg1.set_option("second_generation", 0)
            sg1.set_trace(iters, with_x=True)
            sr1 = qn.BFGS(1e-10, x0, ctx=ctx1)
            sr1.set_trace(iters, with_x=True)
            for sv, ob in ((sg1, obj), (sr1, obj1)):
                try:
                    sv.minimize(qn.MoreThuente(), ob, iters, 20)
                except qn.MaxIterReached:
                    pass
            (tg1, xg1), (tr1_, xr1) = sg1.trace(), sr1.trace()
            case["gen1_close"] = bool(len(tg1) == len(tr1_) == iters
                                      and all((a["ls_cases"], a["n_evals"]) == (c["ls_cases"], c["n_evals"]) for a, c in zip(tg1, tr1_))
                                      and np.linalg.norm(xg1 - xr1) <= 1e-9 * np.linalg.norm(xr1))
            case["gen1_path"] = sg1.stats()["path"]
            case["gen1_bytes"] = sg1.stats()["matrix_bytes_per_pass"]
            if n == 1024:  # More-Thuente's cases 2-4, the modified-updating switch and tu = +inf on the sharded tiles (mt_workloads.py)
                import mt_workloads as W
                ok_all, digits = True, []
                for name, w in W.WORKLOADS.items():
                    dg, bb, xx0 = W.inputs(n, name)
                    runs = []
                    for cx in (ctx, ctx1):
                        ob = qn.Quadratic.synthetic(n, P.SEED, dg, bb, ctx=cx)
                        sv = qn.BFGS(1e-10, xx0, ctx=cx)
                        if w["h0"] is not None:
                            sv.set_approx_inv_hessian(w["h0"] * np.eye(n))
                        sv.set_trace(w["iters"], with_x=True)
                        lsw = qn.MoreThuente()
                        if w["t_max"] is not None:
                            lsw = lsw.with_t_max(w["t_max"])
                        try:
                            sv.minimize(lsw, ob, w["iters"], 20)
                        except qn.MaxIterReached:
                            pass
                        runs.append(sv.trace())
                    (ta, xa_), (tb, xb_) = runs
                    ok = len(ta) == len(tb) and all((u["ls_cases"], u["n_evals"]) == (v["ls_cases"], v["n_evals"]) for u, v in zip(ta, tb))
                    ok = ok and all(abs(u["t"] - v["t"]) <= W.t_tol(name, v["gnorm"], tb[0]["gnorm"], 1e-9) * abs(v["t"]) for u, v in zip(ta, tb))
                    ok = ok and bool(np.linalg.norm(xa_ - xb_) <= 1e-9 * max(1.0, np.linalg.norm(xb_)))
                    ok_all = ok_all and bool(ok)
                    digits += [d_ for r_ in ta for d_ in W.case_digits(r_["ls_cases"])]
                case["mt_cases_ok"] = bool(ok_all)
                case["mt_digits"] = [digits.count(k_) for k_ in (1, 2, 3, 4)]
            # the GENERIC path (closures, log-sum-exp, SR1, bounded variants) runs its H pass on the same sharded tiles
            gen = {}
            for label, cx, ob in (("sh", ctx, obj), ("one", ctx1, obj1)):
                sg = qn.DFP(1e-10, x0, ctx=cx)
                sg.set_option("generic_kernels", 1)
                sg.set_trace(8, with_x=True)
                try:
                    sg.minimize(qn.MoreThuente(), ob, 8, 20)
                except qn.MaxIterReached:
                    pass
                gen[label] = (sg.trace(), sg.stats()["path"], sg.stats()["matrix_bytes_per_pass"])
            (tg, xg), (t1, x1) = gen["sh"][0], gen["one"][0]
            case["generic_close"] = bool(len(tg) == len(t1) and [r_["ls_cases"] for r_ in tg] == [r_["ls_cases"] for r_ in t1]
                                         and np.linalg.norm(xg - x1) <= 1e-9 * np.linalg.norm(x1))
            case["generic_path"] = [gen["sh"][1], gen["one"][1]]
            case["generic_bytes"] = gen["sh"][2]
            if n == 1024:  # log-sum-exp objective (rows of A sharded) + DFP: config 5's shape in small
                rng = np.random.default_rng(7)
                a_ = rng.standard_normal((300, n)) * (3.0 / np.sqrt(n))
                c_ = rng.standard_normal(300)
                xs0 = rng.standard_normal(n)
                outs_l = []
                # sharded / one rank on the default path (round 5: the second-generation structure, qn_sym2g.hip.h -- trial points
                # exchanged as scalars), then the sharded run on the generic path (set_option("second_generation", 0)), then -- host exchange only --
                # the default path again with the exchange in stream order: pipelined, the same bits as the synchronous pump
                variants = [(ctx, None, False), (ctx1, None, False), (ctx, ("second_generation", 0), False)] + ([] if rccl else [(ctx, None, True)])
                for cx, tiling, asyn in variants:
                    if asyn:
                        ctx.set_host_exchange_async(True)
                    lse = qn.LogSumExp(a_, c_, 0.1, ctx=cx)
                    sl = qn.DFP(1e-10, xs0, ctx=cx)
                    if tiling:
                        sl.configure(*tiling)
                    sl.set_trace(10, with_x=True)
                    try:
                        sl.minimize(qn.MoreThuente(), lse, 10, 20)
                    except qn.MaxIterReached:
                        pass
                    stl = sl.stats()
                    outs_l.append((sl.trace(), stl["path"], stl["oracle_evals"], stl["total_xchg_scalar"], stl["total_xchg_vector"]))
                    if asyn:
                        ctx.set_host_exchange_async(False)
                (tl, xl), (tl1, xl1), (tlg, xlg) = outs_l[0][0], outs_l[1][0], outs_l[2][0]
                case["lse_close"] = bool(len(tl) == len(tl1) and [r_["ls_cases"] for r_ in tl] == [r_["ls_cases"] for r_ in tl1]
                                         and np.linalg.norm(xl - xl1) <= 1e-9 * np.linalg.norm(xl1))
                case["lse_path"] = [outs_l[0][1], outs_l[1][1]]
                case["lse_xchg"] = [outs_l[0][2], outs_l[0][3], outs_l[0][4], len(tl)]  # evaluations, scalar and n-vector collectives, iterations
                case["lse_generic_close"] = bool(len(tlg) == len(tl1) and np.linalg.norm(xlg - xl1) <= 1e-9 * np.linalg.norm(xl1))
                case["lse_generic_path"] = outs_l[2][1]
                if not rccl:
                    case["lse_pipelined_equal"] = bool(outs_l[3][0][0] == tl and np.array_equal(outs_l[3][0][1], xl))
                    case["lse_pipelined_path"] = outs_l[3][1]
            # the same exchange in STREAM ORDER (no synchronisation per exchange): the run is then pipelined -- every kernel and
            # every exchange of an iteration enqueued ahead of the device-side decisions, as with RCCL -- and must give the same bits
            ctx.comm_check()
            s_sync = qn.BFGS(1e-10, x0, ctx=ctx)
            s_sync.set_trace(iters, with_x=True)
            try:
                s_sync.minimize(qn.MoreThuente(), obj, iters, 20)
            except qn.MaxIterReached:
                pass
            # the exchange as an all-reduce of the partial vectors (ncclAllReduce with RCCL; gathered and added in rank order by
            # the host-staged stand-in, hence bit-identical to the default there, tolerance-level with RCCL's own order)
            ctx.set_allreduce(True)
            s_ar = qn.BFGS(1e-10, x0, ctx=ctx)
            s_ar.set_trace(iters, with_x=True)
            try:
                s_ar.minimize(qn.MoreThuente(), obj, iters, 20)
            except qn.MaxIterReached:
                pass
            ctx.set_allreduce(False)
            xa, xb = s_ar.trace()[1], s_sync.trace()[1]
            case["allreduce_ok"] = bool(np.array_equal(xa, xb)) if not rccl else bool(np.linalg.norm(xa - xb) <= 1e-9 * np.linalg.norm(xb))
            case["allreduce_x_hex"] = [float(v).hex() for v in xa[-1][:64]]
            # the trial's partial n-vector riding on the scalar exchange (qn_context_set_trial_vector_exchange, DESIGN 9.1's fallback): the same
            # bits as the default exchange, one collective per accepted iteration fewer -- in this context's mode (host exchange: the
            # synchronous pump; RCCL: pipelined) and, host exchange, in stream order (pipelined: unused evaluation slots re-send the same vector)
            tv = []
            for asyn in ((False,) if rccl else (False, True)):
                if asyn:
                    ctx.set_host_exchange_async(True)
                ctx.set_trial_vector_exchange(True)
                s_tv = qn.BFGS(1e-10, x0, ctx=ctx)
                s_tv.set_trace(iters, with_x=True)
                try:
                    s_tv.minimize(qn.MoreThuente(), obj, iters, 20)
                except qn.MaxIterReached:
                    pass
                ctx.set_trial_vector_exchange(False)
                if asyn:
                    ctx.set_host_exchange_async(False)
                st_tv = s_tv.stats()
                tv.append({"equal": bool(s_tv.trace()[0] == s_sync.trace()[0] and np.array_equal(s_tv.trace()[1], s_sync.trace()[1])),
                           "path": st_tv["path"], "iters": st_tv["iterations"], "evals": st_tv["oracle_evals"],
                           "xchg": [st_tv["total_xchg_vector"], st_tv["total_xchg_scalar"]], "syncs": st_tv["host_syncs"]})
            case["trial_vector"] = tv
            case["default_xchg"] = [s_sync.stats()["total_xchg_vector"], s_sync.stats()["total_xchg_scalar"], s_sync.stats()["iterations"]]
            if rccl:  # RCCL runs are pipelined by default: the synchronous pump is the other mode to compare with
                s_sync2 = qn.BFGS(1e-10, x0, ctx=ctx)
                s_sync2.set_sync_mode(1)
                s_sync2.set_trace(iters, with_x=True)
                try:
                    s_sync2.minimize(qn.MoreThuente(), obj, iters, 20)
                except qn.MaxIterReached:
                    pass
                case["pipelined_path"] = [s_sync2.stats()["path"], s_sync.stats()["path"]]
                case["pipelined_equal"] = bool(s_sync2.trace()[0] == s_sync.trace()[0] and np.array_equal(s_sync2.trace()[1], s_sync.trace()[1]))
                case["pipelined_syncs"] = [s_sync2.stats()["host_syncs"], s_sync.stats()["host_syncs"]]
                outs = []
                for sync in (1, 0):
                    sr = qn.BFGS(1e-10, x0, ctx=ctx)
                    sr.set_option("symmetric_storage", 0)
                    sr.set_sync_mode(sync)
                    sr.set_trace(iters, with_x=True)
                    try:
                        sr.minimize(qn.MoreThuente(), obj, iters, 20)
                    except qn.MaxIterReached:
                        pass
                    outs.append((sr.trace()[1], sr.stats()["path"]))
                case["rows_pipelined_equal"] = bool(np.array_equal(outs[0][0], outs[1][0]))
                case["rows_pipelined_path"] = [outs[0][1], outs[1][1]]
                result["cases"].append(case)
                continue
            ctx.set_host_exchange_async(True)
            ctx.comm_check()
            s_pipe = qn.BFGS(1e-10, x0, ctx=ctx)
            s_pipe.set_trace(iters, with_x=True)
            try:
                s_pipe.minimize(qn.MoreThuente(), obj, iters, 20)
            except qn.MaxIterReached:
                pass
            case["pipelined_path"] = [s_sync.stats()["path"], s_pipe.stats()["path"]]
            case["pipelined_equal"] = bool(s_sync.trace()[0] == s_pipe.trace()[0] and np.array_equal(s_sync.trace()[1], s_pipe.trace()[1]))
            case["pipelined_syncs"] = [s_sync.stats()["host_syncs"], s_pipe.stats()["host_syncs"]]
            # ... with a line search whose evaluations per iteration vary (backtracking from a scaled-down H: several halvings at
            # first, one evaluation later): the pipelined pattern is sized to what the run needs and iterations roll over into the
            # next period when it is too short -- the same bits as the synchronous pump, and as many collectives as it enqueues
            bt = {}
            for asyn in (False, True):
                ctx.set_host_exchange_async(asyn)
                sb = qn.DFP(1e-10, x0, ctx=ctx)
                sb.set_approx_inv_hessian(8.0 * np.eye(n))
                sb.set_trace(iters, with_x=True)
                try:
                    sb.minimize(qn.BackTracking(1e-4, 0.5), obj, iters, 20)
                except qn.MaxIterReached:
                    pass
                stb = sb.stats()
                bt[asyn] = (sb.trace(), stb["path"], stb["oracle_evals"], stb["total_xchg_scalar"], stb["total_xchg_vector"])
            case["bt_pipelined_equal"] = bool(bt[False][0][0] == bt[True][0][0] and np.array_equal(bt[False][0][1], bt[True][0][1]))
            case["bt_paths"] = [bt[False][1], bt[True][1]]
            case["bt_evals"] = [bt[False][2], bt[True][2]]
            case["bt_xchg"] = [[bt[False][3], bt[False][4]], [bt[True][3], bt[True][4]]]
            case["bt_evals_per_iteration_max"] = max(r_["n_evals"] for r_ in bt[False][0][0])
            # ... and on the row kernels (grouped exchanges)
            outs = []
            for asyn in (False, True):
                ctx.set_host_exchange_async(asyn)
                sr = qn.BFGS(1e-10, x0, ctx=ctx)
                sr.set_option("symmetric_storage", 0)
                sr.set_trace(iters, with_x=True)
                try:
                    sr.minimize(qn.MoreThuente(), obj, iters, 20)
                except qn.MaxIterReached:
                    pass
                outs.append((sr.trace()[1], sr.stats()["path"]))
            case["rows_pipelined_equal"] = bool(np.array_equal(outs[0][0], outs[1][0]))
            case["rows_pipelined_path"] = [outs[0][1], outs[1][1]]
            ctx.set_host_exchange_async(False)
            result["cases"].append(case)
    gathered = [None] * world
    dist.all_gather_object(gathered, result)
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(gathered, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
