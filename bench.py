#!/usr/bin/env python3
"""bench.py -- BFGS iterations/sec on the synthetic n-dim convex quadratic (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one BFGS iteration (direction, More-Thuente line search, rank-2 inverse-Hessian update) on the
device-resident objective f = 1/2 x'Qx - b'x; Q, H, and every vector are in HBM before the timed region.
N = 1 runs BASELINE.json configs[1] (n = 4096; one rank streams only the symmetric half of H and Q);
N > 1 runs configs[2]'s problem (n = 32768) with the rows of H and Q sharded over the N ranks; every rank streams the circulant
half of its own block-rows (symmetric storage, the same kernels as on one GPU); per iteration the ranks exchange 8 KB of scalars
per evaluation and the partial n-vectors of the accepted point and of the update pass over RCCL ("scaling": "strong" over
N = 2,4,8).  If RCCL cannot be brought up the line carries no value and the run fails: nothing else is a measurement.
Timing: 5 regions, each = reset to (x0, H = I), W untimed warm-up iterations, exactly K timed iterations
between barrier + synchronize brackets, max over ranks; the MEDIAN region is reported (all five are in `timing.region_ms`).
Prints ONE JSON line on rank 0.  The product path is libqn_hip.so (hand-written gfx950 kernels); the CPU
oracle is used only for the `cpu_baseline` leg.  No GPU => hard failure, never a fallback.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED = 0x5EED0001
KAPPA = 1.0e3
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md:36); 6290 GB/s measured-achievable
METRIC = "BFGS iterations/sec on n-dim convex quadratic (f64) at 1/2/4/8 MI355X"


def synth_inputs(n):
    diag = KAPPA ** (np.arange(n, dtype=np.float64) / max(n - 1, 1))
    rng = np.random.Generator(np.random.Philox(key=SEED))
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    return diag, b, x0


def run_iterations(qn, solver, ls, obj, x0, iters):
    """Exactly `iters` BFGS iterations; if the run converges first, restart from (x0, H = I) and keep counting
    (SURVEY.md 8(d)).  Returns the number of restarts."""
    done, restarts = 0, 0
    while done < iters:
        try:
            solver.minimize(ls, obj, iters - done, 20)
            done += solver.k()
            if done < iters:  # Ok(()) before the cap: converged
                solver.reset(x0)
                restarts += 1
        except qn.MaxIterReached:
            done += solver.k()
    return restarts


def timed_regions(run_exact, stats, synchronize, steps, nregions):
    """`nregions` timed regions of exactly `steps` iterations each (run_exact(k) runs k iterations, restarting a run that converges, and
    returns the number of restarts).  What a region did is DIFFERENCED from the solver's cumulative counters, not assumed: round 5's config-5
    leg divided by `steps` while a converged run had done fewer.  Raises when a region's count is not `steps`."""
    out = []
    for _ in range(nregions):
        synchronize()
        st0 = stats()
        t0 = time.perf_counter()
        restarts = run_exact(steps)
        synchronize()
        dt = time.perf_counter() - t0
        st1 = stats()
        its = st1["total_iterations"] - st0["total_iterations"]
        if its != steps:
            raise RuntimeError(f"a timed region ran {its} iterations, not the {steps} it is quoted for")
        out.append({"s": dt, "iterations": its, "s_per_iteration": dt / its, "restarts": restarts,
                    "evaluations": st1["total_oracle_evals"] - st0["total_oracle_evals"]})
    return out


def collectives_share(xprobe, scalars_per_it, vectors_per_it, ms_per_step):
    """What the probed exchange latencies and the COUNTED collectives of the timed region say about where an iteration's time goes (N > 1): the
    n-vector collectives of an iteration alternate between n doubles (q of the accepted point) and 2 n doubles ([u, v] of the update pass).  An
    estimate -- probed latencies x counts, slowest rank -- printed beside the measured ms_per_step; it answers DESIGN section 5's open question
    (is a small RCCL collective ~20 us?) the first time the line is printed on a multi-GPU node."""
    out = {"exchange_probe": xprobe, "collectives_per_iteration": {"scalars_8KB": scalars_per_it, "n_vectors": vectors_per_it}}
    if isinstance(xprobe, dict) and xprobe.get("trial_vector_exchange", {}).get("used"):
        # (the trial's partial vector rode on every "scalar" collective: those moved 8 KB + n doubles per rank, the n-vector ones are the update pass's 2 n)
        out["collectives_per_iteration"] = {"scalars_8KB_plus_n_vector": scalars_per_it, "two_n_vectors": vectors_per_it}
    try:
        ranks = [r for r in xprobe["per_rank"] if isinstance(r, dict) and "error" not in r]
        if not ranks:
            raise ValueError("no rank probed")
        worst = {k: max(r[k]["median_us"] for r in ranks) for k in ("scalars_8KB", "n_vector", "two_n_vectors")}
        if "two_n_vectors" in out["collectives_per_iteration"]:
            est_us = scalars_per_it * max(worst["scalars_8KB"], worst["n_vector"]) + vectors_per_it * worst["two_n_vectors"]
        else:
            est_us = scalars_per_it * worst["scalars_8KB"] + 0.5 * vectors_per_it * (worst["n_vector"] + worst["two_n_vectors"])
        out.update({"slowest_rank_median_us": worst, "estimated_collective_us_per_iteration": est_us,
                    "estimated_fraction_of_ms_per_step": est_us / (1e3 * ms_per_step) if ms_per_step else None,
                    "design_budget_us_per_small_collective": 20.0})
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)
    return out


def totals(solver):
    """Counters that accumulate over every qn_minimize call of the solver (the per-call ones restart with k)."""
    st = solver.stats()
    return {k: st[k] for k in ("total_minimize_calls", "total_iterations", "total_oracle_calls", "total_oracle_evals", "total_h_passes",
                               "total_h_bytes", "total_obj_bytes", "launches", "host_syncs", "total_xchg_vector", "total_xchg_scalar")}


def host_cpu_share():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU quota when there is one.  (The GPU boxes
    show all 256 hardware threads of the host to a container whose quota is 16 CPUs: an OpenMP team of 256 pinned threads on that
    share ran the port at 1.0 it/s -- measured, round 4 -- where 16 threads reach ~100.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.999)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.999)))
            break
        except Exception:  # noqa: BLE001
            continue
    return max(1, n)


def cpu_baseline(n, iters, extra=True):
    """The CPU legs in a FRESH CHILD PROCESS, run to completion BEFORE this process touches the GPU (round 5).
    Round 3 ran the leg in the bench process, where `import torch` had already started an OpenMP runtime (41.7 it/s where the same
    code reached 87-137); round 4 moved it to a child that ran AFTER the GPU measurement, beside a parent that still held a GPU
    context, its runtime threads and 250 MB of pinned memory: 269-457 it/s inside full runs against 499-515 standalone.  Now the
    parent starts the child first thing -- before torch, HIP or the library are loaded: the parent is one thread blocked in
    wait() -- with the pinning in the child's environment before any OpenMP runtime exists; the child takes several sub-samples
    per leg (five for the headline's, round 6) and reports all of them and their median.  Returns {"config2": {...}, "config4": {...}, "config5": {...}} or an
    {"error": ...} object (a failing CPU leg never loses the bench line: ADVICE r4)."""
    env = dict(os.environ)
    # spread: the team's threads one per core, as far apart as the places allow -- on the GPU boxes' 2 x 64-core hosts each of the 16
    # threads of the container's share then has a core complex (and its 32 MB of L3) to itself: 499-514 it/s in three consecutive
    # runs, against 159 +- 0.5 with `close` (16 neighbouring cores) and 345-410 unpinned (round 4, n = 4096)
    env.setdefault("OMP_PROC_BIND", "spread")
    env.setdefault("OMP_PLACES", "cores")
    threads = host_cpu_share()
    env["OMP_NUM_THREADS"] = str(threads)  # (torch.distributed.run exports OMP_NUM_THREADS=1 to its workers; the host shows more CPUs than the share)
    try:
        cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(n), str(iters), str(threads), "1" if extra else "0"],
                            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if cp.returncode != 0:
            return {"error": f"cpu_baseline child failed (rc {cp.returncode}): {cp.stderr[-400:]}"}
        out = json.loads(cp.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001 -- timeout, empty output, a line that is not JSON
        return {"error": "cpu_baseline child: " + repr(e)}
    proc = ("fresh child process run to completion before the bench process loaded torch / HIP, %d threads = this container's CPU share, "
            "OMP_PROC_BIND=%s OMP_PLACES=%s" % (threads, env["OMP_PROC_BIND"], env["OMP_PLACES"]))
    for leg in out.values():
        if isinstance(leg, dict):
            leg["process"] = proc
    return out


def _median(v):
    v = sorted(v)
    return v[len(v) // 2]


def cpu_leg_config2(qo, n, iters, threads):
    """CPU port timed on the host cores (CPU-B): the oracle restatement with the O(n^2) rank-2 update (the
    reference's literal update is O(n^3): 2.7e11 flop per iteration at n = 4096), reference call sequence (5 oracle calls per
    iteration), OpenMP over all cores.  Every sweep is contiguous per thread -- the symmetric H and Q are read by rows = columns,
    four rows per thread in flight, pages first touched by the thread that streams them -- so the port is bandwidth-bound and
    the achieved GB/s is printed next to the core count.  FIVE sub-samples of ~2 s each (round 6, VERDICT r5 item 9: three moved 18 % between
    runs on a shared host); value = their median, with the minimum and the maximum beside it.  (Q's pages are first touched by the threads that
    sweep them -- qo_synth_fill_rows follows the evaluation's partition --, H's by qo_solver_create.)"""
    diag, b, x0 = synth_inputs(n)
    q = qo.synth_rows(n, 0, n, SEED, diag, nthreads=threads)
    o = qo.QuadraticOracle(q, b, nthreads=threads)
    # One run stays inside the pre-convergence window (the restatement needs ~350 iterations on this family; past convergence
    # y's -> 0 and the timings mean nothing), so a sub-sample is a series of runs of `per_run` iterations, each from (x0, H = I) after
    # 3 untimed iterations -- the GPU leg's protocol.  Solver creation is not timed.
    per_run = int(min(max(iters, 30), 250))
    subs, tot_k, tot_dt, tot_moved, tot_runs = [], 0, 0.0, 0.0, 0
    for sub in range(5):
        k, dt, moved, runs = 0, 0.0, 0.0, 0
        while dt < 2.0 and runs < 200:
            s = qo.Solver(qo.BFGS, 1e-10, x0, qo.UPDATE_RANK2, nthreads=threads)
            tw = time.perf_counter()
            s.minimize(qo.morethuente(), o, 3, 20)  # warm the caches / thread pool
            tw = (time.perf_counter() - tw) / 3.0
            if sub == 0 and runs == 0 and tw * per_run > 2.0:  # a host far slower than planned for: keep the whole sample near 10-30 s whatever it is
                per_run = max(3, int(2.0 / tw))
            bytes0, calls0 = s.bytes_streamed, o.calls
            t0 = time.perf_counter()
            s.minimize(qo.morethuente(), o, per_run, 20)  # warm restart: continues from the warmed-up state
            dt += time.perf_counter() - t0
            k += s.k
            moved += (s.bytes_streamed - bytes0) + (o.calls - calls0) * 8.0 * n * n
            runs += 1
            del s
        subs.append({"value": k / dt, "iterations": k, "seconds": dt, "achieved_GBs": moved / dt / 1e9})
        tot_k += k; tot_dt += dt; tot_moved += moved; tot_runs += runs
    val = _median([u["value"] for u in subs])
    out = {"value": val, "unit": "iterations/s", "cores": threads, "kind": "port",
           "sub_samples": [u["value"] for u in subs], "sub_sample_spread": (max(u["value"] for u in subs) - min(u["value"] for u in subs)) / val,
           "sub_sample_min": min(u["value"] for u in subs), "sub_sample_median": val, "sub_sample_max": max(u["value"] for u in subs),
           "achieved_GBs": _median([u["achieved_GBs"] for u in subs]), "bytes_per_iteration": tot_moved / max(tot_k, 1),
           "numa_nodes": len(glob.glob("/sys/devices/system/node/node[0-9]*")) or None,
           "bytes_note": "matrix bytes the port streams: 8 n^2 per oracle call (full Q by rows), 8 n^2 per mat-vec with H (u = H y, "
                         "d = -H g), 16 n^2 for the rank-2 update",
           "sample": f"median of 5 sub-samples; in all {tot_k} BFGS+MoreThuente iterations at n={n} in {tot_runs} runs of {per_run} from (x0, H = I) (same Q, b, x0 as the GPU run), "
                     f"rank-2 O(n^2) update, reference oracle-call sequence (5 calls per iteration), OpenMP x{threads}, {tot_dt:.1f} s"}
    # the reference's own formulation (dense n x n products, single thread as matrixmultiply is built) at a size it finishes
    n_small = 384
    d2, b2, x2 = synth_inputs(n_small)
    q2 = qo.synth_rows(n_small, 0, n_small, SEED, d2)
    s2 = qo.Solver(qo.BFGS, 1e-10, x2, qo.UPDATE_AS_WRITTEN, nthreads=1)
    o2 = qo.QuadraticOracle(q2, b2, nthreads=1)
    t0 = time.perf_counter()
    s2.minimize(qo.morethuente(), o2, 4, 20)
    dt2 = time.perf_counter() - t0
    per_iter = dt2 / max(s2.k, 1)
    out["as_written_1thread"] = {"n": n_small, "s_per_iteration": per_iter,
                                 "extrapolated_s_per_iteration_at_n": per_iter * (n / n_small) ** 3,
                                 "note": "bfgs.rs:115-124 literally (O(n^3)), naive loops, 1 thread; extrapolated with n^3"}
    return out


NEWTON_N = 8192          # BASELINE.json configs[3]
LSE_N = 16384            # BASELINE.json configs[4]
LSE_MU, LSE_A_SCALE, LSE_SEED = 0.1, 2.0, 11


def lse_inputs(n):
    """The config-5 instance of tools/bench_config5.py and tests/test_gpu_logsumexp.py: A ~ N(0, (2 / sqrt n)^2), c, x0 ~ N(0, 1), mu = 0.1."""
    rng = np.random.default_rng(LSE_SEED)
    a = rng.standard_normal((n, n)) * (LSE_A_SCALE / np.sqrt(n))
    c = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    return a, c, x0


def cpu_leg_config4(qo, n_full):
    """Newton (newton/mod.rs:26-49) as the reference runs it: `try_inverse` = LU with partial pivoting + the explicit inverse, 2 n^3
    flop, ONE thread (nalgebra's LU is not threaded) -- at n = 8192 that is 1.1e12 flop, minutes.  Bounded sample: the same
    restatement on the same synthetic family at n = 768, three runs, extrapolated with n^3 (stated in `sample`)."""
    n = 768
    diag, b, x0 = synth_inputs(n)
    q = qo.synth_rows(n, 0, n, SEED, diag)
    per = []
    for _ in range(3):
        oq = qo.QuadraticOracle(q, b)
        s = qo.Solver(qo.NEWTON, 1e-8, x0)
        s.set_hessian(oq)
        t0 = time.perf_counter()
        s.minimize(qo.morethuente(), oq, 10, 20)
        dt = time.perf_counter() - t0
        per.append(dt / max(s.k, 1))
        its = s.k
    sec = _median(per) * (n_full / n) ** 3
    return {"value": 1.0 / sec, "unit": "Newton iterations/s", "cores": 1, "kind": "port",
            "s_per_iteration_extrapolated": sec, "sub_samples_s_per_iteration_at_n768": per,
            "sample": f"oracle Newton (LU + explicit inverse as nalgebra's try_inverse, 1 thread) on the synthetic quadratic at n={n}: {its} iterations "
                      f"per run, 3 runs, median {1e3 * _median(per):.0f} ms per iteration, extrapolated with n^3 to n={n_full}"}


def cpu_leg_config5(qo, n_full, threads):
    """DFP + More-Thuente on the log-sum-exp objective: the oracle's port (rank-2 O(n^2) update, threaded row sweeps of A and H).  At
    n = m = 16384 the port needs 1-2 s per iteration on 16 cores (A and H are 2 GiB each), so the bounded sample is the same family at
    n = m = 8192 -- five sub-samples continuing one run -- scaled with n^2 (every sweep of an iteration is over an n x n or m x n
    matrix: the port is bandwidth-bound); stated in `sample`."""
    n = 8192
    a, c, x0 = lse_inputs(n)
    o = qo.LogSumExpOracle(a, c, LSE_MU, nthreads=threads)
    s = qo.Solver(qo.DFP, 1e-10, x0, qo.UPDATE_RANK2, nthreads=threads)
    tw = time.perf_counter()
    s.minimize(qo.morethuente(), o, 2, 20)
    tw = (time.perf_counter() - tw) / 2.0
    per_sub = max(2, min(40, int(1.0 / max(tw, 1e-3))))
    subs = []
    for _ in range(5):
        t0 = time.perf_counter()
        s.minimize(qo.morethuente(), o, per_sub, 20)  # (a continued call: k restarts, the state does not)
        dt = time.perf_counter() - t0
        subs.append(s.k / dt)
    scale = (n / n_full) ** 2
    val = _median(subs) * scale
    return {"value": val, "unit": "iterations/s", "cores": threads, "kind": "port", "sub_samples_at_n8192": subs,
            "sub_sample_spread": (max(subs) - min(subs)) / _median(subs),
            "sub_sample_min_median_max_at_n8192": [min(subs), _median(subs), max(subs)],
            "sample": f"oracle DFP + MoreThuente on the log-sum-exp family of the GPU leg at n=m={n} (A ~ N(0, (2/sqrt n)^2), mu={LSE_MU}), rank-2 O(n^2) update, "
                      f"OpenMP x{threads}: 5 sub-samples of {per_sub} iterations continuing one run after 2 warm-up iterations, median "
                      f"{_median(subs):.2f} it/s, scaled with (n/{n_full})^2 to n=m={n_full}"}


def cpu_baseline_child(n, iters, threads=None, extra=True):
    from oracle import qn_oracle as qo  # (the first OpenMP runtime of this process: the pinning is in the environment already)
    threads = min(qo.max_threads(), threads) if threads else min(qo.max_threads(), host_cpu_share())
    out = {}
    legs = [("config2", lambda: cpu_leg_config2(qo, n, iters, threads))]
    if extra:
        legs += [("config4", lambda: cpu_leg_config4(qo, NEWTON_N)), ("config5", lambda: cpu_leg_config5(qo, LSE_N, threads))]
    for name, leg in legs:
        try:
            out[name] = leg()
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)}
    return out


MFMA_F64_PEAK_TFLOPS = 78.6  # dense v_mfma_f64 peak of MI355X (/opt/skills/guides/MI355X_MICROARCH.md); a pure v_mfma_f64_16x16x4_f64 loop sustains 47.8 here


def extra_config4(qn, ctx):
    """BASELINE.json configs[3]: Newton (src/newton/mod.rs) on the n = 8192 synthetic quadratic, dense Hessian solve on the GPU.  The
    SPD Hessian takes the blocked Cholesky (f64 MFMA trailing updates); `lu` forces the pivoted LU the reference's try_inverse
    stands for (what an indefinite or non-symmetric Hessian gets).  One Newton iteration = one factorisation + the sweeps of
    d = -H^-1 g and of the decrement's second solve (newton/mod.rs:38-40) + the line search's evaluations."""
    n = NEWTON_N
    diag, b, x0 = synth_inputs(n)
    obj = qn.Quadratic.synthetic(n, SEED, diag, b, ctx=ctx)
    out = {"workload": f"Newton + MoreThuente::default, n={n} convex quadratic (random SPD Q, kappa=1e3, seed 0x5EED0001), f64, 1xMI355X", "dtype": "f64"}
    for label, force_lu, flops in (("cholesky", False, n ** 3 / 3.0), ("lu", True, 2.0 * n ** 3 / 3.0)):
        wall, its = [], 0
        for rep in range(4):  # (the first repetition allocates the factorisation's buffers)
            s = qn.Newton(1e-8, x0, ctx=ctx)
            if force_lu:
                s.set_option("newton_pivoted_lu", 1)
            if rep == 3:
                s.set_profiling(True)
            ctx.synchronize()
            t0 = time.perf_counter()
            s.minimize(qn.MoreThuente(), obj, 10, 20)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            its = s.k()
            if rep in (1, 2):
                wall.append(dt / max(its, 1))
            if rep == 3:
                st = s.stats()
                ev_ms = st["t_newton_ms"] / max(st["n_newton_timed"], 1)
                timeouts = st["newton_lu_sync_timeouts"]
            del s
        ms = 1e3 * min(wall)
        tf = flops / (ev_ms * 1e-3) / 1e12
        out[label] = {"metric": "Newton iterations/s", "value": 1e3 / ms, "unit": "iterations/s", "ms_per_iteration": ms, "iterations_per_run": its,
                      "ms_per_iteration_runs": [1e3 * w for w in wall],
                      "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F64_PEAK_TFLOPS,
                                   "traffic": None, "flops_per_direction": flops, "avg_direction_ms": ev_ms,
                                   "kernel": ("chol_syrk_kernel (depth-256 trailing updates on v_mfma_f64_16x16x4_f64) beside the diagonal-block / panel chain" if not force_lu
                                              else "lu_panel_persist_kernel (one CU's dependent pivot chain) beside lu_gemm2_kernel (MFMA trailing update) on the masked stream"),
                                   "note": "factorisation flops (n^3/3 Cholesky, 2n^3/3 LU) / HIP-event time of one direction request (staging + factorisation + the "
                                           "substitution sweeps) on the solver's stream; the chain of small dependent kernels, not the MFMA rate, bounds both (DESIGN.md 8 f2)"},
                      "lu_sync_timeouts": timeouts}
    return out


def extra_bounded(qn, ctx):
    """SURVEY 8 row f4, driver-visible: BFGSB + MoreThuenteB (bfgs_b.rs, morethuente_b.rs) against BFGS + MoreThuente on the
    headline workload -- n = 4096, a box of which a quarter of the bounds is active at the constrained optimum (the box of tools/bench_bounded.py
    and tests/test_gpu_bounded.py; the unconstrained optimum it is built around comes from two Newton iterations on the GPU, not from the oracle).
    Wall time per iteration of 200 iterations each, after 5 untimed ones, and their ratio."""
    n = 4096
    diag, b, x0 = synth_inputs(n)
    obj = qn.Quadratic.synthetic(n, SEED, diag, b, ctx=ctx)
    nt = qn.Newton(1e-12, x0, ctx=ctx)
    try:
        nt.minimize(qn.MoreThuente(), obj, 3, 20)
    except qn.MaxIterReached:
        pass
    xs = np.array(nt.x(), dtype=np.float64)
    del nt
    lb = xs - 0.3 * np.abs(xs) - 0.05
    ub = xs + 0.1
    k4 = max(1, n // 4)
    lb[:k4] = xs[:k4] + 0.2
    ub[:k4] = xs[:k4] + 1.0
    out = {"workload": f"BFGSB + MoreThuenteB against BFGS + MoreThuente, n={n} convex quadratic (the headline's), a quarter of the box active at the constrained optimum, f64, 1xMI355X"}
    iters = 200
    for name, mk, mkls in (("unbounded", lambda: qn.BFGS(1e-10, x0, ctx=ctx), lambda: qn.MoreThuente()),
                           ("bounded", lambda: qn.BFGSB.new(1e-10, x0, lb, ub, ctx=ctx), lambda: qn.MoreThuenteB.new(n).with_lower_bound(lb).with_upper_bound(ub)),
                           # round 6: BackTrackingB on the second-generation path (projection inside s2_evalr_kernel<true>) against plain BackTracking
                           ("unbounded_backtracking", lambda: qn.BFGS(1e-10, x0, ctx=ctx), lambda: qn.BackTracking(1e-4, 0.5)),
                           ("bounded_backtracking_b", lambda: qn.BFGSB.new(1e-10, x0, lb, ub, ctx=ctx), lambda: qn.BackTrackingB.new(1e-4, 0.5, lb, ub))):
        s = mk()
        ls = mkls()

        def run(k):
            try:
                s.minimize(ls, obj, k, 20)
            except qn.MaxIterReached:
                pass
        run(5)
        ctx.synchronize()
        t0 = time.perf_counter()
        run(iters)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = s.stats()
        its = max(int(st["iterations"]), 1)
        out[name] = {"iterations_per_s": its / dt, "us_per_iteration": 1e6 * dt / its, "iterations": its, "path": st["path"],
                     "second_generation_path": bool(st["path"] & 16), "launches_per_iteration": st["launches"] / its, "host_syncs": st["host_syncs"],
                     "evaluations_per_iteration": st["oracle_evals"] / its}
        del s
    out["ratio_bounded_over_unbounded_time"] = out["bounded"]["us_per_iteration"] / out["unbounded"]["us_per_iteration"]
    out["ratio_backtracking_b_over_backtracking_time"] = out["bounded_backtracking_b"]["us_per_iteration"] / out["unbounded_backtracking"]["us_per_iteration"]
    out["backtracking_b_note"] = ("a BackTrackingB iteration evaluates every trial at a PROJECTED point (projected inside the evaluation kernel at this size) and then the accepted "
                                  "x + t d once more, unprojected (bfgs_b.rs:91-98): one evaluation more per iteration than BackTracking, whose accepted trial IS the next iterate")
    out["value"] = out["bounded"]["iterations_per_s"]
    out["unit"] = "iterations/s"
    return out


def extra_config5(qn, ctx):
    """BASELINE.json configs[4] on one GPU (the config names 4: the driver's scaling run has no slot for it): DFP + More-Thuente on the
    n = m = 16384 log-sum-exp objective f = log sum_i exp(a_i'x + c_i) + mu/2 ||x||^2 (dfp.rs:78-123, morethuente.rs:165-297).  A (2 GiB)
    is generated on the host and uploaded once; the timed region has everything resident."""
    n = LSE_N
    a, c, x0 = lse_inputs(n)
    obj = qn.LogSumExp(a, c, LSE_MU, ctx=ctx)
    del a
    s = qn.DFP(1e-10, x0, ctx=ctx)
    ls = qn.MoreThuente()
    run_iterations(qn, s, ls, obj, x0, 3)  # warm-up (the regions below continue this run; a run that converges restarts from (x0, H = I))
    steps = 20
    regions = timed_regions(lambda k: run_iterations(qn, s, ls, obj, x0, k), s.stats, ctx.synchronize, steps, 3)
    med = sorted(regions, key=lambda r: r["s_per_iteration"])[1]
    # the profiling pass: a LIVE solver -- reset, warmed up again, exactly 8 iterations with restarts as above -- so that every class of launch is timed
    # (round 5 profiled a solver that had converged: one loop-top evaluation and nine predicated-off launches, a `frac` of 2.87)
    s.reset(x0)
    run_iterations(qn, s, ls, obj, x0, 3)
    s.set_profiling(True)
    p0 = s.stats()
    run_iterations(qn, s, ls, obj, x0, 8)
    p1 = s.stats()
    s.set_profiling(False)
    prof_iters = p1["total_iterations"] - p0["total_iterations"]
    n_h, n_e = p1["n_hpass_timed"] - p0["n_hpass_timed"], p1["n_eval_timed"] - p0["n_eval_timed"]
    ms_h = (p1["t_hpass_ms"] - p0["t_hpass_ms"]) / n_h if n_h else None
    ms_e = (p1["t_eval_ms"] - p0["t_eval_ms"]) / n_e if n_e else None
    n_c, n_r = p1["n_ereduce_timed"] - p0["n_ereduce_timed"], p1["n_hreduce_timed"] - p0["n_hreduce_timed"]
    ms_c = (p1["t_ereduce_ms"] - p0["t_ereduce_ms"]) / n_c if n_c else None
    ms_r = (p1["t_hreduce_ms"] - p0["t_hreduce_ms"]) / n_r if n_r else None
    alg_h, alg_e = 2.0 * p1["matrix_bytes_per_pass"], 8.0 * n * n
    e_per_it = med["evaluations"] / med["iterations"]
    b_iter = alg_h + e_per_it * alg_e
    if not (ms_h and ms_e):
        raise RuntimeError("the profiling pass timed no launch of the pass over A or of the update pass")
    if n_e < prof_iters or n_h < prof_iters - 1:
        raise RuntimeError(f"the profiling pass timed {n_e} evaluation and {n_h} update-pass launches for {prof_iters} iterations: not a live run")
    dom_eval = n_e * ms_e > n_h * ms_h
    ach = (alg_e / (ms_e * 1e-3) if dom_eval else alg_h / (ms_h * 1e-3)) / 1e9
    for what, v in (("roofline", ach), ("objective_eval", alg_e / (ms_e * 1e-3) / 1e9), ("update_pass", alg_h / (ms_h * 1e-3) / 1e9)):
        if v > HBM_PEAK_GBS:
            raise RuntimeError(f"config 5: {what} would read {v:.0f} GB/s, above the {HBM_PEAK_GBS:.0f} GB/s roofline -- the accounting is wrong, refusing to print it")
    dt_it = med["s_per_iteration"]
    path = p1["path"]
    return {"workload": f"DFP + MoreThuente::default, n=m={n} log-sum-exp (mu={LSE_MU}, A ~ N(0, ({LSE_A_SCALE:g}/sqrt n)^2), seed {LSE_SEED}), f64, 1xMI355X "
                        "(BASELINE.json configs[4] names 4 GPUs: tests/test_gpu_partitions.py runs that partition)",
            "metric": "DFP iterations/s", "value": 1.0 / dt_it, "unit": "iterations/s", "ms_per_iteration": 1e3 * dt_it, "steps": steps, "dtype": "f64",
            "region_ms": [1e3 * r["s"] for r in regions], "region_iterations": [r["iterations"] for r in regions],
            "region_ms_per_iteration": [1e3 * r["s_per_iteration"] for r in regions], "restarts_after_convergence": sum(r["restarts"] for r in regions),
            "evaluations_per_iteration": e_per_it,
            "algorithmic_bytes_per_iteration": b_iter, "whole_iteration_hbm_frac": b_iter / dt_it / (HBM_PEAK_GBS * 1e9),
            "launches_per_iteration": (p1["launches"] - p0["launches"]) / max(prof_iters, 1), "profiled_iterations": prof_iters,
            "kernels": ("second-generation structure (qn_sym2g.hip.h): the state machine in one-workgroup launches on the device, pipelined; per iteration "
                        "E x (machine, pass over A, combine + staged vectors) + (machine, update tiles, reduce)" if path & 16 else "generic path (synchronous)"),
            "combine_avg_launch_ms": ms_c, "update_reduce_avg_launch_ms": ms_r,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "kernel": ("lse_onepass_kernel (f and the gradient in ONE pass over A: running-maximum softmax)" if dom_eval else
                                    "the update pass over the symmetric half of H (pending rank-2 update + H+ [y, g+])"),
                         "dominant_by": "largest share of the GPU time (launches x average HIP-event duration, synchronous profiling pass)",
                         "objective_eval": {"algorithmic_bytes_per_launch": alg_e, "avg_launch_ms": ms_e, "launches_timed": n_e,
                                            "achieved": alg_e / (ms_e * 1e-3) / 1e9 if n_e else None},
                         "update_pass": {"algorithmic_bytes_per_launch": alg_h, "avg_launch_ms": ms_h, "launches_timed": n_h,
                                         "achieved": alg_h / (ms_h * 1e-3) / 1e9 if n_h else None}}}


def main():
    if len(sys.argv) in (4, 5, 6) and sys.argv[1] == "--cpu-baseline-child":  # (no torch, no GPU: see cpu_baseline)
        print(json.dumps(cpu_baseline_child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) >= 5 else None,
                                            (sys.argv[5] != "0") if len(sys.argv) >= 6 else True)), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dim", type=int, default=None, help="override the problem dimension n")
    ap.add_argument("--ls", default="mt", choices=["mt", "bt"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="N = 1: skip the `extra_configs` legs (BASELINE.json configs[3] Newton n=8192 and configs[4] DFP + log-sum-exp n=16384)")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-scaling-ref", action="store_true",
                    help="N > 1: skip the single-GPU run of the same workload on rank 0 (strong-scaling denominator)")
    ap.add_argument("--tiling", default=None, help="rows_per_block,col_splits")
    ap.add_argument("--sync-mode", type=int, default=None)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        if world > 1:
            sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    # The CPU legs run FIRST, in a child, to completion: this process has not loaded torch, HIP or the library yet (see cpu_baseline).
    n_headline = args.dim or (4096 if world == 1 else 32768)
    want_extra = world == 1 and args.dim is None and not args.no_extra_configs
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(n_headline, 200 if n_headline <= 4096 else 4, extra=want_extra)

    import torch  # device plumbing + torch.distributed rendezvous only
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device is visible and there is no CPU fallback")

    import __graft_entry__ as ge
    qn = ge.load_package()

    n = args.dim or (4096 if world == 1 else 32768)
    steps = args.steps if args.steps is not None else (200 if n <= 8192 else 50)
    warmup = args.warmup

    ctx = None
    rccl_error = None
    host_exchange = os.environ.get("QN_BENCH_EXCHANGE", "rccl") == "host"  # harness rehearsal on a 1-GPU box only
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane: gloo (unique-id broadcast, barriers, max over ranks); data plane: RCCL inside libqn_hip.so
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        ndev = torch.cuda.device_count()
        dev = local_rank % max(ndev, 1)
        err = None
        try:
            ctx = qn.dist.sharded_context(dev, host_exchange=host_exchange)
            if host_exchange:
                ctx.set_host_exchange_async(True)  # stream-ordered: the rehearsal runs the pipelined launch logic, as RCCL runs do
            ctx.comm_check()  # verified all-gathers (single and grouped) before any solver state depends on the communicator
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        okf = torch.tensor([0 if err else 1], dtype=torch.int32)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if int(okf.item()) == 0:
            if host_exchange:
                sys.exit(f"rank {rank}: host-staged exchange failed: {err}")
            # RCCL could not be brought up on some rank.  A scaling point through any other exchange is not a measurement of anything
            # north_star names: print the line WITHOUT a value, with the error, and fail.  The host-staged
            # exchange stays strictly behind QN_BENCH_EXCHANGE=host (harness rehearsal on a one-GPU box).
            rccl_error = err or "failed on another rank"
            print(f"[bench rank {rank}] RCCL exchange unavailable: {rccl_error}", file=sys.stderr)
            if rank == 0:
                print(json.dumps({"metric": METRIC, "value": None, "unit": "iterations/s", "n_gpus": world, "steps": steps, "warmup": warmup,
                                  "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                                  "data": "synthetic", "config": {"workload": f"BFGS + MoreThuente::default, n={n} convex quadratic, f64, {world}xMI355X",
                                                                  "exchange": "rccl", "rccl_error": rccl_error},
                                  "error": "the RCCL communicator could not be brought up on every rank: nothing was measured"}), flush=True)
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)
    else:
        ctx = qn.Context(device=local_rank)

    # N > 1, in front of everything that is timed (round 6): what ONE exchange of each size a row-sharded iteration makes costs
    # on this node -- 8 KB of per-workgroup scalars (one per evaluation), n doubles (q of the accepted point), 2 n doubles ([u, v] of the update
    # pass) per rank.  DESIGN section 5 budgets ~20 us for a small one without ever having run one between devices.
    xprobe = None
    if world > 1:
        try:
            xprobe = {"scalars_8KB": ctx.exchange_probe(1024), "n_vector": ctx.exchange_probe(n), "two_n_vectors": ctx.exchange_probe(2 * n)}
        except Exception as e:  # noqa: BLE001 -- the probe never loses the bench line
            xprobe = {"error": repr(e)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, xprobe)  # (every rank's own clock: a slow link shows on the ranks at its ends)
        xprobe = {"ranks": world, "exchange": "host-staged" if host_exchange else "rccl", "per_rank": per_rank}
        # DESIGN 9.1's fallback (round 6: qn_context_set_trial_vector_exchange): every evaluation's collective also carries the trial point's
        # partial n-vector, the accepted point needs no exchange of its own -- it pays when the latency of a collective, not its bytes, is what
        # costs: E (8 KB + n) against E (8 KB) + n per iteration, E ~ 2 -> when an n-vector exchange costs less than two 8 KB ones.  The rule is
        # PRINTED with the probed numbers (every rank sees the same gathered list, so every rank would decide alike); it is APPLIED only on
        # request (QN_BENCH_TRIAL_VECTOR=1, or =auto to follow the rule): the default line measures the default exchange.
        tv = {"used": False, "recommended": None, "rule": "slowest rank's median: n_vector < 2 x scalars_8KB"}
        try:
            ok = [r for r in per_rank if isinstance(r, dict) and "error" not in r]
            if ok and len(ok) == world:
                tv["recommended"] = bool(max(r["n_vector"]["median_us"] for r in ok) < 2.0 * max(r["scalars_8KB"]["median_us"] for r in ok))
            want = os.environ.get("QN_BENCH_TRIAL_VECTOR", "0")
            if want == "1" or (want == "auto" and tv["recommended"]):
                ctx.set_trial_vector_exchange(True)
                tv["used"] = True
        except Exception as e:  # noqa: BLE001
            tv["error"] = repr(e)
        xprobe["trial_vector_exchange"] = tv

    diag, b, x0 = synth_inputs(n)
    obj = qn.Quadratic.synthetic(n, SEED, diag, b, ctx=ctx)  # Q is generated shard-locally on the device
    solver = qn.BFGS(1e-10, x0, ctx=ctx)
    if args.tiling:
        r, c = (int(v) for v in args.tiling.split(","))
        solver.set_tiling(r, c)
    if args.sync_mode is not None:
        solver.set_sync_mode(args.sync_mode)
    ls = qn.MoreThuente() if args.ls == "mt" else qn.BackTracking(1e-4, 0.5)

    def barrier():
        ctx.synchronize()
        if world > 1:
            dist.barrier()

    # median of 5 timed regions.  Each region is the same thing: back to (x0, H = I), `warmup` untimed iterations,
    # then EXACTLY `steps` timed iterations of the same solve between barrier + synchronize brackets (max over ranks).
    regions = 5
    region_s, region_acc = [], []
    for _ in range(regions):
        solver.reset(x0)
        if warmup > 0:
            run_iterations(qn, solver, ls, obj, x0, warmup)
        barrier()
        t_before = totals(solver)
        t0 = time.perf_counter()
        restarts = run_iterations(qn, solver, ls, obj, x0, steps)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        t_after = totals(solver)
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        barrier()
        acc = {k: t_after[k] - t_before[k] for k in t_after}
        acc["restarts"] = restarts
        region_s.append(dt)
        region_acc.append(acc)
    order = sorted(range(regions), key=lambda i: region_s[i])
    mid = order[regions // 2]
    elapsed = region_s[mid]
    acc = region_acc[mid]
    restarts = acc["restarts"]
    evals, calls = acc["total_oracle_evals"], acc["total_oracle_calls"]
    h_bytes, obj_bytes = acc["total_h_bytes"], acc["total_obj_bytes"]
    its = steps / elapsed

    # fixed cost of one qn_minimize call (entry, the first launch from an idle queue, the last launch's report, the return): wall time
    # of warm calls of 4 and of 24 iterations, the same run continued, medians of 7 -- two points of a straight line whose slope is an
    # iteration and whose intercept is the call.  (Rounds 2-3 timed a call that is allowed 0 iterations: that one enqueues a whole
    # period of launches which then all find nothing to do -- 0.06 ms, twice what a call costs inside a timed region.)
    def timed_call(k):
        ts = []
        for _ in range(7):
            solver.reset(x0)
            run_iterations(qn, solver, ls, obj, x0, 5)
            barrier()
            t0 = time.perf_counter()
            extra_calls = run_iterations(qn, solver, ls, obj, x0, k)
            ctx.synchronize()
            if extra_calls:  # the run converged inside the interval: a reset and a second call were timed with it
                return None
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return 1e3 * ts[len(ts) // 2]
    t4, t24 = timed_call(4), timed_call(24)
    per_call_fixed_ms = (t4 - 4.0 * (t24 - t4) / 20.0) if (t4 is not None and t24 is not None) else None  # (a negative intercept is reported as it is)

    # kernel-level roofline: same workload again with every launch bracketed by HIP events on the solver's stream
    roofline = None
    if not args.no_profile_pass:
        prof_steps = min(steps, 50)
        solver.reset(x0)
        solver.set_profiling(True)
        # Second-generation symmetric path: the pass runs PIPELINED, as the timed region does -- the same back-to-back launch
        # pattern, every launch between two events on the solver's stream; the launches that find their request not pending
        # (a few microseconds in the prologue) are left out of the averages by the library (qn_hip.hip, prof_collect).  Round 2
        # bracketed a synchronous pass instead: every kernel started from an idle queue and read 4 us longer than under rocprofv3.
        # The other paths keep the synchronous pass (only real work is launched there).
        pipelined_pass = bool(solver.stats()["path"] & 16) and args.sync_mode is None
        if not pipelined_pass:
            solver.set_sync_mode(1)
        p0 = solver.stats()
        run_iterations(qn, solver, ls, obj, x0, prof_steps)
        p1 = solver.stats()
        solver.set_profiling(False)
        solver.set_sync_mode(args.sync_mode if args.sync_mode is not None else -1)

        def cls(name):
            nn = p1[f"n_{name}_timed"] - p0[f"n_{name}_timed"]
            tt = p1[f"t_{name}_ms"] - p0[f"t_{name}_ms"]
            return nn, (tt / nn if nn else None)
        (n_h, ms_h), (n_e, ms_e), (n_c, ms_c) = cls("hpass"), cls("eval"), cls("ctl")
        (n_hr, ms_hr), (n_er, ms_er) = cls("hreduce"), cls("ereduce")
        # algorithmic bytes of one update-pass launch on this rank: read + write of what the pass streams -- the rank's n/P x n f64
        # shard, or (one rank, symmetric storage) the symmetric half: off-diagonal 128 x 128 tiles + the diagonal tiles' upper triangles
        mat = float(p1["matrix_bytes_per_pass"])
        alg_h = 2.0 * mat
        alg_q = mat
        # An event / launch / event bracket reports the kernel plus a fixed launch / event cost.  It is NOT subtracted: against
        # rocprofv3 on the same run the raw bracket is 2-3 us above the profiler's kernel duration, so `achieved` below is slightly
        # conservative; the empty-kernel calibration is reported next to it for the record.
        bracket_ms = ctx.event_bracket_overhead_ms(200)
        ach_h = alg_h / (ms_h * 1e-3) / 1e9 if n_h else None
        ach_e = alg_q / (ms_e * 1e-3) / 1e9 if n_e else None
        sym_pass = bool(p1["path"] & 2)
        sym2 = bool(p1["path"] & 16)
        hname = ("s2_hpass_kernel<.., SHARD> (pending rank-2 update in place + row and column sums of [y, g+] over this rank's share of the "
                 "symmetric half of H -- the circulant windows of its block-rows; its prologue runs the solver's state machine)" if (sym2 and world > 1) else
                 "s2_hpass_kernel (pending rank-2 update in place + row and column sums of [y, g+] over the symmetric half of H; its "
                 "prologue runs the solver's state machine)" if sym2 else
                 "sym_hpass_tile_kernel (rank-2 update + row and column dots of [y, g+] over the upper block triangle of H)" if (sym_pass and world == 1) else
                 "sym_hpass_tile_kernel (rank-2 update + row and column dots of [y, g+] over this rank's share of the symmetric half of H: "
                 "the circulant windows of its block-rows)" if sym_pass else
                 "h_pass_fused_kernel (fused rank-2 update + 2-RHS mat-vec over the rank's rows of H)")
        ename = ("s2_eval_kernel<.., SHARD> (this rank's share of f(x + t d) and g(x + t d)'d from its windows of Q: per-workgroup scalars, "
                 "exchanged as 8 KB per rank and summed in rank order by the next launch's prologue)" if (sym2 and world > 1) else
                 "s2_evalr_kernel (f(x + t d) and g(x + t d)'d from the symmetric half of Q; round 6: mover waves stream the workgroup's two tiles and "
                 "its sliver, multiplier waves take the first tile's rows out of the LDS park as they land while wave 0's state machine has long "
                 "decided, the movers multiply the second tile out of their registers; launches of odd parity stream the two tiles in the other order -- an "
                 "evaluation right behind another one starts with the tile the XCD's L2 still holds -- and the update-reduce launch in front of an "
                 "iteration's first evaluation carries workgroups that load rows of its first tiles into that L2)" if (sym2 and n == 4096 and os.environ.get("QN_S2_RING", "1") != "0") else
                 "s2_eval_kernel (f(x + t d) and g(x + t d)'d from the symmetric half of Q: first tile parked in LDS while wave 0 runs the "
                 "solver's state machine, items in groups with one exchange)" if sym2 else
                 "sym_eval_tile_kernel (Q (x + t d) from the upper block triangle of Q)" if sym_pass else
                 "quad_eval_fused_kernel (Q_rows (x + t d), trial point and direction formed on the fly)")
        # The `roofline` object is about the DOMINANT kernel: the one with the largest share of the GPU time of the timed pattern
        # (launches x average duration in this pass; the pass launches exactly the work of the timed region, one request at a time).
        # The other streaming kernel, the small reduce launches and a time-weighted figure over all of them are listed beside it.
        t_e, t_h = (n_e * ms_e if n_e else 0.0), (n_h * ms_h if n_h else 0.0)
        t_small = (n_er * ms_er if n_er else 0.0) + (n_hr * ms_hr if n_hr else 0.0)
        t_all = t_e + t_h + t_small
        dom_is_eval = t_e > t_h
        ach, ms_dom, n_dom, alg_dom = (ach_e, ms_e, n_e, alg_q) if dom_is_eval else (ach_h, ms_h, n_h, alg_h)
        tw = ((n_e or 0) * alg_q + (n_h or 0) * alg_h) / (t_all * 1e-3) / 1e9 if t_all > 0 else None
        hsub = {"kernel": hname, "algorithmic_bytes_per_launch": alg_h, "avg_launch_ms": ms_h, "launches_timed": n_h, "achieved": ach_h,
                "frac": (ach_h / HBM_PEAK_GBS) if ach_h else None, "time_share": (t_h / t_all) if t_all else None,
                # the pass as the iteration pays for it: tile launch + its slot-reduction launch (symmetric storage only)
                "pass_with_reduce": ({"avg_ms": ms_h + ms_hr, "achieved": alg_h / ((ms_h + ms_hr) * 1e-3) / 1e9,
                                      "frac": alg_h / ((ms_h + ms_hr) * 1e-3) / 1e9 / HBM_PEAK_GBS, "reduce_avg_launch_ms": ms_hr,
                                      "reduce_launches_timed": n_hr} if (n_h and n_hr) else None),
                # SURVEY.md 8(d) counts 16 n^2 / P bytes per H pass (full matrix read + written); the symmetric-storage path does
                # the same pass on half of them, so its rate in full-matrix terms is higher than the bytes it really moves
                "full_matrix_equivalent_GBs": (16.0 * n * n / world) / (ms_h * 1e-3) / 1e9 if n_h else None}
        esub = {"kernel": ename, "algorithmic_bytes_per_launch": alg_q, "avg_launch_ms": ms_e, "launches_timed": n_e, "achieved": ach_e,
                "frac": (ach_e / HBM_PEAK_GBS) if ach_e else None, "time_share": (t_e / t_all) if t_all else None,
                "accept_reduce_avg_launch_ms": ms_er, "accept_reduce_launches_timed": n_er}
        roofline = {"bound": "hbm", "kernel": ename if dom_is_eval else hname,
                    "dominant_by": "largest share of the GPU time of the timed launch pattern (launches x average duration)",
                    "time_share": ((t_e if dom_is_eval else t_h) / t_all) if t_all else None,
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (ach / HBM_PEAK_GBS) if ach else None,
                    "traffic": None,
                    "traffic_source": None,
                    "achievable_note": "plain read+write streams reach 4.9-5.4 TB/s on this device (hipMemcpy D2D 5.0 TB/s; "
                                       "profiles/r01_c_bw_probe.txt); peak is the 8 TB/s HBM3E spec",
                    "algorithmic_bytes_per_launch": alg_dom, "avg_launch_ms": ms_dom, "launches_timed": n_dom,
                    "all_kernels_time_weighted": {"achieved": tw, "frac": (tw / HBM_PEAK_GBS) if tw else None,
                                                  "note": "algorithmic bytes of every streaming launch / summed durations of ALL launches of the "
                                                          "pass (tile kernels and their reduce launches)"},
                    "update_pass": hsub,
                    "quad_matvec": esub,
                    "event_bracket_fixed_overhead_ms_not_subtracted": bracket_ms,
                    "ctl_step": {"avg_launch_ms": ms_c, "launches_timed": n_c,
                                 "note": ("the state machine runs inside the prologue of the streaming kernels: there is no control "
                                          "launch in a pipelined run (a synchronous pass has one prologue-only launch per request)") if sym2 else None},
                    "note": "HIP events on the solver stream around every launch of a second pass over the same workload, "
                            + ("pipelined as the timed region is (launches whose request was not pending are left out of the averages)"
                               if pipelined_pass else "in synchronous mode (only real work is launched)")
                            + "; raw brackets: the event packets add 1-3 us to the rocprofv3 kernel durations; profiles/README.md names "
                              "the CSV of this build"}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                key = f"n{n}_p{world}" + ("_sym2" if sym2 else "_sym" if sym_pass else "")
                rec = json.load(open(pmc)).get(key, {})
                roofline["traffic"] = rec.get("quad_eval_bytes_per_launch" if dom_is_eval else "h_pass_bytes_per_launch")
                roofline["update_pass"]["traffic"] = rec.get("h_pass_bytes_per_launch")
                roofline["quad_matvec"]["traffic"] = rec.get("quad_eval_bytes_per_launch")
                if roofline["traffic"] is not None:
                    roofline["traffic_source"] = (f"profiles/pmc_traffic.json[{key}]: HBM bytes per launch from separate rocprofv3 --pmc "
                                                  "FETCH_SIZE / WRITE_SIZE passes (gfx950 x2 correction on FETCH_SIZE); an offline figure, "
                                                  "not a counter read during this run; recorded on: " + str(rec.get("build", "an earlier build"))
                                                  + "; counter collection serialises the dispatches with a cache flush, so what an XCD's L2 keeps from "
                                                    "one launch to the next (round 6: zig-zag order, touch workgroups) does not show: an upper bound")
            except Exception:  # noqa: BLE001
                pass

    if rank == 0:
        mat_bytes = float(solver.stats()["matrix_bytes_per_pass"])  # per rank: the row shard, or the symmetric half (symmetric storage)
        symmetric = bool(solver.stats()["path"] & 2)
        second_gen = bool(solver.stats()["path"] & 16)
        tiles1 = bool(solver.stats()["path"] & 32)  # (round 5) a rank's share of H past the Infinity Cache: first-generation tile kernel behind a machine launch
        # what the default layout of this shape is, and whether the run fell back from it (nothing in the line said so)
        layout_fallback = None
        if not symmetric:
            layout_fallback = ("full rows (2x the bytes): the symmetric-storage layout needs whole 128-row blocks per rank and n >= 1024"
                               if args.tiling is None else "full rows: selected with --tiling")
        elif not second_gen:
            layout_fallback = ("first-generation tile kernels (separate sum / epilogue / control launches): the second-generation path needs "
                               "n = n_pad" if args.tiling is None else "first-generation tile kernels: selected with --tiling")
        b_iter = world * (h_bytes + obj_bytes) / steps  # counted: (passes + read-write passes) x bytes per pass + evaluations x bytes per pass
        out = {
            "metric": METRIC, "value": its, "unit": "iterations/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BFGS + MoreThuente::default (max_iter_line_search 20), n={n} convex quadratic "
                                   f"(random SPD Q, kappa=1e3, seed 0x5EED0001), f64, {world}xMI355X"
                                   + ((", H and Q row-sharded, " + ("host-staged" if host_exchange else "RCCL") + " exchanges (per iteration: 8 KB of scalars per "
                                       "evaluation, the partial n-vectors of the accepted point and of the update pass)") if world > 1 else "")
                                   + (", symmetric storage: only the symmetric half of H and Q is streamed" if symmetric else ""),
                       "matrix_layout": ("symmetric half of H and Q: 128 x 128 tiles above the diagonal + the upper triangles of the diagonal tiles"
                                         if (symmetric and world == 1) else
                                         "rows of H and Q sharded over the ranks; each rank streams the circulant half of its own 128-row "
                                         "blocks (every pair of blocks once across the ranks), per-rank partial n-vectors all-gathered and "
                                         "summed in rank order" if symmetric else "full row-major, row-sharded"),
                       "kernels": ("second generation: the state machine in every kernel's prologue; per iteration E evaluation launches + "
                                   + (("6 (partial sums, vectors, machine, update tiles [one workgroup per tile: the rank's share of H is past the Infinity Cache], "
                                       "partial sums, reduce), E scalar + 2 n-vector collectives") if (world > 1 and tiles1) else
                                      "5 (partial sums, vectors, update tiles, partial sums, reduce), E scalar + 2 n-vector collectives" if world > 1
                                      else "3 (vectors, update tiles, reduce)")) if second_gen else "first generation",
                       "layout_fallback": layout_fallback,
                       "n": n, "line_search": args.ls, "tol": 1e-10, "parallelism": f"row-shard x{world}",
                       "exchange": "none" if world == 1 else ("host-staged gloo, stream-ordered (rehearsal)" if host_exchange else "rccl all-gather"),
                       **({"rccl_error": rccl_error} if rccl_error else {})},
            "timing": {"regions": regions, "reported": "median region", "region_ms": [1e3 * t for t in region_s],
                       "each_region": f"reset to (x0, H = I), {warmup} untimed iterations, then {steps} timed iterations",
                       "qn_minimize_calls_in_region": acc["total_minimize_calls"],
                       "per_call_fixed_ms": per_call_fixed_ms,
                       "per_call_fixed_note": "intercept of the line through the median wall times of warm calls of 4 and of 24 iterations "
                                              "(entry, first launch from an idle queue, the last launch's report to the host, return); "
                                              "it is inside every timed region once per call"},
            "iteration_accounting": {"oracle_calls_reference_sequence": calls, "oracle_evaluations_distinct": evals,
                                     "evaluations_per_iteration": evals / steps, "oracle_calls_per_iteration": calls / steps,
                                     "h_passes": acc["total_h_passes"],
                                     "restarts_after_convergence": restarts,
                                     "algorithmic_bytes_per_iteration": b_iter,
                                     "full_matrix_bytes_per_iteration_survey_8d": 16.0 * n * n + 8.0 * n * n * (evals / steps),
                                     "whole_iteration_hbm_frac": b_iter * its / (world * HBM_PEAK_GBS * 1e9),
                                     "h_bytes_counted": h_bytes, "objective_bytes_counted": obj_bytes,
                                     "launches": acc["launches"], "host_syncs": acc["host_syncs"],
                                     "launches_per_iteration": acc["launches"] / steps,
                                     "collectives_of_n_vectors": acc["total_xchg_vector"], "collectives_of_scalars": acc["total_xchg_scalar"],
                                     "collectives_per_iteration": {"n_vectors": acc["total_xchg_vector"] / steps, "scalars_8KB": acc["total_xchg_scalar"] / steps},
                                     "note": "counters are the solver's cumulative totals differenced around the median timed region"},
            "roofline": roofline,
        }
        if world > 1:
            out["multi_gpu_readiness"] = collectives_share(xprobe, acc["total_xchg_scalar"] / steps, acc["total_xchg_vector"] / steps, 1e3 * elapsed / steps)
        if cpu is not None:
            out["cpu_baseline"] = cpu.get("config2", cpu) if "error" not in cpu else cpu
        if want_extra:
            # BASELINE.json configs[3] and configs[4], driver-visible: timed after the headline region, each with
            # its own roofline and CPU leg.  The headline `value` / `config` stay configs[1].
            del solver, obj
            extra = {}
            for name, leg in (("config4_newton_n8192", lambda: extra_config4(qn, ctx)), ("config5_dfp_logsumexp_n16384", lambda: extra_config5(qn, ctx)),
                              ("bounded_bfgsb_morethuenteb_n4096", lambda: extra_bounded(qn, ctx))):
                try:
                    extra[name] = leg()
                except Exception as e:  # noqa: BLE001 -- an extra leg never loses the bench line
                    extra[name] = {"error": repr(e)}
            if cpu is not None and "error" not in cpu:
                for name, key in (("config4_newton_n8192", "config4"), ("config5_dfp_logsumexp_n16384", "config5")):
                    if isinstance(extra.get(name), dict):
                        extra[name]["cpu_baseline"] = cpu.get(key)
            out["extra_configs"] = extra
        if world > 1 and not args.no_scaling_ref:
            # strong-scaling denominator: the SAME workload unsharded on rank 0's GPU alone (N = 1 of the default run
            # is configs[1], a different problem size, so value(N)/value(1) across default runs is not an efficiency)
            try:
                ctx1 = qn.Context(device=local_rank % max(torch.cuda.device_count(), 1))
                obj1 = qn.Quadratic.synthetic(n, SEED, diag, b, ctx=ctx1)
                s1 = qn.BFGS(1e-10, x0, ctx=ctx1)
                if symmetric and not second_gen:
                    s1.set_option("second_generation", 0)  # the same kernels as the N-rank run: first-generation symmetric-storage tiles
                elif not symmetric:
                    s1.set_option("symmetric_storage", 0)  # the same kernels as the N-rank run: fused row kernels on the full matrices
                # (second generation: the single-GPU default path IS the N-rank run's kernels, unsharded)
                ref_steps = min(steps, 50)
                run_iterations(qn, s1, ls, obj1, x0, min(warmup, 5))
                ctx1.synchronize()
                t1 = time.perf_counter()
                run_iterations(qn, s1, ls, obj1, x0, ref_steps)
                ctx1.synchronize()
                dt1 = time.perf_counter() - t1
                out["strong_scaling_ref"] = {"n_gpus": 1, "value": ref_steps / dt1, "unit": "iterations/s", "steps": ref_steps,
                                             "note": f"same n={n} workload, same storage layout and the same kernels ("
                                                     + ("symmetric half, second-generation kernels" if second_gen else
                                                        "symmetric half, first-generation tile kernels" if symmetric else "full row-major matrices")
                                                     + "), unsharded, on rank 0's GPU after the timed region"}
                # for the record: one GPU on its default path (symmetric storage, second-generation kernels: 5 launches per iteration)
                if second_gen and symmetric:
                    out["strong_scaling_ref"]["single_gpu_default_path_value"] = out["strong_scaling_ref"]["value"]  # (the same run)
                else:
                    s2 = qn.BFGS(1e-10, x0, ctx=ctx1)
                    run_iterations(qn, s2, ls, obj1, x0, min(warmup, 5))
                    ctx1.synchronize()
                    t2 = time.perf_counter()
                    run_iterations(qn, s2, ls, obj1, x0, ref_steps)
                    ctx1.synchronize()
                    out["strong_scaling_ref"]["single_gpu_default_path_value"] = ref_steps / (time.perf_counter() - t2)
                    del s2
                del s1, obj1, ctx1
            except Exception as e:  # noqa: BLE001 -- the reference leg must never lose the bench line
                out["strong_scaling_ref"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
