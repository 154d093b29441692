"""CPU tests of the measurement harness: the `cpu_baseline` legs of bench.py run in a child process without torch or a GPU (they import
numpy and the oracle only), return the keys the bench line promises, and the experiments kept as patches still apply to the tree."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_child_runs_without_torch_and_reports_sub_samples():
    """`bench.py --cpu-baseline-child n iters threads 0`: the config-2 leg alone at a small size -- five sub-samples, their median as the
    value, the spread, the bytes accounting; and the child must not have imported torch (its OpenMP runtime is what round 3 tripped over)."""
    code = ("import sys, json; sys.argv = ['bench.py']; sys.path.insert(0, %r); import bench; "
            "out = bench.cpu_baseline_child(256, 30, 2, extra=False); out['torch_loaded'] = 'torch' in sys.modules; print(json.dumps(out))" % ROOT)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cp = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert cp.returncode == 0, cp.stderr[-2000:]
    out = json.loads(cp.stdout.strip().splitlines()[-1])
    assert out["torch_loaded"] is False
    leg = out["config2"]
    assert leg["kind"] == "port" and leg["cores"] == 2 and leg["unit"] == "iterations/s"
    assert len(leg["sub_samples"]) == 5 and sorted(leg["sub_samples"])[2] == leg["value"] and leg["value"] > 0
    assert leg["sub_sample_min"] <= leg["sub_sample_median"] == leg["value"] <= leg["sub_sample_max"]
    assert "as_written_1thread" in leg and leg["bytes_per_iteration"] > 0


def test_newton_cpu_leg_extrapolates_from_a_bounded_sample():
    sys.path.insert(0, ROOT)
    import bench
    from oracle import qn_oracle as qo
    leg = bench.cpu_leg_config4(qo, 8192)
    assert leg["cores"] == 1 and leg["kind"] == "port" and len(leg["sub_samples_s_per_iteration_at_n768"]) == 3
    assert abs(leg["s_per_iteration_extrapolated"] - sorted(leg["sub_samples_s_per_iteration_at_n768"])[1] * (8192 / 768) ** 3) < 1e-9 * leg["s_per_iteration_extrapolated"]
    assert "n=768" in leg["sample"] and "n^3" in leg["sample"]


def test_experiments_kept_as_patches_still_apply():
    """tools/experiments/*.patch: variants that were built, measured and dropped live as patches (code that is
    not used does not stay beside the benchmark's kernels); a patch that no longer applies is a claim that no longer holds."""
    pdir = os.path.join(ROOT, "tools", "experiments")
    patches = sorted(f for f in os.listdir(pdir) if f.endswith(".patch"))
    assert patches
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        import pytest
        pytest.skip("not a git checkout (the snapshot on the GPU box): git apply --check needs the index")
    for f in patches:
        cp = subprocess.run(["git", "apply", "--check", os.path.join(pdir, f)], cwd=ROOT, capture_output=True, text=True)
        assert cp.returncode == 0, (f, cp.stderr[-1000:])


def test_timed_regions_difference_the_counters_and_refuse_a_short_region():
    """bench.timed_regions (the config-5 leg's accounting): what a region did comes from the solver's cumulative counters;
    a region that ran fewer iterations than it is quoted for -- round 5: a converged run swallowed as Ok -- raises instead of printing."""
    import pytest
    sys.path.insert(0, ROOT)
    import bench

    class Fake:
        def __init__(self, converge_after=None):
            self.tot = {"total_iterations": 0, "total_oracle_evals": 0}
            self.converge_after = converge_after
            self.restarts = 0

        def stats(self):
            return dict(self.tot)

        def run_exact(self, k):  # the shape of bench.run_iterations: exactly k iterations, restarting a run that converges
            done, restarts = 0, 0
            while done < k:
                step = k - done
                if self.converge_after is not None and self.tot["total_iterations"] % self.converge_after + step > self.converge_after:
                    step = self.converge_after - self.tot["total_iterations"] % self.converge_after
                    restarts += 1
                self.tot["total_iterations"] += step
                self.tot["total_oracle_evals"] += 2 * step
                done += step
            return restarts

        def run_swallowing_convergence(self, k):  # round 5's run(): stops at convergence and says nothing
            step = min(k, 7)
            self.tot["total_iterations"] += step
            self.tot["total_oracle_evals"] += step
            return 0

    f = Fake(converge_after=33)
    regs = bench.timed_regions(f.run_exact, f.stats, lambda: None, 20, 3)
    assert [r["iterations"] for r in regs] == [20, 20, 20] and [r["evaluations"] for r in regs] == [40, 40, 40]
    assert sum(r["restarts"] for r in regs) == 1 and all(r["s_per_iteration"] == r["s"] / 20 for r in regs)
    g = Fake()
    with pytest.raises(RuntimeError, match="ran 7 iterations"):
        bench.timed_regions(g.run_swallowing_convergence, g.stats, lambda: None, 20, 3)


def test_collectives_share_estimates_from_probe_and_counts():
    """bench.collectives_share (N > 1): probed latencies x counted collectives, slowest rank; a failed probe is reported, not raised."""
    sys.path.insert(0, ROOT)
    import bench
    pr = lambda a, b, c: {"scalars_8KB": {"median_us": a}, "n_vector": {"median_us": b}, "two_n_vectors": {"median_us": c}}  # noqa: E731
    out = bench.collectives_share({"ranks": 2, "exchange": "rccl", "per_rank": [pr(10.0, 30.0, 50.0), pr(12.0, 28.0, 60.0)]}, 2.0, 2.0, 0.5)
    assert out["slowest_rank_median_us"] == {"scalars_8KB": 12.0, "n_vector": 30.0, "two_n_vectors": 60.0}
    assert abs(out["estimated_collective_us_per_iteration"] - (2 * 12.0 + 30.0 + 60.0)) < 1e-12
    assert abs(out["estimated_fraction_of_ms_per_step"] - 114.0 / 500.0) < 1e-12
    bad = bench.collectives_share({"ranks": 2, "exchange": "rccl", "per_rank": [{"error": "x"}, {"error": "y"}]}, 2.0, 2.0, 0.5)
    assert "error" in bad and "estimated_fraction_of_ms_per_step" not in bad
