"""CPU tests of the measurement harness: the `cpu_baseline` legs of bench.py run in a child process without torch or a GPU (they import
numpy and the oracle only), return the keys the bench line promises, and the experiments kept as patches still apply to the tree."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_child_runs_without_torch_and_reports_sub_samples():
    """`bench.py --cpu-baseline-child n iters threads 0`: the config-2 leg alone at a small size -- three sub-samples, their median as the
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
    assert len(leg["sub_samples"]) == 3 and sorted(leg["sub_samples"])[1] == leg["value"] and leg["value"] > 0
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
