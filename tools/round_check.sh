#!/bin/bash
# One GPU-box call: the whole GPU test suite, the driver's bench line, the default bench line, and a rocprofv3 kernel summary.
# usage (from the repo root, through gpurun): bash tools/round_check.sh <tag>
set -o pipefail
tag=${1:-check}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest.log
tail -3 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.log; tail -1 $out/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench_steps20.err || echo "bench20 failed"
python bench.py > $out/bench_default.json 2> $out/bench_default.err || echo "bench default failed"
repo="${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel 2>/dev/null || pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o prof -- python3 bench.py --steps 200 --no-cpu-baseline --no-profile-pass > $out/bench_under_rocprof.json 2> $out/rocprof.err || echo "rocprof failed"
python tools/show_bench.py $out/bench_steps20.json $out/bench_default.json < /dev/null | cut -c1-1200
ls $out/prof/* < /dev/null | head
