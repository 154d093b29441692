#!/bin/bash
# One GPU-box call: config 3's partition (P ranks, n) recorded with all ranks as threads, then ONE rank replayed alone under
# rocprofv3 (tools/solo_rank.py) -- the per-kernel durations of what a GPU of a P-GPU node executes.
# usage: bash tools/shard_solo.sh <tag> [P] [n] [iters] [ranks...]       e.g.  bash tools/shard_solo.sh r04_a 8 32768 10 0 7
set -o pipefail
tag=${1:-solo}; P=${2:-8}; n=${3:-32768}; iters=${4:-10}; shift 4
ranks=${@:-0}
out=gpurun_out/$tag
repo="${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel 2>/dev/null || pwd)}"
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$repo"
for gen in second first; do
  rec=/tmp/solo_${gen}.npz
  timeout -k 10 600 python3 tools/solo_rank.py record $P $n $iters $rec $([ $gen = first ] && echo first) > $out/record_$gen.json 2> $out/record_$gen.err || { echo "record $gen failed"; tail -5 $out/record_$gen.err; exit 1; }
  cat $out/record_$gen.json
  for r in $ranks; do
    d=$out/${gen}_rank$r
    rm -rf $d; mkdir -p $d
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 tools/solo_rank.py replay $P $n $iters $rec $r $([ $gen = first ] && echo first) > $d/replay.json 2> $d/err.txt || { echo "replay $gen rank $r failed"; tail -5 $d/err.txt; exit 1; }
    cat $d/replay.json
    python3 - "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/p_kernel_stats.csv", recursive=True)
for r in sorted(csv.DictReader(open(f[0])), key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("  %-70s calls %5s  avg %9.2f us  total %8.3f ms  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
  done
done
