#!/bin/bash
# experiments on one rank of the sharded partition: bash tools/shard_exp.sh <tag> "<ENV=... ENV=...>" [P n iters rank]
set -o pipefail
tag=$1; envs=$2; P=${3:-8}; n=${4:-32768}; iters=${5:-10}; r=${6:-0}
out=gpurun_out/$tag; mkdir -p $out
repo="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
rec=/tmp/exp_$tag.npz
env $envs timeout -k 10 600 python3 tools/solo_rank.py record $P $n $iters $rec > $out/record.json 2> $out/record.err || { echo "record failed"; tail -5 $out/record.err; exit 1; }
export $envs
rm -rf $out/prof; mkdir -p $out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 tools/solo_rank.py replay $P $n $iters $rec $r > $out/replay.json 2> $out/err.txt || { echo "replay failed"; tail -5 $out/err.txt; exit 1; }
python3 - "$out/prof" "$tag" "$envs" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/p_kernel_trace.csv", recursive=True)
d = {}
for r in csv.DictReader(open(f[0])):
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d.setdefault(nm, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for nm, v in d.items():
    if not nm.startswith("s2"): continue
    real = [x for x in v if x > 0.5 * max(v)]
    out.append("%s %.1f (n=%d)" % (nm.replace("_kernel", "")[:24], sum(real) / len(real), len(real)))
print(sys.argv[2], "[", sys.argv[3], "]", " | ".join(sorted(out)))
PY
