#!/bin/bash
# A/B on one box: the evaluation tiles at n = 4096 as round 5's two-items-and-a-sliver instance (QN_S2_RING=0) against the mover / multiplier
# kernel of qn_sym2r.hip.h (default), per-kernel averages of rocprofv3 --kernel-trace --stats, alternating repetitions.
#   usage (through gpurun): bash tools/ring_ab.sh [reps] [bench args]
reps=${1:-3}; shift
repo="${GRAFT_REPO_ROOT:-$PWD}"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
for rep in $(seq 1 $reps); do
  for ring in 0 1; do
    out=gpurun_out/ring_ab/r${ring}_$rep
    rm -rf $out; mkdir -p $out
    QN_S2_RING=$ring rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --steps 200 --no-cpu-baseline --no-profile-pass --no-extra-configs "$@" > $out/bench.json 2> $out/err.txt
    python3 - "$ring" "$out" <<'PY'
import sys, csv, glob, json
ring, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
d = {}
for r in rows:
    n = r["Name"]
    for key in ("s2_eval", "s2_hpass", "s2_vec", "s2_hreduce"):
        if key in n and key not in d: d[key] = float(r["TotalDurationNs"]) / max(1, int(r["Calls"]))
try:
    b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1]); v = "%.0f it/s" % b["value"]
except Exception as e:
    v = "bench failed"
tot = 2 * d.get("s2_eval", 0) + d.get("s2_hpass", 0) + d.get("s2_vec", 0) + d.get("s2_hreduce", 0)
print("ring=%s %s | eval %.2f  hpass %.2f  vec %.2f  hreduce %.2f us | 2E+H+V+R = %.2f us" % (ring, v, d.get("s2_eval", 0) / 1e3, d.get("s2_hpass", 0) / 1e3, d.get("s2_vec", 0) / 1e3, d.get("s2_hreduce", 0) / 1e3, tot / 1e3))
PY
  done
done
