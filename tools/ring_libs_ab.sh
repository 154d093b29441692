#!/bin/bash
# per-kernel averages for several builds of the library with the ring kernel on: bash tools/ring_libs_ab.sh "<libA> <libB> ..." [reps]
libs=$1; reps=${2:-2}
repo="${GRAFT_REPO_ROOT:-$PWD}"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
for rep in $(seq 1 $reps); do for lib in $libs; do
  out=gpurun_out/ring_libs/${lib}_$rep; rm -rf $out; mkdir -p $out
  QN_S2_RING=${RING:-1} QN_HIP_LIB=$PWD/optimization-solvers_amd/lib/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --steps 200 --no-cpu-baseline --no-profile-pass --no-extra-configs > $out/bench.json 2> $out/err.txt
  python3 - "$lib" "$out" <<'PY'
import sys, csv, glob, json
lib, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
d = {}
for r in rows:
    for key in ("s2_eval", "s2_hpass", "s2_vec", "s2_hreduce"):
        if key in r["Name"] and key not in d: d[key] = float(r["TotalDurationNs"]) / max(1, int(r["Calls"]))
try: v = "%.0f it/s" % json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])["value"]
except Exception: v = "bench failed"
print("%-20s %s | eval %.2f  hpass %.2f  vec %.2f  hreduce %.2f us | 2E+H+V+R = %.2f us" % (lib, v, d.get("s2_eval", 0) / 1e3, d.get("s2_hpass", 0) / 1e3, d.get("s2_vec", 0) / 1e3, d.get("s2_hreduce", 0) / 1e3, (2 * d.get("s2_eval", 0) + d.get("s2_hpass", 0) + d.get("s2_vec", 0) + d.get("s2_hreduce", 0)) / 1e3))
PY
done; done
