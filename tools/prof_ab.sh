#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of bench.py for several builds of the library, same box:
# bash tools/prof_ab.sh "<libA> <libB> ..." [bench args].  Run-to-run noise of the it/s figure is +-3 %; the kernel averages over
# ~1000 launches each are good to ~0.1 us.
libs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
for lib in $libs; do
  out=gpurun_out/prof_ab/$lib
  rm -rf $out; mkdir -p $out
  QN_HIP_LIB=$PWD/optimization-solvers_amd/lib/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --steps 200 --no-cpu-baseline --no-profile-pass "$@" > $out/bench.json 2> $out/err.txt
  python3 - "$lib" "$out" <<'PY'
import sys, csv, glob, json
lib, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
d = {}
for r in rows:
    n = r["Name"]
    for key in ("s2_eval", "s2_hpass", "s2_vec", "s2_hreduce"):
        if key in n: d[key] = d.get(key, 0.0) + float(r["TotalDurationNs"]) / max(1, int(r["Calls"])) if key not in d else d[key]
try:
    b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1]); v = "%.0f it/s" % b["value"]
except Exception as e:
    v = "bench failed"
tot = 2 * d.get("s2_eval", 0) + d.get("s2_hpass", 0) + d.get("s2_vec", 0) + d.get("s2_hreduce", 0)
print("%-22s %s | eval %.2f  hpass %.2f  vec %.2f  hreduce %.2f us | 2E+H+V+R = %.2f us" % (lib, v, d.get("s2_eval", 0) / 1e3, d.get("s2_hpass", 0) / 1e3, d.get("s2_vec", 0) / 1e3, d.get("s2_hreduce", 0) / 1e3, tot / 1e3))
PY
done
