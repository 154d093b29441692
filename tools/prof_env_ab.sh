#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of bench.py for several ENVIRONMENT settings of one build, same box,
# alternating: bash tools/prof_env_ab.sh "<tag>:<VAR=VAL,VAR=VAL>" ... [-- bench args]
# e.g.  bash tools/prof_env_ab.sh "launch:QN_S2_TRED=0" "tail:QN_S2_TRED=1"
# (the it/s line moves +-3 % between boxes and +-1 % between runs; the kernel averages over ~1000 launches are good to ~0.1 us)
specs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do specs+=("$1"); shift; done
[ "$1" == "--" ] && shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
reps=${REPS:-2}
for rep in $(seq 1 $reps); do
for spec in "${specs[@]}"; do
  tag=${spec%%:*}; envs=${spec#*:}
  out=gpurun_out/prof_env_ab/$tag.$rep
  rm -rf $out; mkdir -p $out
  (
    IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
    rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --steps 200 --no-cpu-baseline --no-profile-pass --no-extra-configs "$@" > $out/bench.json 2> $out/err.txt
  )
  python3 - "$tag.$rep" "$out" <<'PY'
import sys, csv, glob, json
tag, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
d = {}
for r in rows:
    n = r["Name"]
    # (the extra legs are off -- their launches of the same kernels were folded into these averages -- and of several instantiations of a
    # kernel the one with the most calls is the headline run's)
    for key in ("s2_eval", "s2_hpass", "s2_vec", "s2_hreduce"):
        if key in n and int(r["Calls"]) > d.get(key + "#calls", 0):
            d[key] = float(r["TotalDurationNs"]) / max(1, int(r["Calls"])); d[key + "#calls"] = int(r["Calls"])
try:
    b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1]); v = "%.0f it/s" % b["value"]
except Exception as e:
    v = "bench failed"
tot = 2 * d.get("s2_eval", 0) + d.get("s2_hpass", 0) + d.get("s2_vec", 0) + d.get("s2_hreduce", 0)
print("%-14s %s | eval %.2f  hpass %.2f  vec %.2f  hreduce %.2f us | 2E+H+V+R = %.2f us" % (tag, v, d.get("s2_eval", 0) / 1e3, d.get("s2_hpass", 0) / 1e3, d.get("s2_vec", 0) / 1e3, d.get("s2_hreduce", 0) / 1e3, tot / 1e3))
PY
done
done
