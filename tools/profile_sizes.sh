#!/bin/bash
# final-tree kernel evidence at the HBM-regime sizes: rocprofv3 --kernel-trace --stats + one PMC pass each of
# bench.py --dim N.   usage: bash tools/profile_sizes.sh <tag> [sizes...]
tag=${1:-sizes}; shift
sizes=${@:-"8192 32768"}
out=gpurun_out/$tag; mkdir -p $out
repo="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
for n in $sizes; do
  steps=$([ $n -ge 16384 ] && echo 30 || echo 100)
  python3 bench.py --dim $n --steps $steps --no-cpu-baseline > $out/bench_n$n.json 2> $out/bench_n$n.err
  rm -rf $out/prof_n$n; mkdir -p $out/prof_n$n
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_n$n -o p -- python3 bench.py --dim $n --steps $steps --no-cpu-baseline --no-profile-pass > $out/bench_under_rocprof_n$n.json 2> $out/rocprof_n$n.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pmc_${c}_n$n; mkdir -p $out/pmc_${c}_n$n
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${c}_n$n -o c -- python3 bench.py --dim $n --steps 10 --warmup 2 --no-cpu-baseline --no-profile-pass > /dev/null 2> $out/pmc_${c}_n$n.err
  done
  python3 - "$out" "$n" <<'PY'
import csv, glob, json, statistics, sys
out, n = sys.argv[1], sys.argv[2]
st = glob.glob(f"{out}/prof_n{n}/**/p_kernel_stats.csv", recursive=True)
for r in sorted(csv.DictReader(open(st[0])), key=lambda r: -float(r["TotalDurationNs"]))[:6]:
    print("  n=%s %-60s calls %5s avg %9.2f us  %5s %%" % (n, r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{c}_n{n}/**/c_counter_collection.csv", recursive=True)
    if not f: continue
    for key in ("s2_hpass_kernel", "s2_eval_kernel"):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if key in r["Kernel_Name"] and r["Counter_Name"] == c]
        v = [x for x in v if x > 0.5 * max(v)] if v else v
        res[f"{key}_{c}_KB"] = statistics.median(v) if v else None
for key, name in (("s2_hpass_kernel", "h_pass"), ("s2_eval_kernel", "quad_eval")):
    f_, w_ = res.get(f"{key}_FETCH_SIZE_KB"), res.get(f"{key}_WRITE_SIZE_KB")
    if f_ is not None and w_ is not None: res[name + "_bytes_per_launch"] = 2 * f_ * 1024 + w_ * 1024  # (gfx950: FETCH_SIZE counts 64-byte units as 32)
print("  n=%s pmc %s" % (n, json.dumps(res)))
json.dump(res, open(f"{out}/pmc_traffic_n{n}.json", "w"), indent=1)
PY
done
