#!/bin/bash
# HBM traffic per launch of the benchmark's tile kernels from PMC counters: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
# they do not fit one pass on gfx950), --kernel-trace only, the program directly behind `--`.  usage: bash tools/pmc_traffic.sh <tag>
tag=${1:-pmc}
out=gpurun_out/$tag
repo="${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel 2>/dev/null || pwd)}"
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$repo"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-profile-pass > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-profile-pass > /dev/null 2> $out/write.err
python3 - "$out" <<'PY'
import csv, statistics, json, sys, glob
out = sys.argv[1]
def med(path, counter, key):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if key in r["Kernel_Name"] and r["Counter_Name"] == counter]
    v = [x for x in v if x > 0.5 * max(v)] if v else v   # active launches only (a predicated-off launch moves nothing)
    return (statistics.median(v), len(v)) if v else (0.0, 0)
f = glob.glob(out + "/fetch/**/f_counter_collection.csv", recursive=True)[0]
w = glob.glob(out + "/write/**/w_counter_collection.csv", recursive=True)[0]
res = {}
for key, name in (("s2_hpass_kernel", "h_pass"), ("s2_eval", "quad_eval")):  # ("s2_eval": s2_eval_kernel and, at n = 4096 since round 6, s2_evalr_kernel)
    fk, nf = med(f, "FETCH_SIZE", key); wk, nw = med(w, "WRITE_SIZE", key)
    res[name + "_FETCH_SIZE_KB"] = fk; res[name + "_WRITE_SIZE_KB"] = wk
    res[name + "_bytes_per_launch"] = 2 * fk * 1024 + wk * 1024
    res[name + "_launches"] = [nf, nw]
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/pmc_traffic_sym2.json", "w"), indent=1)
PY
