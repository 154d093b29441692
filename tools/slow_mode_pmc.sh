#!/bin/bash
# Root cause of the update kernel's slow mode (n = 4096): tools/slow_mode_probe.py (K solvers in one process, each with its own H)
# under rocprofv3 --pmc, one counter set per pass (--kernel-trace only, program directly behind `--`); per solver: the update
# kernel's median duration and median counter values.  usage: bash tools/slow_mode_pmc.sh <tag> [K]
tag=${1:-slowpmc}; K=${2:-10}
out=gpurun_out/$tag; mkdir -p $out
repo="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_READ_REQ_LATENCY_sum TCC_READ_REQ_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum" "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum"; do
  i=$((i+1)); d=$out/pass$i; rm -rf $d; mkdir -p $d
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o p -- python3 tools/slow_mode_probe.py $K > $d/probe.txt 2> $d/err.txt || { echo "pass $i failed"; tail -3 $d/err.txt; continue; }
  cat $d/probe.txt
  python3 - "$d" "$K" <<'PY'
import csv, glob, sys, statistics
d, K = sys.argv[1], int(sys.argv[2])
cc = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(cc[0])) if "s2_hpass" in r["Kernel_Name"]]
by = {}
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
kt = glob.glob(d + "/**/p_kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        if "s2_hpass" in r["Kernel_Name"]: dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ids = sorted(by)
per = len(ids) // K
names = sorted({k for v in by.values() for k in v})
print("   solver  " + "  ".join("%28s" % n[:28] for n in names) + "   kernel us (median of the launches that worked)")
for s in range(K):
    chunk = ids[s * per:(s + 1) * per]
    real = [i for i in chunk if dur.get(i, 0) > 0.5 * max(dur.get(j, 0) for j in chunk)] if dur else chunk
    line = "   %5d   " % s + "  ".join("%28.0f" % statistics.median(by[i].get(n, 0.0) for i in real) for n in names)
    print(line + ("   %.1f" % statistics.median(dur[i] for i in real) if dur else ""))
PY
done
