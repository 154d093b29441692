#!/bin/bash
# TLB pressure of the update-pass tile kernel in two shapes (rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum, its own run):
# (a) config 5 on one GPU (n = 16384: H is 2 GB), (b) one rank of the P = 8, n = 32768 partition replayed alone (its slab of H is 1 GB).
# usage (through gpurun): bash tools/tlb_pmc.sh <tag>
tag=${1:-tlb}
out=gpurun_out/$tag
repo="${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel 2>/dev/null || pwd)}"
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$repo"
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d $out/c5 -o p -- python3 tools/bench_config5.py 16384 5 > $out/c5.json 2> $out/c5.err || echo "c5 failed"
timeout -k 10 600 python3 tools/solo_rank.py record 8 32768 4 /tmp/solo_tlb.npz > $out/record.json 2> $out/record.err || echo "record failed"
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d $out/solo -o p -- python3 tools/solo_rank.py replay 8 32768 4 /tmp/solo_tlb.npz 0 > $out/solo.json 2> $out/solo.err || echo "solo failed"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("c5", "solo"):
    f = glob.glob(f"{out}/{tag}/**/p_counter_collection.csv", recursive=True)
    if not f:
        print(tag, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][:44]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "TCP_UTCL1_REQUEST_sum": cnt[k] += 1
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("TCP_UTCL1_REQUEST_sum", 0))[:6]:
        rq, ms = v.get("TCP_UTCL1_REQUEST_sum", 0), v.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0)
        print("%-5s %-46s launches %4d  requests/launch %12.0f  misses/launch %10.0f  miss ratio %.5f" % (tag, k, cnt[k], rq / max(cnt[k], 1), ms / max(cnt[k], 1), ms / max(rq, 1)))
PY
