"""Timeline of the last Cholesky factorisation in a rocprofv3 kernel trace (gpurun_out/chol_prof): per-kernel totals by queue, the
period of every outer block (4 panels), and for the outer blocks given on the command line every launch relative to the block's first
panel kernel -- where the chain waits, where the bulk (second stream) runs."""
import csv, glob, sys
f = glob.glob("gpurun_out/chol_prof/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = [i for i, r in enumerate(rows) if "chol_panel" in r["Kernel_Name"]]
start = k[-127]
while start > 0 and "chol_" in rows[start - 1]["Kernel_Name"]: start -= 1
t0 = int(rows[start]["Start_Timestamp"])
d = {}
for r in rows[start:]:
    nm = r["Kernel_Name"].split("(")[0][:24] + " q" + r.get("Queue_Id", "?")
    d.setdefault(nm, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for nm, v in d.items():
    print("  %-32s n=%4d avg %7.1f total %8.1f us" % (nm, len(v), sum(v) / len(v), sum(v)))
end = max(int(r["End_Timestamp"]) for r in rows[start:])
print("factorisation + solves span: %.0f us" % ((end - t0) / 1e3))
blk = [(int(rows[k[-127 + j]]["Start_Timestamp"]) - t0) / 1e3 for j in range(0, 127, 4)]
print("outer block periods (us):", " ".join("%.0f" % (blk[i + 1] - blk[i]) for i in range(len(blk) - 1)))
for b in (int(a) for a in sys.argv[1:]):
    a0 = k[-127 + 4 * b]; a1 = k[-127 + 4 * b + 4] if 4 * b + 4 < 127 else len(rows) - 1
    ta, tb = int(rows[a0]["Start_Timestamp"]), int(rows[a1]["Start_Timestamp"])
    print("outer block %d:" % b)
    for r in rows[max(start, a0 - 3):]:
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s_ > tb: break
        if e_ < ta - 40000: continue
        print("   q%s %-26s %8.1f .. %8.1f  (%6.1f)" % (r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:26], (s_ - ta) / 1e3, (e_ - ta) / 1e3, (e_ - s_) / 1e3))
