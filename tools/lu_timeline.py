"""Timeline of the last LU factorisation in a rocprofv3 kernel trace (gpurun_out/lu_prof): per-kernel totals by queue, and for a few
panels the start / end of every launch relative to the panel's first kernel -- does the bulk (second stream) run beside the chain?"""
import csv, glob, sys
f = glob.glob("gpurun_out/lu_prof/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
loads = [i for i, r in enumerate(rows) if "lu_panel_persist" in r["Kernel_Name"] or "lu_panel_split" in r["Kernel_Name"]]  # one per panel (the one-launch panel)
if len(loads) < 128:
    loads = [i for i, r in enumerate(rows) if "lu_panel_load" in r["Kernel_Name"]]
start = loads[-128]
t0 = int(rows[start]["Start_Timestamp"])
d = {}
for r in rows[start:]:
    nm = r["Kernel_Name"].split("(")[0][:28] + " q" + r.get("Queue_Id", "?")
    d.setdefault(nm, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for nm, v in d.items():
    print("  %-36s n=%5d avg %8.1f total %9.1f us" % (nm, len(v), sum(v) / len(v), sum(v)))
end = max(int(r["End_Timestamp"]) for r in rows[start:])
print("factorisation + solves span: %.0f us" % ((end - t0) / 1e3))
per = [(int(rows[loads[-128 + j]]["Start_Timestamp"]) - t0) / 1e3 for j in range(128)]
print("panel periods (us):", " ".join("%.0f" % (per[i + 1] - per[i]) for i in range(0, 127, 8)))
for pi in (int(a) for a in sys.argv[1:]):
    a = loads[-128 + pi]; b = loads[-128 + pi + 1]
    ta = int(rows[a]["Start_Timestamp"])
    tb = int(rows[b]["Start_Timestamp"])
    print("panel %d:" % pi)
    for r in rows[max(start, a - 4):]:
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s_ > tb + 1000: break
        if e_ < ta: continue
        print("   q%s %-30s %9.1f .. %9.1f  (%7.1f)" % (r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:30], (s_ - ta) / 1e3, (e_ - ta) / 1e3, (e_ - s_) / 1e3))
