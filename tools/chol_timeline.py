import csv,glob,sys
f=glob.glob("gpurun_out/chol_prof/**/p_kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
k=[i for i,r in enumerate(rows) if "chol_diag" in r["Kernel_Name"]]
start=k[-128]
t0=int(rows[start]["Start_Timestamp"])
blk=[(int(rows[k[-128+j]]["Start_Timestamp"])-t0)/1e3 for j in range(0,128,4)]
print("block periods:"," ".join("%.0f"%(blk[i+1]-blk[i]) for i in range(len(blk)-1)), " total to last block start %.0f"%blk[-1])
d={}
for r in rows[start:]:
    nm=r["Kernel_Name"].split("(")[0][:20]+" q"+r.get("Queue_Id","?")
    d.setdefault(nm,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for nm,v in d.items(): print("  %-28s n=%4d avg %7.1f total %8.1f us"%(nm,len(v),sum(v)/len(v),sum(v)))
end=max(int(r["End_Timestamp"]) for r in rows[start:])
print("factorisation + solves span: %.0f us"%((end-t0)/1e3))
