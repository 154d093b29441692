"""print the interesting fields of a bench.py JSON line (diagnostic helper)"""
import json, sys
for path in sys.argv[1:]:
  d = json.loads(open(path).read().strip().splitlines()[-1])
  print("==", path)
  print("value", d["value"], "ms_per_step", d["ms_per_step"])
  print("timing", d.get("timing"))
  print("accounting", d.get("iteration_accounting"))
  r = d.get("roofline") or {}
  print("roofline", {k: r.get(k) for k in ("kernel", "time_share", "achieved", "frac", "avg_launch_ms", "launches_timed", "traffic", "all_kernels_time_weighted")})
  print("update_pass", r.get("update_pass")); print("eval", r.get("quad_matvec")); print("ctl", r.get("ctl_step"))
  if "cpu_baseline" in d: print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["sample"])
