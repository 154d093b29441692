#!/bin/bash
# The driver-style headline line (fresh process each time) repeated N times on one box: the spread between processes (H's placement, the
# box's clocks) that a single BENCH line cannot show.  usage (through gpurun): bash tools/bench_repeat.sh <tag> [N]
tag=${1:-rep}; n=${2:-24}
out=gpurun_out/$tag; mkdir -p $out
: > $out/values.txt
for i in $(seq 1 $n); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-profile-pass > $out/line.json 2> $out/line.err || { echo "run $i failed"; tail -3 $out/line.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$out/line.json').read().strip().splitlines()[-1]); print('%d %.1f %.3f' % ($i, d['value'], d['ms_per_step']*1e3))" | tee -a $out/values.txt
done
python3 - $out/values.txt <<'PY'
import sys
v = sorted(float(l.split()[1]) for l in open(sys.argv[1]))
print("runs %d  min %.0f  p25 %.0f  median %.0f  p75 %.0f  max %.0f it/s  (spread %.1f %% of the median)" % (len(v), v[0], v[len(v)//4], v[len(v)//2], v[3*len(v)//4], v[-1], 100*(v[-1]-v[0])/v[len(v)//2]))
PY
