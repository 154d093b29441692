# Diagnostic: the update-tile kernel has two modes between PROCESSES (26 or 30.5 us in bench.py's event brackets for a whole run, about
# one process in five in the slow one): bench.py N times, one line each.  Round 3 findings, same box: it does not follow H's virtual
# address or a pooled allocation; with non-temporal loads of Q in the evaluation kernel the slow mode did not appear in 20 runs but
# the evaluation lost 1.5 us; with non-temporal accesses to H the update kernel takes 34.6 us always -- at n = 4096 both half
# matrices live in the 256 MB Infinity Cache, and how evenly their physical pages spread over its slices differs from process to process.
# usage: bash tools/modes_ab.sh [runs]
for i in $(seq 1 ${1:-10}); do python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('it/s %.0f  update %.2f us  eval %.2f us' % (d['value'], 1e3 * r['update_pass']['avg_launch_ms'], 1e3 * r['quad_matvec']['avg_launch_ms']))"; done
