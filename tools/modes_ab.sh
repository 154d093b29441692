# Diagnostic: the update-tile kernel has two modes between PROCESSES (26 or 29-30.5 us in bench.py's event brackets for a whole run):
# bench.py N times, one line each, with the placement probe's own timings (QN_H_PLACEMENT=2: candidates' update-kernel times, kept).
# Round 3 findings, same box: it does not follow H's virtual address or a pooled allocation; with non-temporal loads of Q in the
# evaluation kernel the slow mode did not appear in 20 runs but the evaluation lost 1.5 us; with non-temporal accesses to H the update
# kernel takes 34.6 us always -- at n = 4096 both half matrices live in the 256 MB Infinity Cache, and how evenly their physical pages
# spread over its slices differs from process to process.  Round 4: the placement probe (place_h, qn_hip.hip) times the update kernel
# itself on H and on fresh allocations and keeps the fastest.
# usage: [DIM=n] bash tools/modes_ab.sh [runs]
for i in $(seq 1 ${1:-10}); do QN_H_PLACEMENT=2 python bench.py --steps 60 --no-cpu-baseline ${DIM:+--dim $DIM} 2> /tmp/modes_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']; print('it/s %.0f  update %.2f us  eval %.2f us' % (d['value'], 1e3 * r['update_pass']['avg_launch_ms'], 1e3 * r['quad_matvec']['avg_launch_ms']), end='   ')"; grep "H placement" /tmp/modes_err.txt | head -1; done
