# Diagnostic: the update-tile kernel has two modes between PROCESSES (26 or 30.5 us in bench.py's event brackets for a whole run, about
# one process in five in the slow one): bench.py N times, one line each.  Round 3 findings, same box: it does not follow H's virtual
# address or a pooled allocation; with non-temporal loads of Q in the evaluation kernel the slow mode did not appear in 20 runs but
# the evaluation lost 1.5 us; with non-temporal accesses to H the update kernel takes 34.6 us always -- at n = 4096 both half
# matrices live in the 256 MB Infinity Cache, and how evenly their physical pages spread over its slices differs from process to process.
# A placement step was tried as well (time the iteration's access pattern on H where it is and on two freshly allocated copies,
# keep the fastest: a plain read + write-back of H's half takes 13.3 us, 14.3-15.8 us for about one allocation in four): it caught
# some of the slow processes, but one whose H timed as fast as any still ran the update kernel at 30 us -- not adopted.
# usage: bash tools/modes_ab.sh [runs]
for i in $(seq 1 ${1:-10}); do python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('it/s %.0f  update %.2f us  eval %.2f us' % (d['value'], 1e3 * r['update_pass']['avg_launch_ms'], 1e3 * r['quad_matvec']['avg_launch_ms']))"; done
