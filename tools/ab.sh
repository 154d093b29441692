#!/bin/bash
# same-box A/B of two builds of the library: bash tools/ab.sh <libA> <libB> [bench args]; alternating runs of bench.py.
# Prints the rate, the iteration time and the event-bracket averages of the four launch kinds (synchronous profiling pass).
a=$1; b=$2; shift 2
for rep in 1 2 3; do
  for lib in $a $b; do
    v=$(QN_HIP_LIB=$PWD/optimization-solvers_amd/lib/$lib python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline')
line='%.1f it/s  %.2f us' % (d['value'], 1e3*d['ms_per_step'])
if r:
    q=r['quad_matvec']; u=r['update_pass']
    line+=' | eval %.2f vec %.2f hpass %.2f hreduce %.2f us' % (1e3*q['avg_launch_ms'], 1e3*(q['accept_reduce_avg_launch_ms'] or 0), 1e3*u['avg_launch_ms'], 1e3*((u['pass_with_reduce'] or {}).get('reduce_avg_launch_ms') or 0))
print(line)")
    echo "$lib $v"
  done
done
