#!/bin/bash
# same-box comparison of several builds of the library: bash tools/ab3.sh "<libA> <libB> ..." [bench args]; alternating runs of bench.py
libs=$1; shift
for rep in 1 2 3; do
  for lib in $libs; do
    v=$(QN_HIP_LIB=$PWD/optimization-solvers_amd/lib/$lib python bench.py --no-cpu-baseline --no-profile-pass "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f it/s  %.2f us' % (d['value'], 1e3*d['ms_per_step']))")
    echo "$lib $v"
  done
done
