#!/bin/bash
# same-box A/B of two builds of the library: bash tools/ab.sh <libA> <libB> [bench args]; prints value per run, alternating
a=$1; b=$2; shift 2
for rep in 1 2 3; do
  for lib in $a $b; do
    v=$(QN_HIP_LIB=$PWD/optimization-solvers_amd/lib/$lib python bench.py --no-cpu-baseline --no-profile-pass "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s  %.2f us' % (d['value'], 1e3*d['ms_per_step']))")
    echo "$lib $v"
  done
done
