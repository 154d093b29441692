#!/bin/bash
# A/B of role A's split (round 6): for each QN_LU_SPLIT value a rocprofv3 kernel trace of tools/newton_time.py 8192 lu, read by
# tools/lu_timeline.py (per-kernel totals by queue, panel periods, the launches around a few panels) -> gpurun_out/lus/timeline_G<g>.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/lus
for g in ${SPLITS:-1 4}; do
  rm -rf gpurun_out/lu_prof
  QN_LU_SPLIT=$g timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lu_prof -o p -- python3 tools/newton_time.py 8192 lu > gpurun_out/lus/prof_G$g.log 2>&1 || exit 1
  python3 tools/lu_timeline.py ${PANELS:-10 40 64 100} > gpurun_out/lus/timeline_G$g.txt 2>&1 || exit 1
done
rm -rf gpurun_out/lu_prof
