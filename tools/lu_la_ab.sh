#!/bin/bash
# A/B of the LU look-ahead (round 4): QN_LU_LOOKAHEAD, QN_LU_BULK_CUS, QN_LU_BULK_PERSIST -> gpurun_out/lu_la_ab.log
out=gpurun_out/lu_la_ab.log; : > $out
for cfg in ${CFGS:-"1 192 1" "1 192 0" "1 256 1" "1 224 1" "0 192 0"}; do
  set -- $cfg
  echo "== QN_LU_LOOKAHEAD=$1 QN_LU_BULK_CUS=$2 QN_LU_BULK_PERSIST=$3" >> $out
  QN_LU_LOOKAHEAD=$1 QN_LU_BULK_CUS=$2 QN_LU_BULK_PERSIST=$3 timeout -k 10 120 python tools/newton_time.py 8192 lu >> $out 2>&1 || exit 1
done
