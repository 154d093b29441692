#!/bin/bash
# A/B: rows of A as non-temporal loads in lse_onepass_kernel (QN_LSE_NT) at n = m = 16384, alternating runs
for i in 1 2; do
  for nt in 0 1; do
    QN_LSE_NT=$nt python3 tools/bench_config5.py 16384 20 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('QN_LSE_NT=$nt', 'it/s %.1f' % d['iterations_per_s'], 'eval wall ms %.3f' % d['objective_eval']['wall_ms_incl_host_copies'], 'evals/it %.2f' % d['oracle_evals_per_iteration'])"
  done
done
