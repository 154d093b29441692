for i in $(seq 1 ${1:-12}); do QN_DEBUG_PLACEMENT=1 QN_HIP_LIB=$PWD/optimization-solvers_amd/lib/libqn_x1.so python bench.py --steps 60 --no-cpu-baseline 2> /tmp/pl_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('it/s %.0f  update %.2f us  eval %.2f us' % (d['value'], 1e3 * r['update_pass']['avg_launch_ms'], 1e3 * r['quad_matvec']['avg_launch_ms']), end='  |')"; grep "qn placement" /tmp/pl_err.txt | head -3 | sed 's/\[qn placement\]//; s/ at 0x[0-9a-f]*//' | tr '\n' ';'; echo; done
