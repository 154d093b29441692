#!/bin/bash
# One GPU-box call at the end of round 6: the driver's bench line (with extra_configs), the default one, rocprofv3 kernel summaries of
# the headline run and of config 5 on its new path, the PMC traffic passes.  usage (through gpurun): bash tools/round6_final.sh <tag>
tag=${1:-r06_f}
out=gpurun_out/$tag
repo="${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel 2>/dev/null || pwd)}"
mkdir -p $out
for i in 1 2 3; do
  ( time python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_steps20_$i.json 2> $out/bench_steps20_$i.err ) 2> $out/bench_steps20_$i.time || echo "bench20 $i failed"
done
python bench.py --no-extra-configs > $out/bench_default.json 2> $out/bench_default.err || echo "bench default failed"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o prof -- python3 bench.py --steps 200 --no-cpu-baseline --no-profile-pass --no-extra-configs > $out/bench_under_rocprof.json 2> $out/rocprof.err || echo "rocprof failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -o c5 -- python3 tools/bench_config5.py 16384 20 > $out/c5_rocprof.json 2> $out/c5_rocprof.err || echo "rocprof c5 failed"
bash tools/pmc_traffic.sh $tag/pmc > $out/pmc.txt 2>&1
python3 - "$out" <<'PY'
import json, sys, glob, csv
out = sys.argv[1]
for i in (1, 2, 3):
    try:
        d = json.loads(open(f"{out}/bench_steps20_{i}.json").read().strip().splitlines()[-1])
        e = d.get("extra_configs", {})
        c4, c5 = e.get("config4_newton_n8192", {}), e.get("config5_dfp_logsumexp_n16384", {})
        print(f"steps20 #{i}: {d['value']:.0f} it/s, {d['ms_per_step']*1e3:.2f} us; cpu {d['cpu_baseline'].get('value')}; newton chol {c4.get('cholesky',{}).get('ms_per_iteration')} lu {c4.get('lu',{}).get('ms_per_iteration')}; config5 {c5.get('value')} it/s whole {c5.get('whole_iteration_hbm_frac')}; wall {open(f'{out}/bench_steps20_{i}.time').read().split()[1]}")
    except Exception as ex:
        print("steps20", i, "unreadable", ex)
d = json.loads(open(f"{out}/bench_default.json").read().strip().splitlines()[-1])
print(f"default: {d['value']:.0f} it/s, {d['ms_per_step']*1e3:.2f} us, whole {d['iteration_accounting']['whole_iteration_hbm_frac']:.3f}, roofline frac {d['roofline']['frac']:.3f}")
for tag in ("prof", "c5"):
    f = glob.glob(f"{out}/{tag}/**/*kernel_stats.csv", recursive=True)
    if f:
        for r in sorted(csv.DictReader(open(f[0])), key=lambda r: -float(r["TotalDurationNs"]))[:8]:
            print("  %-5s %-64s calls %6s avg %9.2f us %5s %%" % (tag, r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cat $out/pmc.txt | tail -14
# round 6: the evaluation tiles as mover + multiplier waves against round 5's kernel (same bits), alternating repetitions, and their in-kernel stamps
bash tools/ring_ab.sh 3 > $out/ring_ab.txt 2>&1; cat $out/ring_ab.txt
if [ -f optimization-solvers_amd/lib/libqn_hip_stamps.so ]; then
  QN_STAMPS_RING=1 python3 tools/s2_stamps.py 4096 12 > $out/ring_stamps.txt 2>&1
  QN_S2_RING=0 python3 tools/s2_stamps.py 4096 12 > $out/pair_stamps.txt 2>&1
  grep -v "accept-reduce\|update" $out/ring_stamps.txt | sed -n 2,5p
  grep -v "accept-reduce\|update" $out/pair_stamps.txt | sed -n 2,4p
fi
# round 6, second half: what the XCDs' L2 keeps across kernel boundaries -- the zig-zag order and the touch workgroups against the kernels without them
REPS=2 bash tools/prof_env_ab.sh "l2_off:QN_S2_ZIGZAG=0,QN_S2_TOUCH=0,QN_S2_TOUCHQ=0" "zigzag:QN_S2_TOUCH=0,QN_S2_TOUCHQ=0" "default:QN_S2_ZIGZAG=1" > $out/l2_ab.txt 2>&1; cat $out/l2_ab.txt
