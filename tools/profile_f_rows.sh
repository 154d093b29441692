#!/bin/bash
# One GPU-box call for the f-row profiles: config 5 (DFP + log-sum-exp, default and searching instance) and Newton (Cholesky and forced LU):
# rocprofv3 --kernel-trace --stats summaries, and separate --pmc passes (FETCH_SIZE / WRITE_SIZE) for the one-pass log-sum-exp kernel.
# usage (through gpurun, from the repo root): bash tools/profile_f_rows.sh <tag>
tag=${1:-frows}
out=gpurun_out/$tag
repo="${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel 2>/dev/null || pwd)}"
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$repo"
python3 tools/bench_config5.py 16384 20 > $out/config5_default.json 2> $out/config5_default.err
python3 tools/bench_config5.py 16384 20 300 10 30 > $out/config5_search.json 2> $out/config5_search.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -o c5 -- python3 tools/bench_config5.py 16384 10 > $out/c5_rocprof.json 2> $out/c5_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5s -o c5s -- python3 tools/bench_config5.py 16384 10 300 10 30 > $out/c5s_rocprof.json 2> $out/c5s_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/c5_fetch -o f -- python3 tools/bench_config5.py 16384 4 > /dev/null 2> $out/c5_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/c5_write -o w -- python3 tools/bench_config5.py 16384 4 > /dev/null 2> $out/c5_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/newton -o nw -- python3 tools/newton_time.py 8192 > $out/newton_chol.txt 2> $out/newton_chol.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/newton_lu -o nl -- python3 tools/newton_time.py 8192 lu > $out/newton_lu.txt 2> $out/newton_lu.err
cat $out/config5_default.json $out/config5_search.json | cut -c1-900
tail -2 $out/newton_chol.txt $out/newton_lu.txt
find $out -name "*kernel_stats.csv" | head
