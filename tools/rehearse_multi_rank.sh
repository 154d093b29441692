set -o pipefail
mkdir -p gpurun_out/r06_h
export QN_BENCH_EXCHANGE=host
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/r06_h/rehearsal_2ranks.json 2> gpurun_out/r06_h/rehearsal_2ranks.err
echo "rc2=$?"
QN_BENCH_TRIAL_VECTOR=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 10 --warmup 2 --no-scaling-ref > gpurun_out/r06_h/rehearsal_4ranks_tv.json 2> gpurun_out/r06_h/rehearsal_4ranks_tv.err
echo "rc4=$?"
tail -c 1500 gpurun_out/r06_h/rehearsal_2ranks.json; echo; tail -c 600 gpurun_out/r06_h/rehearsal_2ranks.err; tail -c 1500 gpurun_out/r06_h/rehearsal_4ranks_tv.json; tail -c 600 gpurun_out/r06_h/rehearsal_4ranks_tv.err
