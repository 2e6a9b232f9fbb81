#!/bin/bash
# the default bench line (N = 1) and the two-rank rehearsals on the one GPU of the box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_line.json 2> gpurun_out/r4_bench_line.err; echo "bench rc $?"
tail -c 600 gpurun_out/r4_bench_line.json
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --no-host > gpurun_out/r4_two_rank_nccl.json 2> gpurun_out/r4_two_rank_nccl.err; echo "two-rank nccl rc $?"
tail -c 1500 gpurun_out/r4_two_rank_nccl.json; tail -5 gpurun_out/r4_two_rank_nccl.err
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --no-host --no-cpu --backend gloo > gpurun_out/r4_two_rank_gloo.json 2> gpurun_out/r4_two_rank_gloo.err; echo "two-rank gloo rc $?"
tail -c 700 gpurun_out/r4_two_rank_gloo.json
