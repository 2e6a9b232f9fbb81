#!/bin/bash
# two ranks on the one GPU of the box: the sharded path rehearsed (nccl is refused by RCCL there: the guarded fallback), then under gloo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --no-host > gpurun_out/r5_two_rank_nccl.json 2> gpurun_out/r5_two_rank_nccl.err; echo "two-rank nccl rc $?"
tail -c 400 gpurun_out/r5_two_rank_nccl.json; echo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --no-host --no-cpu --backend gloo > gpurun_out/r5_two_rank_gloo.json 2> gpurun_out/r5_two_rank_gloo.err; echo "two-rank gloo rc $?"
tail -c 400 gpurun_out/r5_two_rank_gloo.json; echo
