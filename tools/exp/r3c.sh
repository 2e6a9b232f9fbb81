#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q -k "tiny or zero_vel or converged_box or prepass or confirming or host_program" > gpurun_out/r3c_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3c_pytest.log
tail -4 gpurun_out/r3c_pytest.log
rm -rf gpurun_out/r3c_trace; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3c_trace -- python3 bench.py --no-cpu --no-traffic --no-host --no-hbm-regime --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 0 > gpurun_out/r3c_six_traced.json 2> gpurun_out/r3c_six_traced.err; echo "trace rc=$?"
python tools/exp/trace_six.py gpurun_out/r3c_trace 270 > gpurun_out/r3c_trace_summary.txt 2>&1; cat gpurun_out/r3c_trace_summary.txt
find gpurun_out/r3c_trace -name "*.csv" -size +20M -delete
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --grid 512,512,256 --starts 111 --nstarts 8 --steps 2 --warmup 1 --no-cpu > gpurun_out/r3c_gloo2_512.json 2> gpurun_out/r3c_gloo2_512.err; echo "gloo2 rc=$?"; tail -c 600 gpurun_out/r3c_gloo2_512.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --no-cpu > gpurun_out/r3c_gloo2_241.json 2> gpurun_out/r3c_gloo2_241.err; echo "gloo2-241 rc=$?"; tail -c 300 gpurun_out/r3c_gloo2_241.json
