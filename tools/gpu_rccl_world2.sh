#!/bin/bash
# bench.py --gpus 2 as two PROCESSES on the ONE GPU of the builder's pool, over RCCL (TDMPC2_BENCH_ONE_GPU=1): the N > 1 job logic
# (process-group init, weight broadcast, barrier, all_reduce(MAX), the c5 leg) end to end at world size 2.  A dry run, not a measurement.
# usage: gpurun -- bash tools/gpu_rccl_world2.sh <tag>
cd "$(dirname "$0")/.."
TAG=${1:-r6}
mkdir -p gpurun_out
log=gpurun_out/${TAG}_rccl_world2_one_gpu.log
{
  echo "# torchrun --nproc-per-node 2 bench.py --gpus 2, both ranks on cuda:0 (TDMPC2_BENCH_ONE_GPU=1), backend nccl (= RCCL)"
  TDMPC2_BENCH_ONE_GPU=1 NCCL_DEBUG=WARN HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --envs 64 --steps 5 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>&1 | grep -v "^\s*File\|^    " | tail -120
  echo "# exit status: ${PIPESTATUS[0]}"
} > $log 2>&1
tail -30 $log
