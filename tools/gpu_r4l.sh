#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/gpu_ab.sh r4l2_c4 "base spread" --config c4 --envs 8 --steps 4 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic
bash tools/gpu_ab.sh r4l2_c3 "base spread" --config c3 --envs 30 --steps 8 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic
