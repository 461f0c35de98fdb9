#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_layered.py tests/test_gpu_td_target.py tests/test_gpu_edge.py tests/test_gpu_dist.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Abort|fault|FAILED" | head -5
for rt4 in 1 0; do
python - <<PY
import os, sys, json
if $rt4: os.environ["TDMPC2_GEMM_RT4"]="1"
sys.path.insert(0, ".")
import torch, bench
dev=torch.device("cuda",0)
for name,E,k in (("c3",30,3),("c4",8,2)):
    r=bench.config_leg(name,E,k,dev)
    print("RT4-forced" if $rt4 else "auto", name, "plans/s", r["value"], "stage_ms", r["roofline"]["avg_stage_ms"], "lat1_ms", r["latency_ms_single_env"], flush=True)
PY
done
