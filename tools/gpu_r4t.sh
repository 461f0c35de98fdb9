#!/bin/bash
# XCD rectangles (a row block of the 317M model's hidden GEMMs on 2 XCDs) adopted: layered tests + the two-chain stress test,
# the c4 bench leg with its traffic children (stage traffic), the fetch pass of the c4 geometry, and the A/B against XCD-local rows.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_layered.py tests/test_gpu_philox.py -m gpu -q --tb=short -p no:cacheprovider -x) > gpurun_out/r4t_pytest_layered.log 2>&1
tail -3 gpurun_out/r4t_pytest_layered.log
TDMPC2_BENCH_EXACT_STEPS=1 timeout 600 python bench.py --config c4 --envs 8 --steps 13 --warmup 2 --skip-cpu-baseline --skip-extra-configs > gpurun_out/r4t_c4_bench.json 2> gpurun_out/r4t_c4_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r4t_c4_bench.json").read().strip().splitlines()[-1])
print("c4", d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["extra"].get("bounded_wait_faults"))
PY
PMC_PASSES="fetch grbm sq1" bash tools/gpu_pmc.sh r4t_c4 --config c4 --envs 8 --steps 1 --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic > /dev/null
python tools/pmc_summary.py gpurun_out/pmc_r4t_c4 g_gemm > gpurun_out/r4t_c4_pmc.txt 2>&1
grep -A24 "g_gemm_w<1>  workgroups=512" gpurun_out/r4t_c4_pmc.txt | grep -E "g_gemm_w|pipe|clock|HBM"
rm -rf gpurun_out/pmc_r4t_c4
bash tools/gpu_env_ab.sh r4t "c4 8 13" "A=0" "TDMPC2_GEMM_W_XCD_ROWS=1" > /dev/null; cat gpurun_out/r4t_ab.txt
