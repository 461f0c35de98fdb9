#!/bin/bash
# round-2 evidence run: full -m gpu suite, the default bench line, the torchrun (RCCL, 1 rank) leg, rocprofv3 kernel stats, PMC passes
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider) > gpurun_out/r02aa_pytest_gpu.log 2>&1
tail -8 gpurun_out/r02aa_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r02aa_bench.json 2> gpurun_out/r02aa_bench.err
tail -5 gpurun_out/r02aa_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02aa_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["extra"].get("latency_ms_single_env"), {k:(v.get("value"), v.get("roofline",{}).get("frac")) for k,v in d["extra"].get("configs",{}).items()}, d.get("cpu_baseline",{}).get("value"))
PY
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --skip-cpu-baseline --skip-extra-configs > gpurun_out/r02aa_torchrun_n1.log 2>&1
tail -c 600 gpurun_out/r02aa_torchrun_n1.log; echo
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02aa -o r02aa -- python $R/bench.py --steps 5 --warmup 2 --skip-cpu-baseline --skip-extra-configs > $R/gpurun_out/prof_r02aa.stdout 2> $R/gpurun_out/prof_r02aa.stderr
cd $R
find gpurun_out/prof_r02aa -name "*.csv" | head
KT=$(find gpurun_out/prof_r02aa -name "*kernel_trace.csv" | head -1)
python tools/rocprof_summary.py $KT > gpurun_out/r02aa_kernel_stats_by_grid.txt
cp $(find gpurun_out/prof_r02aa -name "*kernel_stats.csv" | head -1) gpurun_out/r02aa_rocprofv3_kernel_stats.csv
head -12 gpurun_out/r02aa_kernel_stats_by_grid.txt
bash tools/gpu_pmc.sh r02aa
python tools/pmc_summary.py gpurun_out/pmc_r02aa ks_rollout > gpurun_out/r02aa_pmc.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r02aa --json gpurun_out/r02aa_pmc.json 'ks_rollout<\d+, \d, 8, 0, 0>'
head -40 gpurun_out/r02aa_pmc.txt
rm -rf gpurun_out/pmc_r02aa/*/ 2>/dev/null; du -sh gpurun_out | tail -1
