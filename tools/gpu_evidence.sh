#!/bin/bash
# Evidence run of a round (MI355X box, through gpurun): full -m gpu suite, the default bench line, the torchrun
# (RCCL, 1 rank) leg, rocprofv3 kernel stats and the PMC passes of the same command.
# usage: gpurun --timeout 2400 -- bash tools/gpu_evidence.sh <tag> [parts]     parts: any of "tests bench stats pmc_layered ab pmc torchrun" (default: all but ab)
# Parts run in the order of the script (most important first: a call cut short by the GPU budget still leaves the suite, the
# bench line and the kernel stats).  PMC_PASSES (tools/gpu_pmc.sh) limits the counter passes; AB_SPEC / AB_ENVS feed the `ab` part
# (tools/gpu_env_ab.sh).
cd "$(dirname "$0")/.."
R=$PWD
TAG="${1:-r3z}"; PARTS="${2:-tests bench stats pmc_layered pmc torchrun}"
mkdir -p gpurun_out
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has tests; then
  (time timeout 2700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
  tail -8 gpurun_out/${TAG}_pytest_gpu.log
fi
if has bench; then
  timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
  tail -5 gpurun_out/${TAG}_bench.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic"), d["extra"].get("latency_ms_single_env"),
      {k:(v.get("value"), v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("traffic")) for k,v in d["extra"].get("configs",{}).items()}, d.get("cpu_baseline",{}).get("value"))
PY
fi
if has stats; then
  export TMPDIR=/tmp
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -o ${TAG} -- python $R/bench.py --steps 5 --warmup 2 --skip-cpu-baseline --skip-traffic > $R/gpurun_out/prof_${TAG}.stdout 2> $R/gpurun_out/prof_${TAG}.stderr
  cd $R
  KT=$(find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" | head -1)
  python tools/rocprof_summary.py $KT > gpurun_out/${TAG}_kernel_stats_by_grid.txt
  cp $(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_rocprofv3_kernel_stats.csv
  head -14 gpurun_out/${TAG}_kernel_stats_by_grid.txt | cut -c1-160
  rm -rf gpurun_out/prof_${TAG}
fi
if has pmc_layered; then  # the layered family's GEMMs at the c3 / c4 geometry of the bench legs
  PMC_PASSES="${PMC_PASSES_LAYERED:-sq1 grbm fetch}" bash tools/gpu_pmc.sh ${TAG}_c3 --config c3 --envs 30 --steps 2 --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_c3 g_gemm > gpurun_out/${TAG}_c3_pmc.txt 2>&1
  PMC_PASSES="${PMC_PASSES_LAYERED:-sq1 grbm fetch}" bash tools/gpu_pmc.sh ${TAG}_c4 --config c4 --envs 8 --steps 1 --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_c4 g_gemm > gpurun_out/${TAG}_c4_pmc.txt 2>&1
  grep -A16 "g_gemm_w<1, 0>  workgroups=448\|g_gemm_w<1, 0>  workgroups=512" gpurun_out/${TAG}_c3_pmc.txt gpurun_out/${TAG}_c4_pmc.txt | grep -E "g_gemm_w|pipe|clock|HBM" | head -20
  rm -rf gpurun_out/pmc_${TAG}_c3/*/ gpurun_out/pmc_${TAG}_c4/*/ 2>/dev/null
fi
if has ab; then  # environment-switch A/B of the in-tree library (interleaved twice), e.g. AB_SPEC="c3 30 8" AB_ENVS="A=0|TDMPC2_X_GEMM_XCD_ROWS=0"
  IFS='|' read -r -a ABE <<< "${AB_ENVS:-A=0}"
  bash tools/gpu_env_ab.sh ${TAG} "${AB_SPEC:-c3 30 8}" "${ABE[@]}" > /dev/null
  cat gpurun_out/${TAG}_ab.txt
  if [ -n "${AB_SPEC2:-}" ]; then  # a second, single-repetition A/B (the slow configuration)
    AB_REPS=1 bash tools/gpu_env_ab.sh ${TAG}b "${AB_SPEC2}" "${ABE[@]}" > /dev/null
    cat gpurun_out/${TAG}b_ab.txt
  fi
fi
if has pmc; then
  bash tools/gpu_pmc.sh ${TAG}
  python tools/pmc_summary.py gpurun_out/pmc_${TAG} ks_rollout > gpurun_out/${TAG}_pmc.txt 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_${TAG} --json gpurun_out/${TAG}_pmc.json 'ks_rollout<\d+, \d, 8, 0, 0>'
  head -40 gpurun_out/${TAG}_pmc.txt
  rm -rf gpurun_out/pmc_${TAG}/*/ 2>/dev/null
fi
if has torchrun; then
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic > gpurun_out/${TAG}_torchrun_n1.log 2>&1
  tail -c 600 gpurun_out/${TAG}_torchrun_n1.log; echo
  TDMPC2_BENCH_FORCE_C5=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 2 --warmup 1 --skip-cpu-baseline --skip-traffic > gpurun_out/${TAG}_torchrun_c5_leg.log 2>&1
  tail -c 400 gpurun_out/${TAG}_torchrun_c5_leg.log; echo
fi
du -sh gpurun_out | tail -1
