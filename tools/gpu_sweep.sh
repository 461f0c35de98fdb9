#!/bin/bash
# SURVEY.md section 8(d)'s list of configurations, one bench.py line each (run on the GPU box through gpurun):
#   c1 at E = 1, 16, 64, 512 | c2 at I = 6 (metric text) and I = 8 (reference behaviour, A >= 20) | c3: 30 concurrent plans |
#   c4 at the reference dims (L = 1376) and at BASELINE.json's L = 1024 | c5's per-GPU share: c4 with 64 environments
# usage: tools/gpu_sweep.sh <tag>   ->  gpurun_out/sweep_<tag>.jsonl
set -u
TAG="${1:-r01}"
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/sweep_${TAG}.jsonl"
mkdir -p "$R/gpurun_out"; : > "$OUT"
run() {
  echo "== bench.py $*" >&2
  timeout 600 python "$R/bench.py" --skip-cpu-baseline --skip-traffic --skip-extra-configs "$@" 2>/dev/null | tail -1 >> "$OUT" || echo "{\"error\": \"$*\"}" >> "$OUT"
}
run --config c1 --envs 1 --steps 20 --warmup 3
run --config c1 --envs 16 --steps 20 --warmup 3
run --config c1 --envs 64 --steps 20 --warmup 3
run --config c1 --envs 512 --steps 10 --warmup 2
run --config c2 --envs 256 --iterations 6 --steps 20 --warmup 3
run --config c2 --envs 256 --iterations 8 --steps 20 --warmup 3
run --config c3 --envs 30 --steps 5 --warmup 2
run --config c4 --envs 8 --steps 3 --warmup 1
run --config c4_l1024 --envs 8 --steps 3 --warmup 1
run --config c4 --envs 64 --steps 2 --warmup 1
python - "$OUT" <<'PY'
import json, sys
print("| workload | envs | I | family / arithmetic | plans/s | ms per step | rollout TFLOP/s as-written (frac of f16 peak) | E=1 latency ms |")
print("|---|---|---|---|---|---|---|---|")
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
        c = d["config"]
        print(f"| {c['workload'].split(':')[0]} | {c['envs_per_gpu']} | {c['iterations']} | {c['kernel_family']} / {c['arithmetic'].split(' (')[0]} | "
              f"{d['value']} | {d['ms_per_step']} | {d['roofline']['achieved']} ({d['roofline']['frac']}) | {d['extra'].get('latency_ms_single_env')} |")
    except Exception as ex:
        print("| error |", line.strip()[:100], "|")
PY
