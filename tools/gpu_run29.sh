#!/bin/bash
# final check of the round's last library state: full -m gpu suite + the default bench line
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider) > gpurun_out/r02ae_pytest_gpu.log 2>&1
tail -6 gpurun_out/r02ae_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r02ae_bench.json 2> gpurun_out/r02ae_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02ae_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], {k:v for k,v in d["extra"].items() if "latency" in k}, {k:(v.get("value"), v.get("latency_ms_single_env")) for k,v in d["extra"].get("configs",{}).items()})
PY
