#!/bin/bash
# full GPU suite + default bench with the cluster path (heads across members, straight-line shares)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x --timeout 600 > gpurun_out/r02w_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02w_pytest.log
tail -12 gpurun_out/r02w_pytest.log
timeout 300 python tools/probes/cluster_latency.py > gpurun_out/r02w_latency.txt 2>&1; cat gpurun_out/r02w_latency.txt
timeout 900 python bench.py > gpurun_out/r02w_bench.json 2> gpurun_out/r02w_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02w_bench.json').read())
print(d['value'], d['roofline']['avg_launch_ms'], {k:v for k,v in d['extra'].items() if 'latency' in k})
print({k:(v.get('value'), v.get('latency_ms_single_env')) for k,v in d['extra'].get('configs',{}).items()})
PY
