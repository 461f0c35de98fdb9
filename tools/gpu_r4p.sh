#!/bin/bash
# r03k hunt, third step: (1) the stand-alone probe (tools/probes/null_stream_order.hip); (2) the shortest failing pair of
# tests/ under AMD_LOG_LEVEL=4: the AQL packets (queue, barrier bit) of the fused handle's first plan.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
./tools/probes/null_stream_order.bin > gpurun_out/r4p_null_stream_probe.txt 2>&1; cat gpurun_out/r4p_null_stream_probe.txt
CAP='tests/test_gpu_boundary.py::test_plan_is_hip_graph_capturable[c1]'
PRE='tests/test_gpu_layered.py::test_td_target_says_nan_when_a_wait_gave_up'
AMD_LOG_LEVEL=4 timeout 300 python -m pytest -m gpu -q --tb=line -p no:cacheprovider "$PRE" "$CAP" > gpurun_out/r4p_pytest.out 2> /tmp/r4p_amd.log
tail -3 gpurun_out/r4p_pytest.out
wc -l /tmp/r4p_amd.log
grep -E "Dispatch Header|Barrier|ShaderName|hipStreamWaitEvent|hipEventRecord|hipStreamCreate|hipMemsetAsync|hipMemcpy|hipDeviceSynchronize|hipStreamSynchronize|hipFree|hipMalloc" /tmp/r4p_amd.log | cut -c1-420 > /tmp/r4p_filtered.log
wc -l /tmp/r4p_filtered.log
# everything from the fused handle's first kernel (ks_setup) on, and the 400 lines before it
L=$(grep -n "ShaderName : .*ks_setup" /tmp/r4p_filtered.log | head -1 | cut -d: -f1)
echo "first ks_setup at filtered line $L"
if [ -n "$L" ]; then S=$((L > 400 ? L - 400 : 1)); sed -n "${S},$((L + 500))p" /tmp/r4p_filtered.log | gzip > gpurun_out/r4p_amd_log_around_plan.txt.gz; fi
grep -c "barrier=0" /tmp/r4p_filtered.log; grep -c "barrier=1" /tmp/r4p_filtered.log
grep -o "HWq=0x[0-9a-f]*" /tmp/r4p_filtered.log | sort | uniq -c
ls -la gpurun_out/r4p*
