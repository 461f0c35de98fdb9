#!/bin/bash
# Sub-batches again (profiles/README.md r4d), this time WITHOUT the cross-workgroup wait of the fused epilogue (PROBE_UNFUSED=1):
# does the chip fill its rounds when 4 or 6 smaller kernels are in flight and none of their workgroups waits for a peer?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r4x_split_batch_unfused.txt; : > $OUT
echo "== fused epilogue" >> $OUT
timeout 300 python tools/probes/split_batch_probe.py c3 30 2 2>&1 | grep -v amdgpu.ids >> $OUT
echo "== PROBE_UNFUSED=1" >> $OUT
PROBE_UNFUSED=1 timeout 300 python tools/probes/split_batch_probe.py c3 30 2 3 2>&1 | grep -v amdgpu.ids >> $OUT
echo "== PROBE_UNFUSED=1 TDMPC2_GEMM_W256_MIN=128" >> $OUT
PROBE_UNFUSED=1 TDMPC2_GEMM_W256_MIN=128 timeout 300 python tools/probes/split_batch_probe.py c3 30 2 3 2>&1 | grep -v amdgpu.ids >> $OUT
cat $OUT
