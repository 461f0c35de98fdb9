#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
cd /tmp
for c in c4:8 c3:30; do
name=${c%%:*}; E=${c##*:}
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$name -o $name -- python $R/bench.py --config $name --envs $E --steps 2 --warmup 1 --skip-cpu-baseline > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_$name -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/r02j_${name}_kernel_stats_by_grid.txt
grep -v "g_gemm \|l_ln_act<\|l_pi_head \|l_init_x \|l_set_action " $R/gpurun_out/r02j_${name}_kernel_stats_by_grid.txt | head -16 | cut -c1-150
rm -rf $R/gpurun_out/prof_$name
done
