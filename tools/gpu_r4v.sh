#!/bin/bash
# Timelines of a c3 (E = 30) and a c4 (E = 8) step: the kernel trace itself (start / end / queue of every launch), reduced to the
# columns tools/timeline.py reads and gzipped, for an offline look at gaps and at how the two chains overlap.
cd "$(dirname "$0")/.."
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
for spec in "c3 30 4" "c4 8 2"; do
  set -- $spec
  (cd /tmp && TDMPC2_BENCH_EXACT_STEPS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r4v_$1 -o t -- python $R/bench.py --config $1 --envs $2 --steps $3 --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic > /dev/null 2>&1)
  KT=$(find gpurun_out/prof_r4v_$1 -name "*kernel_trace.csv" | head -1)
  python - "$KT" gpurun_out/r4v_$1_timeline.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("start_ns,end_ns,queue,grid,wg,kernel\n")
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        f.write(f'{int(r["Start_Timestamp"]) - t0},{int(r["End_Timestamp"]) - t0},{r.get("Queue_Id", "")},{int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))},{r["Workgroup_Size_X"]},"{name}"\n')
print(len(rows), "launches")
PY
  gzip -f gpurun_out/r4v_$1_timeline.csv
  rm -rf gpurun_out/prof_r4v_$1
done
ls -la gpurun_out/r4v*
