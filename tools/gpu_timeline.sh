#!/bin/bash
# Kernel timelines (start / end / queue of every launch) of bench configurations, reduced to the columns tools/timeline.py reads and
# gzipped, + timeline.py's summary (idle / one / two kernels running per stage, time per kernel class, one stage's launches in order).
# usage: gpurun -- bash tools/gpu_timeline.sh <tag> "<config envs steps>;..." [ENV=... ...]
cd "$(dirname "$0")/.."
R=$PWD; TAG=$1; SPECS=$2; shift 2
EXTRA_ENV=("$@")  # (kept here: `set -- $spec` below replaces the positional parameters)
mkdir -p gpurun_out; export TMPDIR=/tmp
IFS=';' read -r -a SP <<< "$SPECS"
for spec in "${SP[@]}"; do
  set -- $spec
  n=${TAG}_$1e$2
  (cd /tmp && env "${EXTRA_ENV[@]}" TDMPC2_BENCH_EXACT_STEPS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$n -o t -- python $R/bench.py --config $1 --envs $2 --steps $3 --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic > /dev/null 2>&1)
  KT=$(find gpurun_out/prof_$n -name "*kernel_trace.csv" | head -1)
  python - "$KT" gpurun_out/${n}_timeline.csv <<'PY'
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
  gzip -f gpurun_out/${n}_timeline.csv
  python tools/timeline.py gpurun_out/${n}_timeline.csv.gz $2 dump > gpurun_out/${n}_timeline.txt 2>&1
  head -30 gpurun_out/${n}_timeline.txt | cut -c1-130
  rm -rf gpurun_out/prof_$n
done
