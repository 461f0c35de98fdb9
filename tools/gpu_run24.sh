#!/bin/bash
# kernel trace of single plans on the cluster path: per-launch durations (launch 0 carries the policy prior)
R="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$R/gpurun_out/prof_r02z"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
CLUSTER_MODES=1 CLUSTER_ENVS=1 TDMPC2_PLAN_LIB=$R/build/ablate/lib_${LIBV:-pf}.so timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python "$R/tools/probes/cluster_latency.py" c2 > "$OUT/stdout.txt" 2>&1
cd "$R"; python tools/rocprof_summary.py $(find "$OUT" -name "*kernel_trace.csv" | head -1) > gpurun_out/r02z_kernels.txt 2>&1; grep -E "ks_|k_refit|# kernel" gpurun_out/r02z_kernels.txt | head
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_r02z/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
# last plan: find last ks_setup
idx=max(i for i,n in enumerate(names) if 'ks_setup' in n)
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[idx:idx+9]:
    print(r['Kernel_Name'][:40], 'start', (int(r['Start_Timestamp'])-t0)/1e3, 'us  dur', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
