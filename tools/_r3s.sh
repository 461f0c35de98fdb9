mkdir -p gpurun_out
bash tools/gpu_env_ab.sh r3s "c3 30 8;c4 8 3" "A=0" "TDMPC2_GEMM_COL_PAD=1" > /dev/null; cat gpurun_out/r3s_ab.txt
GO=$(python - <<'PY'
import re
t=open("gpurun_out/r3s_ab.txt").read().split("== ")
base,new=[],[]
for blk in t:
    if " c3 " not in blk.split("\n")[0]: continue
    m=re.search(r"lat1_ms ([0-9.]+)", blk)
    if not m: continue
    (new if "COL_PAD" in blk.split("\n")[0] else base).append(float(m.group(1)))
ok = base and new and (sum(new)/len(new)) < 0.97*(sum(base)/len(base))
print(1 if ok else 0)
PY
)
echo "adopt=$GO"
if [ "$GO" = "1" ]; then
  (time TDMPC2_GEMM_COL_PAD=1 timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider) > gpurun_out/r3s_pytest_gpu_col_pad.log 2>&1
  grep -E "passed|failed|^FAILED" gpurun_out/r3s_pytest_gpu_col_pad.log | tail -5
fi
