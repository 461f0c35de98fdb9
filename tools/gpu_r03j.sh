cd /root/repo
mkdir -p gpurun_out
out=gpurun_out/r03j_graph_flaky.txt; : > $out
for k in "act_matches or graph" "save_load or graph" "plan_batch or graph" "errors or graph" ; do
  echo "== $k" >> $out
  timeout 300 python -m pytest tests/test_gpu_boundary.py -q -m gpu -k "$k" -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" >> $out
done
echo "== adversarial + graph" >> $out
timeout 300 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_boundary.py -q -m gpu -k "adversarial or gamma or graph" -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" >> $out
cat $out
