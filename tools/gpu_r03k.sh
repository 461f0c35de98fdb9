cd /root/repo
run() { echo "== $*"; env "$@" timeout 300 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_boundary.py -q -m gpu -k "(gain_does_not_saturate and small) or graph" -p no:cacheprovider 2>&1 | grep -E "passed|failed|AssertionError: assert" | head -3; }
run A=1
run TDMPC2_SIDE_BLOCKING=1
run TDMPC2_SIDE_LEAK=1
run TDMPC2_SIDE_BLOCKING=1 TDMPC2_SIDE_LEAK=1
run HIP_FORCE_DEV_KERNARG=1
run GPU_MAX_HW_QUEUES=1
