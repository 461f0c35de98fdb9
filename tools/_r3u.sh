mkdir -p gpurun_out
bash tools/gpu_env_ab.sh r3u "c3 30 8;c4 8 3" "A=0" "TDMPC2_GEMM_XCD_ROWS=0" > /dev/null; cat gpurun_out/r3u_ab.txt
bash tools/gpu_evidence.sh r3zz "tests bench stats"
