#!/bin/bash
# round 2, call 16: phase timers of the cluster kernel (-DSPLIT_TIMING build), c2 E=1
mkdir -p gpurun_out
CLUSTER_MODES=1 CLUSTER_ENVS=1 TDMPC2_PLAN_LIB=build/ablate/lib_cl48_timing.so TDMPC2_TIMING=1 timeout 300 python tools/probes/cluster_latency.py c2 > gpurun_out/r02m_timing.txt 2>&1
cat gpurun_out/r02m_timing.txt
