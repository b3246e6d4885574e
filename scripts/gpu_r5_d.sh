#!/bin/bash
# Round 5: GPU suite on the current build + frame-time A/B against the round-4 build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 bash scripts/gpu_ab_libs.sh exp_libs/r4.so product > gpurun_out/ab_d.txt 2>&1
tail -n 4 gpurun_out/pytest_gpu.log; cat gpurun_out/ab_d.txt
