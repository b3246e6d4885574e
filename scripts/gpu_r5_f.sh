#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_soak.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_f.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_f.log
tail -n 3 gpurun_out/pytest_f.log
timeout 600 bash scripts/gpu_ab_libs.sh exp_libs/noring.so product > gpurun_out/ab_f.txt 2>&1
cat gpurun_out/ab_f.txt
