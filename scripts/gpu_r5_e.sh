#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 bash scripts/gpu_ab_env.sh KPN_NO_FUSE_H3=1 base KPN_FUSE_H3W=1 KPN_ROW_SCRATCH_MIB=1280 > gpurun_out/ab_e.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_soak.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_e.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_e.log
cat gpurun_out/ab_e.txt; tail -n 3 gpurun_out/pytest_e.log
