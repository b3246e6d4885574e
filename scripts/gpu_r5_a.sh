#!/bin/bash
# Round 5, first GPU run: the WAW reproducer, the GPU suite on the new (compiler-selected) operand splits, both product orders of
# the per-point kernel through the soak gate, A/B of the frame time against the round-4 build.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 120 exp_libs/repro_waw > gpurun_out/repro_waw.txt 2>&1; echo "rc=$?" >> gpurun_out/repro_waw.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log
KPN_TEST_LIB=exp_libs/hlfirst.so timeout 600 python -m pytest tests/test_gpu_soak.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_hlfirst.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_hlfirst.log
timeout 300 python scripts/lib_vs_lib.py exp_libs/hlfirst.so > gpurun_out/hlfirst_frame.txt 2>&1
timeout 900 bash scripts/gpu_ab_libs.sh exp_libs/r4.so product exp_libs/hlfirst.so > gpurun_out/ab_split.txt 2>&1
tail -3 gpurun_out/pytest_gpu.log gpurun_out/pytest_hlfirst.log; cat gpurun_out/repro_waw.txt gpurun_out/hlfirst_frame.txt gpurun_out/ab_split.txt
