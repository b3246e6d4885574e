#!/bin/bash
# Round 5: what the rows kernel would gain without its weight loads in the L1 path (timing experiments, wrong results)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 bash scripts/gpu_ab_libs.sh product exp_libs/now.so exp_libs/sametap.so exp_libs/now_sametap.so > gpurun_out/ab_now.txt 2>&1
KPN_TEST_LIB=exp_libs/hlfirst.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q --deselect tests/test_gpu_parity.py::test_native_library_is_loaded > gpurun_out/pytest_hlfirst2.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_hlfirst2.log
cat gpurun_out/ab_now.txt; tail -n 3 gpurun_out/pytest_hlfirst2.log
