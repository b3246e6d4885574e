#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r3c}
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu_$TAG.log
(timeout 900 python scripts/fuzz_parity.py 200 7) > gpurun_out/fuzz_$TAG.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/fuzz_$TAG.log; cp gpurun_out/fuzz_parity.json gpurun_out/fuzz_parity_mode3_$TAG.json
(KPN_GEO_ROWS_MODE=0 timeout 900 python scripts/fuzz_parity.py 200 7) > gpurun_out/fuzz0_$TAG.log 2>&1; echo "fuzz mode0 rc=$?"; tail -1 gpurun_out/fuzz0_$TAG.log; cp gpurun_out/fuzz_parity.json gpurun_out/fuzz_parity_mode0_$TAG.json
