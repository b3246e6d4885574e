#!/bin/bash
# round-3 A/B: smoke + gpu tests on the product build, then whole-library variants on the bench frame, then phase cycles
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r3a}
cp keypointnerf_amd/_lib/libkpnerf_hip.so exp_libs/product.so
(timeout 600 python __graft_entry__.py --smoke) > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_$TAG.log
(timeout 1200 python -m pytest tests -m gpu -q -x) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu_$TAG.log
shift
bash scripts/gpu_libs_ab.sh $TAG "$@"
for t in exp_libs/*t.so; do [ -f $t ] && for m in 3 2; do timeout 300 python scripts/h2_timing.py $t $m 2>&1 | tail -1; done; done
