#!/bin/bash
# GPU-box check: smoke, gpu tests, bench (+ optional rocprof).  Usage: gpu_check.sh <tag> [prof]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-x}
(timeout 600 python __graft_entry__.py --smoke) > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_$TAG.log
(timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
(timeout 600 python bench.py --steps 5 --warmup 1) > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_$TAG.log | cut -c1-1600
if [ "$2" == "prof" ]; then
  R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
  (timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary) > $R/gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
  # the N>1 flow on this 1-GPU box: bench.py starts 2 ranks itself, both on cuda:0, gloo gather to rank 0
  (timeout 600 python $R/bench.py --gpus 2 --dist-backend gloo --steps 2 --warmup 1) > $R/gpurun_out/bench_2rank_gloo_$TAG.log 2>&1; echo "2-rank rc=$?"; tail -1 $R/gpurun_out/bench_2rank_gloo_$TAG.log | cut -c1-400
fi
