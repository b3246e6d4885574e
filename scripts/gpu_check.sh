#!/bin/bash
# GPU-box check: smoke, gpu tests, bench (+ optional rocprof).  Usage: gpu_check.sh <tag> [prof]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-x}
(timeout 600 python __graft_entry__.py --smoke) > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_$TAG.log
(timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
(timeout 600 python bench.py --steps 5 --warmup 1) > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_$TAG.log | cut -c1-1600
if [ "$2" == "prof" ]; then
  R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
  (timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline) > $R/gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
  (timeout 600 python $R/bench.py --steps 3 --warmup 1 --mask dense --no-cpu-baseline) > $R/gpurun_out/bench_dense_$TAG.log 2>&1; tail -1 $R/gpurun_out/bench_dense_$TAG.log | cut -c1-300
fi
