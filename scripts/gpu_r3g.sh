#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r3g}
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_$TAG.log
python scripts/rccl_smoke.py
T0=$(date +%s); (timeout 1200 python bench.py) > gpurun_out/bench_$TAG.log 2>gpurun_out/bench_$TAG.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"; tail -1 gpurun_out/bench_$TAG.log | cut -c1-3000
(timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 3 --warmup 1 --no-secondary --no-cpu-baseline) > gpurun_out/bench_2rank_gloo_$TAG.log 2>&1; echo "2-rank rc=$?"; tail -1 gpurun_out/bench_2rank_gloo_$TAG.log | cut -c1-600
