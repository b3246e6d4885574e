#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r3h}
(timeout 1500 python -m pytest tests -m gpu -q -x) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
(timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary) > gpurun_out/bench_${TAG}.log 2>&1
echo "[bench] $(tail -1 gpurun_out/bench_${TAG}.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step %.2f  geo avg %.2f ms x %d  frac %.3f share %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], d["roofline"]["frac"], d["roofline"]["kernel_time_share"]))' 2>&1 | tail -1)"
MODE=3 bash scripts/gpu_soak_mode2.sh ${TAG}_mode3 4000000 200
MODE=2 bash scripts/gpu_soak_mode2.sh ${TAG}_mode2 4000000 60
BENCH_ARGS="" bash scripts/gpu_pmc.sh $TAG | tail -3
(timeout 900 python scripts/render_orbit.py --frames 200) > gpurun_out/orbit_$TAG.txt 2>&1; tail -2 gpurun_out/orbit_$TAG.txt
