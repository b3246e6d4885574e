#!/bin/bash
# The GPU suite in the default mode and with mode 2 as the library default, then the bench in both.  Usage: gpu_suite_modes.sh <tag>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-x}
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu_${TAG}_default.log 2>&1; echo "default-mode rc=$?"; tail -2 gpurun_out/pytest_gpu_${TAG}_default.log
(KPN_GEO_ROWS_MODE=0 timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu_${TAG}_mode0.log 2>&1; echo "mode0 rc=$?"; tail -2 gpurun_out/pytest_gpu_${TAG}_mode0.log
for m in 0 2; do
  (timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --geo-rows-mode $m) > gpurun_out/bench_${TAG}_mode$m.log 2>&1
  tail -1 gpurun_out/bench_${TAG}_mode$m.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step %.2f  geo avg %.2f ms x %d  frac %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], d["roofline"]["frac"]))'
done
