#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r3f}
(timeout 600 python __graft_entry__.py --smoke) > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_$TAG.log
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_$TAG.log
for m in 3 2 3; do
  (timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --geo-rows-mode $m) > gpurun_out/bench_${TAG}_mode$m.log 2>&1
  echo "[rows mode $m] $(tail -1 gpurun_out/bench_${TAG}_mode$m.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step %.2f  geo avg %.2f ms x %d  frac %.3f share %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], d["roofline"]["frac"], d["roofline"]["kernel_time_share"]))' 2>&1 | tail -1)"
done
for m in 3 2; do timeout 300 python scripts/h2_timing.py exp_libs/h2t.so $m 2>&1 | tail -1; done
timeout 300 python scripts/fuse_timing.py exp_libs/ft.so 2>&1 | tail -1
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary) > $R/gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $R; DB=$(find gpurun_out/prof -name "${TAG}_bench*.db" | head -1); python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_kernel_stats.md > /dev/null 2>&1; head -9 gpurun_out/${TAG}_kernel_stats.md; rm -rf gpurun_out/prof
