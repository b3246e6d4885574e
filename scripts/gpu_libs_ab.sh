#!/bin/bash
# A/B of whole-library builds (exp_libs/*.so) on the bench frame: each is copied over the product library in turn.
# Usage: gpu_libs_ab.sh <tag> "<lib> <mode> [bench args]" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=$1; shift
cp keypointnerf_amd/_lib/libkpnerf_hip.so /tmp/orig.so
i=0
for spec in "$@"; do
  set -- $spec; lib=$1; mode=$2; shift; shift
  cp exp_libs/$lib.so keypointnerf_amd/_lib/libkpnerf_hip.so
  (timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --geo-rows-mode $mode "$@") > gpurun_out/lab_${TAG}_$i.log 2>&1
  echo "[$lib mode $mode $@] $(tail -1 gpurun_out/lab_${TAG}_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step %.2f  geo avg %.2f ms x %d  frac %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], d["roofline"]["frac"]))' 2>&1 | tail -1)"
  if [ "$mode" != "0" ]; then python scripts/soak_mode2.py --mode $mode --repeats 10 | cut -c1-400; fi
  i=$((i+1))
done
cp /tmp/orig.so keypointnerf_amd/_lib/libkpnerf_hip.so
