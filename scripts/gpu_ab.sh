#!/bin/bash
# A/B of environment knobs on the bench workload.  Usage: gpu_ab.sh <tag> "ENV1=.. ENV2=.." "ENVa=.." ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=$1; shift
i=0
for envs in "$@"; do
  (env $envs timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_ARGS) > gpurun_out/ab_${TAG}_$i.log 2>&1
  echo "[$envs] $(tail -1 gpurun_out/ab_${TAG}_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step %.2f  geo avg %.2f ms x %d (+%d surplus)  frac %.3f  ws %.2f GB" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], d["roofline"].get("surplus_launches",0), d["roofline"]["frac"], d["config"]["render_workspace_bytes"]/1e9))' 2>&1 | tail -1)"
  i=$((i+1))
done
