#!/bin/bash
# rocprofv3 kernel trace of the bench frame in a given geo-rows mode.  Usage: gpu_prof_mode.sh <tag> <mode> [extra bench args]
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; TAG=$1; MODE=$2; shift; shift
cd /tmp; export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --geo-rows-mode $MODE "$@") > $R/gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
tail -1 $R/gpurun_out/rocprof_$TAG.log | cut -c1-300
cd $R; DB=$(find gpurun_out/prof -name "${TAG}_bench*.db" | head -1); python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_kernel_stats.md > /dev/null 2>&1; head -10 gpurun_out/${TAG}_kernel_stats.md; rm -rf gpurun_out/prof
