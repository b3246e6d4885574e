#!/bin/bash
# rocprofv3 kernel table of the bench frame: gpu_rocprof_bench.sh TAG [bench.py flags / NAME=VALUE environment settings first]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=$1; shift
ENVS=""; while [[ "$1" == *=* ]]; do ENVS="$ENVS $1"; shift; done
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
(env $ENVS timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary "$@") > $R/gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $R; DB=$(find gpurun_out/prof -name "${TAG}_bench*.db" | head -1); python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_bench_kernel_stats.md > /dev/null 2>&1; head -24 gpurun_out/${TAG}_bench_kernel_stats.md | cut -c1-160; rm -rf gpurun_out/prof
