#!/bin/bash
# rocprofv3 kernel table of the training iteration (field part).  Usage: gpu_train_prof.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-x}; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
(timeout 600 python $R/scripts/bench_train.py) > $R/gpurun_out/train_$TAG.log 2>&1; tail -1 $R/gpurun_out/train_$TAG.log
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_train -- python $R/scripts/bench_train.py) > $R/gpurun_out/rocprof_train_$TAG.log 2>&1; echo "rocprof rc=$?"
python $R/scripts/rocprof_summary.py $R/gpurun_out/prof/${TAG}_train_results.db $R/gpurun_out/train_kernels_$TAG.md | head -16
