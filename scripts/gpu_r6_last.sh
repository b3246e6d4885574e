cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=r06_z
(timeout 600 python __graft_entry__.py --smoke) > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_$TAG.log
(timeout 2400 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu_$TAG.log | cut -c1-200
(timeout 1500 python bench.py --steps 20 --warmup 5) > gpurun_out/bench_$TAG.json 2>gpurun_out/bench_$TAG.err; echo "bench rc=$?"
(timeout 600 python scripts/rccl_one_rank.py 5) 2>/dev/null | grep '^{' | tail -1 | tee gpurun_out/rccl_one_rank_$TAG.json | cut -c1-400
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary) > $R/gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $R; DB=$(find gpurun_out/prof -name "${TAG}_bench*.db" | head -1); python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_bench_kernel_stats.md > /dev/null 2>&1; head -8 gpurun_out/${TAG}_bench_kernel_stats.md | cut -c1-150; rm -rf gpurun_out/prof
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_train -- python $R/scripts/bench_train.py) > $R/gpurun_out/rocprof_train_$TAG.log 2>&1; echo "rocprof train rc=$?"
cd $R; DB=$(find gpurun_out/prof -name "${TAG}_train*.db" | head -1); python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_train_kernel_stats.md > /dev/null 2>&1; rm -rf gpurun_out/prof
