#!/bin/bash
# round-3 closing evidence run: smoke, full GPU suite, driver-style bench line + rocprofv3 kernel table of the same command,
# training bench (field part and through the drop-in) + kernel table, k_geo_rows_bwd phase cycles, orbit with and without encoders
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r3z}
(timeout 600 python __graft_entry__.py --smoke) > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_$TAG.log
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
(timeout 1200 python bench.py) > gpurun_out/bench_$TAG.json 2>gpurun_out/bench_$TAG.err; echo "bench rc=$?"; tail -1 gpurun_out/bench_$TAG.json | cut -c1-400
python scripts/bwd_timing.py exp_libs/bwdtime.so 2>&1 | tail -14 | tee gpurun_out/bwd_phase_cycles_$TAG.txt
(timeout 600 python scripts/bench_dropin_train.py) 2>&1 | tail -2 | tee gpurun_out/dropin_train_$TAG.txt
(timeout 900 python scripts/render_orbit.py --frames 200; timeout 900 python scripts/render_orbit.py --frames 200 --with-encoders) > gpurun_out/orbit_$TAG.txt 2>&1; grep -v amdgpu gpurun_out/orbit_$TAG.txt | tail -2
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary) > $R/gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $R; DB=$(find gpurun_out/prof -name "${TAG}_bench*.db" | head -1); python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_kernel_stats.md > /dev/null 2>&1; head -12 gpurun_out/${TAG}_kernel_stats.md; rm -rf gpurun_out/prof
bash scripts/gpu_train_prof.sh $TAG 2>&1 | tail -14; rm -rf gpurun_out/prof
