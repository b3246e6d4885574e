#!/bin/bash
# round-3 evidence run: full suite, driver-style bench line, rocprofv3 kernel table of the same command, phase cycles, orbit with
# and without encoders, encoder stand-in timing, training profile, 200-scene parity sweep in rows modes 3 and 0
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r3i}
(timeout 600 python __graft_entry__.py --smoke) > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_$TAG.log
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
(timeout 1200 python bench.py) > gpurun_out/bench_$TAG.json 2>gpurun_out/bench_$TAG.err; echo "bench rc=$?"; tail -1 gpurun_out/bench_$TAG.json | cut -c1-400
for m in 3 2; do timeout 300 python scripts/h2_timing.py exp_libs/h2t.so $m 2>&1 | tail -1; done | tee gpurun_out/phase_cycles_$TAG.txt
timeout 300 python scripts/fuse_timing.py exp_libs/ft.so 2>&1 | tail -1 | tee -a gpurun_out/phase_cycles_$TAG.txt
(timeout 900 python scripts/render_orbit.py --frames 200; timeout 900 python scripts/render_orbit.py --frames 200 --with-encoders; timeout 300 python scripts/encoder_standin.py) > gpurun_out/orbit_$TAG.txt 2>&1; grep -v amdgpu gpurun_out/orbit_$TAG.txt | tail -4
(timeout 900 python scripts/fuzz_parity.py 200 7) > gpurun_out/fuzz_$TAG.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/fuzz_$TAG.log | cut -c1-600; cp gpurun_out/fuzz_parity.json gpurun_out/fuzz_parity_mode3_$TAG.json
(KPN_GEO_ROWS_MODE=0 KPN_FUSE_MODE=0 timeout 900 python scripts/fuzz_parity.py 200 7) > gpurun_out/fuzz0_$TAG.log 2>&1; echo "fuzz fp32 rc=$?"; tail -1 gpurun_out/fuzz0_$TAG.log | cut -c1-600; cp gpurun_out/fuzz_parity.json gpurun_out/fuzz_parity_fp32_$TAG.json
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary) > $R/gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $R; DB=$(find gpurun_out/prof -name "${TAG}_bench*.db" | head -1); python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_kernel_stats.md > /dev/null 2>&1; head -12 gpurun_out/${TAG}_kernel_stats.md; rm -rf gpurun_out/prof
bash scripts/gpu_train_prof.sh $TAG 2>&1 | tail -14
