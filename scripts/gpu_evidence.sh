#!/bin/bash
# End-of-round evidence beyond gpu_check.sh: PMC passes, BASELINE configs[4] at full size, the 200-frame orbit, fuzz sweep.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-x}
bash scripts/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1; echo "pmc rc=$?"
(timeout 900 python bench.py --res 4096 --views 10 --samples 128 --no-fine --mask dense --steps 1 --warmup 0 --no-cpu-baseline --no-secondary) > gpurun_out/bench_configs4_$TAG.log 2>&1; echo "configs4 rc=$?"; tail -1 gpurun_out/bench_configs4_$TAG.log | cut -c1-300
(timeout 600 python scripts/render_orbit.py --frames 200) > gpurun_out/orbit_$TAG.log 2>&1; echo "orbit rc=$?"; tail -1 gpurun_out/orbit_$TAG.log
(timeout 600 python scripts/fuzz_parity.py) > gpurun_out/fuzz_$TAG.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/fuzz_$TAG.log | cut -c1-400
