#!/bin/bash
# Round 6: the two parity sweeps and the configs[4] frame with the conditional re-check of every widened ray
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r06_z}
(timeout 1200 python scripts/fuzz_parity.py 200) > gpurun_out/fuzz_$TAG.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/fuzz_$TAG.log | cut -c1-700; cp gpurun_out/fuzz_parity.json gpurun_out/fuzz_parity_200_scenes_default_$TAG.json
(timeout 1800 python scripts/fuzz_parity.py 400 7) > gpurun_out/fuzz400_$TAG.log 2>&1; echo "fuzz400 rc=$?"; tail -1 gpurun_out/fuzz400_$TAG.log | cut -c1-700; cp gpurun_out/fuzz_parity.json gpurun_out/fuzz_400scenes_seed7_$TAG.json
(timeout 900 python scripts/bench_configs4.py) > gpurun_out/configs4_$TAG.log 2>&1; echo "configs4 rc=$?"; tail -1 gpurun_out/configs4_$TAG.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(d['parity'])[:1500])"
