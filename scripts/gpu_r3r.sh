#!/bin/bash
# round 3, run r: backward tests, training bench, k_geo_rows_bwd phase cycles, kernel table
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r3r}
python -m pytest tests -m gpu -x -q -k "backward or grad or train or rgba2out" 2>&1 | tail -3
python scripts/bench_train.py 2>&1 | tail -1
python scripts/bwd_timing.py exp_libs/bwdtime.so 2>&1 | tail -16 | tee gpurun_out/bwd_phase_cycles_$TAG.txt
bash scripts/gpu_train_prof.sh $TAG 2>&1 | tail -12
