#!/bin/bash
# round 3, run q: backward tests, training bench (product, 256/384 weight-gradient workers), k_geo_rows_bwd phase cycles, kernel table
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -x -q -k "backward or grad or train" 2>&1 | tail -3
python scripts/bench_train.py 2>&1 | tail -1
KPN_EXPERIMENT_LIB=exp_libs/gw256.so python scripts/bench_train.py 2>&1 | tail -1
KPN_EXPERIMENT_LIB=exp_libs/gw384.so python scripts/bench_train.py 2>&1 | tail -1
python scripts/bwd_timing.py exp_libs/bwdtime.so 2>&1 | tail -16 | tee gpurun_out/bwd_phase_cycles_r3q.txt
bash scripts/gpu_train_prof.sh r3q 2>&1 | tail -14
