#!/bin/bash
# soak (run-to-run determinism of mode 1 over 1M points) of experimental library builds: gpu_soak_libs.sh RUNS lib...
cd $GRAFT_REPO_ROOT; RUNS=$1; shift
for l in "$@"; do KPN_EXPERIMENT_LIB=$l SOAK_RUNS=$RUNS timeout 280 python scripts/soak_detail.py 2>&1 | tail -1; done
