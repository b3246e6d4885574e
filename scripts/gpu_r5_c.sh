#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 bash scripts/gpu_ab_libs.sh product exp_libs/now2.so > gpurun_out/ab_now2.txt 2>&1
cat gpurun_out/ab_now2.txt
