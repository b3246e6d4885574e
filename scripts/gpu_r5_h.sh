#!/bin/bash
# Round 5: scheduler-option variants of the kpn_api translation unit (per-point kernel, backward kernels) + training time with them,
# the 200-scene sweep in the fp32 kernels with the extended probe, the operand-range report.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 bash scripts/gpu_ab_libs.sh product exp_libs/ilp.so exp_libs/bias0.so > gpurun_out/ab_h.txt 2>&1; cat gpurun_out/ab_h.txt
for lib in "" exp_libs/ilp.so exp_libs/bias0.so; do echo "train lib=${lib:-product}"; KPN_EXPERIMENT_LIB=$lib timeout 300 python scripts/bench_train.py 2>/dev/null | tail -1; done > gpurun_out/train_h.txt 2>&1; cat gpurun_out/train_h.txt
(timeout 900 python scripts/fuzz_parity.py 200 --fp32) > gpurun_out/fuzz_fp32_h.log 2>&1; echo "fuzz fp32 rc=$?"; tail -1 gpurun_out/fuzz_fp32_h.log | cut -c1-500
(timeout 900 python scripts/range_gate.py) > gpurun_out/range_gate_h.txt 2>&1; echo "range rc=$?"; grep -c "gate: default True" gpurun_out/range_gate_h.txt; tail -2 gpurun_out/range_gate_h.txt | cut -c1-300
