#!/bin/bash
# PMC passes for the vector-memory front end (TA = address unit, TCP = L1) under the bench workload: is the rows kernel's memory
# wait a latency or an L1-throughput effect?  Separate runs per counter group, kernel-trace only.
R=$GRAFT_REPO_ROOT; TAG=${1:-x}; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_ARGS"
run() { name=$1; shift; (timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcl1_$TAG -o $name -- $CMD) > $R/gpurun_out/pmcl1_${TAG}_$name.log 2>&1; echo "$name rc=$?"; }
run ta1 TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run ta3 TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
run tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcp2 TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
run tcp3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
cd $R; python scripts/pmc_summary.py gpurun_out/pmcl1_$TAG > gpurun_out/pmcl1_${TAG}_summary.txt 2>&1; rm -rf gpurun_out/pmcl1_$TAG
grep -A 30 "k_geo_rows_f2p" gpurun_out/pmcl1_${TAG}_summary.txt | head -40
