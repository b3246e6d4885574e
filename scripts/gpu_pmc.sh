#!/bin/bash
# PMC counter passes for the bench workload (separate runs per counter group, kernel-trace only).
R=$GRAFT_REPO_ROOT; TAG=${1:-x}; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_ARGS"
run() { name=$1; shift; (timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$TAG -o $name -- $CMD) > $R/gpurun_out/pmc_${TAG}_$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SALU
run sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA
run sq4 SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
(cd $R; python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_ARGS 2>/dev/null | tail -1 > gpurun_out/bench_r4_pmc_rows.json)   # the rows of one frame (x 2 frames per pass)
ls $R/gpurun_out/pmc_$TAG | head
cd $R; python scripts/pmc_summary.py gpurun_out/pmc_$TAG > gpurun_out/pmc_${TAG}_summary.txt 2>&1; rm -rf gpurun_out/pmc_$TAG
