#!/bin/bash
# PMC counter passes for the training iteration (scripts/bench_train.py), separate runs per counter group, kernel-trace only.
R=$GRAFT_REPO_ROOT; TAG=${1:-x}; cd /tmp; export TMPDIR=/tmp
CMD="python $R/scripts/bench_train.py"
run() { name=$1; shift; (timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmct_$TAG -o $name -- $CMD) > $R/gpurun_out/pmct_${TAG}_$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SALU
run sq4 SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
cd $R; python scripts/pmc_summary.py gpurun_out/pmct_$TAG k_geo_rows_bwd k_color_bwd k_weight_grad k_fuse_bwd > gpurun_out/pmct_${TAG}_summary.txt 2>&1; rm -rf gpurun_out/pmct_$TAG
