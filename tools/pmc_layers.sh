#!/bin/bash
# Counter passes (one rocprofv3 run each) over tools/layer_bench.py for the given layers, under gpurun_out/$1.
# usage (GPU box, repo root): bash tools/pmc_layers.sh NAME conv2 conv3 ...     (MPC_X3_PATCH etc. are inherited)
NAME=$1; shift
O=$PWD/gpurun_out/$NAME; mkdir -p $O
export TMPDIR=/tmp
run() {   # tag counters...
  local tag=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O/$tag -o $tag -- python3 tools/layer_bench.py --reps 3 $LAYERS > $O/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $O/$tag.log; return 1; }
}
LAYERS="$*"
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE && \
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE && \
python3 profiles/summarize_pmc.py $O/sq/sq_counter_collection.csv $O/sq2/sq2_counter_collection.csv $O/fetch/fetch_counter_collection.csv $O/write/write_counter_collection.csv > $O/pmc.csv && \
python3 tools/pmc_report.py $O/pmc.csv
