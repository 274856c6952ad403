#!/bin/bash
# Collects the SQ, FETCH_SIZE and WRITE_SIZE counter passes of a short bench.py run (one pass per rocprofv3
# run) under gpurun_out/$1.  usage (on the GPU box, from the repo root): bash tools/pmc_run.sh NAME [bench args]
set -e
NAME=$1; shift
R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/write.log 2>&1
python3 profiles/summarize_pmc.py $O/sq/sq_counter_collection.csv $O/fetch/fetch_counter_collection.csv $O/write/write_counter_collection.csv > $O/pmc.csv
