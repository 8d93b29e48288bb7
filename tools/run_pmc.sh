#!/bin/bash
# Three counter passes over tools/pmc_call.py (each its own run; TCC: FETCH_SIZE uses 3 of 4 slots, WRITE_SIZE 2).
set -e
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc}
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 tools/pmc_call.py > $OUT/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 tools/pmc_call.py > $OUT/write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- python3 tools/pmc_call.py > $OUT/sq.log 2>&1
python3 tools/pmc_report.py $OUT/pmc_summary.json $OUT/fetch $OUT/write $OUT/sq > $OUT/pmc_summary.txt
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/pmc_summary.txt
