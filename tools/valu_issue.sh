#!/bin/bash
# real and synthetic VALU streams at 1..8 waves per SIMD under ONE instrument (rocprofv3 --pmc),
# with the occupancy claim proven from HW_ID / s_memrealtime inside the same binary.
#   bash tools/valu_issue.sh [filter]  ->  gpurun_out/r03_valu_issue_bisect.txt   (copy to profiles/)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r03_valu
mkdir -p $OUT
TXT=gpurun_out/r03_valu_issue_bisect.txt
FILTER="${1:-}"
{
echo "==== 1. tools/valu_issue (plain run: HIP-event wall time, in-kernel clocks, occupancy proof) ===="
timeout -k 10 300 tools/valu_issue "$FILTER"
} > $TXT 2>&1 || { echo "plain run failed" >> $TXT; exit 1; }
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
timeout -k 10 420 rocprofv3 --pmc $P1 --output-format csv -d $OUT/pmc1 -o pmc -- tools/valu_issue "$FILTER" > $OUT/pmc1.log 2>&1 \
    || { echo "pmc pass failed" >> $TXT; tail -20 $OUT/pmc1.log >> $TXT; exit 1; }
{
echo
echo "==== 2. the same binary under rocprofv3 --pmc $P1 ===="
python3 tools/summarize_valu_issue.py $OUT/pmc1
} >> $TXT 2>&1
echo done
