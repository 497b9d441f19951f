#!/bin/bash
# does a second wave per SIMD keep its VALU issue rate for the two-lane sponge kernel, and is
# the 23 KB unrolled body (instruction supply) what stops it?  Runs sponge_kernel_k2<17,0,BODY> (BODY 0 = unrolled,
# literal constants; BODY 1 = rolled two-round body, ~2 KB) at 1 wave per SIMD (B = 32768), 1.5 (49152) and 2 (65536)
# on 1 MiB messages, plain timings first, then two separate --pmc passes per point (never mixed with traces).
#   bash tools/second_issue_slot.sh   ->  gpurun_out/r02_second_issue_slot.txt (copy to profiles/)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r02_slot
mkdir -p $OUT
TXT=gpurun_out/r02_second_issue_slot.txt
{
echo "# two-lane sponge kernel, SHA3-256, 1 MiB messages; DBG=0 unrolled body, DBG=8 rolled two-round body"
echo "# lanes=2 forces sponge_kernel_k2, lanes=3 the rotating one-/two-lane schedule, lanes=1 the one-lane kernel"
for DBG in 0 8; do
  echo "## DBG=$DBG"
  DBG=$DBG python3 tools/sweep_sha3.py 16384x1048576x2,32768x1048576x2,49152x1048576x2,65536x1048576x2,131072x1048576x2 2>/dev/null
done
echo "## reference (DBG=0): rotating schedule and one-lane kernel at 49152"
DBG=0 python3 tools/sweep_sha3.py 49152x1048576x31 2>/dev/null
} > $TXT 2>&1
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
P2="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU"
for DBG in 0 8; do
  for B in 32768 49152 65536; do
    i=1
    for C in "$P1" "$P2"; do
      export DBG
      timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_d${DBG}_b${B}_p$i -o pmc -- \
          python3 tools/sweep_sha3.py ${B}x1048576x2 > $OUT/pmc_d${DBG}_b${B}_p$i.log 2>&1 || echo "pmc pass d$DBG b$B p$i failed" >> $TXT
      i=$((i+1))
    done
  done
done
python3 tools/summarize_slot.py $OUT >> $TXT
echo done
