#!/bin/bash
# counters for the ragged vs uniform short-message comparison (tools/bench_ragged_dev.py); separate --pmc passes
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r02_ragged
mkdir -p $OUT
python3 tools/bench_ragged_dev.py > $OUT/timing.txt 2>&1
for MODE in ragged uniform; do
  i=1
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "FETCH_SIZE"; do
    export MODE REPS=1
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${MODE}_p$i -o pmc -- python3 tools/bench_ragged_dev.py > $OUT/pmc_${MODE}_p$i.log 2>&1 || echo "pass $MODE $i failed"
    i=$((i+1))
  done
done
python3 - <<PY
import csv, glob, os
for mode in ("ragged", "uniform"):
    agg = {}
    for d in sorted(glob.glob("$OUT/pmc_%s_p*" % mode)):
        if not os.path.isdir(d): continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if "sponge" not in k: continue
                e = agg.setdefault(k, {})
                e[r["Counter_Name"]] = float(r["Counter_Value"])
                e["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                e["_vgpr"] = r["VGPR_Count"]
    for k, e in agg.items():
        w = e.get("SQ_WAVES", 1)
        print(mode, k, "ms %.3f" % (e["_ns"] / 1e6), "waves %d" % w, "vgpr", e["_vgpr"],
              "VALU/wave %.0f" % (e.get("SQ_INSTS_VALU", 0) / w), "SALU/wave %.0f" % (e.get("SQ_INSTS_SALU", 0) / w),
              "LDS/wave %.0f" % (e.get("SQ_INSTS_LDS", 0) / w), "VMEM_RD/wave %.0f" % (e.get("SQ_INSTS_VMEM_RD", 0) / w),
              "wait_any/wave_cyc %.3f" % (e.get("SQ_WAIT_ANY", 0) / e.get("SQ_WAVE_CYCLES", 1)),
              "wait_inst/wave_cyc %.3f" % (e.get("SQ_WAIT_INST_ANY", 0) / e.get("SQ_WAVE_CYCLES", 1)),
              "valu_busy %.3f" % (4 * e.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * e.get("GRBM_GUI_ACTIVE", 1) / 8)),
              "GHz %.2f" % (e.get("GRBM_GUI_ACTIVE", 0) / 8 / e["_ns"]), "fetch_MB %.0f" % (e.get("FETCH_SIZE", 0) * 2 / 1024))
PY
cat $OUT/timing.txt
