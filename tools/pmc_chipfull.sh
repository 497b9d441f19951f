#!/bin/bash
# r03: counters of the chip-full sponge kernels after the blocked / prioritised round: clock, cycles per VALU
# instruction per SIMD, issue stalls, memory waits.  bash tools/pmc_chipfull.sh -> gpurun_out/r03_chipfull_pmc.txt
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r03_chipfull_pmc
mkdir -p $OUT
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
for CFG in 262144x524288x0 2097152x1024x0 131072x1048576x0; do
  timeout -k 10 300 rocprofv3 --pmc $P1 --output-format csv -d $OUT/$CFG -o pmc -- python3 tools/sweep_sha3.py $CFG > $OUT/$CFG.log 2>&1 || { echo "pass $CFG failed"; tail -5 $OUT/$CFG.log; exit 1; }
done
python3 - <<'PY' > gpurun_out/r03_chipfull_pmc.txt
import csv, glob, os
print("%-20s %-52s %5s %8s %5s %10s %8s %8s %6s %9s %8s" % ("batch", "kernel", "vgpr", "ms", "GHz", "VALU/wave", "cyc/inst", "ns/inst", "busy", "wait_inst", "wait_any"))
for d in sorted(glob.glob("gpurun_out/r03_chipfull_pmc/*x*/")):
    rows = {}
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sponge_" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0]
            did = int(r["Dispatch_Id"])
            e = rows.setdefault(k, {"_id": did})
            if did > e["_id"]:
                e.clear(); e["_id"] = did
            if did == e["_id"]:
                e[r["Counter_Name"]] = float(r["Counter_Value"]); e["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); e["_vgpr"] = int(r["VGPR_Count"])
    for k, c in rows.items():
        g = c.get; gui = g("GRBM_GUI_ACTIVE", 0) / 8; insts = g("SQ_INSTS_VALU", 1); waves = g("SQ_WAVES", 1)
        print("%-20s %-52s %5d %8.3f %5.2f %10.0f %8.3f %8.3f %6.3f %9.3f %8.3f" % (os.path.basename(d.rstrip("/")), k[:52], c["_vgpr"], c["_ns"] / 1e6, gui / c["_ns"], insts / waves,
              gui * 1024 / insts, c["_ns"] * 1024 / insts, 4 * g("SQ_ACTIVE_INST_VALU", 0) / (1024 * gui), g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES", 1), g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES", 1)))
PY
cat gpurun_out/r03_chipfull_pmc.txt
