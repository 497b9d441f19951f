#!/usr/bin/env python3
"""Table of the --pmc passes of tools/second_issue_slot.sh: one row per (body form, batch), last dispatch of
sponge_kernel_k2.  Cycles are SQ cycles summed over the chip; per-SIMD figures divide by 1024."""
import csv
import glob
import os
import sys

root = sys.argv[1]
rows = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_d*_b*_p*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)
    dbg, b, _ = name[4:].split("_")
    key = (int(dbg[1:]), int(b[1:]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "sponge_kernel_k2" not in r["Kernel_Name"]:
                continue
            e = rows.setdefault(key, {})
            e[r["Counter_Name"]] = float(r["Counter_Value"])  # last dispatch wins
            e["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            e["_vgpr"] = int(r["VGPR_Count"])
print("\n## PMC (last dispatch of sponge_kernel_k2<17,0,BODY>; body 0 = unrolled, 8 = rolled two-round)")
hdr = ["body", "B", "waves", "ms", "GHz", "VALU/wave", "valu_busy", "wait_inst/wave_cyc", "wait_any/wave_cyc",
       "icache_req", "icache_hit%", "icache_miss", "ifetch", "salu", "smem"]
print(" ".join("%12s" % h for h in hdr))
for (dbg, b), c in sorted(rows.items()):
    g = c.get
    waves = g("SQ_WAVES", 0)
    ghz = g("GRBM_GUI_ACTIVE", 0) / 8 / c["_ns"] if c.get("_ns") else 0
    vals = [dbg, b, int(waves), "%.2f" % (c["_ns"] / 1e6), "%.2f" % ghz,
            "%.0f" % (g("SQ_INSTS_VALU", 0) / waves if waves else 0),
            "%.3f" % (4 * g("SQ_ACTIVE_INST_VALU", 0) / (1024 * g("GRBM_GUI_ACTIVE", 1) / 8)),
            "%.3f" % (g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES", 1)),
            "%.3f" % (g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES", 1)),
            "%.3g" % g("SQC_ICACHE_REQ", float("nan")),
            "%.2f" % (100 * g("SQC_ICACHE_HITS", 0) / g("SQC_ICACHE_REQ", 1) if g("SQC_ICACHE_REQ") else float("nan")),
            "%.3g" % g("SQC_ICACHE_MISSES", float("nan")), "%.3g" % g("SQ_IFETCH", float("nan")),
            "%.3g" % g("SQ_INSTS_SALU", float("nan")), "%.3g" % g("SQ_INSTS_SMEM", float("nan"))]
    print(" ".join("%12s" % v for v in vals))
