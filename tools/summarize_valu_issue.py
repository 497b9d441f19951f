#!/usr/bin/env python3
"""Table of the --pmc pass of tools/valu_issue.sh: one row per (stream, data, waves per SIMD), last dispatch.
Everything in the table comes from the counters and the dispatch timestamps of the SAME pass:
  GHz        = GRBM_GUI_ACTIVE / 8 / duration                       (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
  cyc/inst   = (GRBM_GUI_ACTIVE / 8) * 1024 SIMDs / SQ_INSTS_VALU    cycles per wave64 VALU instruction per SIMD
  ns/inst    = duration * 1024 / SQ_INSTS_VALU                       the same in time (what throughput sees)
  busy       = 4 * SQ_ACTIVE_INST_VALU / (1024 * GRBM_GUI_ACTIVE / 8)   (SQ_* cycle counters count quad-cycles)
  wait_inst  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES"""
import csv
import glob
import os
import re
import sys

REAL = {0: "k1 unrolled", 1: "k1 rolled", 2: "k2 unrolled", 3: "k2 rolled", 4: "fe_mul x2", 5: "fe_sqr x2",
        6: "k1 noalign", 7: "k1 theta-only", 8: "k1 rho-only", 9: "k1 chi-only", 10: "k1 blocked rolled", 11: "k1 paired rolled",
        12: "k1 paired unrolled"}
SYN = {0: "syn xor", 1: "syn bitop3", 2: "syn alignbit", 3: "syn dpp", 4: "syn k1mix", 5: "syn k2mix", 6: "syn mad64",
       7: "syn madmix", 8: "syn add3", 9: "syn k1mix feedback", 10: "syn lshrrev_b64", 11: "syn lshl_add_u64",
       12: "syn sub_co+subb", 13: "syn mul_lo_u32"}
rows = {}
order = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"vi_(real|syn)<(\d+), (\d+)>", r["Kernel_Name"])
        if not m:
            continue
        label = (REAL if m.group(1) == "real" else SYN)[int(m.group(2))]
        key = (label, int(m.group(3)), int(r["Grid_Size"]) // 256 // 256)
        did = int(r["Dispatch_Id"])
        e = rows.setdefault(key, {"_id": did})
        if did > e["_id"]:
            e.clear()
            e["_id"] = did
        if did == e["_id"]:
            e[r["Counter_Name"]] = float(r["Counter_Value"])
            e["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            e["_vgpr"] = int(r["VGPR_Count"])
        if key not in order:
            order.append(key)
print("%-24s %4s %4s %2s %8s %5s %11s %8s %8s %6s %9s %8s" % ("stream", "data", "vgpr", "W", "ms", "GHz", "VALU/wave",
                                                         "cyc/inst", "ns/inst", "busy", "wait_inst", "wait_any"))
for key in order:
    c = rows[key]
    g = c.get
    gui = g("GRBM_GUI_ACTIVE", 0) / 8
    insts = g("SQ_INSTS_VALU", 0)
    waves = g("SQ_WAVES", 0)
    if not insts or not gui:
        continue
    print("%-24s %4s %4d %2d %8.3f %5.2f %11.0f %8.3f %8.3f %6.3f %9.3f %8.3f" % (
        key[0], "rand" if key[1] else "zero", c["_vgpr"], key[2], c["_ns"] / 1e6, gui / c["_ns"], insts / waves if waves else 0,
        gui * 1024 / insts, c["_ns"] * 1024 / insts, 4 * g("SQ_ACTIVE_INST_VALU", 0) / (1024 * gui),
        g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES", 1), g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES", 1)))
