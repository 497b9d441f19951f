#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass) into one JSON, per kernel, last dispatch.
usage: tools/summarize_pmc.py out.json dir1 dir2 ...   (each dir holds pmc_counter_collection.csv)
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads 1/2 of
a wide coalesced stream, so the corrected read side is 2 x FETCH_SIZE x 1024 (an upper bound for the 8-byte-per-lane
loads of the sponge kernels, whose width the guide calls uncalibrated; the raw figure is kept beside it)."""
import collections
import csv
import json
import os
import sys

out, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(dict)
for d in dirs:
    with open(os.path.join(d, "pmc_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] = float(r["Counter_Value"])  # last dispatch wins
            agg[k]["_dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            agg[k]["_grid"] = int(r["Grid_Size"])
            agg[k]["_vgpr"] = int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])
            # (only the passes over bench.py itself: the extra kernels of refresh_profiles.sh run on other batches)
            if os.environ.get("CAPY_PMC_ITEMS") and "sponge_" in k and not d.rstrip("/").endswith(("_short", "_wide")):
                agg[k]["_items"] = int(os.environ["CAPY_PMC_ITEMS"])  # batch size behind the grid (bench.py matches on it)
res = {"_meta": {"kernel_source_digest": os.environ.get("CAPY_PMC_DIGEST", ""),
                 "msg_stride": int(os.environ.get("CAPY_PMC_STRIDE", "0")), "items": int(os.environ.get("CAPY_PMC_ITEMS", "0"))}}
for k, c in agg.items():
    e = dict(c)
    if "FETCH_SIZE" in c:
        e["hbm_read_bytes_raw"] = c["FETCH_SIZE"] * 1024
        e["hbm_read_bytes_corrected_x2"] = c["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in c:
        e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    if "GRBM_GUI_ACTIVE" in c and c.get("_dur_ns"):
        e["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / c["_dur_ns"]
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        e["wait_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
        e["active_inst_any_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
        e["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # SQ_ACTIVE_INST_VALU counts quad-cycles; 1024 SIMDs x (GRBM_GUI_ACTIVE / 8) cycles are available
        e["valu_busy_frac_of_all_simd_cycles"] = 4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    res[k] = e
with open(out, "w") as f:
    json.dump(res, f, indent=1, sort_keys=True)
print("wrote", out, "kernels:", ", ".join(sorted(k for k in res if k != "_meta")))
