#!/bin/bash
# r05: counters of the one-lane-per-sponge fused encrypt / decrypt kernels (csrc/sponge_fused1.h): time, clock, VALU per wave,
# busy / wait fractions, LDS conflicts, HBM bytes read and written against the message bytes.  Separate --pmc passes
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; FETCH_SIZE x 2 on gfx950).
# usage: bash tools/pmc_fused1.sh  -> gpurun_out/r05_fused1_pmc.txt
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r05_fused1_pmc
mkdir -p $OUT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
P2="FETCH_SIZE"
P3="WRITE_SIZE"
P4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
for CFG in ${CFGS:-65536x1048576 131072x1048576 49152x4194304}; do
  N=${CFG%x*}; L=${CFG#*x}
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    NS=$N LEN=$L REPS=1 timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $OUT/$CFG/p$i -o pmc -- python3 tools/sweep_fused1.py > $OUT/$CFG.p$i.log 2>&1 || { echo "pass $CFG p$i failed"; tail -5 $OUT/$CFG.p$i.log; exit 1; }
  done
done
python3 - <<'PY' > gpurun_out/r05_fused1_pmc.txt
import csv, glob, os, collections
print("# rocprofv3 --pmc, one dispatch per row (encrypt and decrypt instances); message bytes = n x len; HBM read = 2 x FETCH_SIZE KiB (gfx950), write = WRITE_SIZE KiB")
print("%-18s %-58s %5s %8s %5s %10s %6s %6s %6s %7s %7s %8s" % ("batch", "kernel", "vgpr", "ms", "GHz", "VALU/wave", "busy", "w_inst", "w_any", "rd/msg", "wr/msg", "ldsconf"))
for d in sorted(glob.glob("gpurun_out/r05_fused1_pmc/*x*/")):
    cfg = os.path.basename(d.rstrip("/")); n, ln = [int(x) for x in cfg.split("x")]
    rows = collections.defaultdict(dict)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fused1" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void capy::", "")
            e = rows[k]
            key = r["Counter_Name"]
            e.setdefault(key, 0.0)
            e[key] += float(r["Counter_Value"])          # sum over the dispatches of a schedule (phases / slices)
            e.setdefault("_ns_" + key, 0)
            e["_ns_" + key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            e["_vgpr"] = int(r["VGPR_Count"])
    for k, c in sorted(rows.items()):
        g = c.get
        ns = g("_ns_GRBM_GUI_ACTIVE", 0) / max(1, sum(1 for x in c if x == "GRBM_GUI_ACTIVE")) or g("_ns_FETCH_SIZE", 1)
        ns = g("_ns_SQ_WAVES", 0) or ns
        gui = g("GRBM_GUI_ACTIVE", 0) / 8 / 2   # collected in two passes (P1, P4): halve
        insts = g("SQ_INSTS_VALU", 1); waves = g("SQ_WAVES", 1)
        msg = float(n) * ln
        print("%-18s %-58s %5d %8.3f %5.2f %10.0f %6.3f %6.3f %6.3f %7.4f %7.4f %8.4f" % (cfg, k[:58], c["_vgpr"], ns / 1e6, (g("GRBM_GUI_ACTIVE", 0) / 16) / max(1, ns), insts / waves,
              4 * g("SQ_ACTIVE_INST_VALU", 0) / (1024 * max(1.0, gui)), g("SQ_WAIT_INST_ANY", 0) / max(1.0, g("SQ_WAVE_CYCLES", 1)), g("SQ_WAIT_ANY", 0) / max(1.0, g("SQ_WAVE_CYCLES", 1)),
              g("FETCH_SIZE", 0) * 2048 / msg, g("WRITE_SIZE", 0) * 1024 / msg, g("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, g("SQ_LDS_IDX_ACTIVE", 1))))
PY
cat gpurun_out/r05_fused1_pmc.txt
