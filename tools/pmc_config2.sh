# BASELINE config 2 under separate rocprofv3 --pmc passes + a kernel trace (LANES=0x100000: the rate-block stores of r03)
export TMPDIR=/tmp
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "WRITE_SIZE" "FETCH_SIZE"; do
n=$(echo $C | cut -d" " -f1)
timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_c2_$n -o pmc -- python3 tools/pmc_config2.py > gpurun_out/pmc_c2_$n.log 2>&1 || { echo pass $n failed; tail -5 gpurun_out/pmc_c2_$n.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_c2_$n/**/*counter_collection.csv", recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "sponge_" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        acc[(r["Kernel_Name"].split("(")[0], "_dur_us")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k,v in sorted(acc.items()): print(k[0], k[1], "last dispatch:", v[-1], "dispatches:", len(v))
PY
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2 -o c2 -- python3 tools/pmc_config2.py > gpurun_out/prof_c2.log 2>&1; grep sponge gpurun_out/prof_c2/c2_kernel_stats.csv | cut -c1-160
