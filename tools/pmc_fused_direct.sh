export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_fused_$C -o pmc -- python3 tools/pmc_fused_direct.py > gpurun_out/pmc_fused_$C.log 2>&1 || { echo pass $C failed; tail -5 gpurun_out/pmc_fused_$C.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/pmc_fused_$C/**/*counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "fused" in r["Kernel_Name"]:
        print("$C", r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"])
PY
done
