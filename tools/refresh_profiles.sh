#!/bin/bash
# Regenerates the round's committed measurements on a GPU box.  Run from the repo root:
#   bash tools/refresh_profiles.sh r01
# Writes raw rocprofv3 output under gpurun_out/ and the summaries under profiles/<round>_*.
# Counter passes are separate runs (rocprofv3 refuses / mis-handles large counter sets; never mix --pmc with traces).
set -o pipefail
R=${1:-r05}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT profiles
python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err && tail -1 $OUT/bench.json > profiles/${R}_bench_line.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$R -o bench -- \
    python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs > $OUT/prof_bench.log 2>&1 \
    && cp $OUT/prof_$R/bench_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    n=$(echo $C | cut -d" " -f1)
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${R}_$n -o pmc -- \
        python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $OUT/pmc_$n.log 2>&1 || echo "PMC pass $n failed"
done
# two more kernels for the summary: the uniform-framing kernel on short messages (2^22 x 64 B) and the wave-per-item
# sha3_encrypt kernel on BASELINE config 3 as specified (128 x 5 MiB); SQ counters only
SQC="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
timeout -k 10 200 rocprofv3 --pmc $SQC --output-format csv -d $OUT/pmc_${R}_short -o pmc -- \
    python3 tools/sweep_sha3.py 4194304x64x0 > $OUT/pmc_short.log 2>&1 || echo "PMC pass short failed"
N_LIST=128 timeout -k 10 200 rocprofv3 --pmc $SQC --output-format csv -d $OUT/pmc_${R}_wide -o pmc -- \
    python3 tools/bench_wide.py > $OUT/pmc_wide.log 2>&1 || echo "PMC pass wide failed"
B=$(python -c "import json;print(json.load(open('profiles/${R}_bench_line.json'))['config']['batch_per_gpu'])")
S=$(python -c "import json;print(json.load(open('profiles/${R}_bench_line.json'))['config']['msg_stride'])")
D=$(python -c "import bench;print(bench.kernel_source_digest())")
CAPY_PMC_ITEMS=$B CAPY_PMC_STRIDE=$S CAPY_PMC_DIGEST=$D python tools/summarize_pmc.py profiles/${R}_pmc_summary.json $OUT/pmc_${R}_short $OUT/pmc_${R}_wide $OUT/pmc_${R}_FETCH_SIZE $OUT/pmc_${R}_WRITE_SIZE $OUT/pmc_${R}_SQ_WAVES
# the contract line last: bench.py reads the traffic figure from the summary written above
python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err && tail -1 $OUT/bench.json > profiles/${R}_bench_line.json
python tools/bench_configs.py 2> /dev/null > $OUT/configs.jsonl && cp $OUT/configs.jsonl profiles/${R}_configs_2_to_5.jsonl
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/prof_$R/bench_kernel_trace.csv")))
with open("profiles/${R}_bench_sponge_dispatches.txt", "w") as f:
    f.write("# per-dispatch durations (ms) of the sponge kernels in the kernel-trace run of bench.py --steps 5 --warmup 1\n"
            "# (bench.py ramps the clocks with unrelated work first, so the warm-up step's launches are like the timed ones)\n")
    for r in rows:
        if "sponge" in r["Kernel_Name"]:
            f.write("%-45s %10.3f\n" % (r["Kernel_Name"].split("(")[0][5:],
                                        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
# profiles/ of the GPU box does not travel back (only gpurun_out/ is merged): leave a copy there
mkdir -p $OUT/profiles_$R && cp profiles/${R}_bench_line.json profiles/${R}_bench_kernel_stats.csv profiles/${R}_pmc_summary.json \
    profiles/${R}_configs_2_to_5.jsonl profiles/${R}_bench_sponge_dispatches.txt $OUT/profiles_$R/ 2> /dev/null
echo done
