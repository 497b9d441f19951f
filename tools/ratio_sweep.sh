# split of the rotating schedule's phases (blocks per two-lane phase / blocks per one-lane phase) at the headline batch
for R in ${RATIOS:-1.50 1.44 1.47 1.53 1.56}; do
CAPY_DEBUG=mixed_ratio=$R timeout -k 10 250 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --ed448-pairs 0 ${B:+--batch $B} 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ratio $R batch', r['config']['batch_per_gpu'], 'GiB/s', round(r['value'],1), 'kernel_ms', round(r['roofline']['kernel_ms'],3))" || exit 1
done
for B in ${BATCHES:-54592}; do
timeout -k 10 250 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --ed448-pairs 0 --batch $B 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ratio default batch', r['config']['batch_per_gpu'], 'GiB/s', round(r['value'],1), 'kernel_ms', round(r['roofline']['kernel_ms'],3), r['roofline']['launches_per_step'])" || exit 1
done
