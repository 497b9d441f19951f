python -c "import torch; f,t=torch.cuda.mem_get_info(); print('free GiB', f/2**30, 'total GiB', t/2**30)"
for B in 49152 52416 54528 55296; do
timeout -k 10 250 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --ed448-pairs 0 --batch $B 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $B ->', r['config']['batch_per_gpu'], 'GiB/s', round(r['value'],1), 'ms/step', round(r['ms_per_step'],2), 'frac', round(r['roofline']['frac'],4), r['roofline'].get('launches_per_step'))" || exit 1
done
