set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
{
echo "# sponge_mixed_kernel<17> (direct per-lane loads, default) vs sponge_mixed_staged_kernel<17> (LDS-staged, debug bit 6)"
echo "# bench.py --steps 3 --warmup 1 --no-cpu-baseline --ed448-pairs 0 [--lanes 16384], same box, alternating"
for i in 1 2 3; do
for L in 0 16384; do
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --ed448-pairs 0 --lanes $L 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run $i', 'staged' if $L else 'direct', 'GiB/s', round(r['value'],1), 'kernel_ms', round(r['roofline']['kernel_ms'],2), 'frac', round(r['roofline']['frac'],4))"
done; done
} | tee gpurun_out/direct_loads_ab.txt
