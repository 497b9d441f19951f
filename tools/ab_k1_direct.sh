# one-lane latency-tuned kernel at <= one wave per SIMD: per-lane loads (default) vs LDS-staged (DBG=64)
for i in 1 2; do
for D in 0 64; do echo "DBG=$D (64 = staged)"
DBG=$D timeout -k 10 200 python tools/sweep_sha3.py 65536x1048704x1,49152x1048704x1,65536x65536x1,65536x8192x1 2>/dev/null || exit 1
DBG=$D N=65536 MAXLEN=262144 MODE=ragged REPS=3 timeout -k 10 200 python tools/bench_ragged_dev.py 2>/dev/null | grep -v amdgpu || exit 1
DBG=$D N=49152 MAXLEN=16384 MODE=ragged REPS=5 timeout -k 10 200 python tools/bench_ragged_dev.py 2>/dev/null | grep -v amdgpu || exit 1
done; done
