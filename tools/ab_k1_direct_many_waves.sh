for i in 1 2; do
for W in 2 4096; do echo "CAPY_DIRECT_MAX_WAVES=$W"
CAPY_DIRECT_MAX_WAVES=$W N=2097152 MAXLEN=2048 MODE=ragged REPS=5 timeout -k 10 200 python tools/bench_ragged_dev.py 2>/dev/null | grep -v amdgpu || exit 1
CAPY_DIRECT_MAX_WAVES=$W N=262144 MAXLEN=65536 MODE=ragged REPS=3 timeout -k 10 200 python tools/bench_ragged_dev.py 2>/dev/null | grep -v amdgpu || exit 1
CAPY_DIRECT_MAX_WAVES=$W N=1048576 MAXLEN=16384 MODE=ragged REPS=3 timeout -k 10 200 python tools/bench_ragged_dev.py 2>/dev/null | grep -v amdgpu || exit 1
done; done
