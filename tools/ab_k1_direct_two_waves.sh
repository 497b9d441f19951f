for i in 1 2; do
for W in 1 2; do echo "CAPY_DIRECT_MAX_WAVES=$W"
CAPY_DIRECT_MAX_WAVES=$W timeout -k 10 200 python tools/sweep_sha3.py 131072x1048704x1,98304x262144x1,131072x65536x1,131072x8192x1 2>/dev/null || exit 1
CAPY_DIRECT_MAX_WAVES=$W N=131072 MAXLEN=65536 MODE=ragged REPS=3 timeout -k 10 200 python tools/bench_ragged_dev.py 2>/dev/null | grep -v amdgpu || exit 1
done; done
