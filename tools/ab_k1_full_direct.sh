# issue-tuned one-lane instance (n > 128 per SIMD, uniform): per-lane loads (default) vs LDS-staged (DBG=64)
for i in 1 2; do
for D in 0 64; do echo "DBG=$D (64 = staged)"
DBG=$D timeout -k 10 200 python tools/sweep_sha3.py 2097152x1024x0,1048576x4096x0,262144x65536x0,262144x524288x0,4194304x64x0 2>/dev/null || exit 1
done; done
