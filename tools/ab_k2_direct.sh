for i in 1 2; do
for D in 0 64; do echo "DBG=$D (64 = staged)"; DBG=$D timeout -k 10 200 python tools/sweep_sha3.py 8192x1048704x2,16384x1048704x2,32768x1048704x2,32768x1048576x2 2>/dev/null || exit 1; done
done
