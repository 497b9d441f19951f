# r06: the 24 S boundary of the one-lane fused kernel on SHORT messages (the boundary was set on 1 MiB / 5 MiB messages)
set -e
for LEN in 1024 16384 131072; do
  export NS=24576,28672,32768 LEN=$LEN REPS=5
  echo "== LEN=$LEN A: four-lane up to 32 S (fused1_min=32769)"; CAPY_DEBUG=fused1_min=32769 python3 tools/sweep_fused1.py
  echo "== LEN=$LEN B: default (one-lane from 24 S)"; python3 tools/sweep_fused1.py
  echo "== LEN=$LEN A again"; CAPY_DEBUG=fused1_min=32769 python3 tools/sweep_fused1.py
  echo "== LEN=$LEN B again"; python3 tools/sweep_fused1.py
done
