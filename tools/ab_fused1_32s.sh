# r06: four-lane fused sha3_encrypt (A, the r05 choice up to 32 S items) against the one-lane form as one lone wave per SIMD (B)
set -e
export NS=24576,26624,28672,29696,30720,31744,32768 LEN=1048576 REPS=3
echo "== A: four-lane up to 32 S (the r05 choice: fused1_min=32769)"; CAPY_DEBUG=fused1_min=32769 python3 tools/sweep_fused1.py
echo "== B: one-lane (fused1_min=0)"; CAPY_DEBUG=fused1_min=0 python3 tools/sweep_fused1.py
echo "== A again"; CAPY_DEBUG=fused1_min=32769 python3 tools/sweep_fused1.py
echo "== B again"; CAPY_DEBUG=fused1_min=0 python3 tools/sweep_fused1.py
export NS=28672,30720,32768 LEN=5242880 REPS=2
echo "== 5 MiB A"; CAPY_DEBUG=fused1_min=32769 python3 tools/sweep_fused1.py
echo "== 5 MiB B"; CAPY_DEBUG=fused1_min=0 python3 tools/sweep_fused1.py
echo "== 5 MiB B, lone_direct=0"; CAPY_DEBUG=fused1_min=0,fused1_lone_direct=0 python3 tools/sweep_fused1.py
export D=256 NS=30720,32768 LEN=1048576 REPS=3
echo "== D256 A"; CAPY_DEBUG=fused1_min=32769 python3 tools/sweep_fused1.py
echo "== D256 B"; CAPY_DEBUG=fused1_min=0 python3 tools/sweep_fused1.py
