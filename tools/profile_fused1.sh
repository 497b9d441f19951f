#!/bin/bash
# r05: everything behind profiles/r05_fused_one_lane.txt on ONE box: the one-lane-per-sponge fused sha3_encrypt / sha3_decrypt
# kernel against the r04 forms (CAPY_DEBUG=fused1_min=100000000: four lanes per item up to 98 304 items, two passes beyond), the
# schedules over the batch size, and the counters.  usage: bash tools/profile_fused1.sh -> gpurun_out/r05_fused_one_lane.txt
export TMPDIR=/tmp
O=gpurun_out/r05_fused_one_lane.txt
{
echo "# profiles/r05_fused_one_lane.txt -- $(date -u +%F) -- $(python3 -c 'import torch;print(torch.cuda.get_device_name(0))' 2>/dev/null)"
echo "## 1. sha3_encrypt / sha3_decrypt D512, device-resident uniform batches, stride = len + 128; best of 2; GiB/s = n x len / 2^30 / s"
echo "### the r04 forms (CAPY_DEBUG=fused1_min=100000000)"
CAPY_DEBUG=fused1_min=100000000 NS=32768,36864,40960,49152,57344,65536,98304,131072 REPS=2 python3 tools/sweep_fused1.py 2>&1 | grep -v amdgpu
CAPY_DEBUG=fused1_min=100000000 LEN=4194304 NS=49152 REPS=2 python3 tools/sweep_fused1.py 2>&1 | grep -v amdgpu
echo "### r05 default (kind 20 / 22: four lanes per item; 23: one lane per sponge, one launch; 24: time slices; 25: rotating occupancy)"
NS=32768,34816,36864,40960,45056,49152,53248,57344,61440,65536,66000,73728,81920,98304,100000,114688,131072,140000,163840 REPS=2 python3 tools/sweep_fused1.py 2>&1 | grep -v amdgpu
LEN=4194304 NS=40960,49152 REPS=2 python3 tools/sweep_fused1.py 2>&1 | grep -v amdgpu
LEN=65536 NS=262144,1048576 REPS=2 python3 tools/sweep_fused1.py 2>&1 | grep -v amdgpu
echo "### D256 (168-byte blocks, filed in two parts)"
D=256 NS=49152,65536,131072 REPS=2 python3 tools/sweep_fused1.py 2>&1 | grep -v amdgpu
echo "### schedules off: one launch for every size (CAPY_DEBUG=fused1_rot=0,fused1_slices=0)"
CAPY_DEBUG=fused1_rot=0,fused1_slices=0 NS=40960,49152,57344,66000,81920,100000,140000 REPS=2 python3 tools/sweep_fused1.py 2>&1 | grep -v amdgpu
echo "## 2. counters (rocprofv3 --pmc, separate passes; sums over the launches of a schedule)"
bash tools/pmc_fused1.sh 2>&1 | tail -9
echo "### plain stores instead of sc1 (CAPY_DEBUG=fused1_store=0)"
CAPY_DEBUG=fused1_store=0 CFGS="131072x1048576 65536x1048576" bash tools/pmc_fused1.sh 2>&1 | tail -5
} > $O 2>&1
cat $O
