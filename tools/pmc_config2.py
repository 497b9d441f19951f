#!/usr/bin/env python3
"""BASELINE config 2 alone (2^20 x KMACXOF256(k_i, "", 8192 bits, "SKE"), 64-byte keys, 1 KiB out per unit), for counter passes:
rocprofv3 --pmc ... -- python3 tools/pmc_config2.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
sp = C.c_void_p(st.cuda_stream)
n = int(os.environ.get("N", str(1 << 20)))
_lib.check(lib.capy_set_sponge_lanes(int(os.environ.get("LANES", "0"), 0)))  # A/B switches (bit 20: rate-block stores of r03)
keys = torch.empty(n * 64, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_fill_random_dev(keys.data_ptr(), keys.numel(), 5, sp))
out = torch.empty(n * 1024, dtype=torch.uint8, device=dev)


def run():
    _lib.check(lib.capy_kmac_xof_batch_dev(512, n, keys.data_ptr(), 64, 64, None, None, None, 0, 0, 8192, b"SKE", 3, out.data_ptr(), 1024, sp))


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
REPS = int(os.environ.get("REPS", "5"))
for _ in range(REPS):
    run()
e1.record(st)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / REPS
print("config 2: n = %d  %.3f ms per call  %.1f M units/s  %.2f G permutations/s  %.0f GB/s written" % (n, ms, n / ms / 1e3, 11 * n / ms / 1e6, n * 1024 / ms / 1e6))
