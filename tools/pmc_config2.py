#!/usr/bin/env python3
"""BASELINE config 2 (2^20 x kmac_xof(k, "", 8192 bits, "SKE", D512)) a few times, for rocprofv3 --pmc / --kernel-trace runs."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = 1 << 20
keys = torch.empty(n * 64, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_fill_random_dev(keys.data_ptr(), n * 64, 2, sp))
out = torch.empty(n * 1024, dtype=torch.uint8, device=dev)
for _ in range(int(os.environ.get("REPS", "5"))):
    _lib.check(lib.capy_kmac_xof_batch_dev(512, n, keys.data_ptr(), 64, 64, None, None, None, 0, 0, 8192, b"SKE", 3, out.data_ptr(), 1024, sp))
torch.cuda.synchronize()
