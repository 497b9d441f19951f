#!/usr/bin/env python3
"""Fixed-base multiplication with constant-address lookups alone (capy_ed448_set_hardened(3)), N items, for counter passes."""
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
n = int(os.environ.get("N", "65536"))
sc = torch.empty(n * 56, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_fill_random_dev(sc.data_ptr(), sc.numel(), 4, sp))
out = torch.empty(n * 112, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_ed448_set_hardened(3))
for _ in range(3):
    _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), out.data_ptr(), sp))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(5):
    _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), out.data_ptr(), sp))
e1.record(st)
torch.cuda.synchronize()
print("hardened fixed base, n = %d: %.3f ms per call" % (n, e0.elapsed_time(e1) / 5))
