#!/usr/bin/env python3
"""Fixed-base (default) or variable-base (OP=vb) multiplication with constant-address lookups alone
(capy_ed448_set_hardened(1)  # CAPY_HARDEN_ALL), N items, for counter passes."""
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
pts = torch.empty(n * 112, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), sp))
_lib.check(lib.capy_ed448_set_hardened(1)  # CAPY_HARDEN_ALL)
vb = os.environ.get("OP", "fb") == "vb"


def run():
    if vb:
        _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), out.data_ptr(), sp))
    else:
        _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), out.data_ptr(), sp))


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(5):
    run()
e1.record(st)
torch.cuda.synchronize()
print("hardened %s base, n = %d: %.3f ms per call" % ("variable" if vb else "fixed", n, e0.elapsed_time(e1) / 5))
