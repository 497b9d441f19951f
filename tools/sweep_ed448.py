#!/usr/bin/env python3
"""Ed448 variable-base / fixed-base throughput over batch size (device-pointer C ABI, HIP events)."""
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


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


nmax = 1 << 20
sc = torch.empty(nmax * 56, dtype=torch.uint8, device=dev)
tsc = torch.empty(nmax * 56, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_fill_random_dev(sc.data_ptr(), nmax * 56, 4, sp))
_lib.check(lib.capy_fill_random_dev(tsc.data_ptr(), nmax * 56, 41, sp))
pts = torch.empty(nmax * 112, dtype=torch.uint8, device=dev)
out = torch.empty(nmax * 112, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_ed448_basemul_batch_dev(nmax, tsc.data_ptr(), pts.data_ptr(), sp))
print("# n        var-base ms   M/s     fixed-base ms   M/s")
for e in range(10, 21, 2):
    n = 1 << e
    vb = timeit(lambda: _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), out.data_ptr(), sp)))
    fb = timeit(lambda: _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), out.data_ptr(), sp)))
    print("2^%-2d  %10.3f  %7.2f   %10.3f  %8.2f" % (e, vb, n / vb / 1e3, fb, n / fb / 1e3), flush=True)
