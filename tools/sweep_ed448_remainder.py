#!/usr/bin/env python3
"""Variable-base multiplication and [a]G + [b]P for batches BETWEEN the wave quanta of the one-item-per-lane kernels
(q x 65 536 + x items): with and without the remainder peeled off into its own launch (CAPY_DEBUG=ed448_peel=0 / 1, read
once per process: run this script twice).  Also checks byte identity of a remainder batch against the unpeeled form when run
with CHECK=1.   -> profiles/r04_ed448_remainder.txt"""
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


def rand(nb, seed):
    t = torch.empty(nb, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), nb, seed, sp))
    return t


def timed(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


tag = os.environ.get("CAPY_DEBUG", "default")
print("# %s: n | variable base ms (M/s) | hardened variable base ms | [a]G + [b]P ms" % tag)
for n in (65536, 65600, 69632, 81920, 98304, 98305, 114688, 131072, 147456, 163840, 196608, 200000, 229376, 262144, 278528, 294912):
    sc, asc, tsc = rand(n * 56, 4), rand(n * 56, 5), rand(n * 56, 41)
    pts = torch.empty(n * 112, dtype=torch.uint8, device=dev)
    o = torch.empty(n * 112, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp))
    _lib.check(lib.capy_ed448_set_hardened(0))
    vb = timed(lambda: _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), o.data_ptr(), sp)))
    ref_vb = o.clone()
    ds = timed(lambda: _lib.check(lib.capy_ed448_double_scalarmul_batch_dev(n, asc.data_ptr(), sc.data_ptr(), pts.data_ptr(), o.data_ptr(), sp)))
    _lib.check(lib.capy_ed448_set_hardened(1))
    vc = timed(lambda: _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), o.data_ptr(), sp)))
    same = torch.equal(o, ref_vb)
    _lib.check(lib.capy_ed448_set_hardened(4))
    print("%8d | %7.3f (%5.2f) | %7.3f | %7.3f | hardened == indexed: %s" % (n, vb, n / vb / 1e3, vc, ds, same), flush=True)
