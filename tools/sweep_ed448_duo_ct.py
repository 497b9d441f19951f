#!/usr/bin/env python3
"""Secret-scalar (constant-address) variable base between 16 384 and 32 768 items: vb_duo_ct_kernel (csrc/ed448_duo.h: two lanes
per item, the window table half in registers and half in LDS, ONE round of waves) against the quad form in two rounds
(CAPY_DEBUG=ed448_duo_ct=0: run the script twice) and against the indexed kernels for byte identity.
usage: python3 tools/sweep_ed448_duo_ct.py [n ...]   -> profiles/r05_ed448_duo_ct.txt"""
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
ns = [int(a) for a in sys.argv[1:]] or [16384, 16400, 18432, 20480, 24576, 28672, 32768, 32800]
nmax = max(ns)
sc, tsc = (torch.empty(nmax * 56, dtype=torch.uint8, device=dev) for _ in range(2))
for t, seed in ((sc, 4), (tsc, 41)):
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), nmax * 56, seed, sp))
pts = torch.empty(nmax * 112, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_ed448_basemul_batch_dev(nmax, tsc.data_ptr(), pts.data_ptr(), sp))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


print("# CAPY_DEBUG=%s" % os.environ.get("CAPY_DEBUG", "default"))
print("#      n | secret scalars (CAPY_HARDEN_ALL) ms, family | public scalars ms, family | identical")
fam = C.c_int(0)
try:
    for n in ns:
        res = {}
        for name, mode in (("secret", 1), ("public", 0)):
            _lib.check(lib.capy_ed448_set_hardened(mode))
            vb = torch.zeros(n * 112, dtype=torch.uint8, device=dev)
            t = timed(lambda: _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), vb.data_ptr(), sp)))
            lib.capy_debug_last_curve_kernel(C.byref(fam), None)
            res[name] = (t, vb, fam.value)
        a, b = res["secret"], res["public"]
        print("%8d | %10.3f  %3d | %10.3f  %3d | %s" % (n, a[0], a[2], b[0], b[2], torch.equal(a[1], b[1])), flush=True)
finally:
    _lib.check(lib.capy_ed448_set_hardened(4))  # CAPY_HARDEN_PROTOCOL
