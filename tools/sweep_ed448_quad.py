#!/usr/bin/env python3
"""Variable-base multiplication over batch sizes in and around the regime of the four-lanes-per-item kernel
(csrc/ed448_quad.h): ms per call and a digest of the outputs, so that two runs -- CAPY_DEBUG=ed448_quad_max=0 (never) against
CAPY_DEBUG=ed448_quad_min=0,ed448_quad_max=100000000 (always) -- can be compared line by line.
usage: python3 tools/sweep_ed448_quad.py [n ...]"""
import ctypes as C
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
sp = C.c_void_p(st.cuda_stream)
ns = [int(a) for a in sys.argv[1:]] or [1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 40960, 49152, 65536]
nmax = max(ns)
sc = torch.empty(nmax * 56, dtype=torch.uint8, device=dev)
tsc = torch.empty(nmax * 56, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_fill_random_dev(sc.data_ptr(), nmax * 56, 4, sp))
_lib.check(lib.capy_fill_random_dev(tsc.data_ptr(), nmax * 56, 41, sp))
pts = torch.empty(nmax * 112, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_ed448_basemul_batch_dev(nmax, tsc.data_ptr(), pts.data_ptr(), sp))
out = torch.empty(nmax * 112, dtype=torch.uint8, device=dev)
fam = C.c_int(0)
print("# CAPY_DEBUG=%s" % os.environ.get("CAPY_DEBUG", ""))
for n in ns:
    def run():
        _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), out.data_ptr(), sp))
    run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        run()
        e1.record(st)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    lib.capy_debug_last_curve_kernel(C.byref(fam), None)
    dig = hashlib.sha256(bytes(out[:n * 112].cpu().numpy())).hexdigest()[:16]
    print("n = %6d  %8.3f ms  %7.2f M/s  family %2d  outputs %s" % (n, best, n / best / 1e3, fam.value, dig), flush=True)
