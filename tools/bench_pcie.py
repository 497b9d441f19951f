#!/usr/bin/env python3
"""PCIe-inclusive SHA3-256 through the host-pointer entry point (pageable host memory): cold = first call on a freshly
written buffer, warm = the same buffer again.  SIZE_GIB (default 4), MSG_KIB (default 64).  CAPY_NO_COPY_PIPELINE=1 for the A/B."""
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
mlen = int(os.environ.get("MSG_KIB", "64")) * 1024
nmsg = int(float(os.environ.get("SIZE_GIB", "4")) * 2**30) // mlen
_lib.check(lib.capy_sha3_batch(256, 1, (C.c_uint8 * 8)(), (C.c_uint64 * 2)(0, 8), (C.c_uint8 * 32)()))  # context up
for rep in range(2):
    host = (C.c_uint8 * (nmsg * mlen))()
    C.memset(host, 0x5A + rep, nmsg * mlen)
    offs = (C.c_uint64 * (nmsg + 1))(*[i * mlen for i in range(nmsg + 1)])
    dig = (C.c_uint8 * (nmsg * 32))()
    t0 = time.perf_counter()
    _lib.check(lib.capy_sha3_batch(256, nmsg, host, offs, dig))
    tc = time.perf_counter() - t0
    t0 = time.perf_counter()
    _lib.check(lib.capy_sha3_batch(256, nmsg, host, offs, dig))
    th = time.perf_counter() - t0
    assert bytes(dig[:32]) == hashlib.sha3_256(bytes(host[:mlen])).digest()
    assert bytes(dig[-32:]) == hashlib.sha3_256(bytes(host[-mlen:])).digest()
    print("%d x %d KiB = %.1f GiB: cold %.3f s (%.1f GiB/s)  warm %.3f s (%.1f GiB/s)" % (
        nmsg, mlen // 1024, nmsg * mlen / 2**30, tc, nmsg * mlen / tc / 2**30, th, nmsg * mlen / th / 2**30), flush=True)
    del host
