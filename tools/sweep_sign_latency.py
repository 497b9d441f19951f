#!/usr/bin/env python3
"""r06, VERDICT r5 item 4 (overlap the message upload of host-buffer `sign` with its compute): what a chunk pipeline could gain.
Prints (a) the latency of capy_schnorr_sign_batch_dev (device-resident, D512, 1 KiB messages) over the chunk sizes a pipeline
would use -- a pipeline's tail is upload_end + latency(last chunk) --, (b) the same with K chunks running concurrently on K
streams, (c) the host-buffer call as it is, and (d) the bare PCIe legs of that call (64 MiB up from pageable memory the way the
library copies it, 7 MiB down).   usage: python3 tools/sweep_sign_latency.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
L, D = 1024, 512
NMAX = 1 << 16
dev = torch.device("cuda", 0)


def rnd(nbytes, seed):
    t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), nbytes, seed, None))
    return t


msgs, pws = rnd(NMAX * L, 1), rnd(NMAX * 64, 2)
h, z = torch.zeros(NMAX * 56, dtype=torch.uint8, device=dev), torch.zeros(NMAX * 56, dtype=torch.uint8, device=dev)


def sign(first, n, stream):
    _lib.check(lib.capy_schnorr_sign_batch_dev(D, n, pws.data_ptr() + first * 64, 64, None, msgs.data_ptr() + first * L, None, L, L,
                                               h.data_ptr() + first * 56, z.data_ptr() + first * 56, C.c_void_p(stream.cuda_stream)))


print("# (a) sign_dev latency per call, device-resident, one stream: n | ms | M/s")
st = torch.cuda.current_stream()
for n in (512, 1024, 2048, 3584, 4096, 8192, 16384, 32768, 65536):
    sign(0, n, st)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        sign(0, n, st)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%6d | %.3f | %.1f" % (n, best * 1e3, n / best / 1e6), flush=True)

print("# (b) K equal chunks of 65536 / K items on K streams at once: K | ms for all | M/s")
for K in (1, 2, 4, 8, 16):
    streams = [torch.cuda.Stream() for _ in range(K)]
    nc = NMAX // K
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(K):
            sign(c * nc, nc, streams[c])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print("%3d | %.3f | %.1f" % (K, el * 1e3, NMAX / el / 1e6), flush=True)

print("# (c) the host-buffer call (capy_schnorr_sign_batch, pageable numpy buffers, warm): ms | M/s")
hm = np.frombuffer(bytes(msgs.cpu().numpy()), dtype=np.uint8).copy()
hp = np.frombuffer(bytes(pws.cpu().numpy()), dtype=np.uint8).copy()
offs = (np.arange(NMAX + 1, dtype=np.uint64) * L)
hh, hz = np.zeros(NMAX * 56, dtype=np.uint8), np.zeros(NMAX * 56, dtype=np.uint8)
best = 1e9
for _ in range(6):
    t0 = time.perf_counter()
    _lib.check(lib.capy_schnorr_sign_batch(D, NMAX, hp.ctypes.data, 64, None, hm.ctypes.data, offs.ctypes.data, hh.ctypes.data, hz.ctypes.data))
    best = min(best, time.perf_counter() - t0)
print("%.3f | %.1f" % (best * 1e3, NMAX / best / 1e6))
assert bytes(hh) == bytes(h.cpu().numpy()) and bytes(hz) == bytes(z.cpu().numpy()), "host and device forms differ"

print("# (d) the PCIe legs alone: torch pageable -> device 64 MiB + 4 MiB, device -> pageable 2 x 3.5 MiB: ms")
tm, tp = torch.from_numpy(hm), torch.from_numpy(hp)
best_up, best_dn = 1e9, 1e9
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    msgs.copy_(tm)
    pws.copy_(tp)
    torch.cuda.synchronize()
    best_up = min(best_up, time.perf_counter() - t0)
    t0 = time.perf_counter()
    a, b = h.cpu(), z.cpu()
    best_dn = min(best_dn, time.perf_counter() - t0)
print("up %.3f  down %.3f  -> transfer floor %.3f ms = %.1f M/s" % (best_up * 1e3, best_dn * 1e3, (best_up + best_dn) * 1e3, NMAX / (best_up + best_dn) / 1e6))
