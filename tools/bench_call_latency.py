#!/usr/bin/env python3
"""Per-call latency of the host-buffer C ABI for small batches of small messages (what a one-message-at-a-time caller of
the reference's API shape pays): capy_sha3_batch, capy_kmac_xof_batch, capy_schnorr_sign_batch, capy_schnorr_verify_batch."""
import ctypes as C
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
rng = random.Random(1)


def per_call(fn, reps=200):  # fn returns a capy status code
    for _ in range(5):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


for n, mlen in ((1, 64), (1, 1024), (1, 65536), (64, 1024), (1024, 1024)):
    msgs = C.create_string_buffer(rng.randbytes(n * mlen), n * mlen)
    offs = (C.c_uint64 * (n + 1))(*[i * mlen for i in range(n + 1)])
    dig = (C.c_uint8 * (n * 32))()
    out = (C.c_uint8 * (n * 64))()
    keys = C.create_string_buffer(rng.randbytes(n * 32), n * 32)
    pubs = (C.c_uint8 * (n * 112))()
    h = (C.c_uint8 * (n * 56))()
    z = (C.c_uint8 * (n * 56))()
    st = (C.c_int32 * n)()
    _lib.check(lib.capy_keypair_batch(512, n, keys, 32, None, pubs))
    t_sha = per_call(lambda: _lib.check(lib.capy_sha3_batch(256, n, msgs, offs, dig)))
    t_kmac = per_call(lambda: _lib.check(lib.capy_kmac_xof_batch(512, n, keys, 32, None, msgs, offs, 512, b"T", 1, out)))
    t_sign = per_call(lambda: _lib.check(lib.capy_schnorr_sign_batch(512, n, keys, 32, None, msgs, offs, h, z)), 50)
    t_ver = per_call(lambda: _lib.check(lib.capy_schnorr_verify_batch(512, n, pubs, msgs, offs, h, z, st)), 50)
    assert not any(st)
    print("n=%5d x %6d B: sha3 %8.1f us  kmac_xof %8.1f us  sign %8.1f us  verify %8.1f us per call" % (n, mlen, t_sha, t_kmac, t_sign, t_ver),
          flush=True)

# ---- the same operations through the device entry points (inputs already on the device, one stream, time per call with
# the stream drained after every call): what is left is kernel time + launch gaps; the difference to the lines above is
# the host-buffer plumbing (allocations, small copies)
import torch  # noqa: E402

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
sp = C.c_void_p(st.cuda_stream)
for n, mlen in ((1, 64), (1, 1024), (64, 1024)):
    msgs = torch.randint(0, 256, (n * mlen,), dtype=torch.uint8, device=dev)
    pws = torch.randint(0, 256, (n * 32,), dtype=torch.uint8, device=dev)
    pubs = torch.zeros(n * 112, dtype=torch.uint8, device=dev)
    h = torch.zeros(n * 56, dtype=torch.uint8, device=dev)
    z = torch.zeros(n * 56, dtype=torch.uint8, device=dev)
    stt = torch.zeros(n, dtype=torch.int32, device=dev)

    def sync_call(fn):
        _lib.check(fn())
        torch.cuda.synchronize()
        return 0

    t_kp = per_call(lambda: sync_call(lambda: lib.capy_keypair_batch_dev(512, n, pws.data_ptr(), 32, None, pubs.data_ptr(), sp)), 50)
    t_sign = per_call(lambda: sync_call(lambda: lib.capy_schnorr_sign_batch_dev(512, n, pws.data_ptr(), 32, None, msgs.data_ptr(), None,
                                                                                 mlen, mlen, h.data_ptr(), z.data_ptr(), sp)), 50)
    t_ver = per_call(lambda: sync_call(lambda: lib.capy_schnorr_verify_batch_dev(512, n, pubs.data_ptr(), msgs.data_ptr(), None, mlen, mlen,
                                                                                  h.data_ptr(), z.data_ptr(), stt.data_ptr(), sp)), 50)
    assert int(stt.sum().item()) == 0
    print("device entry points, n=%3d x %5d B: keypair %7.1f us  sign %7.1f us  verify %7.1f us per call" % (n, mlen, t_kp, t_sign, t_ver),
          flush=True)
