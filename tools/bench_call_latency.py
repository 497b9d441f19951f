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


def per_call(fn, reps=200):
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
