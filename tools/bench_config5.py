#!/usr/bin/env python3
"""BASELINE config 5 through the host-buffer C ABI (PCIe inclusive), plus ECDHIES: 2^16 x 1 KiB messages, D512.
The timed region of each line is one C call (uploads + kernels + downloads)."""
import ctypes as C
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
rng = random.Random(5)
n, mlen = 1 << 16, int(os.environ.get("MLEN", "1024"))
msgs_h = C.create_string_buffer(os.urandom(n * mlen), n * mlen)
pws_h = C.create_string_buffer(rng.randbytes(n * 64), n * 64)
kr_h = C.create_string_buffer(rng.randbytes(n * 56), n * 56)
offs_h = (C.c_uint64 * (n + 1))(*[i * mlen for i in range(n + 1)])
pubs_h = (C.c_uint8 * (n * 112))()
h_h = (C.c_uint8 * (n * 56))()
z_h = (C.c_uint8 * (n * 56))()
zxy_h = (C.c_uint8 * (n * 112))()
tags_h = (C.c_uint8 * (n * 56))()
st_h = (C.c_int32 * n)()


def best(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _lib.check(fn())
        ts.append(time.perf_counter() - t0)
    return min(ts)


_lib.check(lib.capy_keypair_batch(512, n, pws_h, 64, None, pubs_h))  # warm (fixed-base table build)
tk = best(lambda: lib.capy_keypair_batch(512, n, pws_h, 64, None, pubs_h))
ts = best(lambda: lib.capy_schnorr_sign_batch(512, n, pws_h, 64, None, msgs_h, offs_h, h_h, z_h))
tv = best(lambda: lib.capy_schnorr_verify_batch(512, n, pubs_h, msgs_h, offs_h, h_h, z_h, st_h))
assert not any(st_h)
before = bytes(msgs_h[:4096])
te = best(lambda: lib.capy_key_encrypt_batch(512, n, pubs_h, kr_h, msgs_h, offs_h, zxy_h, tags_h), 1)
td = best(lambda: lib.capy_key_decrypt_batch(512, n, pws_h, 64, None, zxy_h, msgs_h, offs_h, tags_h, st_h), 1)
assert not any(st_h) and bytes(msgs_h[:4096]) == before
print("2^16 x %d B, D512, host-buffer C ABI: keypair %.2f ms (%.1f M/s)  sign %.2f ms (%.1f M/s)  verify %.2f ms (%.1f M/s)  "
      "key_encrypt %.2f ms (%.1f M/s)  key_decrypt %.2f ms (%.1f M/s)" % (
          mlen, tk * 1e3, n / tk / 1e6, ts * 1e3, n / ts / 1e6, tv * 1e3, n / tv / 1e6, te * 1e3, n / te / 1e6, td * 1e3, n / td / 1e6))
