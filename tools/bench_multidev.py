#!/usr/bin/env python3
"""Per-call cost of the in-library sharding (capy_set_devices) at EQUAL TOTAL WORK: the same host-buffer call with no
device list, with {0}, {0,0} and {0,0,0,0} -- on a one-GPU box every shard lands on the same card, so what is measured
is the sharding machinery itself (persistent workers, per-worker scratch pools and buffer cache, shard cut), not a
speed-up.  Workloads: BASELINE config 5 sign / verify (2^16 x 1 KiB, D512), capy_sha3_batch 2^16 x 1 KiB and
65 536 x 64 KiB (4 GiB over PCIe).   python tools/bench_multidev.py > gpurun_out/r03_multidev_overhead.txt"""
import ctypes as C
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
rng = random.Random(5)
n = 1 << 16


def median_of(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _lib.check(fn())
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def set_devices(ids):
    arr = (C.c_int * max(1, len(ids)))(*ids)
    _lib.check(lib.capy_set_devices(arr, len(ids)))


def run(label, fn, reps, check=None):
    base = None
    for ids in ([], [0], [0, 0], [0, 0, 0, 0]):
        set_devices(ids)
        _lib.check(fn())  # first call with this list: workers start, pools fill
        _lib.check(fn())
        med, best = median_of(fn, reps)
        if check:
            check()
        if base is None:
            base = med
        print("%-44s devices %-12s median %8.3f ms  best %8.3f ms  %+6.1f %% vs no list" % (
            label, "{" + ",".join(map(str, ids)) + "}" if ids else "(none)", med * 1e3, best * 1e3, 100 * (med / base - 1)), flush=True)
    set_devices([])


mlen = 1024
msgs_h = C.create_string_buffer(os.urandom(n * mlen), n * mlen)
pws_h = C.create_string_buffer(rng.randbytes(n * 64), n * 64)
offs_h = (C.c_uint64 * (n + 1))(*[i * mlen for i in range(n + 1)])
pubs_h = (C.c_uint8 * (n * 112))()
h_h = (C.c_uint8 * (n * 56))()
z_h = (C.c_uint8 * (n * 56))()
st_h = (C.c_int32 * n)()
dig_h = (C.c_uint8 * (n * 32))()
_lib.check(lib.capy_keypair_batch(512, n, pws_h, 64, None, pubs_h))
print("# hardened mode %s" % os.environ.get("CAPY_HARDENED_MODE", "4 = CAPY_HARDEN_PROTOCOL (default)"))
if "CAPY_HARDENED_MODE" in os.environ:
    _lib.check(lib.capy_ed448_set_hardened(int(os.environ["CAPY_HARDENED_MODE"])))
run("config 5 sign   2^16 x 1 KiB", lambda: lib.capy_schnorr_sign_batch(512, n, pws_h, 64, None, msgs_h, offs_h, h_h, z_h), 9)
run("config 5 verify 2^16 x 1 KiB", lambda: lib.capy_schnorr_verify_batch(512, n, pubs_h, msgs_h, offs_h, h_h, z_h, st_h), 9,
    check=lambda: (_ for _ in ()).throw(AssertionError("verify failed")) if any(st_h) else None)
import hashlib

ref = hashlib.sha3_256(bytes(msgs_h[:mlen])).digest()
run("capy_sha3_batch 2^16 x 1 KiB", lambda: lib.capy_sha3_batch(256, n, msgs_h, offs_h, dig_h), 9,
    check=lambda: (_ for _ in ()).throw(AssertionError("digest")) if bytes(dig_h[:32]) != ref else None)
del msgs_h
big = 65536
blen = 65536
big_h = C.create_string_buffer(big * blen)
C.memset(big_h, 0x5A, big * blen)
boffs = (C.c_uint64 * (big + 1))(*[i * blen for i in range(big + 1)])
bdig = (C.c_uint8 * (big * 32))()
run("capy_sha3_batch 65536 x 64 KiB (4 GiB)", lambda: lib.capy_sha3_batch(256, big, big_h, boffs, bdig), 3)
