#!/usr/bin/env python3
"""Ragged host batch through capy_sha3_batch with and without the length-sorted processing order.
Run under rocprofv3 --kernel-trace and compare the two sponge-kernel durations:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ragged -o r -- python3 tools/bench_ragged.py"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
rng = np.random.default_rng(0xCA9C0006)
n = 32768
lens = np.exp(rng.uniform(np.log(1024), np.log(1 << 20), n)).astype(np.uint64)  # log-uniform 1 KiB .. 1 MiB
offs = np.zeros(n + 1, dtype=np.uint64)
offs[1:] = np.cumsum((lens + 7) // 8 * 8)  # 8-byte aligned starts: no re-packing
total = int(offs[-1])
buf = np.frombuffer(rng.bytes(1 << 20) * (total // (1 << 20) + 1), dtype=np.uint8, count=total).copy()
ends = (offs[:-1] + lens).astype(np.uint64)
# the C ABI takes n+1 offsets with len_i = offsets[i+1] - offsets[i]: pass exact-length messages by packing tightly
tight = np.zeros(n + 1, dtype=np.uint64)
tight[1:] = np.cumsum(lens)
packed = np.concatenate([buf[int(offs[i]):int(ends[i])] for i in range(n)])
dig = np.zeros(n * 32, dtype=np.uint8)
for flags, name in ((0, "warm-up"), (4 << 8, "input order"), (0, "longest first"), (4 << 8, "input order"),
                    (0, "longest first")):
    _lib.check(lib.capy_set_sponge_lanes(flags))
    _lib.check(lib.capy_sha3_batch(256, n, packed.ctypes.data_as(C.c_void_p), tight.ctypes.data_as(C.c_void_p),
                                   dig.ctypes.data_as(C.c_void_p)))
    for i in (0, 1, n // 2, n - 1):
        m = packed[int(tight[i]):int(tight[i + 1])].tobytes()
        assert dig[32 * i:32 * i + 32].tobytes() == hashlib.sha3_256(m).digest(), i
    print("%s: ok, %d messages, %.2f GiB, longest %d B" % (name, n, int(tight[-1]) / 2**30, int(lens.max())), flush=True)
