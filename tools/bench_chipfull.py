#!/usr/bin/env python3
"""Chip-full sponge regimes (every SIMD holds >= 2 waves): SHA3-256 over uniform device batches under the automatic
kernel choice, and BASELINE config 2.  r03: before / after the blocked round with raised priority around its rotation
blocks (keccak_dev.h: keccak_round_blocked).   python tools/bench_chipfull.py > gpurun_out/r03_chipfull.txt"""
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
_lib.check(lib.capy_set_sponge_lanes(int(os.environ.get("LANES", "0")) | (int(os.environ.get("DBG", "0")) << 8)))
QUICK = os.environ.get("QUICK") == "1"


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for B, L, reps in ((131072, 1 << 20, 3), (262144, 1 << 19, 3), (1 << 20, 4096, 20), (1 << 21, 1024, 20), (1 << 22, 64, 20),
                   (1 << 22, 136 * 3 + 8, 20), (65536, 1 << 20, 3)):
    if QUICK and (B, L) not in ((262144, 1 << 19), (1 << 20, 4096), (1 << 21, 1024)):
        continue
    msgs = torch.empty(B * L, dtype=torch.uint8, device=dev)
    dig = torch.empty(B * 32, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(msgs.data_ptr(), B * L, 1, sp))
    ms = timeit(lambda: _lib.check(lib.capy_sha3_batch_dev(256, B, msgs.data_ptr(), None, L, L, dig.data_ptr(), sp)), reps)
    ok = all(bytes(dig[i * 32:(i + 1) * 32].cpu().numpy()) == hashlib.sha3_256(bytes(msgs[i * L:(i + 1) * L].cpu().numpy())).digest()
             for i in (0, B // 2 + 1, B - 1))
    print("SHA3-256 %8d x %8d B: %8.3f ms  %7.1f GB/s  %6.2f G msgs/s  ok=%s" % (B, L, ms, B * L / (ms * 1e-3) / 1e9, B / (ms * 1e-3) / 1e9, ok),
          flush=True)
    del msgs, dig
n = 1 << 20
keys = torch.empty(n * 64, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_fill_random_dev(keys.data_ptr(), n * 64, 2, sp))
out = torch.empty(n * 1024, dtype=torch.uint8, device=dev)
ms = timeit(lambda: _lib.check(lib.capy_kmac_xof_batch_dev(512, n, keys.data_ptr(), 64, 64, None, None, None, 0, 0, 8192, b"SKE", 3,
                                                           out.data_ptr(), 1024, sp)), 30)
print("config 2: 2^20 x KMACXOF256 1 KiB squeeze: %.3f ms  %.1f M units/s  %.2f G permutations/s" % (ms, n / ms / 1e3, n * 9 / ms / 1e6), flush=True)
