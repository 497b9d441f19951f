#!/usr/bin/env python3
"""BASELINE config 2's unit (KMACXOF256(k_i, "", 8192 bits, "SKE"), 64-byte keys) over batch sizes: time per call against
waves per SIMD, to separate the steady-state rate from the fill / drain cost of a launch.
usage: python3 tools/sweep_config2.py [n ...]   (default: k x 65 536 for k = 1..8, 12, 16, 32, 64)"""
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
_lib.check(lib.capy_set_sponge_lanes(int(os.environ.get("LANES", "0"), 0)))
ns = [int(a) for a in sys.argv[1:]] or [k << 16 for k in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 32, 64)]
nmax = max(ns)
keys = torch.empty(nmax * 64, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_fill_random_dev(keys.data_ptr(), keys.numel(), 5, sp))
out = torch.empty(nmax * 1024, dtype=torch.uint8, device=dev)
for n in ns:
    def run():
        _lib.check(lib.capy_kmac_xof_batch_dev(512, n, keys.data_ptr(), 64, 64, None, None, None, 0, 0, 8192, b"SKE", 3, out.data_ptr(), 1024, sp))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            run()
        e1.record(st)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print("n = %8d (%5.1f waves per SIMD)  %.4f ms  %.1f M units/s  %.2f G device permutations/s" % (n, n / 65536, best, n / best / 1e3, 9 * n / best / 1e6))
