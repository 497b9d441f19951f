#!/usr/bin/env python3
"""Soak: one-item-per-wave Ed448 kernels against the one-item-per-lane kernels on many random (scalar, point) pairs,
byte for byte (variable base and fixed base, device entry points)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = int(os.environ.get("N", "65536"))
bad = 0
for seed in range(int(os.environ.get("SEEDS", "3"))):
    def rand(nbytes, s):
        t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), 1000 * seed + s, sp))
        return t

    tsc, sc = rand(n * 56, 1), rand(n * 56, 2)
    if seed == 1:  # sparse and dense scalars: long runs of zero / one bits
        sc = sc & rand(n * 56, 3) & rand(n * 56, 4)
    if seed == 2:
        sc = sc | rand(n * 56, 3) | rand(n * 56, 4)
    pts = torch.empty(n * 112, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_ed448_set_wave_max(0))
    _lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp))
    outs = []
    for wmax in (0, 1 << 30):
        _lib.check(lib.capy_ed448_set_wave_max(wmax))
        a = torch.zeros(n * 112, dtype=torch.uint8, device=dev)
        b = torch.zeros(n * 112, dtype=torch.uint8, device=dev)
        _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), a.data_ptr(), sp))
        _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), b.data_ptr(), sp))
        torch.cuda.synchronize()
        outs.append((a, b))
    ev = bool(torch.equal(outs[0][0], outs[1][0]))
    ef = bool(torch.equal(outs[0][1], outs[1][1]))
    print("seed %d: %d pairs, variable base equal=%s, fixed base equal=%s" % (seed, n, ev, ef), flush=True)
    bad += (not ev) + (not ef)
_lib.check(lib.capy_ed448_set_wave_max(-1))
sys.exit(1 if bad else 0)
