#!/usr/bin/env python3
"""Cost of the constant-address table lookups (capy_ed448_set_hardened): variable-base / fixed-base multiplication, key
pair, sign from one item to 2^18, mode 0 (indexed) vs mode 3 (constant-address everywhere; mode 1, the default, covers
key pair and sign).  Run on the GPU box: python tools/bench_hardened.py > gpurun_out/r03_ed448_hardened.txt"""
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


def rand(nbytes, seed):
    t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, sp))
    return t


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("# ms per call, indexed (mode 0) | constant-address (mode 3) | ratio   (1 KiB messages for sign)")
for n in (1, 64, 2048, 1 << 14, 1 << 16, 1 << 18):
    sc, tsc = rand(n * 56, 1), rand(n * 56, 2)
    pts = torch.empty(n * 112, dtype=torch.uint8, device=dev)
    out = torch.empty(n * 112, dtype=torch.uint8, device=dev)
    pws, msgs = rand(n * 64, 3), rand(n * 1024, 4)
    h, z = torch.empty(n * 56, dtype=torch.uint8, device=dev), torch.empty(n * 56, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp))
    ops = {
        "variable-base": lambda: _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), out.data_ptr(), sp)),
        "fixed-base": lambda: _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), out.data_ptr(), sp)),
        "keypair": lambda: _lib.check(lib.capy_keypair_batch_dev(512, n, pws.data_ptr(), 64, None, out.data_ptr(), sp)),
        "sign": lambda: _lib.check(lib.capy_schnorr_sign_batch_dev(512, n, pws.data_ptr(), 64, None, msgs.data_ptr(), None, 1024, 1024,
                                                                   h.data_ptr(), z.data_ptr(), sp)),
    }
    for name, fn in ops.items():
        _lib.check(lib.capy_ed448_set_hardened(0))
        a = timed(fn)
        ref = (out.clone(), h.clone(), z.clone())
        _lib.check(lib.capy_ed448_set_hardened(1))  # CAPY_HARDEN_ALL
        b = timed(fn)
        same = torch.equal(ref[0], out) and torch.equal(ref[1], h) and torch.equal(ref[2], z)
        _lib.check(lib.capy_ed448_set_hardened(4))  # CAPY_HARDEN_PROTOCOL
        print("n=%7d %-14s %9.3f | %9.3f | %5.2fx  %s" % (n, name, a, b, b / a, "identical" if same else "MISMATCH"), flush=True)
