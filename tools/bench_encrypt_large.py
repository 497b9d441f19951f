#!/usr/bin/env python3
"""sha3_encrypt D512 over large batches of 5 MiB / 1 MiB messages: the fused four-lane kernel (n <= 16384) against the
two-pass form (tag kernel + keystream kernel) that larger batches take.  python tools/bench_encrypt_large.py"""
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


for n, L in [tuple(int(v) for v in x.split("x")) for x in os.environ.get("CFG", "16384x5242880,32768x5242880,49152x4194304,32768x1048576,49152x1048576,65536x1048576,262144x65536").split(",")]:
    msgs = rand(n * L, 3)
    pws, zs = rand(n * 64, 31), rand(n * 512, 32)
    tags = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    head = msgs[:4096].clone()

    def enc():
        _lib.check(lib.capy_sha3_encrypt_batch_dev(512, n, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None, L, L,
                                                   tags.data_ptr(), sp))

    def dec():
        _lib.check(lib.capy_sha3_decrypt_batch_dev(512, n, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None, L, L,
                                                   tags.data_ptr(), status.data_ptr(), sp))

    enc(); dec(); torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(st); enc(); e1.record(st); dec(); e2.record(st); torch.cuda.synchronize()
    ok = bool((status == 0).all().item()) and bool(torch.equal(msgs[:4096], head))
    te, td = e0.elapsed_time(e1) * 1e-3, e1.elapsed_time(e2) * 1e-3
    print("sha3_encrypt D512 %7d x %8d B: enc %8.4f s %7.1f GiB/s (%.0f GB/s algorithmic at 2 x len)   dec %8.4f s %7.1f GiB/s  ok=%s" % (
        n, L, te, n * L / te / 2**30, 2 * n * L / te / 1e9, td, n * L / td / 2**30, ok), flush=True)
    del msgs
