#!/usr/bin/env python3
"""A/B of the fused sha3_encrypt / sha3_decrypt kernel: per-lane direct loads and stores (default) against the LDS-staged
round-1 form (debug bit 6).  python tools/ab_fused_direct.py > gpurun_out/fused_direct_ab.txt"""
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
MIB5 = 5242880


def rand(nbytes, seed):
    t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, sp))
    return t


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3


shapes = ((2048, MIB5), (16384, MIB5), (16384, 1 << 20))
if len(sys.argv) > 1:  # e.g. 32768x1048576,262144x65536 (two-pass path: keystream XOR kernel + tag kernel)
    shapes = tuple(tuple(int(v) for v in x.split("x")) for x in sys.argv[1].split(","))
for nmsg, mlen in shapes:
    msgs = rand(nmsg * mlen, 3)
    pws = rand(nmsg * 64, 31)
    zs = rand(nmsg * 512, 32)
    tags = torch.empty(nmsg * 64, dtype=torch.uint8, device=dev)
    status = torch.empty(nmsg, dtype=torch.int32, device=dev)
    before = msgs[:8192].clone()

    def enc():
        _lib.check(lib.capy_sha3_encrypt_batch_dev(512, nmsg, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None,
                                                   mlen, mlen, tags.data_ptr(), sp))

    def dec():
        _lib.check(lib.capy_sha3_decrypt_batch_dev(512, nmsg, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None,
                                                   mlen, mlen, tags.data_ptr(), status.data_ptr(), sp))

    enc(), dec()  # warm
    for rep in range(2):
        for dbg, name in ((0, "direct"), (64, "staged")):
            _lib.check(lib.capy_set_sponge_lanes(dbg << 8))
            te, td = timed(enc), timed(dec)
            ok = bool((status == 0).all().item()) and bool((msgs[:8192] == before).all().item())
            print("%6d x %7d B  %s  enc %.4f s (%.1f GiB/s)  dec %.4f s (%.1f GiB/s)  roundtrip_ok=%s" % (
                nmsg, mlen, name, te, nmsg * mlen / te / 2**30, td, nmsg * mlen / td / 2**30, ok), flush=True)
    _lib.check(lib.capy_set_sponge_lanes(0))
    del msgs
