#!/usr/bin/env python3
"""sha3_encrypt D512 over n x LEN uniform messages around the one-wave-per-SIMD boundary of the fused four-lane kernel
(16 384 items): seconds per call with the time-sliced launches (default) and without (CAPY_DEBUG=fused_slices=0: run the script
twice).  usage: LEN=5242880 python3 tools/sweep_fused_slices.py   -> profiles/r04_fused_slices.txt"""
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
ln = int(os.environ.get("LEN", str(5 << 20)))
stride = ln + 128
print("# %s, %d-byte messages: n | encrypt s | decrypt s | GiB/s (encrypt) | round trip" % (os.environ.get("CAPY_DEBUG", "default"), ln))
for n in [int(x) for x in os.environ.get("NS", "15360,16384,16400,17408,18432,20480,22528,23552,24576,28672,32768,33024,36864,40960,43008,45056,49152,49408,53248,57344,61440,65536").split(",")]:
    if n * stride > 200 * (1 << 30):
        continue
    msgs = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(msgs.data_ptr(), n * stride, 7, sp))
    first = msgs[:ln].clone()
    pws = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    zs = torch.empty(n * 512, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(pws.data_ptr(), n * 64, 8, sp))
    _lib.check(lib.capy_fill_random_dev(zs.data_ptr(), n * 512, 9, sp))
    tags = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    times = []
    for _ in range(2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(st)
        _lib.check(lib.capy_sha3_encrypt_batch_dev(512, n, pws.data_ptr(), 64, None, n * 64, zs.data_ptr(), msgs.data_ptr(), None, ln, stride,
                                                  tags.data_ptr(), sp))
        e[1].record(st)
        _lib.check(lib.capy_sha3_decrypt_batch_dev(512, n, pws.data_ptr(), 64, None, n * 64, zs.data_ptr(), msgs.data_ptr(), None, ln, stride,
                                                  tags.data_ptr(), status.data_ptr(), sp))
        e[2].record(st)
        torch.cuda.synchronize()
        times.append((e[0].elapsed_time(e[1]) / 1e3, e[1].elapsed_time(e[2]) / 1e3))
    enc, dec = min(t[0] for t in times), min(t[1] for t in times)
    ok = bool((status == 0).all()) and torch.equal(msgs[:ln], first)
    print("%6d | %.4f | %.4f | %7.1f | %s" % (n, enc, dec, n * ln / enc / (1 << 30), ok), flush=True)
    del msgs
    torch.cuda.empty_cache()
