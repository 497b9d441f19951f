#!/usr/bin/env python3
"""sha3_encrypt / sha3_decrypt D512 over n x LEN uniform device-resident messages: seconds per call, GiB/s, kernel kind chosen
(capy_debug_last_sponge_kernel) and a round-trip check.  usage: NS=65536,98304 LEN=1048576 python3 tools/sweep_fused1.py"""
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
ln = int(os.environ.get("LEN", str(1 << 20)))
d = int(os.environ.get("D", "512"))
stride = ln + 128
reps = int(os.environ.get("REPS", "3"))
print("# CAPY_DEBUG=%s, D%d, %d-byte messages: n | encrypt s | decrypt s | GiB/s enc | GiB/s dec | kind,launches | round trip"
      % (os.environ.get("CAPY_DEBUG", "default"), d, ln), flush=True)
for n in [int(x) for x in os.environ.get("NS", "32768,49152,65536,98304,131072").split(",")]:
    if n * stride > 230 * (1 << 30):
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
    for _ in range(reps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(st)
        _lib.check(lib.capy_sha3_encrypt_batch_dev(d, n, pws.data_ptr(), 64, None, n * 64, zs.data_ptr(), msgs.data_ptr(), None, ln, stride,
                                                  tags.data_ptr(), sp))
        e[1].record(st)
        _lib.check(lib.capy_sha3_decrypt_batch_dev(d, n, pws.data_ptr(), 64, None, n * 64, zs.data_ptr(), msgs.data_ptr(), None, ln, stride,
                                                  tags.data_ptr(), status.data_ptr(), sp))
        e[2].record(st)
        torch.cuda.synchronize()
        times.append((e[0].elapsed_time(e[1]) / 1e3, e[1].elapsed_time(e[2]) / 1e3))
    k, l = C.c_int(0), C.c_int(0)
    lib.capy_debug_last_sponge_kernel(C.byref(k), C.byref(l))
    enc, dec = min(t[0] for t in times), min(t[1] for t in times)
    ok = bool((status == 0).all()) and torch.equal(msgs[:ln], first)
    print("%6d | %.4f | %.4f | %7.1f | %7.1f | %d,%d | %s" % (n, enc, dec, n * ln / enc / (1 << 30), n * ln / dec / (1 << 30), k.value, l.value, ok), flush=True)
    del msgs
    torch.cuda.empty_cache()
