#!/usr/bin/env python3
"""sha3_encrypt D512 over n x 5 MiB messages, n small: the four-lanes-per-item kernel (sponge_fused.h) against the
wave-per-item kernel (sponge_wide_il.h: two waves per item; sponge_wide.h until r04).  CAPY_DEBUG=wide_max is read once per process, so the debug bits of
capy_set_sponge_lanes select the kernel here (bit 4: never wide, bit 5: always wide).
Run on the GPU box: python tools/bench_wide.py > gpurun_out/r02_wide_vs_fused.txt"""
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
L = int(os.environ.get("MSG", str(5242880)))
CHAIN_NS_PER_INST = 4.04 / 2.38  # two-lane chain: 120 VALU per round, one instruction per 4.04 cycles at 2.38 GHz


def rand(nbytes, seed):
    t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, sp))
    return t


print("# sha3_encrypt D512, n x %d bytes; seconds per call (second of two calls)" % L)
print("%8s %12s %12s %8s %14s %14s" % ("n", "fused s", "wide s", "ratio", "wide GiB/s", "round trip"))
for n in [int(x) for x in os.environ.get("N_LIST", "32,128,256,512,1024,2048").split(",")]:
    msgs = rand(n * L, 3)
    plain = msgs.clone()
    pws, zs = rand(n * 64, 31), rand(n * 512, 32)
    tags = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    res = {}
    for name, dbg in (("fused", 16), ("wide", 32)):
        _lib.check(lib.capy_set_sponge_lanes(dbg << 8))
        best = None
        for rep in range(2):
            msgs.copy_(plain)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            _lib.check(lib.capy_sha3_encrypt_batch_dev(512, n, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(),
                                                       None, L, L, tags.data_ptr(), sp))
            e1.record(st)
            torch.cuda.synchronize()
            best = e0.elapsed_time(e1) * 1e-3
        res[name] = (best, msgs.clone(), tags.clone())
    same = torch.equal(res["fused"][1], res["wide"][1]) and torch.equal(res["fused"][2], res["wide"][2])
    _lib.check(lib.capy_sha3_decrypt_batch_dev(512, n, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None, L, L,
                                               tags.data_ptr(), status.data_ptr(), sp))
    torch.cuda.synchronize()
    ok = same and bool((status == 0).all().item()) and torch.equal(msgs, plain)
    f, w = res["fused"][0], res["wide"][0]
    print("%8d %12.4f %12.4f %8.2f %14.2f %14s" % (n, f, w, f / w, n * L / w / 2**30, "ok" if ok else "MISMATCH"), flush=True)
    del msgs, plain
_lib.check(lib.capy_set_sponge_lanes(0))
