#!/usr/bin/env python3
"""Secret-scalar (constant-address) variable-base multiplication over batch sizes around the regime of vb_quad_ct_kernel
(csrc/ed448_quad.h: four lanes per item, window table in LDS): the one-item-per-lane / one-item-per-wave hardened kernels
(capy_ed448_set_quad_range(0, 0)) against the quad form forced on, CAPY_HARDEN_ALL; ms per call and byte identity.
usage: python3 tools/sweep_ed448_quad_ct.py [n ...]   -> profiles/r04_ed448_quad_ct.txt"""
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
ns = [int(a) for a in sys.argv[1:]] or [2048, 4096, 6144, 8192, 12288, 16384, 20480, 24576, 32768, 49152, 65536]
nmax = max(ns)
sc, tsc = (torch.empty(nmax * 56, dtype=torch.uint8, device=dev) for _ in range(2))
for t, seed in ((sc, 4), (tsc, 41)):
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), nmax * 56, seed, sp))
pts = torch.empty(nmax * 112, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_ed448_basemul_batch_dev(nmax, tsc.data_ptr(), pts.data_ptr(), sp))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


_lib.check(lib.capy_ed448_set_hardened(1))  # CAPY_HARDEN_ALL: the raw call below counts as secret
print("#      n | hardened variable base: lane / wave ms   quad-ct ms   speed-up | identical")
try:
    for n in ns:
        res = {}
        for name, qr in (("other", (0, 0)), ("quad", (0, 1 << 30))):
            _lib.check(lib.capy_ed448_set_quad_range(*qr))
            vb = torch.zeros(n * 112, dtype=torch.uint8, device=dev)
            t_vb = timed(lambda: _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), vb.data_ptr(), sp)))
            res[name] = (t_vb, vb)
        o, q = res["other"], res["quad"]
        print("%8d | %39.3f %12.3f %9.2fx | %s" % (n, o[0], q[0], o[0] / q[0], torch.equal(o[1], q[1])), flush=True)
finally:
    _lib.check(lib.capy_ed448_set_quad_range(-1, -1))
    _lib.check(lib.capy_ed448_set_hardened(4))  # CAPY_HARDEN_PROTOCOL
