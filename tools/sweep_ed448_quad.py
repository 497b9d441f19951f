#!/usr/bin/env python3
"""Variable-base multiplication and the verify-shaped double multiplication over batch sizes in and around the regime of
the four-lanes-per-item (csrc/ed448_quad.h) and two-lanes-per-item (csrc/ed448_duo.h) kernels: both families off
(capy_ed448_set_{quad,duo}_range(0, 0)) against each forced on (0, 2^30): ms per call, and whether the outputs are
byte-identical.
usage: python3 tools/sweep_ed448_quad.py [n ...]   -> profiles/r04_ed448_quad.txt, profiles/r04_ed448_duo.txt"""
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
ns = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 40960, 49152, 65536]
nmax = max(ns)
sc, asc, tsc = (torch.empty(nmax * 56, dtype=torch.uint8, device=dev) for _ in range(3))
for t, seed in ((sc, 4), (asc, 5), (tsc, 41)):
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


print("#      n | variable base: other ms   quad ms    duo ms | [a]G + [b]P: other ms   quad ms    duo ms | identical | default family: vb ms  dsm ms")
FAM = (("other", (0, 0), (0, 0)), ("quad", (0, 1 << 30), (0, 0)), ("duo", (0, 0), (0, 1 << 30)), ("default", (-1, -1), (-1, -1)))
for n in ns:
    res = {}
    for name, qr, dr in FAM:
        _lib.check(lib.capy_ed448_set_quad_range(*qr))
        _lib.check(lib.capy_ed448_set_duo_range(*dr))
        vb = torch.zeros(n * 112, dtype=torch.uint8, device=dev)
        ds = torch.zeros(n * 112, dtype=torch.uint8, device=dev)
        t_vb = timed(lambda: _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), vb.data_ptr(), sp)))
        t_ds = timed(lambda: _lib.check(lib.capy_ed448_double_scalarmul_batch_dev(n, asc.data_ptr(), sc.data_ptr(), pts.data_ptr(),
                                                                                  ds.data_ptr(), sp)))
        res[name] = (t_vb, t_ds, vb, ds)
    same = all(torch.equal(res["other"][j], res[f][j]) for f in ("quad", "duo", "default") for j in (2, 3))
    o, q, d, df = res["other"], res["quad"], res["duo"], res["default"]
    print("%8d | %21.3f %9.3f %9.3f | %20.3f %9.3f %9.3f | %9s | %21.3f %7.3f"
          % (n, o[0], q[0], d[0], o[1], q[1], d[1], same, df[0], df[1]), flush=True)
_lib.check(lib.capy_ed448_set_quad_range(-1, -1))
_lib.check(lib.capy_ed448_set_duo_range(-1, -1))
