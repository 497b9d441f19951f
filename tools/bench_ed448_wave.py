#!/usr/bin/env python3
"""One item per wave (csrc/ed448_wave.h) against one item per lane, over batch size: variable base, fixed base and the
double-scalar multiplication of Signable::verify, device entry points, outputs compared byte for byte.
python tools/bench_ed448_wave.py > gpurun_out/ed448_wave.txt"""
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


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,16,256,1024,2048,4096,8192,16384".split(","))]
nmax = max(sizes)
tsc, sc, sc2 = rand(nmax * 56, 41), rand(nmax * 56, 4), rand(nmax * 56, 5)
pts = torch.empty(nmax * 112, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_ed448_set_wave_max(0))
_lib.check(lib.capy_ed448_basemul_batch_dev(nmax, tsc.data_ptr(), pts.data_ptr(), sp))  # distinct points [t_i]G
torch.cuda.synchronize()
print("# ms per call; lane = one item per lane (vb / fb / dsm kernels), wave = one item per wave; equal = outputs identical")
for n in sizes:
    row = "n=%6d" % n
    for name, call in (
        ("var-base", lambda o: lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), o.data_ptr(), sp)),
        ("fixed-base", lambda o: lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), o.data_ptr(), sp)),
    ):
        outs, ms = [], []
        for wmax in (0, 1 << 30):
            _lib.check(lib.capy_ed448_set_wave_max(wmax))
            o = torch.zeros(n * 112, dtype=torch.uint8, device=dev)
            ms.append(timed(lambda: _lib.check(call(o)), 3 if n <= 4096 else 2))
            outs.append(o)
        row += "  %s lane %7.3f wave %7.3f (%4.1fx) equal=%s" % (name, ms[0], ms[1], ms[0] / ms[1], bool(torch.equal(outs[0], outs[1])))
    print(row, flush=True)
_lib.check(lib.capy_ed448_set_wave_max(-1))
