#!/usr/bin/env python3
"""r06 probe: does vb2_kernel (BASELINE config 4: 2^18 pairs = exactly two waves per SIMD, all starting together and running the
same instruction stream in lock-step) gain from DE-PHASING the two waves of a SIMD?  2^20 pairs run 4-7 % faster per pair than 2^18
(later rounds of waves start staggered), which suggests it.  No source change: CAPY_DEBUG=ed448_pair=1 makes 2^17 pairs take
vb2_kernel too (1024 waves, one per SIMD); two such launches on two streams, the second delayed by torch.cuda._sleep, put two waves
on every SIMD with a chosen phase offset.   usage: CAPY_DEBUG=ed448_pair=1 python3 tools/probe_vb2_dephase.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
nmax = 1 << 20
st = torch.cuda.current_stream()
sp = C.c_void_p(st.cuda_stream)
sc = torch.empty(nmax * 56, dtype=torch.uint8, device=dev)
tsc = torch.empty(nmax * 56, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_fill_random_dev(sc.data_ptr(), nmax * 56, 4, sp))
_lib.check(lib.capy_fill_random_dev(tsc.data_ptr(), nmax * 56, 41, sp))
pts = torch.empty(nmax * 112, dtype=torch.uint8, device=dev)
out = torch.empty(nmax * 112, dtype=torch.uint8, device=dev)
ref = torch.empty(nmax * 112, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_ed448_basemul_batch_dev(nmax, tsc.data_ptr(), pts.data_ptr(), sp))
torch.cuda.synchronize()


def vb(first, n, stream, dst=out):
    _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr() + first * 56, pts.data_ptr() + first * 112, dst.data_ptr() + first * 112,
                                                  C.c_void_p(stream.cuda_stream)))


def wall(fn, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


print("# CAPY_DEBUG=%s" % os.environ.get("CAPY_DEBUG", ""))
print("# (a) one launch: n | ms | M/s")
for n in (1 << 17, 1 << 18, 3 << 17, 1 << 19, 1 << 20):
    vb(0, n, st, ref)
    t = wall(lambda: vb(0, n, st))
    print("%8d | %7.3f | %6.2f" % (n, t * 1e3, n / t / 1e6), flush=True)

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
clk = 2.3e9
print("# (b) 2^18 pairs as two launches of 2^17 (1024 waves each) on two streams, the second delayed: delay ms | total ms | M/s | outputs equal the single launch")
n = 1 << 18
for delay_ms in (0.0, 0.02, 0.1, 0.3, 1.0, 2.0, 4.0):
    def run():
        vb(0, n // 2, s1)
        with torch.cuda.stream(s2):
            if delay_ms:
                torch.cuda._sleep(int(delay_ms * 1e-3 * clk))
        vb(n // 2, n // 2, s2)
    out.zero_()
    t = wall(run)
    ok = torch.equal(out[:n * 112], ref[:n * 112])
    print("%5.2f | %7.3f | %6.2f | %s" % (delay_ms, t * 1e3, n / t / 1e6, ok), flush=True)
