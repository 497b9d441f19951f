#!/usr/bin/env python3
"""Ragged DEVICE batches of short messages through capy_sha3_batch_dev (offsets on the device), against uniform batches
of the same total size.  MODE=ragged|uniform|both, N (default 2^21), MAXLEN (default 2048).
Run on the GPU box: python tools/bench_ragged_dev.py ; under rocprofv3 --pmc for counters."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
sp = C.c_void_p(st.cuda_stream)
n = int(os.environ.get("N", str(1 << 21)))
maxlen = int(os.environ.get("MAXLEN", "2048"))
mode = os.environ.get("MODE", "both")
reps = int(os.environ.get("REPS", "5"))
_lib.check(lib.capy_set_sponge_lanes(int(os.environ.get("LANES", "0")) | (int(os.environ.get("DBG", "0")) << 8)))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rng = np.random.default_rng(0xCA9C0007)
if mode in ("ragged", "both"):
    lens = rng.integers(0, maxlen + 1, n).astype(np.uint64)
    starts = np.zeros(n + 1, dtype=np.uint64)
    starts[1:] = np.cumsum((lens + 7) // 8 * 8)  # 8-byte aligned starts
    total = int(starts[-1])
    # exact lengths: offsets[i+1] - offsets[i] must be the length, so pack tightly but keep starts aligned by padding
    # lengths up -- the library takes lengths from the offsets; use aligned lengths here (what matters is raggedness)
    lens_al = (lens + 7) // 8 * 8
    buf = torch.empty(total + 8, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(buf.data_ptr(), (total + 8) // 8 * 8, 77, sp))
    offs = torch.from_numpy(starts.astype(np.int64)).to(dev)
    dig = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    ms = timed(lambda: _lib.check(lib.capy_sha3_batch_dev(256, n, buf.data_ptr(), offs.data_ptr(), 0, 0, dig.data_ptr(), sp)))
    hb, hd = buf.cpu().numpy(), dig.cpu().numpy()
    for i in (0, 1, n // 2, n - 1):
        m = hb[int(starts[i]):int(starts[i + 1])].tobytes()
        assert hd[32 * i:32 * i + 32].tobytes() == hashlib.sha3_256(m).digest(), i
    print("ragged  n=%d 0..%d B (8-byte multiples): %.3f ms  %.1f GB/s" % (n, maxlen, ms, total / ms / 1e6), flush=True)
    del buf
if mode in ("uniform", "both"):
    L = maxlen // 2
    buf = torch.empty(n * L, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(buf.data_ptr(), n * L, 78, sp))
    dig = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    ms = timed(lambda: _lib.check(lib.capy_sha3_batch_dev(256, n, buf.data_ptr(), None, L, L, dig.data_ptr(), sp)))
    print("uniform n=%d x %d B: %.3f ms  %.1f GB/s" % (n, L, ms, n * L / ms / 1e6), flush=True)
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    ms = timed(lambda: _lib.check(lib.capy_sha3_batch_dev(256, n, buf.data_ptr(), offs.data_ptr(), 0, 0, dig.data_ptr(), sp)))
    print("equal lengths through offsets n=%d x %d B: %.3f ms  %.1f GB/s" % (n, L, ms, n * L / ms / 1e6), flush=True)
