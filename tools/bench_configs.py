#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs 2-5 and the PCIe-inclusive host-buffer rate, one JSON object per line.
Configs 2, 3 and 5 are the functions of bench.py (the driver-run line carries them as `configs`); this tool adds more batch
sizes for config 3, config 4 beside its fixed-base rate, indexed-lookup rates for config 5 and the PCIe leg.
Run on the GPU box: python tools/bench_configs.py > gpurun_out/configs.jsonl ; summaries go to profiles/."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
cx = bench.Ctx(lib, _lib, torch, dev, st)
sp = cx.sp


def emit(**kw):
    print(json.dumps(kw), flush=True)


samples = {}
r, samples[2] = bench.config2(cx)
emit(config=2, **r)
r, samples[3] = bench.config3(cx, saturating=((2048, bench.MSG_BYTES), (16384, bench.MSG_BYTES), (32768, bench.MSG_BYTES), (49152, 4 << 20),
                                              (65536, 1 << 20), (98304, 1 << 20), (131072, 1 << 20), (262144, 1 << 16)))
emit(config=3, **r)

# ---- config 4: Ed448 variable-base / fixed-base, 2^18 pairs
n = 1 << 18
sc, tsc = cx.rand(n * 56, 4), cx.rand(n * 56, 41)
pts = torch.empty(n * 112, dtype=torch.uint8, device=dev)
o = torch.empty(n * 112, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp))
s_fb = cx.timed(lambda: _lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), o.data_ptr(), sp)), 3)
s_vb = cx.timed(lambda: _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), o.data_ptr(), sp)), 3)
emit(config=4, what="Ed448 2^18 pairs", var_base_per_s=n / s_vb, fixed_base_per_s=n / s_fb, var_base_ms=s_vb * 1e3, fixed_base_ms=s_fb * 1e3)

r, samples[5] = bench.config5(cx)
# the same with indexed table lookups for the secret scalars too (capy_ed448_set_hardened(0): r02's default)
_lib.check(lib.capy_ed448_set_hardened(0))
try:
    r_idx, _ = bench.config5(cx)
finally:
    _lib.check(lib.capy_ed448_set_hardened(4))  # CAPY_HARDEN_PROTOCOL, the default
emit(config=5, indexed_lookups_per_s={"keypair": r_idx["keypair_per_s"], "sign": r_idx["sign_per_s"]}, **r)
emit(config="oracle_spot_checks", **bench.check_config_samples(samples))

# ---- PCIe-inclusive SHA3-256 rate through the host-pointer entry point: 65536 x 64 KiB = 4 GiB of pageable host
# memory (the kernel itself takes ~4 ms at this shape, so this measures the staging path).  Cold = first call on a
# freshly written buffer (the library registers the range, then copies); warm = same buffer again.
nmsg, mlen = 65536, 65536
host = (C.c_uint8 * (nmsg * mlen))()
C.memset(host, 0x5A, nmsg * mlen)
offs = (C.c_uint64 * (nmsg + 1))(*[i * mlen for i in range(nmsg + 1)])
dig = (C.c_uint8 * (nmsg * 32))()
t0 = time.perf_counter()
_lib.check(lib.capy_sha3_batch(256, nmsg, host, offs, dig))
tc = time.perf_counter() - t0
t0 = time.perf_counter()
_lib.check(lib.capy_sha3_batch(256, nmsg, host, offs, dig))
th = time.perf_counter() - t0
import hashlib  # noqa: E402

assert bytes(dig[:32]) == hashlib.sha3_256(bytes(host[:mlen])).digest()
emit(config="pcie", what="capy_sha3_batch from pageable host memory, 65536 x 64 KiB (H2D copy + kernel + D2H)",
     cold_seconds=tc, cold_GiBps=nmsg * mlen / tc / 2**30, warm_seconds=th, warm_GiBps=nmsg * mlen / th / 2**30)
