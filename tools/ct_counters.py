#!/usr/bin/env python3
"""Workload of the constant-address counter test (tests/test_constant_address.py, profiles/r04_constant_address_counters.txt):
the Ed448 multiplications by secret scalars at N items on scalar POPULATIONS that drive an indexed table lookup to its
extremes -- all scalars zero (every lane reads the same table row), every digit at its maximum, uniformly random -- so that
the memory-side counters of a run can be compared across populations.

    MODE=1 N=65536 rocprofv3 --pmc <counters> --output-format csv -d <dir> -o pmc -- python3 tools/ct_counters.py

MODE is the CAPY_HARDEN_* value (1: constant-address lookups for every multiplication; 0: indexed lookups).  Segments are
separated by a marker dispatch (fill_random_kernel on 8 bytes); the manifest printed at the end names them in order."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = int(os.environ.get("N", "65536"))
mode = int(os.environ.get("MODE", "1"))
_lib.check(lib.capy_ed448_set_hardened(mode))
_lib.check(lib.capy_ed448_set_wave_max(int(os.environ.get("WAVE_MAX", "0"))))  # 0: lane-per-item kernels at this size

marker = torch.zeros(8, dtype=torch.uint8, device=dev)
rand = torch.empty(n * 56, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_fill_random_dev(rand.data_ptr(), n * 56, 99, sp))
pops = {"zero": torch.zeros(n * 56, dtype=torch.uint8, device=dev),
        "ones": torch.full((n * 56,), 0xFF, dtype=torch.uint8, device=dev),
        "random": rand}
tsc = torch.empty(n * 56, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_fill_random_dev(tsc.data_ptr(), n * 56, 98, sp))
pts = torch.empty(n * 112, dtype=torch.uint8, device=dev)
out = torch.empty(n * 112, dtype=torch.uint8, device=dev)
# points and fixed-base tables are prepared BEFORE the measured segments (table builds are public data)
_lib.check(lib.capy_ed448_set_hardened(0))
_lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp))
_lib.check(lib.capy_ed448_set_hardened(mode))
_lib.check(lib.capy_ed448_basemul_batch_dev(n, rand.data_ptr(), out.data_ptr(), sp))
_lib.check(lib.capy_ed448_scalarmul_batch_dev(n, rand.data_ptr(), pts.data_ptr(), out.data_ptr(), sp))
torch.cuda.synchronize()

manifest = []


def mark(label):
    _lib.check(lib.capy_fill_random_dev(marker.data_ptr(), 8, 1, sp))
    manifest.append(label)


# password populations for the protocol call: one password for everybody (-> one secret scalar: every lane the same
# digits) against random passwords
pw_same = torch.full((n * 32,), 0x5A, dtype=torch.uint8, device=dev)
pw_rand = torch.empty(n * 32, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_fill_random_dev(pw_rand.data_ptr(), n * 32, 97, sp))
for name, sc in pops.items():
    mark("fixed_base/" + name)
    _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), out.data_ptr(), sp))
    torch.cuda.synchronize()
for name, sc in pops.items():
    mark("variable_base/" + name)
    _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), out.data_ptr(), sp))
    torch.cuda.synchronize()
for name, pw in (("same_password", pw_same), ("random_passwords", pw_rand)):
    mark("keypair/" + name)
    _lib.check(lib.capy_keypair_batch_dev(512, n, pw.data_ptr(), 32, None, out.data_ptr(), sp))
    torch.cuda.synchronize()
mark("end")
torch.cuda.synchronize()
print("CT_MANIFEST " + json.dumps({"mode": mode, "n": n, "segments": manifest}))
