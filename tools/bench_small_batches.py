#!/usr/bin/env python3
"""Latency of very small batches of long messages (what the reference's own benches and tests do: ONE 5 MiB message,
benches/benchmark_sha3.rs, benchmark_e448_512.rs, tests/integration_tests.rs): SHA3-256, KMACXOF256 tag, Schnorr
sign + verify for n = 1 .. 2048 messages of 5 MiB, device buffers, with the wave-per-item kernels (sponge_wide_il.h; sponge_wide.h until r04) off
and on.  Run on the GPU box: python tools/bench_small_batches.py > gpurun_out/r02_small_batches.txt"""
import ctypes as C
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
sp = C.c_void_p(st.cuda_stream)
L = 5242880


def rand(nbytes, seed):
    t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, sp))
    return t


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3


print("# n x 5 MiB messages, seconds per call: two-lane kernels (wave-per-item off) | wave-per-item kernels on | ratio")
print("%6s | %9s %9s %6s | %9s %9s %6s | %9s %9s %6s | %9s %9s %6s" % ("n", "sha3 off", "on", "x", "kmac off", "on", "x",
                                                                          "sign off", "on", "x", "verify off", "on", "x"))
# N_LIST overrides the batch sizes; FORCE_WIDE=1 forces the wave-per-item kernels (debug bit 5, n <= 4096) to find the crossover
N_LIST = [int(x) for x in os.environ.get("N_LIST", "1,2,16,128,512,1024,2048").split(",")]
ON_BITS = 32 if os.environ.get("FORCE_WIDE") else 0
for n in N_LIST:
    msgs = rand(n * L, 7)
    keys, pws = rand(n * 64, 8), rand(n * 32, 9)
    dig = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    tag = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    pubs = torch.zeros(n * 112, dtype=torch.uint8, device=dev)
    h = torch.zeros(n * 56, dtype=torch.uint8, device=dev)
    z = torch.zeros(n * 56, dtype=torch.uint8, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    _lib.check(lib.capy_keypair_batch_dev(512, n, pws.data_ptr(), 32, None, pubs.data_ptr(), sp))
    ops = {
        "sha3": lambda: _lib.check(lib.capy_sha3_batch_dev(256, n, msgs.data_ptr(), None, L, L, dig.data_ptr(), sp)),
        "kmac": lambda: _lib.check(lib.capy_kmac_xof_batch_dev(512, n, keys.data_ptr(), 64, 64, None, msgs.data_ptr(), None, L, L,
                                                               512, b"T", 1, tag.data_ptr(), 64, sp)),
        "sign": lambda: _lib.check(lib.capy_schnorr_sign_batch_dev(512, n, pws.data_ptr(), 32, None, msgs.data_ptr(), None, L, L,
                                                                   h.data_ptr(), z.data_ptr(), sp)),
        "verify": lambda: _lib.check(lib.capy_schnorr_verify_batch_dev(512, n, pubs.data_ptr(), msgs.data_ptr(), None, L, L,
                                                                       h.data_ptr(), z.data_ptr(), status.data_ptr(), sp)),
    }
    res, outs = {}, {}
    for name, dbg in (("off", 16), ("on", ON_BITS)):
        _lib.check(lib.capy_set_sponge_lanes(dbg << 8))
        for k, fn in ops.items():
            res[(k, name)] = timed(fn)
        torch.cuda.synchronize()
        outs[name] = (dig.clone(), tag.clone(), h.clone(), z.clone(), status.clone())
    same = all(torch.equal(a, b) for a, b in zip(outs["off"], outs["on"])) and not bool(status.any().item())
    same = same and bytes(dig[:32].cpu().numpy()) == hashlib.sha3_256(bytes(msgs[:L].cpu().numpy())).digest()
    row = "%6d" % n
    for k in ("sha3", "kmac", "sign", "verify"):
        a, b = res[(k, "off")], res[(k, "on")]
        row += " | %9.4f %9.4f %6.2f" % (a, b, a / b)
    print(row + ("   ok" if same else "   MISMATCH"), flush=True)
    del msgs
_lib.check(lib.capy_set_sponge_lanes(0))
