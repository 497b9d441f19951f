#!/usr/bin/env python3
"""Quick GPU check of the one-lane-per-sponge fused sha3_encrypt / sha3_decrypt kernel (csrc/sponge_fused1.h) against the
two-pass form and the oracle.  Run with CAPY_DEBUG=fused1_min=1[,fused1_form=F] so that small batches take the new kernel."""
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402
from oracle import oracle as O  # noqa: E402

lib = _lib.lib()
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = random.Random(5)


def rand(nbytes, seed):
    t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, sp))
    return t


def last_kind():
    k, l = C.c_int(0), C.c_int(0)
    lib.capy_debug_last_sponge_kernel(C.byref(k), C.byref(l))
    return k.value, l.value


bad = 0
cases = []
for d, rb in ((512, 136), (256, 168), (384, 152)):
    for n in (1, 31, 33, 100):
        for ln in (0, 5, rb - 1, rb, rb + 8, 3 * rb + 77, 16 * rb, 17 * rb + 131, 40 * rb + 8):
            for pad in (8, 16, 24, 128 + 8 * (ln % 16)):
                cases.append((d, n, ln, (ln + 7) // 8 * 8 + pad))
rng.shuffle(cases)
cases = cases[: int(os.environ.get("CASES", "160"))]
for d, n, ln, stride in cases:
    pl = 32
    pws, zs, plain = rand(n * pl, 1 + n), rand(n * 512, 2 + n), rand(n * stride + 256, 3 + n + ln)
    off = rng.choice((0, 8, 16, 40, 64, 120))  # the first message's position inside its 128-byte line
    res = {}
    for name, lanes in (("new", 0), ("two-pass", 1 | (1 << 16))):
        _lib.check(lib.capy_set_sponge_lanes(lanes))
        m = plain.clone()
        tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_sha3_encrypt_batch_dev(d, n, pws.data_ptr(), pl, None, n * pl, zs.data_ptr(), m.data_ptr() + off, None, ln, stride,
                                                  tags.data_ptr(), sp))
        torch.cuda.synchronize()
        res[name] = (m, tags, last_kind())
    _lib.check(lib.capy_set_sponge_lanes(0))
    ok = torch.equal(res["new"][0], res["two-pass"][0]) and torch.equal(res["new"][1], res["two-pass"][1])
    kind = res["new"][2]
    m, tags, _ = res["new"]
    for i in {0, n - 1, rng.randrange(n)}:
        want = O.sha3_encrypt(bytes(pws[i * pl:(i + 1) * pl].cpu().numpy()), bytes(zs[i * 512:(i + 1) * 512].cpu().numpy()),
                              bytes(plain[off + i * stride:off + i * stride + ln].cpu().numpy()), d)
        got = (bytes(m[off + i * stride:off + i * stride + ln].cpu().numpy()), bytes(tags[64 * i:64 * i + 64].cpu().numpy()))
        ok = ok and got == want
    # decrypt, one forged tag
    status = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    f = rng.randrange(n)
    tags[64 * f] ^= 1
    ct = m.clone()
    _lib.check(lib.capy_sha3_decrypt_batch_dev(d, n, pws.data_ptr(), pl, None, n * pl, zs.data_ptr(), m.data_ptr() + off, None, ln, stride,
                                              tags.data_ptr(), status.data_ptr(), sp))
    torch.cuda.synchronize()
    kd = last_kind()
    want = plain.clone()
    want[off + f * stride:off + f * stride + ln] = ct[off + f * stride:off + f * stride + ln]
    ok = ok and int(status[f]) == 1 and int((status != 0).sum()) == 1 and torch.equal(m, want)
    if not ok or kind[0] != 23:
        bad += 1
        print("FAIL" if not ok else "KIND", d, n, ln, stride, off, kind, kd, flush=True)
print("cases", len(cases), "bad", bad, "CAPY_DEBUG", os.environ.get("CAPY_DEBUG"))
sys.exit(1 if bad else 0)
