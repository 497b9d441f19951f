#!/usr/bin/env python3
"""GPU check and timing of the bit-interleaved one-wave-per-sponge kernels (csrc/sponge_wide_il.h) against the two-lane digest
kernel, the four-lane fused encrypt kernel, the two-pass form and the oracle.

  python3 tools/check_wide_il.py            correctness: digests (sha3, kmac_xof), sha3_encrypt / sha3_decrypt, forged tags
  python3 tools/check_wide_il.py time       seconds per call: config 3 as specified and the reference's one-message shapes
"""
import ctypes as C
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = random.Random(7)
OLD = 2  # digests: the two-lane kernel
NOWIDE = 16 << 8  # debug bit 4: never a wave-per-item kernel


def rand(nbytes, seed):
    t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, sp))
    return t


def last_kind():
    k, l = C.c_int(0), C.c_int(0)
    lib.capy_debug_last_sponge_kernel(C.byref(k), C.byref(l))
    return k.value, l.value


def encrypt(d, n, pws, zs, m, off, ln, stride, tags):
    _lib.check(lib.capy_sha3_encrypt_batch_dev(d, n, pws.data_ptr(), 32, None, n * 32, zs.data_ptr(), m.data_ptr() + off, None, ln, stride,
                                              tags.data_ptr(), sp))


def decrypt(d, n, pws, zs, m, off, ln, stride, tags, status):
    _lib.check(lib.capy_sha3_decrypt_batch_dev(d, n, pws.data_ptr(), 32, None, n * 32, zs.data_ptr(), m.data_ptr() + off, None, ln, stride,
                                              tags.data_ptr(), status.data_ptr(), sp))


def check():
    from oracle import oracle as O

    bad = 0
    cases = []
    for d, rb in ((512, 136), (256, 168), (384, 152)):
        for n in (1, 2, 7, 100):
            for ln in (0, 1, 3, 4, 5, 8, rb - 4, rb - 3, rb - 1, rb, rb + 4, rb + 8, 2 * rb - 2, 3 * rb + 77, 16 * rb, 17 * rb + 131, 40 * rb + 12):
                cases.append((d, n, ln, (ln + 7) // 8 * 8 + rng.choice((8, 16, 24, 136))))
    rng.shuffle(cases)
    cases = cases[: int(os.environ.get("CASES", "200"))]
    # every SIMD busy, two workgroups per CU and more: the two waves of an item no longer run in step
    cases += [(512, 600, 64 * 136 + 20, 64 * 136 + 32), (256, 1024, 300 * 168, 300 * 168 + 8), (512, 3000, 20 * 136 + 9, 20 * 136 + 24)]
    for d, n, ln, stride in cases:
        pws, zs, plain = rand(n * 32, 1 + n), rand(n * 512, 2 + n), rand(n * stride + 256, 3 + n + ln)
        off = rng.choice((0, 8, 16, 40))
        res = {}
        force = (32 << 8) if n > 1024 else 0  # debug bit 5: the wave-per-item kernels for up to 4096 items
        for name, lanes in (("il", force), ("wide", NOWIDE), ("two-pass", 1 | (1 << 16))):
            _lib.check(lib.capy_set_sponge_lanes(lanes))
            m = plain.clone()
            tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
            encrypt(d, n, pws, zs, m, off, ln, stride, tags)
            torch.cuda.synchronize()
            res[name] = (m, tags, last_kind())
        _lib.check(lib.capy_set_sponge_lanes(0))
        ok = all(torch.equal(res["il"][k], res[o][k]) for o in ("wide", "two-pass") for k in (0, 1))
        m, tags, kind = res["il"]
        i = rng.randrange(n)
        want = O.sha3_encrypt(bytes(pws[i * 32:(i + 1) * 32].cpu().numpy()), bytes(zs[i * 512:(i + 1) * 512].cpu().numpy()),
                              bytes(plain[off + i * stride:off + i * stride + ln].cpu().numpy()), d)
        ok = ok and (bytes(m[off + i * stride:off + i * stride + ln].cpu().numpy()), bytes(tags[64 * i:64 * i + 64].cpu().numpy())) == want
        status = torch.full((n,), 9, dtype=torch.int32, device="cuda")
        f = rng.randrange(n)
        tags[64 * f] ^= 1
        ct = m.clone()
        _lib.check(lib.capy_set_sponge_lanes(force))
        decrypt(d, n, pws, zs, m, off, ln, stride, tags, status)
        torch.cuda.synchronize()
        kd = last_kind()
        _lib.check(lib.capy_set_sponge_lanes(0))
        want = plain.clone()
        want[off + f * stride:off + f * stride + ln] = ct[off + f * stride:off + f * stride + ln]
        ok = ok and int(status[f]) == 1 and int((status != 0).sum()) == 1 and torch.equal(m, want)
        if not ok or kind[0] != 27 or kd[0] != 27 or res["wide"][2][0] != 20:
            bad += 1
            print("FAIL crypt" if not ok else "KIND crypt", d, n, ln, stride, off, kind, kd, res["wide"][2], flush=True)
    # digests: sha3 at four d, kmac_xof with long outputs
    for d in (224, 256, 384, 512):
        for n, ln in ((1, 0), (1, 135), (3, 136), (5, 1000), (64, 4097), (2, 71), (9, 144 * 50 + 3)):
            stride = (ln + 7) // 8 * 8 + 8
            msgs = rand(n * stride + 64, 40 + n + ln)
            outs = {}
            for name, lanes in (("il", 0), ("wide", OLD), ("lane", 1)):
                _lib.check(lib.capy_set_sponge_lanes(lanes))
                out = torch.zeros(n * (d // 8), dtype=torch.uint8, device="cuda")
                _lib.check(lib.capy_sha3_batch_dev(d, n, msgs.data_ptr(), None, ln, stride, out.data_ptr(), sp))
                torch.cuda.synchronize()
                outs[name] = (out, last_kind())
            _lib.check(lib.capy_set_sponge_lanes(0))
            ok = torch.equal(outs["il"][0], outs["wide"][0]) and torch.equal(outs["il"][0], outs["lane"][0])
            if not ok or outs["il"][1][0] != 10 or outs["wide"][1][0] != 2:
                bad += 1
                print("FAIL sha3" if not ok else "KIND sha3", d, n, ln, outs["il"][1], outs["wide"][1], flush=True)
    for d in (256, 512):
        for n, ln, ol in ((1, 0, 32), (2, 100, 64), (3, 1000, 1000), (50, 136, 171), (1, 5000, 4096)):
            stride = (ln + 7) // 8 * 8 + 8
            msgs, keys = rand(n * stride + 64, 60 + n + ln), rand(n * 32, 61 + n)
            outs = {}
            for name, lanes in (("il", 0), ("wide", OLD), ("lane", 1)):
                _lib.check(lib.capy_set_sponge_lanes(lanes))
                out = torch.zeros(n * ol + 8, dtype=torch.uint8, device="cuda")
                _lib.check(lib.capy_kmac_xof_batch_dev(d, n, keys.data_ptr(), 32, 32, None, msgs.data_ptr(), None, ln, stride, 8 * ol, b"T", 1, out.data_ptr(), ol, sp))
                torch.cuda.synchronize()
                outs[name] = (out, last_kind())
            _lib.check(lib.capy_set_sponge_lanes(0))
            ok = torch.equal(outs["il"][0], outs["wide"][0]) and torch.equal(outs["il"][0], outs["lane"][0]) and int(outs["il"][0][n * ol:].sum()) == 0
            if not ok or outs["il"][1][0] != 10:
                bad += 1
                print("FAIL kmac_xof" if not ok else "KIND kmac_xof", d, n, ln, ol, outs["il"][1], flush=True)
    print("cases", len(cases), "+ digests; bad", bad)
    return bad


def timeit(fn, prep=None, reps=3):
    best = 1e9
    for _ in range(reps):
        if prep:
            prep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def timing():
    ln = 5 * 1024 * 1024
    stride = ln + 128
    print("# seconds per call, best of 3; il = sponge_wide_il.h (default); other = the four-lane fused kernel (encrypt / decrypt), the two-lane kernel (digests); D512 / SHA3-256, 5 MiB messages")
    print("# n | sha3_encrypt il (kind) | wide (kind) | ratio | sha3_decrypt il | wide | ratio | SHA3-256 il | wide | ratio")
    for n in [int(x) for x in os.environ.get("NS", "1,16,128,256,512,1024,2048").split(",")]:
        pws, zs, m = rand(n * 32, 1), rand(n * 512, 2), rand(n * stride, 3)
        orig = m.clone()
        tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
        status = torch.zeros(n, dtype=torch.int32, device="cuda")
        out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
        enc = lambda: encrypt(512, n, pws, zs, m, 0, ln, stride, tags)  # noqa: E731
        dec = lambda: decrypt(512, n, pws, zs, m, 0, ln, stride, tags, status)  # noqa: E731
        dig = lambda: _lib.check(lib.capy_sha3_batch_dev(256, n, m.data_ptr(), None, ln, stride, out.data_ptr(), sp))  # noqa: E731
        row = []
        for fn, prep in ((enc, None), (dec, enc), (dig, None)):
            ts = []
            force = (32 << 8) if os.environ.get("FORCE") else 0  # FORCE=1: the wave-per-item kernels for up to 4096 items
            for lanes in (force, OLD if fn is dig else NOWIDE):
                _lib.check(lib.capy_set_sponge_lanes(lanes))
                t = timeit(fn, prep)
                ts.append((t, last_kind()[0]))
                if fn is dec:
                    assert int(status.sum()) == 0 and torch.equal(m, orig)
            _lib.check(lib.capy_set_sponge_lanes(0))
            row.append("%.4f (%d) | %.4f (%d) | %.2f" % (ts[0][0], ts[0][1], ts[1][0], ts[1][1], ts[1][0] / ts[0][0]))
        print("%5d | %s" % (n, " | ".join(row)), flush=True)


def small():
    """short messages: the kernel's set-up (index registers, round constants) against the two-lane kernel's"""
    print("# KMACXOF256, 32-byte keys, 64 bytes out; ms per call (200 calls back to back, best of 3): n | message bytes | il | two-lane | ratio")
    for n in (1, 128, 1024, 2048):
        for ln in (64, 1024, 16384):
            stride = ln + 8
            msgs, keys = rand(n * stride, 5), rand(n * 32, 6)
            out = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
            ts = []
            for lanes in (0, 2):
                _lib.check(lib.capy_set_sponge_lanes(lanes))
                fn = lambda: [_lib.check(lib.capy_kmac_xof_batch_dev(256, n, keys.data_ptr(), 32, 32, None, msgs.data_ptr(), None, ln, stride, 512,  # noqa: E731
                                                                      b"T", 1, out.data_ptr(), 64, sp)) for _ in range(200)]
                fn()
                ts.append((timeit(fn) / 200 * 1e3, last_kind()[0]))
            _lib.check(lib.capy_set_sponge_lanes(0))
            print("%5d | %6d | %.4f (%d) | %.4f (%d) | %.2f" % (n, ln, ts[0][0], ts[0][1], ts[1][0], ts[1][1], ts[1][0] / ts[0][0]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        small()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        timing()
        sys.exit(0)
    sys.exit(1 if check() else 0)
