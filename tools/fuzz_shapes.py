#!/usr/bin/env python3
"""Randomised differential soak of the sponge LAUNCHER: big uniform device batches at random shapes (n up to what 6 GiB hold,
lengths 0 .. 1 MiB with the rate boundaries over-represented, strides with slack) through the automatic kernel choice
(capy_set_sponge_lanes(0): wide / two-lane / mixed / rotating / one-lane / uniform-framing kernels, full-chip heads with
remainders ...) against the same batch with a forced choice (1 = one lane per sponge, 2 = two lanes per sponge): digests,
XOF outputs and in-place ciphertexts must be byte-identical on the device, and a few items of every batch are checked against
the CPU oracle (the checker, as in tests/).  capy_sha3_launch_plan names the kernel the automatic choice took.
usage: SECONDS=240 SEED=1 python3 tools/fuzz_shapes.py   -> profiles/r04_fuzz_soak.txt"""
import ctypes as C
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402
from oracle import oracle as O  # noqa: E402

lib = _lib.lib()
_lib.check(lib.capy_set_device(0))
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
sp = C.c_void_p(st.cuda_stream)
seed = int(os.environ.get("SEED", "1"))
budget = float(os.environ.get("SECONDS", "240"))
cap_bytes = int(os.environ.get("CAP_BYTES", str(6 << 30)))
rng = random.Random(seed)
DS = (224, 256, 384, 512)
stats, failures, plans, crypt_kinds = {}, [], {}, {}


def logn(lo, hi):
    return int(round(lo * (hi / lo) ** rng.random()))


def note(op, ok, detail):
    s = stats.setdefault(op, [0, 0])
    s[0] += 1
    if not ok:
        s[1] += 1
        failures.append((op, detail))
        print("FAIL", op, detail, flush=True)


def shape(d, max_len=1 << 20):
    w, r1 = (1600 - d) // 8, (1600 - 2 * d) // 8
    c = rng.random()
    if c < 0.3:
        ln = max(0, rng.choice((w, r1, 136)) * rng.randint(0, 40) + rng.randint(-9, 9))
    elif c < 0.4:
        ln = rng.randint(0, 64)
    else:
        ln = logn(1, max_len)
    if rng.random() < 0.6:
        ln = ln // 8 * 8  # the aligned fast paths (mixed / rotating / uniform kernels want 8-byte multiples)
    stride = (ln + 7) // 8 * 8 + 8 * rng.choice((0, 0, 1, 16))
    if rng.random() < 0.15:
        stride = ln + rng.choice((0, 1, 3))  # unaligned starts: the generic kernels
    stride = max(stride, 1)
    # batch size: log-uniform, biased towards the thresholds in items per SIMD (1024 SIMDs)
    if rng.random() < 0.5:
        n = int(1024 * rng.choice((1, 2, 16, 32, 48, 64, 96, 128, 192, 256)) * rng.uniform(0.8, 1.25))
    else:
        n = logn(1, 1 << 21)
    n = max(1, min(n, cap_bytes // stride))
    return ln, stride, n


def fill(nbytes, s):
    nbytes = (nbytes + 64 + 7) // 8 * 8
    t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), nbytes, s, sp))
    return t


def item_bytes(t, stride, ln, i):
    return bytes(t[i * stride:i * stride + ln].cpu().numpy())


def picks(n):
    return sorted({0, n - 1, rng.randrange(n), rng.randrange(n)})


def with_lanes(lanes, fn):
    _lib.check(lib.capy_set_sponge_lanes(lanes))
    try:
        return fn()
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))


def op_sha3():
    d = rng.choice(DS)
    ln, stride, n = shape(d)
    msgs = fill(n * stride, rng.getrandbits(32))
    kind, phases = C.c_int(0), C.c_int(0)
    lib.capy_sha3_launch_plan(d, n, ln, stride, C.byref(kind), C.byref(phases))
    plans[kind.value] = plans.get(kind.value, 0) + 1
    outs = {}
    for lanes in (0, rng.choice((1, 2))):
        o = torch.zeros(n * (d // 8), dtype=torch.uint8, device=dev)
        rc = with_lanes(lanes, lambda: lib.capy_sha3_batch_dev(d, n, msgs.data_ptr(), None, ln, stride, o.data_ptr(), sp))
        torch.cuda.synchronize()
        outs[lanes] = (rc, o)
    (rc0, a), (rc1, b) = outs[0], [v for k, v in outs.items() if k != 0][0]
    same = rc0 == 0 and rc1 == 0 and torch.equal(a, b)
    bad = []
    if same:
        host = bytes(a.cpu().numpy()) if n * (d // 8) < (1 << 26) else None
        for i in picks(n):
            got = host[i * (d // 8):(i + 1) * (d // 8)] if host else bytes(a[i * (d // 8):(i + 1) * (d // 8)].cpu().numpy())
            if got != O.sha3(item_bytes(msgs, stride, ln, i), d):
                bad.append(i)
    note("sha3 auto vs forced", same and not bad, (d, n, ln, stride, "plan", kind.value, phases.value, rc0, rc1, same, bad))


def op_kmac():
    d = rng.choice(DS)
    ln, stride, n = shape(d, 1 << 16)
    l = rng.choice((8 * (d // 8), 512, 1024, 8192, 8 * rng.randint(1, 2000)))
    ol = l // 8
    os_ = (ol + 7) // 8 * 8
    n = max(1, min(n, (2 << 30) // max(os_, 1)))
    kl = rng.choice((0, 32, 64, 136))
    s = rng.choice((b"SKE", b"", b"T" * 20))
    absorb = rng.random() < 0.6
    msgs = fill(n * stride, rng.getrandbits(32)) if absorb else None
    keys = fill(max(1, n * kl), rng.getrandbits(32))
    outs = {}
    for lanes in (0, rng.choice((1, 2))):
        o = torch.zeros(n * os_ + 8, dtype=torch.uint8, device=dev)
        rc = with_lanes(lanes, lambda: lib.capy_kmac_xof_batch_dev(d, n, keys.data_ptr(), kl, kl, None, msgs.data_ptr() if absorb else None,
                                                                   None, ln if absorb else 0, stride if absorb else 0, l, s, len(s),
                                                                   o.data_ptr(), os_, sp))
        torch.cuda.synchronize()
        outs[lanes] = (rc, o)
    (rc0, a), (rc1, b) = outs[0], [v for k, v in outs.items() if k != 0][0]
    same = rc0 == 0 and rc1 == 0 and torch.equal(a, b)
    bad = []
    if same:
        for i in picks(n):
            got = bytes(a[i * os_:i * os_ + ol].cpu().numpy())
            key = bytes(keys[i * kl:(i + 1) * kl].cpu().numpy())
            x = item_bytes(msgs, stride, ln, i) if absorb else b""
            if got != O.kmac_xof(key, x, l, s, d):
                bad.append(i)
    note("kmac_xof auto vs forced", same and not bad, (d, n, ln if absorb else None, stride, l, kl, rc0, rc1, same, bad))


def op_encrypt():
    d = rng.choice(DS)
    ln, stride, n = shape(d, 1 << 18)
    stride = (max(ln, 1) + 7) // 8 * 8 + 8 * rng.choice((0, 2))
    n = max(1, min(n, cap_bytes // (2 * stride)))
    pl = rng.choice((0, 16, 64))
    pws = fill(max(1, n * pl), rng.getrandbits(32))
    zs = fill(n * 512, rng.getrandbits(32))
    plain = fill(n * stride, rng.getrandbits(32))
    res = {}
    # the forced form is mostly the two-pass one (bit 16: no fused kernel at all), so that the fused kernels are compared with
    # different code and not with themselves
    for lanes in (0, rng.choice((2, 1 | (1 << 16), 1 | (1 << 16), 1 | (1 << 16)))):
        m = plain.clone()
        tags = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
        rc = with_lanes(lanes, lambda: lib.capy_sha3_encrypt_batch_dev(d, n, pws.data_ptr(), pl, None, n * pl, zs.data_ptr(), m.data_ptr(), None,
                                                                       ln, stride, tags.data_ptr(), sp))
        torch.cuda.synchronize()
        res[lanes] = (rc, m, tags)
        if lanes == 0:
            k, l = C.c_int(0), C.c_int(0)
            lib.capy_debug_last_sponge_kernel(C.byref(k), C.byref(l))
            crypt_kinds[k.value] = crypt_kinds.get(k.value, 0) + 1
    (rc0, m0, t0), (rc1, m1, t1) = res[0], [v for k, v in res.items() if k != 0][0]
    same = rc0 == 0 and rc1 == 0 and torch.equal(m0, m1) and torch.equal(t0, t1)
    bad = []
    if same:
        for i in picks(n)[:3]:
            want = O.sha3_encrypt(bytes(pws[i * pl:(i + 1) * pl].cpu().numpy()), bytes(zs[i * 512:(i + 1) * 512].cpu().numpy()),
                                  item_bytes(plain, stride, ln, i), d)
            if (item_bytes(m0, stride, ln, i), bytes(t0[64 * i:64 * i + 64].cpu().numpy())) != want:
                bad.append(i)
        # and back, under the automatic choice, with item 0's tag forged
        status = torch.full((n,), 7, dtype=torch.int32, device=dev)
        t0[0] ^= 1
        ct0 = item_bytes(m0, stride, ln, 0)
        rc2 = lib.capy_sha3_decrypt_batch_dev(d, n, pws.data_ptr(), pl, None, n * pl, zs.data_ptr(), m0.data_ptr(), None, ln, stride,
                                              t0.data_ptr(), status.data_ptr(), sp)
        torch.cuda.synchronize()
        ok_rest = bool((status[1:] == 0).all()) and (n == 1 or all(
            torch.equal(m0[i * stride:i * stride + ln], plain[i * stride:i * stride + ln]) for i in picks(n) if i))
        same = rc2 == 0 and int(status[0]) == 1 and item_bytes(m0, stride, ln, 0) == ct0 and ok_rest
    note("sha3_encrypt auto vs forced", same and not bad, (d, n, ln, stride, pl, rc0, rc1, same, bad))


OPS = [(op_sha3, 4), (op_kmac, 3), (op_encrypt, 3)]
table = [f for f, wgt in OPS for _ in range(wgt)]
t0 = time.time()
last = t0
while time.time() - t0 < budget:
    f = rng.choice(table)
    try:
        f()
    except Exception as e:
        note(f.__name__, False, "exception: %r" % (e,))
    torch.cuda.empty_cache()
    if time.time() - last > 50:
        last = time.time()
        print("# %4.0f s: %s plans %s" % (last - t0, {k: v[0] for k, v in stats.items()}, dict(sorted(plans.items()))), flush=True)
print("# fuzz_shapes seed %d, %.0f s on %s" % (seed, time.time() - t0, lib.capy_version().decode()))
print("# capy_sha3_launch_plan kinds taken by the automatic choice (see include/capyhip.h):", dict(sorted(plans.items())))
print("# sha3_encrypt schedules taken by the automatic choice (capy_debug_last_sponge_kernel: 20 / 22 four lanes per item, the same in "
      "slices; 23 / 24 / 25 one lane per sponge: one launch, slices, rotating occupancy; 26 two passes; 27 two waves per item):", dict(sorted(crypt_kinds.items())))
for k in sorted(stats):
    print("%-30s calls %5d   failures %d" % (k, stats[k][0], stats[k][1]))
print("# total calls %d, failures %d" % (sum(v[0] for v in stats.values()), len(failures)))
sys.exit(1 if failures else 0)
