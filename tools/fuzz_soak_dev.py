#!/usr/bin/env python3
"""The device-buffer half of tools/fuzz_soak.py: the *_dev entry points with random layouts -- base pointers at any byte
alignment, uniform batches with any stride (>= the length), ragged batches through device offsets with gaps, per-item keys
through strides or device offsets, any output stride the header allows, a side stream or the default one -- against the CPU
oracle (the checker, as in tests/).  Also checks that nothing outside the declared output ranges is written (guard bytes).
usage: SECONDS=240 SEED=1 python3 tools/fuzz_soak_dev.py   -> profiles/r04_fuzz_soak.txt"""
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
seed = int(os.environ.get("SEED", "1"))
budget = float(os.environ.get("SECONDS", "240"))
rng = random.Random(seed)
DS = (224, 256, 384, 512)
GUARD = 0xA5
stats, failures = {}, []
side = torch.cuda.Stream()


def logn(lo, hi):
    return int(round(lo * (hi / lo) ** rng.random()))


def note(op, ok, detail):
    s = stats.setdefault(op, [0, 0])
    s[0] += 1
    if not ok:
        s[1] += 1
        failures.append((op, detail))
        print("FAIL", op, detail, flush=True)


def msg_len(d, cap):
    w, r1 = (1600 - d) // 8, (1600 - 2 * d) // 8
    c = rng.random()
    if c < 0.35:
        return max(0, min(cap, rng.choice((w, r1, 136)) * rng.randint(0, 6) + rng.randint(-4, 4)))
    if c < 0.5:
        return rng.randint(0, 16)
    return logn(1, cap) if cap > 1 else 0


class Layout:
    """n byte strings placed in one device buffer: uniform (stride) or ragged (offsets with random gaps), at a random
    alignment; guard bytes everywhere else."""

    def __init__(self, items, uniform, align=None, stride_slack=True):
        self.items = items
        n = len(items)
        self.base_off = rng.randint(0, 7) if align is None else align
        if uniform:
            ln = len(items[0]) if n else 0
            self.len = ln
            self.stride = ln + (rng.choice((0, 0, 1, 3, 8, 40)) if stride_slack else 0)
            if stride_slack and rng.random() < 0.5:
                self.stride = (self.stride + 7) // 8 * 8
            self.starts = [i * self.stride for i in range(n)]
            self.offsets = None
            total = n * self.stride + 8
        else:
            self.len = 0
            self.stride = 0
            pos, self.starts, offs = 0, [], []
            for it in items:
                self.starts.append(pos)
                offs.append(pos)
                pos += len(it)
            offs.append(pos)
            # device offsets are contiguous (message i ends where i + 1 starts); random gaps would need `lens`
            self.offsets = torch.tensor(offs, dtype=torch.int64, device=dev)
            total = pos + 8
        host = bytearray([GUARD]) * (self.base_off + total + 16)
        for st, it in zip(self.starts, items):
            host[self.base_off + st:self.base_off + st + len(it)] = it
        self.host0 = bytes(host)
        self.t = torch.tensor(list(host), dtype=torch.uint8, device=dev) if len(host) < (1 << 16) else torch.frombuffer(
            bytearray(host), dtype=torch.uint8).to(dev)

    @property
    def ptr(self):
        return self.t.data_ptr() + self.base_off

    @property
    def off_ptr(self):
        return self.offsets.data_ptr() if self.offsets is not None else None

    def read(self):
        """-> (items as they are now, True if every byte outside the items still holds what it held)"""
        now = bytes(self.t.cpu().numpy())
        out, mask = [], bytearray(now)
        ref = bytearray(self.host0)
        for st, it in zip(self.starts, self.items):
            a = self.base_off + st
            out.append(now[a:a + len(it)])
            mask[a:a + len(it)] = bytes(len(it))
            ref[a:a + len(it)] = bytes(len(it))
        return out, bytes(mask) == bytes(ref)


class Out:
    """n outputs of `size` bytes at a stride; guard bytes in between and around"""

    def __init__(self, n, size, stride=None, mult8=False):
        self.n, self.size = n, size
        if stride is None:
            stride = size + rng.choice((0, 0, 8, 24))
            if mult8:
                stride = (stride + 7) // 8 * 8
        self.stride = stride
        self.base_off = 8 * rng.randint(0, 2)
        self.t = torch.full((self.base_off + max(1, n) * max(stride, 1) + 64,), GUARD, dtype=torch.uint8, device=dev)

    @property
    def ptr(self):
        return self.t.data_ptr() + self.base_off

    def read(self):
        now = bytes(self.t.cpu().numpy())
        rows = [now[self.base_off + i * self.stride:self.base_off + i * self.stride + self.size] for i in range(self.n)]
        clean = all(b == GUARD for b in now[:self.base_off])
        for i in range(self.n):
            a = self.base_off + i * self.stride + self.size
            b = self.base_off + (i + 1) * self.stride if i + 1 < self.n else len(now)
            clean = clean and all(x == GUARD for x in now[a:b])
        return rows, clean


def stream_ptr():
    if rng.random() < 0.3:
        return side, C.c_void_p(side.cuda_stream)
    st = torch.cuda.current_stream()
    return st, C.c_void_p(st.cuda_stream)


def run(fn):
    st, sp = stream_ptr()
    if st is side:
        side.wait_stream(torch.cuda.current_stream())
    rc = fn(sp)
    st.synchronize()
    return rc


def keys_layout(n, strided=False):
    """strided: the entry point takes a key stride (kmac_xof); otherwise equally long keys sit back to back"""
    if rng.random() < 0.5:
        kl = rng.choice((0, 1, 16, 32, 64, 135, 136, 200))
        keys = [rng.randbytes(kl) for _ in range(n)]
        return keys, Layout(keys, True, stride_slack=strided), kl
    keys = [rng.randbytes(rng.randint(0, 200)) for _ in range(n)]
    return keys, Layout(keys, False), 0


def messages(n, d, cap):
    uniform = rng.random() < 0.4
    if uniform:
        ln = msg_len(d, cap)
        m = [rng.randbytes(ln) for _ in range(n)]
    else:
        m = [rng.randbytes(msg_len(d, cap)) for _ in range(n)]
    return m, Layout(m, uniform)


def op_sha3():
    d = rng.choice(DS)
    cap = logn(8, 1 << 14)
    n = logn(1, max(1, min(4000, (1 << 20) // cap)))
    m, L = messages(n, d, cap)
    out = Out(n, d // 8, stride=d // 8)
    rc = run(lambda sp: lib.capy_sha3_batch_dev(d, n, L.ptr, L.off_ptr, L.len, L.stride, out.ptr, sp))
    rows, clean = out.read()
    _, untouched = L.read()
    bad = [i for i in range(n) if rows[i] != O.sha3(m[i], d)]
    note("sha3_dev", rc == 0 and not bad and clean and untouched, (d, n, cap, L.base_off, L.stride, rc, bad[:3], clean, untouched))


def op_kmac():
    d = rng.choice(DS)
    cap = logn(4, 1 << 13)
    n = logn(1, max(1, min(3000, (1 << 19) // cap)))
    l = rng.choice((8, 64, 448, 512, 1088, 8 * rng.randint(1, 500)))
    s = rng.choice((b"", b"SKE", b"T", rng.randbytes(rng.randint(0, 40)), b"x" * rng.randint(150, 200)))
    keys, KL, kl = keys_layout(n, strided=True)
    m, L = messages(n, d, cap)
    out = Out(n, l // 8, mult8=True)
    rc = run(lambda sp: lib.capy_kmac_xof_batch_dev(d, n, KL.ptr, kl, KL.stride, KL.off_ptr, L.ptr, L.off_ptr, L.len, L.stride, l,
                                                    s, len(s), out.ptr, out.stride, sp))
    rows, clean = out.read()
    bad = [i for i in range(n) if rows[i] != O.kmac_xof(keys[i], m[i], l, s, d)]
    note("kmac_xof_dev", rc == 0 and not bad and clean, (d, n, cap, l, len(s), kl, L.base_off, KL.base_off, rc, bad[:3], clean))


def op_cshake():
    d = rng.choice(DS)
    cap = logn(4, 1 << 12)
    n = logn(1, 600)
    l = 8 * rng.randint(1, 300)
    nn = rng.choice(((b"", b""), (b"", b"Email"), (b"fn", b""), (b"KMAC", b"custom")))
    m, L = messages(n, d, cap)
    out = Out(n, l // 8, mult8=True)
    rc = run(lambda sp: lib.capy_cshake_batch_dev(d, n, L.ptr, L.off_ptr, L.len, L.stride, l, nn[0], len(nn[0]), nn[1], len(nn[1]),
                                                  out.ptr, out.stride, sp))
    rows, clean = out.read()
    bad = [i for i in range(n) if rows[i] != O.cshake(m[i], l, nn[0], nn[1], d)]
    note("cshake_dev", rc == 0 and not bad and clean, (d, n, cap, l, nn, L.base_off, rc, bad[:3], clean))


def op_sym(kem=False):
    d = rng.choice(DS)
    cap = logn(4, 1 << 14)
    n = logn(1, max(1, min(3000, (1 << 19) // cap)))
    if kem:
        sl = rng.choice((16, 32, 64))
        pw = [rng.randbytes(sl) for _ in range(n)]
        PL, pl = Layout(pw, True, stride_slack=False), sl
    else:
        pw, PL, pl = keys_layout(n)
    m, L = messages(n, d, cap)
    z = [rng.randbytes(512) for _ in range(n)]
    ZL = Layout(z, True, align=0, stride_slack=False)
    tags = Out(n, 64, stride=64)
    pw_bytes = sum(len(p) for p in pw)
    if kem:
        rc = run(lambda sp: lib.capy_kem_sponge_encrypt_batch_dev(d, n, PL.ptr, pl, ZL.ptr, L.ptr, L.off_ptr, L.len, L.stride, tags.ptr, sp))
        want = None
    else:
        rc = run(lambda sp: lib.capy_sha3_encrypt_batch_dev(d, n, PL.ptr, pl, PL.off_ptr, pw_bytes, ZL.ptr, L.ptr, L.off_ptr, L.len,
                                                            L.stride, tags.ptr, sp))
        want = [O.sha3_encrypt(pw[i], z[i], m[i], d) for i in (range(n) if n <= 64 else sorted({0, n - 1, rng.randrange(n), rng.randrange(n)}))]
    ct, untouched = L.read()
    tg, clean = tags.read()
    ok = rc == 0 and untouched and clean
    if want is not None:
        idx = list(range(n)) if n <= 64 else None
        if idx is not None:
            ok = ok and all((ct[i], tg[i]) == want[i] for i in idx)
    # decrypt in place with one wrong password
    w = rng.randrange(n)
    status = torch.full((n + 2,), 77, dtype=torch.int32, device=dev)
    if kem:
        bad_pw = list(pw)
        bad_pw[w] = bytes(x ^ 1 for x in pw[w])
        PL2 = Layout(bad_pw, True, stride_slack=False)
        rc2 = run(lambda sp: lib.capy_kem_sponge_decrypt_batch_dev(d, n, PL2.ptr, pl, ZL.ptr, L.ptr, L.off_ptr, L.len, L.stride, tags.ptr,
                                                                   status.data_ptr(), sp))
    else:
        bad_pw = list(pw)
        bad_pw[w] = pw[w] + b"!" if PL.offsets is not None else bytes(x ^ 1 for x in pw[w]) if pw[w] else pw[w]
        PL2 = Layout(bad_pw, PL.offsets is None, stride_slack=False)
        pl2 = len(bad_pw[0]) if PL.offsets is None else 0
        rc2 = run(lambda sp: lib.capy_sha3_decrypt_batch_dev(d, n, PL2.ptr, pl2, PL2.off_ptr, sum(len(p) for p in bad_pw), ZL.ptr, L.ptr,
                                                             L.off_ptr, L.len, L.stride, tags.ptr, status.data_ptr(), sp))
    back, untouched2 = L.read()
    st = status.cpu().tolist()
    changed = bad_pw[w] != pw[w]
    good = rc2 == 0 and untouched2 and st[n] == 77 and all(
        (st[i] == 0 and back[i] == m[i]) if (i != w or not changed) else (st[i] == 1 and back[i] == ct[i]) for i in range(n))
    note("kem_sponge_dev" if kem else "sha3_encrypt/decrypt_dev", ok and good, (d, n, cap, L.base_off, L.stride, rc, rc2, ok, good))


def dev_bytes(items, align0=True):
    L = Layout(items, True, align=0 if align0 else None, stride_slack=False)
    return L


def op_sign():
    d = rng.choice(DS)
    n = rng.choice((logn(1, 200), logn(200, 9000)))
    cap = logn(1, 1024)
    pw, PL, pl = keys_layout(n)
    m, L = messages(n, d, cap)
    pub = Out(n, 112, stride=112)
    h, z = Out(n, 56, stride=56), Out(n, 56, stride=56)
    rc = run(lambda sp: lib.capy_keypair_batch_dev(d, n, PL.ptr, pl, PL.off_ptr, pub.ptr, sp))
    rc |= run(lambda sp: lib.capy_schnorr_sign_batch_dev(d, n, PL.ptr, pl, PL.off_ptr, L.ptr, L.off_ptr, L.len, L.stride, h.ptr, z.ptr, sp))
    pubs, c1 = pub.read()
    hs, c2 = h.read()
    zs, c3 = z.read()
    idx = sorted({0, n - 1} | {rng.randrange(n) for _ in range(6)})
    bad = [i for i in idx if pubs[i] != O.keypair_pub(pw[i], d) or (hs[i], zs[i]) != O.sign(pw[i], m[i], d)]
    status = torch.full((n + 2,), 77, dtype=torch.int32, device=dev)
    w = rng.randrange(n)
    z.t[z.base_off + 56 * w:z.base_off + 56 * w + 56] = 0
    rc |= run(lambda sp: lib.capy_schnorr_verify_batch_dev(d, n, pub.ptr, L.ptr, L.off_ptr, L.len, L.stride, h.ptr, z.ptr,
                                                           status.data_ptr(), sp))
    st = status.cpu().tolist()
    good = st[n] == 77 and all(st[i] == (1 if i == w else 0) for i in range(n))
    note("keypair/sign/verify_dev", rc == 0 and not bad and good and c1 and c2 and c3, (d, n, cap, rc, bad[:3], good, c1, c2, c3))


def op_key_crypt():
    d = rng.choice(DS)
    n = rng.choice((logn(1, 200), logn(200, 9000)))
    cap = logn(1, 1024)
    pw, PL, pl = keys_layout(n)
    m, L = messages(n, d, cap)
    pub = Out(n, 112, stride=112)
    rc = run(lambda sp: lib.capy_keypair_batch_dev(d, n, PL.ptr, pl, PL.off_ptr, pub.ptr, sp))
    pubs, _ = pub.read()
    k = [rng.randbytes(56) for _ in range(n)]
    KL = dev_bytes(k)
    zxy, tags = Out(n, 112, stride=112), Out(n, 56, stride=56)
    rc |= run(lambda sp: lib.capy_key_encrypt_batch_dev(d, n, pub.ptr, KL.ptr, L.ptr, L.off_ptr, L.len, L.stride, zxy.ptr, tags.ptr, sp))
    ct, untouched = L.read()
    zs, c1 = zxy.read()
    tg, c2 = tags.read()
    idx = sorted({0, n - 1} | {rng.randrange(n) for _ in range(5)})
    bad = [i for i in idx if (ct[i], zs[i], tg[i]) != O.key_encrypt(pubs[i], k[i], m[i], d)]
    status = torch.full((n + 2,), 77, dtype=torch.int32, device=dev)
    w = rng.randrange(n)
    tags.t[tags.base_off + 56 * w] ^= 1  # a forged tag: that item must fail and keep its ciphertext
    rc |= run(lambda sp: lib.capy_key_decrypt_batch_dev(d, n, PL.ptr, pl, PL.off_ptr, zxy.ptr, L.ptr, L.off_ptr, L.len, L.stride, tags.ptr,
                                                        status.data_ptr(), sp))
    back, untouched2 = L.read()
    st = status.cpu().tolist()
    good = st[n] == 77 and all((st[i] == 1 and back[i] == ct[i]) if i == w else (st[i] == 0 and back[i] == m[i]) for i in range(n))
    note("key_encrypt/decrypt_dev", rc == 0 and not bad and good and untouched and untouched2 and c1 and c2,
         (d, n, cap, rc, bad[:3], good, untouched, untouched2, c1, c2))


OPS = [(op_sha3, 3), (op_kmac, 3), (op_cshake, 1), (op_sym, 3), (lambda: op_sym(True), 1), (op_sign, 2), (op_key_crypt, 2)]
table = [f for f, wgt in OPS for _ in range(wgt)]
t0 = time.time()
last = t0
while time.time() - t0 < budget:
    f = rng.choice(table)
    try:
        f()
    except Exception as e:
        note(getattr(f, "__name__", "op"), False, "exception: %r" % (e,))
    if time.time() - last > 50:
        last = time.time()
        print("# %4.0f s: %s" % (last - t0, {k: v[0] for k, v in stats.items()}), flush=True)
print("# fuzz_soak_dev seed %d, %.0f s on %s" % (seed, time.time() - t0, lib.capy_version().decode()))
for k in sorted(stats):
    print("%-28s calls %5d   failures %d" % (k, stats[k][0], stats[k][1]))
print("# total calls %d, failures %d" % (sum(v[0] for v in stats.values()), len(failures)))
sys.exit(1 if failures else 0)
