#!/usr/bin/env python3
"""Randomised differential soak of the whole C ABI against the CPU oracle (oracle/ is the checker here, as in tests/): for
SECONDS (default 240) pick an operation, a security parameter, a batch size (log-uniform, so that every kernel family and
every launcher threshold is crossed: wave / quad / duo / lane Ed448 kernels, wide / fused / two-pass sponge paths, uniform
and ragged batches) and message lengths (ragged, with the rate boundaries over-represented), run it on the GPU and compare
every output -- or, for Ed448 batches too large for the scalar oracle, a random sample plus a byte-for-byte comparison
against the same batch forced through another kernel family.  Prints one line per failure and a summary.
usage: SECONDS=240 SEED=1 python3 tools/fuzz_soak.py   -> profiles/r04_fuzz_soak.txt"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capycrypt_amd import _lib, ops  # noqa: E402
from oracle import oracle as O  # noqa: E402

lib = _lib.lib()
_lib.check(lib.capy_set_device(0))
# DEVICES="0,0,0": the in-library sharding (capy_set_devices) under the same soak -- on a one-GPU box every shard lands on the
# same card, which exercises the shard cut, the per-worker pools and the output slices, not a speed-up
_devs = [int(x) for x in os.environ.get("DEVICES", "").split(",") if x.strip() != ""]
if _devs:
    import ctypes as _C

    _lib.check(lib.capy_set_devices((_C.c_int * len(_devs))(*_devs), len(_devs)))
seed = int(os.environ.get("SEED", "1"))
budget = float(os.environ.get("SECONDS", "240"))
rng = random.Random(seed)
DS = (224, 256, 384, 512)
stats, failures = {}, []


def logn(lo, hi):
    return int(round(lo * (hi / lo) ** rng.random()))


def msg_len(d, cap):
    """lengths around the block boundaries of both rates of d, or anything up to cap"""
    w, r1 = (1600 - d) // 8, (1600 - 2 * d) // 8
    c = rng.random()
    if c < 0.35:
        base = rng.choice((w, r1, 136)) * rng.randint(0, 6)
        return max(0, min(cap, base + rng.randint(-4, 4)))
    if c < 0.5:
        return rng.randint(0, 16)
    return logn(1, cap) if cap > 1 else 0


def msgs_for(n, d, cap, uniform=None):
    if uniform is None:
        uniform = rng.random() < 0.3
    if uniform:
        ln = msg_len(d, cap)
        return [rng.randbytes(ln) for _ in range(n)]
    return [rng.randbytes(msg_len(d, cap)) for _ in range(n)]


def keys_for(n):
    if rng.random() < 0.5:
        kl = rng.choice((0, 1, 16, 32, 64, 135, 136, 200))
        return [rng.randbytes(kl) for _ in range(n)]
    return [rng.randbytes(rng.randint(0, 200)) for _ in range(n)]


def sample(n, k=24):
    idx = set(range(min(n, 4))) | {n - 1} | {rng.randrange(n) for _ in range(k)}
    return sorted(i for i in idx if 0 <= i < n)


def note(op, ok, detail):
    s = stats.setdefault(op, [0, 0])
    s[0] += 1
    if not ok:
        s[1] += 1
        failures.append((op, detail))
        print("FAIL", op, detail, flush=True)


def budget_items(cap_bytes, per_item):
    return max(1, min(20000, cap_bytes // max(1, per_item)))


def op_sha3():
    d = rng.choice(DS)
    cap = logn(8, 1 << 16)
    n = logn(1, budget_items(1 << 22, cap))
    m = msgs_for(n, d, cap)
    got = ops.sha3_batch(m, d)
    bad = [i for i in range(n) if got[i] != O.sha3(m[i], d)]
    note("sha3", not bad, (d, n, cap, bad[:3]))


def op_kmac():
    d = rng.choice(DS)
    cap = logn(4, 1 << 14)
    n = logn(1, budget_items(1 << 21, cap))
    l = rng.choice((8, 64, 448, 512, 1088, 8 * rng.randint(1, 700)))
    s = rng.choice((b"", b"SKE", b"SKA", b"T", rng.randbytes(rng.randint(0, 40)), b"x" * rng.randint(150, 200)))
    k, m = keys_for(n), msgs_for(n, d, cap)
    got = ops.kmac_xof_batch(k, m, l, s, d)
    bad = [i for i in range(n) if got[i] != O.kmac_xof(k[i], m[i], l, s, d)]
    note("kmac_xof", not bad, (d, n, cap, l, len(s), bad[:3]))


def op_cshake():
    d = rng.choice(DS)
    cap = logn(4, 1 << 13)
    n = rng.choice((logn(1, 300), logn(300, 6000)))
    l = 8 * rng.randint(1, 300)
    nn, s = rng.choice(((b"", b""), (b"", b"Email"), (b"fn", b""), (b"KMAC", b"custom"))), None
    m = msgs_for(n, d, cap)
    got = ops.cshake_batch(m, l, nn[0], nn[1], d)
    bad = [i for i in range(n) if got[i] != O.cshake(m[i], l, nn[0], nn[1], d)]
    note("cshake", not bad, (d, n, cap, l, nn, bad[:3]))


def op_sym():
    d = rng.choice(DS)
    cap = logn(4, 1 << 15)
    n = logn(1, budget_items(1 << 21, cap))
    pw, m = keys_for(n), msgs_for(n, d, cap)
    z = [rng.randbytes(512) for _ in range(n)]
    ct, tags = ops.sha3_encrypt_batch(pw, z, m, d)
    idx = sample(n, 40) if n > 64 else range(n)
    bad = [i for i in idx if (ct[i], tags[i]) != O.sha3_encrypt(pw[i], z[i], m[i], d)]
    # one wrong password: that item must fail and keep its ciphertext, the others decrypt
    w = rng.randrange(n)
    pw2 = list(pw)
    pw2[w] = pw[w] + b"!"
    back, ok = ops.sha3_decrypt_batch(pw2, z, ct, tags, d)
    good = all(ok[i] and back[i] == m[i] for i in range(n) if i != w) and (not ok[w]) and back[w] == ct[w]
    note("sha3_encrypt/decrypt", not bad and good, (d, n, cap, bad[:3], "roundtrip", good))


def curve_inputs(n):
    sc = [rng.randbytes(56) for _ in range(n)]
    edge = [(0).to_bytes(56, "big"), (1).to_bytes(56, "big"), b"\xff" * 56]
    for j, e in enumerate(edge):
        if j < n and rng.random() < 0.5:
            sc[rng.randrange(n)] = e
    base = ops.ed448_basemul_batch([rng.randbytes(56) for _ in range(n)])
    return sc, base


def with_families_off(fn):
    _lib.check(lib.capy_ed448_set_quad_range(0, 0))
    _lib.check(lib.capy_ed448_set_duo_range(0, 0))
    try:
        return fn()
    finally:
        _lib.check(lib.capy_ed448_set_quad_range(-1, -1))
        _lib.check(lib.capy_ed448_set_duo_range(-1, -1))


def big_n():
    """mostly small and medium batches; one in eight beyond the wave quantum of the lane kernels (remainder peeling)"""
    return rng.choice((logn(1, 300),) * 4 + (logn(300, 40000),) * 3 + (logn(60000, 140000),))


def op_scalarmul():
    n = big_n()
    hard = rng.choice((ops.HARDEN_OFF, ops.HARDEN_ALL))
    sc, pts = curve_inputs(n)
    ops.ed448_set_hardened(hard)
    try:
        got = ops.ed448_scalarmul_batch(sc, pts)
        other = with_families_off(lambda: ops.ed448_scalarmul_batch(sc, pts)) if n > 4096 else got
    finally:
        ops.ed448_set_hardened(ops.HARDEN_PROTOCOL)
    bad = [i for i in sample(n, 12) if got[i] != O.ed448_scalarmul(sc[i], pts[i])]
    note("ed448_scalarmul", not bad and got == other, (n, hard, bad[:3], "families agree", got == other))


def op_basemul():
    n = rng.choice((logn(1, 300), logn(300, 40000)))
    hard = rng.choice((ops.HARDEN_OFF, ops.HARDEN_ALL))
    sc = [rng.randbytes(56) for _ in range(n)]
    ops.ed448_set_hardened(hard)
    try:
        got = ops.ed448_basemul_batch(sc)
    finally:
        ops.ed448_set_hardened(ops.HARDEN_PROTOCOL)
    bad = [i for i in sample(n, 12) if got[i] != O.ed448_basemul(sc[i])]
    note("ed448_basemul", not bad, (n, hard, bad[:3]))


def op_dsm():
    n = big_n()
    a = [rng.randbytes(56) for _ in range(n)]
    b, pts = curve_inputs(n)
    got = ops.ed448_double_scalarmul_batch(a, b, pts)
    other = with_families_off(lambda: ops.ed448_double_scalarmul_batch(a, b, pts)) if n > 4096 else got
    bad = [i for i in sample(n, 8) if got[i] != O.ed448_add(O.ed448_basemul(a[i]), O.ed448_scalarmul(b[i], pts[i]))]
    note("ed448_double_scalarmul", not bad and got == other, (n, bad[:3], "families agree", got == other))


def op_sign():
    d = rng.choice(DS)
    n = rng.choice((logn(1, 200), logn(200, 20000)))
    cap = logn(1, 2048)
    pw, m = keys_for(n), msgs_for(n, d, cap)
    pub = ops.keypair_batch(pw, d)
    sig = ops.schnorr_sign_batch(pw, m, d)
    idx = sample(n, 10)
    bad = [i for i in idx if pub[i] != O.keypair_pub(pw[i], d) or sig[i] != O.sign(pw[i], m[i], d)]
    ok = ops.schnorr_verify_batch(pub, m, sig, d)
    w = rng.randrange(n)
    sig2 = list(sig)
    sig2[w] = (sig[w][0], bytes(56))
    ok2 = ops.schnorr_verify_batch(pub, m, sig2, d)
    good = all(ok) and not ok2[w] and sum(ok2) == n - 1
    note("keypair/sign/verify", not bad and good, (d, n, cap, bad[:3], "verify", good))


def op_key_crypt():
    d = rng.choice(DS)
    n = rng.choice((logn(1, 200), logn(200, 20000)))
    cap = logn(1, 1024)
    pw, m = keys_for(n), msgs_for(n, d, cap)
    pub = ops.keypair_batch(pw, d)
    k = [rng.randbytes(56) for _ in range(n)]
    ct, zs, tags = ops.key_encrypt_batch(pub, k, m, d)
    bad = [i for i in sample(n, 8) if (ct[i], zs[i], tags[i]) != O.key_encrypt(pub[i], k[i], m[i], d)]
    w = rng.randrange(n)
    pw2 = list(pw)
    pw2[w] = pw[w] + b"?"
    back, ok = ops.key_decrypt_batch(pw2, zs, ct, tags, d)
    good = all(ok[i] and back[i] == m[i] for i in range(n) if i != w) and (not ok[w]) and back[w] == ct[w]
    note("key_encrypt/decrypt", not bad and good, (d, n, cap, bad[:3], "roundtrip", good))


OPS = [(op_sha3, 3), (op_kmac, 3), (op_cshake, 1), (op_sym, 3), (op_scalarmul, 2), (op_basemul, 1), (op_dsm, 1), (op_sign, 2),
       (op_key_crypt, 2)]
table = [f for f, wgt in OPS for _ in range(wgt)]
t0 = time.time()
last = t0
while time.time() - t0 < budget:
    f = rng.choice(table)
    try:
        f()
    except Exception as e:  # an error return from the library is a finding too
        note(f.__name__, False, "exception: %r" % (e,))
    if time.time() - last > 50:
        last = time.time()
        print("# %4.0f s: %s" % (last - t0, {k: v[0] for k, v in stats.items()}), flush=True)
print("# fuzz_soak seed %d, %.0f s on %s%s" % (seed, time.time() - t0, lib.capy_version().decode(),
                                              ", capy_set_devices(%s)" % _devs if _devs else ""))
for k in sorted(stats):
    print("%-26s calls %5d   failures %d" % (k, stats[k][0], stats[k][1]))
print("# total calls %d, failures %d" % (sum(v[0] for v in stats.values()), len(failures)))
sys.exit(1 if failures else 0)
