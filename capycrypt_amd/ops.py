"""Batched operators over the C ABI (include/capyhip.h).  Everything here runs on the GPU through
libcapyhip.so; inputs and outputs are host bytes.  A batch of 1 equals the reference's scalar call.
"""
import ctypes as C

from . import _lib as L


def _d(d):
    return int(getattr(d, "value", d))


def _same_len(items, what, expect=None):
    """All byte strings of one batch argument share a length (the C ABI takes one length per call)."""
    n = expect if expect is not None else (len(items[0]) if items else 0)
    if any(len(x) != n for x in items):
        raise ValueError("%s of one batch must all be %s bytes long" % (what, n if expect is not None else "equally many"))
    return n


def sha3_batch(msgs, d):
    """SHA3-d of each message (shake(), /root/reference/src/sha3/shake_functions.rs:24-32)."""
    d = _d(d)
    n = len(msgs)
    data, offs = L.pack(msgs)
    out = (C.c_uint8 * max(1, n * (d // 8 if d > 0 else 1)))()
    L.check(L.lib().capy_sha3_batch(d, n, data, offs, out))
    dl = d // 8
    raw = bytes(out)
    return [raw[i * dl:(i + 1) * dl] for i in range(n)]


def cshake_batch(xs, l_bits, n_str, s_str, d):
    """cshake(), /root/reference/src/sha3/shake_functions.rs:49-64."""
    d = _d(d)
    n = len(xs)
    data, offs = L.pack(xs)
    ol = l_bits // 8
    out = (C.c_uint8 * max(1, n * ol))()
    nb, sb = bytes(n_str), bytes(s_str)
    L.check(L.lib().capy_cshake_batch(d, n, data, offs, l_bits, L.buf(nb), len(nb), L.buf(sb), len(sb), out))
    raw = bytes(out)
    return [raw[i * ol:(i + 1) * ol] for i in range(n)]


def kmac_xof_batch(keys, xs, l_bits, s_str, d):
    """kmac_xof(), /root/reference/src/sha3/shake_functions.rs:79-89; all keys must have one length."""
    d = _d(d)
    n = len(keys)
    if len(xs) != n:
        raise ValueError("keys and messages differ in count")
    klen = len(keys[0]) if n else 0
    _same_len(keys, "keys")
    data, offs = L.pack(xs)
    ol = l_bits // 8
    out = (C.c_uint8 * max(1, n * ol))()
    sb = bytes(s_str)
    L.check(L.lib().capy_kmac_xof_batch(d, n, L.buf(b"".join(bytes(k) for k in keys)), klen, data, offs, l_bits,
                                        L.buf(sb), len(sb), out))
    raw = bytes(out)
    return [raw[i * ol:(i + 1) * ol] for i in range(n)]


def sha3_encrypt_batch(pws, zs, msgs, d):
    """sha3_encrypt(), /root/reference/src/sha3/encryptable.rs:29-45 -> (ciphertexts, tags)."""
    d = _d(d)
    n = len(msgs)
    plen = len(pws[0]) if n else 0
    _same_len(pws, "passwords")
    _same_len(zs, "nonces", 512)
    data, offs = L.pack(msgs)
    tags = (C.c_uint8 * max(1, 64 * n))()
    L.check(L.lib().capy_sha3_encrypt_batch(d, n, L.buf(b"".join(map(bytes, pws))), plen,
                                            L.buf(b"".join(map(bytes, zs))), data, offs, tags))
    raw, t = bytes(data), bytes(tags)
    return [raw[offs[i]:offs[i + 1]] for i in range(n)], [t[64 * i:64 * i + 64] for i in range(n)]


def sha3_decrypt_batch(pws, zs, cts, tags, d):
    """sha3_decrypt(), /root/reference/src/sha3/encryptable.rs:58-83 -> (messages, ok flags)."""
    d = _d(d)
    n = len(cts)
    plen = len(pws[0]) if n else 0
    _same_len(pws, "passwords")
    data, offs = L.pack(cts)
    status = (C.c_int32 * max(1, n))()
    L.check(L.lib().capy_sha3_decrypt_batch(d, n, L.buf(b"".join(map(bytes, pws))), plen,
                                            L.buf(b"".join(map(bytes, zs))), data, offs,
                                            L.buf(b"".join(map(bytes, tags))), status))
    raw = bytes(data)
    return [raw[offs[i]:offs[i + 1]] for i in range(n)], [status[i] == 0 for i in range(n)]


def kem_sponge_encrypt_batch(secrets, zs, msgs, d):
    """Sponge half of kem_encrypt(), /root/reference/src/kem/encryptable.rs:47-59 -> (ciphertexts, tags)."""
    d = _d(d)
    n = len(msgs)
    slen = len(secrets[0]) if n else 0
    _same_len(secrets, "secrets")
    _same_len(zs, "nonces", 512)
    data, offs = L.pack(msgs)
    tags = (C.c_uint8 * max(1, 64 * n))()
    L.check(L.lib().capy_kem_sponge_encrypt_batch(d, n, L.buf(b"".join(map(bytes, secrets))), slen,
                                                  L.buf(b"".join(map(bytes, zs))), data, offs, tags))
    raw, t = bytes(data), bytes(tags)
    return [raw[offs[i]:offs[i + 1]] for i in range(n)], [t[64 * i:64 * i + 64] for i in range(n)]


def kem_sponge_decrypt_batch(secrets, zs, cts, tags, d):
    """Sponge half of kem_decrypt(), /root/reference/src/kem/encryptable.rs:84-104 -> (messages, ok flags)."""
    d = _d(d)
    n = len(cts)
    slen = len(secrets[0]) if n else 0
    data, offs = L.pack(cts)
    status = (C.c_int32 * max(1, n))()
    L.check(L.lib().capy_kem_sponge_decrypt_batch(d, n, L.buf(b"".join(map(bytes, secrets))), slen,
                                                  L.buf(b"".join(map(bytes, zs))), data, offs,
                                                  L.buf(b"".join(map(bytes, tags))), status))
    raw = bytes(data)
    return [raw[offs[i]:offs[i + 1]] for i in range(n)], [status[i] == 0 for i in range(n)]


# ------------------------------------------------------------------ Ed448
def ed448_scalarmul_batch(scalars_be, points_xy):
    n = len(scalars_be)
    out = (C.c_uint8 * max(1, 112 * n))()
    L.check(L.lib().capy_ed448_scalarmul_batch(n, L.buf(b"".join(map(bytes, scalars_be))),
                                               L.buf(b"".join(map(bytes, points_xy))), out))
    raw = bytes(out)
    return [raw[112 * i:112 * i + 112] for i in range(n)]


def ed448_basemul_batch(scalars_be):
    n = len(scalars_be)
    out = (C.c_uint8 * max(1, 112 * n))()
    L.check(L.lib().capy_ed448_basemul_batch(n, L.buf(b"".join(map(bytes, scalars_be))), out))
    raw = bytes(out)
    return [raw[112 * i:112 * i + 112] for i in range(n)]


def ed448_add_batch(ps, qs):
    n = len(ps)
    out = (C.c_uint8 * max(1, 112 * n))()
    L.check(L.lib().capy_ed448_add_batch(n, L.buf(b"".join(map(bytes, ps))), L.buf(b"".join(map(bytes, qs))), out))
    raw = bytes(out)
    return [raw[112 * i:112 * i + 112] for i in range(n)]


def ed448_double_scalarmul_batch(a_be, b_be, points_xy):
    n = len(a_be)
    out = (C.c_uint8 * max(1, 112 * n))()
    L.check(L.lib().capy_ed448_double_scalarmul_batch(n, L.buf(b"".join(map(bytes, a_be))),
                                                      L.buf(b"".join(map(bytes, b_be))),
                                                      L.buf(b"".join(map(bytes, points_xy))), out))
    raw = bytes(out)
    return [raw[112 * i:112 * i + 112] for i in range(n)]


def keypair_batch(pws, d):
    d = _d(d)
    n = len(pws)
    plen = len(pws[0]) if n else 0
    _same_len(pws, "passwords")
    out = (C.c_uint8 * max(1, 112 * n))()
    L.check(L.lib().capy_keypair_batch(d, n, L.buf(b"".join(map(bytes, pws))), plen, out))
    raw = bytes(out)
    return [raw[112 * i:112 * i + 112] for i in range(n)]


def schnorr_sign_batch(pws, msgs, d):
    d = _d(d)
    n = len(msgs)
    plen = len(pws[0]) if n else 0
    _same_len(pws, "passwords")
    data, offs = L.pack(msgs)
    h = (C.c_uint8 * max(1, 56 * n))()
    z = (C.c_uint8 * max(1, 56 * n))()
    L.check(L.lib().capy_schnorr_sign_batch(d, n, L.buf(b"".join(map(bytes, pws))), plen, data, offs, h, z))
    hb, zb = bytes(h), bytes(z)
    return [(hb[56 * i:56 * i + 56], zb[56 * i:56 * i + 56]) for i in range(n)]


def schnorr_verify_batch(pubs, msgs, sigs, d):
    d = _d(d)
    n = len(msgs)
    data, offs = L.pack(msgs)
    status = (C.c_int32 * max(1, n))()
    L.check(L.lib().capy_schnorr_verify_batch(d, n, L.buf(b"".join(map(bytes, pubs))), data, offs,
                                              L.buf(b"".join(bytes(s[0]) for s in sigs)),
                                              L.buf(b"".join(bytes(s[1]) for s in sigs)), status))
    return [status[i] == 0 for i in range(n)]


def key_encrypt_batch(pubs, k_rands, msgs, d):
    d = _d(d)
    n = len(msgs)
    data, offs = L.pack(msgs)
    zxy = (C.c_uint8 * max(1, 112 * n))()
    tags = (C.c_uint8 * max(1, 56 * n))()
    L.check(L.lib().capy_key_encrypt_batch(d, n, L.buf(b"".join(map(bytes, pubs))),
                                           L.buf(b"".join(map(bytes, k_rands))), data, offs, zxy, tags))
    raw, zb, tb = bytes(data), bytes(zxy), bytes(tags)
    return ([raw[offs[i]:offs[i + 1]] for i in range(n)], [zb[112 * i:112 * i + 112] for i in range(n)],
            [tb[56 * i:56 * i + 56] for i in range(n)])


def key_decrypt_batch(pws, zxys, cts, tags, d):
    d = _d(d)
    n = len(cts)
    plen = len(pws[0]) if n else 0
    _same_len(pws, "passwords")
    data, offs = L.pack(cts)
    status = (C.c_int32 * max(1, n))()
    L.check(L.lib().capy_key_decrypt_batch(d, n, L.buf(b"".join(map(bytes, pws))), plen,
                                           L.buf(b"".join(map(bytes, zxys))), data, offs,
                                           L.buf(b"".join(map(bytes, tags))), status))
    raw = bytes(data)
    return [raw[offs[i]:offs[i + 1]] for i in range(n)], [status[i] == 0 for i in range(n)]
