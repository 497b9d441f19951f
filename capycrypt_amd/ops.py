"""Batched operators over the C ABI (include/capyhip.h).  Everything here runs on the GPU through
libcapyhip.so; inputs and outputs are host bytes.  A batch of 1 equals the reference's scalar call.

Every argument is checked here before a pointer reaches the library: item counts must agree and fixed-size
fields (512-byte nonces, 64/56-byte tags, 56-byte scalars, 112-byte points) must have exactly that size
(ValueError otherwise) -- the C ABI takes plain pointers and would read past a short buffer.  Keys and
passwords may differ in length per item, as in the reference (any &[u8] per message).
"""
import ctypes as C

from . import _lib as L


def _d(d):
    return int(getattr(d, "value", d))


def _count(n, items, what):
    if len(items) != n:
        raise ValueError("%s: %d given for a batch of %d" % (what, len(items), n))


def _fixed(n, items, size, what):
    """n byte strings of exactly `size` bytes, joined."""
    _count(n, items, what)
    items = [bytes(x) for x in items]
    for x in items:
        if len(x) != size:
            raise ValueError("%s must be %d bytes long, got %d" % (what, size, len(x)))
    return L.buf(b"".join(items))


def _keys(n, keys, what):
    """n keys / passwords of any lengths -> (buffer, fixed length, offsets or None) as the C ABI takes them:
    equal lengths go as one key_len (the kernels' uniform path), ragged lengths as n+1 offsets."""
    _count(n, keys, what)
    keys = [bytes(k) for k in keys]
    if any(len(k) > (1 << 20) for k in keys):
        raise ValueError("%s longer than 1 MiB are not supported" % what)
    klen = len(keys[0]) if n else 0
    if all(len(k) == klen for k in keys):
        return L.buf(b"".join(keys)), klen, None
    data, offs = L.pack(keys)
    return data, 0, offs


def _rows(raw, n, size):
    return [raw[size * i:size * i + size] for i in range(n)]


def sha3_batch(msgs, d):
    """SHA3-d of each message (shake(), /root/reference/src/sha3/shake_functions.rs:24-32)."""
    d = _d(d)
    n = len(msgs)
    data, offs = L.pack(msgs)
    out = (C.c_uint8 * max(1, n * (d // 8 if d > 0 else 1)))()
    L.check(L.lib().capy_sha3_batch(d, n, data, offs, out))
    return _rows(bytes(out), n, d // 8)


def cshake_batch(xs, l_bits, n_str, s_str, d):
    """cshake(), /root/reference/src/sha3/shake_functions.rs:49-64."""
    d = _d(d)
    n = len(xs)
    data, offs = L.pack(xs)
    ol = l_bits // 8
    out = (C.c_uint8 * max(1, n * ol))()
    nb, sb = bytes(n_str), bytes(s_str)
    L.check(L.lib().capy_cshake_batch(d, n, data, offs, l_bits, L.buf(nb), len(nb), L.buf(sb), len(sb), out))
    return _rows(bytes(out), n, ol)


def kmac_xof_batch(keys, xs, l_bits, s_str, d):
    """kmac_xof(), /root/reference/src/sha3/shake_functions.rs:79-89; one key per message, any lengths."""
    d = _d(d)
    n = len(xs)
    kbuf, klen, koffs = _keys(n, keys, "keys")
    data, offs = L.pack(xs)
    ol = l_bits // 8
    out = (C.c_uint8 * max(1, n * ol))()
    sb = bytes(s_str)
    L.check(L.lib().capy_kmac_xof_batch(d, n, kbuf, klen, koffs, data, offs, l_bits, L.buf(sb), len(sb), out))
    return _rows(bytes(out), n, ol)


def _sym_encrypt(fn, keys, zs, msgs, d, what, per_item=True):
    d = _d(d)
    n = len(msgs)
    kbuf, klen, koffs = _keys(n, keys, what)
    zbuf = _fixed(n, zs, 512, "nonces")
    data, offs = L.pack(msgs)
    tags = (C.c_uint8 * max(1, 64 * n))()
    if not per_item:  # the KEM entry points take one secret length per call
        L.check(fn(d, n, kbuf, klen, zbuf, data, offs, tags))
    else:
        L.check(fn(d, n, kbuf, klen, koffs, zbuf, data, offs, tags))
    raw = bytes(data)
    return [raw[offs[i]:offs[i + 1]] for i in range(n)], _rows(bytes(tags), n, 64)


def _sym_decrypt(fn, keys, zs, cts, tags, d, what, per_item=True):
    d = _d(d)
    n = len(cts)
    kbuf, klen, koffs = _keys(n, keys, what)
    zbuf = _fixed(n, zs, 512, "nonces")
    tbuf = _fixed(n, tags, 64, "tags")
    data, offs = L.pack(cts)
    status = (C.c_int32 * max(1, n))()
    if not per_item:
        L.check(fn(d, n, kbuf, klen, zbuf, data, offs, tbuf, status))
    else:
        L.check(fn(d, n, kbuf, klen, koffs, zbuf, data, offs, tbuf, status))
    raw = bytes(data)
    return [raw[offs[i]:offs[i + 1]] for i in range(n)], [status[i] == 0 for i in range(n)]


def sha3_encrypt_batch(pws, zs, msgs, d):
    """sha3_encrypt(), /root/reference/src/sha3/encryptable.rs:29-45 -> (ciphertexts, tags)."""
    return _sym_encrypt(L.lib().capy_sha3_encrypt_batch, pws, zs, msgs, d, "passwords")


def sha3_decrypt_batch(pws, zs, cts, tags, d):
    """sha3_decrypt(), /root/reference/src/sha3/encryptable.rs:58-83 -> (messages, ok flags)."""
    return _sym_decrypt(L.lib().capy_sha3_decrypt_batch, pws, zs, cts, tags, d, "passwords")


def _kem_secrets(n, secrets):
    _count(n, secrets, "secrets")
    slen = len(secrets[0]) if n else 0
    if any(len(s) != slen for s in secrets):
        raise ValueError("secrets of one batch must all have one length (the ML-KEM shared secret)")
    return secrets


def kem_sponge_encrypt_batch(secrets, zs, msgs, d):
    """Sponge half of kem_encrypt(), /root/reference/src/kem/encryptable.rs:47-59 -> (ciphertexts, tags)."""
    return _sym_encrypt(L.lib().capy_kem_sponge_encrypt_batch, _kem_secrets(len(msgs), secrets), zs, msgs, d, "secrets",
                        per_item=False)


def kem_sponge_decrypt_batch(secrets, zs, cts, tags, d):
    """Sponge half of kem_decrypt(), /root/reference/src/kem/encryptable.rs:84-104 -> (messages, ok flags)."""
    return _sym_decrypt(L.lib().capy_kem_sponge_decrypt_batch, _kem_secrets(len(cts), secrets), zs, cts, tags, d,
                        "secrets", per_item=False)


# ------------------------------------------------------------------ Ed448
def ed448_scalarmul_batch(scalars_be, points_xy, options=None):
    n = len(scalars_be)
    out = (C.c_uint8 * max(1, 112 * n))()
    _call("capy_ed448_scalarmul_batch", options, n, _fixed(n, scalars_be, 56, "scalars"), _fixed(n, points_xy, 112, "points"), out)
    return _rows(bytes(out), n, 112)


def ed448_basemul_batch(scalars_be, options=None):
    n = len(scalars_be)
    out = (C.c_uint8 * max(1, 112 * n))()
    _call("capy_ed448_basemul_batch", options, n, _fixed(n, scalars_be, 56, "scalars"), out)
    return _rows(bytes(out), n, 112)


def ed448_add_batch(ps, qs):
    n = len(ps)
    out = (C.c_uint8 * max(1, 112 * n))()
    L.check(L.lib().capy_ed448_add_batch(n, _fixed(n, ps, 112, "points"), _fixed(n, qs, 112, "points"), out))
    return _rows(bytes(out), n, 112)


def ed448_double_scalarmul_batch(a_be, b_be, points_xy):
    n = len(a_be)
    out = (C.c_uint8 * max(1, 112 * n))()
    L.check(L.lib().capy_ed448_double_scalarmul_batch(n, _fixed(n, a_be, 56, "scalars"), _fixed(n, b_be, 56, "scalars"),
                                                      _fixed(n, points_xy, 112, "points"), out))
    return _rows(bytes(out), n, 112)


def ed448_validate_batch(points_xy):
    """True per point iff both coordinates are canonical (< p) and the point is on the curve."""
    n = len(points_xy)
    status = (C.c_int32 * max(1, n))()
    L.check(L.lib().capy_ed448_validate_batch(n, _fixed(n, points_xy, 112, "points"), status))
    return [status[i] == 0 for i in range(n)]


def ed448_set_scalar_star(mode):
    """Reading of the curve crate's `Scalar * Scalar` in Signable::sign (include/capyhip.h: capy_ed448_set_scalar_star):
    0 product mod r (default), 1 wrapping at 2^448 with crypto-bigint's sub_mod, 2 wrapping with a reducing subtraction."""
    L.check(L.lib().capy_ed448_set_scalar_star(int(mode)))


HARDEN_OFF, HARDEN_ALL, HARDEN_PROTOCOL = L.CAPY_HARDEN_OFF, L.CAPY_HARDEN_ALL, L.CAPY_HARDEN_PROTOCOL
CallOptions = L.CallOptions  # per-call options (capy_call_options): hardened, scalar_star, generator, stream


def _call(name, options, *args):
    """name(*args), or name + "_ex"(*args, &options) when per-call options are given (include/capyhip.h, r04)."""
    if options is None:
        L.check(getattr(L.lib(), name)(*args))
    else:
        if not isinstance(options, L.CallOptions):
            raise TypeError("options must be a capycrypt_amd.ops.CallOptions")
        L.check(getattr(L.lib(), name + "_ex")(*args, C.byref(options)))


def ed448_generator_create(xy):
    """Register a further generator (a point of the prime order r) -> handle for CallOptions(generator=handle)."""
    if len(bytes(xy)) != 112:
        raise ValueError("the generator is 112 bytes (affine x || y, little-endian)")
    h = C.c_int(-1)
    L.check(L.lib().capy_ed448_generator_create(L.buf(xy), C.byref(h)))
    return h.value


def ed448_set_hardened(mode):
    """Constant-address table lookups (include/capyhip.h: capy_ed448_set_hardened), process-wide default: HARDEN_OFF,
    HARDEN_PROTOCOL (the default: secret scalars of the protocol calls) or HARDEN_ALL (raw scalarmul / basemul calls too).
    A single call chooses for itself with options=CallOptions(hardened=...) (the *_ex entry points)."""
    L.check(L.lib().capy_ed448_set_hardened(int(mode)))


def ed448_set_generator(xy=None):
    """Replace the point that stands for ExtendedPoint::generator() (None: back to the RFC 8032 base point)."""
    if xy is not None and len(bytes(xy)) != 112:
        raise ValueError("the generator is 112 bytes (affine x || y, little-endian)")
    L.check(L.lib().capy_ed448_set_generator(None if xy is None else L.buf(xy)))


def ed448_get_generator():
    out = (C.c_uint8 * 112)()
    L.check(L.lib().capy_ed448_get_generator(out))
    return bytes(out)


def keypair_batch(pws, d, options=None):
    """KeyPair::new, /root/reference/src/ecc/keypair.rs:41-51 -> public keys; one password per key, any lengths."""
    d = _d(d)
    n = len(pws)
    pbuf, plen, poffs = _keys(n, pws, "passwords")
    out = (C.c_uint8 * max(1, 112 * n))()
    _call("capy_keypair_batch", options, d, n, pbuf, plen, poffs, out)
    return _rows(bytes(out), n, 112)


def schnorr_sign_batch(pws, msgs, d, options=None):
    """Signable::sign, /root/reference/src/ecc/signable.rs:40-57 -> [(h, z)]."""
    d = _d(d)
    n = len(msgs)
    pbuf, plen, poffs = _keys(n, pws, "passwords")
    data, offs = L.pack(msgs)
    h = (C.c_uint8 * max(1, 56 * n))()
    z = (C.c_uint8 * max(1, 56 * n))()
    _call("capy_schnorr_sign_batch", options, d, n, pbuf, plen, poffs, data, offs, h, z)
    return list(zip(_rows(bytes(h), n, 56), _rows(bytes(z), n, 56)))


def schnorr_verify_batch(pubs, msgs, sigs, d, options=None):
    """Signable::verify, /root/reference/src/ecc/signable.rs:72-86 -> ok flags."""
    d = _d(d)
    n = len(msgs)
    _count(n, sigs, "signatures")
    data, offs = L.pack(msgs)
    status = (C.c_int32 * max(1, n))()
    _call("capy_schnorr_verify_batch", options, d, n, _fixed(n, pubs, 112, "public keys"), data, offs,
          _fixed(n, [s[0] for s in sigs], 56, "signature hashes"), _fixed(n, [s[1] for s in sigs], 56, "signature scalars"), status)
    return [status[i] == 0 for i in range(n)]


def key_encrypt_batch(pubs, k_rands, msgs, d, options=None):
    """KeyEncryptable::key_encrypt, /root/reference/src/ecc/encryptable.rs:34-50 -> (ciphertexts, Z points, tags)."""
    d = _d(d)
    n = len(msgs)
    data, offs = L.pack(msgs)
    zxy = (C.c_uint8 * max(1, 112 * n))()
    tags = (C.c_uint8 * max(1, 56 * n))()
    _call("capy_key_encrypt_batch", options, d, n, _fixed(n, pubs, 112, "public keys"), _fixed(n, k_rands, 56, "nonces"), data, offs,
          zxy, tags)
    raw = bytes(data)
    return ([raw[offs[i]:offs[i + 1]] for i in range(n)], _rows(bytes(zxy), n, 112), _rows(bytes(tags), n, 56))


def key_decrypt_batch(pws, zxys, cts, tags, d, options=None):
    """KeyEncryptable::key_decrypt, /root/reference/src/ecc/encryptable.rs:72-94 -> (messages, ok flags)."""
    d = _d(d)
    n = len(cts)
    pbuf, plen, poffs = _keys(n, pws, "passwords")
    data, offs = L.pack(cts)
    status = (C.c_int32 * max(1, n))()
    _call("capy_key_decrypt_batch", options, d, n, pbuf, plen, poffs, _fixed(n, zxys, 112, "nonce points"), data, offs,
          _fixed(n, tags, 56, "tags"), status)
    raw = bytes(data)
    return [raw[offs[i]:offs[i + 1]] for i in range(n)], [status[i] == 0 for i in range(n)]
