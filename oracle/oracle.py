"""oracle.py — ctypes front-end of the CPU ORACLE (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Builds oracle/libcapyoracle.so with gcc on first use if it is missing.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcapyoracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("oracle_sponge.c", "keccak_inplace.c", "oracle_ed448.c", "capy_oracle.h")]
    if force or not os.path.exists(_SO) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs
    ):
        subprocess.check_call(["make", "-C", _HERE, "libcapyoracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_left_encode.restype = C.c_size_t
        _lib.oracle_right_encode.restype = C.c_size_t
        _lib.oracle_byte_pad.restype = C.c_size_t
        _lib.oracle_encode_string.restype = C.c_size_t
    return _lib


def _buf(b):
    return (C.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if len(b) else b"\0")


def keccakf1600(state, inplace=False):
    """keccak-f[1600]; inplace=True: the reference's in-place four-rounds-per-trip form (keccak_inplace.c)."""
    arr = (C.c_uint64 * 25)(*state)
    (lib().oracle_keccakf1600_inplace if inplace else lib().oracle_keccakf1600)(arr)
    return list(arr)


def select_keccak(inplace):
    """Which form of the permutation the sponge functions run on (see oracle_sponge.c: oracle_select_keccak)."""
    lib().oracle_select_keccak(1 if inplace else 0)


def left_encode(v):
    out = (C.c_uint8 * 9)()
    n = lib().oracle_left_encode(C.c_uint64(v), out)
    return bytes(out[:n])


def right_encode(v, quirks=1):
    out = (C.c_uint8 * 9)()
    n = lib().oracle_right_encode(C.c_uint64(v), out, quirks)
    return bytes(out[:n])


def byte_pad(x, w, quirks=1):
    out = (C.c_uint8 * (len(x) + 9 + 2 * w))()
    n = lib().oracle_byte_pad(_buf(x), C.c_size_t(len(x)), C.c_uint32(w), out, quirks)
    return bytes(out[:n])


def encode_string(s):
    out = (C.c_uint8 * (len(s) + 9))()
    n = lib().oracle_encode_string(_buf(s), C.c_size_t(len(s)), out)
    return bytes(out[:n])


def sha3(msg, d, quirks=1, want_padded=False):
    out = (C.c_uint8 * (d // 8))()
    padded = (C.c_uint8 * (len(msg) + 400))()
    plen = C.c_size_t(0)
    rc = lib().oracle_sha3(_buf(msg), C.c_size_t(len(msg)), d, quirks, out, padded, C.byref(plen))
    if rc:
        raise ValueError("unsupported security parameter %r" % d)
    if want_padded:
        return bytes(out), bytes(padded[: plen.value])
    return bytes(out)


def cshake(x, l_bits, n, s, d, quirks=1):
    out = (C.c_uint8 * max(1, l_bits // 8))()
    rc = lib().oracle_cshake(_buf(x), C.c_size_t(len(x)), C.c_size_t(l_bits), _buf(n), C.c_size_t(len(n)),
                             _buf(s), C.c_size_t(len(s)), d, quirks, out)
    if rc:
        raise ValueError("unsupported security parameter %r" % d)
    return bytes(out[: l_bits // 8])


def kmac_xof(k, x, l_bits, s, d, quirks=1):
    out = (C.c_uint8 * max(1, l_bits // 8))()
    rc = lib().oracle_kmac_xof(_buf(k), C.c_size_t(len(k)), _buf(x), C.c_size_t(len(x)), C.c_size_t(l_bits),
                               _buf(s), C.c_size_t(len(s)), d, quirks, out)
    if rc:
        raise ValueError("unsupported security parameter %r" % d)
    return bytes(out[: l_bits // 8])


def sha3_encrypt(pw, z, msg, d, quirks=1):
    assert len(z) == 512
    m = (C.c_uint8 * max(1, len(msg))).from_buffer_copy(bytes(msg) if len(msg) else b"\0")
    tag = (C.c_uint8 * 64)()
    lib().oracle_sha3_encrypt(_buf(pw), C.c_size_t(len(pw)), _buf(z), m, C.c_size_t(len(msg)), d, quirks, tag)
    return bytes(m[: len(msg)]), bytes(tag)


def sha3_decrypt(pw, z, ct, tag, d, quirks=1):
    m = (C.c_uint8 * max(1, len(ct))).from_buffer_copy(bytes(ct) if len(ct) else b"\0")
    bad = lib().oracle_sha3_decrypt(_buf(pw), C.c_size_t(len(pw)), _buf(z), m, C.c_size_t(len(ct)), d, quirks,
                                    _buf(tag))
    return bytes(m[: len(ct)]), bad == 0


# ---------------------------------------------------------------- Ed448
def ed448_generator():
    out = (C.c_uint8 * 112)()
    lib().oracle_ed448_generator(out)
    return bytes(out)


def ed448_scalarmul(scalar_be, p_xy):
    out = (C.c_uint8 * 112)()
    lib().oracle_ed448_scalarmul(_buf(scalar_be), _buf(p_xy), out)
    return bytes(out)


def ed448_set_generator(xy=None):
    """The generator of every fixed-base multiplication of the oracle (key pairs, signatures, ECDHIES): a candidate from
    tests/golden/ed448_generator_candidates.json, or None for the RFC 8032 base point (oracle_ed448.c: oracle_ed448_set_generator)."""
    lib().oracle_ed448_set_generator(_buf(xy) if xy is not None else None)


def ed448_basemul(scalar_be):
    out = (C.c_uint8 * 112)()
    lib().oracle_ed448_basemul(_buf(scalar_be), out)
    return bytes(out)


def ed448_add(p_xy, q_xy):
    out = (C.c_uint8 * 112)()
    lib().oracle_ed448_add(_buf(p_xy), _buf(q_xy), out)
    return bytes(out)


def ed448_on_curve(p_xy):
    return bool(lib().oracle_ed448_on_curve(_buf(p_xy)))


def sc448_mul_mod(a, b):
    out = (C.c_uint8 * 56)()
    lib().oracle_sc448_mul_mod(_buf(a), _buf(b), out)
    return bytes(out)


def sc448_sub_mod(a, b):
    out = (C.c_uint8 * 56)()
    lib().oracle_sc448_sub_mod(_buf(a), _buf(b), out)
    return bytes(out)


def sc448_reduce(a):
    out = (C.c_uint8 * 56)()
    lib().oracle_sc448_reduce(_buf(a), out)
    return bytes(out)


def keypair_pub(pw, d):
    out = (C.c_uint8 * 112)()
    lib().oracle_keypair_pub(_buf(pw), C.c_size_t(len(pw)), d, out)
    return bytes(out)


def set_scalar_star(mode):
    """Reading of `Scalar * Scalar` in sign (oracle_ed448.c: oracle_set_scalar_star)."""
    lib().oracle_set_scalar_star(int(mode))


def sign(pw, msg, d):
    h = (C.c_uint8 * 56)()
    z = (C.c_uint8 * 56)()
    lib().oracle_sign(_buf(pw), C.c_size_t(len(pw)), _buf(msg), C.c_size_t(len(msg)), d, h, z)
    return bytes(h), bytes(z)


def verify(pub_xy, msg, d, h, z):
    return lib().oracle_verify(_buf(pub_xy), _buf(msg), C.c_size_t(len(msg)), d, _buf(h), _buf(z)) == 0


def key_encrypt(pub_xy, k_rand, msg, d):
    m = (C.c_uint8 * max(1, len(msg))).from_buffer_copy(bytes(msg) if len(msg) else b"\0")
    zxy = (C.c_uint8 * 112)()
    tag = (C.c_uint8 * 56)()
    lib().oracle_key_encrypt(_buf(pub_xy), _buf(k_rand), m, C.c_size_t(len(msg)), d, zxy, tag)
    return bytes(m[: len(msg)]), bytes(zxy), bytes(tag)


def key_decrypt(pw, z_xy, ct, tag, d):
    m = (C.c_uint8 * max(1, len(ct))).from_buffer_copy(bytes(ct) if len(ct) else b"\0")
    bad = lib().oracle_key_decrypt(_buf(pw), C.c_size_t(len(pw)), _buf(z_xy), m, C.c_size_t(len(ct)), d, _buf(tag))
    return bytes(m[: len(ct)]), bad == 0
