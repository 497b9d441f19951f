"""ed448_ref.py — CPU ORACLE, Ed448 half, python big-int model (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

The reference's curve arithmetic is the un-vendored crate tiny_ed448_goldilocks 0.1.8
(/root/reference/Cargo.toml:22, Cargo.lock:857-869), absent from /root/reference, and the
reference holds no known-answer vector for it ("parity unpinned").  This
model restates the *published* curve: untwisted Edwards x^2 + y^2 = 1 + d x^2 y^2 over
p = 2^448 - 2^224 - 1 with d = -39081 (RFC 7748 §4.2 "edwards448", RFC 8032 §5.2), and is
pinned by the RFC 8032 §7.4 / RFC 7748 §6.2 public-key vectors in tests/test_oracle_ed448.py.

Byte conventions at the reference boundary (call sites in SURVEY.md §8a row 21):
  scalars   56-byte BIG-endian, no reduction   (src/sha3/aux_functions.rs:102-110)
  field el. 56-byte little-endian canonical    (assumption (ii))
"""

P = 2**448 - 2**224 - 1
D = (-39081) % P
R = 2**446 - 13818066809895115352007386748515426880336692474882178609894547503885
GX = 0x4F1970C66BED0DED221D15A622BF36DA9E146570470F1767EA6DE324A3D3A46412AE1AF72AB66511433B80E18B00938E2626A82BC70CC05E
GY = 0x693F46716EB6BC248876203756C9C7624BEA73736CA3984087789C1E05A0C2D73AD3FF1CE67C39C4FDBD132C4ED7C8AD9808795BF230FA14
G = (GX, GY)
IDENT = (0, 1)


def on_curve(pt):
    x, y = pt
    return (x * x + y * y - 1 - D * x * x * y * y) % P == 0


def add(p1, p2):
    """Complete affine Edwards addition (d is a non-square, so denominators never vanish)."""
    x1, y1 = p1
    x2, y2 = p2
    t = D * x1 * x2 % P * y1 % P * y2 % P
    x3 = (x1 * y2 + y1 * x2) * pow(1 + t, -1, P) % P
    y3 = (y1 * y2 - x1 * x2) * pow(1 - t, -1, P) % P
    return (x3, y3)


def _ext_add(a, b):
    X1, Y1, Z1, T1 = a
    X2, Y2, Z2, T2 = b
    A = X1 * X2 % P
    B = Y1 * Y2 % P
    C = D * T1 % P * T2 % P
    Dd = Z1 * Z2 % P
    E = ((X1 + Y1) * (X2 + Y2) - A - B) % P
    F = (Dd - C) % P
    Gg = (Dd + C) % P
    H = (B - A) % P
    return (E * F % P, Gg * H % P, F * Gg % P, E * H % P)


def scalarmul(k, pt):
    """[k]pt by the plain group law, k any non-negative integer (no reduction mod r)."""
    acc = (0, 1, 1, 0)
    base = (pt[0], pt[1], 1, pt[0] * pt[1] % P)
    for bit in bin(k)[2:] if k else "":
        acc = _ext_add(acc, acc)
        if bit == "1":
            acc = _ext_add(acc, base)
    zi = pow(acc[2], -1, P)
    return (acc[0] * zi % P, acc[1] * zi % P)


def fe_to_bytes(x):
    return int(x % P).to_bytes(56, "little")


def fe_from_bytes(b):
    return int.from_bytes(b, "little")


def pt_to_bytes(pt):
    return fe_to_bytes(pt[0]) + fe_to_bytes(pt[1])


def pt_from_bytes(b):
    return (fe_from_bytes(b[:56]), fe_from_bytes(b[56:112]))


def sc_from_bytes(b):
    return int.from_bytes(b, "big")


def sc_to_bytes(k):
    return int(k).to_bytes(56, "big")


# --- RFC 8032 / RFC 7748 helpers used only to pin the model against published vectors ---
def rfc8032_encode(pt):
    x, y = pt
    b = bytearray(int(y).to_bytes(57, "little"))
    b[56] |= (x & 1) << 7
    return bytes(b)


def rfc8032_pubkey(sk):
    import hashlib

    h = bytearray(hashlib.shake_256(sk).digest(114)[:57])
    h[0] &= 0xFC
    h[55] |= 0x80
    h[56] = 0
    s = int.from_bytes(h, "little")
    return rfc8032_encode(scalarmul(s, G))


def x448_u_from_edwards(pt):
    """RFC 7748 §4.2 birational map edwards448 -> curve448: u = y^2 / x^2."""
    x, y = pt
    return y * y % P * pow(x * x % P, -1, P) % P
