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


def rfc8032_decode(b):
    """RFC 8032 §5.2.3 point decoding (57 bytes -> affine (x, y)), None if invalid."""
    y = int.from_bytes(b, "little")
    x0 = y >> 455
    y &= (1 << 455) - 1
    if y >= P:
        return None
    u = (y * y - 1) % P
    v = (D * y * y - 1) % P
    x = pow(u, 3, P) * v % P * pow(pow(u, 5, P) * pow(v, 3, P) % P, (P - 3) // 4, P) % P
    if (v * x * x - u) % P:
        return None
    if x == 0 and x0:
        return None
    if (x & 1) != x0:
        x = P - x
    return (x, y)


def _dom4(ctx):
    return b"SigEd448" + bytes([0, len(ctx)]) + ctx


def rfc8032_secret_scalar(sk):
    import hashlib

    h = hashlib.shake_256(sk).digest(114)
    a = bytearray(h[:57])
    a[0] &= 0xFC
    a[55] |= 0x80
    a[56] = 0
    return int.from_bytes(a, "little"), h[57:]


def rfc8032_challenge(sig_r, pk, msg, ctx=b""):
    """k = SHAKE256(dom4(0, ctx) || R || A || M, 114) mod L (RFC 8032 §5.2.6 / §5.2.7)."""
    import hashlib

    return int.from_bytes(hashlib.shake_256(_dom4(ctx) + sig_r + pk + msg).digest(114), "little") % R


def rfc8032_sign(sk, msg, ctx=b""):
    """Deterministic Ed448 signature, RFC 8032 §5.2.6."""
    import hashlib

    s, prefix = rfc8032_secret_scalar(sk)
    a_enc = rfc8032_encode(scalarmul(s, G))
    r = int.from_bytes(hashlib.shake_256(_dom4(ctx) + prefix + msg).digest(114), "little") % R
    r_enc = rfc8032_encode(scalarmul(r, G))
    k = rfc8032_challenge(r_enc, a_enc, msg, ctx)
    return r_enc + ((r + k * s) % R).to_bytes(57, "little")


def rfc8032_verify(pk, msg, sig, ctx=b"", mul=scalarmul, addp=add):
    """[S]B == R + [k]A (RFC 8032 §5.2.7, cofactorless form).  `mul` / `addp` let a test substitute the
    implementation under test for the group operations."""
    a_pt, r_pt = rfc8032_decode(pk), rfc8032_decode(sig[:57])
    s = int.from_bytes(sig[57:], "little")
    if a_pt is None or r_pt is None or s >= R:
        return False
    k = rfc8032_challenge(sig[:57], pk, msg, ctx)
    return mul(s, G) == addp(r_pt, mul(k, a_pt))


# --- RFC 7748: X448 by the Montgomery ladder (an algorithm independent of the Edwards group law above) ---
A24 = 39081


def x448_clamp(k_bytes):
    kb = bytearray(k_bytes)
    kb[0] &= 252
    kb[55] |= 128
    return int.from_bytes(kb, "little")


def x448_ladder(k, u):
    """u([k]Q) for Q = (u, .) on curve448 (or its twist), k a non-negative integer < 2^448; RFC 7748 §5."""
    x1 = u % P
    x2, z2, x3, z3, swap = 1, 0, x1, 1, 0
    for t in range(447, -1, -1):
        kt = (k >> t) & 1
        swap ^= kt
        if swap:
            x2, x3, z2, z3 = x3, x2, z3, z2
        swap = kt
        a = (x2 + z2) % P
        aa = a * a % P
        b = (x2 - z2) % P
        bb = b * b % P
        e = (aa - bb) % P
        c = (x3 + z3) % P
        d = (x3 - z3) % P
        da = d * a % P
        cb = c * b % P
        x3 = (da + cb) ** 2 % P
        z3 = x1 * (da - cb) ** 2 % P
        x2 = aa * bb % P
        z2 = e * (aa + A24 * e) % P
    if swap:
        x2, x3, z2, z3 = x3, x2, z3, z2
    return x2 * pow(z2, P - 2, P) % P


def x448(k_bytes, u_bytes):
    return x448_ladder(x448_clamp(k_bytes), int.from_bytes(u_bytes, "little")).to_bytes(56, "little")


def curve448_v(u):
    """v with v^2 = u^3 + 156326 u^2 + u, or None when u is on the twist."""
    rhs = (u * u % P * u + 156326 * u * u + u) % P
    v = pow(rhs, (P + 1) // 4, P)
    return v if v * v % P == rhs else None


def edwards_from_curve448(u, v):
    """RFC 7748 §4.2, the map curve448 -> edwards448.  Together with x448_u_from_edwards it composes to
    multiplication by 4 (they are a 4-isogeny and its dual): u(E->M(M->E(Q))) = u([4]Q)."""
    x = 4 * v * (u * u - 1) % P * pow((pow(u, 4, P) - 2 * u * u + 4 * v * v + 1) % P, -1, P) % P
    y = -(pow(u, 5, P) - 2 * pow(u, 3, P) - 4 * u * v * v + u) % P * pow(
        (pow(u, 5, P) - 2 * u * u * v * v - 2 * pow(u, 3, P) - 2 * v * v + u) % P, -1, P) % P
    return (x, y)


def x448_via_edwards(k_bytes, u_bytes, mul=scalarmul):
    """X448(k, u) computed with Edwards VARIABLE-BASE multiplication: lift Q = (u, v) to P' = M->E(Q), multiply by
    k/4 (the clamped scalar is a multiple of 4), map back: E->M([k/4]P') = [k/4][4]Q = [k]Q.  None on the twist."""
    u = int.from_bytes(u_bytes, "little") % P
    v = curve448_v(u)
    if v is None or u in (0, 1, P - 1):
        return None
    k = x448_clamp(k_bytes)
    out = mul(k // 4, edwards_from_curve448(u, v))
    if out[0] == 0:
        return (0).to_bytes(56, "little")
    return x448_u_from_edwards(out).to_bytes(56, "little")


def schnorr_scalars(kb, h, s, star=0):
    """The scalar arithmetic of Signable::sign (/root/reference/src/ecc/signable.rs:46,54) for the three readings of the
    curve crate's `*` / `-` on a value that is not reduced mod r (include/capyhip.h: capy_ed448_set_scalar_star):
    returns (k used for U = [k]G, z).  kb, h, s are integers (s < r)."""
    hs = h * s % R
    if star == 0:
        k = 4 * kb % R
        return k, (k - hs) % R
    k = 4 * kb % 2**448
    if star == 1:
        z = k - hs
        return k, z + R if z < 0 else z
    return k, (k - hs) % R
