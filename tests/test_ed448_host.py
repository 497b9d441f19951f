"""CPU: the field / point / scalar code the HIP kernels run (capycrypt_amd/csrc/ed448_dev.h, ed448_algo.h are
__host__ __device__) compiled for the host and checked against the oracle.  Needs hipcc (present here and on
the GPU box); no GPU."""
import ctypes as C
import os
import random
import shutil
import subprocess

import pytest

from oracle import ed448_ref as E

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "ed448_host_test.cpp")
# CAPY_ED448_TEST_DEFS="-DCAPY_ED448_KARATSUBA=1 ..." builds (and tests) a variant of the device code
DEFS = os.environ.get("CAPY_ED448_TEST_DEFS", "").split()
SO = os.path.join(HERE, "native", "libed448host%s.so" % ("_variant" if DEFS else ""))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def L():
    deps = [SRC] + [os.path.join(HERE, "..", "capycrypt_amd", "csrc", f) for f in ("ed448_dev.h", "ed448_algo.h")]
    if DEFS or not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        if not os.path.exists(HIPCC):
            pytest.skip("hipcc not available")
        subprocess.check_call([HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950"] + DEFS +
                              ["-o", SO, SRC])
    return C.CDLL(SO)


def call(L, fn, *args, outlen=56):
    out = (C.c_uint8 * outlen)()
    getattr(L, fn)(*[C.c_char_p(a) if isinstance(a, bytes) else a for a in args], out)
    return bytes(out)


def fb(x):
    return int(x % E.P).to_bytes(56, "little")


def test_field_ops(L):
    rng = random.Random(5)
    edge = [0, 1, E.P - 1, E.P, E.P + 1, 2**448 - 1, 2**224, 2**224 - 1, E.P - 2, 2]
    for i in range(200):
        a = edge[i] if i < len(edge) else rng.getrandbits(448)
        b = rng.getrandbits(448)
        ab, bb = a.to_bytes(56, "little"), b.to_bytes(56, "little")
        assert call(L, "ht_fe_mul", ab, bb) == fb(a * b)
        assert call(L, "ht_fe_sqr", ab) == fb(a * a)
        assert call(L, "ht_fe_add", ab, bb) == fb(a + b)
        assert call(L, "ht_fe_sub", ab, bb) == fb(a - b)
        assert call(L, "ht_fe_roundtrip", ab) == fb(a)
    for _ in range(8):
        a = rng.getrandbits(448) % E.P or 1
        assert call(L, "ht_fe_inv", a.to_bytes(56, "little")) == fb(pow(a, -1, E.P))


def test_division_step_inversion_equals_fermat(L):
    """fe_inv_gcd (Bernstein-Yang division steps, 44 x 30 steps for every input) against pow(a, -1, p) and against the
    a^(p-2) chain it replaces: zero (-> 0, like the chain), one, p - 1, powers of two and their neighbours, values with
    long runs of zeros / ones, non-canonical encodings (>= p), random values.  No input may need more than the 44 rounds
    the code runs (the paper's bound for 448-bit inputs is 1294 steps = 43.1 rounds)."""
    rng = random.Random(0x6CD)
    vals = [0, 1, 2, 3, E.P - 1, E.P - 2, E.P, E.P + 1, 2**448 - 1, 2**224, 2**224 - 1, 2**224 + 1, 2**447, 2**447 - 1, (E.P - 1) // 2,
            (E.P + 1) // 2, 2**446 - 1, 39081, E.P - 39081]
    vals += [2**k for k in range(1, 448, 13)] + [2**k - 1 for k in range(2, 448, 17)] + [E.P - 2**k for k in range(1, 447, 19)]
    vals += [rng.getrandbits(448) & ~((1 << rng.randrange(1, 440)) - 1) for _ in range(60)]
    vals += [rng.getrandbits(rng.randrange(1, 449)) for _ in range(300)]
    vals += [rng.getrandbits(448) for _ in range(600)]
    L.ht_fe_inv_gcd_rounds.restype = C.c_int
    worst = 0
    for a in vals:
        ab = a.to_bytes(56, "little")
        want = fb(pow(a, -1, E.P)) if a % E.P else bytes(56)
        assert call(L, "ht_fe_inv_gcd", ab) == want, hex(a)
        worst = max(worst, L.ht_fe_inv_gcd_rounds(C.c_char_p(ab)))
    for a in vals[:40]:
        ab = a.to_bytes(56, "little")
        assert call(L, "ht_fe_inv_gcd", ab) == call(L, "ht_fe_inv", ab)
    assert worst <= 44, worst


def test_twisted_curve_additions_and_the_way_back(L):
    """The fixed base accumulates on the 4-isogenous twisted curve (7M mixed additions) and comes back through the dual
    isogeny: phi^(sum of phi(P_i)) = 4 sum P_i.  Random subgroup points, with the negation handling of the kernels
    (an entry for -P used negated), sums that pass through the identity, and the two-results-one-inversion form."""
    rng = random.Random(0x150)
    ident = (0, 1)
    for n in (1, 2, 3, 7, 20):
        pts = [E.scalarmul(rng.getrandbits(446), E.G) for _ in range(n)]
        if n == 2:
            pts[1] = E.pt_neg(pts[0]) if hasattr(E, "pt_neg") else ((-pts[0][0]) % E.P, pts[0][1])  # sum = identity
        want = ident
        for q in pts:
            want = E.add(want, q)
        want = E.scalarmul(4, want) if want != ident else ident
        blob = b"".join(E.pt_to_bytes(q) for q in pts)
        out = (C.c_uint8 * 112)()
        L.ht_tw_sum(C.c_char_p(blob), n, out)
        assert bytes(out) == E.pt_to_bytes(want), n
    p, q = E.scalarmul(rng.getrandbits(446), E.G), E.scalarmul(rng.getrandbits(446), E.G)
    out = (C.c_uint8 * 224)()
    L.ht_tw_pair(C.c_char_p(E.pt_to_bytes(p)), C.c_char_p(E.pt_to_bytes(q)), out)
    assert bytes(out[:112]) == E.pt_to_bytes(E.scalarmul(4, p)) and bytes(out[112:]) == E.pt_to_bytes(E.scalarmul(8, q))


def test_lazy_reduction_chain(L):
    rng = random.Random(6)
    for _ in range(10):
        a, b = rng.getrandbits(448), rng.getrandbits(448)
        x, y = a % E.P, b % E.P
        for _i in range(50):
            t = x * y % E.P
            u = (t - x) ** 2 % E.P
            x, y = (u - y + t * 39081) % E.P, (-t - 2 * u) % E.P
        out = (C.c_uint8 * 56)()
        L.ht_fe_chain(a.to_bytes(56, "little"), b.to_bytes(56, "little"), 50, out)
        assert bytes(out) == fb(x + y)


def test_scalarmul_algorithms(L):
    rng = random.Random(7)
    G = E.pt_to_bytes(E.G)
    L.ht_build_gtab(C.c_char_p(G))
    specials = [0, 1, 2, 8, E.R, 2**448 - 1, int("8" * 112, 16), int("7" * 112, 16)]
    for i in range(12):
        k = specials[i] if i < len(specials) else rng.getrandbits(448)
        P = E.G if i == 0 else E.scalarmul(rng.getrandbits(446), E.G)
        assert call(L, "ht_scalarmul", E.sc_to_bytes(k), E.pt_to_bytes(P), outlen=112) == E.pt_to_bytes(E.scalarmul(k, P))
        assert call(L, "ht_basemul", E.sc_to_bytes(k), outlen=112) == E.pt_to_bytes(E.scalarmul(k, E.G))
    for _ in range(3):
        a, b = rng.getrandbits(448), rng.getrandbits(448)
        P = E.scalarmul(rng.getrandbits(446), E.G)
        got = call(L, "ht_double_scalarmul", E.sc_to_bytes(a), E.sc_to_bytes(b), E.pt_to_bytes(P), outlen=112)
        assert got == E.pt_to_bytes(E.add(E.scalarmul(a, E.G), E.scalarmul(b, P)))


def test_lazy_reduction_bounds_hold(L):
    """The host build audits every fe_mul / fe_sqr operand pair against the column-sum bound (CAPY_FE_CHECK_BOUNDS);
    drive the point formulas with worst-case-ish inputs (limbs at their maxima: coordinates p-1, 2^448-1) and random
    ones and require zero violations."""
    rng = random.Random(9)
    G = E.pt_to_bytes(E.G)
    L.ht_build_gtab(C.c_char_p(G))
    worst = [(E.P - 1).to_bytes(56, "little") * 2, (2**448 - 1).to_bytes(56, "little") * 2, G]
    for xy in worst:
        for k in (2**448 - 1, int("8" * 112, 16), int("7" * 112, 16), rng.getrandbits(448)):
            call(L, "ht_scalarmul", E.sc_to_bytes(k), xy, outlen=112)  # off-curve inputs are fine: only bounds matter
            call(L, "ht_double_scalarmul", E.sc_to_bytes(k), E.sc_to_bytes(k ^ 0x5555), xy, outlen=112)
            call(L, "ht_basemul", E.sc_to_bytes(k), outlen=112)
    # the twisted-curve mixed addition and the dual isogeny (fixed base, r03) on the same worst-case coordinates
    out = (C.c_uint8 * 112)()
    for xy in worst:
        L.ht_tw_sum(C.c_char_p(xy * 5), 5, out)
    assert L.ht_bound_violations() == 0


def test_scalar_field(L):
    rng = random.Random(8)
    edge = [0, 1, E.R, E.R - 1, E.R + 1, 2**448 - 1, 2**446, 2**446 - 1, 2**447, 2**224, 4 * E.R + 123, 2**448 - 2**226]
    pairs = [(a, b) for a in edge for b in edge]
    pairs += [(rng.getrandbits(448), rng.getrandbits(448)) for _ in range(300)]
    pairs += [(rng.getrandbits(rng.randrange(1, 449)), rng.getrandbits(rng.randrange(1, 449))) for _ in range(200)]
    for a, b in pairs:
        assert call(L, "ht_sc_mul_mod", E.sc_to_bytes(a), E.sc_to_bytes(b)) == E.sc_to_bytes(a * b % E.R)
        assert call(L, "ht_sc_sub_mod", E.sc_to_bytes(a), E.sc_to_bytes(b)) == E.sc_to_bytes((a - b) % E.R)
        assert call(L, "ht_sc_mul4_mod", E.sc_to_bytes(a)) == E.sc_to_bytes(4 * a % E.R)


def test_sign_scalars_under_every_reading_of_the_star(L):
    """sc_star4 / sc_sign_z (ed448_algo.h: what sc_star4_kernel and sc_sign_z_kernel run) against the python model
    ed448_ref.schnorr_scalars for the three readings of `bytes_to_scalar(k_bytes) * Scalar::from(4)` and of the `-` that
    consumes it (/root/reference/src/ecc/signable.rs:46,54): random and extreme kb (wrapping and not), h, s."""
    rng = random.Random(0x57A4)
    kbs = [0, 1, 2**446 - 1, 2**446, 2**447, 2**448 - 1, E.R, E.R - 1, (E.R + 3) // 4] + [rng.getrandbits(448) for _ in range(200)]
    for kb in kbs:
        h, s = rng.getrandbits(448), rng.randrange(E.R)
        for star in (0, 1, 2):
            k, z = E.schnorr_scalars(kb, h, s, star)
            kbytes = call(L, "ht_sc_star4", E.sc_to_bytes(kb), star)
            assert kbytes == k.to_bytes(56, "big"), (star, kb)
            assert call(L, "ht_sc_sign_z", kbytes, E.sc_to_bytes(h), E.sc_to_bytes(s), star) == z.to_bytes(56, "big"), (star, kb)


def test_pair_affine_shares_one_inversion(L):
    """pt_pair_to_affine_bytes == two independent conversions, for random projective representatives, Z = 1, Z = p
    (non-canonical zero) and Z = 0 (an invalid point must not spoil its partner and still yields (0, 0))."""
    rng = random.Random(9)
    P = E.P

    def proj(pt, z):
        x, y = pt
        return fb(x * z) + fb(y * z) + int(z % 2**448).to_bytes(56, "little")

    def affine_bytes(pt):
        return fb(pt[0]) + fb(pt[1])

    pts = [E.scalarmul(rng.getrandbits(446), E.G) for _ in range(6)]
    for i in range(0, 6, 2):
        z0, z1 = rng.randrange(1, P), rng.randrange(1, P)
        out = call(L, "ht_pair_affine", proj(pts[i], z0), proj(pts[i + 1], z1), outlen=224)
        assert out == affine_bytes(pts[i]) + affine_bytes(pts[i + 1])
    out = call(L, "ht_pair_affine", proj(pts[0], 1), proj(pts[1], P - 1), outlen=224)
    assert out == affine_bytes(pts[0]) + affine_bytes(pts[1])
    for bad_z in (0, P):  # P is the non-canonical encoding of zero
        bad = fb(5) + fb(7) + int(bad_z).to_bytes(56, "little")
        out = call(L, "ht_pair_affine", bad, proj(pts[2], 12345), outlen=224)
        assert out == bytes(112) + affine_bytes(pts[2])
        out = call(L, "ht_pair_affine", proj(pts[3], 99), bad, outlen=224)
        assert out == affine_bytes(pts[3]) + bytes(112)
    out = call(L, "ht_pair_affine", fb(5) + fb(7) + bytes(56), fb(1) + fb(2) + bytes(56), outlen=224)
    assert out == bytes(224)


def test_point_validation_on_the_host_build(L):
    """pt_validate_bytes (the body of capy_ed448_validate_batch): canonical coordinates and the curve equation; the RFC 8032
    public keys (decoded) are valid points, non-canonical encodings and off-curve pairs are not."""
    import json
    import os

    from oracle import ed448_ref as E

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rfc_ed448.json")) as f:
        v = json.load(f)
    for t in v["rfc8032_7_4"]:
        assert L.ht_validate(C.c_char_p(E.pt_to_bytes(E.rfc8032_decode(bytes.fromhex(t["public"]))))) == 1
    x, y = E.scalarmul(77, E.G)
    assert L.ht_validate(C.c_char_p(E.pt_to_bytes((0, 1)))) == 1
    assert L.ht_validate(C.c_char_p(E.P.to_bytes(56, "little") + E.fe_to_bytes(1))) == 0        # x = p: not canonical
    assert L.ht_validate(C.c_char_p(E.fe_to_bytes(0) + (E.P + 1).to_bytes(56, "little"))) == 0   # y = p + 1
    assert L.ht_validate(C.c_char_p(E.fe_to_bytes(x) + E.fe_to_bytes((y + 1) % E.P))) == 0       # off the curve
    assert L.ht_validate(C.c_char_p(bytes(112))) == 0


def test_constant_address_variant_equals_the_indexed_one(L):
    """vb_scalarmul<true> (every table row read, the wanted one kept by masking: capy_ed448_set_hardened) against the
    oracle for edge scalars and random pairs."""
    from oracle import oracle as O

    rng = random.Random(21)
    ks = [0, 1, 2, 15, 16, 17, 31, 32, 33, E.R - 1, E.R, 2**448 - 1] + [rng.getrandbits(448) for _ in range(12)]
    L.ht_build_gtab_ct(C.c_char_p(E.pt_to_bytes(E.G)))
    for k in ks:
        P = E.pt_to_bytes(E.scalarmul(rng.getrandbits(446), E.G))
        kb = E.sc_to_bytes(k)
        assert call(L, "ht_scalarmul_ct", kb, P, outlen=112) == O.ed448_scalarmul(kb, P)
        assert call(L, "ht_basemul_ct", kb, outlen=112) == O.ed448_basemul(kb)
