"""GPU parity tests of the Ed448 path through the C ABI: committed golden vectors (python big-int model, pinned to
RFC 8032 / RFC 7748), seeded parity vs the C oracle, group-law properties at full batch size, and the src/ecc
protocol round-trips the reference's integration tests run (tests/integration_tests.rs:20-81,116-130,267-281)."""
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
MIB5 = 5242880


@pytest.fixture(scope="module")
def capy():
    import capycrypt_amd

    return capycrypt_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(autouse=True, params=[-1, 0], ids=["wave-per-item-for-small-batches", "lane-per-item-only"])
def ed448_kernel_family(request):
    """Every test runs twice: with the default kernel choice (batches of up to a few thousand scalar multiplications take
    the one-item-per-wave kernels of csrc/ed448_wave.h) and with those switched off (one item per lane at every size)."""
    from capycrypt_amd import _lib

    _lib.check(_lib.lib().capy_ed448_set_wave_max(request.param))
    yield request.param
    _lib.check(_lib.lib().capy_ed448_set_wave_max(-1))


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(HERE, "golden", "ed448_vectors.json")) as f:
        return json.load(f)


def test_golden_scalarmul(capy, vectors):
    v = vectors["scalarmul"]
    got = capy.ops.ed448_scalarmul_batch([bytes.fromhex(t["k"]) for t in v], [bytes.fromhex(t["p"]) for t in v])
    assert [g.hex() for g in got] == [t["out"] for t in v]


def test_golden_sign_and_keypair(capy, vectors):
    """The committed sign / keypair vectors: password and message lengths differ per vector, and each security
    parameter's vectors go through the C ABI as ONE batch (per-item password lengths, include/capyhip.h Conventions),
    then once more one at a time through the reference-shaped API."""
    from capycrypt_amd.message import sign_many, verify_many

    for d in (224, 256, 384, 512):
        v = [t for t in vectors["sign"] if t["d"] == d]
        pws = [bytes.fromhex(t["pw"]) for t in v]
        assert len(set(len(p) for p in pws)) > 1  # really ragged
        kps = capy.KeyPair.new_many(pws, "test key", d)
        assert [k.pub_key.hex() for k in kps] == [t["pub"] for t in v]
        ms = [capy.Message(bytes.fromhex(t["msg"])) for t in v]
        sign_many(ms, kps, d)
        assert [(m.sig.h.hex(), m.sig.z.hex()) for m in ms] == [(t["h"], t["z"]) for t in v]
        assert all(verify_many(ms, [k.pub_key for k in kps]))
        for t in v:
            kp = capy.KeyPair.new(bytes.fromhex(t["pw"]), "test key", d)
            assert kp.pub_key.hex() == t["pub"]
            m = capy.Message(bytes.fromhex(t["msg"]))
            m.sign(kp, d)
            assert (m.sig.h.hex(), m.sig.z.hex()) == (t["h"], t["z"])
            m.verify(kp.pub_key)


def test_ragged_passwords_through_every_protocol(capy, O):
    """One batch, every item with its own password length (0 .. 300 bytes, incl. the bytepad boundaries of D256/D512):
    keypair, sign, verify, key_encrypt -> key_decrypt against the oracle item by item."""
    rng = random.Random(0x9A55)
    for d in (256, 512):
        r2 = (1600 - d) // 8
        plens = [0, 1, 2, 31, 32, 33, 64, r2 - 5, r2 - 4, r2 - 3, r2 - 2, r2, 2 * r2 - 4, 300] + [rng.randrange(0, 200) for _ in range(70)]
        n = len(plens)
        pws = [rng.randbytes(x) for x in plens]
        msgs = [rng.randbytes(rng.randrange(0, 400)) for _ in range(n)]
        pubs = capy.ops.keypair_batch(pws, d)
        assert pubs == [O.keypair_pub(p, d) for p in pws]
        sigs = capy.ops.schnorr_sign_batch(pws, msgs, d)
        assert sigs == [O.sign(p, m, d) for p, m in zip(pws, msgs)]
        assert all(capy.ops.schnorr_verify_batch(pubs, msgs, sigs, d))
        ks = [rng.randbytes(56) for _ in range(n)]
        cts, zs, tags = capy.ops.key_encrypt_batch(pubs, ks, msgs, d)
        wrong = list(pws)
        wrong[5] = wrong[5] + b"x"
        out, ok = capy.ops.key_decrypt_batch(wrong, zs, cts, tags, d)
        assert ok == [i != 5 for i in range(n)]
        assert all(out[i] == (msgs[i] if i != 5 else cts[i]) for i in range(n))


def test_point_validation(capy, O):
    """capy_ed448_validate_batch: on-curve and canonical.  Valid: generator, identity, random multiples.  Invalid: a
    coordinate >= p (non-canonical encoding of a valid point), an off-curve pair, y flipped in one bit."""
    from oracle import ed448_ref as E

    rng = random.Random(3)
    good = [E.pt_to_bytes(E.G), E.pt_to_bytes((0, 1)), E.pt_to_bytes((0, E.P - 1))]
    good += [O.ed448_basemul(rng.randbytes(56)) for _ in range(61)]
    x, y = E.scalarmul(12345, E.G)
    small = E.scalarmul(0, E.G)  # (0, 1): x + p is a non-canonical encoding that still fits 56 bytes
    bad = [
        (small[0] + E.P).to_bytes(56, "little") + E.fe_to_bytes(small[1]),  # x = p  (== 0 mod p)
        E.fe_to_bytes(0) + (1 + E.P).to_bytes(56, "little"),                # y = p + 1
        E.fe_to_bytes(x) + E.fe_to_bytes((y + 1) % E.P),                    # off the curve
        E.fe_to_bytes((x + 1) % E.P) + E.fe_to_bytes(y),
        bytes(112),                                                         # (0, 0)
        b"\xff" * 112,
    ]
    flip = bytearray(good[5])
    flip[70] ^= 4
    bad.append(bytes(flip))
    res = capy.ops.ed448_validate_batch(good + bad)
    assert res == [True] * len(good) + [False] * len(bad)
    assert all(O.ed448_on_curve(p) for p in good)


def _rfc():
    with open(os.path.join(HERE, "golden", "rfc_ed448.json")) as f:
        return json.load(f)


def _gpu_group(capy):
    """Group operations of the python RFC helpers (oracle/ed448_ref.py: rfc8032_verify, x448_via_edwards) routed
    through the C ABI: fixed-base for multiples of G, variable-base otherwise, capy_ed448_add_batch for sums."""
    from oracle import ed448_ref as E

    def mul(k, pt):
        if pt == E.G:
            return E.pt_from_bytes(capy.ops.ed448_basemul_batch([E.sc_to_bytes(k)])[0])
        return E.pt_from_bytes(capy.ops.ed448_scalarmul_batch([E.sc_to_bytes(k)], [E.pt_to_bytes(pt)])[0])

    def var_mul(k, pt):
        return E.pt_from_bytes(capy.ops.ed448_scalarmul_batch([E.sc_to_bytes(k)], [E.pt_to_bytes(pt)])[0])

    def add(a, b):
        return E.pt_from_bytes(capy.ops.ed448_add_batch([E.pt_to_bytes(a)], [E.pt_to_bytes(b)])[0])

    return mul, var_mul, add


def test_rfc8032_key_pairs_and_signatures_on_gpu(capy):
    """RFC 8032 section 7.4, all nine key pairs and seven signatures, through the C ABI:
    A = [s]B by the fixed-base kernel and by the variable-base kernel on G; the verification equation
    [S]B = R + [k]A with [k]A on the variable-base kernel (A is not the generator) and the sum on
    capy_ed448_add_batch; and the same equation as R = [S]B + [L-k]A on capy_ed448_double_scalarmul_batch (the shape of
    Signable::verify, /root/reference/src/ecc/signable.rs:77)."""
    from oracle import ed448_ref as E

    H = bytes.fromhex
    v = _rfc()["rfc8032_7_4"]
    ss = [E.sc_to_bytes(E.rfc8032_secret_scalar(H(t["secret"]))[0]) for t in v]
    pubs = capy.ops.ed448_basemul_batch(ss)
    assert [E.rfc8032_encode(E.pt_from_bytes(p)).hex() for p in pubs] == [t["public"] for t in v]
    assert capy.ops.ed448_scalarmul_batch(ss, [E.pt_to_bytes(E.G)] * len(v)) == pubs
    mul, _, add = _gpu_group(capy)
    sv = [t for t in v if "signature" in t]
    assert len(sv) == 7
    a_be, b_be, pts, rs = [], [], [], []
    for t in sv:
        pk, msg, sig, ctx = H(t["public"]), H(t["message"]), H(t["signature"]), H(t["context"])
        assert E.rfc8032_verify(pk, msg, sig, ctx, mul=mul, addp=add), t["name"]
        bad = bytearray(sig)
        bad[100] ^= 0x10  # a different S: [S]B moves, the equation must fail
        assert not E.rfc8032_verify(pk, msg, bytes(bad), ctx, mul=mul, addp=add)
        k = E.rfc8032_challenge(sig[:57], pk, msg, ctx)
        a_be.append(E.sc_to_bytes(int.from_bytes(sig[57:], "little")))
        b_be.append(E.sc_to_bytes((E.R - k) % E.R))
        pts.append(E.pt_to_bytes(E.rfc8032_decode(pk)))
        rs.append(E.pt_to_bytes(E.rfc8032_decode(sig[:57])))
    assert capy.ops.ed448_double_scalarmul_batch(a_be, b_be, pts) == rs


def test_rfc7748_x448_through_the_variable_base_kernel(capy):
    """RFC 7748 values reproduced with capy_ed448_scalarmul_batch through the 4-isogeny of RFC 7748 section 4.2
    (oracle/ed448_ref.py: x448_via_edwards): both public keys and the Diffie-Hellman shared secret of section 6.2
    (a variable-base multiplication of a non-generator point), the on-curve vector of section 5.2, and the 1000-fold
    iteration of section 5.2: 1000 chained variable-base multiplications, every input point the previous output."""
    from oracle import ed448_ref as E

    H = bytes.fromhex
    v = _rfc()
    _, var_mul, _ = _gpu_group(capy)
    d = v["rfc7748_6_2"]
    five = (5).to_bytes(56, "little")
    assert E.x448_via_edwards(H(d["alice_private"]), five, mul=var_mul).hex() == d["alice_public"]
    assert E.x448_via_edwards(H(d["bob_private"]), five, mul=var_mul).hex() == d["bob_public"]
    assert E.x448_via_edwards(H(d["alice_private"]), H(d["bob_public"]), mul=var_mul).hex() == d["shared_secret"]
    assert E.x448_via_edwards(H(d["bob_private"]), H(d["alice_public"]), mul=var_mul).hex() == d["shared_secret"]
    t = v["rfc7748_5_2"][0]
    assert E.x448_via_edwards(H(t["scalar"]), H(t["u"]), mul=var_mul).hex() == t["out"]
    it = v["rfc7748_5_2_iterated"]
    k = u = H(it["start"])
    for i in range(1000):
        k, u = E.x448_via_edwards(k, u, mul=var_mul), k
        if i == 0:
            assert k.hex() == it["after_1"]
    assert k.hex() == it["after_1000"]


def test_openssl_generated_vectors_on_gpu(capy):
    """The OpenSSL-generated fixture (tests/golden/openssl_ed448.json, an independent implementation) through the C
    ABI, batched: 24 public keys on the fixed-base kernel; 24 verification equations R = [S]B + [L-k]A on the
    double-multiplication kernel and [k]A on the variable-base kernel; 48 X448 shared secrets on the variable-base
    kernel through the 4-isogeny."""
    from oracle import ed448_ref as E

    H = bytes.fromhex
    with open(os.path.join(HERE, "golden", "openssl_ed448.json")) as f:
        v = json.load(f)
    ed = v["ed448"]
    ss = [E.sc_to_bytes(E.rfc8032_secret_scalar(H(t["secret"]))[0]) for t in ed]
    pubs = capy.ops.ed448_basemul_batch(ss)
    assert [E.rfc8032_encode(E.pt_from_bytes(p)).hex() for p in pubs] == [t["public"] for t in ed]
    a_be, b_be, ks, rs = [], [], [], []
    for t in ed:
        pk, msg, sig = H(t["public"]), H(t["message"]), H(t["signature"])
        k = E.rfc8032_challenge(sig[:57], pk, msg)
        a_be.append(E.sc_to_bytes(int.from_bytes(sig[57:], "little")))
        b_be.append(E.sc_to_bytes((E.R - k) % E.R))
        ks.append(E.sc_to_bytes(k))
        rs.append(E.pt_to_bytes(E.rfc8032_decode(sig[:57])))
    assert capy.ops.ed448_double_scalarmul_batch(a_be, b_be, pubs) == rs
    ka = capy.ops.ed448_scalarmul_batch(ks, pubs)  # [k]A, A not the generator
    assert capy.ops.ed448_add_batch(rs, ka) == capy.ops.ed448_basemul_batch(a_be)  # R + [k]A = [S]B
    # X448: one batch of 48 variable-base multiplications of lifted public keys by clamped scalars / 4
    scal, pts, want = [], [], []
    for t in v["x448"]:
        for priv, peer in ((t["a"], t["b_public"]), (t["b"], t["a_public"])):
            u = int.from_bytes(H(peer), "little") % E.P
            scal.append(E.sc_to_bytes(E.x448_clamp(H(priv)) // 4))
            pts.append(E.pt_to_bytes(E.edwards_from_curve448(u, E.curve448_v(u))))
            want.append(t["shared"])
    got = capy.ops.ed448_scalarmul_batch(scal, pts)
    assert [E.x448_u_from_edwards(E.pt_from_bytes(g)).to_bytes(56, "little").hex() for g in got] == want


def test_variable_base_kernel_vs_montgomery_ladder(capy):
    """768 random (scalar, point) pairs in one batch, points anywhere on the curve (4-torsion components included):
    the variable-base kernel against the Montgomery ladder of RFC 7748 section 5, an algorithm that shares nothing with
    the Edwards formulas of the kernel or of the oracle."""
    from oracle import ed448_ref as E

    rng = random.Random(0x7748)
    ks, us, pts = [], [], []
    while len(ks) < 768:
        u = rng.getrandbits(448) % E.P
        vv = E.curve448_v(u)
        if vv is None or u in (0, 1, E.P - 1):
            continue
        ks.append(rng.randbytes(56))
        us.append(u)
        pts.append(E.pt_to_bytes(E.edwards_from_curve448(u, vv)))
    got = capy.ops.ed448_scalarmul_batch([E.sc_to_bytes(E.x448_clamp(k) // 4) for k in ks], pts)
    for k, u, g in zip(ks, us, got):
        assert E.x448_u_from_edwards(E.pt_from_bytes(g)) == E.x448_ladder(E.x448_clamp(k), u)


def test_config4_full_size_distinct_points(capy, O):
    """BASELINE config 4 at full size with 2^18 DISTINCT points: P_i = [t_i]G from the fixed-base
    kernel, uniform 448-bit k_i, and for EVERY item [k_i]P_i (variable-base kernel, per-lane window table in HBM)
    must equal [k_i * t_i mod r]G (fixed-base kernel; the products are python big-ints).  Call sites:
    /root/reference/src/ecc/encryptable.rs:37,78, src/ecc/signable.rs:77."""
    import ctypes as C

    from capycrypt_amd import _lib
    from oracle import ed448_ref as E

    lib = _lib.lib()
    n = 1 << 18
    rng = random.Random(0xCA9C0004)
    tb = rng.randbytes(56 * n)
    kb = rng.randbytes(56 * n)
    ts = [int.from_bytes(tb[56 * i:56 * i + 56], "big") >> 2 for i in range(n)]  # 446-bit t_i
    tb = b"".join(t.to_bytes(56, "big") for t in ts)
    pts = (C.c_uint8 * (112 * n))()
    _lib.check(lib.capy_ed448_basemul_batch(n, _lib.buf(tb), pts))
    assert len(set(bytes(pts)[112 * i:112 * i + 112] for i in range(0, n, 257))) == len(range(0, n, 257))
    out = (C.c_uint8 * (112 * n))()
    _lib.check(lib.capy_ed448_scalarmul_batch(n, _lib.buf(kb), pts, out))
    prod = b"".join(((int.from_bytes(kb[56 * i:56 * i + 56], "big") * ts[i]) % E.R).to_bytes(56, "big") for i in range(n))
    exp = (C.c_uint8 * (112 * n))()
    _lib.check(lib.capy_ed448_basemul_batch(n, _lib.buf(prod), exp))
    got, want = bytes(out), bytes(exp)
    if got != want:
        bad = [i for i in range(n) if got[112 * i:112 * i + 112] != want[112 * i:112 * i + 112]]
        raise AssertionError("%d of %d items differ, first %s" % (len(bad), n, bad[:8]))
    raw = bytes(pts)
    for i in (0, 1, 63, 64, 4095, 4096, n // 3, n - 2, n - 1):  # and a sample against the CPU oracle
        assert got[112 * i:112 * i + 112] == O.ed448_scalarmul(kb[56 * i:56 * i + 56], raw[112 * i:112 * i + 112]), i


def test_seeded_parity_vs_oracle(capy, O):
    rng = random.Random(0xCA9C0004)
    n = 200  # > 3 waves, ragged tail
    sc = [rng.randbytes(56) for _ in range(n)]
    sc[0], sc[1], sc[2] = bytes(56), bytes(55) + b"\x01", b"\xff" * 56
    pts = [O.ed448_basemul(rng.randbytes(56)) for _ in range(n)]
    assert capy.ops.ed448_scalarmul_batch(sc, pts) == [O.ed448_scalarmul(s, p) for s, p in zip(sc, pts)]
    assert capy.ops.ed448_basemul_batch(sc) == [O.ed448_basemul(s) for s in sc]
    assert capy.ops.ed448_add_batch(pts, pts[::-1]) == [O.ed448_add(p, q) for p, q in zip(pts, pts[::-1])]
    a = [rng.randbytes(56) for _ in range(n)]
    exp = [O.ed448_add(O.ed448_basemul(x), O.ed448_scalarmul(s, p)) for x, s, p in zip(a, sc, pts)]
    assert capy.ops.ed448_double_scalarmul_batch(a, sc, pts) == exp


def test_group_law_properties_large_batch(capy, O):
    """Size-independent checks on a batch larger than the oracle could verify item by item:
    [a]P + [b]P == [a+b]P, [r]P == identity, fixed-base == variable-base on G."""
    from oracle import ed448_ref as E

    rng = random.Random(77)
    n = 4096
    G = O.ed448_generator()
    t = [rng.randbytes(56) for _ in range(n)]
    P = capy.ops.ed448_basemul_batch(t)
    assert P == capy.ops.ed448_scalarmul_batch(t, [G] * n)
    a = [rng.getrandbits(440) for _ in range(n)]
    b = [rng.getrandbits(440) for _ in range(n)]
    aP = capy.ops.ed448_scalarmul_batch([E.sc_to_bytes(x) for x in a], P)
    bP = capy.ops.ed448_scalarmul_batch([E.sc_to_bytes(x) for x in b], P)
    abP = capy.ops.ed448_scalarmul_batch([E.sc_to_bytes(x + y) for x, y in zip(a, b)], P)
    assert capy.ops.ed448_add_batch(aP, bP) == abP
    ident = (0).to_bytes(56, "little") + (1).to_bytes(56, "little")
    assert capy.ops.ed448_scalarmul_batch([E.sc_to_bytes(E.R)] * 256, P[:256]) == [ident] * 256
    for i in (0, 1000, 4095):
        assert O.ed448_on_curve(aP[i]) and aP[i] == O.ed448_scalarmul(E.sc_to_bytes(a[i]), P[i])


@pytest.mark.parametrize("d", [256, 512])
def test_schnorr_and_ecdhies_batches_vs_oracle(capy, O, d):
    rng = random.Random(500 + d)
    n = 70
    pws = [rng.randbytes(16) for _ in range(n)]
    msgs = [rng.randbytes(rng.choice((0, 1, 100, 133, 136, 137, 1000, 5000))) for _ in range(n)]
    pubs = capy.ops.keypair_batch(pws, d)
    assert pubs == [O.keypair_pub(p, d) for p in pws]
    sigs = capy.ops.schnorr_sign_batch(pws, msgs, d)
    assert sigs == [O.sign(p, m, d) for p, m in zip(pws, msgs)]
    assert all(capy.ops.schnorr_verify_batch(pubs, msgs, sigs, d))
    tampered = list(msgs)
    tampered[5] += b"x"
    ok = capy.ops.schnorr_verify_batch(pubs, tampered, sigs, d)
    assert ok.count(False) == 1 and not ok[5]
    kr = [rng.randbytes(56) for _ in range(n)]
    cts, zs, tags = capy.ops.key_encrypt_batch(pubs, kr, msgs, d)
    exp = [O.key_encrypt(pk, k, m, d) for pk, k, m in zip(pubs, kr, msgs)]
    assert (cts, zs, tags) == ([e[0] for e in exp], [e[1] for e in exp], [e[2] for e in exp])
    pts, ok = capy.ops.key_decrypt_batch(pws, zs, cts, tags, d)
    assert all(ok) and pts == msgs
    pws2 = list(pws)
    pws2[7] = b"y" * 16
    pts, ok = capy.ops.key_decrypt_batch(pws2, zs, cts, tags, d)
    assert ok.count(False) == 1 and not ok[7] and pts[7] == cts[7] and pts[8] == msgs[8]


# ---- the reference's integration tests, same shape (5 MiB random messages)
@pytest.mark.parametrize("d,pwlen", [(256, 64), (512, 32)])
def test_key_gen_enc_dec(capy, d, pwlen):
    msg = capy.Message(capy.get_random_bytes(MIB5))
    original = bytes(msg.msg)
    key_pair = capy.KeyPair.new(capy.get_random_bytes(pwlen), "test key", capy.SecParam.try_from(d))
    msg.key_encrypt(key_pair.pub_key, d)
    assert bytes(msg.msg) != original
    msg.key_decrypt(key_pair.priv_key)
    assert bytes(msg.msg) == original


@pytest.mark.parametrize("d", [256, 512])
def test_signature(capy, d):
    msg = capy.Message(capy.get_random_bytes(MIB5))
    key_pair = capy.KeyPair.new(capy.get_random_bytes(64), "test key", capy.SecParam.try_from(d))
    msg.sign(key_pair, d)
    msg.verify(key_pair.pub_key)
    msg.msg[12345] ^= 1
    with pytest.raises(capy.OperationError) as e:
        msg.verify(key_pair.pub_key)
    assert e.value.variant == "SignatureVerificationFailure"


def test_key_decrypt_handling_bad_input(capy):
    new_msg = capy.Message(capy.get_random_bytes(125))
    kp1 = capy.KeyPair.new(capy.get_random_bytes(32), "test key", capy.SecParam.D512)
    kp2 = capy.KeyPair.new(capy.get_random_bytes(32), "test key", capy.SecParam.D512)
    new_msg.key_encrypt(kp1.pub_key, capy.SecParam.D512)
    before = bytes(new_msg.msg)
    with pytest.raises(capy.OperationError) as e:
        new_msg.key_decrypt(kp2.priv_key)
    assert e.value.variant == "KeyDecryptionError" and bytes(new_msg.msg) == before


def test_sig_timing_side_channel_shape(capy):
    """tests/integration_tests.rs:137-156: sign with passwords of 1..512 bytes, verify each."""
    for i in (1, 2, 64, 257, 512):
        msg = capy.Message(capy.get_random_bytes(4096))
        kp = capy.KeyPair.new(capy.get_random_bytes(i), "test key", capy.SecParam.D512)
        msg.sign(kp, capy.SecParam.D512)
        msg.verify(kp.pub_key)


def test_dev_api_protocols_roundtrip(capy, O):
    """Device-buffer forms of keypair / sign / verify / key_encrypt / key_decrypt on a uniformly strided batch."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    n, L, stride, d = 300, 1000, 1008, 512

    def rand(nb, seed):
        t = torch.empty((nb + 7) // 8 * 8, dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, None))
        return t

    msgs, pws, kr = rand(n * stride, 1), rand(n * 32, 2), rand(n * 56, 3)
    pubs = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
    h = torch.zeros(n * 56, dtype=torch.uint8, device="cuda")
    z = torch.zeros(n * 56, dtype=torch.uint8, device="cuda")
    st = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    _lib.check(lib.capy_keypair_batch_dev(d, n, pws.data_ptr(), 32, None, pubs.data_ptr(), None))
    _lib.check(lib.capy_schnorr_sign_batch_dev(d, n, pws.data_ptr(), 32, None, msgs.data_ptr(), None, L, stride, h.data_ptr(),
                                               z.data_ptr(), None))
    _lib.check(lib.capy_schnorr_verify_batch_dev(d, n, pubs.data_ptr(), msgs.data_ptr(), None, L, stride, h.data_ptr(),
                                                 z.data_ptr(), st.data_ptr(), None))
    torch.cuda.synchronize()
    assert not st.cpu().numpy().any()
    hm, hp = bytes(msgs.cpu().numpy()), bytes(pws.cpu().numpy())
    hh, hz, hpub = bytes(h.cpu().numpy()), bytes(z.cpu().numpy()), bytes(pubs.cpu().numpy())
    for i in (0, 63, 64, 299):
        m, pw = hm[i * stride:i * stride + L], hp[32 * i:32 * i + 32]
        assert hpub[112 * i:112 * i + 112] == O.keypair_pub(pw, d)
        assert (hh[56 * i:56 * i + 56], hz[56 * i:56 * i + 56]) == O.sign(pw, m, d)
    zxy = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
    tags = torch.zeros(n * 56, dtype=torch.uint8, device="cuda")
    work = msgs.clone()
    _lib.check(lib.capy_key_encrypt_batch_dev(d, n, pubs.data_ptr(), kr.data_ptr(), work.data_ptr(), None, L, stride,
                                              zxy.data_ptr(), tags.data_ptr(), None))
    torch.cuda.synchronize()
    hc, hk = bytes(work.cpu().numpy()), bytes(kr.cpu().numpy())
    ect, ez, etag = O.key_encrypt(hpub[:112], hk[:56], hm[:L], d)
    assert hc[:L] == ect and bytes(zxy[:112].cpu().numpy()) == ez and bytes(tags[:56].cpu().numpy()) == etag
    _lib.check(lib.capy_key_decrypt_batch_dev(d, n, pws.data_ptr(), 32, None, zxy.data_ptr(), work.data_ptr(), None, L, stride,
                                              tags.data_ptr(), st.data_ptr(), None))
    torch.cuda.synchronize()
    assert not st.cpu().numpy().any() and bytes(work.cpu().numpy()) == hm
    # every protocol call scrubs its secret intermediates (s, k, W, ke || ka) from the pooled scratch when it returns
    # (one launch for all ranges, csrc/sponge.hip: workspace_scrub_many): check it after each kind of call
    import ctypes as C
    nz = C.c_uint64(1)
    _lib.check(lib.capy_debug_secret_scratch_nonzero(None, C.byref(nz)))
    assert nz.value == 0  # key_decrypt: s, W, ke || ka
    _lib.check(lib.capy_schnorr_sign_batch_dev(d, n, pws.data_ptr(), 32, None, msgs.data_ptr(), None, L, stride, h.data_ptr(),
                                               z.data_ptr(), None))
    _lib.check(lib.capy_debug_secret_scratch_nonzero(None, C.byref(nz)))
    assert nz.value == 0  # sign: s, k
    _lib.check(lib.capy_key_encrypt_batch_dev(d, n, pubs.data_ptr(), kr.data_ptr(), work.data_ptr(), None, L, stride,
                                              zxy.data_ptr(), tags.data_ptr(), None))
    _lib.check(lib.capy_debug_secret_scratch_nonzero(None, C.byref(nz)))
    assert nz.value == 0  # key_encrypt: k, W, ke || ka


def test_full_batch_pair_inversion_kernels(capy, O):
    """2^18 + 77 items (BASELINE config 4 size, ragged last wave): from 262 144 items two items per lane share one
    inversion (ed448.hip: vb2_kernel / fb2_kernel).  Size-independent property: [k]G by the fixed-base kernel equals
    [k]G by the variable-base kernel for every item; a sample is checked against the oracle."""
    import ctypes as C

    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    n = (1 << 18) + 77
    sc = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_fill_random_dev(sc.data_ptr(), n * 56, 0xCA9C0004, None))
    G = O.ed448_generator()
    gs = torch.tensor(list(G), dtype=torch.uint8, device="cuda").repeat(n)
    fb = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
    vb = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), fb.data_ptr(), None))
    _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), gs.data_ptr(), vb.data_ptr(), None))
    torch.cuda.synchronize()
    hf, hv, hs = bytes(fb.cpu().numpy()), bytes(vb.cpu().numpy()), bytes(sc.cpu().numpy())
    assert hf == hv
    for i in (0, 63, 64, 127, 128, n // 2, n - 78, n - 77, n - 14, n - 1):
        assert hf[112 * i:112 * i + 112] == O.ed448_basemul(hs[56 * i:56 * i + 56]), i


# ---------------------------------------------------------------- the reference's remaining integration tests
def test_sig_key_lengths_1_to_512(capy, O):
    """tests/integration_tests.rs:137-156 (test_sig_timing_side_channel): passwords of 2^0 .. 2^9 bytes, sign then
    verify at D512 -- here as ONE batch with ten different password lengths, checked against the oracle, plus the
    reference's one-message-at-a-time form on a large message."""
    from capycrypt_amd.message import sign_many, verify_many

    rng = random.Random(1 << 9)
    pws = [rng.randbytes(1 << i) for i in range(10)]
    msgs = [rng.randbytes(3000 + 17 * i) for i in range(10)]
    kps = capy.KeyPair.new_many(pws, "test key", 512)
    assert [k.pub_key for k in kps] == [O.keypair_pub(p, 512) for p in pws]
    ms = [capy.Message(m) for m in msgs]
    sign_many(ms, kps, 512)
    assert [(m.sig.h, m.sig.z) for m in ms] == [O.sign(p, x, 512) for p, x in zip(pws, msgs)]
    assert all(verify_many(ms, [k.pub_key for k in kps]))
    big = capy.Message(rng.randbytes(MIB5))
    kp = capy.KeyPair.new(pws[9], "test key", capy.SecParam.D512)
    big.sign(kp, capy.SecParam.D512)
    big.verify(kp.pub_key)


def test_reading_writing_keypair_and_message(capy, tmp_path):
    """tests/integration_tests.rs:158-235: KeyPair and Message survive write_to_file / read_from_file; a key pair read
    back signs, a message read back (before and after signing) verifies."""
    rng = random.Random(77)
    kp = capy.KeyPair.new(rng.randbytes(32), "test key", capy.SecParam.D512)
    kpath = str(tmp_path / "read_write_keypair.json")
    kp.write_to_file(kpath)
    back = capy.KeyPair.read_from_file(kpath)
    assert (back.owner, back.pub_key, back.priv_key, back.date_created) == (kp.owner, kp.pub_key, kp.priv_key, kp.date_created)
    msg = capy.Message(rng.randbytes(200000))
    msg.sign(back, capy.SecParam.D512)
    msg.verify(back.pub_key)
    mpath = str(tmp_path / "temp_message.json")
    capy.Message(rng.randbytes(50000)).write_to_file(mpath)
    initial = capy.Message.read_from_file(mpath)
    initial.sign(kp, capy.SecParam.D512)
    initial.write_to_file(mpath)
    signed = capy.Message.read_from_file(mpath)
    signed.verify(kp.pub_key)
    signed.msg[0] ^= 1
    with pytest.raises(capy.OperationError):
        signed.verify(kp.pub_key)
    # a reference-written key file carries pub_key in the curve crate's layout: the affine bytes come back from priv_key
    import json

    doc = json.loads(kp.to_json())
    doc.pop("capyhip_curve_layout")
    doc["pub_key"] = {"opaque": "whatever tiny_ed448_goldilocks writes"}
    foreign = capy.KeyPair.from_json(json.dumps(doc))
    assert foreign.pub_key == b"" and foreign.derive_pub_key(512) == kp.pub_key


def test_generator_can_be_replaced(capy, O):
    """capy_ed448_set_generator: the hedge for assumption (i) of DESIGN.md section 2.  With G' = [5]G every fixed-base
    result moves to the new base ([k]G' = [5k]G), key pairs / signatures / ECDHIES stay self-consistent, and NULL
    restores the RFC 8032 base point (the RFC key pairs come out again).  Off-curve generators are refused."""
    from capycrypt_amd import _lib
    from oracle import ed448_ref as E

    rng = random.Random(55)
    G = capy.ops.ed448_get_generator()
    assert G == E.pt_to_bytes(E.G) == O.ed448_generator()
    ks = [rng.randbytes(56) for _ in range(70)]
    try:
        g5 = O.ed448_basemul(E.sc_to_bytes(5))
        capy.ops.ed448_set_generator(g5)
        assert capy.ops.ed448_get_generator() == g5
        assert capy.ops.ed448_basemul_batch(ks) == [O.ed448_scalarmul(k, g5) for k in ks]
        pts = [O.ed448_basemul(rng.randbytes(56)) for _ in ks]
        exp = [O.ed448_add(O.ed448_scalarmul(a, g5), O.ed448_scalarmul(b, p)) for a, b, p in zip(ks, ks[::-1], pts)]
        assert capy.ops.ed448_double_scalarmul_batch(ks, ks[::-1], pts) == exp
        pws = [rng.randbytes(rng.randrange(1, 80)) for _ in range(20)]
        msgs = [rng.randbytes(rng.randrange(0, 500)) for _ in range(20)]
        pubs = capy.ops.keypair_batch(pws, 512)
        assert pubs != [O.keypair_pub(p, 512) for p in pws]  # a different base: different public keys
        sigs = capy.ops.schnorr_sign_batch(pws, msgs, 512)
        assert all(capy.ops.schnorr_verify_batch(pubs, msgs, sigs, 512))
        cts, zs, tags = capy.ops.key_encrypt_batch(pubs, ks[:20], msgs, 512)
        out, ok = capy.ops.key_decrypt_batch(pws, zs, cts, tags, 512)
        assert all(ok) and out == msgs
        bad = bytearray(g5)
        bad[60] ^= 1
        with pytest.raises(_lib.CapyHipError):
            capy.ops.ed448_set_generator(bytes(bad))
        assert capy.ops.ed448_get_generator() == g5
        # a generator whose order is NOT the prime r is refused: G + (0, -1) = (-Gx, -Gy), of order 2r (the fixed-base
        # tables are built from scalars reduced mod r and, on the twisted curve, from [1/4 mod r] G)
        gx, gy = E.G
        with pytest.raises(_lib.CapyHipError):
            capy.ops.ed448_set_generator(E.pt_to_bytes(((-gx) % E.P, (-gy) % E.P)))
        assert capy.ops.ed448_get_generator() == g5
    finally:
        capy.ops.ed448_set_generator(None)
    assert capy.ops.ed448_get_generator() == G
    assert capy.ops.ed448_basemul_batch(ks) == [O.ed448_basemul(k) for k in ks]


def test_hardened_mode_is_bit_identical(capy, O):
    """capy_ed448_set_hardened: CAPY_HARDEN_PROTOCOL (the default) runs the multiplications by secret scalars inside the
    protocol calls on the constant-address kernels (every row of the window table read per window, no address depends on
    a scalar), CAPY_HARDEN_ALL the raw multiplications as well, CAPY_HARDEN_OFF none.  Every result must be the same in all three and equal the
    oracle's: raw operations, key pairs, signatures, ECDHIES -- with the one-item-per-wave kernels and without (the
    autouse fixture), i.e. through wave::*<true> and vb_ct_kernel / fb_ct7_kernel (the fixed base with its lookups on the
    matrix cores, csrc/ed448_fb7.h; a full-size ragged batch of it: test_hardened_pair_kernels_match)."""
    rng = random.Random(0xC7)
    n = 130
    ks = [rng.randbytes(56) for _ in range(n)]
    ks[0], ks[1], ks[2] = bytes(56), bytes(55) + b"\x01", b"\xff" * 56
    pts = [O.ed448_basemul(rng.randbytes(56)) for _ in range(n)]
    pws = [rng.randbytes(rng.randrange(0, 90)) for _ in range(n)]
    msgs = [rng.randbytes(rng.randrange(0, 600)) for _ in range(n)]

    def run():
        r = {"vb": capy.ops.ed448_scalarmul_batch(ks, pts), "fb": capy.ops.ed448_basemul_batch(ks),
             "pub": capy.ops.keypair_batch(pws, 512), "sig": capy.ops.schnorr_sign_batch(pws, msgs, 512)}
        r["ver"] = capy.ops.schnorr_verify_batch(r["pub"], msgs, r["sig"], 512)
        r["enc"] = capy.ops.key_encrypt_batch(r["pub"], ks, msgs, 512)
        c, z, t = r["enc"]
        r["dec"] = capy.ops.key_decrypt_batch(pws, z, c, t, 512)
        return r

    try:
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_OFF)
        plain = run()
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_PROTOCOL)
        default = run()
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_ALL)
        hard = run()
    finally:
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_PROTOCOL)
    for k in plain:
        assert hard[k] == plain[k], k
        assert default[k] == plain[k], k
    assert plain["vb"] == [O.ed448_scalarmul(k, p) for k, p in zip(ks, pts)]
    assert plain["fb"] == [O.ed448_basemul(k) for k in ks] and all(plain["ver"]) and all(plain["dec"][1])


def test_scalar_star_modes_match_oracle(capy, O):
    """capy_ed448_set_scalar_star: the three readings of `bytes_to_scalar(k_bytes) * Scalar::from(4)` in Signable::sign
    (/root/reference/src/ecc/signable.rs:46,54).  In every mode the GPU's (h, z) equal the oracle's in the same mode and
    verify; the modes differ from each other."""
    rng = random.Random(0x57A2)
    n = 70
    pws = [rng.randbytes(rng.randrange(1, 70)) for _ in range(n)]
    msgs = [rng.randbytes(rng.randrange(0, 400)) for _ in range(n)]
    pubs = capy.ops.keypair_batch(pws, 512)
    got = {}
    try:
        for mode in (0, 1, 2):
            capy.ops.ed448_set_scalar_star(mode)
            O.set_scalar_star(mode)
            sigs = capy.ops.schnorr_sign_batch(pws, msgs, 512)
            assert sigs == [O.sign(pw, m, 512) for pw, m in zip(pws, msgs)], mode
            assert all(capy.ops.schnorr_verify_batch(pubs, msgs, sigs, 512)), mode
            got[mode] = sigs
    finally:
        capy.ops.ed448_set_scalar_star(0)
        O.set_scalar_star(0)
    assert got[0] != got[1] and got[0] != got[2] and got[1] != got[2]


def test_hardened_pair_kernels_match(capy, O, ed448_kernel_family):
    """The kernels of batches from 262 144 items: two items per lane sharing one inversion (vb2_kernel, fb2_kernel<false>)
    against their constant-address counterparts (vb_ct_kernel, fb_ct7_kernel: 7-bit windows selected by a one-hot matrix
    product, every one of the 262 214 results compared).  CAPY_DEBUG=ed448_pair cannot be switched at
    run time, so a batch of the real threshold size (with a ragged last wave) runs once in mode 0 and once in mode 3 and
    must agree item by item; a sample is checked against the oracle."""
    import ctypes as C

    from capycrypt_amd import _lib

    if ed448_kernel_family != -1:
        pytest.skip("batch size is far above the wave-kernel threshold either way")
    import torch

    lib = _lib.lib()
    n = (1 << 18) + 70  # ragged last wave of 128
    dev = torch.device("cuda", 0)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sc = torch.empty(n * 56, dtype=torch.uint8, device=dev)
    tsc = torch.empty(n * 56, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(sc.data_ptr(), n * 56, 71, sp))
    _lib.check(lib.capy_fill_random_dev(tsc.data_ptr(), n * 56, 72, sp))
    pts = torch.empty(n * 112, dtype=torch.uint8, device=dev)
    outs = {}
    try:
        for mode in (0, 1):  # CAPY_HARDEN_OFF, CAPY_HARDEN_ALL
            capy.ops.ed448_set_hardened(mode)
            if mode == 0:
                _lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp))
            vb = torch.empty(n * 112, dtype=torch.uint8, device=dev)
            fb = torch.empty(n * 112, dtype=torch.uint8, device=dev)
            _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), vb.data_ptr(), sp))
            _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), fb.data_ptr(), sp))
            torch.cuda.synchronize()
            outs[mode] = (vb, fb)
    finally:
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_PROTOCOL)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    sch, pth, vbh = bytes(sc.cpu().numpy()), bytes(pts.cpu().numpy()), bytes(outs[1][0].cpu().numpy())
    for i in (0, 63, 64, 127, 128, n - 71, n - 1):
        assert vbh[112 * i:112 * i + 112] == O.ed448_scalarmul(sch[56 * i:56 * i + 56], pth[112 * i:112 * i + 112]), i


def test_many_forms_of_the_python_mirror(capy, O):
    """The batched helpers over lists of Message (capycrypt_amd/message.py) equal the one-at-a-time reference-shaped
    methods: tagged hash, ECDHIES encrypt / decrypt incl. restore-on-failure, with ragged passwords."""
    from capycrypt_amd.message import compute_tagged_hash_many, key_decrypt_many, key_encrypt_many

    rng = random.Random(0xAB)
    n = 40
    pws = [rng.randbytes(rng.randrange(0, 100)) for _ in range(n)]
    bodies = [rng.randbytes(rng.randrange(0, 3000)) for _ in range(n)]
    ms = [capy.Message(b) for b in bodies]
    compute_tagged_hash_many(ms, pws, "T", 512)
    assert [m.digest for m in ms] == [O.kmac_xof(p, b, 512, b"T", 512) for p, b in zip(pws, bodies)]
    kps = capy.KeyPair.new_many(pws, "k", 512)
    ks = [rng.randbytes(56) for _ in range(n)]
    key_encrypt_many(ms, [k.pub_key for k in kps], 512, ks)
    one = capy.Message(bodies[7])
    one.key_encrypt(kps[7].pub_key, 512, ks[7])
    assert bytes(ms[7].msg) == bytes(one.msg) and ms[7].digest == one.digest and ms[7].asym_nonce == one.asym_nonce
    wrong = list(pws)
    wrong[3] = wrong[3] + b"?"
    ms[9].digest = ms[9].digest[:10]  # malformed tag: a failure, message untouched
    ct9 = bytes(ms[9].msg)
    ok = key_decrypt_many(ms, wrong)
    assert ok == [i not in (3, 9) for i in range(n)]
    assert all(bytes(ms[i].msg) == bodies[i] for i in range(n) if i not in (3, 9)) and bytes(ms[9].msg) == ct9


def test_wave_kernels_equal_lane_kernels_on_edge_cases(capy, O, ed448_kernel_family):
    """One item per wave (csrc/ed448_wave.h) against one item per lane, byte for byte: random and extreme scalars
    (0, 1, 2^448 - 1, r, r - 1, single bits), distinct subgroup points, the identity, the point of order 2, and a
    non-curve point (both kernel families run the same formulas, so even that must agree); variable base, fixed base and
    [a]G + [b]P; a few results are also checked against the C oracle."""
    import ctypes as C

    from capycrypt_amd import _lib

    if ed448_kernel_family != -1:
        pytest.skip("sets the kernel family itself")
    lib = _lib.lib()
    rng = random.Random(448)
    r = (1 << 446) - 0x8335dc163bb124b65129c96fde933d8d723a70aadc873d6d54a7bb0d
    ks = [0, 1, 2, (1 << 448) - 1, r, r - 1, r + 1, 1 << 447, 1 << 224, (1 << 224) - 1, 0x0F0F << 430]
    ks += [rng.getrandbits(448) for _ in range(40)]
    n = len(ks)
    kb = [k.to_bytes(56, "big") for k in ks]
    ts = [rng.getrandbits(446).to_bytes(56, "big") for _ in range(n)]
    pts = capy.ops.ed448_basemul_batch(ts)
    ident = (0).to_bytes(56, "little") + (1).to_bytes(56, "little")
    p = (1 << 448) - (1 << 224) - 1
    order2 = (0).to_bytes(56, "little") + (p - 1).to_bytes(56, "little")
    pts[3], pts[4], pts[5] = ident, order2, (5).to_bytes(56, "little") + (7).to_bytes(56, "little")
    results = []
    try:
        for wmax, mode in ((0, 0), (1 << 20, 0), (0, 1), (1 << 20, 1)):  # lane / wave kernels, indexed / constant-address
            _lib.check(lib.capy_ed448_set_wave_max(wmax))
            capy.ops.ed448_set_hardened(mode)
            vb = capy.ops.ed448_scalarmul_batch(kb, pts)
            fb = capy.ops.ed448_basemul_batch(kb)
            out = (C.c_uint8 * (n * 112))()
            _lib.check(lib.capy_ed448_double_scalarmul_batch(n, b"".join(kb[::-1]), b"".join(kb), b"".join(pts), out))
            results.append((vb, fb, bytes(out)))
    finally:
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_PROTOCOL)
    for idx, other in enumerate(results[1:]):
        # the constant-address lane kernel uses 4-bit windows: a different operation sequence, which must agree on the
        # curve only -- the non-curve point (item 5) is compared between the two indexed families alone
        keep = [i for i in range(n) if idx == 0 or i != 5]
        assert [results[0][0][i] for i in keep] == [other[0][i] for i in keep]
        assert results[0][1] == other[1]
        assert [results[0][2][112 * i:112 * i + 112] for i in keep] == [other[2][112 * i:112 * i + 112] for i in keep]
    for i in (0, 1, 3, 6, 11, n - 1):
        assert results[1][0][i] == O.ed448_scalarmul(kb[i], pts[i]), i
        assert results[1][1][i] == O.ed448_basemul(kb[i]), i


def test_matrix_core_fixed_base_on_structured_scalars(capy, O, ed448_kernel_family):
    """csrc/ed448_fb7.h: the hardened fixed base selects its table entries with a one-hot matrix product over 7-bit signed
    windows.  Scalars that put every window at its extremes -- the same 7-bit pattern everywhere (0, 1, 63, 64, 65, 127),
    one window at a time at 64 (digit -64 and a carry), at 63, or cleared in an all-ones scalar, repeated bytes, values
    around multiples of the group order, 2^448 - 1 -- must give the indexed kernel's and the oracle's points."""
    from capycrypt_amd import _lib

    if ed448_kernel_family != 0:
        pytest.skip("forces the lane-per-item kernels itself")
    lib = _lib.lib()
    R = (1 << 446) - 0x8335dc163bb124b65129c96fde933d8d723a70aadc873d6d54a7bb0d
    ks = [0, 1, 2, 63, 64, 65, 127, 128, (1 << 448) - 1, 1 << 447, (1 << 447) - 1, R, R - 1, R + 1, 2 * R, 3 * R + 5]
    ks += [sum(d << (7 * i) for i in range(64)) & ((1 << 448) - 1) for d in (0, 1, 63, 64, 65, 127)]
    for i in range(64):
        ks += [64 << (7 * i), 63 << (7 * i), ((1 << 448) - 1) ^ (127 << (7 * i))]
    ks += [int.from_bytes(bytes([b]) * 56, "big") for b in (0x55, 0xAA, 0x7F, 0x80, 0xFE, 0x01)]
    kb = [k.to_bytes(56, "big") for k in ks]
    try:
        _lib.check(lib.capy_ed448_set_wave_max(0))
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_ALL)
        hard = capy.ops.ed448_basemul_batch(kb)
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_OFF)
        plain = capy.ops.ed448_basemul_batch(kb)
    finally:
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_PROTOCOL)
        _lib.check(lib.capy_ed448_set_wave_max(-1))
    assert hard == plain
    for i in list(range(0, 22)) + list(range(22, len(ks), 17)):
        assert hard[i] == O.ed448_basemul(kb[i]), hex(ks[i])
