"""CPU: pin the sponge oracle against every known-answer vector the reference's tests hold
(tests/golden/reference_kats.json, each entry cites its reference file:line) and against hashlib."""
import hashlib
import random

import pytest

from oracle import oracle as O


def test_sha3_kats(kats):
    for v in kats["sha3"]:
        assert O.sha3(bytes.fromhex(v["msg_hex"]), v["d"]).hex() == v["digest_hex"], v["src"]


def test_tagged_hash_kats(kats):
    for v in kats["tagged_hash"]:
        got = O.kmac_xof(bytes.fromhex(v["pw_hex"]), bytes.fromhex(v["msg_hex"]), v["d"], v["s"].encode(), v["d"])
        assert got.hex() == v["digest_hex"], v["src"]


def test_cshake_kats(kats):
    for v in kats["cshake"]:
        got = O.cshake(bytes.fromhex(v["x_hex"]), v["l_bits"], v["n"].encode(), v["s"].encode(), v["d"])
        assert got.hex() == v["out_hex"], v["src"]


def test_kmac_kats(kats):
    for v in kats["kmac_xof"]:
        got = O.kmac_xof(bytes.fromhex(v["k_hex"]), bytes.fromhex(v["x_hex"]), v["l_bits"], v["s"].encode(), v["d"])
        assert got.hex() == v["out_hex"], v["src"]


def test_encodings(kats):
    e = kats["encodings"]
    for v, exp in e["right_encode"]:
        assert list(O.right_encode(v)) == exp
    for v, exp in e["left_encode"]:
        assert list(O.left_encode(v)) == exp
    assert list(O.byte_pad(b"test", 4)) == e["byte_pad"][0]["out"]
    assert O.byte_pad(bytes.fromhex(e["byte_pad"][1]["x_hex"]), 200).hex() == e["byte_pad"][1]["out_hex"]


@pytest.mark.parametrize("d,h", [(224, hashlib.sha3_224), (256, hashlib.sha3_256), (384, hashlib.sha3_384),
                                 (512, hashlib.sha3_512)])
def test_fips_mode_equals_hashlib(d, h):
    rng = random.Random(d)
    for n in list(range(0, 300)) + [1000, 4096, 10007]:
        m = rng.randbytes(n)
        assert O.sha3(m, d, quirks=0) == h(m).digest(), n


def test_reference_quirks_d256_is_fips_everywhere():
    rng = random.Random(1)
    for n in range(0, 600):
        m = rng.randbytes(n)
        assert O.sha3(m, 256, quirks=1) == hashlib.sha3_256(m).digest()


def test_reference_quirk_set_sha3_512():
    """shake() decides 0x06/0x86 on len % 136 whatever d is (shake_functions.rs:25): SHA3-512 differs
    from FIPS 202 exactly when one of len = 71 (mod 72), len = 135 (mod 136) holds."""
    rng = random.Random(2)
    diff = []
    for n in range(0, 300):
        m = rng.randbytes(n)
        if O.sha3(m, 512, quirks=1) != hashlib.sha3_512(m).digest():
            diff.append(n)
    expect = [n for n in range(0, 300) if (n % 72 == 71) != (n % 136 == 135)]
    assert diff == expect


def test_shake_mutation_visible_to_caller():
    d, padded = O.sha3(b"test", 256, want_padded=True)
    assert padded == b"test" + b"\x06" + b"\0" * 130 + b"\x80" and len(padded) == 136


def test_kmac_missing_pad_quirk_differs_from_standard():
    # KMAC D512: len(X) = 133 (mod 136) lands the 0x04 suffix on a block boundary -> no 0x80 (sponge.rs:13)
    x = bytes(range(133))
    assert O.kmac_xof(b"k" * 32, x, 256, b"S", 512, quirks=1) != O.kmac_xof(b"k" * 32, x, 256, b"S", 512, quirks=0)
    x = bytes(range(134))
    assert O.kmac_xof(b"k" * 32, x, 256, b"S", 512, quirks=1) == O.kmac_xof(b"k" * 32, x, 256, b"S", 512, quirks=0)


def test_sha3_encrypt_roundtrip_and_restore():
    rng = random.Random(3)
    for d in (224, 256, 384, 512):
        for n in (0, 1, 523, 5000):
            pw, z, m = rng.randbytes(64), rng.randbytes(512), rng.randbytes(n)
            ct, tag = O.sha3_encrypt(pw, z, m, d)
            assert len(ct) == n and len(tag) == 64
            pt, ok = O.sha3_decrypt(pw, z, ct, tag, d)
            assert ok and pt == m
            pt, ok = O.sha3_decrypt(rng.randbytes(64), z, ct, tag, d)  # tests/integration_tests.rs:250-262
            assert not ok and pt == ct


def test_keccakf_zero_state():
    st = O.keccakf1600([0] * 25)
    assert st[0] == 0xF1258F7940E1DDE7 and st[24] == 0xEAF1FF7B5CECA249  # XKCP KeccakF-1600-IntermediateValues


def test_inplace_keccak_equals_textbook_round():
    """oracle/keccak_inplace.c (the in-place four-rounds-per-trip form of /root/reference/src/sha3/keccakf.rs:56-422, what
    bench.py's cpu_baseline times) against the textbook permutation, and the sponge built on either."""
    import hashlib
    import random

    rng = random.Random(0x1600)
    assert O.keccakf1600([0] * 25, inplace=True)[0] == 0xF1258F7940E1DDE7
    for _ in range(200):
        st = [rng.getrandbits(64) for _ in range(25)]
        assert O.keccakf1600(st, inplace=True) == O.keccakf1600(st)
    O.select_keccak(True)
    try:
        for n in (0, 1, 135, 136, 137, 5000):
            m = rng.randbytes(n)
            assert O.sha3(m, 256) == hashlib.sha3_256(m).digest()
            assert O.sha3(m, 512, quirks=0) == hashlib.sha3_512(m).digest()
    finally:
        O.select_keccak(False)


def _openssl_kmac():
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "openssl_kmac.json")) as f:
        return json.load(f)["kmac_xof"]


def kmac_conflicts_with_sp800_185(d, klen, xlen):
    """Where the reference's kmac_xof differs from SP 800-185: the cSHAKE suffix byte
    lands on a block boundary so that no 0x80 is ever added (sponge.rs:13), or bytepad(encode_string(K)) is already
    aligned and byte_pad appends a whole extra block (aux_functions.rs:14)."""
    w = (1600 - d) // 8
    enc = 2 + len(O.left_encode(8 * klen)) + klen  # left_encode(w) || left_encode(8|K|) || K
    return (xlen + 3) % w == 0 or enc % w == 0


def test_kmac_against_openssl_generated_vectors():
    """tests/golden/openssl_kmac.json: 96 KMACXOF128 / KMACXOF256 outputs from the OpenSSL command-line tool (an
    independent implementation; tests/golden/gen_openssl_kmac.py), lengths drawn around the rate boundaries.  The
    oracle's SP 800-185 mode (quirks = 0) must reproduce every one; its reference mode (quirks = 1, what the GPU path is
    checked against) must reproduce every one OUTSIDE the documented conflict set and differ inside it."""
    v = _openssl_kmac()
    assert len(v) == 96
    inside = 0
    for t in v:
        k, x, s = bytes.fromhex(t["k"]), bytes.fromhex(t["x"]), t["s"].encode()
        assert O.kmac_xof(k, x, t["l_bits"], s, t["d"], quirks=0).hex() == t["out"]
        ref = O.kmac_xof(k, x, t["l_bits"], s, t["d"], quirks=1).hex()
        if kmac_conflicts_with_sp800_185(t["d"], len(k), len(x)):
            inside += 1
            assert ref != t["out"], (t["d"], len(k), len(x))
        else:
            assert ref == t["out"], (t["d"], len(k), len(x))
    assert 8 <= inside <= 24
