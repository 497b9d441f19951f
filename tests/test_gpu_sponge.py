"""GPU parity tests of the sponge path, through the C ABI (capycrypt_amd -> libcapyhip.so), against
 (1) the reference's own known-answer vectors (tests/golden/reference_kats.json),
 (2) the CPU oracle on seeded inputs (bit-exact, reference quirks included),
 (3) size-independent properties at BASELINE.json's full sizes (5 MiB messages).
They read like the reference's tests (src/sha3/shake_functions.rs:92-288, tests/integration_tests.rs)."""
import hashlib
import random

import pytest

pytestmark = pytest.mark.gpu

MIB5 = 5242880


@pytest.fixture(scope="module")
def capy():
    import capycrypt_amd

    from capycrypt_amd import _lib

    assert _lib.lib().capy_device_count() >= 1, "no GPU visible"
    return capycrypt_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(autouse=True, params=[1, 2, 1 | (1 << 16), 2 | (1 << 8), 2 | (32 << 8), 1 | (64 << 8)],
                ids=["lane-per-sponge", "two-lanes-per-sponge", "lane-per-sponge,two-pass-encrypt",
                     "two-lanes,no-uniform-addressing", "wave-per-item-kernels", "lane-per-sponge,lds-staged-loads"])
def sponge_lanes(request):
    """Every test runs against both sponge kernels (sponge_kernels.h / sponge_kernels_k2.h), with the fused
    one-pass encrypt kernel (sponge_fused.h) on and off, with the wave-uniform addressing path off, and with the
    wave-per-item kernels (sponge_wide_il.h: encrypt / decrypt and digests) forced for every batch of up to 4096 items
    they can take, and with the wave-cooperative loads through LDS of round 1 (debug bit 6; the default is per-lane loads)."""
    from capycrypt_amd import _lib

    global _CURRENT_LANES
    _CURRENT_LANES = request.param
    _lib.check(_lib.lib().capy_set_sponge_lanes(request.param))
    yield request.param
    _CURRENT_LANES = 0
    _lib.check(_lib.lib().capy_set_sponge_lanes(0))


_CURRENT_LANES = 0


def sponge_lanes_current():
    return _CURRENT_LANES


# ---------------------------------------------------------------- (1) the reference's KATs, via the mirrored API
def test_shake_kats_via_message(capy, kats):
    for v in kats["sha3"]:  # test_shake_224/256/384/512, test_hashable
        data = capy.Message(bytes.fromhex(v["msg_hex"]))
        data.compute_sha3_hash(capy.SecParam.try_from(v["d"]))
        assert data.digest.hex() == v["digest_hex"], v["src"]


def test_compute_tagged_hash_kats(capy, kats):
    for v in kats["tagged_hash"]:
        data = capy.Message(bytes.fromhex(v["msg_hex"]))
        data.compute_tagged_hash(bytes.fromhex(v["pw_hex"]), v["s"], capy.SecParam.try_from(v["d"]))
        assert data.digest.hex() == v["digest_hex"], v["src"]


def test_cshake_kats(capy, kats):
    for v in kats["cshake"]:
        res = capy.cshake(bytes.fromhex(v["x_hex"]), v["l_bits"], v["n"], v["s"], v["d"])
        assert res.hex() == v["out_hex"], v["src"]


def test_kmac_kats(capy, kats):
    for v in kats["kmac_xof"]:
        res = capy.kmac_xof(bytes.fromhex(v["k_hex"]), bytes.fromhex(v["x_hex"]), v["l_bits"], v["s"], v["d"])
        assert res.hex() == v["out_hex"], v["src"]


def test_compute_sha3_hash_leaves_padding_in_msg(capy, O):
    m = capy.Message(b"test")
    m.compute_sha3_hash(capy.SecParam.D256)
    assert bytes(m.msg) == O.sha3(b"test", 256, want_padded=True)[1]


# ---------------------------------------------------------------- (2) seeded parity vs the oracle
LENS = [0, 1, 7, 8, 9, 63, 64, 71, 72, 73, 103, 104, 105, 135, 136, 137, 143, 144, 145, 151, 152, 165, 166, 167,
        168, 169, 171, 172, 173, 271, 272, 273, 335, 336, 337, 343, 344, 500, 1000, 4096, 10000, 70001]


@pytest.mark.parametrize("d", [224, 256, 384, 512])
def test_sha3_matches_oracle_all_lengths(capy, O, d):
    rng = random.Random(d)
    msgs = [rng.randbytes(n) for n in LENS + list(range(0, 300, 1))]
    assert capy.ops.sha3_batch(msgs, d) == [O.sha3(m, d) for m in msgs]


def test_sha3_256_is_fips202(capy):
    rng = random.Random(1)
    msgs = [rng.randbytes(n) for n in range(0, 700, 3)]
    assert capy.ops.sha3_batch(msgs, 256) == [hashlib.sha3_256(m).digest() for m in msgs]


@pytest.mark.parametrize("d", [224, 256, 384, 512])
def test_kmac_xof_matches_oracle(capy, O, d):
    rng = random.Random(100 + d)
    msgs = [rng.randbytes(n) for n in LENS]
    for klen, lbits, s in ((0, 256, b""), (32, 512, b"My Tagged Application"), (64, 8192, b"SKE"), (576, 1024, b"S"),
                           (131, 448, b"T")):
        keys = [rng.randbytes(klen) for _ in msgs]
        got = capy.ops.kmac_xof_batch(keys, msgs, lbits, s, d)
        assert got == [O.kmac_xof(k, m, lbits, s, d) for k, m in zip(keys, msgs)], (klen, lbits)


def test_kmac_for_the_aes_module_callers(capy, O):
    """SURVEY 8(f) rank 3, literally: the two kmac_xof calls of /root/reference/src/aes/encryptable.rs:38,42 (and
    :140,145 on the decrypt side) -- key derivation kmac_xof(iv || key, "", 512, "AES", D256) for 16-byte IVs with
    128/192/256-bit AES keys, then the tag kmac_xof(ka, msg, 512, "AES", D256) over the message -- as one batch each."""
    rng = random.Random(0xAE5)
    for keylen in (16, 24, 32):
        ivkeys = [rng.randbytes(16 + keylen) for _ in LENS]
        keka = capy.ops.kmac_xof_batch(ivkeys, [b""] * len(LENS), 512, b"AES", 256)
        assert keka == [O.kmac_xof(k, b"", 512, b"AES", 256) for k in ivkeys]
        kas = [kk[keylen:] for kk in keka]  # ka = the bytes behind ke (encryptable.rs:40)
        msgs = [rng.randbytes(n) for n in LENS]
        assert capy.ops.kmac_xof_batch(kas, msgs, 512, b"AES", 256) == [O.kmac_xof(k, m, 512, b"AES", 256) for k, m in zip(kas, msgs)]


@pytest.mark.parametrize("d", [224, 256, 384, 512])
def test_cshake_matches_oracle(capy, O, d):
    rng = random.Random(200 + d)
    msgs = [rng.randbytes(n) for n in LENS]
    for n, s, l in ((b"", b"Email Signature", 256), (b"KMAC", b"x" * 200, 2048), (b"fn", b"", 8)):
        assert capy.ops.cshake_batch(msgs, l, n, s, d) == [O.cshake(m, l, n, s, d) for m in msgs]


@pytest.mark.parametrize("d", [224, 256, 384, 512])
def test_cshake_empty_n_and_s_matches_the_reference_corner(capy, O, d):
    """cshake(x, l, "", "", d), /root/reference/src/sha3/shake_functions.rs:59-61: the dropped shake() call mutates
    the framed buffer (SHA3 suffix + pad to the SHA3-d rate) before the absorb at capacity d.  The oracle models it
    (oracle_sponge.c, quirks = 1); lengths on both rates' block boundaries and the 135-mod-136 suffix switch."""
    rng = random.Random(900 + d)
    w, r1 = (1600 - d) // 8, (1600 - 2 * d) // 8
    lens = sorted(set(LENS + [136 - (w + 1) % 136 + 135 - 136 * k for k in (0, -1)] + [r1 - 2, r1 - 1, r1, 2 * r1 - w % r1]
                      + _cshake_empty_aligned_lens(d)))
    msgs = [rng.randbytes(max(0, n)) for n in lens]
    for l in (8, 256, 1600):
        assert capy.ops.cshake_batch(msgs, l, b"", b"", d) == [O.cshake(m, l, b"", b"", d) for m in msgs]


def _cshake_empty_aligned_lens(d):
    """Message lengths for which the buffer the dropped shake() call leaves behind -- one block of w bytes, x, 04, the
    SHA3 suffix, pad10*1 to the SHA3-d rate r1 -- is ALSO a whole number of blocks of w bytes: the absorb at capacity d then
    ends on a block boundary with nothing appended.  (The wave-per-item digest kernel took such an item's digest from a stale
    state until tools/fuzz_soak.py found it in r04.)"""
    import math

    w, r1 = (1600 - d) // 8, (1600 - 2 * d) // 8
    m = w * r1 // math.gcd(w, r1)
    lo, hi = m - w - 2 - (r1 - 1), m - w - 2
    return [lo, lo + 1, (lo + hi) // 2, hi - 1, hi, hi + 1, 2 * m - w - 2 - 5]


@pytest.mark.parametrize("d", [256, 384, 512])
def test_cshake_empty_n_and_s_on_a_block_boundary_of_both_rates(capy, O, d):
    """The aligned corner of the corner (see _cshake_empty_aligned_lens) on its own: uniform batches of one, two, three and
    130 such messages (the wave-per-item kernel holds two items per wave; 130 leaves it) and a ragged one."""
    rng = random.Random(990 + d)
    lens = [n for n in _cshake_empty_aligned_lens(d)]
    for n_items in (1, 2, 3, 130):
        for ln in lens[:5]:
            msgs = [rng.randbytes(ln) for _ in range(n_items)]
            assert capy.ops.cshake_batch(msgs, 512, b"", b"", d) == [O.cshake(m, 512, b"", b"", d) for m in msgs], (n_items, ln)
    msgs = [rng.randbytes(ln) for ln in lens * 3]
    assert capy.ops.cshake_batch(msgs, 1088, b"", b"", d) == [O.cshake(m, 1088, b"", b"", d) for m in msgs]


@pytest.mark.parametrize("d", [224, 256, 384, 512])
def test_cshake_empty_n_and_s_device_form_matches_the_reference_corner(capy, O, d):
    """The same corner through capy_cshake_batch_dev (r03: answered CAPY_ERR_UNSUPPORTED): uniform batches (one length per
    call, with a message stride) and a ragged batch through device offsets, lengths on the block boundaries of both rates."""
    import ctypes as C

    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(950 + d)
    w, r1 = (1600 - d) // 8, (1600 - 2 * d) // 8
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lens = sorted(set([0, 1, 7, 8, 100, 135, 136, 1000] + [136 - (w + 1) % 136 + 135 - 136 * k for k in (0, -1)] + [r1 - 2, r1 - 1, r1,
                                                                                                            2 * r1 - w % r1]))
    lens = [n for n in lens + _cshake_empty_aligned_lens(d)[:4] if n >= 0]
    ol = 64
    for n_bytes in lens:  # uniform: 5 messages of this length, stride rounded up to 8 plus one slack word
        stride = (n_bytes + 7) // 8 * 8 + 8
        msgs = [rng.randbytes(n_bytes) for _ in range(5)]
        buf = torch.tensor(list(b"".join(m + bytes(stride - n_bytes) for m in msgs)), dtype=torch.uint8, device="cuda")
        out = torch.zeros(5 * ol, dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_cshake_batch_dev(d, 5, buf.data_ptr(), None, n_bytes, stride, 8 * ol, b"", 0, b"", 0, out.data_ptr(), ol, sp))
        got = bytes(out.cpu().numpy())
        assert [got[ol * i:ol * (i + 1)] for i in range(5)] == [O.cshake(m, 8 * ol, b"", b"", d) for m in msgs], n_bytes
    msgs = [rng.randbytes(n) for n in lens]
    offs, data = [0], b""
    for m in msgs:
        data += m
        offs.append(len(data))
    buf = torch.tensor(list(data + bytes(8)), dtype=torch.uint8, device="cuda")
    doff = torch.tensor(offs, dtype=torch.int64, device="cuda")
    out = torch.zeros(len(msgs) * ol, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_cshake_batch_dev(d, len(msgs), buf.data_ptr(), doff.data_ptr(), 0, 0, 8 * ol, b"", 0, b"", 0, out.data_ptr(), ol, sp))
    got = bytes(out.cpu().numpy())
    assert [got[ol * i:ol * (i + 1)] for i in range(len(msgs))] == [O.cshake(m, 8 * ol, b"", b"", d) for m in msgs]


def test_unsupported_security_parameter(capy):
    from capycrypt_amd._lib import CapyHipError

    with pytest.raises(CapyHipError) as e:
        capy.ops.sha3_batch([b"abc"], 300)
    assert e.value.code == -1
    with pytest.raises(capy.OperationError):
        capy.Message(b"x").compute_sha3_hash(300)


def test_empty_batch_and_empty_messages(capy, O):
    assert capy.ops.sha3_batch([], 256) == []
    assert capy.ops.kmac_xof_batch([], [], 256, b"", 256) == []
    assert capy.ops.sha3_batch([b""] * 130, 512) == [O.sha3(b"", 512)] * 130


@pytest.mark.parametrize("d", [224, 256, 384, 512])
def test_sha3_encrypt_decrypt_matches_oracle(capy, O, d):
    rng = random.Random(300 + d)
    msgs = [rng.randbytes(n) for n in LENS]
    pws = [rng.randbytes(64) for _ in msgs]
    zs = [rng.randbytes(512) for _ in msgs]
    cts, tags = capy.ops.sha3_encrypt_batch(pws, zs, msgs, d)
    exp = [O.sha3_encrypt(p, z, m, d) for p, z, m in zip(pws, zs, msgs)]
    assert cts == [e[0] for e in exp] and tags == [e[1] for e in exp]
    pts, ok = capy.ops.sha3_decrypt_batch(pws, zs, cts, tags, d)
    assert all(ok) and pts == msgs
    # wrong password on some items -> Err and ciphertext restored (tests/integration_tests.rs:250-262)
    pws2 = list(pws)
    for i in (0, 5, len(msgs) - 1):
        pws2[i] = rng.randbytes(64)
    pts, ok = capy.ops.sha3_decrypt_batch(pws2, zs, cts, tags, d)
    for i in range(len(msgs)):
        if i in (0, 5, len(msgs) - 1):
            assert not ok[i] and pts[i] == cts[i]
        else:
            assert ok[i] and pts[i] == msgs[i]


def test_kem_sponge_half_matches_oracle(capy, O):
    """kem_encrypt's sponge half (src/kem/encryptable.rs:51-57) = the same KMAC flow with
    tags KEMKE / KEMKA; the oracle side is composed from kmac_xof exactly as the reference composes it."""
    rng = random.Random(42)
    msgs = [rng.randbytes(n) for n in (0, 1, 135, 136, 1000, 70001)]
    secrets = [rng.randbytes(32) for _ in msgs]
    zs = [rng.randbytes(512) for _ in msgs]
    for d in (256, 512):
        cts, tags = capy.ops.kem_sponge_encrypt_batch(secrets, zs, msgs, d)
        for m, k, z, c, t in zip(msgs, secrets, zs, cts, tags):
            ke_ka = O.kmac_xof(z + k, b"", 1024, b"S", d)
            assert t == O.kmac_xof(ke_ka[64:], m, 512, b"KEMKA", d)
            ks = O.kmac_xof(ke_ka[:64], b"", len(m) * 8, b"KEMKE", d)
            assert c == bytes(a ^ b for a, b in zip(m, ks))
        pts, ok = capy.ops.kem_sponge_decrypt_batch(secrets, zs, cts, tags, d)
        assert all(ok) and pts == msgs
        bad = [secrets[0]] + [rng.randbytes(32)] + secrets[2:]
        pts, ok = capy.ops.kem_sponge_decrypt_batch(bad, zs, cts, tags, d)
        assert not ok[1] and pts[1] == cts[1] and ok[0] and ok[2]


def test_sha3_decrypt_handling_bad_input_like_reference(capy):
    pw1, pw2 = capy.get_random_bytes(64), capy.get_random_bytes(64)
    new_msg = capy.Message(capy.get_random_bytes(523))
    new_msg.sha3_encrypt(pw1, capy.SecParam.D512)
    msg2 = bytes(new_msg.msg)
    with pytest.raises(capy.OperationError) as e:
        new_msg.sha3_decrypt(pw2)
    assert e.value.variant == "SHA3DecryptionFailure" and bytes(new_msg.msg) == msg2


def test_batch_of_one_equals_batch_of_many(capy):
    rng = random.Random(9)
    msgs = [rng.randbytes(rng.randrange(0, 3000)) for _ in range(200)]
    many = capy.ops.sha3_batch(msgs, 256)
    for i in (0, 1, 63, 64, 65, 199):
        assert capy.ops.sha3_batch([msgs[i]], 256)[0] == many[i]


def test_config2_keystream_units(capy, O):
    """BASELINE config 2: unit = kmac_xof(k, "", 8192 bits, "SKE", D512), 64-byte keys."""
    rng = random.Random(0xCA9C0002)
    keys = [rng.randbytes(64) for _ in range(4096)]
    got = capy.ops.kmac_xof_batch(keys, [b""] * len(keys), 8192, b"SKE", 512)
    for i in list(range(0, 4096, 97)) + [4095]:
        assert got[i] == O.kmac_xof(keys[i], b"", 8192, b"SKE", 512)
    assert len(set(got)) == len(got)


@pytest.mark.parametrize("d,n,L,stride", [(256, 73728, 80000, 80000), (256, 66001, 70001, 70008), (512, 100000, 40003, 40008),
                                          (384, 120000, 60000, 60000), (224, 81920, 90007, 90008)])
def test_rotating_occupancy_schedule_matches_one_lane_kernel(capy, sponge_lanes, d, n, L, stride):
    """r04: uniform digest batches between one and two one-lane waves per SIMD (64 S < n < 128 S) with long messages take
    the rotating-occupancy schedule (sponge_rot.h: P phase launches of 512-lane workgroups, one per compute unit, a
    rotating subset of them doubled up, + a resume launch).  Every digest must equal the one-lane kernel's (forced lanes =
    1: one launch of the paired latency-tuned instance), and hashlib's where the reference is FIPS 202 for that length."""
    import ctypes as C
    import hashlib

    import torch

    from capycrypt_amd import _lib

    if sponge_lanes != 1:
        pytest.skip("sets the kernel choice itself")
    lib = _lib.lib()
    msgs = _dev_rand(n * stride, 13)
    outs = []
    for lanes in (1, 0):
        _lib.check(lib.capy_set_sponge_lanes(lanes))
        kind, phases = C.c_int(0), C.c_int(0)
        _lib.check(lib.capy_sha3_launch_plan(d, n, L, stride, C.byref(kind), C.byref(phases)))
        assert (kind.value, phases.value >= 2) == ((8, True) if lanes == 0 else (1, False)), (kind.value, phases.value)
        dig = torch.zeros(n * (d // 8), dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_sha3_batch_dev(d, n, msgs.data_ptr(), None, L, stride, dig.data_ptr(), None))
        torch.cuda.synchronize()
        outs.append(dig)
    assert torch.equal(outs[0], outs[1]), (d, n, L)
    if d == 256:
        hd = bytes(outs[1].cpu().numpy())
        for i in (0, 255, 256, n // 2 + 3, n - 257, n - 1):
            assert hd[32 * i:32 * i + 32] == hashlib.sha3_256(bytes(msgs[i * stride:i * stride + L].cpu().numpy())).digest(), i
    # the keyed form (KMACXOF: head blocks by a head-only launch first) against the one-lane kernel
    keys = _dev_rand(n * 64, 14)
    ko = []
    for lanes in (1, 0):
        _lib.check(lib.capy_set_sponge_lanes(lanes))
        out = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_kmac_xof_batch_dev(d, n, keys.data_ptr(), 64, 64, None, msgs.data_ptr(), None, L, stride, 512, b"T", 1,
                                               out.data_ptr(), 64, None))
        torch.cuda.synchronize()
        ko.append(out)
    assert torch.equal(ko[0], ko[1]), (d, n, L, "kmac")


@pytest.mark.parametrize("d", [512, 256, 384, 224])
def test_long_squeeze_leaves_as_whole_lines(capy, O, d):
    """r04: XOF squeezes of at least one 128-byte line per item leave the one-lane kernels as whole lines assembled in
    LDS (sponge_kernels.h, squeeze), not as rate blocks at their own offsets.  Device-pointer kmac_xof at output lengths
    around the line / block boundaries, row strides that do and do not allow 16-byte stores, full and partial waves, the
    latency-tuned and (n > 128 per SIMD) the issue-tuned instance: every row must equal the rate-block form's (debug
    bit 10) and sampled rows the oracle's; bytes between rows stay untouched."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    lanes = sponge_lanes_current()
    cases = [(200, 128, 128), (200, 136, 144), (200, 144, 160), (200, 1024, 1024), (200, 1040, 1056), (200, 2176, 2176),
             (200, 2304, 2320), (200, 1000, 1000), (200, 1024, 1032), (70000, 1024, 1024), (140032, 272, 288),
             (140000, 1024, 1024)]
    for n, out_len, stride in cases:
        if n > 1000 and d in (384, 224) and out_len != 1024:
            continue
        keys = _dev_rand(n * 64, 100 + out_len)
        hkeys = bytes(keys.cpu().numpy())
        outs = []
        # automatic choice (n > 128 per SIMD: sponge_uniform.h) / the fixture's kernel, whole lines / the same, rate blocks
        for flags in (0, lanes, lanes | (1 << 20)):
            _lib.check(lib.capy_set_sponge_lanes(flags))
            out = torch.full((n * stride + 16,), 0xA5, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_kmac_xof_batch_dev(d, n, keys.data_ptr(), 64, 64, None, None, None, 0, 0, 8 * out_len, b"SKE", 3,
                                                   out.data_ptr(), stride, None))
            torch.cuda.synchronize()
            outs.append(out)
        _lib.check(lib.capy_set_sponge_lanes(lanes))
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (n, out_len, stride)
        rows = outs[0][:n * stride].view(n, stride)
        assert bool((rows[:, out_len:] == 0xA5).all()) and bool((outs[0][n * stride:] == 0xA5).all()), (n, out_len, stride)
        for i in sorted({0, 1, 63, 64, 127, n // 2, n - 65, n - 2, n - 1}):
            assert bytes(rows[i, :out_len].cpu().numpy()) == O.kmac_xof(hkeys[64 * i:64 * i + 64], b"", 8 * out_len, b"SKE", d), \
                (n, out_len, stride, i)


# ---------------------------------------------------------------- device-pointer API, unaligned inputs
def test_dev_api_unaligned_and_strided(capy, O):
    import ctypes as C

    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(11)
    L, n = 1000, 100
    for shift in (0, 1, 3, 8):
        stride = 1024 + (8 if shift == 8 else 0)
        raw = bytearray(rng.randbytes(shift + n * stride))
        t = torch.tensor(list(raw), dtype=torch.uint8, device="cuda")
        out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_sha3_batch_dev(256, n, t.data_ptr() + shift, None, L, stride, out.data_ptr(), None))
        torch.cuda.synchronize()
        got = bytes(out.cpu().numpy())
        for i in range(n):
            m = bytes(raw[shift + i * stride: shift + i * stride + L])
            assert got[32 * i:32 * i + 32] == O.sha3(m, 256), (shift, i)


# ---------------------------------------------------------------- (3) full-size properties (5 MiB messages)
def test_config1_sha3_256_of_5mib(capy, O):
    rng = random.Random(0xCA9C0001)
    msg = rng.randbytes(MIB5)
    m = capy.Message(msg)
    m.compute_sha3_hash(capy.SecParam.D256)
    assert m.digest == hashlib.sha3_256(msg).digest() == O.sha3(msg, 256)


def test_5mib_batch_order_and_independence(capy):
    rng = random.Random(5)
    base = rng.randbytes(MIB5)
    msgs = []
    for i in range(70):  # more than one wave, ragged around 5 MiB
        cut = MIB5 - (i % 5) * 137
        msgs.append(base[:cut - 1] + bytes([i]))
    got = capy.ops.sha3_batch(msgs, 256)
    assert len(set(got)) == 70
    for i in (0, 33, 69):
        assert got[i] == hashlib.sha3_256(msgs[i]).digest()


def test_config3_encrypt_roundtrip_5mib(capy, O):
    """sha3_encrypt over 5 MiB messages (test_symmetric_encryptable / benchmark sym_enc): tag and first/last
    ciphertext blocks against the oracle for one message, round-trip + wrong-password restore for all."""
    rng = random.Random(0xCA9C0003)
    n = 6
    msgs = [rng.randbytes(MIB5 - i) for i in range(n)]
    pws = [rng.randbytes(64) for _ in range(n)]
    zs = [rng.randbytes(512) for _ in range(n)]
    cts, tags = capy.ops.sha3_encrypt_batch(pws, zs, msgs, 512)
    ect, etag = O.sha3_encrypt(pws[0], zs[0], msgs[0], 512)
    assert tags[0] == etag and cts[0] == ect
    assert all(len(c) == len(m) and c != m for c, m in zip(cts, msgs))
    pws_bad = list(pws)
    pws_bad[2] = rng.randbytes(64)
    pts, ok = capy.ops.sha3_decrypt_batch(pws_bad, zs, cts, tags, 512)
    assert ok == [True, True, False, True, True, True]
    assert pts[2] == cts[2] and all(pts[i] == msgs[i] for i in (0, 1, 3, 4, 5))


def test_cpp_host_mirror_selftest():
    """The C++ mirror of the reference interface (capycrypt_amd/host/capycrypt.hpp) run as its own process."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "capycrypt_amd", "host", "host_selftest")
    if not os.path.exists(exe):
        pytest.skip("host_selftest not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


# ---------------------------------------------------------------- device-pointer API: every kernel instance
def _dev_rand(nbytes, seed):
    import ctypes as C

    import torch

    from capycrypt_amd import _lib

    t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().capy_fill_random_dev(t.data_ptr(), t.numel(), seed, None))
    torch.cuda.synchronize()
    return t


@pytest.mark.parametrize("n,L", [(100, 1000), (4100, 2731 * 8), (70000, 1000), (140000, 304)])
def test_dev_api_uniform_batches_all_instances(capy, O, n, L):
    """Uniformly strided device batches (the bench layout) through sha3 / kmac_xof / sha3_encrypt / sha3_decrypt:
    n = 100 and 4100 use the small-batch kernels (two-lane or latency-tuned, fused encrypt), n = 140000 the
    issue-tuned full-chip instances.  Spot-checked against the oracle, round trip checked for every item."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    stride = (L + 15) // 16 * 16
    msgs = _dev_rand(n * stride, 7)
    keys = _dev_rand(n * 64, 8)
    zs = _dev_rand(n * 512, 9)
    host = bytes(msgs.cpu().numpy())
    hkeys = bytes(keys.cpu().numpy())
    hz = bytes(zs.cpu().numpy())
    picks = sorted({0, 1, 63, 64, n // 2, n - 2, n - 1})

    dig = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_sha3_batch_dev(256, n, msgs.data_ptr(), None, L, stride, dig.data_ptr(), None))
    out = torch.zeros(n * 72, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_kmac_xof_batch_dev(512, n, keys.data_ptr(), 64, 64, None, msgs.data_ptr(), None, L, stride, 576,
                                           b"T", 1, out.data_ptr(), 72, None))
    torch.cuda.synchronize()
    hd, ho = bytes(dig.cpu().numpy()), bytes(out.cpu().numpy())
    for i in picks:
        m = host[i * stride:i * stride + L]
        assert hd[32 * i:32 * i + 32] == O.sha3(m, 256), i
        assert ho[72 * i:72 * i + 72] == O.kmac_xof(hkeys[64 * i:64 * i + 64], m, 576, b"T", 512), i

    for d in (512, 256):
        work = msgs.clone()
        tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
        status = torch.full((n,), 7, dtype=torch.int32, device="cuda")
        _lib.check(lib.capy_sha3_encrypt_batch_dev(d, n, keys.data_ptr(), 64, None, 0, zs.data_ptr(), work.data_ptr(), None, L,
                                                   stride, tags.data_ptr(), None))
        torch.cuda.synchronize()
        hc, ht = bytes(work.cpu().numpy()), bytes(tags.cpu().numpy())
        for i in picks:
            ect, etag = O.sha3_encrypt(hkeys[64 * i:64 * i + 64], hz[512 * i:512 * i + 512], host[i * stride:i * stride + L], d)
            assert hc[i * stride:i * stride + L] == ect and ht[64 * i:64 * i + 64] == etag, (d, i)
            assert hc[i * stride + L:(i + 1) * stride] == host[i * stride + L:(i + 1) * stride]  # padding untouched
        # corrupt one tag: that item must fail and keep its ciphertext, all others decrypt
        bad = n // 3
        tags[64 * bad] ^= 1
        _lib.check(lib.capy_sha3_decrypt_batch_dev(d, n, keys.data_ptr(), 64, None, 0, zs.data_ptr(), work.data_ptr(), None, L,
                                                   stride, tags.data_ptr(), status.data_ptr(), None))
        torch.cuda.synchronize()
        st = status.cpu().numpy()
        assert st[bad] == 1 and int(st.sum()) == 1
        hp = bytes(work.cpu().numpy())
        assert hp[bad * stride:bad * stride + L] == hc[bad * stride:bad * stride + L]
        ref = bytearray(host)
        ref[bad * stride:bad * stride + L] = hc[bad * stride:bad * stride + L]
        assert hp == bytes(ref)


@pytest.mark.parametrize("d,n,L,stride", [(256, 40000, 200003, 200008), (512, 34001, 100001, 100008),
                                          (224, 57344, 160000, 160000), (384, 45000, 140007, 140008),
                                          (256, 60000, 150001, 150008)])
def test_rotating_schedule_matches_one_lane_kernel(capy, sponge_lanes, d, n, L, stride):
    """Batches between half a chip and a full chip of one-lane sponges take the rotating one-/two-lane schedule
    (sponge_mixed.h: P phase launches + a resume launch).  Every digest must equal the one-lane kernel's, and
    hashlib's where the reference is FIPS 202 for that length (d = 256 always; the other lengths here are off the
    135 (mod 136) / r-1 (mod r) cases of SURVEY.md 8a row 9).  Covers ragged last groups and partial waves."""
    import ctypes as C
    import hashlib

    import torch

    from capycrypt_amd import _lib

    if sponge_lanes != 1:
        pytest.skip("sets the kernel choice itself")
    lib = _lib.lib()
    msgs = _dev_rand(n * stride, 11)
    outs = []
    for lanes in (1, 3, 3 | (64 << 8)):  # one-lane kernel, rotating schedule, its LDS-staged A/B twin (debug bit 6)
        _lib.check(lib.capy_set_sponge_lanes(lanes))
        kind, phases = C.c_int(0), C.c_int(0)
        _lib.check(lib.capy_sha3_launch_plan(d, n, L, stride, C.byref(kind), C.byref(phases)))
        assert (kind.value, phases.value >= 2) == ((3, True) if lanes & 0xFF == 3 else (1, False))
        dig = torch.zeros(n * (d // 8), dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_sha3_batch_dev(d, n, msgs.data_ptr(), None, L, stride, dig.data_ptr(), None))
        torch.cuda.synchronize()
        outs.append(dig)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    h = getattr(hashlib, "sha3_%d" % d)
    hd = bytes(outs[1].cpu().numpy())
    for i in sorted({0, 31, 32, 63, 64, n // 2, n - 65, n - 2, n - 1}):
        m = bytes(msgs[i * stride:i * stride + L].cpu().numpy())
        assert hd[i * (d // 8):(i + 1) * (d // 8)] == h(m).digest(), i


def test_cshake_dev_api_and_rotating_schedule(capy, O, sponge_lanes):
    """capy_cshake_batch_dev: a small ragged-free batch against the oracle under every kernel choice, and (once) a
    batch large enough for the rotating schedule, whose shared prefix arrives folded into the initial state."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    n, L, stride = 300, 777, 784
    msgs = _dev_rand(n * stride, 21)
    host = bytes(msgs.cpu().numpy())
    for d, lbits in ((256, 512), (512, 256), (224, 448), (384, 136 * 8 * 2)):
        out = torch.zeros(n * (lbits // 8), dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_cshake_batch_dev(d, n, msgs.data_ptr(), None, L, stride, lbits, b"FN", 2, b"custom", 6,
                                             out.data_ptr(), lbits // 8, None))
        torch.cuda.synchronize()
        ho = bytes(out.cpu().numpy())
        for i in (0, 1, 63, 64, n - 1):
            assert ho[i * (lbits // 8):(i + 1) * (lbits // 8)] == O.cshake(host[i * stride:i * stride + L], lbits, b"FN",
                                                                          b"custom", d), (d, i)
    if sponge_lanes != 1:
        return
    n, L, stride = 40000, 150003, 150008
    big = _dev_rand(n * stride, 22)
    outs = []
    for lanes in (1, 3):
        _lib.check(lib.capy_set_sponge_lanes(lanes))
        out = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_cshake_batch_dev(512, n, big.data_ptr(), None, L, stride, 512, b"", 0, b"S", 1,
                                             out.data_ptr(), 64, None))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    ho = bytes(outs[1].cpu().numpy())
    for i in (0, 33, n - 1):
        m = bytes(big[i * stride:i * stride + L].cpu().numpy())
        assert ho[64 * i:64 * i + 64] == O.cshake(m, 512, b"", b"S", 512), i


def test_kmac_rotating_schedule_with_per_item_keys(capy, O, sponge_lanes):
    """compute_tagged_hash-sized batches (hashable.rs:33-35): per-item key blocks go through a head-only launch,
    the body through the rotating schedule, tail and squeeze through the resume launch."""
    import torch

    from capycrypt_amd import _lib

    if sponge_lanes != 1:
        pytest.skip("sets the kernel choice itself")
    lib = _lib.lib()
    for d, n, L, stride, klen in ((512, 40000, 150003, 150008, 64), (256, 36000, 180001, 180008, 200)):
        msgs = _dev_rand(n * stride, 31)
        keys = _dev_rand(n * klen, 32)
        outs = []
        for lanes in (1, 3):
            _lib.check(lib.capy_set_sponge_lanes(lanes))
            out = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_kmac_xof_batch_dev(d, n, keys.data_ptr(), klen, klen, None, msgs.data_ptr(), None, L, stride, 512,
                                                   b"T", 1, out.data_ptr(), 64, None))
            torch.cuda.synchronize()
            outs.append(out)
        assert torch.equal(outs[0], outs[1])
        ho, hk = bytes(outs[1].cpu().numpy()), bytes(keys.cpu().numpy())
        for i in (0, 32, n // 2, n - 1):
            m = bytes(msgs[i * stride:i * stride + L].cpu().numpy())
            assert ho[64 * i:64 * i + 64] == O.kmac_xof(hk[klen * i:klen * (i + 1)], m, 512, b"T", d), (d, i)


def test_wave_quantisation_split_matches_one_lane(capy, O, sponge_lanes):
    """Uniform batches between 64 and 128 sponges per SIMD: since r03 ONE launch of the paired latency-tuned instance
    (blocked round with priority; digest and keystream-XOR modes) -- the automatic choice --, before that a full-chip
    head + a remainder on the two-lane kernel (sponge.hip: launch_sponge; still reachable with bit 18 of
    capy_set_sponge_lanes, which also selects the plain round).  Forced one-lane, automatic and split launches must agree
    for every item in digests, KMAC outputs and sha3_encrypt (fused paired kernel / two-pass with bit 16), and with the
    oracle for a sample."""
    import torch

    from capycrypt_amd import _lib

    if sponge_lanes != 1:
        pytest.skip("sets the kernel choice itself")
    lib = _lib.lib()
    n, L, stride = 70003, 1000, 1008
    msgs = _dev_rand(n * stride, 41)
    keys = _dev_rand(n * 64, 42)
    zs = _dev_rand(n * 512, 43)
    res = []
    for lanes in (1, 0, 1 << 18, (1 << 18) | (1 << 16), 1 << 16):
        _lib.check(lib.capy_set_sponge_lanes(lanes))
        dig = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
        out = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
        work = msgs.clone()
        tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_sha3_batch_dev(256, n, msgs.data_ptr(), None, L, stride, dig.data_ptr(), None))
        _lib.check(lib.capy_kmac_xof_batch_dev(512, n, keys.data_ptr(), 64, 64, None, msgs.data_ptr(), None, L, stride, 512,
                                               b"T", 1, out.data_ptr(), 64, None))
        _lib.check(lib.capy_sha3_encrypt_batch_dev(512, n, keys.data_ptr(), 64, None, 0, zs.data_ptr(), work.data_ptr(), None, L,
                                                   stride, tags.data_ptr(), None))
        torch.cuda.synchronize()
        res.append((dig, out, work, tags))
    _lib.check(lib.capy_set_sponge_lanes(1))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)
    host, hk, hz = bytes(msgs.cpu().numpy()), bytes(keys.cpu().numpy()), bytes(zs.cpu().numpy())
    hd, hc, ht = (bytes(t.cpu().numpy()) for t in (res[1][0], res[1][2], res[1][3]))
    for i in (0, 65535, 65536, 65537, n - 1):
        m = host[i * stride:i * stride + L]
        assert hd[32 * i:32 * i + 32] == O.sha3(m, 256), i
        ect, etag = O.sha3_encrypt(hk[64 * i:64 * i + 64], hz[512 * i:512 * i + 512], m, 512)
        assert hc[i * stride:i * stride + L] == ect and ht[64 * i:64 * i + 64] == etag, i


def test_dev_api_ragged_offsets_longest_first(capy, O):
    """Ragged batches given as device offsets are processed longest-first (sponge.hip: device_order, a counting
    sort on a logarithmic length scale).  Outputs stay at the item's own index: every digest / KMAC output /
    ciphertext / tag is checked against the oracle, and a corrupted tag fails only its own item."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(77)
    n = 700
    lens = [rng.choice([0, 1, 7, 8, 135, 136, 137, 271, 272, 1000, 5000, 40000]) if rng.random() < 0.5
            else rng.randrange(0, 3000) for _ in range(n)]
    offs, pos = [], 0
    for x in lens:
        offs.append(pos)
        pos += (x + 7) // 8 * 8
    offs.append(pos)
    data = _dev_rand(max(8, pos), 51)
    host = bytes(data.cpu().numpy())
    d_offs = torch.tensor(offs, dtype=torch.int64, device="cuda")
    d_lens = None
    keys = _dev_rand(n * 64, 52)
    zs = _dev_rand(n * 512, 53)
    hk, hz = bytes(keys.cpu().numpy()), bytes(zs.cpu().numpy())
    # offsets[i+1]-offsets[i] is the padded length: pass exact lengths through a re-packed view instead -> use the
    # padded lengths as the message lengths (the padding bytes are part of the random buffer)
    plens = [offs[i + 1] - offs[i] for i in range(n)]
    msgs = [host[offs[i]:offs[i] + plens[i]] for i in range(n)]

    dig = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_sha3_batch_dev(256, n, data.data_ptr(), d_offs.data_ptr(), 0, 0, dig.data_ptr(), None))
    out = torch.zeros(n * 40, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_kmac_xof_batch_dev(256, n, keys.data_ptr(), 64, 64, None, data.data_ptr(), d_offs.data_ptr(), 0, 0, 320,
                                           b"T", 1, out.data_ptr(), 40, None))
    torch.cuda.synchronize()
    hd, ho = bytes(dig.cpu().numpy()), bytes(out.cpu().numpy())
    for i in range(n):
        assert hd[32 * i:32 * i + 32] == O.sha3(msgs[i], 256), (i, plens[i])
    for i in range(0, n, 7):
        assert ho[40 * i:40 * i + 40] == O.kmac_xof(hk[64 * i:64 * i + 64], msgs[i], 320, b"T", 256), i

    work = data.clone()
    tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
    status = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    _lib.check(lib.capy_sha3_encrypt_batch_dev(512, n, keys.data_ptr(), 64, None, 0, zs.data_ptr(), work.data_ptr(),
                                               d_offs.data_ptr(), 0, 0, tags.data_ptr(), None))
    torch.cuda.synchronize()
    hc, ht = bytes(work.cpu().numpy()), bytes(tags.cpu().numpy())
    for i in range(0, n, 5):
        ect, etag = O.sha3_encrypt(hk[64 * i:64 * i + 64], hz[512 * i:512 * i + 512], msgs[i], 512)
        assert hc[offs[i]:offs[i] + plens[i]] == ect and ht[64 * i:64 * i + 64] == etag, i
    bad = next(i for i in range(n // 2, n) if plens[i] > 0)
    tags[64 * bad + 3] ^= 0x10
    _lib.check(lib.capy_sha3_decrypt_batch_dev(512, n, keys.data_ptr(), 64, None, 0, zs.data_ptr(), work.data_ptr(),
                                               d_offs.data_ptr(), 0, 0, tags.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert st[bad] == 1 and int(st.sum()) == 1
    hp = bytes(work.cpu().numpy())
    ref = bytearray(host)
    ref[offs[bad]:offs[bad] + plens[bad]] = hc[offs[bad]:offs[bad] + plens[bad]]
    assert hp == bytes(ref)


def test_ragged_device_batch_above_128_sponges_per_simd(capy):
    """150 000 short ragged messages given as device offsets (more than 128 sponges per SIMD: ragged batches stay on
    the latency-tuned one-lane instance, in longest-first order; the fixture also runs the two-lane kernel); every 97th
    digest is checked with hashlib (SHA3-256 is FIPS 202 at every length)."""
    import numpy as np
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    n = 150000
    rng = np.random.default_rng(5)
    lens = (rng.integers(0, 60, n) * 8).astype(np.int64)  # device offsets imply lengths: multiples of 8 here
    lens[::1000] = 4096
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    data = _dev_rand(int(offs[-1]) + 8, 61)
    d_offs = torch.from_numpy(offs).cuda()
    dig = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_sha3_batch_dev(256, n, data.data_ptr(), d_offs.data_ptr(), 0, 0, dig.data_ptr(), None))
    torch.cuda.synchronize()
    host, hd = bytes(data.cpu().numpy()), bytes(dig.cpu().numpy())
    for i in range(0, n, 97):
        assert hd[32 * i:32 * i + 32] == hashlib.sha3_256(host[offs[i]:offs[i + 1]]).digest(), i
    # the processing order must be a permutation of the batch: every digest equals the input-order run's
    flags = lib.capy_set_sponge_lanes  # (the fixture restores the setting afterwards)
    _lib.check(flags(1 | (4 << 8)))
    dig2 = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_sha3_batch_dev(256, n, data.data_ptr(), d_offs.data_ptr(), 0, 0, dig2.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(dig, dig2)


def test_host_ragged_batch_larger_than_a_neighbourhood(capy):
    """10 000 ragged messages through the host-buffer API (more than two 4096-item neighbourhoods of the processing
    order, unaligned starts -> re-packed): every SHA3-256 digest against hashlib, KMAC outputs must be pairwise
    consistent with a second call in reverse order."""
    rng = random.Random(4242)
    msgs = [rng.randbytes(rng.choice([0, 1, 7, 8, 135, 136, 137, 300, 1000]) if rng.random() < 0.4 else rng.randrange(0, 600))
            for _ in range(10000)]
    got = capy.ops.sha3_batch(msgs, 256)
    assert got == [hashlib.sha3_256(m).digest() for m in msgs]
    keys = [rng.randbytes(32) for _ in msgs]
    a = capy.ops.kmac_xof_batch(keys, msgs, 256, b"T", 512)
    b = capy.ops.kmac_xof_batch(keys[::-1], msgs[::-1], 256, b"T", 512)
    assert a == b[::-1]


def test_device_fill_equals_host_harness_prng(capy, sponge_lanes):
    """capy_fill_random_dev and capycrypt_amd.harness_prng produce the same stream (any shard's inputs
    can be regenerated on either side)."""
    from capycrypt_amd import harness_prng as H

    if sponge_lanes != 1:
        pytest.skip("independent of the sponge kernel choice")
    for seed, nbytes in ((0, 64), (0xCA9C0001, 1 << 16), (2**64 - 5, 8 * 1000)):
        t = _dev_rand(nbytes, seed)
        assert bytes(t.cpu().numpy()) == H.fill(seed, nbytes)


def test_concurrent_host_threads(capy, O):
    """include/capyhip.h promises thread safety: four host threads issue different batched calls at once
    (ctypes drops the GIL during the call) and every result must still be bit-exact."""
    import threading

    rng = random.Random(77)
    jobs = []
    for t in range(4):
        msgs = [rng.randbytes(rng.randrange(0, 4000)) for _ in range(150 + 37 * t)]
        keys = [rng.randbytes(32) for _ in msgs]
        jobs.append((msgs, keys, 256 if t % 2 else 512))
    results, errors = [None] * 4, []

    def work(i):
        try:
            msgs, keys, d = jobs[i]
            out = []
            for _ in range(5):
                out.append((capy.ops.sha3_batch(msgs, d), capy.ops.kmac_xof_batch(keys, msgs, 512, b"T", d),
                            capy.ops.sha3_encrypt_batch(keys, [k * 16 for k in keys], msgs, d)))
            results[i] = out
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for i, (msgs, keys, d) in enumerate(jobs):
        exp_sha = [O.sha3(m, d) for m in msgs]
        exp_kmac = [O.kmac_xof(k, m, 512, b"T", d) for k, m in zip(keys, msgs)]
        exp_enc = [O.sha3_encrypt(k, k * 16, m, d) for k, m in zip(keys, msgs)]
        for sha, kmac, (cts, tags) in results[i]:
            assert sha == exp_sha and kmac == exp_kmac
            assert cts == [e[0] for e in exp_enc] and tags == [e[1] for e in exp_enc]


def test_d224_prefix_is_staged_without_synchronising(capy, O, sponge_lanes):
    """cSHAKE / KMAC at d = 224 carry raw prefix bytes (r = 172 but 168 consumed per block, sponge.rs:48-55).  The *_dev
    entry points stage them through the argument of a writer kernel -- stream-ordered, no host synchronisation -- and
    fall back to a synchronous copy only for customisation strings beyond the inline buffer.  Back-to-back launches with
    DIFFERENT prefixes on one stream must each see their own bytes."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    n, L, stride = 200, 500, 504
    msgs = _dev_rand(n * stride, 61)
    keys = _dev_rand(n * 32, 62)
    customs = [b"", b"A", b"Email Signature", b"x" * 100, b"y" * 161, b"z" * 163, b"w" * 400]
    outs = [torch.zeros(n * 64, dtype=torch.uint8, device="cuda") for _ in customs]
    for cs, out in zip(customs, outs):  # all enqueued before any is read back
        _lib.check(lib.capy_kmac_xof_batch_dev(224, n, keys.data_ptr(), 32, 32, None, msgs.data_ptr(), None, L, stride, 512,
                                               cs, len(cs), out.data_ptr(), 64, None))
    torch.cuda.synchronize()
    hk = bytes(keys.cpu().numpy())
    for cs, out in zip(customs, outs):
        ho = bytes(out.cpu().numpy())
        for i in (0, 63, 64, n - 1):
            m = bytes(msgs[i * stride:i * stride + L].cpu().numpy())
            assert ho[64 * i:64 * i + 64] == O.kmac_xof(hk[32 * i:32 * i + 32], m, 512, cs, 224), (len(cs), i)


def test_device_key_offsets_and_workspace_release(capy, O, sponge_lanes):
    """Per-item key lengths through the DEVICE entry points (key_offsets / pw_offsets are device arrays), and
    capy_release_workspace(): scratch is handed back and the next call simply allocates again."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(5)
    n, L, stride = 150, 300, 304
    msgs = _dev_rand(n * stride, 71)
    klens = [rng.choice([0, 1, 31, 64, 130, 131, 132, 200, 333]) for _ in range(n)]
    keys_h = [rng.randbytes(k) for k in klens]
    offs = [0]
    for k in klens:
        offs.append(offs[-1] + k)
    kbuf = torch.tensor(list(b"".join(keys_h)) or [0], dtype=torch.uint8, device="cuda")
    koff = torch.tensor(offs, dtype=torch.int64, device="cuda")
    out = torch.zeros(n * 56, dtype=torch.uint8, device="cuda")
    zs = _dev_rand(n * 512, 72)
    tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
    status = torch.ones(n, dtype=torch.int32, device="cuda")
    plain = msgs.clone()
    for round_ in range(2):
        _lib.check(lib.capy_kmac_xof_batch_dev(512, n, kbuf.data_ptr(), 0, 0, koff.data_ptr(), msgs.data_ptr(), None, L, stride,
                                               448, b"T", 1, out.data_ptr(), 56, None))
        torch.cuda.synchronize()
        ho = bytes(out.cpu().numpy())
        for i in range(0, n, 7):
            m = bytes(msgs[i * stride:i * stride + L].cpu().numpy())
            assert ho[56 * i:56 * i + 56] == O.kmac_xof(keys_h[i], m, 448, b"T", 512), (klens[i], i)
        _lib.check(lib.capy_sha3_encrypt_batch_dev(256, n, kbuf.data_ptr(), 0, koff.data_ptr(), offs[-1], zs.data_ptr(),
                                                   msgs.data_ptr(), None, L, stride, tags.data_ptr(), None))
        torch.cuda.synchronize()
        hz = bytes(zs.cpu().numpy())
        for i in (0, 1, 77, n - 1):
            ect, etag = O.sha3_encrypt(keys_h[i], hz[512 * i:512 * i + 512], bytes(plain[i * stride:i * stride + L].cpu().numpy()), 256)
            assert bytes(msgs[i * stride:i * stride + L].cpu().numpy()) == ect
            assert bytes(tags[64 * i:64 * i + 64].cpu().numpy()) == etag
        _lib.check(lib.capy_sha3_decrypt_batch_dev(256, n, kbuf.data_ptr(), 0, koff.data_ptr(), offs[-1], zs.data_ptr(),
                                                   msgs.data_ptr(), None, L, stride, tags.data_ptr(), status.data_ptr(), None))
        torch.cuda.synchronize()
        assert not status.cpu().numpy().any() and torch.equal(msgs, plain)
        _lib.check(lib.capy_release_workspace())  # the second round runs on freshly allocated scratch


def test_kmac_against_openssl_generated_vectors_on_gpu(capy):
    """The OpenSSL-generated KMACXOF fixture through the C ABI: per security parameter ONE batch per output length with
    ragged key and message lengths; outside the documented conflict set (where the reference itself deviates from
    SP 800-185) every output must equal OpenSSL's."""
    from test_oracle_sponge import _openssl_kmac, kmac_conflicts_with_sp800_185

    v = [t for t in _openssl_kmac() if not kmac_conflicts_with_sp800_185(t["d"], len(t["k"]) // 2, len(t["x"]) // 2)]
    assert len(v) >= 70
    groups = {}
    for t in v:
        groups.setdefault((t["d"], t["l_bits"], t["s"]), []).append(t)
    for (d, l_bits, s), ts in groups.items():
        got = capy.ops.kmac_xof_batch([bytes.fromhex(t["k"]) for t in ts], [bytes.fromhex(t["x"]) for t in ts], l_bits,
                                      s.encode(), d)
        assert [g.hex() for g in got] == [t["out"] for t in ts], (d, l_bits, s)


def test_small_ragged_batches_of_long_messages(capy, O, sponge_lanes):
    """Batches of a few hundred messages with widely different lengths (0 .. 300 KB): in automatic mode these take the
    wave-per-item digest kernel (two items of different length per wave, length-sorted order), under the forced modes
    the one- and two-lane kernels.  SHA3-256 against hashlib (FIPS 202 = the reference at d = 256), KMAC and SHA3-512
    against the oracle for a sample."""
    import hashlib

    from capycrypt_amd import _lib

    rng = random.Random(0x51AB)
    n = 301
    lens = [rng.choice([0, 1, 135, 136, 137, 4096]) if i % 5 == 0 else rng.randrange(0, 300000) for i in range(n)]
    msgs = [rng.randbytes(x) for x in lens]
    keys = [rng.randbytes(rng.randrange(0, 200)) for _ in range(n)]
    for auto in (False, True):
        if auto:
            _lib.check(_lib.lib().capy_set_sponge_lanes(0))  # automatic choice: the wave-per-item kernel
        assert capy.ops.sha3_batch(msgs, 256) == [hashlib.sha3_256(m).digest() for m in msgs]
        got5 = capy.ops.sha3_batch(msgs, 512)
        gotk = capy.ops.kmac_xof_batch(keys, msgs, 448, b"T", 512)
        for i in range(0, n, 13):
            assert got5[i] == O.sha3(msgs[i], 512), lens[i]
            assert gotk[i] == O.kmac_xof(keys[i], msgs[i], 448, b"T", 512), (len(keys[i]), lens[i])


@pytest.mark.parametrize("d", [224, 256, 384, 512])
def test_uniform_framing_kernel_matches_generic_kernel(capy, O, sponge_lanes, d):
    """Uniform, 8-byte aligned digest batches of more than 128 items per SIMD take sponge_uniform.h (wave-uniform
    framing decided by scalar code; r02-r03: sponge_short.h for at most four blocks).  Every digest must equal the generic kernel's
    (debug bit 7) at the lengths where the framing changes shape -- empty, one byte, word boundaries, r - 1, r, r + 1
    around every block boundary, the reference's 135 (mod 136) suffix rule -- and the oracle's for sampled items;
    cSHAKE with a one-block output goes the same way."""
    import ctypes as C

    import torch

    from capycrypt_amd import _lib

    if sponge_lanes != 1:
        pytest.skip("sets the kernel choice itself")
    lib = _lib.lib()
    r = (1600 - 2 * d) // 8
    n = 140000
    lengths = sorted({0, 1, 7, 8, 9, 63, 64, 135, 136, 137, r - 8, r - 1, r, r + 1, 2 * r - 1, 2 * r, 2 * r + 7, 271, 3 * r - 2,
                      3 * r, 4 * r - 2, 4 * r, 5 * r + 3, 11 * r - 1})
    for L in lengths:
        stride = max(8, (L + 7) // 8 * 8)
        msgs = _dev_rand(n * stride, 100 + L)
        outs = []
        for lanes in (0, 128 << 8):
            _lib.check(lib.capy_set_sponge_lanes(lanes))
            kind, phases = C.c_int(0), C.c_int(0)
            _lib.check(lib.capy_sha3_launch_plan(d, n, L, stride, C.byref(kind), C.byref(phases)))
            # the plan shares its predicates with the launcher since r03: dense 28-byte digests (d = 224) are not
            # 8-byte aligned, so that batch really runs on the generic kernel -- and the plan now says so
            assert (kind.value == 7) == (lanes == 0 and (d // 8) % 8 == 0), (L, lanes, kind.value)
            dig = torch.zeros(n * (d // 8), dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_sha3_batch_dev(d, n, msgs.data_ptr(), None, L, stride, dig.data_ptr(), None))
            torch.cuda.synchronize()
            outs.append(dig)
        assert torch.equal(outs[0], outs[1]), (d, L)
        hd = bytes(outs[0].cpu().numpy())
        for i in (0, 63, 64, n - 1):
            m = bytes(msgs[i * stride:i * stride + L].cpu().numpy())
            assert hd[i * (d // 8):(i + 1) * (d // 8)] == O.sha3(m, d), (d, L, i)
    # cSHAKE (prefix folded into the initial state, suffix 04), output of one rate block at most
    L, stride = 100, 104
    msgs = _dev_rand(n * stride, 7)
    host = bytes(msgs[:4 * stride].cpu().numpy())
    lbits = 256
    outs = []
    for lanes in (0, 128 << 8):
        _lib.check(lib.capy_set_sponge_lanes(lanes))
        out = torch.zeros(n * (lbits // 8), dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_cshake_batch_dev(d, n, msgs.data_ptr(), None, L, stride, lbits, b"FN", 2, b"custom", 6,
                                             out.data_ptr(), lbits // 8, None))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    ho = bytes(outs[0].cpu().numpy())
    for i in range(4):
        assert ho[i * 32:(i + 1) * 32] == O.cshake(host[i * stride:i * stride + L], lbits, b"FN", b"custom", d), (d, i)
    # KMACXOF with equally long keys (the head is built from aligned key words with one launch-wide funnel shift), bodies
    # from empty to several blocks, one-block and multi-line outputs
    rk = (1600 - d) // 8
    for klen, L, out_len in [(64, 0, 1024), (56, 1000, 64), (8, rk - 3, 64), (24, rk - 4, 48), (32, 2 * rk, 160), (136, 5, 64),
                             (168 - 8, 7, 64), (400, 3 * rk + 1, 256), (64, 8 * rk - 3, 64), (0, 9, 64)]:
        stride = max(8, (L + 7) // 8 * 8)
        kstride = max(8, klen)
        msgs, keys = _dev_rand(n * stride, 300 + L), _dev_rand(n * kstride, 400 + klen)
        outs = []
        for lanes in (0, 128 << 8):
            _lib.check(lib.capy_set_sponge_lanes(lanes))
            out = torch.zeros(n * out_len, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_kmac_xof_batch_dev(d, n, keys.data_ptr(), klen, kstride, None, msgs.data_ptr(), None, L, stride,
                                                   8 * out_len, b"T", 1, out.data_ptr(), out_len, None))
            torch.cuda.synchronize()
            outs.append(out)
        assert torch.equal(outs[0], outs[1]), (d, klen, L, out_len)
        for i in (0, 63, 64, n - 1):
            k = bytes(keys[i * kstride:i * kstride + klen].cpu().numpy())
            m = bytes(msgs[i * stride:i * stride + L].cpu().numpy())
            assert bytes(outs[0][i * out_len:(i + 1) * out_len].cpu().numpy()) == O.kmac_xof(k, m, 8 * out_len, b"T", d), (d, klen, L, i)


def test_wave_per_item_digest_pairs_of_unequal_length(capy, O, sponge_lanes):
    """The wave-per-item digest kernel switches between a tight body loop over directly loaded blocks and the generic step
    (r03-r04: two items per wave, each half in its own phase; since r05 one item per wave, sponge_il_digest_kernel).  Unsorted
    device batches (no length sort on this path) whose neighbours differ wildly -- long next to empty, one block next to thousands, aligned next to unaligned
    offsets, an odd item count -- must still give hashlib's digests (SHA3-256) and the oracle's (SHA3-512: the reference's
    135 mod 136 suffix rule applies at every d), with the kernel forced (debug bit 5)."""
    import hashlib

    import torch

    from capycrypt_amd import _lib

    if sponge_lanes != 1:
        pytest.skip("sets the kernel choice itself")
    lib = _lib.lib()
    rng = random.Random(0xD1E5)
    base = [200000, 0, 1, 136 * 700, 136 * 700 + 8, 135, 136, 137, 99999, 136 * 3, 7, 136 * 1200 - 1, 64, 136 * 2 + 8, 150001]
    for align in (8, 1):
        lens = list(base)
        offs, pos = [], 0
        for x in lens:
            offs.append(pos)
            pos += x if align == 1 else (x + 7) // 8 * 8
        total = pos
        blob = rng.randbytes(total + 8)
        # the C ABI takes n + 1 offsets and lens = differences: lay the messages out back to back at the chosen alignment
        # by hashing the padded extents when align == 8 (the padding bytes are part of the message then)
        ext = [(offs[i + 1] if i + 1 < len(offs) else total) - offs[i] for i in range(len(offs))]
        data = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
        d_offs = torch.tensor(offs + [total], dtype=torch.int64).cuda()
        n = len(lens)
        for d, ref in ((256, lambda m: hashlib.sha3_256(m).digest()), (512, lambda m: O.sha3(m, 512))):
            dig = torch.zeros(n * d // 8, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_set_sponge_lanes(32 << 8))
            try:
                _lib.check(lib.capy_sha3_batch_dev(d, n, data.data_ptr(), d_offs.data_ptr(), 0, 0, dig.data_ptr(), None))
                torch.cuda.synchronize()
            finally:
                _lib.check(lib.capy_set_sponge_lanes(0))
            got = bytes(dig.cpu().numpy())
            for i in range(n):
                assert got[i * d // 8:(i + 1) * d // 8] == ref(blob[offs[i]:offs[i] + ext[i]]), (align, d, i, ext[i])
