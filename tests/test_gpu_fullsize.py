"""Size-independent properties at BASELINE.json's full per-GPU sizes, through the device-pointer C ABI:
the oracle cannot check these batches item by item, so each test pairs a structural property that must hold for
EVERY item (round trip, batch-size independence, exactly-one-failure) with oracle / hashlib checks on a sample."""
import ctypes as C
import hashlib
import random

import pytest

pytestmark = pytest.mark.gpu
MIB5 = 5242880


def _rand(nbytes, seed):
    import torch

    from capycrypt_amd import _lib

    t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().capy_fill_random_dev(t.data_ptr(), t.numel(), seed, None))
    return t


def test_config1_full_hbm_batch_is_batch_size_independent():
    """As many 5 MiB messages as bench.py hashes (54 528, or what fits): EVERY digest must equal the one a second kernel family
    computes over the same resident batch (one lane per sponge in one launch instead of the rotating one-/two-lane schedule in
    phase launches; asserted through capy_debug_last_sponge_kernel), a sampled sub-batch hashed again on its own (a third
    choice: two lanes per sponge) must be identical, and three digests are checked with hashlib."""
    import ctypes as C

    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    stride = MIB5 + 128
    free, _ = torch.cuda.mem_get_info()
    n = min(54528, int((free - (8 << 30)) // stride) // 2048 * 2048)
    assert n >= 2048
    msgs = _rand(n * stride, 0xCA9C0001)
    dig = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_sha3_batch_dev(256, n, msgs.data_ptr(), None, MIB5, stride, dig.data_ptr(), None))
    kind, launches = C.c_int(0), C.c_int(0)
    lib.capy_debug_last_sponge_kernel(C.byref(kind), C.byref(launches))
    first = (kind.value, launches.value)
    dig_all = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    try:
        _lib.check(lib.capy_set_sponge_lanes(1))
        _lib.check(lib.capy_sha3_batch_dev(256, n, msgs.data_ptr(), None, MIB5, stride, dig_all.data_ptr(), None))
        lib.capy_debug_last_sponge_kernel(C.byref(kind), C.byref(launches))
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))
    torch.cuda.synchronize()
    assert kind.value == 1 and launches.value == 1 and (n != 54528 or first[0] == 3), (first, kind.value, launches.value)
    assert torch.equal(dig, dig_all)
    del dig_all
    sub0, m = n - 1500, 1024  # a window reaching into the last groups of the schedule
    dig2 = torch.zeros(m * 32, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_sha3_batch_dev(256, m, msgs.data_ptr() + sub0 * stride, None, MIB5, stride, dig2.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(dig[sub0 * 32:(sub0 + m) * 32], dig2)
    hd = bytes(dig.cpu().numpy())
    for i in (0, n // 2 + 1, n - 1):
        assert hd[32 * i:32 * i + 32] == hashlib.sha3_256(bytes(msgs[i * stride:i * stride + MIB5].cpu().numpy())).digest()
    del msgs, dig, dig2
    torch.cuda.empty_cache()  # hand the 286 GB back: the library allocates outside torch's cache


def test_config2_full_batch_units_are_batch_size_independent():
    """2^20 keystream units (kmac_xof(k, "", 8192 bits, "SKE", D512)): a sampled run of 512 units recomputed as its own
    batch (two-lane kernel instead of the issue-tuned one) is identical; a few units are checked against the oracle."""
    import torch

    from capycrypt_amd import _lib
    from oracle import oracle as O

    lib = _lib.lib()
    n = 1 << 20
    keys = _rand(n * 64, 0xCA9C0002)
    out = torch.zeros(n * 1024, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_kmac_xof_batch_dev(512, n, keys.data_ptr(), 64, 64, None, None, None, 0, 0, 8192, b"SKE", 3,
                                           out.data_ptr(), 1024, None))
    s0, m = 777777, 512
    out2 = torch.zeros(m * 1024, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_kmac_xof_batch_dev(512, m, keys.data_ptr() + s0 * 64, 64, 64, None, None, None, 0, 0, 8192, b"SKE", 3,
                                           out2.data_ptr(), 1024, None))
    torch.cuda.synchronize()
    assert torch.equal(out[s0 * 1024:(s0 + m) * 1024], out2)
    hk = bytes(keys.cpu().numpy())
    for i in (0, s0 + 5, n - 1):
        assert bytes(out[i * 1024:(i + 1) * 1024].cpu().numpy()) == O.kmac_xof(hk[64 * i:64 * i + 64], b"", 8192, b"SKE", 512)


def test_config3_per_gpu_share_round_trips():
    """128 x 5 MiB (the 8-GPU split of config 3): sha3_encrypt then sha3_decrypt restores every byte; with one tag
    corrupted exactly that item fails and keeps its ciphertext."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    n = 128
    msgs = _rand(n * MIB5, 0xCA9C0003)
    plain = msgs.clone()
    pws, zs = _rand(n * 64, 31), _rand(n * 512, 32)
    tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
    status = torch.full((n,), 5, dtype=torch.int32, device="cuda")
    _lib.check(lib.capy_sha3_encrypt_batch_dev(512, n, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None, MIB5, MIB5,
                                               tags.data_ptr(), None))
    torch.cuda.synchronize()
    assert not torch.equal(msgs, plain)
    cipher = msgs.clone()
    tags[64 * 77 + 9] ^= 4
    _lib.check(lib.capy_sha3_decrypt_batch_dev(512, n, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None, MIB5, MIB5,
                                               tags.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert st[77] == 1 and int(st.sum()) == 1
    ok = torch.ones(n, dtype=torch.bool)
    ok[77] = False
    m2, p2, c2 = msgs.view(n, MIB5), plain.view(n, MIB5), cipher.view(n, MIB5)
    assert torch.equal(m2[ok.cuda()], p2[ok.cuda()]) and torch.equal(m2[77], c2[77])


def test_config3_as_specified_whole_batch_on_one_gpu():
    """BASELINE config 3 AS SPECIFIED, all of it on one GPU (r06): 1024 x 5 MiB through sha3_encrypt D512 -- two
    waves per item, bit-interleaved Keccak lanes (sponge_il_crypt_kernel, kind 27; /root/reference/src/sha3/encryptable.rs:29-45).
    EVERY tag and EVERY ciphertext byte equal the two-pass form's (tag kernel + keystream kernel: another kernel family
    altogether), three items equal the oracle's, and sha3_decrypt (:58-83) restores every plaintext byte except the one item whose
    tag was forged, which keeps its ciphertext."""
    import ctypes as C

    import torch

    from capycrypt_amd import _lib
    from oracle import oracle

    lib = _lib.lib()
    n = 1024
    plain = _rand(n * MIB5, 0xCA9C0003)
    pws, zs = _rand(n * 64, 31), _rand(n * 512, 32)
    outs, kinds = [], []
    try:
        for flags in (0, 1 | (1 << 16)):
            _lib.check(lib.capy_set_sponge_lanes(flags))
            work = plain.clone()
            tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_sha3_encrypt_batch_dev(512, n, pws.data_ptr(), 64, None, 0, zs.data_ptr(), work.data_ptr(), None, MIB5, MIB5,
                                                       tags.data_ptr(), None))
            torch.cuda.synchronize()
            k, l = C.c_int(0), C.c_int(0)
            lib.capy_debug_last_sponge_kernel(C.byref(k), C.byref(l))
            kinds.append(k.value)
            outs.append((work, tags))
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))
    assert kinds == [27, 26], kinds  # two waves per item; the two-pass form
    assert torch.equal(outs[0][1], outs[1][1]), "tags differ from the two-pass form"
    assert torch.equal(outs[0][0], outs[1][0]), "ciphertexts differ from the two-pass form"
    work, tags = outs[0]
    del outs
    hp, hz, ht = bytes(pws.cpu().numpy()), bytes(zs.cpu().numpy()), bytes(tags.cpu().numpy())
    for i in (0, 517, n - 1):
        ect, etag = oracle.sha3_encrypt(hp[64 * i:64 * i + 64], hz[512 * i:512 * i + 512], bytes(plain[i * MIB5:(i + 1) * MIB5].cpu().numpy()), 512)
        assert bytes(work[i * MIB5:(i + 1) * MIB5].cpu().numpy()) == ect and ht[64 * i:64 * i + 64] == etag, i
    bad = 700
    cipher_bad = work[bad * MIB5:(bad + 1) * MIB5].clone()
    tags[64 * bad + 63] ^= 0x80
    status = torch.full((n,), 5, dtype=torch.int32, device="cuda")
    _lib.check(lib.capy_sha3_decrypt_batch_dev(512, n, pws.data_ptr(), 64, None, 0, zs.data_ptr(), work.data_ptr(), None, MIB5, MIB5,
                                               tags.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert st[bad] == 1 and int(st.sum()) == 1
    w2, p2 = work.view(n, MIB5), plain.view(n, MIB5)
    ok = torch.ones(n, dtype=torch.bool, device="cuda")
    ok[bad] = False
    assert torch.equal(w2[ok], p2[ok]) and torch.equal(w2[bad], cipher_bad)


def test_config3_saturating_batch_fused_paired_kernel(O=None):
    """Config 3's GPU-saturating variant: 32 768 x 1 MiB through sha3_encrypt -- since r06 one lone wave per SIMD of the
    one-lane-per-sponge fused kernel (r03-r05: two waves per SIMD of the four-lane kernel).  Every tag must equal the two-pass form's (tag kernel +
    keystream kernel, bit 16 of capy_set_sponge_lanes), every ciphertext byte too; decrypt restores every plaintext byte;
    two items are checked against the oracle."""
    import torch

    from capycrypt_amd import _lib
    from oracle import oracle

    lib = _lib.lib()
    n, L = 32768, 1 << 20
    plain = _rand(n * L, 0xCA9C0013)
    pws, zs = _rand(n * 64, 33), _rand(n * 512, 34)
    outs = []
    try:
        for flags in (0, 1 << 16):
            _lib.check(lib.capy_set_sponge_lanes(flags))
            work = plain.clone()
            tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_sha3_encrypt_batch_dev(512, n, pws.data_ptr(), 64, None, 0, zs.data_ptr(), work.data_ptr(), None, L, L,
                                                       tags.data_ptr(), None))
            torch.cuda.synchronize()
            outs.append((work, tags))
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0])
    work, tags = outs[0]
    del outs
    hp, hz, ht = bytes(pws.cpu().numpy()), bytes(zs.cpu().numpy()), bytes(tags.cpu().numpy())
    for i in (0, n - 1):
        ect, etag = oracle.sha3_encrypt(hp[64 * i:64 * i + 64], hz[512 * i:512 * i + 512], bytes(plain[i * L:(i + 1) * L].cpu().numpy()), 512)
        assert bytes(work[i * L:(i + 1) * L].cpu().numpy()) == ect and ht[64 * i:64 * i + 64] == etag, i
    status = torch.full((n,), 5, dtype=torch.int32, device="cuda")
    _lib.check(lib.capy_sha3_decrypt_batch_dev(512, n, pws.data_ptr(), 64, None, 0, zs.data_ptr(), work.data_ptr(), None, L, L,
                                               tags.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    assert int(status.sum().item()) == 0 and torch.equal(work, plain)


def test_config5_full_batch_sign_then_verify():
    """2^16 x 1 KiB: every signature verifies; after flipping one message byte exactly that item fails; a sample of
    signatures equals the oracle's."""
    import torch

    from capycrypt_amd import _lib
    from oracle import oracle as O

    lib = _lib.lib()
    n, L = 1 << 16, 1024
    msgs, pws = _rand(n * L, 0xCA9C0005), _rand(n * 64, 51)
    pubs = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
    h = torch.zeros(n * 56, dtype=torch.uint8, device="cuda")
    z = torch.zeros(n * 56, dtype=torch.uint8, device="cuda")
    st = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    _lib.check(lib.capy_keypair_batch_dev(512, n, pws.data_ptr(), 64, None, pubs.data_ptr(), None))
    _lib.check(lib.capy_schnorr_sign_batch_dev(512, n, pws.data_ptr(), 64, None, msgs.data_ptr(), None, L, L, h.data_ptr(),
                                               z.data_ptr(), None))
    _lib.check(lib.capy_schnorr_verify_batch_dev(512, n, pubs.data_ptr(), msgs.data_ptr(), None, L, L, h.data_ptr(),
                                                 z.data_ptr(), st.data_ptr(), None))
    torch.cuda.synchronize()
    assert not st.cpu().numpy().any()
    hm, hp = bytes(msgs.cpu().numpy()), bytes(pws.cpu().numpy())
    hh, hz = bytes(h.cpu().numpy()), bytes(z.cpu().numpy())
    for i in (0, 40000, n - 1):
        assert (hh[56 * i:56 * i + 56], hz[56 * i:56 * i + 56]) == O.sign(hp[64 * i:64 * i + 64], hm[L * i:L * i + L], 512)
    bad = 31337
    msgs[bad * L + 500] ^= 1
    _lib.check(lib.capy_schnorr_verify_batch_dev(512, n, pubs.data_ptr(), msgs.data_ptr(), None, L, L, h.data_ptr(),
                                                 z.data_ptr(), st.data_ptr(), None))
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    assert s[bad] == 1 and int(s.sum()) == 1
