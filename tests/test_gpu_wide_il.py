"""The bit-interleaved one-wave-per-sponge kernels (csrc/sponge_wide_il.h, r05): digests with one item per wave, sha3_encrypt /
sha3_decrypt with two cooperating waves per item (keystream wave -> LDS -> tag wave, one barrier per block).

Reference behaviour: /root/reference/src/sha3/sponge.rs:10-95 and shake_functions.rs:24-89 (digests),
/root/reference/src/sha3/encryptable.rs:29-45, :58-83 (encrypt; decrypt with the ciphertext restored on failure).
Every case compares the kernel with another kernel family byte for byte over the WHOLE buffer, with the oracle on sampled
items, and asserts through capy_debug_last_sponge_kernel that these kernels are the ones that ran (kinds 10 and 27).
tests/test_gpu_sponge.py additionally runs its whole suite with these kernels forced ("wave-per-item-kernels")."""
import ctypes as C
import random

import pytest

pytestmark = pytest.mark.gpu

IL_DIGEST, TWO_LANES, IL_CRYPT, FOUR_LANES = 10, 2, 27, 20
FORCE_WIDE = 32 << 8  # debug bit 5: the wave-per-item kernels for every batch of up to 4096 items
NEVER_WIDE = 16 << 8  # debug bit 4


@pytest.fixture(scope="module")
def env():
    import torch

    from capycrypt_amd import _lib
    from oracle import oracle

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    yield _lib, _lib.lib(), oracle, torch
    _lib.lib().capy_set_sponge_lanes(0)


def _rand(env, nbytes, seed):
    _lib, lib, _, torch = env
    t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, None))
    return t


def _last(lib):
    k, l = C.c_int(0), C.c_int(0)
    lib.capy_debug_last_sponge_kernel(C.byref(k), C.byref(l))
    return k.value


def _simds(torch):
    return 4 * torch.cuda.get_device_properties(0).multi_processor_count


def _crypt_case(env, d, n, ln, stride, off, lanes=0, seed=1):
    """encrypt under `lanes` and with the four-lane kernel, compare everything; decrypt with one forged tag"""
    _lib, lib, O, torch = env
    rng = random.Random(seed)
    pws, zs, plain = _rand(env, n * 32, 1 + n), _rand(env, n * 512, 2 + n), _rand(env, n * stride + off + 256, 3 + n + ln)
    optr = None

    def span(i):
        return off + i * stride, off + i * stride + ln

    res = {}
    try:
        for name, mode in (("il", lanes), ("other", NEVER_WIDE)):
            _lib.check(lib.capy_set_sponge_lanes(mode))
            m = plain.clone()
            tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_sha3_encrypt_batch_dev(d, n, pws.data_ptr(), 32, None, n * 32, zs.data_ptr(), m.data_ptr() + off, optr, ln, stride,
                                                      tags.data_ptr(), None))
            torch.cuda.synchronize()
            res[name] = (m, tags, _last(lib))
        assert res["il"][2] == IL_CRYPT and res["other"][2] != IL_CRYPT
        assert torch.equal(res["il"][0], res["other"][0]) and torch.equal(res["il"][1], res["other"][1])
        m, tags, _ = res["il"]
        for i in {0, n - 1, rng.randrange(n)}:
            a, b = span(i)
            want = O.sha3_encrypt(bytes(pws[i * 32:(i + 1) * 32].cpu().numpy()), bytes(zs[i * 512:(i + 1) * 512].cpu().numpy()),
                                  bytes(plain[a:b].cpu().numpy()), d)
            assert (bytes(m[a:b].cpu().numpy()), bytes(tags[64 * i:64 * i + 64].cpu().numpy())) == want
        # decrypt on the same kernel: one forged tag fails alone and keeps its ciphertext
        _lib.check(lib.capy_set_sponge_lanes(lanes))
        status = torch.full((n,), 9, dtype=torch.int32, device="cuda")
        f = rng.randrange(n)
        tags[64 * f + 5] ^= 0x40
        ct = m.clone()
        _lib.check(lib.capy_sha3_decrypt_batch_dev(d, n, pws.data_ptr(), 32, None, n * 32, zs.data_ptr(), m.data_ptr() + off, optr, ln, stride,
                                                  tags.data_ptr(), status.data_ptr(), None))
        torch.cuda.synchronize()
        assert _last(lib) == IL_CRYPT
        want = plain.clone()
        a, b = span(f)
        want[a:b] = ct[a:b]
        assert int(status[f]) == 1 and int((status != 0).sum()) == 1
        assert torch.equal(m, want)
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))


@pytest.mark.parametrize("d,rb", [(512, 136), (256, 168), (384, 152)])
def test_encrypt_decrypt_every_tail_length_class(env, d, rb):
    """message lengths around every boundary the two-wave kernel distinguishes: empty, inside the first 32-bit half of a word,
    on it, inside the second, whole words, the last word of a block, whole blocks, blocks + a ragged tail"""
    for k, ln in enumerate((0, 1, 3, 4, 5, 7, 8, 12, rb - 8, rb - 5, rb - 4, rb - 1, rb, rb + 1, rb + 4, 2 * rb, 2 * rb + 133, 9 * rb + 6)):
        _crypt_case(env, d, (1, 2, 7, 33)[k % 4], ln, (ln + 7) // 8 * 8 + (8, 16, 136)[k % 3], (0, 8, 40)[k % 3], seed=k)


def test_the_two_waves_of_an_item_out_of_step(env):
    """Once every SIMD is busy the two waves of an item stop running in lock step -- the race the first form of this kernel lost
    (it let the tag wave read the message from memory while the keystream wave overwrote it in place): half a wave per SIMD,
    one per SIMD, two per SIMD (the largest automatic batch), and six per SIMD under the debug switch."""
    S = _simds(env[3])
    _crypt_case(env, 512, S // 4 + 44, 64 * 136 + 20, 64 * 136 + 32, 0)
    _crypt_case(env, 256, S // 2, 300 * 168, 300 * 168 + 8, 8)
    _crypt_case(env, 512, S, 40 * 136 + 9, 40 * 136 + 16, 0)
    _crypt_case(env, 512, 3000, 20 * 136 + 9, 20 * 136 + 24, 16, lanes=FORCE_WIDE)


def test_ragged_host_batch(env):
    """A ragged batch reaches the kernel through the host-buffer ABI (device offsets are not inspected for alignment): every
    tail length, per-item lengths, and with 150 items the longest-first processing order; every item against the four-lane
    kernel and the oracle, decrypt with a forged tag through the same path."""
    _lib, lib, O, torch = env
    from capycrypt_amd import ops

    rng = random.Random(3)
    n = 150
    msgs = [rng.randbytes(rng.choice((0, 1, 4, 135, 136, 137, 1000, rng.randrange(0, 5000)))) for _ in range(n)]
    pws = [rng.randbytes(17) for _ in range(n)]
    zs = [rng.randbytes(512) for _ in range(n)]
    res = {}
    try:
        for name, lanes in (("il", 0), ("other", NEVER_WIDE)):
            _lib.check(lib.capy_set_sponge_lanes(lanes))
            res[name] = ops.sha3_encrypt_batch(pws, zs, msgs, 512) + (_last(lib),)
        assert res["il"][2] == IL_CRYPT and res["other"][2] == FOUR_LANES
        assert res["il"][0] == res["other"][0] and res["il"][1] == res["other"][1]
        cts, tags, _ = res["il"]
        for i in range(0, n, 7):
            assert (cts[i], tags[i]) == O.sha3_encrypt(pws[i], zs[i], msgs[i], 512), i
        f = rng.randrange(n)
        tags = list(tags)
        tags[f] = bytes([tags[f][0] ^ 1]) + tags[f][1:]
        _lib.check(lib.capy_set_sponge_lanes(0))
        back, ok = ops.sha3_decrypt_batch(pws, zs, cts, tags, 512)
        assert _last(lib) == IL_CRYPT
        assert [i for i in range(n) if not ok[i]] == [f]
        assert back[f] == cts[f] and all(back[i] == msgs[i] for i in range(n) if i != f)
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))


@pytest.mark.parametrize("d", [224, 256, 384, 512])
def test_sha3_one_item_per_wave_equals_the_other_kernels(env, d):
    _lib, lib, O, torch = env
    try:
        for n, ln in ((1, 0), (1, 135), (3, 136), (5, 1000), (64, 4097), (2, 71), (9, 144 * 50 + 3), (130, 517)):
            stride = (ln + 7) // 8 * 8 + 8
            msgs = _rand(env, n * stride + 64, 40 + n + ln)
            outs = {}
            for name, lanes in (("il", 0), ("two", 2), ("lane", 1)):
                _lib.check(lib.capy_set_sponge_lanes(lanes))
                out = torch.zeros(n * (d // 8) + 8, dtype=torch.uint8, device="cuda")
                _lib.check(lib.capy_sha3_batch_dev(d, n, msgs.data_ptr(), None, ln, stride, out.data_ptr(), None))
                torch.cuda.synchronize()
                outs[name] = (out, _last(lib))
            assert outs["il"][1] == IL_DIGEST and outs["two"][1] == TWO_LANES
            assert torch.equal(outs["il"][0], outs["two"][0]) and torch.equal(outs["il"][0], outs["lane"][0])
            assert int(outs["il"][0][n * (d // 8):].sum()) == 0
            for i in (0, n - 1):  # the oracle, not hashlib: the reference's pad rule differs from FIPS 202 at some lengths (sponge.rs:23-33)
                assert bytes(outs["il"][0][i * (d // 8):(i + 1) * (d // 8)].cpu().numpy()) == O.sha3(bytes(msgs[i * stride:i * stride + ln].cpu().numpy()), d)
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))


def test_kmac_xof_long_squeezes_and_unaligned_outputs(env):
    """squeezes of several blocks, output lengths that end inside a 32-bit half, output rows that are only byte aligned"""
    _lib, lib, O, torch = env
    try:
        for d in (256, 512):
            for n, ln, ol in ((1, 0, 32), (2, 100, 64), (3, 1000, 1000), (50, 136, 171), (1, 5000, 4096), (7, 9, 3), (5, 300, 137)):
                stride = (ln + 7) // 8 * 8 + 8
                msgs, keys = _rand(env, n * stride + 64, 60 + n + ln), _rand(env, n * 32, 61 + n)
                outs = {}
                for name, lanes in (("il", 0), ("two", 2), ("lane", 1)):
                    _lib.check(lib.capy_set_sponge_lanes(lanes))
                    out = torch.zeros(n * ol + 16, dtype=torch.uint8, device="cuda")
                    _lib.check(lib.capy_kmac_xof_batch_dev(d, n, keys.data_ptr(), 32, 32, None, msgs.data_ptr(), None, ln, stride, 8 * ol, b"T", 1,
                                                          out.data_ptr() + 1, ol, None))
                    torch.cuda.synchronize()
                    outs[name] = (out, _last(lib))
                assert outs["il"][1] == IL_DIGEST
                assert torch.equal(outs["il"][0], outs["two"][0]) and torch.equal(outs["il"][0], outs["lane"][0])
                assert int(outs["il"][0][0]) == 0 and int(outs["il"][0][1 + n * ol:].sum()) == 0
                i = n - 1
                want = O.kmac_xof(bytes(keys[i * 32:(i + 1) * 32].cpu().numpy()), bytes(msgs[i * stride:i * stride + ln].cpu().numpy()), 8 * ol, b"T", d)
                assert bytes(outs["il"][0][1 + i * ol:1 + (i + 1) * ol].cpu().numpy()) == want
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))


def test_launch_plan_reports_the_kernel(env):
    """capy_sha3_launch_plan: one item per wave up to two items per SIMD, then two lanes per sponge"""
    _lib, lib, _, torch = env
    S = _simds(torch)
    kind, phases = C.c_int(0), C.c_int(0)
    for n, want in ((1, 10), (S, 10), (S + 1, 10), (2 * S, 10), (2 * S + 1, 2)):
        _lib.check(lib.capy_sha3_launch_plan(256, n, 1 << 20, (1 << 20) + 8, C.byref(kind), C.byref(phases)))
        assert (kind.value, phases.value) == (want, 1), (n, kind.value)
