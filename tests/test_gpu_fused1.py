"""The one-lane-per-sponge fused sha3_encrypt / sha3_decrypt kernel (csrc/sponge_fused1.h, r05): two lanes per item, whole-line
stores through an LDS ring, three schedules (single launch, time slices of two waves per SIMD, rotating occupancy).

Reference behaviour: /root/reference/src/sha3/encryptable.rs:29-45 (encrypt), :58-83 (decrypt, ciphertext restored on failure).
Every case compares the new kernel with the two-pass form (tag kernel + keystream kernel: capy_set_sponge_lanes bit 16) byte for
byte over the WHOLE buffer (so the bytes between the messages are covered too), with the oracle on sampled items, and asserts
through capy_debug_last_sponge_kernel that the schedule the case means to cover is the one that ran."""
import ctypes as C
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONE_LANE, ONE_LANE_SLICED, ONE_LANE_ROT, TWO_PASS = 23, 24, 25, 26


@pytest.fixture(scope="module")
def env():
    import torch

    from capycrypt_amd import _lib
    from oracle import oracle

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return _lib, _lib.lib(), oracle, torch


def _rand(lib, _lib, torch, nbytes, seed):
    t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, None))
    return t


def _last(lib):
    k, l = C.c_int(0), C.c_int(0)
    lib.capy_debug_last_sponge_kernel(C.byref(k), C.byref(l))
    return k.value, l.value


def _simds(torch):
    return 4 * torch.cuda.get_device_properties(0).multi_processor_count


def _round_trip(env, d, n, ln, stride, off, want_kind, offsets=None, pl=32, samples=3, seed=1):
    """encrypt with the automatic kernel choice and with the two-pass form, compare everything, decrypt with one forged tag"""
    _lib, lib, O, torch = env
    rng = random.Random(seed)
    total = (offsets[-1] if offsets is not None else n * stride) + off + 256
    pws, zs, plain = _rand(lib, _lib, torch, n * pl, 1 + n), _rand(lib, _lib, torch, n * 512, 2 + n), _rand(lib, _lib, torch, total, 3 + n + ln)
    offs_dev = torch.tensor(offsets, dtype=torch.int64, device="cuda") if offsets is not None else None
    optr = offs_dev.data_ptr() if offs_dev is not None else None

    def start(i):
        return off + (offsets[i] if offsets is not None else i * stride)

    def length(i):
        return offsets[i + 1] - offsets[i] if offsets is not None else ln

    res = {}
    try:
        for name, lanes in (("auto", 0), ("two-pass", 1 | (1 << 16))):
            _lib.check(lib.capy_set_sponge_lanes(lanes))
            m = plain.clone()
            tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_sha3_encrypt_batch_dev(d, n, pws.data_ptr(), pl, None, n * pl, zs.data_ptr(), m.data_ptr() + off, optr, ln, stride,
                                                      tags.data_ptr(), None))
            torch.cuda.synchronize()
            res[name] = (m, tags, _last(lib))
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))
    kind, launches = res["auto"][2]
    assert kind == want_kind, (kind, launches, want_kind)
    assert res["two-pass"][2][0] == TWO_PASS
    if want_kind in (ONE_LANE_SLICED, ONE_LANE_ROT):
        assert launches > 1
    assert torch.equal(res["auto"][1], res["two-pass"][1]), "tags differ from the two-pass form"
    assert torch.equal(res["auto"][0], res["two-pass"][0]), "ciphertexts (or bytes outside the messages) differ from the two-pass form"
    m, tags, _ = res["auto"]
    for i in sorted({0, n - 1, 31, 32, n // 2} | {rng.randrange(n) for _ in range(samples)}):
        if i >= n:
            continue
        a, b = start(i), start(i) + length(i)
        want = O.sha3_encrypt(bytes(pws[i * pl:(i + 1) * pl].cpu().numpy()), bytes(zs[i * 512:(i + 1) * 512].cpu().numpy()),
                              bytes(plain[a:b].cpu().numpy()), d)
        assert (bytes(m[a:b].cpu().numpy()), bytes(tags[64 * i:64 * i + 64].cpu().numpy())) == want, (d, n, i)
    # decrypt: item f's tag forged -> it alone fails and keeps its ciphertext (encryptable.rs:77-82)
    f = rng.randrange(n)
    tags[64 * f + 5] ^= 0x40
    ct = m[start(f):start(f) + length(f)].clone()
    status = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    _lib.check(lib.capy_sha3_decrypt_batch_dev(d, n, pws.data_ptr(), pl, None, n * pl, zs.data_ptr(), m.data_ptr() + off, optr, ln, stride,
                                              tags.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    assert _last(lib)[0] == want_kind
    want = plain.clone()
    want[start(f):start(f) + length(f)] = ct
    assert int(status[f]) == 1 and int((status != 0).sum()) == 1
    assert torch.equal(m, want)


@pytest.mark.parametrize("form", [1, 2, 4, "1,fused1_lone_direct=1"])
def test_every_instance_on_small_batches_in_a_child_process(form):
    """tools/check_fused1.py with CAPY_DEBUG=fused1_min=1,fused1_form=F (the knobs are read once per process): 150 shapes --
    three rates, batch sizes 1 / 31 / 33 / 100 (partial waves), lengths from empty to 40 blocks with and without tails, strides
    and starting offsets that put the messages at every 8-byte position of their 128-byte lines -- each against the two-pass
    form over the whole buffer, the oracle, and a decrypt with one forged tag."""
    env = dict(os.environ, CAPY_DEBUG="fused1_min=1,fused1_form=%s" % form, CASES="150")  # (form 1 with and without per-lane stores)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_fused1.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "bad 0" in r.stdout


def test_single_launch_two_waves_per_simd(env):
    """n = 64 items per SIMD exactly: one launch of the unrolled instance, 1 KiB-class messages with a tail."""
    S = _simds(env[3])
    _round_trip(env, 512, 64 * S, 136 * 9 + 50, 136 * 9 + 56, 8, ONE_LANE)


def test_lone_wave_form_takes_uniform_batches_from_24_items_per_simd(env):
    """r06 (the 10 % step at n = 32 S).  Up to 32 items per SIMD the one-lane form is ONE lone wave per SIMD
    (FORM 1: compiled so that a second wave does not fit, per-lane stores) and beats the four-lane form from 24 items per SIMD on
    (profiles/r06_fused_32s_ab.txt), so UNIFORM batches switch there: 24 S and 32 S take kind 23 in one launch, 24 S - 1 still the
    four-lane kernel (kind 20).  Each against the two-pass form over the whole buffer, the oracle on samples, and a decrypt with a
    forged tag (_round_trip); a partial last wave and a tail at 32 S - 13."""
    S = _simds(env[3])
    _round_trip(env, 512, 24 * S, 136 * 9 + 50, 136 * 9 + 56, 8, ONE_LANE)
    _round_trip(env, 512, 32 * S, 136 * 7 + 3, 136 * 7 + 8, 0, ONE_LANE)
    _round_trip(env, 256, 32 * S - 13, 168 * 5 + 161, 168 * 6, 8, ONE_LANE)
    _round_trip(env, 512, 24 * S - 1, 136 * 9 + 50, 136 * 9 + 56, 8, 20)


def test_ragged_batches_keep_the_four_lane_form_up_to_32_items_per_simd(env):
    """... and RAGGED batches do not: half as many items per wave wait for the wave's longest message in the four-lane form.  A
    host batch of 30 S items of 0 .. 5 blocks: kind 20, equal to the two-pass form and the oracle."""
    _lib, lib, O, torch = env
    from capycrypt_amd import ops

    S = _simds(torch)
    n = 30 * S
    rng = random.Random(17)
    msgs = [rng.randbytes(rng.randrange(0, 5 * 136 + 9)) for _ in range(n)]
    pws = [rng.randbytes(16) for _ in range(n)]
    zs = [rng.randbytes(512) for _ in range(n)]
    res = {}
    try:
        for name, lanes in (("auto", 0), ("two-pass", 1 | (1 << 16))):
            _lib.check(lib.capy_set_sponge_lanes(lanes))
            res[name] = ops.sha3_encrypt_batch(pws, zs, msgs, 512) + (_last(lib)[0],)
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))
    assert res["auto"][2] == 20 and res["two-pass"][2] == TWO_PASS, (res["auto"][2], res["two-pass"][2])
    assert res["auto"][:2] == res["two-pass"][:2]
    for i in (0, n - 1, rng.randrange(n)):
        assert (res["auto"][0][i], res["auto"][1][i]) == O.sha3_encrypt(pws[i], zs[i], msgs[i], 512), i


def test_single_launch_four_waves_per_simd_short_messages(env):
    """more than two waves per SIMD, messages too short for slices: the rolled 128-register instance, D256 (two-part filing)"""
    S = _simds(env[3])
    _round_trip(env, 256, 100 * S + 17, 168 * 5, 168 * 5 + 8, 0, ONE_LANE)


def test_ragged_host_batch_single_launch(env):
    """A chip-filling RAGGED batch reaches the kernel through the host-buffer ABI only (device offsets are not inspected for
    alignment): messages of 0 .. 6 blocks with every tail length, re-packed to aligned starts with per-item lengths and a
    longest-first processing order on upload.  One launch; waves run to their longest item.  Against the two-pass form for
    every item and the oracle for a sample; decrypt with a forged tag through the same path."""
    _lib, lib, O, torch = env
    from capycrypt_amd import ops

    S = _simds(torch)
    n = 36 * S + 5
    rng = random.Random(7)
    msgs = [rng.randbytes(rng.randrange(0, 6 * 152 + 9)) for _ in range(n)]
    pws = [rng.randbytes(24) for _ in range(n)]
    zs = [rng.randbytes(512) for _ in range(n)]
    res = {}
    try:
        for name, lanes in (("auto", 0), ("two-pass", 1 | (1 << 16))):
            _lib.check(lib.capy_set_sponge_lanes(lanes))
            res[name] = ops.sha3_encrypt_batch(pws, zs, msgs, 384) + (_last(lib)[0],)
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))
    assert res["auto"][2] == ONE_LANE and res["two-pass"][2] == TWO_PASS
    assert res["auto"][0] == res["two-pass"][0] and res["auto"][1] == res["two-pass"][1]
    cts, tags, _ = res["auto"]
    for i in (0, 1, n - 1, rng.randrange(n), rng.randrange(n)):
        assert (cts[i], tags[i]) == O.sha3_encrypt(pws[i], zs[i], msgs[i], 384), i
    f = rng.randrange(n)
    tags = list(tags)
    tags[f] = bytes([tags[f][0] ^ 1]) + tags[f][1:]
    back, ok = ops.sha3_decrypt_batch(pws, zs, cts, tags, 384)
    assert _last(lib)[0] == ONE_LANE
    assert [i for i in range(n) if not ok[i]] == [f]
    assert back[f] == cts[f] and all(back[i] == msgs[i] for i in range(n) if i != f)


def test_rotating_occupancy_schedule_between_one_and_two_waves(env):
    """32 < n / SIMDs < 64 with long messages: phase launches of sponge_fused1_rot_kernel + one resume launch; a batch size that is
    not a multiple of 32 or 128, a length with a tail, messages at 8 mod 16"""
    S = _simds(env[3])
    _round_trip(env, 512, 33 * S + 77, 136 * 600 + 77, 136 * 600 + 80 + 8, 8, ONE_LANE_ROT)


def test_rotating_occupancy_schedule_d256(env):
    S = _simds(env[3])
    _round_trip(env, 256, 52 * S, 168 * 520, 168 * 520 + 128, 0, ONE_LANE_ROT)


def test_time_slices_above_two_waves_per_simd(env):
    """n just above 64 items per SIMD, long messages: slices of exactly two waves per SIMD, wave-groups taking turns"""
    S = _simds(env[3])
    _round_trip(env, 512, 64 * S + 500, 136 * 515 + 16, 136 * 515 + 16 + 24, 0, ONE_LANE_SLICED)


def test_kem_and_ecdhies_keys_take_the_same_kernel(env):
    """the other callers of the symmetric half: KEM (32-byte secrets, tags KEMKE / KEMKA, src/kem/encryptable.rs:47-59) through
    its device entry point at a chip-filling size; result against the two-pass form"""
    _lib, lib, O, torch = env
    S = _simds(torch)
    n, ln = 36 * S, 136 * 4 + 9
    stride = (ln + 7) // 8 * 8
    secrets, zs, plain = _rand(lib, _lib, torch, n * 32, 5), _rand(lib, _lib, torch, n * 512, 6), _rand(lib, _lib, torch, n * stride, 7)
    res = {}
    try:
        for name, lanes in (("auto", 0), ("two-pass", 1 | (1 << 16))):
            _lib.check(lib.capy_set_sponge_lanes(lanes))
            m = plain.clone()
            tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_kem_sponge_encrypt_batch_dev(512, n, secrets.data_ptr(), 32, zs.data_ptr(), m.data_ptr(), None, ln, stride, tags.data_ptr(), None))
            torch.cuda.synchronize()
            res[name] = (m, tags, _last(lib)[0])
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))
    assert res["auto"][2] == ONE_LANE and res["two-pass"][2] == TWO_PASS
    assert torch.equal(res["auto"][0], res["two-pass"][0]) and torch.equal(res["auto"][1], res["two-pass"][1])


def test_ecdhies_symmetric_half_takes_the_kernel_with_56_byte_keys_and_tags(env):
    """KeyEncryptable (src/ecc/encryptable.rs:34-94) at a chip-filling size: its symmetric half hands the kernel 56-byte ke / ka
    at a 112-byte stride and asks for 56-byte tags (the sponge callers: 64 / 128 / 64) -- heads, tag stores and the D256 rate
    differ.  key_encrypt against the two-pass form for every byte, two items against the oracle, then key_decrypt with one
    forged tag."""
    _lib, lib, O, torch = env
    S = _simds(torch)
    n, ln, d = 33 * S + 9, 168 * 3 + 21, 256
    stride = (ln + 7) // 8 * 8 + 8
    rng = random.Random(0xEC)
    pw_len = 24
    pws, ks, plain = _rand(lib, _lib, torch, n * pw_len, 41), _rand(lib, _lib, torch, n * 56, 42), _rand(lib, _lib, torch, n * stride, 43)
    pubs = torch.empty(n * 112, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_keypair_batch_dev(d, n, pws.data_ptr(), pw_len, None, pubs.data_ptr(), None))
    res = {}
    try:
        for name, lanes in (("auto", 0), ("two-pass", 1 | (1 << 16))):
            _lib.check(lib.capy_set_sponge_lanes(lanes))
            m = plain.clone()
            zxy = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
            tags = torch.zeros(n * 56, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_key_encrypt_batch_dev(d, n, pubs.data_ptr(), ks.data_ptr(), m.data_ptr(), None, ln, stride, zxy.data_ptr(), tags.data_ptr(), None))
            torch.cuda.synchronize()
            res[name] = (m, zxy, tags, _last(lib)[0])
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))
    assert res["auto"][3] == ONE_LANE and res["two-pass"][3] == TWO_PASS
    for k in range(3):
        assert torch.equal(res["auto"][k], res["two-pass"][k]), k
    m, zxy, tags, _ = res["auto"]
    for i in (0, n - 1):
        want = O.key_encrypt(bytes(pubs[112 * i:112 * i + 112].cpu().numpy()), bytes(ks[56 * i:56 * i + 56].cpu().numpy()),
                             bytes(plain[i * stride:i * stride + ln].cpu().numpy()), d)
        got = (bytes(m[i * stride:i * stride + ln].cpu().numpy()), bytes(zxy[112 * i:112 * i + 112].cpu().numpy()), bytes(tags[56 * i:56 * i + 56].cpu().numpy()))
        assert got == want, i
    f = rng.randrange(n)
    tags[56 * f] ^= 0x80
    ct = m[f * stride:f * stride + ln].clone()
    status = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    _lib.check(lib.capy_key_decrypt_batch_dev(d, n, pws.data_ptr(), pw_len, None, zxy.data_ptr(), m.data_ptr(), None, ln, stride, tags.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    assert _last(lib)[0] == ONE_LANE
    want = plain.clone()
    want[f * stride:f * stride + ln] = ct
    assert int(status[f]) == 1 and int((status != 0).sum()) == 1 and torch.equal(m, want)
