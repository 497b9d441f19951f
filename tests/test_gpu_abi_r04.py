"""GPU tests of the C-ABI additions of round 4 (include/capyhip.h): per-call options (capy_call_options, *_ex entry
points) against the process-wide setters, generator handles, the device-buffer forms that were missing (KEM sponge half,
point addition, verify-shaped double multiplication), the output-length range check, and the edge scalars of the
twisted-curve fixed-base kernels.  Parity is against the oracle and against the existing entry points."""
import ctypes as C
import random
import threading

import pytest

pytestmark = pytest.mark.gpu

R = (1 << 446) - 0x8335dc163bb124b65129c96fde933d8d723a70aadc873d6d54a7bb0d


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


def _last_kernels(lib):
    vb, fb = C.c_int(0), C.c_int(0)
    lib.capy_debug_last_curve_kernel(C.byref(vb), C.byref(fb))
    return vb.value & 15, fb.value & 15  # 1 indexed, 2 constant-address (bit 4: the wave-per-item family)


def test_hardened_values_and_kernel_choice_of_raw_calls(capy, O):
    """CAPY_HARDEN_ALL (1, the r02 meaning) covers the raw scalarmul / basemul calls, CAPY_HARDEN_PROTOCOL (4,
    the default) only the secret scalars of the protocol calls, r03's 2 and 3 are refused -- asserted on the kernel
    family each call really launched (capy_debug_last_curve_kernel), with identical results in every mode."""
    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(0xAB1)
    n = 70
    ks = [rng.randbytes(56) for _ in range(n)]
    pts = [O.ed448_basemul(rng.randbytes(56)) for _ in range(n)]
    pws = [rng.randbytes(20) for _ in range(n)]
    for bad in (2, 3, 5, -1):
        assert lib.capy_ed448_set_hardened(bad) == _lib.CAPY_ERR_ARG
    want = {capy.ops.HARDEN_OFF: ((1, 1), 1), capy.ops.HARDEN_PROTOCOL: ((1, 1), 2), capy.ops.HARDEN_ALL: ((2, 2), 2)}
    res = {}
    try:
        for mode, (raw, proto) in want.items():
            capy.ops.ed448_set_hardened(mode)
            vb = capy.ops.ed448_scalarmul_batch(ks, pts)
            fb = capy.ops.ed448_basemul_batch(ks)
            assert _last_kernels(lib) == raw, mode
            pub = capy.ops.keypair_batch(pws, 256)
            assert _last_kernels(lib)[1] == proto, mode
            res[mode] = (vb, fb, pub)
    finally:
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_PROTOCOL)
    assert res[capy.ops.HARDEN_OFF] == res[capy.ops.HARDEN_PROTOCOL] == res[capy.ops.HARDEN_ALL]
    assert res[capy.ops.HARDEN_OFF][1][:5] == [O.ed448_basemul(k) for k in ks[:5]]


def test_call_options_let_two_threads_differ(capy, O):
    """capy_call_options through the *_ex entry points: two host threads run the same calls at the same time, one with
    indexed lookups and one with constant-address lookups for everything, while the process default stays in force for
    calls without options.  Each thread sees its own kernel family; all results are equal."""
    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(0xAB2)
    n = 96
    ks = b"".join(rng.randbytes(56) for _ in range(n))
    pws = b"".join(rng.randbytes(32) for _ in range(n))
    msgs = [rng.randbytes(100) for _ in range(n)]
    mbuf, moff = _lib.pack(msgs)
    out = {}
    errs = []

    def worker(name, hardened):
        try:
            opt = _lib.CallOptions(hardened=hardened)
            fams = set()
            for _ in range(4):
                fb = (C.c_uint8 * (n * 112))()
                _lib.check(lib.capy_ed448_basemul_batch_ex(n, ks, fb, C.byref(opt)))
                fams.add(_last_kernels(lib)[1])
                pub = (C.c_uint8 * (n * 112))()
                _lib.check(lib.capy_keypair_batch_ex(512, n, pws, 32, None, pub, C.byref(opt)))
                fams.add(_last_kernels(lib)[1])
                h, z = (C.c_uint8 * (n * 56))(), (C.c_uint8 * (n * 56))()
                _lib.check(lib.capy_schnorr_sign_batch_ex(512, n, pws, 32, None, mbuf, moff, h, z, C.byref(opt)))
                st = (C.c_int32 * n)()
                _lib.check(lib.capy_schnorr_verify_batch_ex(512, n, pub, mbuf, moff, h, z, st, C.byref(opt)))
                assert not any(st)
            out[name] = (bytes(fb), bytes(pub), bytes(h), bytes(z), fams)
        except Exception as e:  # noqa: BLE001
            errs.append((name, repr(e)))

    ts = [threading.Thread(target=worker, args=("off", _lib.CAPY_HARDEN_OFF)),
          threading.Thread(target=worker, args=("all", _lib.CAPY_HARDEN_ALL))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert out["off"][4] == {1} and out["all"][4] == {2}
    assert out["off"][:4] == out["all"][:4]
    # a call without options still follows the process default (raw call: indexed; key pair: constant-address)
    fb = (C.c_uint8 * (n * 112))()
    _lib.check(lib.capy_ed448_basemul_batch(n, ks, fb))
    assert _last_kernels(lib)[1] == 1 and bytes(fb) == out["off"][0]
    # malformed options are refused
    bad = _lib.CallOptions(hardened=3)
    assert lib.capy_ed448_basemul_batch_ex(n, ks, fb, C.byref(bad)) == _lib.CAPY_ERR_ARG
    short = _lib.CallOptions()
    short.struct_size = 8
    assert lib.capy_ed448_basemul_batch_ex(n, ks, fb, C.byref(short)) == _lib.CAPY_ERR_ARG
    unknown = _lib.CallOptions(generator=63)
    assert lib.capy_ed448_basemul_batch_ex(n, ks, fb, C.byref(unknown)) == _lib.CAPY_ERR_ARG


def test_generator_handles(capy, O):
    """capy_ed448_generator_create + capy_call_options::generator: a second generator G' = [5]G serves fixed-base
    multiplications, key pairs, signatures and verification of the calls that select it (indexed and constant-address
    tables, lane and wave kernels), next to the process generator, which is untouched."""
    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(0xAB3)
    g2 = O.ed448_basemul((5).to_bytes(56, "big"))
    handle = C.c_int(-1)
    _lib.check(lib.capy_ed448_generator_create(g2, C.byref(handle)))
    again = C.c_int(-1)
    _lib.check(lib.capy_ed448_generator_create(g2, C.byref(again)))
    assert handle.value >= 1 and again.value == handle.value
    # a point outside the prime-order subgroup is refused
    p = (1 << 448) - (1 << 224) - 1
    order2 = (0).to_bytes(56, "little") + (p - 1).to_bytes(56, "little")
    assert lib.capy_ed448_generator_create(order2, C.byref(again)) == _lib.CAPY_ERR_ARG
    for n in (40, 9000):  # wave-per-item and lane-per-item kernels
        ks = [rng.randbytes(56) for _ in range(n)]
        kb = b"".join(ks)
        for hardened in (_lib.CAPY_HARDEN_OFF, _lib.CAPY_HARDEN_ALL):
            opt = _lib.CallOptions(hardened=hardened, generator=handle.value)
            fb2 = (C.c_uint8 * (n * 112))()
            _lib.check(lib.capy_ed448_basemul_batch_ex(n, kb, fb2, C.byref(opt)))
            want = capy.ops.ed448_scalarmul_batch(ks, [g2] * n)
            assert [bytes(fb2[112 * i:112 * i + 112]) for i in range(n)] == want, (n, hardened)
        fb1 = capy.ops.ed448_basemul_batch(ks[:8])
        assert fb1 == [O.ed448_basemul(k) for k in ks[:8]]
    # a signature made with G' verifies with G' and fails with G
    n = 33
    pws = b"".join(rng.randbytes(24) for _ in range(n))
    msgs = [rng.randbytes(rng.randrange(0, 300)) for _ in range(n)]
    mbuf, moff = _lib.pack(msgs)
    opt = _lib.CallOptions(generator=handle.value)
    pub, h, z = (C.c_uint8 * (n * 112))(), (C.c_uint8 * (n * 56))(), (C.c_uint8 * (n * 56))()
    _lib.check(lib.capy_keypair_batch_ex(512, n, pws, 24, None, pub, C.byref(opt)))
    _lib.check(lib.capy_schnorr_sign_batch_ex(512, n, pws, 24, None, mbuf, moff, h, z, C.byref(opt)))
    st = (C.c_int32 * n)()
    _lib.check(lib.capy_schnorr_verify_batch_ex(512, n, pub, mbuf, moff, h, z, st, C.byref(opt)))
    assert not any(st)
    _lib.check(lib.capy_schnorr_verify_batch(512, n, pub, mbuf, moff, h, z, st))
    assert all(st)


def test_new_device_buffer_forms(capy, O):
    """capy_ed448_add_batch_dev, capy_ed448_double_scalarmul_batch_dev, capy_kem_sponge_{encrypt,decrypt}_batch_dev (r04)
    against their host-buffer forms and the oracle."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(0xAB4)

    def dev(b):
        return torch.tensor(list(b), dtype=torch.uint8, device="cuda") if b else torch.zeros(8, dtype=torch.uint8, device="cuda")

    for n in (50, 9000):
        a = [rng.randbytes(56) for _ in range(n)]
        b = [rng.randbytes(56) for _ in range(n)]
        pts = capy.ops.ed448_basemul_batch([rng.randbytes(56) for _ in range(n)])
        qts = capy.ops.ed448_basemul_batch([rng.randbytes(56) for _ in range(n)])
        dp, dq, da, db = dev(b"".join(pts)), dev(b"".join(qts)), dev(b"".join(a)), dev(b"".join(b))
        o1 = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
        o2 = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_ed448_add_batch_dev(n, dp.data_ptr(), dq.data_ptr(), o1.data_ptr(), None))
        _lib.check(lib.capy_ed448_double_scalarmul_batch_dev(n, da.data_ptr(), db.data_ptr(), dp.data_ptr(), o2.data_ptr(), None))
        torch.cuda.synchronize()
        h1, h2 = (C.c_uint8 * (n * 112))(), (C.c_uint8 * (n * 112))()
        _lib.check(lib.capy_ed448_add_batch(n, b"".join(pts), b"".join(qts), h1))
        _lib.check(lib.capy_ed448_double_scalarmul_batch(n, b"".join(a), b"".join(b), b"".join(pts), h2))
        assert bytes(o1.cpu().numpy()) == bytes(h1) and bytes(o2.cpu().numpy()) == bytes(h2)
        for i in (0, n // 2, n - 1):
            assert bytes(h1[112 * i:112 * i + 112]) == O.ed448_add(pts[i], qts[i])
    assert lib.capy_ed448_add_batch_dev(4, None, None, None, None) == _lib.CAPY_ERR_ARG
    # KEM sponge half on device buffers: uniform messages, then ragged ones through offsets
    n, L, stride = 300, 1000, 1008
    secrets, zs = rng.randbytes(n * 32), rng.randbytes(n * 512)
    raw = rng.randbytes(n * stride)
    dsec, dz = dev(secrets), dev(zs)
    work = dev(raw)
    tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
    _lib.check(lib.capy_kem_sponge_encrypt_batch_dev(512, n, dsec.data_ptr(), 32, dz.data_ptr(), work.data_ptr(), None, L, stride,
                                                     tags.data_ptr(), None))
    torch.cuda.synchronize()
    msgs = [raw[i * stride:i * stride + L] for i in range(n)]
    hbuf, hoff = _lib.pack(msgs)
    htags = (C.c_uint8 * (n * 64))()
    _lib.check(lib.capy_kem_sponge_encrypt_batch(512, n, secrets, 32, zs, hbuf, hoff, htags))
    hw = bytes(work.cpu().numpy())
    assert b"".join(hw[i * stride:i * stride + L] for i in range(n)) == bytes(hbuf)[:n * L]
    assert bytes(tags.cpu().numpy()) == bytes(htags)
    assert all(hw[i * stride + L:(i + 1) * stride] == raw[i * stride + L:(i + 1) * stride] for i in range(n))
    status = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    tags[64 * 7] ^= 1
    _lib.check(lib.capy_kem_sponge_decrypt_batch_dev(512, n, dsec.data_ptr(), 32, dz.data_ptr(), work.data_ptr(), None, L, stride,
                                                     tags.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert st[7] == 1 and int(st.sum()) == 1
    back = bytes(work.cpu().numpy())
    for i in range(n):
        want = hw if i == 7 else raw
        assert back[i * stride:i * stride + L] == want[i * stride:i * stride + L], i


def test_output_length_is_range_checked(capy):
    """l_bits / 8 used to be narrowed to 32 bits without a check (the reference takes l: usize,
    src/sha3/shake_functions.rs:79); an output of 2^32 bytes or more per item is now refused, not truncated."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    key, out = torch.zeros(64, dtype=torch.uint8, device="cuda"), torch.zeros(64, dtype=torch.uint8, device="cuda")
    big = 1 << 35
    assert lib.capy_kmac_xof_batch_dev(512, 1, key.data_ptr(), 64, 64, None, None, None, 0, 0, big, b"S", 1, out.data_ptr(),
                                       1 << 32, None) == _lib.CAPY_ERR_ARG
    assert lib.capy_cshake_batch_dev(512, 1, key.data_ptr(), None, 8, 8, big, b"N", 1, b"S", 1, out.data_ptr(), 1 << 32,
                                     None) == _lib.CAPY_ERR_ARG
    hk, ho = (C.c_uint8 * 64)(), (C.c_uint8 * 64)()
    off = (C.c_uint64 * 2)(0, 0)
    assert lib.capy_kmac_xof_batch(512, 1, hk, 64, None, hk, off, big, b"S", 1, ho) == _lib.CAPY_ERR_ARG
    assert lib.capy_cshake_batch(512, 1, hk, off, big, b"N", 1, b"S", 1, ho) == _lib.CAPY_ERR_ARG
    # the largest length below the limit is accepted as far as the arguments go (a tiny out_stride then fails the next check)
    assert lib.capy_kmac_xof_batch_dev(512, 1, key.data_ptr(), 64, 64, None, None, None, 0, 0, big - 8, b"S", 1, out.data_ptr(),
                                       64, None) == _lib.CAPY_ERR_ARG
    assert b"out_stride" in lib.capy_last_error()


def test_twisted_fixed_base_kernels_on_edge_scalars(capy, O):
    """the fixed base accumulates on the twisted curve E' (a = -1, non-square; d' = -39082 is a square), whose
    7M additions are complete only on the odd-order subgroup the tables live in.  Scalars k = 0, 1, r - 1, r, r + 1, 2r,
    2^448 - 1 and their neighbours -- where the accumulator passes through or ends at the identity -- must come out right
    from every kernel that works on E': fb_kernel<true> and fb_ct7_kernel<true> (one item per lane) and, from 262 144
    items, fb2_kernel<false, true> and fb_ct7_pair_kernel<true> (two items per lane, ragged tail)."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    edge = [0, 1, 2, R - 2, R - 1, R, R + 1, 2 * R - 1, 2 * R, 2 * R + 1, 3 * R, 4 * R - 1, (1 << 448) - 1, (1 << 448) - 2, 1 << 447,
            (1 << 446), 4 * R, 4 * R + 3]
    edge = [k % (1 << 448) for k in edge]
    want = {k: O.ed448_basemul(k.to_bytes(56, "big")) for k in edge}
    ident = (0).to_bytes(56, "little") + (1).to_bytes(56, "little")
    assert want[0] == ident and want[R] == ident and want[2 * R] == ident
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        _lib.check(lib.capy_ed448_set_wave_max(0))  # lane-per-item kernels at every size
        for n in (len(edge), 9000, (1 << 18) + 37):
            sc = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_fill_random_dev(sc.data_ptr(), n * 56, 4242, sp))
            flat = sc.view(n, 56)
            eb = torch.tensor(list(b"".join(k.to_bytes(56, "big") for k in edge)), dtype=torch.uint8, device="cuda").view(-1, 56)
            flat[:len(edge)] = eb            # first lanes of the first wave
            flat[n - len(edge):] = eb        # the ragged tail
            outs = []
            for mode in (capy.ops.HARDEN_OFF, capy.ops.HARDEN_ALL):
                capy.ops.ed448_set_hardened(mode)
                o = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
                _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), o.data_ptr(), sp))
                torch.cuda.synchronize()
                outs.append(o)
            assert torch.equal(outs[0], outs[1]), n
            ho = bytes(outs[0].cpu().numpy())
            for j, k in enumerate(edge):
                assert ho[112 * j:112 * j + 112] == want[k], (n, hex(k))
                t = n - len(edge) + j
                assert ho[112 * t:112 * t + 112] == want[k], (n, "tail", hex(k))
    finally:
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_PROTOCOL)
        _lib.check(lib.capy_ed448_set_wave_max(-1))


def test_python_mirror_takes_call_options(capy, O):
    """capycrypt_amd.ops: options=CallOptions(...) routes a helper through the *_ex entry point."""
    rng = random.Random(0xAB5)
    n = 20
    pws = [rng.randbytes(rng.randrange(0, 40)) for _ in range(n)]
    msgs = [rng.randbytes(rng.randrange(0, 200)) for _ in range(n)]
    off = capy.ops.CallOptions(hardened=capy.ops.HARDEN_OFF)
    hard = capy.ops.CallOptions(hardened=capy.ops.HARDEN_ALL)
    pub = capy.ops.keypair_batch(pws, 384, options=off)
    assert pub == capy.ops.keypair_batch(pws, 384, options=hard) == capy.ops.keypair_batch(pws, 384)
    assert pub[:3] == [O.keypair_pub(pw, 384) for pw in pws[:3]]
    sigs = capy.ops.schnorr_sign_batch(pws, msgs, 384, options=hard)
    assert sigs == capy.ops.schnorr_sign_batch(pws, msgs, 384, options=off)
    assert all(capy.ops.schnorr_verify_batch(pub, msgs, sigs, 384, options=off))
    g2 = O.ed448_basemul((7).to_bytes(56, "big"))
    h = capy.ops.ed448_generator_create(g2)
    ks = [rng.randbytes(56) for _ in range(n)]
    assert capy.ops.ed448_basemul_batch(ks, options=capy.ops.CallOptions(generator=h)) == capy.ops.ed448_scalarmul_batch(ks, [g2] * n)
    with pytest.raises(TypeError):
        capy.ops.keypair_batch(pws, 384, options={"hardened": 1})


@pytest.mark.parametrize("family", ["quad", "duo"])
def test_lanes_per_item_kernels_equal_the_other_families(capy, O, family):
    """csrc/ed448_quad.h, csrc/ed448_duo.h (r04): batches of 4096 < n <= 16384 public-scalar multiplications put X, Y, Z, T of
    the accumulator into the four lanes of a quad, batches of 16384 < n <= 32768 put (Y, Z) and (X, T) into the two lanes of
    a pair.  Variable base and the verify-shaped double multiplication must give the bytes of the lane-per-item /
    wave-per-item kernels (capy_ed448_set_{quad,duo}_range(0, 0) switch the families off) at sizes with full and ragged last
    waves, on edge scalars (0, 1, r - 1, r, 2^448 - 1), the identity, a point of order 2 and random points; a sample is
    checked against the oracle, and the protocol call that uses it (verify) accepts what sign produced."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(0xAB6)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = (1 << 448) - (1 << 224) - 1
    ident = (0).to_bytes(56, "little") + (1).to_bytes(56, "little")
    order2 = (0).to_bytes(56, "little") + (p - 1).to_bytes(56, "little")
    edge_k = [0, 1, 2, R - 1, R, R + 1, (1 << 448) - 1, 1 << 447]
    fam = C.c_int(0)
    # "on" forces the family under test for every n and switches the other one off; "off" switches both off
    on = {"quad": ((0, 1 << 30), (0, 0)), "duo": ((0, 0), (0, 1 << 30))}[family]
    tag = {"quad": 33, "duo": 65}[family]

    def set_ranges(quad_range, duo_range):
        _lib.check(lib.capy_ed448_set_quad_range(*quad_range))
        _lib.check(lib.capy_ed448_set_duo_range(*duo_range))

    try:
        for n in {"quad": (4097, 9000, 16384, 32768), "duo": (33, 4097, 16385, 32768, 40001)}[family]:
            sc = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
            asc = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
            tsc = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
            for t, seed in ((sc, 1), (asc, 2), (tsc, 3)):
                _lib.check(lib.capy_fill_random_dev(t.data_ptr(), n * 56, 500 + seed + n, sp))
            eb = torch.tensor(list(b"".join(k.to_bytes(56, "big") for k in edge_k)), dtype=torch.uint8, device="cuda")
            sc[:eb.numel()] = eb
            sc[(n - len(edge_k)) * 56:] = eb
            pts = torch.empty(n * 112, dtype=torch.uint8, device="cuda")
            set_ranges((0, 0), (0, 0))
            _lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp))
            special = torch.tensor(list(ident + order2), dtype=torch.uint8, device="cuda")
            pts[8 * 112:10 * 112] = special
            outs = {}
            for name, rng_ in (("other", ((0, 0), (0, 0))), ("quad", on)):
                set_ranges(*rng_)
                vb = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
                ds = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
                _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), vb.data_ptr(), sp))
                lib.capy_debug_last_curve_kernel(C.byref(fam), None)
                assert (fam.value == tag) == (name == "quad"), (n, name, fam.value)
                _lib.check(lib.capy_ed448_double_scalarmul_batch_dev(n, asc.data_ptr(), sc.data_ptr(), pts.data_ptr(), ds.data_ptr(), sp))
                torch.cuda.synchronize()
                outs[name] = (vb, ds)
            assert torch.equal(outs["other"][0], outs["quad"][0]), n
            assert torch.equal(outs["other"][1], outs["quad"][1]), n
            hs, hp, hv = bytes(sc.cpu().numpy()), bytes(pts.cpu().numpy()), bytes(outs["quad"][0].cpu().numpy())
            for i in list(range(0, 10)) + [n // 2, n - 3, n - 1]:
                assert hv[112 * i:112 * i + 112] == O.ed448_scalarmul(hs[56 * i:56 * i + 56], hp[112 * i:112 * i + 112]), (n, i)
        # verify (double multiplication inside the protocol call) on a batch in the family's default range
        set_ranges((-1, -1), (-1, -1))
        n = {"quad": 5000, "duo": 16500}[family]
        pws = [rng.randbytes(16) for _ in range(n)]
        msgs = [rng.randbytes(40) for _ in range(n)]
        pub = capy.ops.keypair_batch(pws, 256)
        sigs = capy.ops.schnorr_sign_batch(pws, msgs, 256)
        ok = capy.ops.schnorr_verify_batch(pub, msgs, sigs, 256)
        assert all(ok)
        sigs[7] = (sigs[7][0], bytes(56))
        ok = capy.ops.schnorr_verify_batch(pub, msgs, sigs, 256)
        assert not ok[7] and sum(ok) == n - 1
    finally:
        set_ranges((-1, -1), (-1, -1))


def test_constant_address_quad_kernel_equals_the_other_hardened_kernels(capy, O):
    """vb_quad_ct_kernel (csrc/ed448_quad.h, r04): secret-scalar variable-base multiplications of 4096 < n <= 16 384 items with
    four lanes per item and the window table in LDS, every row read per window; vb_duo_ct_kernel (csrc/ed448_duo.h, r05) for
    16 384 < n <= 32 768: two lanes per item, rows 1-4 of the table in registers and rows 5-8 in LDS, one round of waves where the
    quad form needed two.  Bytes must equal those of the
    one-item-per-lane / one-item-per-wave hardened kernels (capy_ed448_set_quad_range(0, 0)) and of the indexed kernels, on
    edge scalars, the identity and a point of order 2, at sizes with ragged last waves and beyond one round of waves; a
    sample is checked against the oracle; and the protocol calls that use it (key_encrypt / key_decrypt) round-trip."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(0xC7C7)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = (1 << 448) - (1 << 224) - 1
    ident = (0).to_bytes(56, "little") + (1).to_bytes(56, "little")
    order2 = (0).to_bytes(56, "little") + (p - 1).to_bytes(56, "little")
    edge_k = [0, 1, 2, R - 1, R, R + 1, (1 << 448) - 1, 1 << 447, 8, 0x8888, (1 << 448) - 8]
    fam = C.c_int(0)
    try:
        for n in (4097, 9001, 16384, 20000, 32768):
            sc = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
            tsc = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
            for t, seed in ((sc, 1), (tsc, 3)):
                _lib.check(lib.capy_fill_random_dev(t.data_ptr(), n * 56, 900 + seed + n, sp))
            eb = torch.tensor(list(b"".join(k.to_bytes(56, "big") for k in edge_k)), dtype=torch.uint8, device="cuda")
            sc[:eb.numel()] = eb
            sc[(n - len(edge_k)) * 56:] = eb
            pts = torch.empty(n * 112, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp))
            pts[12 * 112:14 * 112] = torch.tensor(list(ident + order2), dtype=torch.uint8, device="cuda")
            outs = {}
            for name, mode, qr in (("indexed", capy.ops.HARDEN_OFF, (0, 0)), ("lane_ct", capy.ops.HARDEN_ALL, (0, 0)),
                                   ("quad_ct", capy.ops.HARDEN_ALL, (-1, -1))):
                capy.ops.ed448_set_hardened(mode)
                _lib.check(lib.capy_ed448_set_quad_range(*qr))
                _lib.check(lib.capy_ed448_set_duo_range(*((0, 0) if qr == (0, 0) else (-1, -1))))
                vb = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
                _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), vb.data_ptr(), sp))
                lib.capy_debug_last_curve_kernel(C.byref(fam), None)
                # r05: between 16 and 32 items per SIMD the constant-address form is the two-lanes-per-item kernel (66: table half
                # in registers, half in LDS, one round of waves); below, the quad form (34)
                S = 4 * torch.cuda.get_device_properties(0).multi_processor_count
                want_ct = 66 if 16 * S < n <= 32 * S else 34
                assert (fam.value == want_ct) == (name == "quad_ct"), (n, name, fam.value)
                torch.cuda.synchronize()
                outs[name] = vb
            assert torch.equal(outs["indexed"], outs["quad_ct"]), n
            assert torch.equal(outs["lane_ct"], outs["quad_ct"]), n
            hs, hp, hv = bytes(sc.cpu().numpy()), bytes(pts.cpu().numpy()), bytes(outs["quad_ct"].cpu().numpy())
            for i in list(range(0, 14)) + [n // 2, n - 3, n - 1]:
                assert hv[112 * i:112 * i + 112] == O.ed448_scalarmul(hs[56 * i:56 * i + 56], hp[112 * i:112 * i + 112]), (n, i)
        # the protocol calls whose secret-scalar multiplications take it in the default mode
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_PROTOCOL)
        _lib.check(lib.capy_ed448_set_quad_range(-1, -1))
        _lib.check(lib.capy_ed448_set_duo_range(-1, -1))
        n = 4500
        pws = [rng.randbytes(12) for _ in range(n)]
        msgs = [rng.randbytes(33) for _ in range(n)]
        pub = capy.ops.keypair_batch(pws, 256)
        ks = [rng.randbytes(56) for _ in range(n)]
        cts, zs, tags = capy.ops.key_encrypt_batch(pub, ks, msgs, 256)
        lib.capy_debug_last_curve_kernel(C.byref(fam), None)
        assert fam.value == 34, fam.value
        for i in (0, 1, n - 1):
            assert (cts[i], zs[i], tags[i]) == O.key_encrypt(pub[i], ks[i], msgs[i], 256), i
        back, ok = capy.ops.key_decrypt_batch(pws, zs, cts, tags, 256)
        assert all(ok) and back == msgs
    finally:
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_PROTOCOL)
        _lib.check(lib.capy_ed448_set_quad_range(-1, -1))
        _lib.check(lib.capy_ed448_set_duo_range(-1, -1))


def test_batches_between_the_wave_quanta_are_split_and_stay_byte_identical(capy, O):
    """csrc/ed448.hip: peel_remainder (r04).  A batch of q x 65 536 + x items (0 < x <= 32 768) in the one-item-per-lane regime
    is launched as its remainder (in the kernel family of ITS size) followed by the whole quanta.  The outputs must be those of
    item-by-item evaluation: the indexed and the constant-address paths (which split into different kernel pairs) agree byte
    for byte, the items on both sides of the cut and at both ends match the oracle, and verify (the double multiplication,
    split the same way) accepts what sign produced for such a batch."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(0x9EE1)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        for n, x in ((65536 + 37, 37), (65536 + 5000, 5000), (65536 + 20000, 20000), (131072 + 32768, 32768)):
            sc = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
            asc = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
            tsc = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
            for t, seed in ((sc, 1), (asc, 2), (tsc, 3)):
                _lib.check(lib.capy_fill_random_dev(t.data_ptr(), n * 56, 700 + seed + n, sp))
            pts = torch.empty(n * 112, dtype=torch.uint8, device="cuda")
            _lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp))
            outs = {}
            for mode in (capy.ops.HARDEN_OFF, capy.ops.HARDEN_ALL):
                capy.ops.ed448_set_hardened(mode)
                vb = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
                _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), vb.data_ptr(), sp))
                torch.cuda.synchronize()
                outs[mode] = vb
            assert torch.equal(outs[capy.ops.HARDEN_OFF], outs[capy.ops.HARDEN_ALL]), n
            ds = torch.zeros(n * 112, dtype=torch.uint8, device="cuda")
            capy.ops.ed448_set_hardened(capy.ops.HARDEN_OFF)
            _lib.check(lib.capy_ed448_double_scalarmul_batch_dev(n, asc.data_ptr(), sc.data_ptr(), pts.data_ptr(), ds.data_ptr(), sp))
            hs, ha, hp = bytes(sc.cpu().numpy()), bytes(asc.cpu().numpy()), bytes(pts.cpu().numpy())
            hv, hd = bytes(outs[capy.ops.HARDEN_OFF].cpu().numpy()), bytes(ds.cpu().numpy())
            for i in (0, 1, n - x - 2, n - x - 1, n - x, n - x + 1, n - 2, n - 1, rng.randrange(n)):
                k, a, pt = hs[56 * i:56 * i + 56], ha[56 * i:56 * i + 56], hp[112 * i:112 * i + 112]
                want = O.ed448_scalarmul(k, pt)
                assert hv[112 * i:112 * i + 112] == want, (n, i)
                assert hd[112 * i:112 * i + 112] == O.ed448_add(O.ed448_basemul(a), want), (n, i)
    finally:
        capy.ops.ed448_set_hardened(capy.ops.HARDEN_PROTOCOL)


def test_time_sliced_fused_encrypt_matches_the_two_pass_form(capy, O):
    """csrc/sponge_launch.hip (r04): sha3_encrypt / sha3_decrypt of 16 384 < n <= 22 528 uniform long messages run the fused
    four-lane kernel in TIME SLICES (one wave per SIMD per launch, the groups of 16 items taking turns, states carried in
    scratch) instead of putting a second wave on some SIMDs.  Ciphertexts and tags must be those of the two-pass form
    (capy_set_sponge_lanes(1 | 1 << 16): no fused kernel) byte for byte, items across the batch match the oracle, and decrypt
    restores the plaintext -- with one forged tag failing alone and keeping its ciphertext.  Lengths with and without a tail,
    a batch size that is not a multiple of 16, the smallest and the largest sliced batch."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(0x51CE)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        # four lanes per item in slices of one wave per SIMD (16 384 < n <= 22 528: kind 22); beyond 32 items per SIMD the
        # one-lane-per-sponge kernel on its rotating schedule (kind 25; r04's second and third slice levels are gone)
        for d, n, ln, want_kind in ((512, 16400, 136 * 600 + 77, 22), (256, 20003, 168 * 520, 22), (512, 22528, 136 * 1030 + 8, 22),
                                    (512, 33000, 136 * 520 + 16, 25), (384, 50001, 152 * 515, 25)):
            stride = (ln + 7) // 8 * 8 + 8
            pl = 32
            def rand(nbytes, seed):
                t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device="cuda")
                _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, sp))
                return t
            pws, zs, plain = rand(n * pl, 1 + n), rand(n * 512, 2 + n), rand(n * stride, 3 + n)
            res = {}
            for name, lanes in (("sliced", 0), ("two-pass", 1 | (1 << 16))):
                _lib.check(lib.capy_set_sponge_lanes(lanes))
                m = plain.clone()
                tags = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
                _lib.check(lib.capy_sha3_encrypt_batch_dev(d, n, pws.data_ptr(), pl, None, n * pl, zs.data_ptr(), m.data_ptr(), None, ln, stride,
                                                          tags.data_ptr(), sp))
                torch.cuda.synchronize()
                res[name] = (m, tags)
                if name == "sliced":  # assert the schedule that ran
                    kind, launches = C.c_int(0), C.c_int(0)
                    lib.capy_debug_last_sponge_kernel(C.byref(kind), C.byref(launches))
                    S = 4 * torch.cuda.get_device_properties(0).multi_processor_count
                    if S == 1024:
                        assert kind.value == want_kind and launches.value > 1, (n, kind.value, launches.value)
            _lib.check(lib.capy_set_sponge_lanes(0))
            assert torch.equal(res["sliced"][0], res["two-pass"][0]), (d, n, ln)
            assert torch.equal(res["sliced"][1], res["two-pass"][1]), (d, n, ln)
            m, tags = res["sliced"]
            for i in (0, 15, 16, 16383, 16384, n - 1, rng.randrange(n)):
                want = O.sha3_encrypt(bytes(pws[i * pl:(i + 1) * pl].cpu().numpy()), bytes(zs[i * 512:(i + 1) * 512].cpu().numpy()),
                                      bytes(plain[i * stride:i * stride + ln].cpu().numpy()), d)
                got = (bytes(m[i * stride:i * stride + ln].cpu().numpy()), bytes(tags[64 * i:64 * i + 64].cpu().numpy()))
                assert got == want, (d, n, ln, i)
            # bytes between the messages are not touched
            gap = torch.arange(n, device="cuda").unsqueeze(1) * stride + torch.arange(ln, stride, device="cuda").unsqueeze(0)
            assert torch.equal(m[gap.flatten()], plain[gap.flatten()]), (d, n, ln)
            # and back: item 7's tag forged
            status = torch.full((n,), 9, dtype=torch.int32, device="cuda")
            tags[64 * 7] ^= 1
            ct7 = m[7 * stride:7 * stride + ln].clone()
            _lib.check(lib.capy_sha3_decrypt_batch_dev(d, n, pws.data_ptr(), pl, None, n * pl, zs.data_ptr(), m.data_ptr(), None, ln, stride,
                                                      tags.data_ptr(), status.data_ptr(), sp))
            torch.cuda.synchronize()
            assert int(status[7]) == 1 and int((status != 0).sum()) == 1
            assert torch.equal(m[7 * stride:7 * stride + ln], ct7)
            keep = torch.ones(n, dtype=torch.bool, device="cuda")
            keep[7] = False
            idx = (torch.arange(n, device="cuda")[keep].unsqueeze(1) * stride + torch.arange(ln, device="cuda").unsqueeze(0)).flatten()
            assert torch.equal(m[idx], plain[idx]), (d, n, ln)
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))


def test_time_sliced_uniform_digests_match_the_one_lane_kernel(capy, O):
    """csrc/sponge_launch.hip: try_launch_uniform_sliced (r04).  Uniform digest batches just above two, three and four waves per
    SIMD (131 072 / 196 608 / 262 144 items on this chip) run as a sequence of launches of exactly that many waves per SIMD,
    groups of 64 items taking turns.  SHA3-256, SHA3-512 and keyed KMACXOF (per-item head blocks, a long output) must give the
    bytes of the forced one-lane kernel (capy_set_sponge_lanes(1)), and items across the batch -- first and last group, both
    sides of a launch boundary -- those of the oracle.  Batch sizes that are not multiples of 64, lengths with and without a
    tail."""
    import torch

    from capycrypt_amd import _lib

    lib = _lib.lib()
    rng = random.Random(0x51CF)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def rand(nbytes, seed):
        t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device="cuda")
        _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, sp))
        return t

    try:
        for d, n, ln, keyed in ((256, 133121, 136 * 520 + 40, False), (512, 200704, 72 * 600, False), (512, 266245, 136 * 515 + 8, True)):
            stride = (ln + 7) // 8 * 8 + 8
            msgs = rand(n * stride, 11 + n)
            keys = rand(n * 64, 12 + n)
            ol = 208 if keyed else d // 8  # a multiple of 16: long outputs leave the uniform-framing kernel as whole lines
            os_ = (ol + 15) // 16 * 16
            outs = {}
            for name, lanes in (("sliced", 0), ("one-lane", 1)):
                _lib.check(lib.capy_set_sponge_lanes(lanes))
                o = torch.zeros(n * os_ + 16, dtype=torch.uint8, device="cuda")
                if keyed:
                    _lib.check(lib.capy_kmac_xof_batch_dev(d, n, keys.data_ptr(), 64, 64, None, msgs.data_ptr(), None, ln, stride, 8 * ol, b"T", 1,
                                                          o.data_ptr(), os_, sp))
                else:
                    _lib.check(lib.capy_sha3_batch_dev(d, n, msgs.data_ptr(), None, ln, stride, o.data_ptr(), sp))
                torch.cuda.synchronize()
                outs[name] = o
                if name == "sliced":  # the sliced path must be the one that ran (9: uniform-framing kernel in time slices)
                    kind, launches = C.c_int(0), C.c_int(0)
                    lib.capy_debug_last_sponge_kernel(C.byref(kind), C.byref(launches))
                    S = 4 * torch.cuda.get_device_properties(0).multi_processor_count
                    if S == 1024:
                        assert kind.value == 9 and launches.value > 1, (n, kind.value, launches.value)
                    elif kind.value != 9:
                        pytest.skip("batch sizes of this test are sliced on a 1024-SIMD device only (this one: %d SIMDs)" % S)
            _lib.check(lib.capy_set_sponge_lanes(0))
            assert torch.equal(outs["sliced"], outs["one-lane"]), (d, n, ln, keyed)
            row = os_ if keyed else d // 8
            for i in (0, 63, 64, 65535, 65536, 131071, 131072, n - 65, n - 1, rng.randrange(n)):
                x = bytes(msgs[i * stride:i * stride + ln].cpu().numpy())
                got = bytes(outs["sliced"][i * row:i * row + ol].cpu().numpy())
                want = O.kmac_xof(bytes(keys[i * 64:(i + 1) * 64].cpu().numpy()), x, 8 * ol, b"T", d) if keyed else O.sha3(x, d)
                assert got == want, (d, n, ln, keyed, i)
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))
