"""GPU: the multi-device path INSIDE the C ABI (capy_set_devices, include/capyhip.h).
The GPU box has one card, so the device list is {0, 0} (and {0, 0, 0}): two (three) worker threads, each with its own
scratch set, sharing the device's fixed-base table -- the same code that runs on ids {0..7} of an 8-GPU node.  Every
result must be bit-identical to the single-device call, in input order, including in-place effects and per-item status."""
import ctypes as C
import random

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def devices():
    from capycrypt_amd import _lib

    lib = _lib.lib()

    def set_ids(ids):
        arr = (C.c_int * max(1, len(ids)))(*ids)
        _lib.check(lib.capy_set_devices(arr, len(ids)))
        got = (C.c_int * 8)()
        assert lib.capy_get_devices(got, 8) == len(ids) and list(got[:len(ids)]) == list(ids)

    yield set_ids
    _lib.check(lib.capy_set_devices(None, 0))


def _batch(rng, n):
    msgs = [rng.randbytes(rng.choice([0, 1, 135, 136, 137, 1000, 5000, 70000]) if i % 3 else rng.randrange(0, 3000))
            for i in range(n)]
    msgs[n // 2] = rng.randbytes(400000)  # one heavy item: the byte-balanced cut is not the count-balanced one
    return msgs


@pytest.mark.parametrize("ids", [[0, 0], [0, 0, 0], [0]])
def test_sharded_calls_equal_the_single_device_call(devices, ids):
    from capycrypt_amd import ops

    rng = random.Random(0xD0 + len(ids))
    n = 203
    msgs = _batch(rng, n)
    pws = [rng.randbytes(rng.choice([0, 5, 32, 64, 200])) for _ in range(n)]
    zs = [rng.randbytes(512) for _ in range(n)]
    ks = [rng.randbytes(56) for _ in range(n)]
    sc = [rng.randbytes(56) for _ in range(n)]

    def run():
        r = {}
        r["sha3"] = ops.sha3_batch(msgs, 256)
        r["cshake"] = ops.cshake_batch(msgs, 512, b"", b"Email Signature", 512)
        r["kmac"] = ops.kmac_xof_batch(pws, msgs, 448, b"T", 256)
        r["enc"] = ops.sha3_encrypt_batch(pws, zs, msgs, 512)
        cts, tags = r["enc"]
        tags = list(tags)
        tags[7] = bytes(64)
        r["dec"] = ops.sha3_decrypt_batch(pws, zs, cts, tags, 512)
        r["kem"] = ops.kem_sponge_encrypt_batch([p[:0] + bytes(32) for p in pws], zs, msgs, 256)
        r["pub"] = ops.keypair_batch(pws, 512)
        r["sig"] = ops.schnorr_sign_batch(pws, msgs, 512)
        sig = list(r["sig"])
        sig[11] = (sig[11][0], bytes(56))
        r["ver"] = ops.schnorr_verify_batch(r["pub"], msgs, sig, 512)
        r["kenc"] = ops.key_encrypt_batch(r["pub"], ks, msgs, 512)
        c2, z2, t2 = r["kenc"]
        r["kdec"] = ops.key_decrypt_batch(pws[:3] + [b"nope"] + pws[4:], z2, c2, t2, 512)
        r["fb"] = ops.ed448_basemul_batch(sc)
        r["vb"] = ops.ed448_scalarmul_batch(sc, r["pub"])
        r["add"] = ops.ed448_add_batch(r["pub"], r["fb"])
        r["dsm"] = ops.ed448_double_scalarmul_batch(sc, ks, r["pub"])
        r["val"] = ops.ed448_validate_batch(r["pub"][:-1] + [bytes(112)])
        return r

    single = run()
    devices(ids)
    sharded = run()
    for k in single:
        assert sharded[k] == single[k], k
    assert single["dec"][1] == [i != 7 for i in range(n)] and single["ver"] == [i != 11 for i in range(n)]
    assert single["kdec"][1] == [i != 3 for i in range(n)] and single["val"] == [True] * (n - 1) + [False]


def test_sharded_error_is_reported_and_batch_smaller_than_device_list(devices):
    from capycrypt_amd import _lib, ops

    devices([0, 0, 0, 0])
    assert ops.sha3_batch([b"abc"], 256) == ops.sha3_batch([b"abc"], 256)  # one item, four devices: one shard works
    assert ops.sha3_batch([], 256) == []
    with pytest.raises(_lib.CapyHipError) as e:
        ops.sha3_batch([b"a", b"b", b"c", b"d"], 300)
    assert e.value.code == _lib.CAPY_ERR_UNSUPPORTED_SECPARAM
    lib = _lib.lib()
    assert lib.capy_set_devices((C.c_int * 1)(99), 1) == _lib.CAPY_ERR_ARG
