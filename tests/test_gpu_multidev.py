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


@pytest.mark.parametrize("ids", [[0, 0], [0, 0, 0], [0], [0] * 8])
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


def test_persistent_workers_survive_list_changes_and_concurrent_callers(devices):
    """r03: the workers of capy_set_devices are long-lived (one per list position, own scratch pools and staging-buffer
    cache).  Changing the list stops the old workers and starts new ones with the next sharded call; releasing the calling
    thread's workspace between calls must not disturb them; sharded calls from several host threads take turns and every
    one of them gets the single-device result."""
    import threading

    from capycrypt_amd import _lib, ops

    rng = random.Random(0x70B)
    msgs = [rng.randbytes(rng.randrange(0, 4000)) for _ in range(150)]
    pws = [rng.randbytes(20) for _ in msgs]
    want_sha, want_sig = ops.sha3_batch(msgs, 512), ops.schnorr_sign_batch(pws, msgs, 256)
    for ids in ([0, 0], [0, 0, 0, 0], [0], [0, 0]):
        devices(ids)
        for _ in range(3):  # repeated calls reuse the workers' pools; sizes differ from call to call
            k = rng.randrange(1, len(msgs))
            assert ops.sha3_batch(msgs[:k], 512) == want_sha[:k]
        assert ops.schnorr_sign_batch(pws, msgs, 256) == want_sig
        _lib.check(_lib.lib().capy_release_workspace())
    devices([0, 0, 0])
    errors = []

    def caller(seed):
        r = random.Random(seed)
        try:
            for _ in range(4):
                k = r.randrange(1, len(msgs))
                if ops.sha3_batch(msgs[:k], 512) != want_sha[:k] or ops.schnorr_sign_batch(pws[:k], msgs[:k], 256) != want_sig[:k]:
                    errors.append("mismatch in thread %d" % seed)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=caller, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_staging_buffer_cache_reuse_and_release():
    """r03: the host-buffer entry points take their device staging buffers from a per-thread cache (common.h: DevBuf).
    Calls of growing, shrinking and equal sizes must give the same results as fresh allocations would, secrets are
    zeroed before a block is reused (a later, larger KMAC with a shorter key must not see the previous key's bytes), and
    capy_release_workspace() empties the cache without affecting later calls."""
    from capycrypt_amd import _lib, ops
    from oracle import oracle as O

    rng = random.Random(0xCAC4E)
    for n, mlen, klen in ((300, 2000, 64), (7, 90000, 200), (300, 2000, 3), (1, 5, 0), (64, 136, 64), (300, 1999, 64)):
        msgs = [rng.randbytes(mlen) for _ in range(n)]
        keys = [rng.randbytes(klen) for _ in range(n)]
        got = ops.kmac_xof_batch(keys, msgs, 512, b"cache", 512)
        for i in (0, n // 2, n - 1):
            assert got[i] == O.kmac_xof(keys[i], msgs[i], 512, b"cache", 512), (n, mlen, klen, i)
        if n == 7:
            _lib.check(_lib.lib().capy_release_workspace())


def test_every_visible_device_once_each(devices):
    """r06: capy_set_devices with EVERY visible device exactly once -- one entry on this pool, eight on the
    real node, the same test -- over the three BASELINE shapes that shard: a config-1-shaped SHA3-256 batch (equal long messages),
    config 4 (Ed448 variable base on distinct points) and config 5 (key pairs, sign, verify), each compared with the call on
    device 0 alone; the cut is the one capy_shard_plan announces; every device reports a topology entry (PCI bus id; the CPU set is
    empty only where sysfs hides the device).  What this cannot prove on one card -- distinct devices, more than one PCIe link,
    NUMA pinning on a two-socket host -- it will the first time it runs on a node with more."""
    import torch

    from capycrypt_amd import _lib, ops, sharding

    lib = _lib.lib()
    ndev = lib.capy_device_count()
    assert ndev == torch.cuda.device_count() >= 1
    ids = list(range(ndev))
    topo = [sharding.device_topology(i) for i in ids]
    assert len({t["pci_bus_id"] for t in topo}) == ndev and all(t["pci_bus_id"] for t in topo), topo
    for t in topo:
        assert t["n_cpus"] == len(t["cpu_ids"]) and set(t["cpu_ids"]) <= set(__import__("os").sched_getaffinity(0))
    rng = random.Random(0xA11)
    n1 = 96 * ndev + 5                                  # config-1 shape: equal long messages (5 MiB there, 192 KiB here)
    msgs1 = [rng.randbytes(192 * 1024) for _ in range(n1)]
    n4 = 1536 * ndev + 7                                # config 4: distinct (scalar, point) pairs
    sc = [rng.randbytes(56) for _ in range(n4)]
    ts = [rng.randbytes(56) for _ in range(n4)]
    n5 = 640 * ndev + 3                                 # config 5: 1 KiB messages
    msgs5 = [rng.randbytes(1024) for _ in range(n5)]
    pws = [rng.randbytes(64) for _ in range(n5)]

    def run():
        r = {"sha3": ops.sha3_batch(msgs1, 256)}
        pts = ops.ed448_basemul_batch(ts)
        r["vb"] = ops.ed448_scalarmul_batch(sc, pts)
        r["pub"] = ops.keypair_batch(pws, 512)
        r["sig"] = ops.schnorr_sign_batch(pws, msgs5, 512)
        sig = list(r["sig"])
        sig[n5 - 1] = (bytes(56), sig[n5 - 1][1])
        r["ver"] = ops.schnorr_verify_batch(r["pub"], msgs5, sig, 512)
        return r

    devices([0])
    single = run()
    devices(ids)
    sharded = run()
    for k in single:
        assert sharded[k] == single[k], k
    assert single["ver"] == [True] * (n5 - 1) + [False]
    import hashlib

    assert single["sha3"][:4] == [hashlib.sha3_256(m).digest() for m in msgs1[:4]]
    # the announced cut: contiguous, covers everything, count-balanced for equal messages
    bounds = (C.c_uint64 * (ndev + 1))()
    offs = (C.c_uint64 * (n1 + 1))(*[i * 192 * 1024 for i in range(n1 + 1)])
    _lib.check(lib.capy_shard_plan(n1, ndev, offs, bounds))
    b = list(bounds)
    assert b[0] == 0 and b[-1] == n1 and all(0 <= b[i + 1] - b[i] - n1 // ndev <= 1 for i in range(ndev)), b
