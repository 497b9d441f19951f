"""Seeded differential fuzz of the sponge path: random security parameters, batch sizes and ragged lengths (empty
messages, lengths on and around every block boundary, unaligned starts) through the host-buffer C ABI under each
kernel choice; every output must equal the CPU oracle's (reference quirks on), and encrypt -> decrypt must round-trip.
Sizes are small enough for the oracle to finish in seconds."""
import os
import random

import pytest

pytestmark = pytest.mark.gpu

RATES = {224: (144, 172), 256: (136, 168), 384: (104, 152), 512: (72, 136)}  # (sha3 rate, cshake/kmac rate) bytes


def _lengths(rng, n, d):
    r1, r2 = RATES[d]
    special = [0, 1, 7, 8, 9, r1 - 2, r1 - 1, r1, r1 + 1, r2 - 4, r2 - 3, r2 - 2, r2 - 1, r2, r2 + 1, 2 * r1 - 1, 2 * r1,
               2 * r2 - 3, 135, 136, 167, 168, 3 * r2]
    if rng.random() < 0.3:  # equal lengths: the host batch is recognised as uniform (strided fast path)
        return [rng.choice(special + [rng.randrange(0, 3000)])] * n
    out = []
    for _ in range(n):
        c = rng.random()
        if c < 0.45:
            out.append(rng.choice(special))
        elif c < 0.9:
            out.append(rng.randrange(0, 700))
        else:
            out.append(rng.randrange(700, 5000))
    return out


@pytest.mark.parametrize("lanes", [0, 1, 2, 2 | (1 << 8), 1 | (1 << 16), 32 << 8, 64 << 8, 1 | (64 << 8) | (1 << 16)],
                         ids=["auto", "one-lane", "two-lane", "two-lane,no-uniform", "one-lane,two-pass",
                              "auto,wave-per-item-kernels", "auto,lds-staged-loads", "one-lane,two-pass,lds-staged-loads"])
@pytest.mark.parametrize("seed", list(range(1, 1 + int(os.environ.get("CAPY_FUZZ_SEEDS", "8")))))  # soak: raise it
def test_sponge_fuzz_against_oracle(lanes, seed):
    from capycrypt_amd import _lib, ops
    from oracle import oracle as O

    rng = random.Random(0xF022 + seed)
    _lib.check(_lib.lib().capy_set_sponge_lanes(lanes))
    try:
        for d in (224, 256, 384, 512):
            n = rng.choice([1, 2, 31, 33, 63, 65, 100, 130, 200, 333])  # >= 128 ragged: length-sorted processing order
            lens = _lengths(rng, n, d)
            msgs = [rng.randbytes(x) for x in lens]
            picks = sorted(set([0, n - 1] + [rng.randrange(n) for _ in range(10)]))

            got = ops.sha3_batch(msgs, d)
            for i in picks:
                assert got[i] == O.sha3(msgs[i], d), ("sha3", d, lens[i])

            lbits = rng.choice([8, 448, 512, 1088, 1600, 8 * 400])
            cs = rng.randbytes(rng.randrange(0, 40))
            got = ops.cshake_batch(msgs, lbits, b"FN", cs, d)
            for i in picks:
                assert got[i] == O.cshake(msgs[i], lbits, b"FN", cs, d), ("cshake", d, lens[i], lbits)

            klen = rng.choice([0, 1, 32, 56, 64, 130, 200])
            # half of the draws: one key / password length per ITEM (the reference takes any &[u8] per message,
            # src/sha3/hashable.rs:33-35, src/sha3/encryptable.rs:29), incl. the lengths around the bytepad boundary
            r2 = RATES[d][1]
            ragged_keys = rng.random() < 0.5
            def klen_i():
                return rng.choice([0, 1, 31, 32, 33, 56, 64, r2 - 5, r2 - 4, r2 - 3, r2, 2 * r2 - 4, 300]) if ragged_keys else klen
            keys = [rng.randbytes(klen_i()) for _ in range(n)]
            got = ops.kmac_xof_batch(keys, msgs, lbits, cs, d)
            for i in picks:
                assert got[i] == O.kmac_xof(keys[i], msgs[i], lbits, cs, d), ("kmac", d, len(keys[i]), lens[i], lbits)

            pws = [rng.randbytes(klen_i()) for _ in range(n)]
            zs = [rng.randbytes(512) for _ in range(n)]
            cts, tags = ops.sha3_encrypt_batch(pws, zs, msgs, d)
            for i in picks:
                ect, etag = O.sha3_encrypt(pws[i], zs[i], msgs[i], d)
                assert cts[i] == ect and tags[i] == etag, ("encrypt", d, len(pws[i]), lens[i])
            bad = rng.randrange(n)
            tags2 = list(tags)
            tags2[bad] = bytes([tags[bad][0] ^ 0x40]) + tags[bad][1:]
            pts, ok = ops.sha3_decrypt_batch(pws, zs, cts, tags2, d)
            for i in range(n):
                if i == bad:
                    assert not ok[i] and pts[i] == cts[i]  # failure keeps the ciphertext (encryptable.rs:77-82)
                else:
                    assert ok[i] and pts[i] == msgs[i], ("decrypt", d, lens[i])
    finally:
        _lib.check(_lib.lib().capy_set_sponge_lanes(0))
