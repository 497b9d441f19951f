"""The measurement harness's deterministic input generator, host side.

Counter-mode SplitMix64: 8-byte word i of a stream is mix(seed + (i + 1) * 0x9E3779B97F4A7C15).  The device side is
`capy_fill_random_dev` (csrc/sponge.hip: fill_random_kernel); both produce the same bytes, so any shard of a batch can
be regenerated on either side from (seed, word offset) alone.  Seeds follow 0xCA9C0000 + config index (+ rank).
Not a CSPRNG: the reference draws nonces from `thread_rng` (aux_functions.rs:80-84); here randomness is an input.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def words(seed, n_words, first_word=0):
    """n_words little-endian u64 values of stream `seed`, starting at word index first_word."""
    with np.errstate(over="ignore"):
        i = np.arange(first_word + 1, first_word + 1 + n_words, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def fill(seed, n_bytes, first_byte=0):
    """n_bytes of stream `seed` starting at byte offset first_byte (any alignment)."""
    w0 = first_byte // 8
    w1 = (first_byte + n_bytes + 7) // 8
    raw = words(seed, w1 - w0, w0).astype("<u8").tobytes()
    lo = first_byte - 8 * w0
    return raw[lo:lo + n_bytes]
