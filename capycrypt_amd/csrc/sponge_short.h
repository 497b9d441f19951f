// sponge_short.h — digests of MANY SHORT, equally long messages (SHA3 / SHAKE-type framing without keys).
//
// A batch of 64-byte messages is one permutation per item; in the generic kernel (sponge_kernels.h) all of it goes
// through the byte-granular tail path, whose region tests work on per-lane head fields since the per-item-key rework
// (2^22 x 64 B: 0.72 ms against 0.39 ms for the bare permutations).  With equal lengths everything about the framing is
// WAVE-UNIFORM -- which word holds the last message bytes, where the suffix bytes and the pad bit go, how many blocks
// there are -- so this kernel decides it with scalar code and keeps vector work to the loads, the one partially
// filled word and the permutation.  Same SpongeParams, same quirks (the reference's suffix rule, pad-only-if-unaligned),
// bit-identical digests; used for uniform batches of at least 128 items per SIMD whose messages are at most
// four rate blocks long (suffix included), 8-byte aligned, without prefix, head, mask, order or a squeeze longer than one block.
#pragma once
#include "sponge_kernels.h"

namespace capy {


// the 8 stream bytes at body position `pos` (a multiple of 8) of a message of `len` bytes followed by the suffix, zeros
// and the final pad bit: the trailer branch of stream_word() with every test on wave-uniform values
__device__ __forceinline__ uint64_t short_word(const uint8_t *msg, uint64_t pos, uint64_t len, uint64_t sfx, uint32_t sfx_len,
                                               bool pad80, uint64_t padded)
{
    if (pos + 8 <= len) return load_global_u64(msg + pos);  // uniform
    uint64_t v = 0;
    uint32_t off = 0;
    if (pos < len) {  // uniform: the word that holds the last 1..7 message bytes
        off = (uint32_t)(len - pos);
        v = load_global_u64(msg + pos) & ((1ULL << (8 * off)) - 1);
    }
    const uint64_t s0 = pos > len ? pos - len : 0;
    if (s0 < sfx_len) v |= (sfx >> (8 * s0)) << (8 * off);
    if (pad80 && pos + 8 == padded) v |= 0x80ULL << 56;
    return v;
}

template <int RW>
__global__ __launch_bounds__(64, 4) void sponge_short_kernel(const SpongeParams p)
{
    constexpr uint32_t RB = RW * 8;
    const uint32_t lane = threadIdx.x;
    const uint64_t item0 = (uint64_t)blockIdx.x * 64;
    const bool active = item0 + lane < p.n;
    const uint64_t item = active ? item0 + lane : p.n - 1;  // lanes past the batch redo the last item and store nothing
    const uint8_t *msg = p.msgs + item * p.msg_stride;

    // framing, all scalar: suffix byte, padded length, block count (sponge_kernel's prologue with uniform inputs)
    const uint64_t len = p.uniform_len;
    uint64_t sfx = p.suffix;
    if (p.sha3_suffix_rule && (len % 136) == 135) sfx = (p.suffix & ~0xffULL) | 0x86;
    if (p.suffix_len < 8) sfx &= (1ULL << (8 * p.suffix_len)) - 1;
    const uint64_t total = len + p.suffix_len;
    const uint32_t rem = (uint32_t)(total % RB);
    const bool pad80 = p.fips_pad || rem != 0;
    const uint64_t padded = rem ? total + (RB - rem) : total;
    const uint32_t nb = (uint32_t)(padded / RB);

    KState a;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        a.lo[i] = (uint32_t)p.init_state[i];
        a.hi[i] = (uint32_t)(p.init_state[i] >> 32);
    }
    for (uint32_t b = 0; b < nb; b++) {
        const uint64_t base = (uint64_t)b * RB;
#pragma unroll
        for (int w = 0; w < RW; w++) xor_word(a, w, short_word(msg, base + 8 * w, len, sfx, p.suffix_len, pad80, padded));
        keccakf1600_paired<CAPY_PAIRED_PRIO>(a);
    }
    // one squeeze block at most: out_len <= 8 * sq_words bytes, rows 8-byte aligned (checked by the launcher)
    if (active) {
        uint8_t *o = p.out + item * p.out_stride;
#pragma unroll
        for (int w = 0; w < RW; w++)
            if ((uint32_t)(8 * w) < p.out_len) {
                const uint64_t v = state_word(a, w);
                if ((uint32_t)(8 * w + 8) <= p.out_len)
                    store_global_u64(o + 8 * w, v);
                else
                    for (uint32_t j = 0; 8 * w + j < p.out_len; j++) o[8 * w + j] = (uint8_t)(v >> (8 * j));
            }
    }
}

}  // namespace capy
