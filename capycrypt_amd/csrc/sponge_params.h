// sponge_params.h — launch parameters and the byte-granular view of a sponge's input stream
// (shared by the one-lane and two-lane kernels).  See sponge_kernels.h for the framing.
#pragma once
#include "keccak_dev.h"

namespace capy {

// set by the launcher (never by the caller) in SpongeParams::debug_flags: the one-lane kernels load message blocks per
// lane instead of cooperatively through LDS (the default; cleared by debug bit 6 for A/B runs)
constexpr uint32_t SPONGE_DIRECT_LOADS = 1u << 16;
// debug bit 10 (A/B): long squeezes leave as rate blocks at their own offsets (r01-r03) instead of whole 128-byte lines
constexpr uint32_t SPONGE_BLOCK_OUT = 1u << 10;

struct SpongeParams {
    uint64_t init_state[25];  // state after the batch-shared prefix (zeros for SHA3)
    // batch-shared prefix bytes that could NOT be folded into init_state (only when the prefix is
    // not a whole number of absorb blocks, i.e. cSHAKE/KMAC at D224 where r = 172 but 168 B are consumed)
    const uint8_t *pre;
    uint32_t pre_len;
    // per-item head = hdr bytes || key bytes || zeros up to head_len   (head_len = 0: no head)
    const uint8_t *keys;
    uint64_t key_stride;
    // optional n+1 byte offsets into keys: key i = keys[key_offsets[i] .. key_offsets[i+1]), every item with its own
    // length (the reference takes any &[u8] per message: src/ecc/signable.rs:40-43, src/sha3/hashable.rs:33-35).  The
    // header bytes and the head length then follow from the item's key length (item_head below) and key_stride,
    // key_len, hdr_len, hdr0/1, head_len are ignored; bytepad_w is the SP 800-185 bytepad width w.
    const uint64_t *key_offsets;
    uint32_t bytepad_w;
    uint32_t key_len;
    uint32_t hdr_len;
    uint64_t hdr0, hdr1;  // up to 16 header bytes, little-endian packed
    uint32_t head_len;
    // body / xor target
    const uint8_t *msgs;
    const uint64_t *offsets;  // n+1 byte offsets into msgs, or null: item i at msgs + i*msg_stride
    const uint64_t *lens;     // optional n lengths (aligned re-packed batches); null: offsets[i+1]-offsets[i]
    uint64_t msg_stride;
    uint64_t uniform_len;
    uint32_t absorb_body;  // 0: the body is not absorbed (keystream mode: X = "")
    // trailer
    uint64_t suffix;  // up to 8 suffix bytes, little-endian packed
    uint32_t suffix_len;
    uint32_t sha3_suffix_rule;  // reference shake(): first suffix byte is 0x86 iff len % 136 == 135
    uint32_t fips_pad;          // 0: reference pad rule (pad only if unaligned); 1: FIPS 202 pad10*1
    uint32_t stride_bytes;      // the reference's `r` (172 for cSHAKE/KMAC at D224), else 8*RW
    // output
    uint32_t out_mode;  // 0: write out_len bytes per item; 1: XOR keystream into msgs in place
    uint32_t sq_words;  // words emitted per squeeze block
    uint8_t *out;
    uint64_t out_stride;
    uint32_t out_len;
    const int32_t *mask;  // optional: only items with mask[i] != 0 are processed
    // optional processing order (ragged batches): slot k of the grid works on item order[k].  The launcher sorts by
    // length so that the lanes of a wave finish together; outputs stay at the item's own index.
    const uint32_t *order;
    uint32_t debug_flags;  // bit 0: do not use the wave-uniform addressing path (A/B measurements); SPONGE_DIRECT_LOADS
    // resume (one-lane digest kernel only): the first resume_blocks blocks were absorbed by sponge_mixed_kernel,
    // whose states sit word-major in resume_state[25][resume_pad]; only the tail blocks and the squeeze remain
    const uint64_t *resume_state;
    uint64_t resume_pad;
    uint32_t resume_blocks;
    // head-only launch (one-lane digest kernel only): absorb the per-item head blocks, write the states word-major
    // to head_state[25][resume_pad] and stop (the body then goes through sponge_mixed_kernel)
    uint64_t *head_state;
    uint64_t n;
    // time-sliced launches of the uniform-framing kernel (sponge_uniform.h, r04): wave w of launch sl_launch works on the group
    // of 64 items (sl_launch * gridDim.x + w) mod sl_groups for at most sl_blocks full body blocks, resuming from / saving to
    // sl_state ([group][50][64] half-words) with its progress in sl_done[group] (0xffffffff: not started, 0xfffffffe:
    // finished).  sl_groups == 0: the whole job in one launch.
    uint32_t sl_groups, sl_launch, sl_blocks;
    uint32_t *sl_done;
    uint32_t *sl_state;
};

struct ItemCtx {
    const uint8_t *key;
    const uint8_t *msg;
    uint64_t len;     // absorbed body length
    uint64_t padded;  // head + body + suffix + pad
    uint64_t suffix;
    bool pad80;
    // the item's head: bytepad(encode_string(K_i), w) = hdr (hdr_len bytes of hdr0) || K_i || zeros up to head_len
    uint32_t key_len, hdr_len, head_len;
    uint64_t hdr0, hdr1;
};

// Fill the head fields (and the key pointer) of item `item`.  Uniform keys: copies of the launch parameters (scalar
// registers).  Per-item keys (p.key_offsets): hdr = left_encode(w) || left_encode(8 |K_i|), built here exactly as the
// reference's byte_pad(encode_string(k), w) does (src/sha3/aux_functions.rs:11-49, src/sha3/shake_functions.rs:84):
// head_len = z + (w - z % w) with z = hdr_len + |K_i| (a full extra block of zeros when z is a multiple of w).
__device__ __forceinline__ void item_head(const SpongeParams &p, uint64_t item, bool active, ItemCtx &c)
{
    if (p.key_offsets == nullptr) {  // wave-uniform
        c.key = active ? p.keys + item * p.key_stride : nullptr;
        c.key_len = p.key_len;
        c.hdr_len = p.hdr_len;
        c.hdr0 = p.hdr0;
        c.hdr1 = p.hdr1;
        c.head_len = p.head_len;
        return;
    }
    uint64_t o0 = 0;
    uint32_t klen = 0;
    if (active) {
        o0 = p.key_offsets[item];
        klen = (uint32_t)(p.key_offsets[item + 1] - o0);
    }
    c.key = p.keys + o0;
    c.key_len = klen;
    const uint32_t w = p.bytepad_w, bits = klen * 8;  // w < 256, |K_i| <= 2^20 (checked on the host): bits < 2^24
    const uint32_t nb = bits < 0x100 ? 1 : (bits < 0x10000 ? 2 : 3);
    uint64_t h = 1 | ((uint64_t)w << 8) | ((uint64_t)nb << 16);
    // big-endian bytes of `bits` at positions 3 .. 3+nb-1
    const uint32_t be = nb == 1 ? bits : (nb == 2 ? ((bits >> 8) | ((bits & 0xff) << 8))
                                                  : ((bits >> 16) | (bits & 0xff00) | ((bits & 0xff) << 16)));
    h |= (uint64_t)be << 24;
    c.hdr0 = h;
    c.hdr1 = 0;
    c.hdr_len = 3 + nb;
    const uint32_t z = c.hdr_len + klen;
    c.head_len = z + (w - z % w);
}

__device__ __forceinline__ uint32_t stream_byte(const SpongeParams &p, const ItemCtx &c, uint64_t pos)
{
    uint32_t v = 0;
    if (pos < p.pre_len) {
        v = p.pre[pos];
        if (c.pad80 && pos + 1 == c.padded) v |= 0x80;
        return v;
    }
    pos -= p.pre_len;
    if (pos < c.head_len) {
        if (pos < c.hdr_len) {
            v = (uint32_t)((pos < 8 ? c.hdr0 >> (8 * pos) : c.hdr1 >> (8 * (pos - 8))) & 0xff);
        } else {
            uint64_t k = pos - c.hdr_len;
            if (k < c.key_len) v = c.key[k];
        }
    } else {
        uint64_t q = pos - c.head_len;
        if (q < c.len) {
            v = c.msg[q];
        } else {
            q -= c.len;
            if (q < p.suffix_len) v = (uint32_t)((c.suffix >> (8 * q)) & 0xff);
        }
    }
    if (c.pad80 && pos + p.pre_len + 1 == c.padded) v |= 0x80;
    return v;
}

__device__ __forceinline__ uint64_t stream_word(const SpongeParams &p, const ItemCtx &c, uint64_t pos)
{
    const uint64_t body0 = (uint64_t)p.pre_len + c.head_len;
    if (pos >= body0) {
        const uint64_t body_end = body0 + c.len;
        const uint8_t *a = c.msg + (pos - body0);
        const bool aligned = ((uintptr_t)a & 7) == 0;
        if (pos + 8 <= body_end) {
            // whole word inside the body and 8-byte aligned in memory: one load
            if (aligned) return *reinterpret_cast<const uint64_t *>(a);
        } else if (pos >= body_end || aligned) {
            // trailer word: the last 1..7 body bytes (if any), then suffix bytes, zeros and the final pad bit, built
            // without a byte loop.  The 8-byte load stays inside the aligned word that holds the last body byte.
            uint64_t v = 0;
            uint32_t off = 0;
            if (pos < body_end) {
                off = (uint32_t)(body_end - pos);
                v = *reinterpret_cast<const uint64_t *>(a) & ((1ULL << (8 * off)) - 1);
            }
            const uint64_t s0 = pos > body_end ? pos - body_end : 0;
            if (s0 < p.suffix_len) {
                const uint64_t sfx = p.suffix_len >= 8 ? c.suffix : (c.suffix & ((1ULL << (8 * p.suffix_len)) - 1));
                v |= (sfx >> (8 * s0)) << (8 * off);
            }
            if (c.pad80 && pos + 8 == c.padded) v |= 0x80ULL << 56;
            return v;
        }
    }
    // whole word inside the per-item key (any alignment) or inside the head's zero fill
    if (pos >= (uint64_t)p.pre_len + c.hdr_len && pos + 8 <= body0) {
        const uint64_t k = pos - p.pre_len - c.hdr_len;
        if (k >= c.key_len) return 0;
        if (k + 8 <= c.key_len) {
            const uint8_t *a = c.key + k;
            if (((uintptr_t)a & 7) == 0) return *reinterpret_cast<const uint64_t *>(a);
            uint64_t w = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) w |= (uint64_t)a[j] << (8 * j);
            return w;
        }
    }
    uint64_t w = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) w |= (uint64_t)stream_byte(p, c, pos + j) << (8 * j);
    return w;
}

// Cooperative block loads of ragged batches are unconditional (17..21 back-to-back loads per block step; predicating
// them serialised the loads behind exec-mask regions: the ragged issue-tuned instance waited 56 % of its cycles).  A slot
// whose message has run out re-reads that message's own first bytes, clamped to the last 8 bytes of the batch buffer:
// always mapped, and a different line for every message (one shared fallback address was a hot L2 line).
__device__ __forceinline__ const uint8_t *batch_last_word(const uint8_t *msgs, const uint64_t *offsets, uint64_t n,
                                                          uint64_t stride, uint64_t uniform_len)
{
    const uint64_t total = offsets ? offsets[n] : (n ? (n - 1) * stride + uniform_len : 0);
    return msgs + (total >= 8 ? total - 8 : 0);
}
// 8-byte load through an explicitly GLOBAL pointer.  Message pointers rebuilt from integers (LDS tables, offsets) are
// generic ("flat") pointers to the compiler, which then orders every such load against the LDS stores of the staging
// buffer in between: load, wait, store, load, ... -- 17..21 serial round trips per block step.
__device__ __forceinline__ uint64_t load_global_u64(const uint8_t *q)
{
    return *reinterpret_cast<const __attribute__((address_space(1))) uint64_t *>(reinterpret_cast<uintptr_t>(q));
}
__device__ __forceinline__ void store_global_u64(uint8_t *q, uint64_t v)
{
    *reinterpret_cast<__attribute__((address_space(1))) uint64_t *>(reinterpret_cast<uintptr_t>(q)) = v;
}
__device__ __forceinline__ const uint8_t *ragged_src(bool live, const uint8_t *base, uint64_t byte_off, const uint8_t *last_word)
{
    const uint8_t *fallback = base < last_word ? base : last_word;
    return live ? base + byte_off : fallback;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

}  // namespace capy
