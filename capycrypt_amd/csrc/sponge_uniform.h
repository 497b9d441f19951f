// sponge_uniform.h — chip-full batches whose framing is WAVE-UNIFORM: one sponge per lane, every decision scalar.
//
// The generic one-lane kernel (sponge_kernels.h) serves ragged batches, per-item key lengths, raw prefixes, masks and
// processing orders; its issue-tuned instance pays for that generality with ~6000 basic blocks, 220 bytes of scratch per
// lane and byte-granular head / tail paths (BASELINE config 2, r03: 40 650 VALU per wave for 9 permutations, 16 % of
// the wave cycles in s_waitcnt, 1.32x the output bytes written).  When every item of a launch has the same key length,
// the same message length and the same output length, everything about the byte stream
//     head = hdr || K_i || zeros (whole rate blocks)  ||  body (message)  ||  suffix  ||  pad
// is known per LAUNCH -- which word holds which bytes, where the suffix and the pad bit go, the block counts -- so this
// kernel decides it with scalar code and leaves the vector unit the loads, one funnel shift per head word and the
// permutations.  Same SpongeParams, same reference quirks (shake()'s suffix rule, pad-only-if-unaligned:
// /root/reference/src/sha3/shake_functions.rs:24-32, sponge.rs:10-17), bit-identical output.
//
// Squeeze (/root/reference/src/sha3/sponge.rs:25-34): a digest of at most one block leaves per lane; a longer output
// (XOF: config 2 squeezes 1 KiB per item, the keystream of /root/reference/src/sha3/encryptable.rs:41) leaves as whole
// 128-byte LINES: every lane files its squeeze words into its item's 16-word row of an LDS buffer at (stream position
// mod 128), and whenever the rows are full the wave writes 64 lines with 8 store instructions of 16 bytes per lane
// (8 lanes = one line, 8 items per instruction).  A rate block written at its own offset straddles lines whose halves
// arrive a permutation apart: that was the 1.32x.
//
// Taken by launch_sponge() for uniform, 8-byte aligned digest launches of more than 128 items per SIMD (it replaces
// sponge_short.h of r02, which covered the key-less, at-most-four-blocks corner of the same idea).
#pragma once
#include "sponge_kernels.h"

namespace capy {

// the 8 stream bytes at body position `pos` (a multiple of 8) of a message of `len` bytes followed by the suffix, zeros
// and the final pad bit: the trailer branch of stream_word() with every test on wave-uniform values
__device__ __forceinline__ uint64_t uniform_tail_word(const uint8_t *msg, uint64_t pos, uint64_t len, uint64_t sfx, uint32_t sfx_len,
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

// 98..106 VGPRs, no scratch: four waves per SIMD fit; the launcher can cap the occupancy with dynamic LDS (A/B)
// SLICED (r04): the instance of the time-sliced launches (SpongeParams::sl_*; the launcher caps the occupancy at the level's
// waves per SIMD): a launch absorbs at most sl_blocks full body blocks per group of 64 items, heads in the group's first turn,
// trailer and squeeze in its last, the state in sl_state in between.  The plain instance compiles none of it.
template <int RW, bool SLICED = false>
__global__ __launch_bounds__(64, 4) void sponge_uniform_kernel(const SpongeParams p)
{
    constexpr uint32_t RB = RW * 8;
    __shared__ uint64_t s_rows[64 * RW];  // [item][word]: row stride RW words (bank-conflict-free for per-lane b64 access)

    const uint32_t lane = threadIdx.x;
    uint32_t grp = blockIdx.x, sl_t0 = 0;
    bool sl_resume = false;
    if constexpr (SLICED) {
        grp = (uint32_t)(((uint64_t)p.sl_launch * gridDim.x + blockIdx.x) % p.sl_groups);
        const uint32_t done = p.sl_done[grp];
        if (done == 0xfffffffeu) return;  // an extra turn of a finished group
        sl_resume = done != 0xffffffffu;
        sl_t0 = sl_resume ? done : 0;
    }
    const uint64_t item0 = (uint64_t)grp * 64;
    const bool active = item0 + lane < p.n;
    const uint64_t item = active ? item0 + lane : p.n - 1;  // lanes past the batch redo the last item and store nothing

    // ---- framing, all scalar (sponge_kernel's prologue with uniform inputs)
    const uint64_t len = p.absorb_body ? p.uniform_len : 0;
    uint64_t sfx = p.suffix;
    if (p.sha3_suffix_rule && (len % 136) == 135) sfx = (p.suffix & ~0xffULL) | 0x86;
    if (p.suffix_len < 8) sfx &= (1ULL << (8 * p.suffix_len)) - 1;
    const uint64_t total = len + p.suffix_len;  // after the head, which is a whole number of blocks
    const uint32_t rem = (uint32_t)(total % RB);
    const bool pad80 = p.fips_pad || rem != 0;
    const uint64_t padded = rem ? total + (RB - rem) : total;
    const uint32_t nb = (uint32_t)(padded / RB);  // body + trailer blocks
    const uint32_t nfull = (uint32_t)(len / RB);  // of which straight from the message
    const uint32_t hb = p.head_len / RB;

    KState a;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        a.lo[i] = (uint32_t)p.init_state[i];
        a.hi[i] = (uint32_t)(p.init_state[i] >> 32);
    }

    // ---- head blocks: stream byte s = hdr_len + k holds key byte k.  With K[x] the x-th aligned 8-byte word of the key
    // (zero outside [0, key_len), key_len a multiple of 8), stream word j = K[j + i0] >> 8 sh | K[j + i0 + 1] << (64 - 8 sh)
    // for the launch-wide i0 = floor(-hdr_len / 8), sh = -hdr_len mod 8; the hdr bytes are ORed into words 0 and 1.
    if (SLICED && sl_resume) {
        const uint32_t *st = p.sl_state + (size_t)grp * 50 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 25; i++) {
            a.lo[i] = st[(2 * i) * 64];
            a.hi[i] = st[(2 * i + 1) * 64];
        }
    } else if (hb) {
        const uint8_t *key = p.keys + item * p.key_stride;
        const int32_t i0 = -(int32_t)((p.hdr_len + 7) / 8);
        const uint32_t sh = (8 - (p.hdr_len & 7)) & 7;
        const int32_t kwords = (int32_t)(p.key_len / 8);
        for (uint32_t b = 0; b < hb; b++) {
            const int32_t x0 = (int32_t)(b * RW) + i0;
            uint64_t k[RW + 1];
#pragma unroll
            for (int w = 0; w <= RW; w++) {
                const int32_t x = x0 + w;
                k[w] = (x >= 0 && x < kwords) ? load_global_u64(key + 8 * x) : 0;  // uniform predicate
            }
#pragma unroll
            for (int w = 0; w < RW; w++) {
                uint64_t v = sh ? ((k[w] >> (8 * sh)) | (k[w + 1] << (64 - 8 * sh))) : k[w];
                if (b == 0 && w == 0) v |= p.hdr0;
                if (b == 0 && w == 1) v |= p.hdr1;
                xor_word(a, w, v);
            }
            keccakf1600_paired<CAPY_PAIRED_PRIO>(a);
        }
    }

    // ---- body: full blocks straight from the message (per-lane 8-byte loads: the lines a wave touches are shared by its
    // next blocks and stay in the vector cache / L2), then the trailer blocks
    const uint8_t *msg = p.msgs + (p.absorb_body ? item * p.msg_stride : 0);
    uint32_t t_end = nfull;
    if constexpr (SLICED) {
        msg += (uint64_t)sl_t0 * RB;
        if (nfull - sl_t0 > p.sl_blocks) t_end = sl_t0 + p.sl_blocks;
    }
    for (uint32_t t = sl_t0; t < t_end; t++) {
#pragma unroll
        for (int w = 0; w < RW; w++) xor_word(a, w, load_global_u64(msg + 8 * w));
        msg += RB;
        keccakf1600_paired<CAPY_PAIRED_PRIO>(a);
    }
    if constexpr (SLICED) {
        if (t_end < nfull) {  // scalar: more full blocks remain for a later launch
            uint32_t *st = p.sl_state + (size_t)grp * 50 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 25; i++) {
                st[(2 * i) * 64] = a.lo[i];
                st[(2 * i + 1) * 64] = a.hi[i];
            }
            if (lane == 0) p.sl_done[grp] = t_end;
            return;
        }
        if (lane == 0) p.sl_done[grp] = 0xfffffffeu;
    }
    for (uint32_t b = nfull; b < nb; b++) {
        const uint64_t base = (uint64_t)(b - nfull) * RB, left = len - (uint64_t)nfull * RB;
#pragma unroll
        for (int w = 0; w < RW; w++)
            xor_word(a, w, uniform_tail_word(msg, base + 8 * w, left, sfx, p.suffix_len, pad80, padded - (uint64_t)nfull * RB));
        keccakf1600_paired<CAPY_PAIRED_PRIO>(a);
    }

    // ---- squeeze
    if (p.out_len <= 8 * p.sq_words) {
        // one block: per-lane stores (rows 8-byte aligned, checked by the launcher)
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
        return;
    }
    if constexpr (RW >= 16) {
        // whole lines (sq_words == RW, out_len a multiple of 16, rows 16-byte aligned: checked by the launcher)
        uint8_t *wave_out = p.out + item0 * p.out_stride;  // SGPR pair
        const uint32_t row = lane * RW;
        const uint32_t q = lane & 7, sub = lane >> 3;
        const uint32_t items_here = p.n - item0 < 64 ? (uint32_t)(p.n - item0) : 64u;
        uint32_t goff = sub * (uint32_t)p.out_stride + 16 * q;  // store k: item 8k + lane / 8, 16-byte chunk lane % 8
        const uint32_t gstep = 8 * (uint32_t)p.out_stride;
        const uint32_t total_words = p.out_len / 8;
        uint32_t done = 0, fill = 0;
        for (;;) {
            const uint32_t nw = total_words - done < (uint32_t)RW ? total_words - done : (uint32_t)RW;
            uint32_t w0 = 0;
            while (w0 < nw) {
                const uint32_t cnt = (16 - fill) < (nw - w0) ? (16 - fill) : (nw - w0);
#pragma unroll
                for (int i = 0; i < RW; i++)
                    if ((uint32_t)i >= w0 && (uint32_t)i < w0 + cnt) s_rows[row + fill + i - w0] = state_word(a, i);
                fill += cnt;
                w0 += cnt;
                done += cnt;
                if (fill == 16 || done == total_words) {
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const uint64_t *src = &s_rows[(8 * k + sub) * RW + 2 * q];
                        const uint64_t v0 = src[0], v1 = src[1];
                        if (2 * q < fill && 8 * k + sub < items_here) {
                            typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
                            u32x4 v = {(uint32_t)v0, (uint32_t)(v0 >> 32), (uint32_t)v1, (uint32_t)(v1 >> 32)};
                            *reinterpret_cast<__attribute__((address_space(1))) u32x4 *>(
                                reinterpret_cast<uintptr_t>(wave_out + (goff + (uint32_t)k * gstep))) = v;
                        }
                    }
                    __syncthreads();
                    goff += 128;
                    fill = 0;
                }
            }
            if (done == total_words) break;
            keccakf1600_paired<CAPY_PAIRED_PRIO>(a);
        }
    }
}

}  // namespace capy
