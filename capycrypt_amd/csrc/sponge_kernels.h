// sponge_kernels.h — batched keccak sponge for gfx950: one sponge per lane.
//
// One kernel template covers the reference's whole sponge path
//   shake / cshake / kmac_xof      /root/reference/src/sha3/shake_functions.rs:24-89
//   sponge_absorb / sponge_squeeze /root/reference/src/sha3/sponge.rs:10-34
//   xor_bytes (keystream ^ msg)    /root/reference/src/sha3/aux_functions.rs:90-93
// by treating every call as the absorption of a per-item byte stream
//     head (bytepad(encode_string(K_i), w), built in-kernel)  ||  body (message)  ||  suffix  ||  pad
// on top of an initial state that already contains the batch-shared prefix block(s)
// (bytepad(encode_string(N) || encode_string(S), w)), followed by a squeeze that either writes
// `out_len` bytes per item (MODE 0) or XORs a len_i-byte keystream into the message in place (MODE 1).
//
// Data movement: each lane owns one sponge (25 x u64 in 50 VGPRs), but messages are contiguous per item,
// so rate-sized blocks are fetched by the whole wave with 8-byte loads (17..21 consecutive lanes cover one
// message block: slot k, lane l -> element i = 64k + l -> (item, word) = (i / RW, i % RW)), staged through
// LDS in [item][word] order (linear in i, so the writes are contiguous; the per-lane ds_read_b64 at stride
// RW*8 is bank-conflict-free) and, when the chip is not full, prefetched one block ahead in registers.
// For uniformly strided batches the loads use one wave-uniform base (SGPR pair) plus a fixed 32-bit
// per-lane offset, so the block loop spends no VALU on addressing.
//
// Template parameters
//   RW        absorb rate in 64-bit words (9, 13, 17, 18, 19, 21)
//   FULLCHIP  false: latency-tuned instance (few waves: unrolled permutation, register prefetch);
//             true : issue-tuned instance (many waves per SIMD: rolled permutation, small register budget)
//   MODE      0 digest output, 1 in-place keystream XOR
#pragma once
#include "sponge_params.h"


namespace capy {

// Measured on MI355X (profiles/r01_keccak_loop_forms.txt): a lone wave hides nothing, so with <= 1 wave per SIMD
// the fully unrolled permutation with literal round constants wins (905 vs 734..825 GB/s-equivalent at 768
// waves); with many waves per SIMD the rolled form with constants fetched one trip ahead wins (10.7 vs 9.1 G
// permutations/s at 16k waves).  Since r03 the many-waves form is the BLOCKED round with raised priority around its
// rotation blocks (keccak_dev.h: keccak_round_blocked): two waves of a SIMD then share it at 2 cycles per simple
// instruction, 13.9 instead of 9.9 G permutations/s (profiles/r03_valu_issue_bisect.txt).
// PAIRED (latency-tuned instance only): the launch puts two waves on a SIMD (64 < items per SIMD <= 128, ragged
// batches of any size): the unrolled permutation on the blocked round.
template <bool FULLCHIP, bool PAIRED = false>
__device__ __forceinline__ void keccak_hot(KState &a)
{
    if constexpr (FULLCHIP)
        keccakf1600_paired<CAPY_PAIRED_PRIO>(a);
    else if constexpr (PAIRED)
        keccakf1600_paired_unrolled<CAPY_PAIRED_PRIO>(a);
    else
        keccakf1600_unrolled(a);
}
template <bool FULLCHIP, bool PAIRED = false>
__device__ __forceinline__ void keccak_cold(KState &a)
{
    if constexpr (FULLCHIP || PAIRED)
        keccakf1600_paired<CAPY_PAIRED_PRIO>(a);
    else
        keccakf1600(a);
}

__device__ __forceinline__ void xor_word(KState &a, int w, uint64_t v)
{
    a.lo[w] ^= (uint32_t)v;
    a.hi[w] ^= (uint32_t)(v >> 32);
}
__device__ __forceinline__ uint64_t state_word(const KState &a, int w) { return ((uint64_t)a.hi[w] << 32) | a.lo[w]; }

// register budget: the latency-tuned instance must still fit two waves per SIMD (it serves up to 128 items per SIMD);
// the issue-tuned instance is held to 168 VGPRs for three.
// r03: three waves (168 VGPRs) since the blocked round pairs the waves of a SIMD: +1.4 / +4.8 / +7 % over four waves at
// 128 VGPRs on 2^18 x 512 KiB / 2^20 x 4 KiB / 2^21 x 1 KiB, config 2 +4 % (profiles/r03_chipfull.txt)
#ifndef CAPY_FULLCHIP_WAVES
#define CAPY_FULLCHIP_WAVES 3
#endif
// 1: the issue-tuned instance also keeps the next block in registers under the permutation (A/B, r03: spills at four
// waves (0.70x), -3 % at three)
#ifndef CAPY_FULLCHIP_PREFETCH
#define CAPY_FULLCHIP_PREFETCH 0
#endif
// WAVES = waves per SIMD the register budget is sized for (2: latency-tuned, 256 VGPRs; 3: issue-tuned, 168 VGPRs).
// A 3-wave copy of the issue-tuned instance for ragged batches was tried and dropped: the latency-tuned instance is
// faster there (see launch_sponge in sponge.hip).
// WAVES = 1 (the latency-tuned instance, taken for at most one wave per SIMD): compiled so that a second wave of the SAME kernel
// does not fit on the SIMD (CAPY_WAVES_PER_SIMD below) -- behind a launch whose waves end staggered the dispatcher otherwise
// doubles waves up on the SIMDs that happen to be free and the launch takes up to twice as long (profiles/r04_placement.txt).
template <int RW, bool FULLCHIP, int MODE, int WAVES = (FULLCHIP ? CAPY_FULLCHIP_WAVES : 1), bool PAIRED = false>
// (the register cap and the pin both through amdgpu_waves_per_eu(min, max): a second __launch_bounds__ argument beside the
// attribute left the WAVES = 1 instances of the small rates, SHA3-384 / SHA3-512, at the 216 / 248 registers they use -- two
// waves fitted; tests/test_kernel_resources.py reads the kernel descriptors)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES == 1 ? 1 : 8))) void sponge_kernel(const SpongeParams p)
{
    constexpr uint32_t RB = RW * 8;
    __shared__ uint64_t s_stage[64 * RW];
    __shared__ uint64_t s_base[64];
    __shared__ uint32_t s_nfull[64];

    const uint32_t lane = threadIdx.x;
    const uint64_t item0 = (uint64_t)blockIdx.x * 64;
    const bool in_range = item0 + lane < p.n;
    const uint64_t item = in_range ? (p.order ? (uint64_t)p.order[item0 + lane] : item0 + lane) : p.n;
    const bool active = in_range && (p.mask == nullptr || p.mask[item] != 0);

    ItemCtx c;
    c.key = nullptr;
    c.msg = nullptr;
    uint64_t tgt_len = 0;  // length of the per-item message buffer (absorb body and/or xor target)
    if (active) {
        if (p.offsets) {
            uint64_t o0 = p.offsets[item];
            tgt_len = p.lens ? p.lens[item] : p.offsets[item + 1] - o0;
            c.msg = p.msgs + o0;
        } else {
            tgt_len = p.uniform_len;
            c.msg = p.msgs + item * p.msg_stride;
        }
    }
    item_head(p, item, active, c);
    c.len = p.absorb_body ? tgt_len : 0;
    c.suffix = p.suffix;
    if (p.sha3_suffix_rule && (c.len % 136) == 135) c.suffix = (p.suffix & ~0xffULL) | 0x86;
    const uint64_t total = (uint64_t)p.pre_len + c.head_len + c.len + p.suffix_len;
    const uint32_t rem = (uint32_t)(total % p.stride_bytes);
    c.pad80 = p.fips_pad || rem != 0;
    c.padded = rem ? total + (p.stride_bytes - rem) : total;
    const uint32_t nb = active ? (uint32_t)(c.padded / p.stride_bytes) : 0;

    const bool grid_aligned = ((p.pre_len + (p.key_offsets ? 0u : p.head_len)) % RB == 0) && (p.stride_bytes == RB);  // wave-uniform
    // head blocks: wave-uniform unless the keys have per-item lengths (then a multiple of w = RB per lane)
    const uint32_t hb = grid_aligned ? (p.pre_len + c.head_len) / RB : 0;
    const uint32_t hb_max = p.key_offsets ? wave_max_u32(hb) : hb;
    const bool msg_aligned = active && grid_aligned && (((uintptr_t)c.msg & 7) == 0);
    const uint32_t nfull = (msg_aligned && p.absorb_body) ? (uint32_t)(c.len / RB) : 0;

    // Wave-uniform fast addressing: a full wave of equally long, equally strided, 8-byte aligned messages whose
    // 64 blocks sit within 4 GiB of the wave's first message.
    const bool uniform = !(p.debug_flags & 1) && p.offsets == nullptr && p.mask == nullptr && p.order == nullptr && item0 + 64 <= p.n &&
                         grid_aligned &&
                         (((uintptr_t)p.msgs | p.msg_stride) & 7) == 0 && p.msg_stride * 64 < 0xfff00000ULL;
    const uint8_t *wave_base = p.msgs + item0 * p.msg_stride;  // SGPR pair

    KState a;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        a.lo[i] = (uint32_t)p.init_state[i];
        a.hi[i] = (uint32_t)(p.init_state[i] >> 32);
    }
    const bool resume = p.resume_state != nullptr;  // wave-uniform; phases H and B already done elsewhere
    if (resume && active) {
#pragma unroll
        for (int i = 0; i < 25; i++) {
            const uint64_t v = p.resume_state[(uint64_t)i * p.resume_pad + item];
            a.lo[i] = (uint32_t)v;
            a.hi[i] = (uint32_t)(v >> 32);
        }
    }

    // ---------------- phase H: per-item head blocks (uniform trip count)
    for (uint32_t b = 0; b < (resume ? 0u : hb_max); b++) {
        if (active && b < hb) {
#pragma unroll
            for (int w = 0; w < RW; w++) xor_word(a, w, stream_word(p, c, (uint64_t)b * RB + 8 * w));
            keccak_cold<FULLCHIP, PAIRED>(a);
        }
    }

    if (p.head_state != nullptr) {  // wave-uniform
        if (active) {
#pragma unroll
            for (int i = 0; i < 25; i++) p.head_state[(uint64_t)i * p.resume_pad + item] = state_word(a, i);
        }
        return;
    }

    // per-lane element offsets of the cooperative transfers (uniform path): element i = 64k + lane
    auto elem_offset = [&](int k) -> uint32_t {
        const uint32_t i = k * 64 + lane;
        const uint32_t m = i / RW, w = i - m * RW;
        return m * (uint32_t)p.msg_stride + 8 * w;
    };

    // ---------------- phase B: full body blocks, wave-cooperative loads through LDS
    if (resume) {
        // nothing: blocks [0, resume_blocks) are in the loaded state
    } else if (uniform) {
        const uint32_t nf = p.absorb_body ? (uint32_t)(p.uniform_len / RB) : 0;  // same for every lane
        if (nf) {
            uint32_t voff[RW];
#pragma unroll
            for (int k = 0; k < RW; k++) voff[k] = elem_offset(k);
            if (p.debug_flags & SPONGE_DIRECT_LOADS) {
                // every lane loads the words of its own block, next block in flight under the rounds -- no LDS, no
                // barrier.  The lines a wave's 64 blocks touch do not all stay in the CU's 32 KB vector cache at 2-4
                // waves per SIMD, but L2 absorbs the re-fetches: +2-4 % uniform, +6-27 % ragged over the
                // wave-cooperative loads through LDS (profiles/r02_direct_loads_ab.txt)
                const uint8_t *mine = wave_base + (uint64_t)lane * p.msg_stride;
                if constexpr (FULLCHIP && !CAPY_FULLCHIP_PREFETCH) {
                    // 128-VGPR budget: no registers held across the permutation, the other waves of the SIMD cover the loads
                    for (uint32_t t = 0; t < nf; t++) {
#pragma unroll
                        for (int w = 0; w < RW; w++) xor_word(a, w, load_global_u64(mine + 8 * w));
                        mine += RB;
                        keccak_hot<FULLCHIP, PAIRED>(a);
                    }
                } else {
                uint64_t pf[RW];
#pragma unroll
                for (int w = 0; w < RW; w++) pf[w] = load_global_u64(mine + 8 * w);
                for (uint32_t t = 0; t < nf; t++) {
#pragma unroll
                    for (int w = 0; w < RW; w++) xor_word(a, w, pf[w]);
                    if (t + 1 < nf) {
                        mine += RB;
#pragma unroll
                        for (int w = 0; w < RW; w++) pf[w] = load_global_u64(mine + 8 * w);
                    }
                    keccak_hot<FULLCHIP, PAIRED>(a);
                }
                }
            } else if constexpr (!FULLCHIP) {
                uint64_t pf[RW];
#pragma unroll
                for (int k = 0; k < RW; k++) pf[k] = *reinterpret_cast<const uint64_t *>(wave_base + voff[k]);
                for (uint32_t t = 0; t < nf; t++) {
#pragma unroll
                    for (int k = 0; k < RW; k++) s_stage[k * 64 + lane] = pf[k];
                    __syncthreads();
                    uint64_t wv[RW];
#pragma unroll
                    for (int w = 0; w < RW; w++) wv[w] = s_stage[lane * RW + w];
                    __syncthreads();
                    if (t + 1 < nf) {
                        const uint8_t *bt = wave_base + (uint64_t)(t + 1) * RB;
#pragma unroll
                        for (int k = 0; k < RW; k++) pf[k] = *reinterpret_cast<const uint64_t *>(bt + voff[k]);
                    }
#pragma unroll
                    for (int w = 0; w < RW; w++) xor_word(a, w, wv[w]);
                    keccak_hot<FULLCHIP, PAIRED>(a);
                }
            } else {
                for (uint32_t t = 0; t < nf; t++) {
                    const uint8_t *bt = wave_base + (uint64_t)t * RB;
#pragma unroll
                    for (int k = 0; k < RW; k++) s_stage[k * 64 + lane] = *reinterpret_cast<const uint64_t *>(bt + voff[k]);
                    __syncthreads();
#pragma unroll
                    for (int w = 0; w < RW; w++) xor_word(a, w, s_stage[lane * RW + w]);
                    __syncthreads();
                    keccak_hot<FULLCHIP, PAIRED>(a);
                }
            }
        }
    } else {
        s_base[lane] = (uint64_t)(uintptr_t)(c.msg ? c.msg : p.msgs);
        s_nfull[lane] = nfull;
        __syncthreads();
        const uint32_t max_full = wave_max_u32(nfull);
        const uint8_t *last_word = batch_last_word(p.msgs, p.offsets, p.n, p.msg_stride, p.uniform_len);
        if (!FULLCHIP && (p.debug_flags & SPONGE_DIRECT_LOADS)) {  // ragged batches never take the issue-tuned instance
            if (max_full) {
                const uint8_t *mine = c.msg ? c.msg : p.msgs;
                uint64_t pf[RW];
                auto own_load = [&](uint32_t t) {
                    const bool live = t < nfull;
                    const uint8_t *src = live ? mine + (uint64_t)t * RB : last_word;
#pragma unroll
                    for (int w = 0; w < RW; w++) pf[w] = load_global_u64(live ? src + 8 * w : src);
                };
                own_load(0);
                for (uint32_t t = 0; t < max_full; t++) {
                    const bool live = t < nfull;
                    if (live) {
#pragma unroll
                        for (int w = 0; w < RW; w++) xor_word(a, w, pf[w]);
                    }
                    if (t + 1 < max_full) own_load(t + 1);
                    if (live) keccak_hot<FULLCHIP, PAIRED>(a);
                }
            }
        } else if constexpr (!FULLCHIP) {
            if (max_full) {
                // source pointer and block limit of every (slot, lane) pair, hoisted out of the block loop
                const uint8_t *src[RW];
                uint32_t lim[RW];
#pragma unroll
                for (int k = 0; k < RW; k++) {
                    const uint32_t i = k * 64 + lane;
                    const uint32_t m = i / RW, w = i - m * RW;
                    lim[k] = s_nfull[m];
                    src[k] = reinterpret_cast<const uint8_t *>(s_base[m]) + 8 * w;
                }
                uint64_t pf[RW];
                auto coop_load = [&](uint32_t t) {
#pragma unroll
                    for (int k = 0; k < RW; k++)
                        pf[k] = load_global_u64(ragged_src(t < lim[k], src[k], (uint64_t)t * RB, last_word));
                };
                coop_load(0);
                for (uint32_t t = 0; t < max_full; t++) {
#pragma unroll
                    for (int k = 0; k < RW; k++) s_stage[k * 64 + lane] = pf[k];
                    __syncthreads();
                    uint64_t wv[RW];
#pragma unroll
                    for (int w = 0; w < RW; w++) wv[w] = s_stage[lane * RW + w];
                    __syncthreads();
                    if (t + 1 < max_full) coop_load(t + 1);
                    if (t < nfull) {
#pragma unroll
                        for (int w = 0; w < RW; w++) xor_word(a, w, wv[w]);
                        keccak_hot<FULLCHIP, PAIRED>(a);
                    }
                }
            }
        } else {
            for (uint32_t t = 0; t < max_full; t++) {
#pragma unroll
                for (int k = 0; k < RW; k++) {
                    const uint32_t i = k * 64 + lane;
                    const uint32_t m = i / RW, w = i - m * RW;
                    s_stage[i] = load_global_u64(ragged_src(
                        t < s_nfull[m], reinterpret_cast<const uint8_t *>(s_base[m]) + 8 * w, (uint64_t)t * RB, last_word));
                }
                __syncthreads();
                if (t < nfull) {
#pragma unroll
                    for (int w = 0; w < RW; w++) xor_word(a, w, s_stage[lane * RW + w]);
                }
                __syncthreads();
                if (t < nfull) keccak_hot<FULLCHIP, PAIRED>(a);
            }
        }
    }

    // ---------------- phase T: remaining blocks (tail of the body, suffix, pad) byte-granular
    {
        const uint32_t first = resume ? p.resume_blocks : hb + nfull;
        const uint32_t cnt = nb > first ? nb - first : 0;
        const uint32_t max_cnt = wave_max_u32(cnt);
        for (uint32_t j = 0; j < max_cnt; j++) {
            if (j < cnt) {
                const uint64_t base = (uint64_t)(first + j) * RB;
#pragma unroll
                for (int w = 0; w < RW; w++) xor_word(a, w, stream_word(p, c, base + 8 * w));
                keccak_cold<FULLCHIP, PAIRED>(a);
            }
        }
    }

    // ---------------- squeeze
    if constexpr (MODE == 0) {
        uint8_t *o = active ? p.out + item * p.out_stride : nullptr;
        uint32_t produced = 0;
        // Long outputs (XOF squeezes of at least one rate block per item): whole blocks leave through LDS so that 17..21
        // consecutive lanes write one item's 136..168 contiguous bytes, instead of every lane writing 8 bytes to its own
        // row (config 2: 1.37x write amplification, 64-byte fabric writes).  Needs a full wave of in-order items, rows
        // 8-byte aligned and the squeeze width equal to the rate (cSHAKE / KMAC).
        const bool coop_out = p.sq_words == (uint32_t)RW && p.out_len >= RB && item0 + 64 <= p.n && p.order == nullptr &&
                              p.mask == nullptr && (((uintptr_t)p.out | p.out_stride) & 7) == 0 && p.out_stride * 64 < 0xfff00000ULL;
        // r04: whole 128-byte LINES leave, not rate blocks.  A 136..168-byte block written at its own offset straddles
        // lines, and the two halves of a straddled line are written a permutation apart: 1.32x the output bytes reached
        // HBM (config 2, profiles/r02_config2_pmc.txt).  Here every lane files its squeeze words into its item's row of
        // the staging buffer at (stream position mod 128); whenever the rows hold a full line the wave writes 64 lines
        // with 8 store instructions of 16 bytes per lane (8 lanes = one line, 8 items per instruction).  The row index
        // and the emit decision are wave-uniform (equal out_len), so this costs scalar branches only.
        const bool line_out = RW >= 16 && coop_out && !(p.debug_flags & SPONGE_BLOCK_OUT) && p.out_len >= 128 && (p.out_len & 15) == 0 &&
                              (((uintptr_t)p.out | p.out_stride) & 15) == 0;
        if (line_out) {
            uint8_t *wave_out = p.out + item0 * p.out_stride;  // SGPR pair
            const uint32_t row = lane * RW;                     // this lane's row of s_stage (RW words; 16 are used)
            const uint32_t q = lane & 7;
            uint32_t goff = (lane >> 3) * (uint32_t)p.out_stride + 16 * q;  // store k: item 8k + lane/8, chunk lane%8
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
                        if ((uint32_t)i >= w0 && (uint32_t)i < w0 + cnt) s_stage[row + fill + i - w0] = state_word(a, i);
                    fill += cnt;
                    w0 += cnt;
                    done += cnt;
                    if (fill == 16 || done == total_words) {
                        __syncthreads();
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const uint64_t *src = &s_stage[(8 * k + (lane >> 3)) * RW + 2 * q];
                            const uint64_t v0 = src[0], v1 = src[1];
                            if (2 * q < fill) {
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
                keccak_hot<FULLCHIP, PAIRED>(a);
            }
            produced = p.out_len;
        } else if (coop_out) {
            uint32_t ooff[RW];
#pragma unroll
            for (int k = 0; k < RW; k++) {
                const uint32_t i = k * 64 + lane;
                const uint32_t m = i / RW, w = i - m * RW;
                ooff[k] = m * (uint32_t)p.out_stride + 8 * w;
            }
            uint8_t *wave_out = p.out + item0 * p.out_stride;  // SGPR pair
            const uint32_t nblk = p.out_len / RB;
            for (uint32_t t = 0; t < nblk; t++) {
#pragma unroll
                for (int w = 0; w < RW; w++) s_stage[lane * RW + w] = state_word(a, w);
                __syncthreads();
                uint8_t *bt = wave_out + (uint64_t)t * RB;
#pragma unroll
                for (int k = 0; k < RW; k++) *reinterpret_cast<uint64_t *>(bt + ooff[k]) = s_stage[k * 64 + lane];
                __syncthreads();
                produced += RB;
                if (produced < p.out_len) keccak_hot<FULLCHIP, PAIRED>(a);
            }
        }
        while (produced < p.out_len) {
#pragma unroll
            for (int w = 0; w < 25; w++) {
                if ((uint32_t)w < p.sq_words) {
                    if (active && produced < p.out_len) {
                        const uint64_t v = state_word(a, w);
                        if (produced + 8 <= p.out_len && (((uintptr_t)(o + produced)) & 7) == 0) {
                            *reinterpret_cast<uint64_t *>(o + produced) = v;
                        } else {
                            for (uint32_t j = 0; j < 8 && produced + j < p.out_len; j++) o[produced + j] = (uint8_t)(v >> (8 * j));
                        }
                    }
                    produced += 8;
                }
            }
            // the reference permutes once more after the last block (sponge.rs:30); that state is
            // never observable, so the permutation is skipped here.
            if (produced < p.out_len) keccak_hot<FULLCHIP, PAIRED>(a);
        }
    } else {
        // keystream XOR in place: msg[i] ^= squeeze(len); squeeze block = RW words (cSHAKE/KMAC only)
        uint32_t xfull = msg_aligned ? (uint32_t)(tgt_len / RB) : 0;
        if (!FULLCHIP && (p.debug_flags & SPONGE_DIRECT_LOADS)) {
            // every lane XORs its own message in place with 8-byte loads and stores, next block in flight.  Not for the
            // issue-tuned instance: at 4 waves per SIMD the per-lane partial-line stores cost 18 % (429 -> 351 GiB/s at
            // 262 144 x 64 KiB, profiles/r02_direct_loads_ab.txt) where the staged, coalesced stores do not.
            if (uniform) xfull = (uint32_t)(p.uniform_len / RB);
            uint8_t *mine = const_cast<uint8_t *>(c.msg ? c.msg : p.msgs);
            const uint32_t max_x = wave_max_u32(xfull);
            if (max_x) {
                const uint8_t *last_word = batch_last_word(p.msgs, p.offsets, p.n, p.msg_stride, p.uniform_len);
                uint64_t pf[RW];
                auto own_load = [&](uint32_t t) {
                    const bool live = t < xfull;
                    const uint8_t *src = live ? mine + (uint64_t)t * RB : last_word;
#pragma unroll
                    for (int w = 0; w < RW; w++) pf[w] = load_global_u64(live ? src + 8 * w : src);
                };
                own_load(0);
                for (uint32_t t = 0; t < max_x; t++) {
                    const bool live = t < xfull;
                    if (live) {
                        uint8_t *bt = mine + (uint64_t)t * RB;
#pragma unroll
                        for (int w = 0; w < RW; w++) store_global_u64(bt + 8 * w, pf[w] ^ state_word(a, w));
                    }
                    if (t + 1 < max_x) own_load(t + 1);
                    if (live && (uint64_t)(t + 1) * RB < tgt_len) keccak_hot<FULLCHIP, PAIRED>(a);
                }
            }
        } else if (uniform) {
            const uint32_t nx = (uint32_t)(p.uniform_len / RB);
            xfull = nx;
            if (nx) {
                uint32_t voff[RW];
#pragma unroll
                for (int k = 0; k < RW; k++) voff[k] = elem_offset(k);
                uint8_t *wb = const_cast<uint8_t *>(wave_base);
                for (uint32_t t = 0; t < nx; t++) {
                    uint8_t *bt = wb + (uint64_t)t * RB;
#pragma unroll
                    for (int k = 0; k < RW; k++) s_stage[k * 64 + lane] = *reinterpret_cast<const uint64_t *>(bt + voff[k]);
                    __syncthreads();
#pragma unroll
                    for (int w = 0; w < RW; w++) s_stage[lane * RW + w] ^= state_word(a, w);
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < RW; k++) *reinterpret_cast<uint64_t *>(bt + voff[k]) = s_stage[k * 64 + lane];
                    __syncthreads();
                    if ((uint64_t)(t + 1) * RB < tgt_len) keccak_hot<FULLCHIP, PAIRED>(a);
                }
            }
        } else {
            __syncthreads();
            s_base[lane] = (uint64_t)(uintptr_t)(c.msg ? c.msg : p.msgs);
            s_nfull[lane] = xfull;
            const uint8_t *last_word = batch_last_word(p.msgs, p.offsets, p.n, p.msg_stride, p.uniform_len);
            __syncthreads();
            const uint32_t max_x = wave_max_u32(xfull);
            for (uint32_t t = 0; t < max_x; t++) {
#pragma unroll
                for (int k = 0; k < RW; k++) {
                    const uint32_t i = k * 64 + lane;
                    const uint32_t m = i / RW, w = i - m * RW;
                    s_stage[i] = load_global_u64(ragged_src(
                        t < s_nfull[m], reinterpret_cast<const uint8_t *>(s_base[m]) + 8 * w, (uint64_t)t * RB, last_word));
                }
                __syncthreads();
                if (t < xfull) {
#pragma unroll
                    for (int w = 0; w < RW; w++) s_stage[lane * RW + w] ^= state_word(a, w);
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < RW; k++) {
                    const uint32_t i = k * 64 + lane;
                    const uint32_t m = i / RW, w = i - m * RW;
                    if (t < s_nfull[m])
                        store_global_u64(reinterpret_cast<uint8_t *>(s_base[m]) + (uint64_t)t * RB + 8 * w, s_stage[i]);
                }
                __syncthreads();
                if (t < xfull && (uint64_t)(t + 1) * RB < tgt_len) keccak_hot<FULLCHIP, PAIRED>(a);
            }
        }
        // leftover bytes (unaligned messages: everything) byte-granular
        uint64_t pos = (uint64_t)xfull * RB;
        const uint64_t left = tgt_len - pos;
        const uint32_t cnt = (uint32_t)((left + RB - 1) / RB);
        const uint32_t max_cnt = wave_max_u32(cnt);
        uint8_t *m = const_cast<uint8_t *>(c.msg);
        for (uint32_t j = 0; j < max_cnt; j++) {
            if (j < cnt) {
#pragma unroll
                for (int w = 0; w < RW; w++) {
                    const uint64_t v = state_word(a, w);
                    for (int b = 0; b < 8; b++)
                        if (pos + 8 * w + b < tgt_len) m[pos + 8 * w + b] ^= (uint8_t)(v >> (8 * b));
                }
                pos += RB;
                if (pos < tgt_len) keccak_cold<FULLCHIP, PAIRED>(a);
            }
        }
    }
}

}  // namespace capy
