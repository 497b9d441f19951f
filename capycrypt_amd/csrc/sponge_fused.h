// sponge_fused.h — sha3_encrypt / sha3_decrypt in ONE pass over the message, four lanes per item: batches of up to 32 items per SIMD
// (beyond: one lane per sponge, sponge_fused1.h; up to one item per SIMD: two waves per item, sponge_wide_il.h).
//
// The reference computes, per message (src/sha3/encryptable.rs:39-42 and :71-75),
//     t = kmac_xof(ka, m, 512, "SKA")          -- absorbs the whole plaintext
//     c = kmac_xof(ke, "", |m|, "SKE") XOR m   -- squeezes |m| bytes of keystream
// two independent sponges that each walk the message block by block.  Small batches are latency
// bound (one permutation after another per sponge), so this kernel runs both sponges of an item in
// lock-step on four lanes: lanes (4q, 4q+1) hold the tag sponge of item q as (lo, hi) halves, lanes
// (4q+2, 4q+3) the keystream sponge, exactly the two-lane layout of sponge_kernels_k2.h.  Every block
// of the message is fetched once (coalesced, through LDS), XORed with the keystream and absorbed into
// the tag sponge in the same step, and ONE permutation instruction stream advances both sponges:
// half the permutation latency and 2 x len of HBM traffic instead of 3 x len.
//   encrypt: tag sponge absorbs the block as loaded (plaintext), then the block is XORed and stored.
//   decrypt: the block is XORed first (-> plaintext), absorbed, stored; the launcher re-XORs items
//            whose tag does not verify (encryptable.rs:77-82) with the masked keystream kernel.
// Restricted to rate-aligned KMAC framings (D256/D384/D512: head and prefix are whole blocks) and
// 8-byte aligned messages; the launcher falls back to the two-pass path otherwise.
#pragma once
#include "sponge_kernels_k2.h"

namespace capy {

struct FusedParams {
    uint64_t init_tag[25];  // state after bytepad(encode_string("KMAC") || encode_string(ka_custom), w)
    uint64_t init_ks[25];   // same for ke_custom
    const uint8_t *keka;    // per item: ke (key_len bytes) at keka + i*keka_stride, ka at + ka_offset
    uint64_t keka_stride;
    uint32_t ka_offset;
    uint32_t key_len;
    uint32_t hdr_len;
    uint64_t hdr0, hdr1;
    uint32_t head_len;  // multiple of the rate
    uint8_t *msgs;
    const uint64_t *offsets;
    const uint64_t *lens;
    uint64_t msg_stride, uniform_len;
    uint8_t *tags;  // tag_len bytes per item at tags + i*tag_stride
    uint64_t tag_stride;
    uint32_t tag_len;
    uint32_t decrypt;
    const uint32_t *order;  // optional processing order, see SpongeParams::order
    uint32_t wide;          // launcher's choice: 1 = two waves per item (sponge_wide_il.h), 0 = four lanes per item (here)
    uint32_t staged;        // A/B (debug bit 6): the round-1 form of this kernel, blocks staged through LDS
    uint32_t paired;        // launcher's choice: more than one wave per SIMD -> the blocked round with priority
    uint64_t n;
    // time-sliced launches (r04; uniform batches between one and ~1.45 waves per SIMD, see the launcher): wave w of launch
    // `sl_launch` works on wave-group (sl_launch * gridDim.x + w) mod sl_groups for at most sl_blocks full message blocks,
    // resuming from / saving to sl_state ([group][25][64] half-words) with its progress in sl_done[group]
    // (SLICE_FRESH: not started, SLICE_FINISHED: tag written).  sl_groups == 0: the whole job in one launch.
    uint32_t sl_groups, sl_launch, sl_blocks, sl_grid;  // sl_grid: waves per launch
    uint32_t *sl_done;
    uint32_t *sl_state;
    // the one-lane-per-sponge form (sponge_fused1.h, r05: two lanes per item, chip-filling batches): launcher's choice of the
    // instance (0: not this form; 1 lone wave per SIMD, 2 two waves, 4 three or four), A/B switch for per-lane stores
    // instead of whole lines, occupancy cap in waves per SIMD (0: none), and the rotating-occupancy schedule's phase
    // (sponge_fused1.h: fused1_rot_kernel; rot_G == 0: not that schedule)
    uint32_t one_lane, direct_stores, lone_direct, cap_waves, store_policy;
    uint32_t rot_phase, rot_Cp, rot_G, rot_nb1, rot_nb2;
};
constexpr uint32_t SLICE_FRESH = 0xffffffffu, SLICE_FINISHED = 0xfffffffeu;

// partner sponge's word: lanes (4q + 2, 4q + 3) <-> (4q, 4q + 1)
__device__ __forceinline__ uint32_t quad_swap_pairs(uint32_t v)
{
    // quad_perm:[2,3,0,1] -> dpp_ctrl = 2 | 3<<2 | 0<<4 | 1<<6 = 0x4E
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
}

// STAGED = false (default): every lane loads its own 32-bit half of every word of its item's block (the tag lane and
// the keystream lane of a half load the same word in the same instruction, so the tag lane always sees the bytes as
// they were before this step), the keystream lanes store the XORed half straight back, and on decrypt the tag lanes
// take the keystream from their partner lanes with one DPP move per word.  No LDS and no barrier in the block loop.
// STAGED = true: the round-1 form (wave-cooperative 8-byte transfers through LDS, four barriers per block), kept for A/B.
// PAIRED (r03): the launch puts two or three waves on a SIMD (16 384 < n <= 49 152): the hot loop runs the blocked round
// with raised priority around its DPP / rotation blocks (sponge_kernels_k2.h: keccak_round_k2_blocked).
// PAIRED: 0 = one wave per SIMD (and compiled for exactly one); 1 = two waves (22 528 < n <= 32 768: the blocked round with priority)
template <int RW, bool STAGED = false, int PAIRED = 0>
__global__ __launch_bounds__(64) CAPY_WAVES_PER_SIMD(PAIRED == 0 ? 1 : 8) void sponge_fused_crypt_kernel(const FusedParams fp)
{
    constexpr uint32_t RB = RW * 8;
    constexpr int NIT = 16;                       // items per wave
    constexpr int NLOAD = (NIT * RW + 63) / 64;   // 8-byte cooperative transfers per block step
    __shared__ uint64_t s_stage[NLOAD * 64];
    __shared__ uint64_t s_base[NIT];
    __shared__ uint32_t s_nfull[NIT];

    const uint32_t lane = threadIdx.x;
    const uint32_t h = lane & 1, role = (lane >> 1) & 1, q = lane >> 2;  // role 0 = tag sponge, 1 = keystream sponge
    const uint32_t hmask = 0u - h;
    const uint32_t grp = fp.sl_groups ? (uint32_t)(((uint64_t)fp.sl_launch * gridDim.x + blockIdx.x) % fp.sl_groups) : blockIdx.x;
    const uint64_t slot = (uint64_t)grp * NIT + q;
    uint32_t sl_t0 = 0;  // first full block of this launch (wave-uniform)
    bool sl_resume = false;
    if (fp.sl_groups) {
        const uint32_t done = fp.sl_done[grp];
        if (done == SLICE_FINISHED) return;  // an extra turn of a group that has its tag already
        sl_resume = done != SLICE_FRESH;
        sl_t0 = sl_resume ? done : 0;
    }
    const bool active = slot < fp.n;
    const uint64_t item = active ? (fp.order ? (uint64_t)fp.order[slot] : slot) : fp.n;

    // Both sponges are described with the generic stream machinery (sponge_params.h): the tag sponge absorbs
    // head || msg || 00 01 04 || pad, the keystream sponge absorbs head || 00 01 04 || pad.
    SpongeParams p;
    p.pre = nullptr;
    p.pre_len = 0;
    p.key_offsets = nullptr;
    p.key_len = fp.key_len;
    p.hdr_len = fp.hdr_len;
    p.hdr0 = fp.hdr0;
    p.hdr1 = fp.hdr1;
    p.head_len = fp.head_len;
    p.suffix = 0x040100ULL;
    p.suffix_len = 3;
    p.fips_pad = 0;
    p.stride_bytes = RB;

    ItemCtx c;
    c.key = nullptr;
    c.msg = nullptr;
    uint64_t tgt_len = 0;
    if (active) {
        if (fp.offsets) {
            const uint64_t o0 = fp.offsets[item];
            tgt_len = fp.lens ? fp.lens[item] : fp.offsets[item + 1] - o0;
            c.msg = fp.msgs + o0;
        } else {
            tgt_len = fp.uniform_len;
            c.msg = fp.msgs + item * fp.msg_stride;
        }
        c.key = fp.keka + item * fp.keka_stride + (role == 0 ? fp.ka_offset : 0);
    }
    // ke / ka are derived keys of one fixed length: the head is the same for every item
    c.key_len = fp.key_len;
    c.hdr_len = fp.hdr_len;
    c.hdr0 = fp.hdr0;
    c.hdr1 = fp.hdr1;
    c.head_len = fp.head_len;
    c.len = role == 0 ? tgt_len : 0;
    c.suffix = p.suffix;
    const uint64_t total = (uint64_t)fp.head_len + c.len + 3;
    const uint32_t rem = (uint32_t)(total % RB);
    c.pad80 = rem != 0;
    c.padded = rem ? total + (RB - rem) : total;
    const uint32_t hb = fp.head_len / RB;
    const uint32_t nfull = active ? (uint32_t)(tgt_len / RB) : 0;  // full message blocks (both roles walk them)

    KHalf a;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        const uint64_t v = role == 0 ? fp.init_tag[i] : fp.init_ks[i];
        a.a[i] = h ? (uint32_t)(v >> 32) : (uint32_t)v;
    }

    auto absorb_slow = [&](uint64_t base) {
#pragma unroll
        for (int w = 0; w < RW; w++) {
            const uint64_t v = stream_word(p, c, base + 8 * w);
            a.a[w] ^= h ? (uint32_t)(v >> 32) : (uint32_t)v;
        }
        keccakf1600_k2(a, hmask);
    };

    // ---- heads (both roles), then the keystream sponge's only other block (00 01 04 || pad)
    if (sl_resume) {
        const uint32_t *st = fp.sl_state + (size_t)grp * 25 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 25; i++) a.a[i] = st[i * 64];
    } else {
        for (uint32_t b = 0; b < hb; b++)
            if (active) absorb_slow((uint64_t)b * RB);
        if (active && role == 1) absorb_slow((uint64_t)hb * RB);
    }
    // from here on the keystream sponge's state IS keystream block 0

    // ---- full blocks, one pass
    uint32_t *stage32 = reinterpret_cast<uint32_t *>(s_stage);
    if constexpr (!STAGED) {
        const uint32_t all_full = wave_max_u32(nfull);
        // a time slice ends after sl_blocks blocks (uniform lengths: every lane of the batch has the same nfull)
        const uint32_t max_full = (fp.sl_groups && all_full - sl_t0 > fp.sl_blocks) ? sl_t0 + fp.sl_blocks : all_full;
        if (max_full > sl_t0) {
            const uint8_t *last_word = batch_last_word(fp.msgs, fp.offsets, fp.n, fp.msg_stride, fp.uniform_len);
            uint8_t *mine = (active ? const_cast<uint8_t *>(c.msg) : fp.msgs) + 4 * h;
            uint32_t pf[RW];
            auto own_load = [&](uint32_t t) {
                const bool live = t < nfull;
                const uint8_t *src = live ? mine + (uint64_t)t * RB : last_word;
#pragma unroll
                for (int w = 0; w < RW; w++)
                    pf[w] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(
                        reinterpret_cast<uintptr_t>(live ? src + 8 * w : src));
            };
            own_load(sl_t0);
            for (uint32_t t = sl_t0; t < max_full; t++) {
                const bool live = t < nfull;
                uint32_t wv[RW];
                if (fp.decrypt) {  // wave-uniform; the DPP move runs in every lane
#pragma unroll
                    for (int w = 0; w < RW; w++) wv[w] = pf[w] ^ quad_swap_pairs(a.a[w]);  // tag lanes: plaintext
                } else {
#pragma unroll
                    for (int w = 0; w < RW; w++) wv[w] = pf[w];
                }
                if (live && role == 1) {
                    uint8_t *dstp = mine + (uint64_t)t * RB;
#pragma unroll
                    for (int w = 0; w < RW; w++)
                        *reinterpret_cast<__attribute__((address_space(1))) uint32_t *>(reinterpret_cast<uintptr_t>(dstp + 8 * w)) =
                            pf[w] ^ a.a[w];
                }
                if (t + 1 < max_full) own_load(t + 1);
                // tag sponge: absorb + permute; keystream sponge: permute while more keystream is needed
                const bool perm = live && (role == 0 || (uint64_t)(t + 1) * RB < tgt_len);
                if (perm) {
                    if (role == 0) {
#pragma unroll
                        for (int w = 0; w < RW; w++) a.a[w] ^= wv[w];
                    }
                    if constexpr (PAIRED)
                        keccakf1600_k2_paired_unrolled<true>(a, hmask);
                    else
                        keccakf1600_k2_unrolled(a, hmask);
                }
            }
        }
        if (fp.sl_groups && max_full < all_full) {  // wave-uniform: more full blocks remain for a later launch
            uint32_t *st = fp.sl_state + (size_t)grp * 25 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 25; i++) st[i * 64] = a.a[i];
            if (lane == 0) fp.sl_done[grp] = max_full;
            return;
        }
    } else {
    if (lane < NIT) {
        const uint64_t sl = (uint64_t)blockIdx.x * NIT + lane;
        const uint64_t it = sl < fp.n ? (fp.order ? (uint64_t)fp.order[sl] : sl) : fp.n;
        uint64_t base = (uint64_t)(uintptr_t)fp.msgs;  // slots past the batch: any mapped address
        uint32_t nf = 0;
        if (it < fp.n) {
            uint64_t len;
            if (fp.offsets) {
                const uint64_t o0 = fp.offsets[it];
                len = fp.lens ? fp.lens[it] : fp.offsets[it + 1] - o0;
                base = (uint64_t)(uintptr_t)(fp.msgs + o0);
            } else {
                len = fp.uniform_len;
                base = (uint64_t)(uintptr_t)(fp.msgs + it * fp.msg_stride);
            }
            nf = (uint32_t)(len / RB);
        }
        s_base[lane] = base;
        s_nfull[lane] = nf;
    }
    __syncthreads();
    const uint32_t max_full = wave_max_u32(nfull);
    if (max_full) {
        uint8_t *dst[NLOAD];
        uint32_t lim[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; k++) {
            const uint32_t i = k * 64 + lane;
            const uint32_t m = i / RW, w = i - m * RW;
            const bool in = m < NIT;
            lim[k] = in ? s_nfull[in ? m : 0] : 0;
            dst[k] = in ? reinterpret_cast<uint8_t *>(s_base[m]) + 8 * w : fp.msgs;
        }
        const uint8_t *last_word = batch_last_word(fp.msgs, fp.offsets, fp.n, fp.msg_stride, fp.uniform_len);
        uint64_t pf[NLOAD];
        auto coop_load = [&](uint32_t t) {
#pragma unroll
            for (int k = 0; k < NLOAD; k++) {
                pf[k] = load_global_u64(ragged_src(t < lim[k], dst[k], (uint64_t)t * RB, last_word));
            }
        };
        coop_load(0);
        for (uint32_t t = 0; t < max_full; t++) {
#pragma unroll
            for (int k = 0; k < NLOAD; k++) s_stage[k * 64 + lane] = pf[k];
            __syncthreads();
            const bool mine = t < nfull;
            uint32_t wv[RW];
            if (!fp.decrypt) {
                // tag sponge reads the plaintext before the keystream lanes overwrite it
#pragma unroll
                for (int w = 0; w < RW; w++) wv[w] = stage32[(q * RW + w) * 2 + h];
                __syncthreads();
                if (mine && role == 1) {
#pragma unroll
                    for (int w = 0; w < RW; w++) stage32[(q * RW + w) * 2 + h] ^= a.a[w];
                }
                __syncthreads();
            } else {
                if (mine && role == 1) {
#pragma unroll
                    for (int w = 0; w < RW; w++) stage32[(q * RW + w) * 2 + h] ^= a.a[w];
                }
                __syncthreads();
#pragma unroll
                for (int w = 0; w < RW; w++) wv[w] = stage32[(q * RW + w) * 2 + h];
                __syncthreads();
            }
#pragma unroll
            for (int k = 0; k < NLOAD; k++) {
                const uint64_t v = s_stage[k * 64 + lane];
                if (t < lim[k]) store_global_u64(dst[k] + (uint64_t)t * RB, v);
            }
            __syncthreads();
            if (t + 1 < max_full) coop_load(t + 1);
            // tag sponge: absorb + permute; keystream sponge: permute while more keystream is needed
            const bool perm = mine && (role == 0 || (uint64_t)(t + 1) * RB < tgt_len);
            if (perm) {
                if (role == 0) {
#pragma unroll
                    for (int w = 0; w < RW; w++) a.a[w] ^= wv[w];
                }
                keccakf1600_k2_unrolled(a, hmask);
            }
        }
    }

    }

    // ---- tail: fewer than RB message bytes remain.  The keystream lanes read them (once), XOR, store, and hand the
    // PLAINTEXT half-words to the tag lanes through LDS, so no lane ever re-reads bytes another lane has just written.
    const uint64_t pos = (uint64_t)nfull * RB;
    const uint32_t left = active ? (uint32_t)(tgt_len - pos) : 0;
    uint8_t *m = const_cast<uint8_t *>(c.msg);
    __syncthreads();
    if (role == 1) {
#pragma unroll
        for (int w = 0; w < RW; w++) {
            const uint32_t at = 8 * w + 4 * h;
            uint32_t in = 0;
            for (int b = 0; b < 4; b++)
                if (at + b < left) in |= (uint32_t)m[pos + at + b] << (8 * b);
            const uint32_t nvalid = at < left ? (left - at < 4 ? left - at : 4) : 0;
            const uint32_t vmask = nvalid >= 4 ? 0xffffffffu : ((1u << (8 * nvalid)) - 1u);
            const uint32_t out = (in ^ a.a[w]) & vmask;
            for (int b = 0; b < 4; b++)
                if (at + b < left) m[pos + at + b] = (uint8_t)(out >> (8 * b));
            stage32[(q * RW + w) * 2 + h] = fp.decrypt ? out : in;  // plaintext
        }
    }
    __syncthreads();
    {
        // tag sponge: plaintext tail || 00 01 04 || 0* [80]   (1 or 2 blocks)
        const uint32_t tl = left + 3;
        const uint32_t cnt = (role == 0 && active) ? (tl + RB - 1) / RB : 0;
        const bool pad80 = (tl % RB) != 0;
        const uint32_t max_cnt = wave_max_u32(cnt);
        for (uint32_t j = 0; j < max_cnt; j++) {
            if (j < cnt) {
#pragma unroll
                for (int w = 0; w < RW; w++) {
                    const uint32_t rel0 = j * RB + 8 * w + 4 * h;
                    uint32_t v = j == 0 ? stage32[(q * RW + w) * 2 + h] : 0u;
                    for (int b = 0; b < 4; b++) {
                        const uint32_t rel = rel0 + b;
                        uint32_t byte = 0;
                        if (rel >= left && rel - left < 3) byte = (0x040100u >> (8 * (rel - left))) & 0xff;
                        if (pad80 && rel + 1 == cnt * RB) byte |= 0x80;
                        v |= byte << (8 * b);
                    }
                    a.a[w] ^= v;
                }
                keccakf1600_k2(a, hmask);
            }
        }
    }

    if (fp.sl_groups && lane == 0) fp.sl_done[grp] = SLICE_FINISHED;
    // ---- tag
    if (role == 0 && active) {
        uint8_t *o = fp.tags + item * fp.tag_stride;
#pragma unroll
        for (int w = 0; w < RW; w++) {
            const uint32_t at = 8 * w + 4 * h;
            if (at + 4 <= fp.tag_len) {
                const uint32_t v = a.a[w];
                if ((((uintptr_t)(o + at)) & 3) == 0)
                    *reinterpret_cast<uint32_t *>(o + at) = v;
                else
                    for (int b = 0; b < 4; b++) o[at + b] = (uint8_t)(v >> (8 * b));
            }
        }
    }
}

}  // namespace capy
