// sponge_fused1.h — sha3_encrypt / sha3_decrypt in ONE pass over the message for CHIP-FILLING batches: one lane per sponge.
//
// The reference computes, per message (/root/reference/src/sha3/encryptable.rs:29-45 and :58-83),
//     t = kmac_xof(ka, m, 512, "SKA")          -- absorbs the whole plaintext
//     c = kmac_xof(ke, "", |m|, "SKE") XOR m   -- squeezes |m| bytes of keystream
// sponge_fused.h runs both sponges of an item on four lanes (two two-lane pairs): 240 lane-instructions per sponge-round,
// right while a batch cannot fill the chip (a two-lane sponge has the shorter chain) and pure loss once it can: from 32 768
// items on there are >= 65 536 sponges, one per lane fills every SIMD, and the one-lane round costs 180.  Here an item takes
// TWO lanes: the even lane holds the whole tag sponge (25 x u64 in 50 VGPRs), the odd lane the whole keystream sponge, and
// both run the one-lane round of keccak_dev.h in lock-step -- 32 items per wave, 4320 VALU per 136-byte block for both
// sponges of 32 items (the four-lane form: 2880 for 16).
//   * both lanes of a pair load the same 8-byte words of the block (one request per line: HBM sees the message once);
//   * the keystream lane XORs them with its rate words -- ciphertext on encrypt, plaintext on decrypt;
//   * the tag lane absorbs the plaintext: the words as loaded (encrypt) or XORed with the partner's rate words, which cross
//     with two v_mov_b32_dpp quad_perm:[1,0,3,2] per word (decrypt) -- never through memory;
//   * the XORed words do NOT go back per lane (32 partial-line stores per instruction: 1.16x the bytes written and 18 % of
//     the time in the kernels that tried, profiles/r02_direct_loads_ab.txt): the keystream lane files them into its item's
//     256-byte ring in LDS at (address mod 256), and whenever a 128-byte line of the message is complete the wave writes it
//     with 16-byte stores, 8 lanes per line, 8 items per store instruction -- whole lines only, traffic 2 x len.
// Restricted like sponge_fused.h to rate-aligned KMAC framings (D256 / D384 / D512) and 8-byte aligned messages; keys are
// the derived ke || ka of the protocol layers (a multiple of 8 bytes, 8-byte aligned).  Decrypt restores the ciphertext of
// items whose tag does not verify with the masked keystream pass of the launcher (encryptable.rs:77-82).
//
// FORM: 1 = a lone wave per SIMD (plain unrolled round, next block prefetched; the lone role of the rotating schedule)
//       2 = two waves per SIMD (blocked round with raised priority, unrolled, next block prefetched; <= 256 VGPRs)
//       3 = the doubled-up role of the rotating-occupancy schedule (two waves per SIMD: the blocked round ROLLED -- two compute
//           units share an instruction cache, and both roles unrolled evict each other, sponge_rot.h --, next block prefetched)
//       4 = three or four waves per SIMD (the same round rolled, blocks loaded at the top of the step; <= 128 VGPRs)
#pragma once
#include "sponge_fused.h"
#include "sponge_kernels.h"

namespace capy {

#ifndef CAPY_F1_ROT_DOUBLED
#define CAPY_F1_ROT_DOUBLED 3  // 2: the doubled-up role unrolled as well (A/B)
#endif
#ifndef CAPY_F1_FORM2_WAVES
#define CAPY_F1_FORM2_WAVES 2  // 3: the unrolled instance squeezed into 168 VGPRs (A/B)
#endif
#ifndef CAPY_F1_LB
#define CAPY_F1_LB 4
#endif
constexpr int FUSED1_ITEMS = 32;                          // items per wave
constexpr uint32_t FUSED1_RING = 256;                     // bytes per item: two lines
constexpr uint32_t FUSED1_LDS_WAVE = FUSED1_ITEMS * FUSED1_RING + FUSED1_ITEMS * 16 + FUSED1_ITEMS * 4;  // ring + record + block limit per item

typedef uint32_t __attribute__((ext_vector_type(4))) fused1_u32x4;
typedef __attribute__((address_space(3))) uint8_t fused1_lds_u8;
typedef __attribute__((address_space(3))) uint64_t fused1_lds_u64;
typedef __attribute__((address_space(3))) fused1_u32x4 fused1_lds_u32x4;

// The ring and the records are private to ONE wave, and the LDS serves a wave's instructions in order: a read sees what the
// same wave wrote before it.  No s_barrier (the rotating schedule puts eight waves with different trip counts in a
// workgroup); this only keeps the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void fused1_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int FORM>
__device__ __forceinline__ void fused1_hot(KState &a)
{
    if constexpr (FORM == 1)
        keccakf1600_unrolled(a);
    else if constexpr (FORM == 2)
        keccakf1600_paired_unrolled<CAPY_PAIRED_PRIO>(a);
    else  // 3, 4
        keccakf1600_paired<CAPY_PAIRED_PRIO>(a);
}
template <int FORM>
__device__ __forceinline__ void fused1_cold(KState &a)
{
    if constexpr (FORM == 1)
        keccakf1600(a);
    else
        keccakf1600_paired<CAPY_PAIRED_PRIO>(a);
}

// what the tag sponge absorbs after the message: 00 01 (right_encode(0), shake_functions.rs:86) 04 (cSHAKE suffix, :57),
// zeros, and 0x80 in the last byte of the last block when the stream is not a whole number of blocks (sponge.rs:13).
// The 8 bytes at offset `rel` of the stream that starts right behind the last full block; `left` message bytes precede.
__device__ __forceinline__ uint64_t fused1_trailer_word(uint32_t rel, uint32_t left, bool pad80, uint32_t padded)
{
    const int32_t d = (int32_t)left - (int32_t)rel;  // where the three bytes start, relative to this word
    uint64_t v = 0;
    if (d >= 0 && d < 8) v = 0x040100ULL << (8 * d);
    if (d < 0 && d > -3) v = 0x040100ULL >> (8 * -d);
    if (pad80 && rel + 8 == padded) v |= 0x80ULL << 56;
    return v;
}

// The store side of an item lives in LDS, not in registers (the rolled round at four waves per SIMD has none to spare across
// the permutation): one 16-byte record per item = {endp: the address behind the last block filed, avail: the bytes the ring
// holds, counted from the first byte of the line the next store pass writes (that line starts at endp - avail), front: the
// bytes of that line in front of the range (never written)}, kept by the keystream lane.
//
// One pass of line stores.  FINAL = false: items whose ring holds a complete line (avail >= 128) write it; FINAL = true: every
// item writes what it still holds (the last, partial line of its range).  Lane l serves the 16-byte chunk l % 8 of item
// 8 k + l / 8 for k = 0 .. 3, straight from that item's record.  Returns false (wave-uniform) when no item had anything.
// REGS (the instances with registers to spare: FORM 2, 3): the owner lanes keep their record in registers (`mine`) and publish it
// here; otherwise (FORM 4) it lives in LDS only and `mine` is a scratch copy.
template <bool FINAL, bool ROLLED, bool REGS>
__device__ __forceinline__ bool fused1_flush_pass(fused1_u32x4 &mine, uint32_t role, uint32_t q, uint32_t lane, fused1_lds_u8 *ring,
                                                  fused1_lds_u32x4 *home, uint32_t policy)
{
    if constexpr (!REGS) mine = home[q];
    const bool has = FINAL ? mine.z > mine.w : mine.z >= 128;
    if (__builtin_amdgcn_ballot_w64(has) == 0) return false;
    if constexpr (REGS) {
        if (role == 1) home[q] = mine;
        fused1_wave_sync();
    }
    const uint32_t c16 = 16 * (lane & 7);
    auto serve = [&](uint32_t k) {
        const uint32_t j = 8 * k + (lane >> 3);
        const fused1_u32x4 r = home[j];
        const bool hasj = FINAL ? r.z > r.w : r.z >= 128;
        const uint32_t limit = r.z < 128 ? r.z : 128;
        if (hasj && c16 >= r.w && c16 + 16 <= limit) {
            const uint64_t gl = (((uint64_t)r.y << 32) | r.x) - r.z;
            const uint32_t ro = (((uint32_t)gl & 255) + 16 * j + c16) & 255;
            const fused1_u32x4 v = *reinterpret_cast<const fused1_lds_u32x4 *>(ring + j * FUSED1_RING + ro);
            // policy (wave-uniform; A/B, CAPY_DEBUG=fused1_store): 0 plain stores -- the written lines stay in the XCD's L2 --,
            // 1 sc1 (written through and dropped from L2), 2 nt
            const uint64_t dst = gl + c16;
            if (policy == 1)
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
            else if (policy == 2)
                asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
            else
                *reinterpret_cast<__attribute__((address_space(1))) fused1_u32x4 *>(dst) = v;
        }
    };
    if constexpr (ROLLED) {  // the 128-register instance: one item at a time; the other waves of the SIMD cover the LDS round trips
#pragma unroll 1
        for (uint32_t k = 0; k < 4; k++) serve(k);
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) serve(k);
    }
    fused1_wave_sync();
    if (has) {
        mine.z -= mine.z < 128 ? mine.z : 128;
        mine.w = 0;
        if constexpr (!REGS) {
            if (role == 1) home[q] = mine;
        }
    }
    if constexpr (!REGS) fused1_wave_sync();
    return true;
}

// An item's message of this wave-group slot: both lanes of pair q work on item order[32 grp + q].  `lane` comes through an
// empty asm where the caller wants the values RE-computed rather than kept in registers across the block loop.
struct Fused1Item {
    uint64_t item, tgt_len;
    uint8_t *msg;
    bool active;
};
__device__ __forceinline__ Fused1Item fused1_item(const FusedParams &fp, uint32_t grp, uint32_t lane)
{
    Fused1Item it;
    const uint64_t slot = (uint64_t)grp * FUSED1_ITEMS + (lane >> 1);
    it.active = slot < fp.n;
    it.item = it.active ? (fp.order ? (uint64_t)fp.order[slot] : slot) : 0;
    it.msg = fp.msgs;
    it.tgt_len = 0;
    if (it.active) {
        if (fp.offsets) {
            const uint64_t o0 = fp.offsets[it.item];
            it.tgt_len = fp.lens ? fp.lens[it.item] : fp.offsets[it.item + 1] - o0;
            it.msg = fp.msgs + o0;
        } else {
            it.tgt_len = fp.uniform_len;
            it.msg = fp.msgs + it.item * fp.msg_stride;
        }
    }
    return it;
}

// Full message blocks [t0, min(nfull, t0 + tcap)) of wave-group `grp` (32 items), preceded by the heads when `fresh` and
// followed by tail + tag when no full block remains afterwards; otherwise the states go to fp.sl_state for a later launch.
// lds: FUSED1_LDS_WAVE bytes of this wave's own.  Returns the first full block that is left (wave-uniform), or 0xffffffff
// when the group is finished.
template <int RW, int FORM, bool DECRYPT>
__device__ __forceinline__ uint32_t fused1_body(const FusedParams &fp, uint32_t grp, uint32_t t0, uint32_t tcap, bool fresh, fused1_lds_u8 *lds)
{
    constexpr uint32_t RB = RW * 8;
    fused1_lds_u8 *ring = lds;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t hb = fp.head_len / RB;

    KState a;
    if (fresh) {
        const Fused1Item it = fused1_item(fp, grp, lane);
        const uint32_t role = lane & 1;  // 0 = tag sponge, 1 = keystream sponge
#pragma unroll
        for (int i = 0; i < 25; i++) {
            const uint64_t v = role == 0 ? fp.init_tag[i] : fp.init_ks[i];
            a.lo[i] = (uint32_t)v;
            a.hi[i] = (uint32_t)(v >> 32);
        }
        // ---- heads bytepad(encode_string(K), w) = hdr || K || zeros, whole blocks: stream byte hdr_len + k holds key byte k.
        // With K[x] the x-th aligned word of the key (zero outside it), stream word j = K[j + i0] >> 8 sh | K[j + i0 + 1] << (64 - 8 sh)
        // for the launch-wide i0 = floor(-hdr_len / 8), sh = -hdr_len mod 8 (sponge_uniform.h); the tag lane reads ka, its partner ke
        const uint8_t *key = fp.keka + it.item * fp.keka_stride + (role == 0 ? fp.ka_offset : 0);
        const int32_t i0 = -(int32_t)((fp.hdr_len + 7) / 8);
        const uint32_t sh = (8 - (fp.hdr_len & 7)) & 7;
        const int32_t kwords = (int32_t)(fp.key_len / 8);
        for (uint32_t b = 0; b < hb; b++) {
            const int32_t x0 = (int32_t)(b * RW) + i0;
            uint64_t kprev = (x0 >= 0 && x0 < kwords) ? load_global_u64(key + 8 * x0) : 0;  // uniform predicates
#pragma unroll
            for (int w = 0; w < RW; w++) {
                const int32_t x = x0 + w + 1;
                const uint64_t knext = (x >= 0 && x < kwords) ? load_global_u64(key + 8 * x) : 0;
                uint64_t v = sh ? ((kprev >> (8 * sh)) | (knext << (64 - 8 * sh))) : kprev;
                if (b == 0 && w == 0) v |= fp.hdr0;
                if (b == 0 && w == 1) v |= fp.hdr1;
                xor_word(a, w, v);
                kprev = knext;
            }
            fused1_cold<FORM>(a);
        }
        // the keystream sponge's only other block: 00 01 04 || 0* || 80 (X = "", encryptable.rs:41)
        if (role == 1) {
            a.lo[0] ^= 0x040100u;
            a.hi[RW - 1] ^= 0x80000000u;
            fused1_cold<FORM>(a);
        }
    } else {
        const uint32_t *st = fp.sl_state + (size_t)grp * 50 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 25; i++) {
            a.lo[i] = st[(2 * i) * 64];
            a.hi[i] = st[(2 * i + 1) * 64];
        }
    }
    // from here on the keystream sponge's state IS the keystream block of the step

    // ---- full blocks.  Nothing but the state is live across the permutation: the store side sits in LDS (above), the rest
    // is wave-uniform or recomputed from the lane number.
    uint32_t all_full, t_end;
    {
        fused1_lds_u32x4 *home = reinterpret_cast<fused1_lds_u32x4 *>(lds + FUSED1_ITEMS * FUSED1_RING);
        __attribute__((address_space(3))) uint32_t *myend =
            reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(lds + FUSED1_ITEMS * (FUSED1_RING + 16));
        const uint32_t role = lane & 1, q = lane >> 1;
        {
            const Fused1Item it = fused1_item(fp, grp, lane);
            const uint32_t nfull = (uint32_t)(it.tgt_len / RB);  // full message blocks (both roles walk them)
            all_full = wave_max_u32(nfull);
            t_end = (all_full > t0 && all_full - t0 > tcap) ? t0 + tcap : all_full;  // wave-uniform
            const uint32_t my_end = nfull < t_end ? nfull : t_end;
            const uint64_t endp = (uint64_t)(uintptr_t)(it.msg + (uint64_t)t0 * RB);
            const uint32_t front = my_end > t0 ? (uint32_t)(endp & 127) : 0;
            if (role == 1) {
                const fused1_u32x4 r = {(uint32_t)endp, (uint32_t)(endp >> 32), front, front};
                home[q] = r;
                myend[q] = my_end;
            }
            fused1_wave_sync();
        }
        if (t_end > t0) {
            // fp.lone_direct: FORM 1 (a lone wave per SIMD: the lone role of the rotating schedule) stores per lane instead.  A lone
            // wave pays every LDS round trip and every instruction of the ring in full (474 against 520 GiB/s,
            // profiles/r05_fused_one_lane.txt), and at one wave per SIMD most partial lines of a block's two stores still meet in L2
            // (bytes written 1.08-1.16 x; at two waves per SIMD it is 1.42 x, which is what the ring is for).  The launcher sets it
            // where lone waves dominate the schedule (below 44 items per SIMD: +5-8 % there, nothing above).
            // fp.direct_stores: the same for every instance (A/B).
            constexpr bool ALWAYS_DIRECT = false;
            constexpr bool REGS = FORM != 4;  // the store side of an item in registers (and published to LDS for the store passes)
            const bool direct = fp.direct_stores || (FORM == 1 && fp.lone_direct);  // wave-uniform
            uint64_t pf[RW];
#pragma unroll
            for (int w = 0; w < RW; w++) pf[w] = 0;
            // one base register + immediate offsets; lanes whose message has run out load nothing (one exec region per step)
            auto load_block = [&](const uint8_t *at) {
#pragma unroll
                for (int w = 0; w < RW; w++) pf[w] = load_global_u64(at + 8 * w);
            };
            fused1_u32x4 r = home[q];
            uint32_t my_end = myend[q];
            if constexpr (FORM != 4) {
                if (t0 < my_end) load_block(reinterpret_cast<const uint8_t *>(((uint64_t)r.y << 32) | r.x));
            }
            // The ring holds two lines: what is left of a line after a store pass (at most 120 bytes) plus the words filed since
            // must not exceed 256 bytes.  A 136-byte block fits in one go; the 152- and 168-byte blocks of D384 / D256 are filed
            // in two parts with a store pass in between (the first 128 bytes complete a line whatever was left).
            constexpr int W1 = (RB + 120 > FUSED1_RING) ? 16 : RW;
            auto advance = [&](uint32_t bytes) {
                const uint64_t e2 = (((uint64_t)r.y << 32) | r.x) + bytes;
                r.x = (uint32_t)e2;
                r.y = (uint32_t)(e2 >> 32);
                r.z += bytes;
            };
            for (uint32_t t = t0; t < t_end; t++) {
                if constexpr (!REGS) {
                    r = home[q];
                    my_end = myend[q];
                }
                const bool live = t < my_end;
                uint8_t *blk = reinterpret_cast<uint8_t *>(((uint64_t)r.y << 32) | r.x);
                if (live) {
                    if constexpr (FORM == 4) load_block(blk);
                    // the tag sponge absorbs the plaintext: the words as loaded, or XORed with the partner's rate words (the
                    // keystream lane's state does not change here: its mask is 0)
                    const uint32_t am = role == 0 ? 0xffffffffu : 0u;
                    if constexpr (DECRYPT) {  // both lanes of a pair are live together
#pragma unroll
                        for (int w = 0; w < RW; w++) {
                            const uint32_t pl = dpp_swap_pair(a.lo[w]), ph = dpp_swap_pair(a.hi[w]);
                            a.lo[w] ^= ((uint32_t)pf[w] ^ pl) & am;
                            a.hi[w] ^= ((uint32_t)(pf[w] >> 32) ^ ph) & am;
                        }
                    } else {
#pragma unroll
                        for (int w = 0; w < RW; w++) {
                            a.lo[w] ^= (uint32_t)pf[w] & am;
                            a.hi[w] ^= (uint32_t)(pf[w] >> 32) & am;
                        }
                    }
                    if (role == 1) {
                        if (direct) {
#pragma unroll
                            for (int w = 0; w < RW; w++) store_global_u64(blk + 8 * w, pf[w] ^ state_word(a, w));
                        } else if constexpr (!ALWAYS_DIRECT) {
                            // per-item rotation of the ring by 16 q: the 8 items of a store group hit different banks
                            uint32_t rp = (r.x + 16 * q) & 255;
                            fused1_lds_u8 *row = ring + q * FUSED1_RING;
#pragma unroll
                            for (int w = 0; w < W1; w++) {
                                *reinterpret_cast<fused1_lds_u64 *>(row + rp) = pf[w] ^ state_word(a, w);
                                rp = (rp + 8) & 255;
                            }
                            // a 16-byte chunk that straddles the first / last byte of the range is not written by the store passes:
                            // the keystream lane stores that word itself (RB = 8 mod 16: a block's last word ends where its first
                            // began, mod 16)
                            if (t == t0 && (r.x & 8)) store_global_u64(blk, pf[0] ^ state_word(a, 0));
                            if (t + 1 == my_end && !(r.x & 8)) store_global_u64(blk + 8 * (RW - 1), pf[RW - 1] ^ state_word(a, RW - 1));
                        }
                    }
                    // (both lanes of a pair advance their copy; only the keystream lane's is ever published)
                    advance(direct ? RB : 8 * W1);
                    if constexpr (!REGS) {
                        if (role == 1) home[q] = r;
                    }
                }
                if constexpr (!ALWAYS_DIRECT && W1 < RW) {
                    if (!direct) {  // wave-uniform
                        fused1_wave_sync();
                        while (fused1_flush_pass<false, FORM == 4, REGS>(r, role, q, lane, ring, home, fp.store_policy)) {
                        }
                        if (live) {
                            if (role == 1) {
                                uint32_t rp = (r.x + 16 * q) & 255;
                                fused1_lds_u8 *row = ring + q * FUSED1_RING;
#pragma unroll
                                for (int w = W1; w < RW; w++) {
                                    *reinterpret_cast<fused1_lds_u64 *>(row + rp) = pf[w] ^ state_word(a, w);
                                    rp = (rp + 8) & 255;
                                }
                            }
                            advance(8 * (RW - W1));
                            if constexpr (!REGS) {
                                if (role == 1) home[q] = r;
                            }
                        }
                    }
                }
                if constexpr (FORM != 4) {
                    if (t + 1 < my_end) load_block(blk + RB);
                }
                if constexpr (!ALWAYS_DIRECT) {
                    if (!direct) {  // wave-uniform
                        fused1_wave_sync();
                        while (fused1_flush_pass<false, FORM == 4, REGS>(r, role, q, lane, ring, home, fp.store_policy)) {
                        }
                    }
                }
                if (live) fused1_hot<FORM>(a);
            }
            if constexpr (!ALWAYS_DIRECT) {
                if (!direct) {
                    fused1_wave_sync();
                    while (fused1_flush_pass<true, true, REGS>(r, role, q, lane, ring, home, fp.store_policy)) {
                    }
                }
            }
        }
    }
    if (t_end < all_full) {  // wave-uniform: more full blocks remain for a later launch
        uint32_t *st = fp.sl_state + (size_t)grp * 50 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 25; i++) {
            st[(2 * i) * 64] = a.lo[i];
            st[(2 * i + 1) * 64] = a.hi[i];
        }
        return t_end;
    }

    // ---- tail: fewer than RB message bytes remain.  Both lanes of a pair load them (before either stores); the keystream lane
    // XORs and stores, the tag lane forms the plaintext from the same bytes and its partner's rate words.
    uint32_t lane2 = lane;
    asm volatile("" : "+v"(lane2));  // recompute the item's values: nothing of them stays in registers across the block loop
    const Fused1Item it = fused1_item(fp, grp, lane2);
    const uint32_t role = lane2 & 1;
    const uint64_t pos = it.tgt_len / RB * RB;
    const uint32_t left_all = (uint32_t)(it.tgt_len - pos);
    const uint32_t tl = left_all + 3;
    const uint32_t cnt = it.active ? (tl + RB - 1) / RB : 0;  // 1 or 2 blocks: tail || 00 01 04 || 0* [80]
    const bool pad80 = (tl % RB) != 0;
    const uint32_t max_cnt = wave_max_u32(cnt);
#pragma unroll 1
    for (uint32_t j = 0; j < max_cnt; j++) {
        const bool mine = j < cnt;
        // `left` goes through an empty asm in every trip: the RW byte masks below would otherwise be hoisted out of this loop of
        // one or two trips as 2 RW loop-invariant registers, which the 128-register instance spilled (r05: 23-62 VGPRs of
        // scratch in FORM 4; tests/test_kernel_resources.py now holds every instance of this kernel to zero)
        uint32_t left = left_all;
        asm volatile("" : "+v"(left));
#pragma unroll
        for (int w = 0; w < RW; w++) {
            const uint32_t at = 8 * w;
            uint64_t v = fused1_trailer_word(j * RB + at, left, pad80, cnt * RB);
            const uint64_t ks = state_word(a, w);
            const uint64_t pks = ((uint64_t)dpp_swap_pair(a.hi[w]) << 32) | dpp_swap_pair(a.lo[w]);  // the partner's rate word
            if (j == 0 && at < left) {
                const uint32_t nvalid = left - at < 8 ? left - at : 8;
                const uint64_t vmask = nvalid >= 8 ? ~0ULL : ((1ULL << (8 * nvalid)) - 1);
                uint8_t *p = it.msg + pos + at;
                const uint64_t in = load_global_u64(p) & vmask;  // stays inside the aligned word that holds the last byte
                if (role == 1) {
                    const uint64_t out = (in ^ ks) & vmask;
                    if (nvalid >= 8)
                        store_global_u64(p, out);
                    else
                        for (uint32_t b = 0; b < nvalid; b++) p[b] = (uint8_t)(out >> (8 * b));
                }
                v |= DECRYPT ? (in ^ pks) & vmask : in;
            }
            if (mine && role == 0) xor_word(a, w, v);
        }
        if (mine && role == 0) fused1_cold<FORM>(a);
    }

    // ---- tag
    if (role == 0 && it.active) {
        uint8_t *o = fp.tags + it.item * fp.tag_stride;
#pragma unroll
        for (int w = 0; w < RW; w++) {
            const uint32_t at = 8 * w;
            if (at + 8 <= fp.tag_len && (((uintptr_t)(o + at)) & 7) == 0) {
                store_global_u64(o + at, state_word(a, w));
            } else {
                const uint64_t v = state_word(a, w);
                for (uint32_t b = 0; b < 8; b++)
                    if (at + b < fp.tag_len) o[at + b] = (uint8_t)(v >> (8 * b));
            }
        }
    }
    return 0xffffffffu;
}

// The rotating-OCCUPANCY schedule of sponge_rot.h for this kernel: batches between one and two waves per SIMD (32 S < n < 64 S
// items).  One launch of the whole batch takes the two-waves time however few SIMDs hold a second wave; time slices of one
// wave per SIMD run every wave at the lone-wave rate, which for this kernel is 0.75 of the two-waves throughput.  Here the
// wave-groups are bundled in fours (= one compute unit at one wave per SIMD, 128 items); in each of P phase launches Cp compute
// units hold TWO bundles -- two waves per SIMD, rot_nb2 blocks each -- and the others one bundle, rot_nb1 = ratio x rot_nb2
// blocks; roles rotate (bundle g has role (g + phase 2 Cp) mod G), every bundle is doubled up in the same number of phases, all
// waves of a phase finish together.  A workgroup is 512 lanes and the kernel is compiled for exactly two waves per SIMD, so a
// compute unit holds exactly one workgroup; in the lone role waves 4..7 leave at once.  Progress per wave-group in sl_done,
// states in sl_state, as for the time slices; what the phases leave (a few blocks, tail, tag) is done by one sliced launch of
// sponge_fused1_kernel.
template <int RW, bool DECRYPT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void sponge_fused1_rot_kernel(const FusedParams fp)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[8 * FUSED1_LDS_WAVE];
    const uint32_t b = blockIdx.x, wave = threadIdx.x >> 6;
    const bool doubled = b < fp.rot_Cp;  // workgroup-uniform
    if (!doubled && wave >= 4) return;
    const uint32_t rho = doubled ? 2 * b + (wave >> 2) : fp.rot_Cp + b;
    const uint32_t shift = (uint32_t)(((uint64_t)fp.rot_phase * 2 * fp.rot_Cp) % fp.rot_G);
    const uint32_t g = rho >= shift ? rho - shift : rho + fp.rot_G - shift;
    const uint32_t grp = g * 4 + (wave & 3);
    if ((uint64_t)grp * FUSED1_ITEMS >= fp.n) return;  // wave-uniform: an empty bundle / the idle part of the last one
    const uint32_t done = fp.sl_done[grp];
    if (done == SLICE_FINISHED) return;
    const bool fresh = done == SLICE_FRESH;
    fused1_lds_u8 *lds = (fused1_lds_u8 *)s_lds + wave * FUSED1_LDS_WAVE;
    uint32_t next;
    if (doubled)
        next = fused1_body<RW, CAPY_F1_ROT_DOUBLED, DECRYPT>(fp, grp, fresh ? 0 : done, fp.rot_nb2, fresh, lds);
    else
        next = fused1_body<RW, 1, DECRYPT>(fp, grp, fresh ? 0 : done, fp.rot_nb1, fresh, lds);
    if ((threadIdx.x & 63) == 0) fp.sl_done[grp] = next == 0xffffffffu ? SLICE_FINISHED : next;
}

// One launch: wave w works on wave-group w (sl_groups == 0), or -- time slices, as in sponge_fused.h -- on wave-group
// (sl_launch * gridDim.x + w) mod sl_groups for at most sl_blocks full blocks, progress in sl_done, states in sl_state.
template <int RW, int FORM, bool DECRYPT>
// Occupancy is pinned from both sides by amdgpu_waves_per_eu(min, max) alone (a second __launch_bounds__ argument next to
// it left the descriptor at the registers used: 160 for FORM 1 / 2, so that three waves fitted a SIMD -- ADVICE r5): min caps
// the registers the compiler may use, max pads the count in the kernel descriptor so that one more wave of THIS kernel does not
// fit.  FORM 1: exactly one wave per SIMD (>= 257 VGPRs in the descriptor), FORM 2: exactly two (>= 171), FORM 4: four (<= 128).
__global__ __launch_bounds__(64)
    __attribute__((amdgpu_waves_per_eu((FORM == 1 ? 1 : (FORM == 2 ? CAPY_F1_FORM2_WAVES : CAPY_F1_LB)),
                                       (FORM == 1 ? 1 : (FORM == 2 ? CAPY_F1_FORM2_WAVES : 4))))) void
    sponge_fused1_kernel(const FusedParams fp)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[FUSED1_LDS_WAVE];
    fused1_lds_u8 *lds = (fused1_lds_u8 *)s_lds;
    uint32_t grp = blockIdx.x, t0 = 0, tcap = 0xffffffffu;
    bool fresh = true;
    if (fp.sl_groups) {
        grp = (uint32_t)(((uint64_t)fp.sl_launch * gridDim.x + blockIdx.x) % fp.sl_groups);
        const uint32_t done = fp.sl_done[grp];
        if (done == SLICE_FINISHED) return;  // an extra turn of a group that has its tag already
        fresh = done == SLICE_FRESH;
        t0 = fresh ? 0 : done;
        tcap = fp.sl_blocks;
    }
    const uint32_t next = fused1_body<RW, FORM, DECRYPT>(fp, grp, t0, tcap, fresh, lds);
    if (fp.sl_groups && (threadIdx.x & 63) == 0) fp.sl_done[grp] = next == 0xffffffffu ? SLICE_FINISHED : next;
}

}  // namespace capy
