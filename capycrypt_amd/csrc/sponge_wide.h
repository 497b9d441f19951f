// sponge_wide.h — sha3_encrypt / sha3_decrypt for VERY small batches of long messages: one WAVE per item.
//
// BASELINE config 3 as specified leaves 128 messages of 5 MiB per GPU: 256 sponges (tag + keystream), each a strict
// chain of ~38 553 permutations, on a chip with 1024 SIMDs.  Nothing but the latency of one permutation matters
// there.  The two-lane form of sponge_fused.h issues 120 VALU instructions per round per sponge pair (~4.8 us per
// permutation); this kernel spreads a sponge over 25 lanes -- one 64-bit Keccak lane per GPU lane -- so that a round
// is ~22 VALU instructions plus 14 ds_bpermute_b32 in two dependent LDS round trips (3.16 us per permutation, r03;
// 3.7-3.8 us with 18 gathers in three trips in r02: tools/probe_wide.hip, profiles/r02_wide_lane_probe.txt,
// profiles/r03_wide_round_probe.txt).
//
//   lanes  0..24   tag sponge        kmac_xof(ka, m, 8 tag_len, "..A")       Keccak lane i = x + 5y in GPU lane i
//   lanes 32..56   keystream sponge  kmac_xof(ke, "", |m|, "..E") XOR m      same layout at lane offset 32
//   theta   column parity: 4 + 4 gathers from rows y+1..y+4, then C[x-1], C[x+1]: whole-wave DPP rotations by one lane
//   rho     the lane's own rotation amount: v_alignbit_b32 with a VGPR shift, selects for >= 32 and for 0
//   pi+chi  B[x], B[x+1], B[x+2] gathered straight from the rho output (pi folded into the gather index): 3 + 3
// Message blocks need no staging: GPU lane i < RW loads word i of the block (both sponges read the same 8 RW bytes),
// the keystream lanes XOR and store, the tag lanes absorb.  Same FusedParams, framing restrictions (rate-aligned
// KMAC framing, 8-byte aligned messages) and decrypt protocol as sponge_fused.h, bit-identical results; the launcher
// picks this kernel when the batch is small enough that every wave still has a SIMD (almost) to itself.
#pragma once
#include "sponge_fused.h"

namespace capy {

__device__ __forceinline__ uint32_t wide_bperm(uint32_t byte_index, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_index, (int)v);
}
// a ^ (b & c)
__device__ __forceinline__ uint32_t xor_and(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x78);
}

// bitwise select: (m & a) | (~m & b), one v_bitop3_b32 (truth table 0xCA with the mask as first operand)
__device__ __forceinline__ uint32_t wide_sel(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }

// rho as two 64-bit shifts (1) or as selects around v_alignbit_b32 (0): see wide_round
#ifndef CAPY_WIDE_RHO_SHIFT64
#define CAPY_WIDE_RHO_SHIFT64 1
#endif
struct WideIdx {
    uint32_t up[4];                 // byte index of GPU lane (x, y+k), k = 1..4
    uint32_t b0, b1, b2;            // pi sources of B[x], B[x+1], B[x+2] in row y
    uint32_t sh;                    // alignbit shift of the lane's rho offset r: (32 - r % 32) % 32
    uint32_t m_swap, m_zero, m_l0;  // all-ones masks: r >= 32, r % 32 == 0, "this is Keccak lane 0"
    uint32_t self;                  // byte index of the lane whose state this lane holds (itself, or the lane it mirrors)
};

__device__ __forceinline__ uint32_t wide_rho(uint32_t i)
{
    // rho offsets indexed x + 5y, packed six bits each (FIPS 202 table 2; the table of CAPY_RHO in keccak_dev.h)
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) r = i == (uint32_t)k ? (uint32_t)CAPY_RHO(k) : r;
    return r;
}

// Which Keccak lane a GPU lane stands for.  Lanes 0..24 / 32..56 are the two sponges; the idle lanes MIRROR a lane
// (same index registers, so they compute the same values).  Four of them are placed where the whole-wave rotations of
// theta wrap to (r03): the column parity C does not depend on y, so in the x + 5y layout lane i - 1 / i + 1 always holds
// C[x - 1] / C[x + 1] -- except at the ends of a sponge's 25 lanes, where the rotation reads an idle lane:
//   lane 63 mirrors (4, 0) of the lower sponge   (wave_ror:1 feeds lane 0 from lane 63)
//   lane 25 mirrors (0, 0) of the lower sponge   (wave_rol:1 feeds lane 24 from lane 25)
//   lane 31 mirrors (4, 0) of the upper sponge   (lane 32 <- lane 31)
//   lane 57 mirrors (0, 0) of the upper sponge   (lane 56 <- lane 57)
// the other idle lanes mirror (4, 4) of their own half as before; nobody reads them.
__device__ __forceinline__ WideIdx wide_setup()
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = lane & 32, i = lane & 31;
    if (lane == 63) {
        base = 0;
        i = 4;
    } else if (lane == 25) {
        i = 0;
    } else if (lane == 31) {
        base = 32;
        i = 4;
    } else if (lane == 57) {
        i = 0;
    } else if (i > 24) {
        i = 24;
    }
    const uint32_t x = i % 5, y = i / 5;
    auto at = [&](uint32_t xx, uint32_t yy) { return 4 * (base + (xx % 5) + 5 * (yy % 5)); };
    WideIdx w;
#pragma unroll
    for (int k = 0; k < 4; k++) w.up[k] = at(x, y + 1 + k);
    auto src = [&](uint32_t X, uint32_t Y) { return at((X + 3 * Y) % 5, X % 5); };  // B[X,Y] = rho(E)[(X+3Y)%5, X]
    w.b0 = src(x, y);
    w.b1 = src(x + 1, y);
    w.b2 = src(x + 2, y);
    const uint32_t r = wide_rho(i);
#if CAPY_WIDE_RHO_SHIFT64
    w.sh = r;                   // left shift amount
    w.m_swap = (64 - r) & 63;   // right shift amount (0 for r = 0: v | v)
    w.m_zero = 0;
#else
    w.sh = (32 - (r & 31)) & 31;
    w.m_swap = r >= 32 ? ~0u : 0u;
    w.m_zero = (r & 31) == 0 ? ~0u : 0u;
#endif
    w.m_l0 = i == 0 ? ~0u : 0u;
    w.self = 4 * (base + i);
    return w;
}

// whole-wave rotations by one lane (DPP_WF_RR1 / DPP_WF_RL1): lane i <- lane i - 1 (lane 0 <- lane 63) / lane i <- lane i + 1
// (every lane has a source, so there is no "old" value to keep: mov_dpp, not update_dpp with a zero to materialise)
__device__ __forceinline__ uint32_t wave_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x13C, 0xF, 0xF, false); }
__device__ __forceinline__ uint32_t wave_rol1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x134, 0xF, 0xF, false); }

__device__ __forceinline__ void wide_round(uint32_t &lo, uint32_t &hi, const WideIdx &w, uint32_t rc_lo, uint32_t rc_hi)
{
    // theta: the column parity lands in every lane of the column
    const uint32_t g0 = wide_bperm(w.up[0], lo), g1 = wide_bperm(w.up[1], lo), g2 = wide_bperm(w.up[2], lo),
                   g3 = wide_bperm(w.up[3], lo);
    const uint32_t h0 = wide_bperm(w.up[0], hi), h1 = wide_bperm(w.up[1], hi), h2 = wide_bperm(w.up[2], hi),
                   h3 = wide_bperm(w.up[3], hi);
    const uint32_t cl = xor3(xor3(lo, g0, g1), g2, g3), ch = xor3(xor3(hi, h0, h1), h2, h3);
    // C[x - 1] and C[x + 1] by whole-wave rotations instead of a second LDS round trip (four DPP moves for four
    // ds_bpermute + a wait: timing skeletons 404 -> 338 cycles per round at one wave per CU,
    // profiles/r03_wide_round_probe.txt); the lanes the rotation wraps to mirror the right columns (wide_setup)
    const uint32_t ml = wave_ror1(cl), mh = wave_ror1(ch), pl = wave_rol1(cl), ph = wave_rol1(ch);
    const uint32_t rl = __builtin_amdgcn_alignbit(pl, ph, 31), rh = __builtin_amdgcn_alignbit(ph, pl, 31);  // rol 1
    uint32_t el = xor3(lo, ml, rl), eh = xor3(hi, mh, rh);
    // rho: rotate left by this lane's own offset
#if CAPY_WIDE_RHO_SHIFT64
    // two 64-bit shifts and two ORs (the shifter takes the amount mod 64, so an offset of 0 gives v | v); the 32-bit form
    // below needs two selects before and two after its two alignbits
    const uint64_t v = ((uint64_t)eh << 32) | el;
    const uint64_t rot = (v << w.sh) | (v >> w.m_swap);
    el = (uint32_t)rot;
    eh = (uint32_t)(rot >> 32);
#else
    const uint32_t a = wide_sel(w.m_swap, eh, el), b = wide_sel(w.m_swap, el, eh);
    const uint32_t ra = __builtin_amdgcn_alignbit(a, b, w.sh), rb = __builtin_amdgcn_alignbit(b, a, w.sh);
    el = wide_sel(w.m_zero, a, ra);
    eh = wide_sel(w.m_zero, b, rb);
#endif
    // pi + chi (the gathers read lanes 0..24 / 32..56 only, so a mirroring lane ends the round with its original's state)
    const uint32_t b0l = wide_bperm(w.b0, el), b1l = wide_bperm(w.b1, el), b2l = wide_bperm(w.b2, el);
    const uint32_t b0h = wide_bperm(w.b0, eh), b1h = wide_bperm(w.b1, eh), b2h = wide_bperm(w.b2, eh);
    lo = xor_and(chi3(b0l, b1l, b2l), rc_lo, w.m_l0);
    hi = xor_and(chi3(b0h, b1h, b2h), rc_hi, w.m_l0);
}

template <int... Rs>
__device__ __forceinline__ void wide_permute_impl(uint32_t &lo, uint32_t &hi, const WideIdx &w, std::integer_sequence<int, Rs...>)
{
    (wide_round(lo, hi, w, (uint32_t)keccak_rc64(Rs), (uint32_t)(keccak_rc64(Rs) >> 32)), ...);
}
__device__ __forceinline__ void wide_permute(uint32_t &lo, uint32_t &hi, const WideIdx &w)
{
    // the mirroring lanes take their original's state (absorbed words, restored states: whatever happened between two
    // permutations happened in lanes 0..24 / 32..56): two gathers per permutation
    lo = wide_bperm(w.self, lo);
    hi = wide_bperm(w.self, hi);
    wide_permute_impl(lo, hi, w, std::make_integer_sequence<int, 24>{});
}

template <int RW>
__global__ __launch_bounds__(64) CAPY_WAVES_PER_SIMD(1) void sponge_wide_crypt_kernel(const FusedParams fp)
{
    constexpr uint32_t RB = RW * 8;
    const uint32_t lane = threadIdx.x, role = lane >> 5, i = lane & 31;  // role 0 = tag sponge, 1 = keystream sponge
    const bool word_lane = i < (uint32_t)RW;                              // this lane owns word i of every block
    const uint64_t slot = blockIdx.x;
    if (slot >= fp.n) return;
    const uint64_t item = fp.order ? (uint64_t)fp.order[slot] : slot;
    const WideIdx w = wide_setup();

    // the generic stream machinery (sponge_params.h) describes both sponges: tag = head || msg || 00 01 04 || pad,
    // keystream = head || 00 01 04 || pad
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

    uint64_t tgt_len;
    uint8_t *msg;
    if (fp.offsets) {
        const uint64_t o0 = fp.offsets[item];
        tgt_len = fp.lens ? fp.lens[item] : fp.offsets[item + 1] - o0;
        msg = fp.msgs + o0;
    } else {
        tgt_len = fp.uniform_len;
        msg = fp.msgs + item * fp.msg_stride;
    }
    ItemCtx c;
    c.msg = msg;
    c.key = fp.keka + item * fp.keka_stride + (role == 0 ? fp.ka_offset : 0);
    c.key_len = fp.key_len;
    c.hdr_len = fp.hdr_len;
    c.hdr0 = fp.hdr0;
    c.hdr1 = fp.hdr1;
    c.head_len = fp.head_len;
    c.len = 0;  // the keystream sponge's view; the tag sponge's message blocks are absorbed directly below
    c.suffix = p.suffix;
    {
        const uint64_t total = (uint64_t)fp.head_len + 3;
        const uint32_t rem = (uint32_t)(total % RB);
        c.pad80 = rem != 0;
        c.padded = rem ? total + (RB - rem) : total;
    }
    const uint32_t hb = fp.head_len / RB;

    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) {
        const uint64_t v = role ? fp.init_ks[k] : fp.init_tag[k];
        if (i == (uint32_t)k) {
            lo = (uint32_t)v;
            hi = (uint32_t)(v >> 32);
        }
    }
    auto absorb = [&](uint64_t v) {
        lo ^= (uint32_t)v;
        hi ^= (uint32_t)(v >> 32);
    };

    // ---- heads of both sponges, then the keystream sponge's only other block (00 01 04 || pad); the tag lanes keep
    // their state across that step (the permutation is wave-wide)
    for (uint32_t b = 0; b < hb; b++) {
        if (word_lane) absorb(stream_word(p, c, (uint64_t)b * RB + 8 * i));
        wide_permute(lo, hi, w);
    }
    {
        const uint32_t klo = lo, khi = hi;
        if (word_lane && role == 1) absorb(stream_word(p, c, (uint64_t)hb * RB + 8 * i));
        wide_permute(lo, hi, w);
        if (role == 0) {
            lo = klo;
            hi = khi;
        }
    }
    // from here on the keystream sponge's state IS keystream block 0

    // ---- full blocks: one pass, block t+1 in flight while block t is permuted
    const uint32_t nfull = (uint32_t)(tgt_len / RB);
    const uint32_t partner = 4 * (lane ^ 32);
    uint8_t *my = msg + 8 * i;
    uint64_t pf = 0;
    if (nfull && word_lane) pf = load_global_u64(my);
    for (uint32_t t = 0; t < nfull; t++) {
        const uint64_t in = pf;
        if (t + 1 < nfull && word_lane) pf = load_global_u64(my + (uint64_t)(t + 1) * RB);
        const uint64_t out = in ^ (((uint64_t)hi << 32) | lo);  // meaningful in the keystream lanes
        if (word_lane && role == 1) store_global_u64(my + (uint64_t)t * RB, out);
        uint64_t plain = in;
        if (fp.decrypt) {  // wave-uniform: the tag sponge absorbs the PLAINTEXT = what the keystream lanes just produced
            const uint32_t pl = wide_bperm(partner, (uint32_t)out), ph = wide_bperm(partner, (uint32_t)(out >> 32));
            plain = ((uint64_t)ph << 32) | pl;
        }
        if (word_lane && role == 0) absorb(plain);
        wide_permute(lo, hi, w);
    }

    // ---- tail: fewer than RB message bytes remain.  The keystream lanes XOR and store them and hand the plaintext
    // words to the tag lanes through a lane exchange (never through memory another lane has just written).
    const uint64_t pos = (uint64_t)nfull * RB;
    const uint32_t left = (uint32_t)(tgt_len - pos);
    uint64_t tail_plain = 0;
    {
        uint64_t in = 0, vmask = 0;
        const uint32_t at = 8 * i;
        if (word_lane && at < left) {
            in = load_global_u64(my + pos);  // reads at most 7 bytes past the end of an 8-byte aligned message
            const uint32_t nvalid = left - at < 8 ? left - at : 8;
            vmask = nvalid >= 8 ? ~0ULL : ((1ULL << (8 * nvalid)) - 1ULL);
            in &= vmask;
        }
        const uint64_t out = (in ^ (((uint64_t)hi << 32) | lo)) & vmask;
        if (word_lane && role == 1 && at < left) {
            if (vmask == ~0ULL) {
                store_global_u64(my + pos, out);
            } else {
                for (uint32_t b = 0; b < 8 && at + b < left; b++) my[pos + b] = (uint8_t)(out >> (8 * b));
            }
        }
        const uint64_t mine = fp.decrypt ? out : in;
        const uint32_t pl = wide_bperm(partner, (uint32_t)mine), ph = wide_bperm(partner, (uint32_t)(mine >> 32));
        tail_plain = ((uint64_t)ph << 32) | pl;  // tag lane i: plaintext word i of the tail (zero beyond `left`)
    }
    {
        // tag sponge: plaintext tail || 00 01 04 || 0* [80]   (1 or 2 blocks); the keystream state is no longer needed
        const uint32_t tl = left + 3;
        const uint32_t cnt = (tl + RB - 1) / RB;
        const bool pad80 = (tl % RB) != 0;
        for (uint32_t j = 0; j < cnt; j++) {
            if (word_lane && role == 0) {
                uint64_t v = j == 0 ? tail_plain : 0;
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    const uint32_t rel = j * RB + 8 * i + b;
                    uint64_t byte = 0;
                    if (rel >= left && rel - left < 3) byte = (0x040100u >> (8 * (rel - left))) & 0xff;
                    if (pad80 && rel + 1 == cnt * RB) byte |= 0x80;
                    v |= byte << (8 * b);
                }
                absorb(v);
            }
            wide_permute(lo, hi, w);
        }
    }

    // ---- tag
    if (role == 0 && 8 * i + 8 <= fp.tag_len) {
        uint8_t *o = fp.tags + item * fp.tag_stride + 8 * i;
        const uint64_t v = ((uint64_t)hi << 32) | lo;
        if ((((uintptr_t)o) & 7) == 0)
            store_global_u64(o, v);
        else
            for (int b = 0; b < 8; b++) o[b] = (uint8_t)(v >> (8 * b));
    }
}

}  // namespace capy

namespace capy {

// ---------------------------------------------------------------------------------------------------------------
// sponge_wide_digest_kernel<RW> -- SHA3 / cSHAKE / KMACXOF digests (MODE 0 of sponge_kernels.h) for very small batches
// of long messages: two items per wave (lanes 0..24 and 32..56), each sponge spread over 25 lanes as above.  The
// reference's own benches and integration tests hash / sign ONE 5 MiB message at a time
// (benches/benchmark_sha3.rs:11-19, tests/integration_tests.rs:62-81): a batch that small is nothing but one serial
// chain per sponge, and this form runs the chain 1.6x faster than the two-lane kernel.
//
// Same SpongeParams and stream semantics as sponge_kernel / sponge_kernel_k2 (shared prefix folded into init_state,
// per-item head, body, suffix, pad; every reference quirk is a parameter of the framing), except that raw prefix bytes
// (pre_len != 0, i.e. cSHAKE/KMAC at D224) are not handled here -- the launcher keeps those on the other kernels.
// Block b of an item: GPU lane i < RW absorbs word i.  Full body blocks of 8-byte aligned messages are loaded
// directly (one block in flight ahead of the permutation); head, tail, suffix and pad words come from stream_word.
// The two items of a wave may differ in length: the wave runs max(blocks) steps and a half whose item is finished
// keeps its state across the remaining wave-wide permutations.
template <int RW>
__global__ __launch_bounds__(64) CAPY_WAVES_PER_SIMD(1) void sponge_wide_digest_kernel(const SpongeParams p)
{
    constexpr uint32_t RB = RW * 8;
    const uint32_t lane = threadIdx.x, half = lane >> 5, i = lane & 31;
    const bool word_lane = i < (uint32_t)RW;
    const uint64_t slot = (uint64_t)blockIdx.x * 2 + half;
    const bool in_range = slot < p.n;
    const uint64_t item = in_range ? (p.order ? (uint64_t)p.order[slot] : slot) : p.n;
    const bool active = in_range && (p.mask == nullptr || p.mask[item] != 0);
    const WideIdx w = wide_setup();

    ItemCtx c;
    c.key = nullptr;
    c.msg = nullptr;
    uint64_t tgt_len = 0;
    if (active) {
        if (p.offsets) {
            const uint64_t o0 = p.offsets[item];
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
    const uint64_t total = (uint64_t)c.head_len + c.len + p.suffix_len;
    const uint32_t rem = (uint32_t)(total % RB);
    c.pad80 = p.fips_pad || rem != 0;
    c.padded = rem ? total + (RB - rem) : total;
    const uint32_t nb = active ? (uint32_t)(c.padded / RB) : 0;  // absorb blocks of my item
    const uint32_t hb = c.head_len / RB;
    const bool msg_aligned = active && (((uintptr_t)c.msg & 7) == 0);
    uint32_t nfull = (msg_aligned && p.absorb_body) ? (uint32_t)(c.len / RB) : 0;  // directly loaded blocks
    // An item's LAST absorb block always takes the generic step, which is where the state the squeeze starts from is set
    // aside.  It is a directly loadable body block only when the stream ends on a block boundary with no suffix behind the
    // body (cshake with N = S = "": the caller-framed trailer, suffix_len = 0): found by tools/fuzz_soak.py in r04 -- such an
    // item's digest came from a stale state.
    if (nfull && hb + nfull == nb) nfull--;

    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) {
        if (i == (uint32_t)k) {
            lo = (uint32_t)p.init_state[k];
            hi = (uint32_t)(p.init_state[k] >> 32);
        }
    }

    // ---- absorb: step s handles block s of both items.  A half is FAST at step s when its block s is a directly loaded
    // body block, IDLE when its item is finished (or the half is empty); while every half is one or the other the wave
    // runs the tight loop (a load, an XOR, the permutation -- the generic step's framing logic costs 5-8 % of a block,
    // profiles/r03_wide_round_probe.txt), otherwise one generic step.  A finished half's state is set aside once
    // (flo, fhi) instead of being restored after every wave-wide permutation.
    const uint32_t steps = wave_max_u32(nb);
    const uint8_t *my = c.msg ? c.msg + 8 * i : nullptr;
    const uint32_t fast_end = hb + nfull;
    uint64_t pf = 0;
    uint32_t flo = lo, fhi = hi;
    if (word_lane && nfull && hb == 0) pf = load_global_u64(my);
    uint32_t s = 0;
    while (s < steps) {  // wave-uniform
        const bool fast = s >= hb && s < fast_end;
        if (wave_max_u32((fast || s >= nb) ? 0u : 1u) == 0) {
            const uint32_t end = ~wave_max_u32(~(fast ? fast_end : steps));  // first step at which some half leaves its range
            const bool mine = word_lane && fast;
            for (; s < end; s++) {
                const uint64_t word = pf;
                if (mine && s + 1 < fast_end) pf = load_global_u64(my + (uint64_t)(s + 1 - hb) * RB);
                if (mine) {
                    lo ^= (uint32_t)word;
                    hi ^= (uint32_t)(word >> 32);
                }
                wide_permute(lo, hi, w);
            }
            continue;
        }
        uint64_t word = pf;
        // the block after this one, while this one is permuted
        if (word_lane && s + 1 >= hb && s + 1 < fast_end) pf = load_global_u64(my + (uint64_t)(s + 1 - hb) * RB);
        if (!fast) word = (word_lane && s < nb) ? stream_word(p, c, (uint64_t)s * RB + 8 * i) : 0;
        if (word_lane && s < nb) {
            lo ^= (uint32_t)word;
            hi ^= (uint32_t)(word >> 32);
        }
        wide_permute(lo, hi, w);
        if (s + 1 == nb) {  // my item's last absorb block: this is the state the squeeze starts from
            flo = lo;
            fhi = hi;
        }
        s++;
    }
    lo = flo;
    hi = fhi;

    // ---- squeeze: sq_words words per block, out_len bytes per item
    uint8_t *o = active ? p.out + item * p.out_stride : nullptr;
    uint32_t produced = 0;
    while (produced < p.out_len) {  // wave-uniform
        const uint32_t at = produced + 8 * i;
        if (active && i < p.sq_words && at < p.out_len) {
            const uint64_t v = ((uint64_t)hi << 32) | lo;
            if (at + 8 <= p.out_len && (((uintptr_t)(o + at)) & 7) == 0)
                store_global_u64(o + at, v);
            else
                for (uint32_t b = 0; b < 8 && at + b < p.out_len; b++) o[at + b] = (uint8_t)(v >> (8 * b));
        }
        produced += 8 * p.sq_words;
        if (produced < p.out_len) wide_permute(lo, hi, w);
    }
}

}  // namespace capy
