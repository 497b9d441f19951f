// sponge_wide.h — a sponge spread over 25 GPU lanes, one 64-bit Keccak lane per GPU lane, two sponges per wave: the digest
// kernel for batches of up to two items per SIMD (sponge_wide_digest_kernel below), and the round primitives that the
// bit-interleaved one-sponge-per-wave kernels of sponge_wide_il.h (batches of up to ONE item per SIMD, and every
// sha3_encrypt / sha3_decrypt of that size) share.
//
// Such batches are nothing but serial chains -- the reference's own benches and tests hash ONE 5 MiB message at a time,
// BASELINE config 3 as specified leaves 128 messages per GPU -- and only the latency of one permutation matters.  The
// two-lane form of sponge_kernels_k2.h issues 120 VALU instructions per round (~4.8 us per permutation); here a round is
// ~22 VALU instructions plus 14 ds_bpermute_b32 in two dependent LDS round trips (3.0 us per permutation; 3.7-3.8 us with
// 18 gathers in three trips in r02: tools/probe_wide.hip, profiles/r02_wide_lane_probe.txt, profiles/r03_wide_round_probe.txt).
//
//   lanes  0..24   item 2k        Keccak lane i = x + 5y in GPU lane i
//   lanes 32..56   item 2k + 1    same layout at lane offset 32
//   theta   column parity: 4 + 4 gathers from rows y+1..y+4, then C[x-1], C[x+1]: whole-wave DPP rotations by one lane
//   rho     the lane's own rotation amount: two 64-bit shifts by VGPR amounts
//   pi+chi  B[x], B[x+1], B[x+2] gathered straight from the rho output (pi folded into the gather index): 3 + 3
// (r02-r04 also had a sha3_encrypt kernel in this form, tag and keystream sponge side by side in one wave; the two-wave kernel
// of sponge_wide_il.h is 1.03-1.24x faster at every batch size it was taken for and replaced it in r05.)
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
