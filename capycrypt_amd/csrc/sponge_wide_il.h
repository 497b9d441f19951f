// sponge_wide_il.h — one WAVE per sponge, the 1600-bit state spread over the wave's lanes with BIT-INTERLEAVED Keccak lanes:
// the shortest permutation this library has, for batches so small that nothing but the latency of one permutation matters
// (BASELINE config 3 as specified: 128 messages of 5 MiB per GPU; the reference's own benches and tests: ONE 5 MiB message
// at a time, benches/benchmark_sha3.rs:11-19, tests/integration_tests.rs:62-81).  The lane-per-sponge kernels need 4320
// dependent-issue VALU instructions per permutation (7.4 us in a lone wave), the two-lane form 2880 (4.8 us).
//
// r02-r04 (sponge_wide.h, gone): one 64-bit Keccak lane per GPU lane in two VGPRs, two sponges per wave (lanes 0..24 and
// 32..56): 22 VALU + 14 ds_bpermute per round in two dependent LDS round trips, 3.0 us per permutation.  Here a wave holds ONE
// sponge and a Keccak lane is split by bit parity:
//   GPU lanes  0..24   the even bits (0, 2, .. 62) of Keccak lane i = x + 5y, as one 32-bit word
//   GPU lanes 32..56   the odd bits
//   theta   column parity: 4 gathers from rows y+1..y+4 of the same half; C[x-1], C[x+1] by whole-wave DPP rotations by one
//           lane (C does not depend on y, so lane i -+ 1 of the x + 5y layout always holds C[x -+ 1]; four idle lanes mirror
//           the lanes the rotations wrap to)
//   rho     a 64-bit rotation by r is a 32-bit rotation of each half by a lane constant -- r = 2k: both halves by k;
//           r = 2k + 1: the odd half by k + 1 BECOMES the even half, the even half by k becomes the odd one -- and that
//           exchange costs nothing: the pi gather simply reads the other half's lane
//   pi+chi  B[x], B[x+1], B[x+2] gathered straight from the rho output (pi and the half exchange folded into the index)
// Every bitwise step works on a half alone; only theta's rol(C[x+1], 1) needs the partner half of a DIFFERENT lane's value:
// one v_permlane32_swap_b32 (new on gfx950) of two copies + a select.  Per round 12 VALU + 7 ds_bpermute in the same two LDS
// round trips: 2.5 us per permutation (250 cycles per round; timing skeletons tools/gen_valu_census.py wide2 / wide4:
// 337 -> 247 cycles; profiles/r05_wide_interleaved.txt).
// Message words enter and leave through a 16-instruction bit (de)interleave per 32-bit half and one more lane swap.
//
//   sponge_il_digest_kernel<RW, LONE>    one item per wave; taken for up to two items per SIMD
//   sponge_il_crypt_kernel<RW, DECRYPT, LONE>  one item per 128-lane workgroup: wave 0 = tag sponge, wave 1 = keystream sponge;
//                                        taken for up to one item per SIMD.
//                                        The message is turned in place, so only the keystream wave reads and writes it;
//                                        it hands each plaintext block to the tag wave through 512 B of LDS, one
//                                        s_barrier per block, the tag wave two blocks behind so that neither waits.
// SpongeParams / FusedParams and their framing as in the other kernels; bit-identical results (tests/test_gpu_wide_il.py, and
// the whole of tests/test_gpu_sponge.py with these kernels forced).
#pragma once
#include "sponge_fused.h"
#include "occupancy.h"

namespace capy {

__device__ __forceinline__ uint32_t wide_bperm(uint32_t byte_index, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_index, (int)v);
}
// bitwise select: (m & a) | (~m & b), one v_bitop3_b32 (truth table 0xCA with the mask as first operand)
__device__ __forceinline__ uint32_t wide_sel(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }
// whole-wave rotations by one lane (DPP_WF_RR1 / DPP_WF_RL1): lane i <- lane i - 1 (lane 0 <- lane 63) / lane i <- lane i + 1
// (every lane has a source, so there is no "old" value to keep: mov_dpp, not update_dpp with a zero to materialise)
__device__ __forceinline__ uint32_t wave_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x13C, 0xF, 0xF, false); }
__device__ __forceinline__ uint32_t wave_rol1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x134, 0xF, 0xF, false); }
__device__ __forceinline__ uint32_t wide_rho(uint32_t i)
{
    // rho offsets indexed x + 5y (FIPS 202 table 2; the table of CAPY_RHO in keccak_dev.h)
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) r = i == (uint32_t)k ? (uint32_t)CAPY_RHO(k) : r;
    return r;
}


// even / odd bits of a 64-bit constant (round constants, compile time)
__host__ __device__ constexpr uint32_t il_bits(uint64_t v, int parity)
{
    uint32_t r = 0;
    for (int k = 0; k < 32; k++) r |= (uint32_t)((v >> (2 * k + parity)) & 1) << k;
    return r;
}
// rho offsets that are odd, as a bit mask over i = x + 5y
__host__ __device__ constexpr uint32_t il_rho_odd_mask()
{
    uint32_t m = 0;
    for (int k = 0; k < 25; k++) m |= (uint32_t)(CAPY_RHO(k) & 1) << k;
    return m;
}

struct IlIdx {
    uint32_t up[4];       // byte index of GPU lane (x, y+k) of the same half, k = 1..4
    uint32_t b0, b1, b2;  // pi sources of B[x], B[x+1], B[x+2] in row y: the lane AND the half the rotation left the bits in
    uint32_t sh_rho;      // v_alignbit shift of this lane's rho amount (32 - amount) % 32
    uint32_t sh_c;        // ... of rol(C[x+1], 1): 31 in the even half (the odd bits rotated by one), 0 in the odd half
    uint32_t m_hi;        // all ones in GPU lanes 32..63
    uint32_t self;        // byte index of the lane this lane holds the state of (itself, or the lane it mirrors)
    uint32_t sel;         // v_perm_b32 selector of the word <-> halves conversions
    uint32_t rc[24];      // iota: this lane's half of the round constant (zero unless the lane stands for Keccak lane 0)
};

template <int R>
__device__ __forceinline__ uint32_t il_rc(bool lane0, uint32_t e)
{
    constexpr uint32_t even = il_bits(keccak_rc64(R), 0), odd = il_bits(keccak_rc64(R), 1);
    return lane0 ? (e ? odd : even) : 0u;
}
template <int... Rs>
__device__ __forceinline__ void il_rc_fill(IlIdx &w, bool lane0, uint32_t e, std::integer_sequence<int, Rs...>)
{
    ((w.rc[Rs] = il_rc<Rs>(lane0, e)), ...);
}

// The idle lanes MIRROR a lane (same index registers, so they compute the same values): 63 -> (4, 0) of the even half, 25 -> (0, 0) even, 31 -> (4, 0) odd,
// 57 -> (0, 0) odd -- the lanes theta's whole-wave rotations wrap to; the others mirror (4, 4) and are never read.
__device__ __forceinline__ IlIdx il_setup()
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
    const uint32_t e = base >> 5, x = i % 5, y = i / 5;
    IlIdx w;
#pragma unroll
    for (int k = 0; k < 4; k++) w.up[k] = 4 * (base + x + 5 * ((y + 1 + k) % 5));
    // B[X, Y] = rho(E)[j], j = ((X + 3Y) % 5) + 5 X; half e of it sits in half e ^ (rho(j) odd) of GPU lane j
    auto src = [&](uint32_t X, uint32_t Y) {
        X %= 5;
        const uint32_t j = (X + 3 * Y) % 5 + 5 * X;
        constexpr uint32_t odd = il_rho_odd_mask();
        return 4 * (32 * (e ^ ((odd >> j) & 1)) + j);
    };
    w.b0 = src(x, y);
    w.b1 = src(x + 1, y);
    w.b2 = src(x + 2, y);
    const uint32_t r = wide_rho(i);
    const uint32_t amount = ((r >> 1) + (r & 1 & e)) & 31;
    w.sh_rho = (32 - amount) & 31;
    w.sh_c = e ? 0 : 31;
    w.m_hi = (lane & 32) ? ~0u : 0u;
    w.self = 4 * (base + i);
    w.sel = (lane & 32) ? 0x07060302u : 0x05040100u;
    il_rc_fill(w, i == 0, e, std::make_integer_sequence<int, 24>{});
    return w;
}

// The value the OTHER half of the wave holds in the same lane position: lanes 0..31 receive lanes 32..63 and the reverse.
// v_permlane32_swap_b32 a, b exchanges a[32..63] with b[0..31]; with a = b = v that leaves (v.low, v.low) and (v.high, v.high).
struct IlPair {
    uint32_t low, high;  // the values of GPU lane (l & 31) and of GPU lane (l & 31) + 32, in every lane l
};
__device__ __forceinline__ IlPair il_both(uint32_t v)
{
    const auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return {s[0], s[1]};
}

template <int R>
__device__ __forceinline__ void il_round(uint32_t &a, const IlIdx &w)
{
    // theta: the column parity of this half lands in every lane of the column
    const uint32_t g0 = wide_bperm(w.up[0], a), g1 = wide_bperm(w.up[1], a), g2 = wide_bperm(w.up[2], a), g3 = wide_bperm(w.up[3], a);
    const uint32_t c = xor3(xor3(a, g0, g1), g2, g3);
    const uint32_t m = wave_ror1(c), p = wave_rol1(c);  // C[x - 1], C[x + 1] of this half
    const IlPair pp = il_both(p);
    const uint32_t q = wide_sel(w.m_hi, pp.low, pp.high);          // C[x + 1] of the other half
    const uint32_t rot = __builtin_amdgcn_alignbit(q, q, w.sh_c);  // rol(C[x + 1], 1): even <- rol32(odd, 1), odd <- even
    uint32_t t = xor3(a, m, rot);
    // rho: this lane's rotation; which half the bits now belong to is the gather's business
    t = __builtin_amdgcn_alignbit(t, t, w.sh_rho);
    // pi + chi + iota
    const uint32_t b0 = wide_bperm(w.b0, t), b1 = wide_bperm(w.b1, t), b2 = wide_bperm(w.b2, t);
    a = chi3(b0, b1, b2) ^ w.rc[R];
}
template <int... Rs>
__device__ __forceinline__ void il_permute_impl(uint32_t &a, const IlIdx &w, std::integer_sequence<int, Rs...>)
{
    (il_round<Rs>(a, w), ...);
}
__device__ __forceinline__ void il_permute(uint32_t &a, const IlIdx &w)
{
    a = wide_bperm(w.self, a);  // the mirroring lanes take their original's state
    il_permute_impl(a, w, std::make_integer_sequence<int, 24>{});
}

// 32 bits -> even bits in the low half-word, odd bits in the high one (four delta swaps), and back
__device__ __forceinline__ uint32_t il_delta(uint32_t t, uint32_t mask, int sh)
{
    const uint32_t x = (t ^ (t >> sh)) & mask;
    return t ^ x ^ (x << sh);
}
__device__ __forceinline__ uint32_t il_unzip32(uint32_t t)
{
    t = il_delta(t, 0x22222222u, 1);
    t = il_delta(t, 0x0C0C0C0Cu, 2);
    t = il_delta(t, 0x00F000F0u, 4);
    return il_delta(t, 0x0000FF00u, 8);
}
__device__ __forceinline__ uint32_t il_zip32(uint32_t t)
{
    t = il_delta(t, 0x0000FF00u, 8);
    t = il_delta(t, 0x00F000F0u, 4);
    t = il_delta(t, 0x0C0C0C0Cu, 2);
    return il_delta(t, 0x22222222u, 1);
}
// GPU lane i holds the LOW 32 bits of a 64-bit word, GPU lane i + 32 the HIGH 32 bits  ->  lane i its even bits, lane i + 32
// its odd bits (wave-wide: every lane must call it)
__device__ __forceinline__ uint32_t il_in(uint32_t dword, const IlIdx &w)
{
    const IlPair t = il_both(il_unzip32(dword));
    return __builtin_amdgcn_perm(t.high, t.low, w.sel);  // even: low half-words of (high, low); odd: the high half-words
}
// the inverse: the state's halves -> the word's low / high 32 bits
__device__ __forceinline__ uint32_t il_out(uint32_t half, const IlIdx &w)
{
    const IlPair t = il_both(half);  // low = even bits, high = odd bits
    return il_zip32(__builtin_amdgcn_perm(t.high, t.low, w.sel));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL loads and stores
// (vmcnt(0)): the keystream wave would sit out the latency of its store and of the block it fetches ahead once per block
// (measured: 2.98 instead of 2.6 us per block).  The message is only ever touched by that one wave, so LDS order is all the
// hand-over needs.
__device__ __forceinline__ void il_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint32_t il_load_u32(const uint8_t *q) { return *reinterpret_cast<const uint32_t *>(q); }
__device__ __forceinline__ void il_store_u32(uint8_t *q, uint32_t v) { *reinterpret_cast<uint32_t *>(q) = v; }

// ---------------------------------------------------------------------------------------------------------------
// Digests: MODE 0 of sponge_kernels.h, one item per wave.  GPU lane (i, half) absorbs / squeezes 32 bits of word i.
// LONE = the launch holds at most one wave per SIMD: compiled so that a second wave of this kernel does not fit beside it
// (occupancy.h: behind a launch whose waves end staggered the dispatcher otherwise doubles waves up on the SIMDs that free first).
template <int RW, bool LONE>
__global__ __launch_bounds__(64) CAPY_WAVES_PER_SIMD(LONE ? 1 : 8) void sponge_il_digest_kernel(const SpongeParams p)
{
    constexpr uint32_t RB = RW * 8;
    const uint32_t lane = threadIdx.x, e = lane >> 5, i = lane & 31;
    const bool word_lane = i < (uint32_t)RW;
    const uint64_t slot = blockIdx.x;
    if (slot >= p.n) return;
    const uint64_t item = p.order ? (uint64_t)p.order[slot] : slot;
    if (p.mask != nullptr && p.mask[item] == 0) return;  // wave-uniform
    const IlIdx w = il_setup();

    ItemCtx c;
    uint64_t tgt_len;
    if (p.offsets) {
        const uint64_t o0 = p.offsets[item];
        tgt_len = p.lens ? p.lens[item] : p.offsets[item + 1] - o0;
        c.msg = p.msgs + o0;
    } else {
        tgt_len = p.uniform_len;
        c.msg = p.msgs + item * p.msg_stride;
    }
    item_head(p, item, true, c);
    c.len = p.absorb_body ? tgt_len : 0;
    c.suffix = p.suffix;
    if (p.sha3_suffix_rule && (c.len % 136) == 135) c.suffix = (p.suffix & ~0xffULL) | 0x86;
    const uint64_t total = (uint64_t)c.head_len + c.len + p.suffix_len;
    const uint32_t rem = (uint32_t)(total % RB);
    c.pad80 = p.fips_pad || rem != 0;
    c.padded = rem ? total + (RB - rem) : total;
    const uint32_t nb = (uint32_t)(c.padded / RB), hb = c.head_len / RB;
    const bool msg_aligned = (((uintptr_t)c.msg) & 7) == 0;
    uint32_t nfull = (msg_aligned && p.absorb_body) ? (uint32_t)(c.len / RB) : 0;  // body blocks loaded directly
    // An item's LAST absorb block always takes the generic step.  It is a directly loadable body block only when the stream ends
    // on a block boundary with no suffix behind the body (cshake with N = S = "": the caller-framed trailer, suffix_len = 0) --
    // the corner in which r03's wave-per-item kernel squeezed from a stale state (found by tools/fuzz_soak.py in r04).
    if (nfull && hb + nfull == nb) nfull--;

    auto half_of = [&](uint64_t v) { return e ? (uint32_t)(v >> 32) : (uint32_t)v; };
    uint32_t a;
    {
        uint64_t v = 0;
#pragma unroll
        for (int k = 0; k < 25; k++) v = i == (uint32_t)k ? p.init_state[k] : v;
        a = il_in(half_of(v), w);
    }
    auto generic_step = [&](uint32_t s) {
        const uint64_t v = word_lane ? stream_word(p, c, (uint64_t)s * RB + 8 * i) : 0;
        const uint32_t h = il_in(half_of(v), w);
        if (word_lane) a ^= h;
        il_permute(a, w);
    };
    uint32_t s = 0;
    for (; s < hb && s < nb; s++) generic_step(s);
    if (nfull) {
        const uint8_t *my = c.msg + 8 * i + 4 * e;
        uint32_t pf = word_lane ? il_load_u32(my) : 0;
        for (uint32_t t = 0; t < nfull; t++, s++) {
            const uint32_t d = pf;
            if (word_lane && t + 1 < nfull) pf = il_load_u32(my + (uint64_t)(t + 1) * RB);
            const uint32_t h = il_in(d, w);
            if (word_lane) a ^= h;
            il_permute(a, w);
        }
    }
    for (; s < nb; s++) generic_step(s);

    // squeeze: sq_words words per block, out_len bytes per item
    uint8_t *o = p.out + item * p.out_stride;
    uint32_t produced = 0;
    while (produced < p.out_len) {  // wave-uniform
        const uint32_t d = il_out(a, w);
        const uint32_t at = produced + 8 * i + 4 * e;
        if (i < p.sq_words && at < p.out_len) {
            if (at + 4 <= p.out_len && (((uintptr_t)(o + at)) & 3) == 0)
                il_store_u32(o + at, d);
            else
                for (uint32_t b = 0; b < 4 && at + b < p.out_len; b++) o[at + b] = (uint8_t)(d >> (8 * b));
        }
        produced += 8 * p.sq_words;
        if (produced < p.out_len) il_permute(a, w);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sha3_encrypt / sha3_decrypt and the other symmetric halves: the protocol, parameters and restrictions of the
// four-lane kernel of sponge_fused.h (rate-aligned KMAC framing, 8-byte aligned messages), two waves per item.
template <int RW, bool DECRYPT, bool LONE>
__global__ __launch_bounds__(128) CAPY_WAVES_PER_SIMD(LONE ? 1 : 8) void sponge_il_crypt_kernel(const FusedParams fp)
{
    constexpr uint32_t RB = RW * 8;
    __shared__ uint32_t hand[2][64];         // plaintext blocks on their way from the keystream wave to the tag wave
    const uint32_t role = threadIdx.x >> 6;         // wave-uniform: 0 = tag sponge, 1 = keystream sponge
    const uint32_t lane = threadIdx.x & 63, e = lane >> 5, i = lane & 31;
    const bool word_lane = i < (uint32_t)RW;
    const uint64_t slot = blockIdx.x;
    if (slot >= fp.n) return;
    const uint64_t item = fp.order ? (uint64_t)fp.order[slot] : slot;
    const IlIdx w = il_setup();

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
    c.len = 0;  // the keystream sponge's view: head || 00 01 04 || pad
    c.suffix = p.suffix;
    {
        const uint64_t total = (uint64_t)fp.head_len + 3;
        const uint32_t rem = (uint32_t)(total % RB);
        c.pad80 = rem != 0;
        c.padded = rem ? total + (RB - rem) : total;
    }
    const uint32_t hb = fp.head_len / RB;
    auto half_of = [&](uint64_t v) { return e ? (uint32_t)(v >> 32) : (uint32_t)v; };

    uint32_t a;
    {
        uint64_t v = 0;
#pragma unroll
        for (int k = 0; k < 25; k++) v = i == (uint32_t)k ? (role ? fp.init_ks[k] : fp.init_tag[k]) : v;
        a = il_in(half_of(v), w);
    }
    // heads; then the keystream sponge's only other block, after which its state IS keystream block 0
    for (uint32_t b = 0; b < hb + role; b++) {
        const uint64_t v = word_lane ? stream_word(p, c, (uint64_t)b * RB + 8 * i) : 0;
        const uint32_t h = il_in(half_of(v), w);
        if (word_lane) a ^= h;
        il_permute(a, w);
    }

    // ---- full blocks.  Only the keystream wave touches the message: it loads block t, turns it and stores it IN PLACE, and
    // hands the plaintext block (what it read when encrypting, what it wrote when decrypting) to the tag wave through LDS.
    // Everything that is not the permutation runs BESIDE a permutation, in the same basic block, so that it fills the LDS
    // round trips of the gathers instead of lengthening the chain:
    //   step t, keystream wave: copy the state (keystream block t), start the permutation to block t + 1; in its shadow
    //                           convert the copy, XOR, write the LDS slot, request block t + 1 (into the OTHER fetch
    //                           register: steps come in pairs so that no register copy waits for a load); block t - 1 is
    //                           stored at the start of the step
    //   step t, tag wave:       absorb block t - 2 (converted a step ago), permute; in its shadow read the slot of block
    //                           t - 1 and convert it
    // One LDS-only barrier opens every step; the slots alternate (t & 1).
    const uint32_t nfull = (uint32_t)(tgt_len / RB);
    uint8_t *my = msg + 8 * (i < (uint32_t)RW ? i : (uint32_t)RW - 1) + 4 * e;  // the lanes behind the rate fetch (not store) its last word
    const uint32_t steps = nfull ? nfull + 2 : 0;
    const uint32_t word_mask = word_lane ? ~0u : 0u;
    if (role == 1) {
        uint32_t fa = nfull ? il_load_u32(my) : 0, fb = 0, out_prev = 0;
        auto step = [&](uint32_t t, uint32_t &cur, uint32_t &nxt) {
            il_lds_barrier();
            if (t > nfull) return;
            // the one wait for the fetched block -- BEFORE block t - 1 is stored and block t + 1 requested: the memory counter
            // retires in order, so a wait behind the store would sit out the store's latency as well
            asm volatile("" : "+v"(cur) : : "memory");
            if (t >= 1 && word_lane) il_store_u32(msg + 8 * i + 4 * e + (uint64_t)(t - 1) * RB, out_prev);
            if (t == nfull) return;
            const uint32_t ks = a;
            nxt = il_load_u32(my + (uint64_t)(t + 1 < nfull ? t + 1 : t) * RB);
            const uint32_t out = cur ^ il_out(ks, w);
            hand[t & 1][lane] = DECRYPT ? out : cur;
            il_permute(a, w);
            out_prev = out;
        };
        for (uint32_t t = 0; t < steps; t += 2) {
            step(t, fa, fb);
            if (t + 1 < steps) step(t + 1, fb, fa);
        }
    } else {
        uint32_t h_prev = 0;
        for (uint32_t t = 0; t < steps; t++) {
            il_lds_barrier();
            if (t >= 2) {
                a ^= h_prev & word_mask;
                h_prev = il_in(hand[(t - 1) & 1][lane], w);  // block t - 1 (the last step reads a stale slot, unused)
                il_permute(a, w);
            } else if (t == 1) {
                h_prev = il_in(hand[0][lane], w);
            }
        }
    }

    // ---- tail: fewer than RB message bytes remain
    const uint64_t pos = (uint64_t)nfull * RB;
    const uint32_t left = (uint32_t)(tgt_len - pos), at = 8 * i + 4 * e;
    uint32_t tail_plain = 0;
    {
        if (role == 1) {
            uint32_t in = 0, vmask = 0;
            if (word_lane && at < left) {
                in = il_load_u32(msg + at + pos);  // at most 3 bytes past the end of an 8-byte aligned message's last word
                const uint32_t nvalid = left - at < 4 ? left - at : 4;
                vmask = nvalid >= 4 ? ~0u : ((1u << (8 * nvalid)) - 1u);
                in &= vmask;
            }
            const uint32_t out = (in ^ il_out(a, w)) & vmask;
            if (word_lane && at < left) {
                if (vmask == ~0u) {
                    il_store_u32(msg + at + pos, out);
                } else {
                    for (uint32_t b = 0; b < 4 && at + b < left; b++) msg[at + pos + b] = (uint8_t)(out >> (8 * b));
                }
            }
            hand[0][lane] = DECRYPT ? out : in;  // slot 0 is free: the loop's last reads lie before its last barrier
        }
        il_lds_barrier();
        tail_plain = hand[0][lane];
    }
    if (role == 1) return;  // no barrier follows

    {
        // tag sponge: plaintext tail || 00 01 04 || 0* [80]   (1 or 2 blocks)
        const uint32_t tl = left + 3;
        const uint32_t cnt = (tl + RB - 1) / RB;
        const bool pad80 = (tl % RB) != 0;
        for (uint32_t j = 0; j < cnt; j++) {
            uint32_t v = j == 0 ? tail_plain : 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t rel = j * RB + at + b;
                uint32_t byte = 0;
                if (rel >= left && rel - left < 3) byte = (0x040100u >> (8 * (rel - left))) & 0xff;
                if (pad80 && rel + 1 == cnt * RB) byte |= 0x80;
                v |= byte << (8 * b);
            }
            const uint32_t h = il_in(v, w);
            if (word_lane) a ^= h;
            il_permute(a, w);
        }
    }
    const uint32_t d = il_out(a, w);
    if (at + 4 <= fp.tag_len) {  // tag lengths are multiples of 4 (the launcher's fused_shape)
        uint8_t *o = fp.tags + item * fp.tag_stride + at;
        if ((((uintptr_t)o) & 3) == 0)
            il_store_u32(o, d);
        else
            for (int b = 0; b < 4; b++) o[b] = (uint8_t)(d >> (8 * b));
    }
}

}  // namespace capy
