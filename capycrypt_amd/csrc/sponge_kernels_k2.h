// sponge_kernels_k2.h — the sponge kernel for SMALL batches of LONG messages: two lanes per sponge.
//
// A sponge over one message is a strict chain of permutations, and a wave issues at most one VALU
// instruction every ~4.9 cycles however many SIMDs sit idle (tools/microbench.hip).  With fewer than
// ~64k independent sponges (the 5 MiB-message configurations of BASELINE.json cannot hold more in
// HBM) the chip is not full and throughput = sponges / (instructions per permutation per wave).  This
// kernel therefore splits every 64-bit Keccak lane across an adjacent lane pair: the even GPU lane
// holds the low 32-bit halves of all 25 Keccak lanes, the odd lane the high halves.
//   theta / chi / iota are bit-parallel            -> identical code on both halves (25 VGPRs of state)
//   rotations need the partner's half              -> one v_mov_b32_dpp quad_perm:[1,0,3,2] + one v_alignbit_b32
// 120 VALU per round per wave instead of 180, for 32 sponges per wave instead of 64: 1.48x the
// per-sponge speed of the one-lane-per-sponge kernel whenever the chip is under-occupied (40.4 vs 59.7 ms per MiB
// of message), 0.75x when it is full (so the launcher picks this kernel only for small batches, see sponge.hip).
// The DPP move must carry bound_ctrl: without it the compiler materialises the "old" operand with a v_mov_b32 per
// swap (+29 VALU per round, measured 1.12x instead of 1.48x).
//
// Same stream semantics, parameters and phases as sponge_kernels.h (which documents the framing).
#pragma once
#include "sponge_params.h"

namespace capy {

__device__ __forceinline__ uint32_t dpp_swap_pair(uint32_t v)
{
    // quad_perm:[1,0,3,2] -> dpp_ctrl = 1 | 0<<2 | 3<<4 | 2<<6 = 0xB1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

// rotate-left of a 64-bit lane split as (own half, partner half); valid for both halves
template <int R>
__device__ __forceinline__ uint32_t rol64_half(uint32_t own, uint32_t partner)
{
    static_assert(R != 32, "no rho offset of 32 in keccak");
    if constexpr (R == 0)
        return own;
    else if constexpr (R < 32)
        return __builtin_amdgcn_alignbit(own, partner, 32 - R);
    else
        return __builtin_amdgcn_alignbit(partner, own, 64 - R);
}

struct KHalf {
    uint32_t a[25];
};

template <int I>
__device__ __forceinline__ void rho_pi_half(const KHalf &e, KHalf &b)
{
    constexpr int x = I % 5, y = I / 5;
    constexpr int dst = y + 5 * ((2 * x + 3 * y) % 5);
    if constexpr (I == 0)
        b.a[dst] = e.a[0];
    else
        b.a[dst] = rol64_half<CAPY_RHO(I)>(e.a[I], dpp_swap_pair(e.a[I]));
}
template <int... Is>
__device__ __forceinline__ void rho_pi_half_all(const KHalf &e, KHalf &b, std::integer_sequence<int, Is...>)
{
    (rho_pi_half<Is>(e, b), ...);
}

// hmask = 0 on the low-half lane, ~0 on the high-half lane
__device__ __forceinline__ void keccak_round_k2(KHalf &s, uint32_t rc_lo, uint32_t rc_x, uint32_t hmask)
{
    uint32_t c[5], r[5];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = xor3(xor3(s.a[x], s.a[x + 5], s.a[x + 10]), s.a[x + 15], s.a[x + 20]);
#pragma unroll
    for (int x = 0; x < 5; x++) r[x] = rol64_half<1>(c[x], dpp_swap_pair(c[x]));
    KHalf e, b;
#pragma unroll
    for (int i = 0; i < 25; i++) e.a[i] = xor3(s.a[i], c[(i % 5 + 4) % 5], r[(i % 5 + 1) % 5]);
    rho_pi_half_all(e, b, std::make_integer_sequence<int, 25>{});
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) s.a[x + 5 * y] = chi3(b.a[x + 5 * y], b.a[(x + 1) % 5 + 5 * y], b.a[(x + 2) % 5 + 5 * y]);
    // iota: my half of RC = rc_lo ^ ((rc_lo ^ rc_hi) & hmask)
    s.a[0] = xor3(s.a[0], rc_x & hmask, rc_lo);
}

// The two-lane round for TWO OR MORE WAVES PER SIMD (keccak_dev.h: keccak_round_blocked; profiles/
// r03_valu_issue_bisect.txt): 62 simple instructions (parity, theta-apply, chi + iota) and 58 four-cycle ones (a DPP move +
// v_alignbit_b32 per rotation) in two blocks that run at raised priority, kept apart by sched_barrier.  Synthetic stream
// of this shape at two waves per SIMD: 4.07 -> 2.74 cycles per instruction.
template <bool PRIO>
__device__ __forceinline__ void keccak_round_k2_blocked(KHalf &s, uint32_t rc_lo, uint32_t rc_x, uint32_t hmask)
{
    uint32_t c[5], r[5];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = xor3(xor3(s.a[x], s.a[x + 5], s.a[x + 10]), s.a[x + 15], s.a[x + 20]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int x = 0; x < 5; x++) r[x] = rol64_half<1>(c[x], dpp_swap_pair(c[x]));
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    KHalf e, b;
#pragma unroll
    for (int i = 0; i < 25; i++) e.a[i] = xor3(s.a[i], c[(i % 5 + 4) % 5], r[(i % 5 + 1) % 5]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    rho_pi_half_all(e, b, std::make_integer_sequence<int, 25>{});
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) s.a[x + 5 * y] = chi3(b.a[x + 5 * y], b.a[(x + 1) % 5 + 5 * y], b.a[(x + 2) % 5 + 5 * y]);
    s.a[0] = xor3(s.a[0], rc_x & hmask, rc_lo);
}
template <bool PRIO, int... Rs>
__device__ __forceinline__ void keccakf1600_k2_paired_unrolled_impl(KHalf &s, uint32_t hmask, std::integer_sequence<int, Rs...>)
{
    // literal round constants: an SGPR operand would turn iota into a 4-cycle instruction inside a simple block
    (keccak_round_k2_blocked<PRIO>(s, (uint32_t)keccak_rc64(Rs), (uint32_t)keccak_rc64(Rs) ^ (uint32_t)(keccak_rc64(Rs) >> 32), hmask), ...);
}
template <bool PRIO>
__device__ __forceinline__ void keccakf1600_k2_paired_unrolled(KHalf &s, uint32_t hmask)
{
    keccakf1600_k2_paired_unrolled_impl<PRIO>(s, hmask, std::make_integer_sequence<int, 24>{});
}

template <int... Rs>
__device__ __forceinline__ void keccakf1600_k2_unrolled_impl(KHalf &s, uint32_t hmask, std::integer_sequence<int, Rs...>)
{
    // re-aligned to 8 bytes after every round, see keccak_round_aligned (keccak_dev.h)
    ((keccak_round_k2(s, (uint32_t)keccak_rc64(Rs), (uint32_t)keccak_rc64(Rs) ^ (uint32_t)(keccak_rc64(Rs) >> 32), hmask),
      [&] { asm volatile(".p2align 3" : "+v"(s.a[0])); }()),
     ...);
}
__device__ __forceinline__ void keccakf1600_k2_unrolled(KHalf &s, uint32_t hmask)
{
    keccakf1600_k2_unrolled_impl(s, hmask, std::make_integer_sequence<int, 24>{});
}

__device__ __forceinline__ void keccakf1600_k2(KHalf &s, uint32_t hmask)
{
#pragma unroll 2
    for (int r = 0; r < 24; r++) {
        const uint32_t lo = KECCAK_RC32[2 * r], hi = KECCAK_RC32[2 * r + 1];
        keccak_round_k2(s, lo, lo ^ hi, hmask);
        asm volatile(".p2align 3" : "+v"(s.a[0]));
    }
}

// Rolled two-round body with the next pair of round constants fetched one trip ahead (the scalar-load latency never
// sits between two rounds): ~2 KB of instructions instead of the 23 KB of the unrolled form.  The experiment of
// profiles/r02_second_issue_slot.txt: does a second wave on the SIMD keep its issue rate when the body is small?
__device__ __forceinline__ void keccakf1600_k2_pipelined(KHalf &s, uint32_t hmask)
{
    uint32_t c0 = KECCAK_RC32[0], c1 = KECCAK_RC32[1], c2 = KECCAK_RC32[2], c3 = KECCAK_RC32[3];
#pragma unroll 1
    for (int r = 0; r < 24; r += 2) {
        const int nx = (r + 2 < 24) ? r + 2 : 0;
        const uint32_t n0 = KECCAK_RC32[2 * nx], n1 = KECCAK_RC32[2 * nx + 1], n2 = KECCAK_RC32[2 * nx + 2],
                       n3 = KECCAK_RC32[2 * nx + 3];
        keccak_round_k2(s, c0, c0 ^ c1, hmask);
        asm volatile(".p2align 3" : "+v"(s.a[0]));
        keccak_round_k2(s, c2, c2 ^ c3, hmask);
        asm volatile(".p2align 3" : "+v"(s.a[0]));
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
    }
}

// BODY 0: fully unrolled permutation with literal round constants in the block loop (default);
// BODY 1: the rolled two-round form above (A/B instance, selected by debug bit 3 of capy_set_sponge_lanes)
// BODY 2: the blocked round with raised priority (keccak_round_k2_blocked), for two waves per SIMD (A/B instance for
//         SHA3-256 digests, forced two-lane launches of more than 32 sponges per SIMD: profiles/r03_chipfull.txt)
// BODY 0 / 1 serve at most 32 sponges per SIMD = one wave per SIMD: see CAPY_WAVES_PER_SIMD (keccak_dev.h)
template <int RW, int MODE, int BODY = 0>
__global__ __launch_bounds__(64) CAPY_WAVES_PER_SIMD(BODY == 2 ? 8 : 1) void sponge_kernel_k2(const SpongeParams p)
{
    constexpr uint32_t RB = RW * 8;
    constexpr int NSP = 32;                        // sponges per wave
    constexpr int NLOAD = (NSP * RW + 63) / 64;    // 8-byte cooperative loads per block step
    __shared__ uint64_t s_stage[NLOAD * 64];
    __shared__ uint64_t s_base[NSP];
    __shared__ uint32_t s_nfull[NSP];

    const uint32_t lane = threadIdx.x;
    const uint32_t h = lane & 1, j = lane >> 1;
    const uint32_t hmask = 0u - h;
    const uint64_t slot = (uint64_t)blockIdx.x * NSP + j;
    const bool in_range = slot < p.n;
    const uint64_t item = in_range ? (p.order ? (uint64_t)p.order[slot] : slot) : p.n;
    const bool active = in_range && (p.mask == nullptr || p.mask[item] != 0);

    ItemCtx c;
    c.key = nullptr;
    c.msg = nullptr;
    uint64_t tgt_len = 0;
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

    const bool grid_aligned = ((p.pre_len + (p.key_offsets ? 0u : p.head_len)) % RB == 0) && (p.stride_bytes == RB);
    // head blocks: wave-uniform unless the keys have per-item lengths (then a multiple of w = RB per lane)
    const uint32_t hb = grid_aligned ? (p.pre_len + c.head_len) / RB : 0;
    const uint32_t hb_max = p.key_offsets ? wave_max_u32(hb) : hb;
    const bool msg_aligned = active && grid_aligned && (((uintptr_t)c.msg & 7) == 0);
    const uint32_t nfull = (msg_aligned && p.absorb_body) ? (uint32_t)(c.len / RB) : 0;

    KHalf a;
#pragma unroll
    for (int i = 0; i < 25; i++) a.a[i] = h ? (uint32_t)(p.init_state[i] >> 32) : (uint32_t)p.init_state[i];

    auto absorb_slow = [&](uint64_t base) {
#pragma unroll
        for (int w = 0; w < RW; w++) {
            uint64_t v = stream_word(p, c, base + 8 * w);
            a.a[w] ^= h ? (uint32_t)(v >> 32) : (uint32_t)v;
        }
        keccakf1600_k2(a, hmask);
    };

    // ---------------- phase H
    for (uint32_t b = 0; b < hb_max; b++)
        if (active && b < hb) absorb_slow((uint64_t)b * RB);  // a pair is active or inactive as a whole

    // ---------------- phase B: each lane of a pair loads its own 32-bit half of every word of its sponge's block, the
    // next block in flight under the rounds.  A pair's block lies in two or three 128-byte lines that stay in the CU's
    // vector cache across its 17..21 loads (one wave per SIMD: 4 x 32 sponges per CU).  1.5-2.5 % faster than the
    // wave-cooperative loads through LDS of round 1 (profiles/r02_direct_loads_ab.txt), 28 fewer VGPRs.
    const uint8_t *last_word = batch_last_word(p.msgs, p.offsets, p.n, p.msg_stride, p.uniform_len);
    if (h == 0) s_base[j] = (uint64_t)(uintptr_t)(c.msg ? c.msg : p.msgs);  // MODE 1 reads it after its own barrier
    {
        const uint32_t max_full = wave_max_u32(nfull);
        if (max_full) {
            const uint8_t *mine = (c.msg ? c.msg : p.msgs) + 4 * h;
            const uint8_t *safe = last_word;  // 8 readable bytes for lanes past their own last full block
            auto own_load = [&](uint32_t t, uint32_t (&pf)[RW]) {
                const uint8_t *q = t < nfull ? mine + (uint64_t)t * RB : safe;
#pragma unroll
                for (int w = 0; w < RW; w++)
                    pf[w] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(
                        reinterpret_cast<uintptr_t>(t < nfull ? q + 8 * w : q));
            };
            uint32_t pf[RW];
            own_load(0, pf);
            for (uint32_t t = 0; t < max_full; t++) {
                if (t < nfull) {
#pragma unroll
                    for (int w = 0; w < RW; w++) a.a[w] ^= pf[w];
                }
                if (t + 1 < max_full) own_load(t + 1, pf);
                if (t < nfull) {
                    if constexpr (BODY == 1)
                        keccakf1600_k2_pipelined(a, hmask);
                    else if constexpr (BODY == 2)
                        keccakf1600_k2_paired_unrolled<true>(a, hmask);
                    else
                        keccakf1600_k2_unrolled(a, hmask);
                }
            }
        }
    }

    // ---------------- phase T
    {
        const uint32_t first = hb + nfull;
        const uint32_t cnt = nb > first ? nb - first : 0;
        const uint32_t max_cnt = wave_max_u32(cnt);
        for (uint32_t q = 0; q < max_cnt; q++)
            if (q < cnt) absorb_slow((uint64_t)(first + q) * RB);
    }

    // ---------------- squeeze
    if constexpr (MODE == 0) {
        uint8_t *o = active ? p.out + item * p.out_stride : nullptr;
        uint32_t produced = 0;
        while (produced < p.out_len) {
#pragma unroll
            for (int w = 0; w < 25; w++) {
                if ((uint32_t)w < p.sq_words) {
                    const uint32_t at = produced + 4 * h;  // my half of word w
                    if (active && at < p.out_len) {
                        const uint32_t v = a.a[w];
                        if (at + 4 <= p.out_len && (((uintptr_t)(o + at)) & 3) == 0) {
                            *reinterpret_cast<uint32_t *>(o + at) = v;
                        } else {
                            for (uint32_t b = 0; b < 4 && at + b < p.out_len; b++) o[at + b] = (uint8_t)(v >> (8 * b));
                        }
                    }
                    produced += 8;
                }
            }
            if (produced < p.out_len) keccakf1600_k2(a, hmask);
        }
    } else {
        // keystream XOR in place (cSHAKE/KMAC: squeeze block = RW words)
        const uint32_t xfull = msg_aligned ? (uint32_t)(tgt_len / RB) : 0;
        __syncthreads();
        if (h == 0) s_nfull[j] = xfull;
        __syncthreads();
        const uint32_t max_x = wave_max_u32(xfull);
        uint32_t *stage32 = reinterpret_cast<uint32_t *>(s_stage);
        if (max_x) {
            uint8_t *dst[NLOAD];
            uint32_t lim[NLOAD];
#pragma unroll
            for (int k = 0; k < NLOAD; k++) {
                const uint32_t i = k * 64 + lane;
                const uint32_t m = i / RW, w = i - m * RW;
                const bool in = m < NSP;
                lim[k] = in ? s_nfull[in ? m : 0] : 0;
                dst[k] = in ? reinterpret_cast<uint8_t *>(s_base[m]) + 8 * w : const_cast<uint8_t *>(p.msgs);
            }
            uint64_t pf[NLOAD];
            auto coop_load = [&](uint32_t t) {
#pragma unroll
                for (int k = 0; k < NLOAD; k++) {
                    pf[k] = load_global_u64(ragged_src(t < lim[k], dst[k], (uint64_t)t * RB, last_word));
                }
            };
            coop_load(0);
            for (uint32_t t = 0; t < max_x; t++) {
#pragma unroll
                for (int k = 0; k < NLOAD; k++) s_stage[k * 64 + lane] = pf[k];
                __syncthreads();
                if (t < xfull) {
#pragma unroll
                    for (int w = 0; w < RW; w++) stage32[(j * RW + w) * 2 + h] ^= a.a[w];
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < NLOAD; k++) {
                    const uint64_t v = s_stage[k * 64 + lane];
                    if (t < lim[k]) store_global_u64(dst[k] + (uint64_t)t * RB, v);
                }
                __syncthreads();
                if (t + 1 < max_x) coop_load(t + 1);
                if (t < xfull && (uint64_t)(t + 1) * RB < tgt_len) keccakf1600_k2(a, hmask);
            }
        }
        uint64_t pos = (uint64_t)xfull * RB;
        const uint64_t left = tgt_len - pos;
        const uint32_t cnt = (uint32_t)((left + RB - 1) / RB);
        const uint32_t max_cnt = wave_max_u32(cnt);
        uint8_t *m = const_cast<uint8_t *>(c.msg);
        for (uint32_t q = 0; q < max_cnt; q++) {
            if (q < cnt) {
#pragma unroll
                for (int w = 0; w < RW; w++) {
                    const uint32_t v = a.a[w];
                    for (int b = 0; b < 4; b++) {
                        const uint64_t at = pos + 8 * w + 4 * h + b;
                        if (at < tgt_len) m[at] ^= (uint8_t)(v >> (8 * b));
                    }
                }
                pos += RB;
                if (pos < tgt_len) keccakf1600_k2(a, hmask);
            }
        }
    }
}

}  // namespace capy
