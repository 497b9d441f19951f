// keccak_dev.h — keccak-f[1600] for one sponge per lane on gfx950 (CDNA4).
//
// Replaces the reference's keccakf_1600 (/root/reference/src/sha3/keccakf.rs:8-423).
// The 25 x u64 state lives in 50 VGPRs as (lo, hi) 32-bit halves.  Per round:
//   theta  column parity with v_bitop3_b32 (3-input XOR, truth table 0x96)      20 ops
//          rol1(C) with v_alignbit_b32                                          10 ops
//          A ^= C[x-1] ^ rol1(C[x+1]) as one bitop3 per half                    50 ops
//   rho    two v_alignbit_b32 per lane (rotations are compile-time constants)   48 ops
//   pi     register renaming (free)
//   chi    a ^ (~b & c) as one v_bitop3_b32 (truth table 0xD2) per half         50 ops
//   iota   2 ops
// = 180 VALU ops / round, 4320 / permutation, no cross-lane traffic, no LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "occupancy.h"

namespace capy {

__device__ __constant__ const uint32_t KECCAK_RC32[48] = {
    0x00000001u, 0x00000000u, 0x00008082u, 0x00000000u, 0x0000808Au, 0x80000000u, 0x80008000u, 0x80000000u,
    0x0000808Bu, 0x00000000u, 0x80000001u, 0x00000000u, 0x80008081u, 0x80000000u, 0x00008009u, 0x80000000u,
    0x0000008Au, 0x00000000u, 0x00000088u, 0x00000000u, 0x80008009u, 0x00000000u, 0x8000000Au, 0x00000000u,
    0x8000808Bu, 0x00000000u, 0x0000008Bu, 0x80000000u, 0x00008089u, 0x80000000u, 0x00008003u, 0x80000000u,
    0x00008002u, 0x80000000u, 0x00000080u, 0x80000000u, 0x0000800Au, 0x00000000u, 0x8000000Au, 0x80000000u,
    0x80008081u, 0x80000000u, 0x00008080u, 0x80000000u, 0x80000001u, 0x00000000u, 0x80008008u, 0x80000000u};

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
// a ^ (~b & c)
__device__ __forceinline__ uint32_t chi3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xD2);
}

struct KState {
    uint32_t lo[25];
    uint32_t hi[25];
};

// 64-bit rotate-left by a compile-time amount on (lo, hi) halves.
template <int R>
__device__ __forceinline__ void rol64c(uint32_t lo, uint32_t hi, uint32_t &olo, uint32_t &ohi)
{
    if constexpr (R == 0) {
        olo = lo;
        ohi = hi;
    } else if constexpr (R == 32) {
        olo = hi;
        ohi = lo;
    } else if constexpr (R < 32) {
        olo = __builtin_amdgcn_alignbit(lo, hi, 32 - R);
        ohi = __builtin_amdgcn_alignbit(hi, lo, 32 - R);
    } else {
        olo = __builtin_amdgcn_alignbit(hi, lo, 64 - R);
        ohi = __builtin_amdgcn_alignbit(lo, hi, 64 - R);
    }
}

// rho offsets indexed [x + 5y] (FIPS 202 table 2)
#define CAPY_RHO(i)                                                                                          \
    ((i) == 0 ? 0 : (i) == 1 ? 1 : (i) == 2 ? 62 : (i) == 3 ? 28 : (i) == 4 ? 27 : (i) == 5 ? 36 : (i) == 6 ? 44 \
     : (i) == 7 ? 6 : (i) == 8 ? 55 : (i) == 9 ? 20 : (i) == 10 ? 3 : (i) == 11 ? 10 : (i) == 12 ? 43        \
     : (i) == 13 ? 25 : (i) == 14 ? 39 : (i) == 15 ? 41 : (i) == 16 ? 45 : (i) == 17 ? 15 : (i) == 18 ? 21    \
     : (i) == 19 ? 8 : (i) == 20 ? 18 : (i) == 21 ? 2 : (i) == 22 ? 61 : (i) == 23 ? 56 : 14)

template <int I>
__device__ __forceinline__ void rho_pi_one(const KState &e, KState &b)
{
    constexpr int x = I % 5, y = I / 5;
    constexpr int dst = y + 5 * ((2 * x + 3 * y) % 5);
    rol64c<CAPY_RHO(I)>(e.lo[I], e.hi[I], b.lo[dst], b.hi[dst]);
}

template <int... Is>
__device__ __forceinline__ void rho_pi_all(const KState &e, KState &b, std::integer_sequence<int, Is...>)
{
    (rho_pi_one<Is>(e, b), ...);
}

__device__ __forceinline__ void keccak_round(KState &a, uint32_t rc_lo, uint32_t rc_hi)
{
    uint32_t cl[5], ch[5], rl[5], rh[5];
#pragma unroll
    for (int x = 0; x < 5; x++) {
        cl[x] = xor3(xor3(a.lo[x], a.lo[x + 5], a.lo[x + 10]), a.lo[x + 15], a.lo[x + 20]);
        ch[x] = xor3(xor3(a.hi[x], a.hi[x + 5], a.hi[x + 10]), a.hi[x + 15], a.hi[x + 20]);
    }
#pragma unroll
    for (int x = 0; x < 5; x++) rol64c<1>(cl[x], ch[x], rl[x], rh[x]);
    KState e, b;
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) {
            e.lo[x + 5 * y] = xor3(a.lo[x + 5 * y], cl[(x + 4) % 5], rl[(x + 1) % 5]);
            e.hi[x + 5 * y] = xor3(a.hi[x + 5 * y], ch[(x + 4) % 5], rh[(x + 1) % 5]);
        }
    rho_pi_all(e, b, std::make_integer_sequence<int, 25>{});
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) {
            a.lo[x + 5 * y] = chi3(b.lo[x + 5 * y], b.lo[(x + 1) % 5 + 5 * y], b.lo[(x + 2) % 5 + 5 * y]);
            a.hi[x + 5 * y] = chi3(b.hi[x + 5 * y], b.hi[(x + 1) % 5 + 5 * y], b.hi[(x + 2) % 5 + 5 * y]);
        }
    a.lo[0] ^= rc_lo;
    a.hi[0] ^= rc_hi;
}

// gfx950 fetches instructions in 8-byte granules: a stream of 8-byte VOP3 encodings (v_bitop3_b32, v_alignbit_b32)
// that sits at 4 mod 8 issues ~20 % slower for a lone wave (measured: the same 4361-instruction loop body ran 363 ms
// vs 300 ms depending only on that phase).  The iota XOR is a 4-byte encoding whenever the round constant half is an
// inline constant, so the phase can flip between rounds: re-align after every unrolled round (at most one s_nop).
// The directive is tied to the lane the iota XORs just wrote, so it lands between two rounds.
__device__ __forceinline__ void keccak_round_aligned(KState &a, uint32_t rc_lo, uint32_t rc_hi)
{
    keccak_round(a, rc_lo, rc_hi);
    asm volatile(".p2align 3" : "+v"(a.lo[0]), "+v"(a.hi[0]));
}
__device__ __forceinline__ void keccakf1600(KState &a)
{
#pragma unroll 2
    for (int r = 0; r < 24; r++) keccak_round_aligned(a, KECCAK_RC32[2 * r], KECCAK_RC32[2 * r + 1]);
}

// Rolled form with the next pair of round constants fetched one trip ahead, so the scalar-load latency
// never sits between two rounds (a lone wave cannot hide it).
__device__ __forceinline__ void keccakf1600_pipelined(KState &a)
{
    uint32_t c0 = KECCAK_RC32[0], c1 = KECCAK_RC32[1], c2 = KECCAK_RC32[2], c3 = KECCAK_RC32[3];
#pragma unroll 1
    for (int r = 0; r < 24; r += 2) {
        const int nx = (r + 2 < 24) ? r + 2 : 0;
        const uint32_t n0 = KECCAK_RC32[2 * nx], n1 = KECCAK_RC32[2 * nx + 1], n2 = KECCAK_RC32[2 * nx + 2],
                       n3 = KECCAK_RC32[2 * nx + 3];
        keccak_round(a, c0, c1);  // many waves per SIMD: fetch phase is hidden, no re-alignment
        keccak_round(a, c2, c3);
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
    }
}

// Round constants as compile-time values: the fully unrolled permutation of the hot block loop carries
// them as instruction literals (no scalar loads, no loop counter).
__device__ __host__ constexpr uint64_t keccak_rc64(int r)
{
    constexpr uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
        0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    return RC[r];
}

// ---- the round for TWO OR MORE WAVES PER SIMD (profiles/r03_valu_issue_bisect.txt) -------------------------------------
// Measured on MI355X with one instrument (rocprofv3 --pmc + in-kernel clocks): a SIMD issues a wave64 VALU instruction
// of the SIMPLE class (v_bitop3_b32, v_xor/and/or/not, v_add/sub_u32, v_lshrrev_b32, v_mov_b32, f32 add/mul/fma) in
// 2 cycles, so two waves that each issue every 4 cycles share it at full rate -- but every other opcode
// (v_alignbit_b32, v_lshlrev_b32, DPP moves, every 3-operand integer op but bitop3, all multiplies, all 64-bit ops,
// carry-out adds) holds the SIMD for 4 cycles, and the arbiter gives the OLDER wave every issue window: a younger
// wave advances through simple instructions only and stops at its first 4-cycle instruction until the older wave is
// done.  The keccak round is 122 simple + 58 v_alignbit_b32, so a second wave added nothing (4.0 cycles per
// instruction at 1, 2, 4, 7 waves per SIMD).  Raising the wave's priority around its 4-cycle blocks lets the younger
// wave take those windows too: the two waves then overlap their simple blocks two instructions per window and run
// their rotation blocks one after the other -- (58 + 58 + 122) windows for two rounds instead of 360; measured on a
// synthetic stream of this shape 2.5 cycles per instruction at two waves per SIMD (1.45x), 2.3 at four.
// The compiler must not move instructions between the blocks (sched_barrier), and a lone wave pays ~8 % for the four
// s_setprio, so the launchers take this form only when every SIMD holds at least two waves.
#ifndef CAPY_PAIRED_PRIO
#define CAPY_PAIRED_PRIO true  // false: the same blocked round without s_setprio (A/B builds only)
#endif
template <bool PRIO>
__device__ __forceinline__ void keccak_round_blocked(KState &a, uint32_t rc_lo, uint32_t rc_hi)
{
    uint32_t cl[5], ch[5], rl[5], rh[5];
#pragma unroll
    for (int x = 0; x < 5; x++) {
        cl[x] = xor3(xor3(a.lo[x], a.lo[x + 5], a.lo[x + 10]), a.lo[x + 15], a.lo[x + 20]);
        ch[x] = xor3(xor3(a.hi[x], a.hi[x + 5], a.hi[x + 10]), a.hi[x + 15], a.hi[x + 20]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int x = 0; x < 5; x++) rol64c<1>(cl[x], ch[x], rl[x], rh[x]);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    KState e, b;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        e.lo[i] = xor3(a.lo[i], cl[(i % 5 + 4) % 5], rl[(i % 5 + 1) % 5]);
        e.hi[i] = xor3(a.hi[i], ch[(i % 5 + 4) % 5], rh[(i % 5 + 1) % 5]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    rho_pi_all(e, b, std::make_integer_sequence<int, 25>{});
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) {
            a.lo[x + 5 * y] = chi3(b.lo[x + 5 * y], b.lo[(x + 1) % 5 + 5 * y], b.lo[(x + 2) % 5 + 5 * y]);
            a.hi[x + 5 * y] = chi3(b.hi[x + 5 * y], b.hi[(x + 1) % 5 + 5 * y], b.hi[(x + 2) % 5 + 5 * y]);
        }
    a.lo[0] ^= rc_lo;
    a.hi[0] ^= rc_hi;
}

// Fully unrolled, literal round constants (an SGPR operand would turn the iota XOR into a 4-cycle instruction inside a
// simple block): the form for exactly two waves per SIMD, where nothing but the partner wave hides a scalar load.
template <bool PRIO, int... Rs>
__device__ __forceinline__ void keccakf1600_paired_unrolled_impl(KState &a, std::integer_sequence<int, Rs...>)
{
    (keccak_round_blocked<PRIO>(a, (uint32_t)keccak_rc64(Rs), (uint32_t)(keccak_rc64(Rs) >> 32)), ...);
}
template <bool PRIO>
__device__ __forceinline__ void keccakf1600_paired_unrolled(KState &a)
{
    keccakf1600_paired_unrolled_impl<PRIO>(a, std::make_integer_sequence<int, 24>{});
}

// Rolled two-round body on the blocked round, constants one trip ahead (the many-waves form of keccakf1600_pipelined).
template <bool PRIO>
__device__ __forceinline__ void keccakf1600_paired(KState &a)
{
    uint32_t c0 = KECCAK_RC32[0], c1 = KECCAK_RC32[1], c2 = KECCAK_RC32[2], c3 = KECCAK_RC32[3];
#pragma unroll 1
    for (int r = 0; r < 24; r += 2) {
        const int nx = (r + 2 < 24) ? r + 2 : 0;
        const uint32_t n0 = KECCAK_RC32[2 * nx], n1 = KECCAK_RC32[2 * nx + 1], n2 = KECCAK_RC32[2 * nx + 2],
                       n3 = KECCAK_RC32[2 * nx + 3];
        keccak_round_blocked<PRIO>(a, c0, c1);
        keccak_round_blocked<PRIO>(a, c2, c3);
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
    }
}

template <int... Rs>
__device__ __forceinline__ void keccakf1600_unrolled_impl(KState &a, std::integer_sequence<int, Rs...>)
{
    (keccak_round_aligned(a, (uint32_t)keccak_rc64(Rs), (uint32_t)(keccak_rc64(Rs) >> 32)), ...);
}
__device__ __forceinline__ void keccakf1600_unrolled(KState &a)
{
    keccakf1600_unrolled_impl(a, std::make_integer_sequence<int, 24>{});
}

}  // namespace capy
