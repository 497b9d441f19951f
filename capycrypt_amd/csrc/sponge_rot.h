// sponge_rot.h — body absorb for batches BETWEEN ONE AND TWO one-lane waves per SIMD (64 S < n < 128 S sponges).
//
// A batch of 64 S + x one-lane sponges (0 < x < 64 S) in one launch puts a second wave on x / 64 SIMDs.  Two waves of a
// SIMD each advance at 1 / 1.75 of a lone wave's rate (the blocked round with raised priority; 1 / 1.96 with the plain
// round), so those SIMDs finish last and the launch takes the two-waves time however small x is: 73 728 x 1 MiB ran at
// 826 GB/s where 65 536 ran at 1167 and 131 072 at 1340 (profiles/r03_chipfull.txt) -- 12.5 % more work, 59 % more time.
//
// The remedy is the rotation of sponge_mixed.h applied to OCCUPANCY instead of lane width: the batch is cut into groups
// of 256 sponges (four one-lane waves = one compute unit's worth at one wave per SIMD); in each of P phase launches
// 2 Cp groups run DOUBLED UP -- two groups on one compute unit, two waves per SIMD, nb2 blocks each -- while the other
// groups have a compute unit to themselves and absorb nb1 = 1.5 nb2 blocks; the roles rotate so that after P phases
// every group has been doubled up in exactly `a` of them, and the states cross phases through the word-major buffer the
// mixed schedule uses (200 B per sponge and phase).  All waves of a phase finish together, and the batch takes
//     T = T2 / (f + ratio (1 - f)),   f = a / P = 2 Cp / (C + Cp)      (C compute units, T2 the two-waves time)
// instead of T2.  Measured (profiles/r04_chipfull.txt): 66 048 / 73 728 / 81 920 / 98 304 / 114 688 x 1 MiB
// 94.2 / 93.5 / 94.0 / 95.3 / 97.6 ms -> 61.8 / 63.3 / 68.5 / 80.2 / 91.4 ms (735 ... 1232 -> 1120 ... 1316 GB/s).
//
// Placement is made explicit rather than left to the dispatcher: a workgroup is 512 lanes and the kernel is compiled for
// exactly two waves per SIMD (amdgpu_waves_per_eu(2, 2): the register allocation is rounded up so that a third wave
// does not fit), so every compute unit holds exactly one workgroup -- eight waves, two per SIMD, when its role is
// "doubled up"; in the other role waves 4..7 leave at once and four waves run one per SIMD on the unrolled plain round
// (the lone-wave form).  The doubled-up role runs the ROLLED blocked round with priority: two compute units share a
// 64 KB instruction cache, and with both roles unrolled (2 x 35 KB) neighbours of different roles evicted each other --
// 77.5 instead of 65.5 ms at 73 728 x 1 MiB (CAPY_ROT_DOUBLED_ROLLED=0 for the A/B).
//
// Scope as for sponge_mixed.h: the uniform digest absorb (equal lengths, fixed stride, 8-byte aligned); heads before,
// tail / padding / squeeze after, by the generic kernel through SpongeParams::head_state / resume_state.
#pragma once
#include "sponge_kernels.h"

namespace capy {

#ifndef CAPY_ROT_DOUBLED_ROLLED
#define CAPY_ROT_DOUBLED_ROLLED 1
#endif

struct RotParams {
    const uint8_t *msgs;
    uint64_t msg_stride;
    uint64_t n;
    uint64_t *state;  // [25][n_pad] words, word-major
    uint64_t n_pad;
    uint64_t init_state[25];
    uint32_t load_state;  // 0: first phase and no head blocks: start from init_state
    uint32_t phase;       // this launch's phase, 0 .. P - 1
    uint32_t Cp, G;       // doubled-up compute units per phase; groups of 256 sponges (G = C + Cp; groups past n are empty)
    uint32_t nb1, nb2;    // blocks per phase of a group on its own / doubled up
};

// blocks of group g absorbed before phase `phase`
__device__ __forceinline__ uint32_t rot_blocks_before(const RotParams &q, uint32_t g, uint32_t phase)
{
    uint32_t done = 0, rho = g;
    for (uint32_t p = 0; p < phase; p++) {
        done += rho < 2 * q.Cp ? q.nb2 : q.nb1;
        rho += 2 * q.Cp;
        if (rho >= q.G) rho -= q.G;
    }
    return done;
}

template <int RW, bool DOUBLED>
__device__ __forceinline__ void rot_body(const RotParams &q, uint64_t item0, uint32_t first, uint32_t nf)
{
    constexpr uint32_t RB = RW * 8;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t item = item0 + lane;
    const bool active = item < q.n;
    const uint64_t it = active ? item : q.n - 1;  // lanes past the batch redo the last sponge and store nothing
    KState a;
    if (q.load_state) {
#pragma unroll
        for (int i = 0; i < 25; i++) {
            const uint64_t v = q.state[(uint64_t)i * q.n_pad + it];
            a.lo[i] = (uint32_t)v;
            a.hi[i] = (uint32_t)(v >> 32);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 25; i++) {
            a.lo[i] = (uint32_t)q.init_state[i];
            a.hi[i] = (uint32_t)(q.init_state[i] >> 32);
        }
    }
    if (nf) {
        const uint8_t *mine = q.msgs + it * q.msg_stride + (uint64_t)first * RB;
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
            if constexpr (DOUBLED) {
#if CAPY_ROT_DOUBLED_ROLLED
                keccakf1600_paired<CAPY_PAIRED_PRIO>(a);
#else
                keccakf1600_paired_unrolled<CAPY_PAIRED_PRIO>(a);
#endif
            } else {
                keccakf1600_unrolled(a);
            }
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 25; i++) q.state[(uint64_t)i * q.n_pad + item] = state_word(a, i);
    }
}

template <int RW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void sponge_rot_kernel(const RotParams q)
{
    const uint32_t b = blockIdx.x, wave = threadIdx.x >> 6;
    const bool doubled = b < q.Cp;  // workgroup-uniform
    if (!doubled && wave >= 4) return;
    // role index of this half workgroup in the rotation, and the group that holds it in this phase
    const uint32_t rho = doubled ? 2 * b + (wave >> 2) : q.Cp + b;  // 2 Cp + (b - Cp)
    const uint32_t shift = (uint32_t)(((uint64_t)q.phase * 2 * q.Cp) % q.G);
    const uint32_t g = rho >= shift ? rho - shift : rho + q.G - shift;
    const uint64_t item0 = (uint64_t)g * 256 + (wave & 3) * 64;
    if (item0 >= q.n) return;  // wave-uniform: an empty group / the idle part of the last one
    const uint32_t first = rot_blocks_before(q, g, q.phase);
    if (doubled)
        rot_body<RW, true>(q, item0, first, q.nb2);
    else
        rot_body<RW, false>(q, item0, first, q.nb1);
}

}  // namespace capy
