// sponge_mixed.h — body absorb for batches that fill MORE than half but LESS than all of the chip's lanes.
//
// The headline workload is capacity-bound: 288 GB of HBM hold fewer than 56 k messages of 5 MiB, i.e. at most 870 waves
// of one-lane-per-sponge work for 1024 SIMDs, and a sixth to a quarter of the chip idles while every sponge advances at the
// one-lane rate (180 VALU per round).  The two-lane form (sponge_kernels_k2.h, 120 VALU per round) advances a sponge
// 1.48x faster but needs twice the lanes, so it cannot take the whole batch either.  This kernel runs BOTH forms
// side by side in one grid so that every SIMD holds exactly one wave: in each of P phases one group of n/P sponges is
// processed two-lanes-wide while the others run one-lane-wide, the groups rotate, and the states cross phases through
// a small HBM buffer (200 B per sponge).  Every sponge gets one fast phase of nb2 blocks and P-1 slow phases of nb1
// blocks, nb2 / nb1 = the measured speed ratio, so all waves of a phase finish together.  Expected gain over the
// one-lane kernel = (ratio + P - 1) / P: 1.16x at P = 3 (49 152 sponges on 1024 SIMDs), 1.10x at P = 5 (54 528, the
// headline batch: 852 one-lane waves would leave 172 SIMDs idle).
//
// Scope: the uniform digest absorb only (equal lengths, fixed stride, 8-byte aligned).  Per-item head blocks (KMAC
// keys) are absorbed first by a head-only launch of sponge_kernel<RW, false, 0> (SpongeParams::head_state); the tail,
// padding and squeeze are finished by the same kernel resuming from the state buffer (SpongeParams::resume_state).
// Everything else takes the generic kernels.
#pragma once
#include "sponge_kernels_k2.h"

namespace capy {

struct MixedParams {
    const uint8_t *msgs;
    uint64_t msg_stride;
    uint64_t n;
    uint64_t *state;    // [25][n_pad] words, word-major (coalesced for both lane layouts)
    uint64_t n_pad;
    uint64_t init_state[25];
    uint32_t load_state;  // 0: this is the first phase, start from init_state
    // two-lane group of this phase: items [k2_begin, k2_end), blocks [k2_first, k2_first + k2_count)
    uint64_t k2_begin, k2_end;
    uint32_t k2_waves;  // blockIdx.x < k2_waves: two-lane role; the remaining waves take 64 items each of the rest
    uint32_t k2_first, k2_count;
    // one-lane waves: items below k2_begin have had their fast phase already, items from k2_end on have not
    uint32_t k1_first_lo, k1_first_hi, k1_count;
    uint32_t staged;  // A/B (debug bit 6): take sponge_mixed_staged_kernel, the round-1 form with LDS-staged loads
};

// STAGED = false (default): every lane loads the words of its own message straight into registers, the next block in
// flight under the 24 rounds of the current one.  A lane's block lies in two or three 128-byte lines that stay in the
// CU's vector cache between its 17 loads, so HBM sees each line once (profiles/r02_pmc_summary.json: traffic 1.03x).
// STAGED = true: the wave loads 64 blocks cooperatively (coalesced) and transposes them through LDS; two barriers and
// 2 x 17 LDS operations per block cost 2-4 % on the headline (profiles/r02_direct_loads_ab.txt).
template <int RW, bool STAGED>
__device__ __forceinline__ void mixed_body_k1(const MixedParams &q, uint32_t wave, uint64_t *s_stage)
{
    constexpr uint32_t RB = RW * 8;
    const uint32_t lane = threadIdx.x;
    uint64_t item0 = (uint64_t)wave * 64;
    uint32_t first = q.k1_first_lo;
    if (item0 >= q.k2_begin) {
        item0 += q.k2_end - q.k2_begin;
        first = q.k1_first_hi;
    }
    if (item0 >= q.n) return;  // wave-uniform
    const uint64_t item = item0 + lane;
    const bool active = item < q.n;
    const uint64_t last = q.n - 1 - item0;  // clamp for the partial last wave: loads stay inside the batch

    KState a;
    if (q.load_state) {
        const uint64_t it = active ? item : q.n - 1;
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

    const uint32_t nf = q.k1_count;
    if constexpr (!STAGED) {
        if (nf) {
        const uint8_t *mine = q.msgs + (active ? item : q.n - 1) * q.msg_stride + (uint64_t)first * RB;
        uint64_t pf[RW];
#pragma unroll
        for (int w = 0; w < RW; w++) pf[w] = load_global_u64(mine + 8 * w);
        for (uint32_t t = 0; t < nf; t++) {
#pragma unroll
            for (int w = 0; w < RW; w++) {
                a.lo[w] ^= (uint32_t)pf[w];
                a.hi[w] ^= (uint32_t)(pf[w] >> 32);
            }
            if (t + 1 < nf) {
                mine += RB;
#pragma unroll
                for (int w = 0; w < RW; w++) pf[w] = load_global_u64(mine + 8 * w);
            }
            keccakf1600_unrolled(a);
        }
        }
    } else if (nf) {
        uint32_t voff[RW];
#pragma unroll
        for (int k = 0; k < RW; k++) {
            const uint32_t i = k * 64 + lane;
            uint32_t m = i / RW;
            const uint32_t w = i - m * RW;
            m = m < last ? m : (uint32_t)last;
            voff[k] = m * (uint32_t)q.msg_stride + 8 * w;
        }
        const uint8_t *wave_base = q.msgs + item0 * q.msg_stride + (uint64_t)first * RB;
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
            for (int w = 0; w < RW; w++) {
                a.lo[w] ^= (uint32_t)wv[w];
                a.hi[w] ^= (uint32_t)(wv[w] >> 32);
            }
            keccakf1600_unrolled(a);
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 25; i++) q.state[(uint64_t)i * q.n_pad + item] = ((uint64_t)a.hi[i] << 32) | a.lo[i];
    }
}

template <int RW, bool STAGED>
__device__ __forceinline__ void mixed_body_k2(const MixedParams &q, uint32_t wave, uint64_t *s_stage)
{
    constexpr uint32_t RB = RW * 8;
    constexpr int NSP = 32;
    constexpr int NLOAD = (NSP * RW + 63) / 64;
    const uint32_t lane = threadIdx.x;
    const uint32_t h = lane & 1, j = lane >> 1;
    const uint32_t hmask = 0u - h;
    const uint64_t item0 = q.k2_begin + (uint64_t)wave * NSP;
    if (item0 >= q.k2_end) return;  // wave-uniform
    const uint64_t item = item0 + j;
    const bool active = item < q.k2_end;
    const uint64_t last = q.k2_end - 1 - item0;

    KHalf a;
    uint32_t *st32 = reinterpret_cast<uint32_t *>(q.state);
    if (q.load_state) {
        const uint64_t it = active ? item : q.k2_end - 1;
#pragma unroll
        for (int i = 0; i < 25; i++) a.a[i] = st32[((uint64_t)i * q.n_pad + it) * 2 + h];
    } else {
#pragma unroll
        for (int i = 0; i < 25; i++) a.a[i] = h ? (uint32_t)(q.init_state[i] >> 32) : (uint32_t)q.init_state[i];
    }

    const uint32_t nf = q.k2_count;
    if constexpr (!STAGED) {
        if (nf) {
        // each lane of a pair loads its own 32-bit half of every word of its sponge's block, next block in flight
        const uint8_t *mine = q.msgs + (active ? item : q.k2_end - 1) * q.msg_stride + (uint64_t)q.k2_first * RB + 4 * h;
        uint32_t pf[RW];
#pragma unroll
        for (int w = 0; w < RW; w++)
            pf[w] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(reinterpret_cast<uintptr_t>(mine + 8 * w));
        for (uint32_t t = 0; t < nf; t++) {
#pragma unroll
            for (int w = 0; w < RW; w++) a.a[w] ^= pf[w];
            if (t + 1 < nf) {
                mine += RB;
#pragma unroll
                for (int w = 0; w < RW; w++)
                    pf[w] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(reinterpret_cast<uintptr_t>(mine + 8 * w));
            }
            keccakf1600_k2_unrolled(a, hmask);
        }
        }
    } else if (nf) {
        uint32_t voff[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; k++) {
            const uint32_t i = k * 64 + lane;
            uint32_t m = i / RW;
            const uint32_t w = i - m * RW;
            m = m < last ? m : (uint32_t)last;  // also folds the elements past 32 sponges onto a valid address
            voff[k] = m * (uint32_t)q.msg_stride + 8 * w;
        }
        const uint8_t *wave_base = q.msgs + item0 * q.msg_stride + (uint64_t)q.k2_first * RB;
        uint64_t pf[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; k++) pf[k] = *reinterpret_cast<const uint64_t *>(wave_base + voff[k]);
        const uint32_t *stage32 = reinterpret_cast<const uint32_t *>(s_stage);
        for (uint32_t t = 0; t < nf; t++) {
#pragma unroll
            for (int k = 0; k < NLOAD; k++) s_stage[k * 64 + lane] = pf[k];
            __syncthreads();
            uint32_t wv[RW];
#pragma unroll
            for (int w = 0; w < RW; w++) wv[w] = stage32[(j * RW + w) * 2 + h];
            __syncthreads();
            if (t + 1 < nf) {
                const uint8_t *bt = wave_base + (uint64_t)(t + 1) * RB;
#pragma unroll
                for (int k = 0; k < NLOAD; k++) pf[k] = *reinterpret_cast<const uint64_t *>(bt + voff[k]);
            }
#pragma unroll
            for (int w = 0; w < RW; w++) a.a[w] ^= wv[w];
            keccakf1600_k2_unrolled(a, hmask);
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 25; i++) st32[((uint64_t)i * q.n_pad + item) * 2 + h] = a.a[i];
    }
}

template <int RW>
__global__ __launch_bounds__(64) CAPY_WAVES_PER_SIMD(1) void sponge_mixed_kernel(const MixedParams q)
{
    if (blockIdx.x < q.k2_waves)
        mixed_body_k2<RW, false>(q, blockIdx.x, nullptr);
    else
        mixed_body_k1<RW, false>(q, blockIdx.x - q.k2_waves, nullptr);
}

template <int RW>
__global__ __launch_bounds__(64) void sponge_mixed_staged_kernel(const MixedParams q)
{
    // the two-lane role stages ceil(32 * RW / 64) * 64 words, the one-lane role 64 * RW
    __shared__ uint64_t s_stage[64 * RW];
    if (blockIdx.x < q.k2_waves)
        mixed_body_k2<RW, true>(q, blockIdx.x, s_stage);
    else
        mixed_body_k1<RW, true>(q, blockIdx.x - q.k2_waves, s_stage);
}

}  // namespace capy
