// sponge_kernels.h — batched keccak sponge for gfx950: one sponge per lane.
//
// One kernel covers the reference's whole sponge path
//   shake / cshake / kmac_xof      /root/reference/src/sha3/shake_functions.rs:24-89
//   sponge_absorb / sponge_squeeze /root/reference/src/sha3/sponge.rs:10-34
//   xor_bytes (keystream ^ msg)    /root/reference/src/sha3/aux_functions.rs:90-93
// by treating every call as the absorption of a per-item byte stream
//     head (bytepad(encode_string(K_i), w), built in-kernel)  ||  body (message)  ||  suffix  ||  pad
// on top of an initial state that already contains the batch-shared prefix block(s)
// (bytepad(encode_string(N) || encode_string(S), w)), followed by a squeeze that either writes
// `out_len` bytes per item or XORs a len_i-byte keystream into the message in place.
//
// Data movement: each lane owns one sponge (25 x u64 in 50 VGPRs), but messages are contiguous
// per item, so rate-sized blocks are fetched by the whole wave with coalesced 8-byte loads
// (17..21 consecutive lanes cover one message block), staged through LDS ([item][word] layout,
// bank-conflict-free for the per-lane ds_read_b64 at stride RW*8), and prefetched one block ahead
// in registers so the HBM latency hides under the 24 rounds.
#pragma once
#include "keccak_dev.h"

namespace capy {

struct SpongeParams {
    uint64_t init_state[25];  // state after the batch-shared prefix (zeros for SHA3)
    // batch-shared prefix bytes that could NOT be folded into init_state (only when the prefix is
    // not a whole number of absorb blocks, i.e. cSHAKE/KMAC at D224 where r = 172 but 168 B are consumed)
    const uint8_t *pre;
    uint32_t pre_len;
    // per-item head = hdr bytes || key bytes || zeros up to head_len   (head_len = 0: no head)
    const uint8_t *keys;
    uint64_t key_stride;
    uint32_t key_len;
    uint32_t hdr_len;
    uint64_t hdr0, hdr1;  // up to 16 header bytes, little-endian packed
    uint32_t head_len;
    // body / xor target
    const uint8_t *msgs;
    const uint64_t *offsets;  // n+1 byte offsets into msgs, or null: item i at msgs + i*msg_stride
    const uint64_t *lens;     // optional n lengths (aligned re-packed batches); null: offsets[i+1]-offsets[i]
    uint64_t msg_stride;
    uint64_t uniform_len;
    uint32_t absorb_body;  // 0: the body is not absorbed (keystream mode: X = "")
    // trailer
    uint64_t suffix;  // up to 8 suffix bytes, little-endian packed
    uint32_t suffix_len;
    uint32_t sha3_suffix_rule;  // reference shake(): first suffix byte is 0x86 iff len % 136 == 135
    uint32_t fips_pad;          // 0: reference pad rule (pad only if unaligned); 1: FIPS 202 pad10*1
    uint32_t stride_bytes;      // the reference's `r` (172 for cSHAKE/KMAC at D224), else 8*RW
    // output
    uint32_t out_mode;  // 0: write out_len bytes per item; 1: XOR keystream into msgs in place
    uint32_t sq_words;  // words emitted per squeeze block
    uint8_t *out;
    uint64_t out_stride;
    uint32_t out_len;
    const int32_t *mask;  // optional: only items with mask[i] != 0 are processed
    uint64_t n;
};

struct ItemCtx {
    const uint8_t *key;
    const uint8_t *msg;
    uint64_t len;     // absorbed body length
    uint64_t padded;  // head + body + suffix + pad
    uint64_t suffix;
    bool pad80;
};

__device__ __forceinline__ uint32_t stream_byte(const SpongeParams &p, const ItemCtx &c, uint64_t pos)
{
    uint32_t v = 0;
    if (pos < p.pre_len) {
        v = p.pre[pos];
        if (c.pad80 && pos + 1 == c.padded) v |= 0x80;
        return v;
    }
    pos -= p.pre_len;
    if (pos < p.head_len) {
        if (pos < p.hdr_len) {
            v = (uint32_t)((pos < 8 ? p.hdr0 >> (8 * pos) : p.hdr1 >> (8 * (pos - 8))) & 0xff);
        } else {
            uint64_t k = pos - p.hdr_len;
            if (k < p.key_len) v = c.key[k];
        }
    } else {
        uint64_t q = pos - p.head_len;
        if (q < c.len) {
            v = c.msg[q];
        } else {
            q -= c.len;
            if (q < p.suffix_len) v = (uint32_t)((c.suffix >> (8 * q)) & 0xff);
        }
    }
    if (c.pad80 && pos + p.pre_len + 1 == c.padded) v |= 0x80;
    return v;
}

__device__ __forceinline__ uint64_t stream_word(const SpongeParams &p, const ItemCtx &c, uint64_t pos)
{
    const uint64_t body0 = (uint64_t)p.pre_len + p.head_len;
    // whole word inside the body and 8-byte aligned in memory: one load
    if (pos >= body0 && pos + 8 <= body0 + c.len) {
        const uint8_t *a = c.msg + (pos - body0);
        if (((uintptr_t)a & 7) == 0) return *reinterpret_cast<const uint64_t *>(a);
    }
    // whole word inside the zero fill (between suffix and the final pad byte)
    if (pos >= body0 + c.len + p.suffix_len && pos + 8 < c.padded) return 0;
    uint64_t w = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) w |= (uint64_t)stream_byte(p, c, pos + j) << (8 * j);
    return w;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// FULLCHIP = false: launches that cannot fill the chip (<= 2 waves per SIMD).  A lone wave hides nothing, so the
//   hot loops carry the permutation fully unrolled with literal round constants and prefetch the next block
//   into registers (measured: 905 vs 734..825 GB/s-equivalent at 768 waves, profiles/r01_keccak_loop_forms.txt).
// FULLCHIP = true: launches with many waves per SIMD.  Throughput is VALU issue; the rolled permutation with
//   constants fetched one trip ahead is the fastest form there (10.7 vs 9.1 G permutations/s) and the
//   register budget is kept low for occupancy (no register prefetch: other waves cover the latency).
template <bool FULLCHIP>
__device__ __forceinline__ void keccak_hot(KState &a)
{
    if constexpr (FULLCHIP)
        keccakf1600_pipelined(a);
    else
        keccakf1600_unrolled(a);
}
template <bool FULLCHIP>
__device__ __forceinline__ void keccak_cold(KState &a)
{
    if constexpr (FULLCHIP)
        keccakf1600_pipelined(a);
    else
        keccakf1600(a);
}

template <int RW, bool FULLCHIP>
__global__ __launch_bounds__(64) void sponge_kernel(const SpongeParams p)
{
    constexpr uint32_t RB = RW * 8;
    __shared__ uint64_t s_stage[64 * RW];
    __shared__ uint64_t s_base[64];
    __shared__ uint32_t s_nfull[64];

    const uint32_t lane = threadIdx.x;
    const uint64_t item = (uint64_t)blockIdx.x * 64 + lane;
    const bool active = item < p.n && (p.mask == nullptr || p.mask[item] != 0);

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
        c.key = p.keys + item * p.key_stride;
    }
    c.len = p.absorb_body ? tgt_len : 0;
    c.suffix = p.suffix;
    if (p.sha3_suffix_rule && (c.len % 136) == 135) c.suffix = (p.suffix & ~0xffULL) | 0x86;
    const uint64_t total = (uint64_t)p.pre_len + p.head_len + c.len + p.suffix_len;
    const uint32_t rem = (uint32_t)(total % p.stride_bytes);
    c.pad80 = p.fips_pad || rem != 0;
    c.padded = rem ? total + (p.stride_bytes - rem) : total;
    const uint32_t nb = active ? (uint32_t)(c.padded / p.stride_bytes) : 0;

    const bool grid_aligned = ((p.pre_len + p.head_len) % RB == 0) && (p.stride_bytes == RB);  // wave-uniform
    const uint32_t hb = grid_aligned ? (p.pre_len + p.head_len) / RB : 0;
    const bool msg_aligned = active && grid_aligned && (((uintptr_t)c.msg & 7) == 0);
    const uint32_t nfull = (msg_aligned && p.absorb_body) ? (uint32_t)(c.len / RB) : 0;

    KState a;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        a.lo[i] = (uint32_t)p.init_state[i];
        a.hi[i] = (uint32_t)(p.init_state[i] >> 32);
    }

    // ---------------- phase H: per-item head blocks (uniform trip count)
    for (uint32_t b = 0; b < hb; b++) {
        if (active) {
#pragma unroll
            for (int w = 0; w < RW; w++) {
                uint64_t v = stream_word(p, c, (uint64_t)b * RB + 8 * w);
                a.lo[w] ^= (uint32_t)v;
                a.hi[w] ^= (uint32_t)(v >> 32);
            }
            keccak_cold<FULLCHIP>(a);
        }
    }

    // ---------------- phase B: full body blocks, wave-cooperative coalesced loads through LDS
    s_base[lane] = (uint64_t)(uintptr_t)c.msg;
    s_nfull[lane] = nfull;
    __syncthreads();
    const uint32_t max_full = wave_max_u32(nfull);
    if (!FULLCHIP && max_full) {
        // source pointer and block limit of every (load slot, lane) pair, hoisted out of the block loop
        const uint8_t *src[RW];
        uint32_t lim[RW];
#pragma unroll
        for (int k = 0; k < RW; k++) {
            const uint32_t i = k * 64 + lane;
            const uint32_t m = i / RW, w = i - m * RW;
            lim[k] = s_nfull[m];
            src[k] = reinterpret_cast<const uint8_t *>(s_base[m]) + 8 * w;
        }
        const uint8_t *safe = p.msgs;  // readable whenever any message of this wave has a full block
        uint64_t pf[RW];
        auto coop_load = [&](uint32_t t) {
#pragma unroll
            for (int k = 0; k < RW; k++) {
                const uint8_t *q = t < lim[k] ? src[k] + (uint64_t)t * RB : safe;
                pf[k] = *reinterpret_cast<const uint64_t *>(q);
            }
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
                for (int w = 0; w < RW; w++) {
                    a.lo[w] ^= (uint32_t)wv[w];
                    a.hi[w] ^= (uint32_t)(wv[w] >> 32);
                }
                keccak_hot<FULLCHIP>(a);
            }
        }
    }

    if (FULLCHIP && max_full) {
        // no register prefetch, no hoisted address arrays: keeps the kernel at 4 waves per SIMD
        for (uint32_t t = 0; t < max_full; t++) {
#pragma unroll
            for (int k = 0; k < RW; k++) {
                const uint32_t i = k * 64 + lane;
                const uint32_t m = i / RW, w = i - m * RW;
                const bool in = t < s_nfull[m];
                const uint8_t *q = in ? reinterpret_cast<const uint8_t *>(s_base[m]) + (uint64_t)t * RB + 8 * w : p.msgs;
                s_stage[i] = *reinterpret_cast<const uint64_t *>(q);
            }
            __syncthreads();
            if (t < nfull) {
#pragma unroll
                for (int w = 0; w < RW; w++) {
                    const uint64_t v = s_stage[lane * RW + w];
                    a.lo[w] ^= (uint32_t)v;
                    a.hi[w] ^= (uint32_t)(v >> 32);
                }
            }
            __syncthreads();
            if (t < nfull) keccak_hot<FULLCHIP>(a);
        }
    }

    // ---------------- phase T: remaining blocks (tail of the body, suffix, pad) byte-granular
    {
        const uint32_t first = hb + nfull;
        const uint32_t cnt = nb > first ? nb - first : 0;
        const uint32_t max_cnt = wave_max_u32(cnt);
        for (uint32_t j = 0; j < max_cnt; j++) {
            if (j < cnt) {
                const uint64_t base = (uint64_t)(first + j) * RB;
#pragma unroll
                for (int w = 0; w < RW; w++) {
                    uint64_t v = stream_word(p, c, base + 8 * w);
                    a.lo[w] ^= (uint32_t)v;
                    a.hi[w] ^= (uint32_t)(v >> 32);
                }
                keccak_cold<FULLCHIP>(a);
            }
        }
    }

    // ---------------- squeeze
    if (p.out_mode == 0) {
        uint8_t *o = active ? p.out + item * p.out_stride : nullptr;
        uint32_t produced = 0;
        while (produced < p.out_len) {
#pragma unroll
            for (int w = 0; w < 25; w++) {
                if ((uint32_t)w < p.sq_words) {
                    if (active && produced < p.out_len) {
                        uint64_t v = ((uint64_t)a.hi[w] << 32) | a.lo[w];
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
            if (produced < p.out_len) keccak_hot<FULLCHIP>(a);
        }
    } else {
        // keystream XOR in place: msg[i] ^= squeeze(len) ; squeeze block = RW words (cSHAKE/KMAC only)
        const uint32_t xfull = msg_aligned ? (uint32_t)(tgt_len / RB) : 0;
        __syncthreads();
        s_nfull[lane] = xfull;
        __syncthreads();
        const uint32_t max_x = wave_max_u32(xfull);
        if (max_x) {
            uint8_t *dst[RW];
            uint32_t lim[RW];
#pragma unroll
            for (int k = 0; k < RW; k++) {
                const uint32_t i = k * 64 + lane;
                const uint32_t m = i / RW, w = i - m * RW;
                lim[k] = s_nfull[m];
                dst[k] = reinterpret_cast<uint8_t *>(s_base[m]) + 8 * w;
            }
            uint8_t *safe = const_cast<uint8_t *>(p.msgs);
            uint64_t pf[RW];
            auto coop_load = [&](uint32_t t) {
#pragma unroll
                for (int k = 0; k < RW; k++) {
                    const uint8_t *q = t < lim[k] ? dst[k] + (uint64_t)t * RB : safe;
                    pf[k] = *reinterpret_cast<const uint64_t *>(q);
                }
            };
            coop_load(0);
            for (uint32_t t = 0; t < max_x; t++) {
#pragma unroll
                for (int k = 0; k < RW; k++) s_stage[k * 64 + lane] = pf[k];
                __syncthreads();
                if (t < xfull) {
#pragma unroll
                    for (int w = 0; w < RW; w++) s_stage[lane * RW + w] ^= ((uint64_t)a.hi[w] << 32) | a.lo[w];
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < RW; k++) {
                    const uint64_t v = s_stage[k * 64 + lane];
                    if (t < lim[k]) *reinterpret_cast<uint64_t *>(dst[k] + (uint64_t)t * RB) = v;
                }
                __syncthreads();
                if (t + 1 < max_x) coop_load(t + 1);
                if (t < xfull && (uint64_t)(t + 1) * RB < tgt_len) keccak_hot<FULLCHIP>(a);
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
                    uint64_t v = ((uint64_t)a.hi[w] << 32) | a.lo[w];
                    for (int b = 0; b < 8; b++)
                        if (pos + 8 * w + b < tgt_len) m[pos + 8 * w + b] ^= (uint8_t)(v >> (8 * b));
                }
                pos += RB;
                if (pos < tgt_len) keccak_cold<FULLCHIP>(a);
            }
        }
    }
}

}  // namespace capy
