// ed448_wave.h — Ed448 scalar multiplication with ONE ITEM PER WAVE, for small batches.
//
// The batched kernels (ed448.hip: vb / vb2 / fb / fb2 / dsm) give every item one lane: ~1.35 M dependent VALU
// instructions per variable-base multiplication, 3 ms however small the batch (a batch of one signature verification
// takes as long as 65 536 of them).  Here a wave works on one item: a field element is spread over a 16-lane row, one
// 28-bit limb per lane, and the four rows of the wave hold the four coordinates (X, Y, Z, T) of an extended point, so one
// "row vector" register is a whole point and one multiplication pass does the four independent field multiplications of
// a level of the addition / doubling formulas side by side.
//
//   field multiplication (rv_mul): a_i is broadcast inside each row (ds_swizzle), b * x^i mod p (x = 2^28,
//   x^16 = x^8 + 1) is kept rotating through the row (two fused DPP operations per step), every lane accumulates one
//   64-bit column: 16 v_mad_u64_u32 per lane, ~100 instructions, 240 ns at one wave per SIMD against 554 ns for the
//   in-lane form -- and four of them at once (tools/probe_ed448_sliced.hip checks it against fe_mul bit for bit).
//   doubling = 2 passes, addition = 3 passes, rows are moved with ds_bpermute.
//
// The algorithm is the one of ed448_algo.h step for step (same signed 5-bit windows, same table, same formulas, same
// fixed-base table), so the outputs are the same field values and hence the same bytes.  Every lane of the wave follows
// the same control flow; the scalar's digits are wave-uniform.  CT = true (r03) is the constant-address form for secret
// scalars: every row of the LDS table (variable base) / every entry of the window's row of the 5-bit hardened table
// (fixed base) is read and the wanted one kept by a wave-uniform mask, the sign is applied by masks too -- 17 extra LDS
// reads per window for the variable base (free next to ~1000 instructions of point arithmetic), 90 instead of 39
// additions for the fixed base (1.4x, the inversion dominates either way).
//
// Device-only (DPP / swizzle builtins); covered by GPU tests that compare it with the batched kernels.
#pragma once
#include "ed448_algo.h"

namespace capy {
namespace wave {

typedef uint32_t RV;  // one limb per lane: lane = 16 * row + limb; four field elements per register

struct WC {            // per-lane constants
    uint32_t l4;       // 4 * limb index (byte address of my limb inside a row for ds_bpermute)
    uint32_t m8, m89, m8b;  // all-ones in limb lane 8 / lanes 8 and 9 / lanes 8..11
    uint32_t r0, r1, r2, r3;  // all-ones in row 0 / 1 / 2 / 3
    uint32_t twop, fourp;     // my limb of 2p and 4p
    uint32_t one;             // my limb of the field element 1
};

__device__ __forceinline__ WC wc_init()
{
    const uint32_t lane = threadIdx.x & 63, l = lane & 15, row = lane >> 4;
    WC c;
    c.l4 = l * 4;
    c.m8 = l == 8 ? ~0u : 0u;
    c.m89 = (l == 8 || l == 9) ? ~0u : 0u;
    c.m8b = (l >= 8 && l <= 11) ? ~0u : 0u;
    c.r0 = row == 0 ? ~0u : 0u;
    c.r1 = row == 1 ? ~0u : 0u;
    c.r2 = row == 2 ? ~0u : 0u;
    c.r3 = row == 3 ? ~0u : 0u;
    c.twop = l == 8 ? 2 * (M28 - 1) : 2 * M28;
    c.fourp = 2 * c.twop;
    c.one = l == 0 ? 1u : 0u;
    // keep the masks as AND operands (the compiler would otherwise re-derive the conditions and use v_cndmask)
    asm volatile("" : "+v"(c.m8), "+v"(c.m89), "+v"(c.m8b), "+v"(c.r0), "+v"(c.r1), "+v"(c.r2), "+v"(c.r3));
    return c;
}

template <int N>
__device__ __forceinline__ uint32_t row_ror(uint32_t v)  // lane k of every row <- lane (k - N) mod 16 of that row
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xF, 0xF, true);
}
template <int I>
__device__ __forceinline__ uint32_t limb_bcast(uint32_t v)  // every lane of a row <- lane I of that row
{
    // ds_swizzle bit mode: lane' = ((lane & and) | or) ^ xor inside each group of 32; and = 0x10 keeps the row
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x10 | (I << 5));
}
template <int R>
__device__ __forceinline__ RV row_bcast(const WC &c, RV v)  // every row <- row R
{
    return (RV)__builtin_amdgcn_ds_bpermute((int)(c.l4 + R * 64), (int)v);
}

// carry-save normalisation: limbs < 2^31 in, limbs <= 2^28 + 8 out (fe_weak_reduce)
__device__ __forceinline__ RV rv_weak(const WC &c, RV x)
{
    const uint32_t cy = x >> 28;
    return (x & M28) + row_ror<1>(cy) + (row_ror<9>(cy) & c.m8);
}
__device__ __forceinline__ RV rv_add_nr(RV a, RV b) { return a + b; }
__device__ __forceinline__ RV rv_sub_nr(const WC &c, RV a, RV b) { return a + c.twop - b; }
__device__ __forceinline__ RV rv_neg_nr(const WC &c, RV a) { return c.twop - a; }
__device__ __forceinline__ RV rv_sub(const WC &c, RV a, RV b) { return rv_weak(c, a + c.twop - b); }
__device__ __forceinline__ RV rv_sub4(const WC &c, RV a, RV b) { return rv_weak(c, a + c.fourp - b); }

// Row-wise a * b mod p.  The same polynomial product folded with the same x^16 = x^8 + 1 as fe_mul, so a column
// collects the same products with the same multiplicities: exact while 38 * max_limb(a) * max_limb(b) < 2^64, and every
// operand bound audited for the point formulas of ed448_dev.h carries over (the limbs of b * x^i are sums of at most
// three limbs of b: < 2^32).  Output limbs <= 2^28 + 8.
__device__ __forceinline__ RV rv_mul(const WC &c, RV a, RV b)
{
    uint32_t ai[16];
    ai[0] = limb_bcast<0>(a);
    ai[1] = limb_bcast<1>(a);
    ai[2] = limb_bcast<2>(a);
    ai[3] = limb_bcast<3>(a);
    ai[4] = limb_bcast<4>(a);
    ai[5] = limb_bcast<5>(a);
    ai[6] = limb_bcast<6>(a);
    ai[7] = limb_bcast<7>(a);
    ai[8] = limb_bcast<8>(a);
    ai[9] = limb_bcast<9>(a);
    ai[10] = limb_bcast<10>(a);
    ai[11] = limb_bcast<11>(a);
    ai[12] = limb_bcast<12>(a);
    ai[13] = limb_bcast<13>(a);
    ai[14] = limb_bcast<14>(a);
    ai[15] = limb_bcast<15>(a);
    // B[i] = b * x^i mod p in four interleaved chains (steps of x^4), so that the DPP moves of one chain cover the
    // latency of the others:  * x: limb 15 -> limbs 0 and 8;  * x^2: limbs 14, 15 -> 0, 1 and 8, 9;  * x^4: 12..15 -> 0..3 and 8..11
    uint32_t B[16];
    B[0] = b;
    B[1] = row_ror<1>(b) + (row_ror<9>(b) & c.m8);
    B[2] = row_ror<2>(b) + (row_ror<10>(b) & c.m89);
    B[3] = row_ror<1>(B[2]) + (row_ror<9>(B[2]) & c.m8);
#pragma unroll
    for (int i = 4; i < 16; i++) B[i] = row_ror<4>(B[i - 4]) + (row_ror<12>(B[i - 4]) & c.m8b);
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += (uint64_t)ai[i] * B[i];
    const uint32_t lo = (uint32_t)acc & M28, mid = (uint32_t)(acc >> 28) & M28, hi = (uint32_t)(acc >> 56);
    const uint32_t x = lo + row_ror<1>(mid) + row_ror<2>(hi) + (row_ror<9>(mid) & c.m8) + (row_ror<10>(hi) & c.m89);
    return rv_weak(c, x);
}

// a * (-39081) mod p  (fe_mul_d), limbs of a <= 2^29
__device__ __forceinline__ RV rv_mul_d(const WC &c, RV a)
{
    const uint64_t v = (uint64_t)a * ED448_D_ABS;
    const uint32_t lo = (uint32_t)v & M28, hi = (uint32_t)(v >> 28);
    const RV t = rv_weak(c, lo + row_ror<1>(hi) + (row_ror<9>(hi) & c.m8));
    return rv_weak(c, c.twop - t);
}

__device__ __forceinline__ RV rv_identity(const WC &c) { return c.one & (c.r1 | c.r2); }  // (0, 1, 1, 0)

// the second-level operands of both formulas: rows (E, G, F, E) x (F, H, G, H) = (X3, Y3, Z3, T3)
__device__ __forceinline__ RV rv_finish(const WC &c, RV E, RV F, RV G, RV H)
{
    const RV L = (E & (c.r0 | c.r3)) | (G & c.r1) | (F & c.r2);
    const RV R = (F & c.r0) | (H & (c.r1 | c.r3)) | (G & c.r2);
    return rv_mul(c, L, R);
}

// pt_add_cached: P = (X1, Y1, Z1, T1), Q = (X2, Y2, Z2, d T2) in rows 0..3; limbs of Q rows 0 and 3 <= 2^29
__device__ __forceinline__ RV wv_add(const WC &c, RV P, RV Q)
{
    const RV M = rv_mul(c, P, Q);  // (A, B, D, C)
    const RV u = row_bcast<0>(c, P) + row_bcast<1>(c, P);  // X1 + Y1 in every row
    const RV v = row_bcast<0>(c, Q) + row_bcast<1>(c, Q);  // X2 + Y2
    RV E = rv_mul(c, u, v);
    const RV A = row_bcast<0>(c, M), B = row_bcast<1>(c, M), D = row_bcast<2>(c, M), C = row_bcast<3>(c, M);
    E = rv_sub(c, rv_sub_nr(c, E, A), B);
    const RV F = rv_sub_nr(c, D, C);
    const RV G = rv_add_nr(D, C);
    const RV H = rv_sub_nr(c, B, A);
    return rv_finish(c, E, F, G, H);
}

// pt_dbl<true>
__device__ __forceinline__ RV wv_dbl(const WC &c, RV P)
{
    const RV xy = row_bcast<0>(c, P) + row_bcast<1>(c, P);
    const RV V = (P & ~c.r3) | (xy & c.r3);  // (X, Y, Z, X + Y)
    const RV S = rv_mul(c, V, V);            // (A, B, Z^2, (X + Y)^2)
    const RV A = row_bcast<0>(c, S), B = row_bcast<1>(c, S), Z2 = row_bcast<2>(c, S), E0 = row_bcast<3>(c, S);
    const RV C = rv_add_nr(Z2, Z2);
    const RV E = rv_sub(c, rv_sub_nr(c, E0, A), B);
    const RV G = rv_add_nr(A, B);
    const RV F = rv_sub4(c, G, C);
    const RV H = rv_sub_nr(c, A, B);
    return rv_finish(c, E, F, G, H);
}

// cached form of a point: (X, Y, Z, d T)
__device__ __forceinline__ RV wv_cached(const WC &c, RV P) { return (P & ~c.r3) | (rv_mul_d(c, P) & c.r3); }
// -(x, y) = (-x, y): negate rows 0 and 3 of a cached point when neg is set (wave-uniform)
__device__ __forceinline__ RV wv_cond_neg(const WC &c, RV Q, bool neg)
{
    const RV nq = (Q & (c.r1 | c.r2)) | (rv_neg_nr(c, Q) & (c.r0 | c.r3));
    return neg ? nq : Q;
}

// a^(p-2), the chain of fe_inv, in every row
__device__ __forceinline__ RV rv_sqrn(const WC &c, RV a, int n)
{
#pragma unroll 1
    for (int i = 0; i < n; i++) a = rv_mul(c, a, a);
    return a;
}
__device__ __forceinline__ RV rv_inv(const WC &c, RV a)
{
    const RV x2 = rv_mul(c, rv_mul(c, a, a), a);
    const RV x3 = rv_mul(c, rv_mul(c, x2, x2), a);
    const RV x6 = rv_mul(c, rv_sqrn(c, x3, 3), x3);
    const RV x9 = rv_mul(c, rv_sqrn(c, x6, 3), x3);
    const RV x18 = rv_mul(c, rv_sqrn(c, x9, 9), x9);
    const RV x19 = rv_mul(c, rv_mul(c, x18, x18), a);
    const RV x37 = rv_mul(c, rv_sqrn(c, x19, 18), x18);
    const RV x74 = rv_mul(c, rv_sqrn(c, x37, 37), x37);
    const RV x111 = rv_mul(c, rv_sqrn(c, x74, 37), x37);
    const RV x222 = rv_mul(c, rv_sqrn(c, x111, 111), x111);
    const RV x223 = rv_mul(c, rv_mul(c, x222, x222), a);
    const RV t = rv_mul(c, rv_sqrn(c, x223, 223), x222);
    return rv_mul(c, rv_sqrn(c, t, 2), a);
}

// limb k of the 56-byte little-endian integer at `in`
__device__ __forceinline__ uint32_t limb_from_bytes(const uint8_t *in, uint32_t k)
{
    const uint32_t bit = 28 * k, b0 = bit >> 3;
    uint64_t v = 0;
#pragma unroll
    for (uint32_t b = 0; b < 5; b++)
        if (b0 + b < 56) v |= (uint64_t)in[b0 + b] << (8 * b);
    return (uint32_t)(v >> (bit & 7)) & M28;
}

// pt_from_affine_bytes: (x, y, 1, x y)
__device__ __forceinline__ RV wv_from_affine_bytes(const WC &c, const uint8_t *xy)
{
    const uint32_t l = threadIdx.x & 15;
    const RV x = limb_from_bytes(xy, l), y = limb_from_bytes(xy + 56, l);  // x and y in every row
    const RV t = rv_mul(c, x, y);
    return (x & c.r0) | (y & c.r1) | (c.one & c.r2) | (t & c.r3);
}

// pt_to_affine_bytes through LDS.  CAPY_ED448_WAVE_GCD_INV = 1 (default): the point's four rows are staged, every lane
// reads Z as a whole field element and runs the division-step inversion of ed448_dev.h in the LANE form (wave-uniform
// work, ~40 000 instructions against 458 row-form multiplications of ~200 ns each), and lanes 0 and 1 multiply x and y
// out and finish them with the in-lane canonical reduction.  0: the a^(p-2) chain in row form (A/B).
#ifndef CAPY_ED448_WAVE_GCD_INV
#define CAPY_ED448_WAVE_GCD_INV 1
#endif
__device__ __forceinline__ void wv_to_affine_bytes(const WC &c, uint8_t *xy, RV P, uint32_t *stage /* [64] */)
{
    const uint32_t lane = threadIdx.x & 63;
#if CAPY_ED448_WAVE_GCD_INV
    stage[lane] = P;
    __syncthreads();
    Fe z;
#pragma unroll
    for (int i = 0; i < 16; i++) z.l[i] = stage[32 + i];
    const Fe zi = fe_inv_gcd(z);
    if (lane < 2) {
        Fe f;
#pragma unroll
        for (int i = 0; i < 16; i++) f.l[i] = stage[lane * 16 + i];
        fe_to_bytes(xy + 56 * lane, fe_mul(f, zi));
    }
    __syncthreads();
#else
    const RV zi = rv_inv(c, row_bcast<2>(c, P));
    const RV r = rv_mul(c, P, zi);  // (x, y, 1, t)
    if (lane < 32) stage[lane] = r;
    __syncthreads();
    if (lane < 2) {
        Fe f;
#pragma unroll
        for (int i = 0; i < 16; i++) f.l[i] = stage[lane * 16 + i];
        fe_to_bytes(xy + 56 * lane, f);
    }
    __syncthreads();
#endif
}

// ---- variable base: table {0..16} P (cached) in LDS, one row vector per entry
struct VbTable {
    uint32_t e[TAB_ENTRIES][64];
};
__device__ __forceinline__ void wv_build_table(const WC &c, VbTable &t, RV P)
{
    const uint32_t lane = threadIdx.x & 63;
    const RV Pc = wv_cached(c, P);
    RV acc = rv_identity(c);
#pragma unroll 1
    for (int j = 0; j < TAB_ENTRIES; j++) {
        t.e[j][lane] = wv_cached(c, acc);
        if (j + 1 < TAB_ENTRIES) acc = wv_add(c, acc, Pc);
    }
}
// -(x, y) by masks, no select on the (secret) sign
__device__ __forceinline__ RV wv_cond_neg_ct(const WC &c, RV Q, bool neg)
{
    const uint32_t nm = ct_mask(neg) & (c.r0 | c.r3);  // rows 0 and 3 of a negative digit
    return (Q & ~nm) | (rv_neg_nr(c, Q) & nm);
}
template <bool CT>
__device__ __forceinline__ RV wv_add_digit(const WC &c, RV acc, const VbTable &t, int digit)
{
    const bool neg = digit < 0;
    const int idx = neg ? -digit : digit;
    const uint32_t lane = threadIdx.x & 63;
    if constexpr (CT) {
        RV Q = 0;
#pragma unroll
        for (int j = 0; j < TAB_ENTRIES; j++) Q = ct_take(Q, t.e[j][lane], ct_mask(j == idx));
        return wv_add(c, acc, wv_cond_neg_ct(c, Q, neg));
    } else {
        const RV Q = t.e[idx][lane];
        return wv_add(c, acc, wv_cond_neg(c, Q, neg));
    }
}
// vb_scalarmul
template <bool CT>
__device__ __forceinline__ RV wv_scalarmul(const WC &c, const uint8_t *k_be, RV P, VbTable &t)
{
    wv_build_table(c, t, P);
    __syncthreads();
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<WBITS>(w, k);
    sc_msb_align<WBITS>(w);
    RV acc = wv_add_digit<CT>(c, rv_identity(c), t, (int)top);
#pragma unroll 1
    for (int i = 0; i < NWIN; i++) {
#pragma unroll 1
        for (int j = 0; j < WBITS; j++) acc = wv_dbl(c, acc);
        acc = wv_add_digit<CT>(c, acc, t, sc_next_digit_msb<WBITS>(w));
    }
    return acc;
}

// ---- fixed base: the shared affine tables of ed448_algo.h (x, y, d x y per entry, 16 limbs each): the 12-bit table
// indexed by the digit, or (CT) the 5-bit hardened table with every entry of the window's row read
template <bool CT>
__device__ __forceinline__ RV wv_fb_add_digit(const WC &c, RV acc, const uint32_t *gtab, int row, int digit)
{
    const bool neg = digit < 0;
    const int idx = neg ? -digit : digit;
    const uint32_t lane = threadIdx.x & 63, l = lane & 15, r = lane >> 4;
    // rows 0, 1, 3 read x, y, d x y; row 2 is Z2 = 1
    const uint32_t off = r == 2 ? l : (r == 3 ? 32u : r * 16u) + l;
    uint32_t v;
    if constexpr (CT) {
        const uint32_t *e = gtab + (size_t)row * FBCT_ENTRIES * FB_ENTRY_DWORDS + off;
        v = 0;
#pragma unroll
        for (int j = 0; j < FBCT_ENTRIES; j++) v = ct_take(v, e[j * FB_ENTRY_DWORDS], ct_mask(j == idx));
    } else {
        v = gtab[((size_t)row * FB_TAB_ENTRIES + idx) * FB_ENTRY_DWORDS + off];
    }
    const RV Q = (v & ~c.r2) | (c.one & c.r2);
    return wv_add(c, acc, CT ? wv_cond_neg_ct(c, Q, neg) : wv_cond_neg(c, Q, neg));
}
template <bool CT>
__device__ __forceinline__ RV wv_fb_accumulate(const WC &c, RV acc, const uint8_t *k_be, const uint32_t *gtab)
{
    constexpr int W = CT ? FBCT_WBITS : FB_WBITS;
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<W>(w, k);
    acc = wv_fb_add_digit<CT>(c, acc, gtab, Win<W>::NWIN, (int)top);
#pragma unroll 1
    for (int i = 0; i < Win<W>::NWIN; i++) acc = wv_fb_add_digit<CT>(c, acc, gtab, i, sc_next_digit_lsb<W>(w));
    return acc;
}

// ---- kernels: grid = n waves
template <bool CT>
__global__ __launch_bounds__(64) void vb_wave_kernel(uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride,
                                                     const uint8_t *points_xy, uint64_t point_stride, uint8_t *out_xy)
{
    __shared__ VbTable tab;
    __shared__ uint32_t stage[64];
    const uint64_t i = blockIdx.x;
    const WC c = wc_init();
    const RV P = wv_from_affine_bytes(c, points_xy + i * point_stride);
    const RV r = wv_scalarmul<CT>(c, scalars_be + i * scalar_stride, P, tab);
    wv_to_affine_bytes(c, out_xy + i * 112, r, stage);
}

template <bool CT>  // CT: gtab is the hardened table (FBCT_WBITS-bit windows)
__global__ __launch_bounds__(64) void fb_wave_kernel(uint64_t n, const uint8_t *scalars_be, uint8_t *out_xy, const uint32_t *gtab)
{
    __shared__ uint32_t stage[64];
    const uint64_t i = blockIdx.x;
    const WC c = wc_init();
    const RV r = wv_fb_accumulate<CT>(c, rv_identity(c), scalars_be + i * 56, gtab);
    wv_to_affine_bytes(c, out_xy + i * 112, r, stage);
}

// [a]G + [b]P (double_scalarmul)
__global__ __launch_bounds__(64) void dsm_wave_kernel(uint64_t n, const uint8_t *a_be, const uint8_t *b_be, const uint8_t *points_xy,
                                                      uint8_t *out_xy, const uint32_t *gtab)
{
    __shared__ VbTable tab;
    __shared__ uint32_t stage[64];
    const uint64_t i = blockIdx.x;
    const WC c = wc_init();
    const RV P = wv_from_affine_bytes(c, points_xy + i * 112);
    RV r = wv_scalarmul<false>(c, b_be + i * 56, P, tab);  // verification: public scalars
    r = wv_fb_accumulate<false>(c, r, a_be + i * 56, gtab);
    wv_to_affine_bytes(c, out_xy + i * 112, r, stage);
}

}  // namespace wave
}  // namespace capy
