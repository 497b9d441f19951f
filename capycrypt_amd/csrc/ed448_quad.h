// ed448_quad.h — Ed448 variable-base multiplication with FOUR LANES PER ITEM (r04), for batches between the
// one-item-per-wave kernels (ed448_wave.h, efficient to ~4 k items) and the one-item-per-lane kernels (ed448.hip,
// efficient from ~64 k): a lane carries one dependent chain of ~1.27 M instructions whatever the batch size, so 8 k .. 32 k
// items -- what BASELINE configs 4 and 5 leave per GPU when they are split over eight -- took the same ~2.4 ms as 65 k.
//
// Not a split of the field multiplication (priced and rejected: Karatsuba's three half products do not divide by two and
// the recombination is not lane-symmetric, profiles/r04_ed448_two_lane.txt) but of the POINT: the four lanes of a quad
// hold X, Y, Z, T of the accumulator, a whole field element each (16 x 28-bit limbs in registers), and the independent
// field multiplications of one level of the doubling / addition formulas run side by side, each lane a complete fe_mul /
// fe_sqr on its own operands.  Operands cross lanes only BETWEEN levels, by v_mov_b32_dpp quad_perm (one instruction moves
// a limb for all four lanes at once, every lane from a source of its own):
//   doubling   level 1: (X, Y, Z, X + Y)^2          -> A, B, C', S          1 fe_sqr
//              between: G = A + B, H = A - B, E = S - G, F = G - 2 C'       3 permutes + limb arithmetic
//              level 2: (E, G, F, H) x (F, H, G, E)  -> X3, Y3, Z3, T3       1 permute + 1 fe_mul
//   addition   level 1: (X1, Y1, Z1, T1) x (X2, Y2, Z2, dT2)  -> A, B, D, C      1 fe_mul (each lane reads ITS field of the entry)
//              level 1b: (X1, Y1) x (Y2, X2)          -> the two halves of E    1 permute + 1 fe_mul (lanes 2, 3 idle)
//              between: E = sum, G = D + C, F = D - C, H = B - A            3 permutes + limb arithmetic
//              level 2: as for the doubling
// ~870 instructions per doubling and ~1100 per addition in one lane's stream against 2056 and 3023 for one item per lane:
// the chain of an item is ~2.4x shorter (the final inversion, ~40 k instructions, does not shrink), at 16 items per wave
// and ~15 % more multiply-adds per item (7 products in 8 slots, 9 in 12).  Same group law, same window recoding, same table
// layout (ed448_algo.h), same canonical affine output: byte-identical results.
//
// Indexed table lookups only (public scalars: verification, raw calls); secret scalars keep the constant-address kernels.
#pragma once
#include "ed448_algo.h"

namespace capy {
namespace quad {

#if defined(__HIP_DEVICE_COMPILE__)

// lane q of every quad receives `a` from quad lane P_q
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ Fe fe_perm(const Fe &a)
{
    Fe r;
#pragma unroll
    for (int i = 0; i < 16; i++)
        r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true);
    return r;
}
__device__ __forceinline__ Fe fe_sel(bool take_b, const Fe &a, const Fe &b)
{
    Fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.l[i] = take_b ? b.l[i] : a.l[i];
    return r;
}
__device__ __forceinline__ uint32_t p2_limb(int i) { return i == 8 ? 2 * (M28 - 1) : 2 * M28; }  // limbs of 2p

// (E, G, F, H) in lanes (0, 1, 2, 3)  ->  (E F, G H, F G, H E) = (X3, Y3, Z3, T3)
__device__ __forceinline__ Fe final_products(const Fe &egfh)
{
    const Fe other = fe_perm<2, 3, 1, 0>(egfh);  // (F, H, G, E)
    return fe_mul(egfh, other);
}

// own = (X, Y, Z, .) -> (X3, Y3, Z3, T3) of the doubled point.  Bounds as in pt_dbl_core (ed448_dev.h): A, B, C', S are
// R; G <= 2^29, H <= 2^29.58, E and F reduced.
__device__ __forceinline__ Fe dbl(const Fe &own, uint32_t q)
{
    // level 1: lane 3 squares X + Y, the others their own coordinate
    const Fe x3 = fe_perm<0, 1, 2, 0>(own), y3 = fe_perm<0, 1, 2, 1>(own);  // lane 3: X, Y (others: own twice)
    Fe op;
#pragma unroll
    for (int i = 0; i < 16; i++) op.l[i] = q == 3 ? x3.l[i] + y3.l[i] : own.l[i];  // <= 2^29: 40 * 2^58 < 2^64
    const Fe sq = fe_sqr(op);  // (A, B, C', S)
    // between: u = A + B (lanes 0, 1, 2) / H = A - B + 2p (lane 3); then E = S + 4p - u (lane 0), F = u + 4p - 2 C' (lane 2)
    const Fe a = fe_perm<0, 0, 0, 0>(sq), b = fe_perm<1, 1, 1, 1>(sq), s = fe_perm<3, 3, 3, 3>(sq);
    Fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t u = q == 3 ? a.l[i] + p2_limb(i) - b.l[i] : a.l[i] + b.l[i];  // G (<= 2^29) or H (<= 2^29.58)
        const uint32_t pos = q == 0 ? s.l[i] : u, neg = q == 0 ? u : 2 * sq.l[i];   // lane 0: S - u; lane 2: u - 2 C'
        const uint32_t ef = pos + 2 * p2_limb(i) - neg;                             // < 2^31
        r.l[i] = (q & 1) ? u : ef;
    }
    Fe red = r;
    fe_weak_reduce(red);  // E and F must be reduced (they meet G / H of 2^29.58 in the products)
#pragma unroll
    for (int i = 0; i < 16; i++) r.l[i] = (q & 1) ? r.l[i] : red.l[i];
    return final_products(r);  // (E, G, F, H) sit in lanes (0, 1, 2, 3)
}

// acc = (X1, Y1, Z1, T1), e = this lane's field of a cached entry (X2, Y2, Z2, d T2; X2 and d T2 possibly fe_neg_nr'ed:
// <= 2^29)  ->  acc + entry.  E = X1 Y2 + Y1 X2 is formed from two products instead of (X1 + Y1)(X2 + Y2) - A - B: the
// two lanes that hold X1, Y1 compute one each, and no carry pass is needed (E <= 2^29 + 2^11, like G).
__device__ __forceinline__ Fe add_cached(const Fe &acc, const Fe &e, uint32_t q)
{
    const Fe m = fe_mul(acc, e);                               // (A, B, D, C)
    const Fe cross = fe_mul(acc, fe_perm<1, 0, 2, 3>(e));      // (X1 Y2, Y1 X2, -, -)
    // U, V per lane: lane 0: cross0, cross1 -> E = U + V;  lane 1: D, C -> G = U + V;  lane 2: D, C -> F = U - V;  lane 3: B, A -> H = U - V
    const Fe mu = fe_perm<0, 2, 2, 1>(m), mv = fe_perm<0, 3, 3, 0>(m), cv = fe_perm<1, 1, 1, 1>(cross);
    Fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t u = q == 0 ? cross.l[i] : mu.l[i], v = q == 0 ? cv.l[i] : mv.l[i];
        r.l[i] = q < 2 ? u + v : u + p2_limb(i) - v;           // E, G <= 2^29 + 2^11;  F, H <= 2^29.58
    }
    // (E, G, F, H): products E F and H E meet 2^29.01 x 2^29.58 = 2^58.6 < 2^58.7, G H as in the one-lane form; F G too
    return final_products(r);
}

// [k]P for the item of this quad; tab = the item's VB_TABLE_DWORDS of scratch (layout of vb_build_table: entry j at
// tab + 64 j, fields X, Y, Z, dT at +0, +16, +32, +48 dwords; lane q of the quad owns field q).  Returns the lane's
// coordinate of the result (X, Y, Z, T).
__device__ __forceinline__ Fe scalarmul(const uint8_t *k_be, const uint8_t *xy, uint32_t *tab, uint32_t q)
{
    // the point in cached form, one field per lane: (x, y, 1, d x y); every lane reads both coordinates
    const Fe px = fe_from_bytes(xy), py = fe_from_bytes(xy + 56);
    Fe p_own = q == 0 ? px : py;
    {
        const Fe t = fe_mul(px, py), one = fe_one();
        p_own = fe_sel(q == 2, p_own, one);
        p_own = fe_sel(q == 3, p_own, t);
    }
    const Fe p_cached = fe_sel(q == 3, p_own, fe_mul_d(p_own));
    uint32_t *mine = tab + q * 16;
    // table {0 .. WHALF} P in cached form: the identity (0, 1, 1, 0), then repeated additions of P
    Fe acc = fe_zero();
    acc.l[0] = (q == 1 || q == 2) ? 1u : 0u;
#pragma unroll 1
    for (int j = 0; j < TAB_ENTRIES; j++) {
        store_fe(mine + j * 64, fe_sel(q == 3, acc, fe_mul_d(acc)));
        if (j + 1 < TAB_ENTRIES) acc = add_cached(acc, p_cached, q);
    }
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<WBITS>(w, k);
    sc_msb_align<WBITS>(w);
    const bool flips = (q == 0 || q == 3);  // -(x, y) = (-x, y): X2 and d T2 change sign
    auto entry = [&](int digit) -> Fe {
        const bool neg = digit < 0;
        const int idx = neg ? -digit : digit;
        const Fe e = load_fe(mine + idx * 64);
        return fe_sel(neg && flips, e, fe_neg_nr(e));  // <= 2^29, within add_cached's operand bounds
    };
    acc = fe_zero();
    acc.l[0] = (q == 1 || q == 2) ? 1u : 0u;
    acc = add_cached(acc, entry((int)top), q);
#pragma unroll 1
    for (int i = 0; i < NWIN; i++) {
        // the window's entry is requested BEFORE its doublings (the digit is known: most significant first), so the
        // table read (a different line per quad) hides under ~4300 instructions of arithmetic
        const Fe e = entry(sc_next_digit_msb<WBITS>(w));
#pragma unroll 1
        for (int j = 0; j < WBITS; j++) acc = dbl(acc, q);
        acc = add_cached(acc, e, q);
    }
    return acc;
}

// ---- constant-address form (secret scalars: KeyEncryptable's k V and s Z, /root/reference/src/ecc/encryptable.rs:37,78).
// The table of an item is 8 rows (1 .. CtWin::HALF times P; the identity row is not stored: a digit of 0 matches no row and
// the zeros it leaves become the cached identity) of 4 fields x 64 B, one field per lane: 2 KiB per item, 32 KiB per wave of
// 16 items -- it lives in LDS (four waves per compute unit = one per SIMD: 128 of 160 KiB), laid out so that lane l owns
// bytes 16 l .. 16 l + 15 of every 1 KiB line: every ds_read_b128 / ds_write_b128 of the wave is one conflict-free line, and
// its address depends on the lane and the row counter only.  Every row is read for every window and the wanted one kept by
// an arithmetic mask (ct_mask / ct_take, ed448_algo.h): 32 LDS reads + 128 v_bitop3_b32 per window, no HBM traffic at all
// (the one-item-per-lane hardened kernel reads its per-item tables from HBM: ~60 GB per 2^18 multiplications).
constexpr int QUAD_CT_ROWS = CtWin::HALF;                    // rows 1 .. HALF
constexpr int QUAD_CT_LDS_DWORDS = QUAD_CT_ROWS * 4 * 256;   // rows x 16-byte pieces x (64 lanes x 4 dwords)

__device__ __forceinline__ void ct_store_row(uint32_t *lds, int row, const Fe &a)
{
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int pc = 0; pc < 4; pc++) {
        const uint4 v = {a.l[4 * pc], a.l[4 * pc + 1], a.l[4 * pc + 2], a.l[4 * pc + 3]};
        *reinterpret_cast<uint4 *>(lds + ((row * 4 + pc) * 64 + lane) * 4) = v;
    }
}

// this lane's field of sign(digit) * tab[|digit|], every row read
__device__ __forceinline__ Fe ct_entry(const uint32_t *lds, int digit, uint32_t q)
{
    const uint32_t lane = threadIdx.x & 63;
    const bool neg = digit < 0;
    const uint32_t idx = (uint32_t)(neg ? -digit : digit);
    Fe e = fe_zero();
#pragma unroll
    for (int row = 0; row < QUAD_CT_ROWS; row++) {
        const uint32_t m = ct_mask((uint32_t)(row + 1) == idx);
#pragma unroll
        for (int pc = 0; pc < 4; pc++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(lds + ((row * 4 + pc) * 64 + lane) * 4);
            e.l[4 * pc] = ct_take(e.l[4 * pc], v.x, m);
            e.l[4 * pc + 1] = ct_take(e.l[4 * pc + 1], v.y, m);
            e.l[4 * pc + 2] = ct_take(e.l[4 * pc + 2], v.z, m);
            e.l[4 * pc + 3] = ct_take(e.l[4 * pc + 3], v.w, m);
        }
    }
    e.l[0] |= ct_mask(idx == 0) & ((q == 1 || q == 2) ? 1u : 0u);   // digit 0: the cached identity (0, 1, 1, 0)
    const bool flips = (q == 0 || q == 3);
    return fe_sel(neg && flips, e, fe_neg_nr(e));                    // a select of data, not of an address
}

// [k]P with constant-address lookups; lds = the wave's QUAD_CT_LDS_DWORDS.  Windows of CT_WBITS bits (as the other
// hardened kernels: fewer rows to read per window outweigh the extra windows).
__device__ __forceinline__ Fe scalarmul_ct(const uint8_t *k_be, const uint8_t *xy, uint32_t *lds, uint32_t q)
{
    const Fe px = fe_from_bytes(xy), py = fe_from_bytes(xy + 56);
    Fe p_own = q == 0 ? px : py;
    {
        const Fe t = fe_mul(px, py), one = fe_one();
        p_own = fe_sel(q == 2, p_own, one);
        p_own = fe_sel(q == 3, p_own, t);
    }
    const Fe p_cached = fe_sel(q == 3, p_own, fe_mul_d(p_own));
    Fe acc = fe_zero();
    acc.l[0] = (q == 1 || q == 2) ? 1u : 0u;
#pragma unroll 1
    for (int j = 0; j < QUAD_CT_ROWS; j++) {
        acc = add_cached(acc, p_cached, q);
        ct_store_row(lds, j, fe_sel(q == 3, acc, fe_mul_d(acc)));
    }
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<CT_WBITS>(w, k);
    sc_msb_align<CT_WBITS>(w);
    acc = fe_zero();
    acc.l[0] = (q == 1 || q == 2) ? 1u : 0u;
    acc = add_cached(acc, ct_entry(lds, (int)top, q), q);
#pragma unroll 1
    for (int i = 0; i < CtWin::NWIN; i++) {
        const Fe e = ct_entry(lds, sc_next_digit_msb<CT_WBITS>(w), q);
#pragma unroll 1
        for (int j = 0; j < CT_WBITS; j++) acc = dbl(acc, q);
        acc = add_cached(acc, e, q);
    }
    return acc;
}

// acc += [a]G from the shared fixed-base table on E (rows of FB_TAB_ENTRIES affine cached entries (x, y, d x y), 12-bit
// signed windows, row FbWin::NWIN = the recoding carry): 39 additions in the same quad form -- lane q reads field q of
// the entry, the Z lane multiplies by one.  The entry of the next window is requested before the current addition.
__device__ __forceinline__ Fe add_fixed_base(Fe acc, const uint8_t *a_be, const uint32_t *gtab, uint32_t q)
{
    uint32_t ka[14], wa[15];
    sc_from_be(ka, a_be);
    const uint32_t topa = sc_recode_signed<FB_WBITS>(wa, ka);
    const bool flips = (q == 0 || q == 3);
    const uint32_t field = q == 3 ? 2 : q;  // (x, y, -, d x y) -> fields 0, 1, -, 2
    auto entry = [&](int row, int digit) -> Fe {
        const bool neg = digit < 0;
        const int idx = neg ? -digit : digit;
        Fe e = load_fe(gtab + ((size_t)row * FB_TAB_ENTRIES + idx) * FB_ENTRY_DWORDS + field * 16);
        e = fe_sel(q == 2, e, fe_one());
        return fe_sel(neg && flips, e, fe_neg_nr(e));
    };
    Fe e = entry(FbWin::NWIN, (int)topa);
#pragma unroll 1
    for (int i = 0; i <= FbWin::NWIN; i++) {
        Fe nxt = e;
        if (i < FbWin::NWIN) nxt = entry(i, sc_next_digit_lsb<FB_WBITS>(wa));
        acc = add_cached(acc, e, q);
        e = nxt;
    }
    return acc;
}

#endif  // __HIP_DEVICE_COMPILE__

}  // namespace quad
}  // namespace capy
