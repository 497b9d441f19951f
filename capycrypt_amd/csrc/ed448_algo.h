// ed448_algo.h — scalar-multiplication algorithms shared by the HIP kernels and the host unit test.
//
// Variable base: signed fixed windows of WBITS bits over all 448 scalar bits (scalars arrive unreduced,
// /root/reference/src/sha3/aux_functions.rs:102-106), per-item table {0..2^(WBITS-1)}P in "cached" extended form
// kept in HBM (lane-major so every lane streams whole 128-B lines), uniform control flow (every window does WBITS
// doublings + 1 complete addition; the digit only selects the table row and a sign).
// Fixed base: (FbWin::NWIN+1) x FbWin::ENTRIES affine table of j*2^(FB_WBITS i)*G shared by all lanes, one mixed
// addition per window, no doublings.  WBITS = 5: 90 windows, 17 entries (4352 B per item); FB_WBITS = 8: 56 windows,
// 57 x 129 entries (1.41 MB, L2-resident; every lane of a wave reads the same row).
#pragma once
#include "ed448_dev.h"

namespace capy {

// unroll factor of the doubling loop inside a window (1 = rolled: smallest code; see DESIGN.md for the measurements)
#ifndef CAPY_ED448_DBL_UNROLL
#define CAPY_ED448_DBL_UNROLL 1
#endif
#define CAPY_PRAGMA_(x) _Pragma(#x)
#define CAPY_UNROLL(n) CAPY_PRAGMA_(unroll n)

constexpr int VB_TABLE_DWORDS = TAB_ENTRIES * 64;  // per item: entries x (X, Y, Z, dT) x 16 limbs
constexpr int FB_ROWS = FbWin::NWIN + 1;           // one row per window plus the recoding carry
constexpr int FB_TAB_ENTRIES = FbWin::ENTRIES;
constexpr int FB_ENTRY_DWORDS = 48;                // (x, y, d*x*y) x 16 limbs
constexpr int FB_TABLE_DWORDS = FB_ROWS * FB_TAB_ENTRIES * FB_ENTRY_DWORDS;

CAPY_HD inline void store_fe(uint32_t *dst, const Fe &a)
{
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        uint4 v = {a.l[i], a.l[i + 1], a.l[i + 2], a.l[i + 3]};
        *reinterpret_cast<uint4 *>(dst + i) = v;
    }
}
CAPY_HD inline Fe load_fe(const uint32_t *src)
{
    Fe a;
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        uint4 v = *reinterpret_cast<const uint4 *>(src + i);
        a.l[i] = v.x;
        a.l[i + 1] = v.y;
        a.l[i + 2] = v.z;
        a.l[i + 3] = v.w;
    }
    return a;
}

// Build the per-item table {0,1,..,WHALF}P (cached form) at tab[0 .. VB_TABLE_DWORDS).
CAPY_HD inline void vb_build_table(uint32_t *tab, const Pt &P)
{
    const Fe Pd = fe_mul_d(P.T);
    Pt acc = pt_identity();
#pragma unroll 1
    for (int j = 0; j < TAB_ENTRIES; j++) {
        store_fe(tab + j * 64, acc.X);
        store_fe(tab + j * 64 + 16, acc.Y);
        store_fe(tab + j * 64 + 32, acc.Z);
        store_fe(tab + j * 64 + 48, fe_mul_d(acc.T));
        if (j + 1 < TAB_ENTRIES) acc = pt_add_cached(acc, P.X, P.Y, P.Z, Pd);
    }
}

// acc += sign(digit) * tab[|digit|]
CAPY_HD inline Pt vb_add_digit(const Pt &acc, const uint32_t *tab, int digit)
{
    const bool neg = digit < 0;
    const int idx = neg ? -digit : digit;
    const uint32_t *e = tab + idx * 64;
    Fe X2 = load_fe(e), Y2 = load_fe(e + 16), Z2 = load_fe(e + 32), Td2 = load_fe(e + 48);
    // -(x, y) = (-x, y): negate X and T
    X2 = fe_select(neg, X2, fe_neg_nr(X2));  // <= 2^29, within pt_add_cached's operand bounds
    Td2 = fe_select(neg, Td2, fe_neg_nr(Td2));
    return pt_add_cached(acc, X2, Y2, Z2, Td2);
}

// [k]P, k = 56 big-endian bytes (all 448 bits used), tab = VB_TABLE_DWORDS of scratch for this item.
CAPY_HD inline Pt vb_scalarmul(const uint8_t *k_be, const Pt &P, uint32_t *tab)
{
    vb_build_table(tab, P);
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<WBITS>(w, k);
    sc_msb_align<WBITS>(w);
    Pt acc = vb_add_digit(pt_identity(), tab, (int)top);
#pragma unroll 1
    for (int i = 0; i < NWIN; i++) {
        // one doubling body keeps the loop inside the I-cache; the compiler sinks the T product (dead in all but
        // the last trip) out of the loop, so this runs 4S+3M per doubling plus one multiplication per window
        CAPY_UNROLL(CAPY_ED448_DBL_UNROLL)
        for (int j = 0; j < WBITS; j++) acc = pt_dbl<true>(acc);
        acc = vb_add_digit(acc, tab, sc_next_digit_msb<WBITS>(w));
    }
    return acc;
}

// acc += sign(digit) * G16[row][|digit|]   (affine cached entries: x, y, d*x*y)
CAPY_HD inline Pt fb_add_digit(const Pt &acc, const uint32_t *gtab, int row, int digit)
{
    const bool neg = digit < 0;
    const int idx = neg ? -digit : digit;
    const uint32_t *e = gtab + (row * FB_TAB_ENTRIES + idx) * FB_ENTRY_DWORDS;
    Fe x2 = load_fe(e), y2 = load_fe(e + 16), td2 = load_fe(e + 32);
    x2 = fe_select(neg, x2, fe_neg_nr(x2));
    td2 = fe_select(neg, td2, fe_neg_nr(td2));
    return pt_add_affine_cached(acc, x2, y2, td2);
}

// [k]G from the shared table gtab[FB_TABLE_DWORDS]
CAPY_HD inline Pt fb_scalarmul(const uint8_t *k_be, const uint32_t *gtab)
{
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<FB_WBITS>(w, k);
    Pt acc = fb_add_digit(pt_identity(), gtab, FbWin::NWIN, (int)top);
#pragma unroll 1
    for (int i = 0; i < FbWin::NWIN; i++) acc = fb_add_digit(acc, gtab, i, sc_next_digit_lsb<FB_WBITS>(w));
    return acc;
}

// [a]G + [b]P in one pass (Straus): the doublings of the variable-base loop are shared; the G part
// uses row 0 of the fixed-base table (j*G, j = 0..2^(FB_WBITS-1), of which the WBITS-wide digits reach 0..2^(WBITS-1)).
CAPY_HD inline Pt double_scalarmul(const uint8_t *a_be, const uint8_t *b_be, const Pt &P, uint32_t *tab,
                                   const uint32_t *gtab)
{
    vb_build_table(tab, P);
    uint32_t ka[14], kb[14], wa[15], wb[15];
    sc_from_be(ka, a_be);
    sc_from_be(kb, b_be);
    const uint32_t topa = sc_recode_signed<WBITS>(wa, ka), topb = sc_recode_signed<WBITS>(wb, kb);
    sc_msb_align<WBITS>(wa);
    sc_msb_align<WBITS>(wb);
    Pt acc = vb_add_digit(pt_identity(), tab, (int)topb);
    acc = fb_add_digit(acc, gtab, 0, (int)topa);
#pragma unroll 1
    for (int i = 0; i < NWIN; i++) {
#pragma unroll 1
        for (int j = 0; j < WBITS; j++) acc = pt_dbl<true>(acc);
        acc = vb_add_digit(acc, tab, sc_next_digit_msb<WBITS>(wb));
        acc = fb_add_digit(acc, gtab, 0, sc_next_digit_msb<WBITS>(wa));
    }
    return acc;
}

// ------------------------------------------------------------------ scalars mod r (Schnorr / ECDHIES glue)
// r = 2^446 - 0x8335dc163bb124b65129c96fde933d8d723a70aadc873d6d54a7bb0d, as 14 LE words.
// The curve crate's Scalar ops (`mul_mod`, `*`, `-`; call sites /root/reference/src/ecc/keypair.rs:43,
// signable.rs:42,46,54, encryptable.rs:36,77) are taken as arithmetic mod r with reduced results
// (assumption (iii)).  These run once or twice per signature, so they are simple
// bit-serial routines without dynamically indexed register arrays.
CAPY_HD inline uint32_t sc_r_word(int i)
{
    constexpr uint32_t R[14] = {0xab5844f3u, 0x2378c292u, 0x8dc58f55u, 0x216cc272u, 0xaed63690u,
                                0xc44edb49u, 0x7cca23e9u, 0xffffffffu, 0xffffffffu, 0xffffffffu,
                                0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
    return R[i];
}

// x -= r if x >= r   (x < 2^448)
CAPY_HD inline void sc_cond_sub_r(uint32_t x[14])
{
    uint32_t t[14];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        uint64_t v = (uint64_t)x[i] - sc_r_word(i) - borrow;
        t[i] = (uint32_t)v;
        borrow = (v >> 63) & 1;
    }
#pragma unroll
    for (int i = 0; i < 14; i++) x[i] = borrow ? x[i] : t[i];
}

// x mod r for any 448-bit x: 2^448 < 5r, so four conditional subtractions suffice
CAPY_HD inline void sc_reduce(uint32_t x[14])
{
#pragma unroll 1
    for (int i = 0; i < 4; i++) sc_cond_sub_r(x);
}

// x = 2x mod r, x < r
CAPY_HD inline void sc_dbl_mod(uint32_t x[14])
{
#pragma unroll
    for (int i = 13; i > 0; i--) x[i] = (x[i] << 1) | (x[i - 1] >> 31);
    x[0] <<= 1;
    sc_cond_sub_r(x);
}

// x = x + y mod r, both < r
CAPY_HD inline void sc_add_mod(uint32_t x[14], const uint32_t y[14])
{
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        uint64_t v = (uint64_t)x[i] + y[i] + c;
        x[i] = (uint32_t)v;
        c = v >> 32;
    }
    sc_cond_sub_r(x);
}

// out = a * b mod r (inputs: arbitrary 448-bit values)
CAPY_HD inline void sc_mul_mod(uint32_t out[14], const uint32_t a_in[14], const uint32_t b_in[14])
{
    uint32_t a[14], b[14], acc[14];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        a[i] = a_in[i];
        b[i] = b_in[i];
        acc[i] = 0;
    }
    sc_reduce(b);
#pragma unroll 1
    for (int bit = 0; bit < 448; bit++) {
        sc_dbl_mod(acc);
        const bool one = (a[13] >> 31) != 0;
#pragma unroll
        for (int i = 13; i > 0; i--) a[i] = (a[i] << 1) | (a[i - 1] >> 31);
        a[0] <<= 1;
        uint32_t t[14];
#pragma unroll
        for (int i = 0; i < 14; i++) t[i] = acc[i];
        sc_add_mod(t, b);
#pragma unroll
        for (int i = 0; i < 14; i++) acc[i] = one ? t[i] : acc[i];
    }
#pragma unroll
    for (int i = 0; i < 14; i++) out[i] = acc[i];
}

// out = 4 a mod r
CAPY_HD inline void sc_mul4_mod(uint32_t out[14], const uint32_t a_in[14])
{
#pragma unroll
    for (int i = 0; i < 14; i++) out[i] = a_in[i];
    sc_reduce(out);
    sc_dbl_mod(out);
    sc_dbl_mod(out);
}

// out = a - b mod r
CAPY_HD inline void sc_sub_mod(uint32_t out[14], const uint32_t a_in[14], const uint32_t b_in[14])
{
    uint32_t a[14], b[14];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        a[i] = a_in[i];
        b[i] = b_in[i];
    }
    sc_reduce(a);
    sc_reduce(b);
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        uint64_t v = (uint64_t)a[i] - b[i] - borrow;
        a[i] = (uint32_t)v;
        borrow = (v >> 63) & 1;
    }
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        uint64_t v = (uint64_t)a[i] + (borrow ? sc_r_word(i) : 0u) + c;
        out[i] = (uint32_t)v;
        c = v >> 32;
    }
}

}  // namespace capy
