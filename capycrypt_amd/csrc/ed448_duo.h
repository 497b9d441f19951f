// ed448_duo.h — Ed448 variable-base multiplication with TWO LANES PER ITEM (r04), for batches of 16 k .. 32 k items: what
// BASELINE config 4 (2^18 pairs) leaves per GPU when it is split over eight.  There the four-lanes-per-item kernels
// (ed448_quad.h) need two waves per SIMD, which run one after the other (every instruction of these kernels is a 4-cycle
// one: nothing pairs, profiles/r03_valu_issue_bisect.txt), and the one-item-per-lane kernels leave half the SIMDs idle.
// Two lanes per item put exactly one wave of 32 items on every SIMD at 32 768 items.
//
// The split is the quad kernels' (of the POINT, not of the field multiplication), folded once more: lane 0 of a pair holds
// (u, v) = (Y, Z), lane 1 holds (u, v) = (X, T); every lane runs whole fe_mul / fe_sqr on its own operands and operands
// cross lanes only between levels (v_mov_b32_dpp quad_perm:[1,0,3,2], one instruction per limb for both directions):
//   doubling   level 1a: (Y, X)^2                     -> B | A                        1 fe_sqr
//              level 1b: (Z, X + Y)^2                 -> C' | S                       1 swap + 1 fe_sqr
//              between : G = A + B, H = A - B in both lanes; F = G - 2 C' | E = S - G  1 swap + limb arithmetic + one carry pass
//              level 2a: H x (G | E)                  -> Y3 | T3 ... stored as u | v  (see dbl: which product lands where)
//              level 2b: (F | F) x (G | E)            -> Z3 | X3                      1 swap + 2 fe_mul
//   addition   level 1 : (Y1, X1) x (Y2, X2) -> B | A;  (Z1, T1) x (Z2, dT2) -> D | C;  (Y1, X1) x (X2, Y2) -> halves of E
//              between : E, G = D + C, F = D - C, H = B - A (three swaps, no carry pass: bounds as in pt_add_cached)
//              level 2 : as for the doubling
// 2 S + 2 M (~1400 instructions) per doubling and 5 M (~1800) per addition in one lane's stream against 2056 / 3023 for one
// item per lane and 870 / 1100 at four lanes per item: per ITEM 0.78x the wave-instructions of the quad form (7 products
// in 8 slots and 10 in 10, against 7 in 8 and 9 in 12; the final inversion is shared by 32 items instead of 16).
// Same group law, window recoding and canonical affine output: byte-identical results.  Indexed table lookups only (public
// scalars); the table of an item is private to its pair and uses the pair's own layout (lane p owns dwords p * 32 .. of
// every 64-dword entry: its two fields, contiguous).
#pragma once
#include "ed448_quad.h"

namespace capy {
namespace duo {

#if defined(__HIP_DEVICE_COMPILE__)

using quad::fe_perm;
using quad::fe_sel;
using quad::p2_limb;

__device__ __forceinline__ Fe fe_swap(const Fe &a) { return fe_perm<1, 0, 3, 2>(a); }

struct Half {  // one lane's half of a point: lane 0 (u, v) = (Y, Z), lane 1 (u, v) = (X, T)
    Fe u, v;
};

// lane 0 has G, H and its own ef = F; lane 1 has G, H and its own ef = E  ->  (Y3, Z3) | (X3, T3)
__device__ __forceinline__ Half final_products(const Fe &G, const Fe &H, const Fe &ef, bool p)
{
    const Fe of = fe_swap(ef);                  // E | F
    const Fe common = fe_sel(p, G, ef);         // G | E
    Half r;
    r.u = fe_mul(common, fe_sel(p, H, of));     // G H = Y3 | E F = X3
    r.v = fe_mul(common, fe_sel(p, ef, H));     // G F = Z3 | E H = T3
    return r;
}

// Bounds as in pt_dbl_core (ed448_dev.h): A, B, C', S are R; G <= 2^29 + 2^11, H <= 2^29.58, E and F reduced.
__device__ __forceinline__ Half dbl(const Half &a, bool p)
{
    const Fe sq1 = fe_sqr(a.u);                 // B | A
    const Fe su = fe_swap(a.u);                 // X | Y
    Fe op;
#pragma unroll
    for (int i = 0; i < 16; i++) op.l[i] = p ? a.u.l[i] + su.l[i] : a.v.l[i];  // Z | X + Y (<= 2^29: 40 * 2^58 < 2^64)
    const Fe sq2 = fe_sqr(op);                  // C' | S
    const Fe o1 = fe_swap(sq1);                 // A | B
    Fe G, H, ef;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t A = p ? sq1.l[i] : o1.l[i];
        G.l[i] = sq1.l[i] + o1.l[i];
        H.l[i] = (A << 1) + p2_limb(i) - G.l[i];                               // A - B + 2p, limb-wise the same value
        const uint32_t pos = p ? sq2.l[i] : G.l[i], neg = p ? G.l[i] : 2 * sq2.l[i];
        ef.l[i] = pos + 2 * p2_limb(i) - neg;                                  // G - 2 C' + 4p | S - G + 4p   (< 2^31)
    }
    fe_weak_reduce(ef);                          // E and F must be reduced (they meet G / H of 2^29.58 in the products)
    return final_products(G, H, ef, p);
}

// acc + entry; e = this lane's half of a cached entry: (Y2, Z2) | (X2, d T2), lane 1's possibly fe_neg_nr'ed (<= 2^29).
// E = X1 Y2 + Y1 X2 from two products, one per lane, as in the quad form: no carry pass between the levels.
__device__ __forceinline__ Half add_cached(const Half &a, const Half &e, bool p)
{
    const Fe mA = fe_mul(a.u, e.u);             // B | A
    const Fe mB = fe_mul(a.v, e.v);             // D | C
    const Fe mC = fe_mul(a.u, fe_swap(e.u));    // Y1 X2 | X1 Y2
    const Fe oA = fe_swap(mA), oB = fe_swap(mB), oC = fe_swap(mC);
    Fe G, H, ef;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t B = p ? oA.l[i] : mA.l[i], D = p ? oB.l[i] : mB.l[i];
        const uint32_t sumA = mA.l[i] + oA.l[i], sumB = mB.l[i] + oB.l[i];
        G.l[i] = sumB;                                                         // D + C <= 2^29 + 2^11
        H.l[i] = (B << 1) + p2_limb(i) - sumA;                                 // B - A + 2p <= 2^29.58
        const uint32_t F = (D << 1) + p2_limb(i) - sumB;                       // D - C + 2p <= 2^29.58
        ef.l[i] = p ? mC.l[i] + oC.l[i] : F;                                   // F | E (<= 2^29 + 2^11)
    }
    // E F and E H meet 2^29.01 x 2^29.58 = 2^58.6 < 2^58.7; G H and F G as in the one-lane form
    return final_products(G, H, ef, p);
}

// the identity (0, 1, 1, 0): (Y, Z) = (1, 1) | (X, T) = (0, 0)
__device__ __forceinline__ Half identity(bool p)
{
    Half r;
    r.u = fe_zero();
    r.u.l[0] = p ? 0u : 1u;
    r.v = r.u;
    return r;
}

// [k]P for the item of this pair; tab = the item's VB_TABLE_DWORDS of scratch (entry j at tab + 64 j; lane p owns dwords
// 32 p .. 32 p + 31 of it: u then v, v of lane 1 multiplied by d).
__device__ __forceinline__ Half scalarmul(const uint8_t *k_be, const uint8_t *xy, uint32_t *tab, bool p)
{
    const Fe px = fe_from_bytes(xy), py = fe_from_bytes(xy + 56);
    Half pc;  // P in cached form: (y, 1) | (x, d x y)
    pc.u = fe_sel(p, py, px);
    pc.v = fe_sel(p, fe_one(), fe_mul_d(fe_mul(px, py)));
    uint32_t *mine = tab + (p ? 32 : 0);
    Half acc = identity(p);
#pragma unroll 1
    for (int j = 0; j < TAB_ENTRIES; j++) {
        store_fe(mine + j * 64, acc.u);
        store_fe(mine + j * 64 + 16, fe_sel(p, acc.v, fe_mul_d(acc.v)));
        if (j + 1 < TAB_ENTRIES) acc = add_cached(acc, pc, p);
    }
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<WBITS>(w, k);
    sc_msb_align<WBITS>(w);
    auto entry = [&](int digit) -> Half {
        const bool neg = digit < 0;
        const int idx = neg ? -digit : digit;
        Half e;
        e.u = load_fe(mine + idx * 64);
        e.v = load_fe(mine + idx * 64 + 16);
        // -(x, y) = (-x, y): X2 and d T2 change sign -- both live in lane 1
        e.u = fe_sel(neg && p, e.u, fe_neg_nr(e.u));
        e.v = fe_sel(neg && p, e.v, fe_neg_nr(e.v));
        return e;
    };
    acc = add_cached(identity(p), entry((int)top), p);
#pragma unroll 1
    for (int i = 0; i < NWIN; i++) {
        // the entry is requested before the window's doublings (most significant digit first: it is known)
        const Half e = entry(sc_next_digit_msb<WBITS>(w));
#pragma unroll 1
        for (int j = 0; j < WBITS; j++) acc = dbl(acc, p);
        acc = add_cached(acc, e, p);
    }
    return acc;
}

// ---- constant-address form (secret scalars, 16 k .. 32 k items: KeyEncryptable's k V and s Z,
// /root/reference/src/ecc/encryptable.rs:37,78), r05.  The constant-address quad kernel (ed448_quad.h) needs 32 KiB of LDS per
// wave of 16 items, so a compute unit holds four of its waves and 32 768 items take two rounds (2.35 ms against 1.25 ms for
// 16 384).  Two lanes per item put those 32 768 items on one wave per SIMD -- if the window table of 32 items fits 40 KiB.
// It does not in LDS alone (8 rows x 4 fields x 64 B = 2 KiB per item), so the table is SPLIT: rows 1 .. 4 of an item stay in
// the REGISTERS of its two lanes (each lane its own two fields of a row: 32 dwords, 128 VGPRs for four rows -- the kernel runs
// at one wave per SIMD, where 512 registers per lane are there to be used), rows 5 .. 8 in LDS (32 pieces of 16 bytes per lane,
// lane l owning bytes 16 l .. 16 l + 15 of every 1 KiB line: conflict-free, 32 KiB per wave).  Every row of both halves is read
// for every window and the wanted one kept by an arithmetic mask (ct_mask / ct_take, ed448_algo.h): 128 v_bitop3_b32 on the
// register rows + 32 LDS reads and 128 v_bitop3_b32 on the others; no address, branch or exec mask depends on the scalar.
constexpr int DUO_CT_ROWS = CtWin::HALF;                       // rows 1 .. HALF (8)
constexpr int DUO_CT_REG_ROWS = DUO_CT_ROWS / 2;               // 1 .. 4 in registers
constexpr int DUO_CT_LDS_ROWS = DUO_CT_ROWS - DUO_CT_REG_ROWS;  // 5 .. 8 in LDS
constexpr int DUO_CT_LDS_DWORDS = DUO_CT_LDS_ROWS * 8 * 256;   // rows x 16-byte pieces (2 fields x 4) x (64 lanes x 4 dwords)

__device__ __forceinline__ void ct_store_row(uint32_t *lds, int row, const Half &h)
{
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int pc = 0; pc < 8; pc++) {
        const Fe &f = pc < 4 ? h.u : h.v;
        const int b = 4 * (pc & 3);
        const uint4 v = {f.l[b], f.l[b + 1], f.l[b + 2], f.l[b + 3]};
        *reinterpret_cast<uint4 *>(lds + ((row * 8 + pc) * 64 + lane) * 4) = v;
    }
}
__device__ __forceinline__ Half ct_load_row(const uint32_t *lds, int row)
{
    const uint32_t lane = threadIdx.x & 63;
    Half h;
#pragma unroll
    for (int pc = 0; pc < 8; pc++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(lds + ((row * 8 + pc) * 64 + lane) * 4);
        Fe &f = pc < 4 ? h.u : h.v;
        const int b = 4 * (pc & 3);
        f.l[b] = v.x, f.l[b + 1] = v.y, f.l[b + 2] = v.z, f.l[b + 3] = v.w;
    }
    return h;
}

// this lane's half of sign(digit) * tab[|digit|]: (Y2, Z2) | (X2, d T2); every row of both halves of the table is read
__device__ __forceinline__ Half ct_entry(const Half (&regs)[DUO_CT_REG_ROWS], const uint32_t *lds, int digit, bool p)
{
    const uint32_t lane = threadIdx.x & 63;
    const bool neg = digit < 0;
    const uint32_t idx = (uint32_t)(neg ? -digit : digit);
    Half e;
    e.u = fe_zero();
    e.v = fe_zero();
#pragma unroll
    for (int row = 0; row < DUO_CT_REG_ROWS; row++) {
        const uint32_t m = ct_mask((uint32_t)(row + 1) == idx);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            e.u.l[i] = ct_take(e.u.l[i], regs[row].u.l[i], m);
            e.v.l[i] = ct_take(e.v.l[i], regs[row].v.l[i], m);
        }
    }
#pragma unroll
    for (int row = 0; row < DUO_CT_LDS_ROWS; row++) {
        const uint32_t m = ct_mask((uint32_t)(row + 1 + DUO_CT_REG_ROWS) == idx);
#pragma unroll
        for (int pc = 0; pc < 8; pc++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(lds + ((row * 8 + pc) * 64 + lane) * 4);
            Fe &f = pc < 4 ? e.u : e.v;
            const int b = 4 * (pc & 3);
            f.l[b] = ct_take(f.l[b], v.x, m);
            f.l[b + 1] = ct_take(f.l[b + 1], v.y, m);
            f.l[b + 2] = ct_take(f.l[b + 2], v.z, m);
            f.l[b + 3] = ct_take(f.l[b + 3], v.w, m);
        }
    }
    // digit 0: no row matched; the zeros become the cached identity (Y2, Z2) = (1, 1) | (X2, d T2) = (0, 0)
    const uint32_t one0 = ct_mask(idx == 0) & (p ? 0u : 1u);
    e.u.l[0] |= one0;
    e.v.l[0] |= one0;
    // -(x, y) = (-x, y): X2 and d T2 change sign, both live in lane 1 -- a select of data, not of an address
    e.u = fe_sel(neg && p, e.u, fe_neg_nr(e.u));
    e.v = fe_sel(neg && p, e.v, fe_neg_nr(e.v));
    return e;
}

// [k]P with constant-address lookups; lds = the wave's DUO_CT_LDS_DWORDS.  4-bit signed windows as the other hardened kernels.
__device__ __forceinline__ Half scalarmul_ct(const uint8_t *k_be, const uint8_t *xy, uint32_t *lds, bool p)
{
    const Fe px = fe_from_bytes(xy), py = fe_from_bytes(xy + 56);
    Half pc;  // P in cached form: (y, 1) | (x, d x y)
    pc.u = fe_sel(p, py, px);
    pc.v = fe_sel(p, fe_one(), fe_mul_d(fe_mul(px, py)));
    // rows 1 .. 4 through LDS into registers, then rows 5 .. 8 into the same LDS slots (two rolled loops: the additions are not
    // unrolled eight times)
    Half acc = identity(p);
#pragma unroll 1
    for (int j = 0; j < DUO_CT_REG_ROWS; j++) {
        acc = add_cached(acc, pc, p);
        Half row = acc;
        row.v = fe_sel(p, acc.v, fe_mul_d(acc.v));
        ct_store_row(lds, j, row);
    }
    Half regs[DUO_CT_REG_ROWS];
#pragma unroll
    for (int j = 0; j < DUO_CT_REG_ROWS; j++) regs[j] = ct_load_row(lds, j);
#pragma unroll 1
    for (int j = 0; j < DUO_CT_LDS_ROWS; j++) {
        acc = add_cached(acc, pc, p);
        Half row = acc;
        row.v = fe_sel(p, acc.v, fe_mul_d(acc.v));
        ct_store_row(lds, j, row);
    }
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<CT_WBITS>(w, k);
    sc_msb_align<CT_WBITS>(w);
    acc = add_cached(identity(p), ct_entry(regs, lds, (int)top, p), p);
#pragma unroll 1
    for (int i = 0; i < CtWin::NWIN; i++) {
        const Half e = ct_entry(regs, lds, sc_next_digit_msb<CT_WBITS>(w), p);
#pragma unroll 1
        for (int j = 0; j < CT_WBITS; j++) acc = dbl(acc, p);
        acc = add_cached(acc, e, p);
    }
    return acc;
}

// acc += [a]G from the shared fixed-base table on E (rows of FB_TAB_ENTRIES affine cached entries (x, y, d x y), 12-bit
// signed windows, row FbWin::NWIN = the recoding carry): 39 additions in pair form, (y, 1) | (x, d x y) per entry.
__device__ __forceinline__ Half add_fixed_base(Half acc, const uint8_t *a_be, const uint32_t *gtab, bool p)
{
    uint32_t ka[14], wa[15];
    sc_from_be(ka, a_be);
    const uint32_t topa = sc_recode_signed<FB_WBITS>(wa, ka);
    auto entry = [&](int row, int digit) -> Half {
        const bool neg = digit < 0;
        const int idx = neg ? -digit : digit;
        const uint32_t *src = gtab + ((size_t)row * FB_TAB_ENTRIES + idx) * FB_ENTRY_DWORDS;
        Half e;
        e.u = load_fe(src + (p ? 0 : 16));
        e.v = load_fe(src + 32);  // lane 0 replaces it by one (the load keeps the lanes' instruction streams identical)
        e.v = fe_sel(p, fe_one(), e.v);
        e.u = fe_sel(neg && p, e.u, fe_neg_nr(e.u));
        e.v = fe_sel(neg && p, e.v, fe_neg_nr(e.v));
        return e;
    };
    Half e = entry(FbWin::NWIN, (int)topa);
#pragma unroll 1
    for (int i = 0; i <= FbWin::NWIN; i++) {
        Half nxt = e;
        if (i < FbWin::NWIN) nxt = entry(i, sc_next_digit_lsb<FB_WBITS>(wa));
        acc = add_cached(acc, e, p);
        e = nxt;
    }
    return acc;
}

// affine bytes of the pair's result: both lanes invert Z (lane 0's v), lane 1 writes x, lane 0 writes y
__device__ __forceinline__ void store_affine(uint8_t *out_xy, const Half &r, bool p, bool live)
{
    const Fe zi = fe_inv_out(fe_perm<0, 0, 2, 2>(r.v));
    const Fe c = fe_mul(r.u, zi);
    if (live) fe_to_bytes(out_xy + (p ? 0 : 56), c);
}

#endif  // __HIP_DEVICE_COMPILE__

}  // namespace duo
}  // namespace capy
