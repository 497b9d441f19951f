// ed448_algo.h — scalar-multiplication algorithms shared by the HIP kernels and the host unit test.
//
// Variable base: signed fixed windows of WBITS bits over all 448 scalar bits (scalars arrive unreduced,
// /root/reference/src/sha3/aux_functions.rs:102-106), per-item table {0..2^(WBITS-1)}P in "cached" extended form
// kept in HBM (lane-major so every lane streams whole 128-B lines), uniform control flow (every window does WBITS
// doublings + 1 complete addition; the digit only selects the table row and a sign).
// Fixed base: (FbWin::NWIN+1) x FbWin::ENTRIES affine table of j*2^(FB_WBITS i)*G shared by all lanes, one mixed
// addition per window, no doublings.  WBITS = 5: 90 windows, 17 entries (4352 B per item); FB_WBITS = 12: 38 windows,
// 39 x 2049 entries (15.3 MB: L2 / Infinity Cache; every lane of a wave reads the same row).  Measured
// 8 / 9 / 10 / 11 / 12 bits: 1.56 / 1.43 / 1.33 / 1.24 / 1.19 ms per 2^18 fixed-base multiplications.
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

// ---- table entries fetched AHEAD through LDS (device only; r03).  At one wave per SIMD nothing hides the latency of an
// indexed table read (a different line per lane): the wait before every addition was 8 % of a variable-base and 12 % of a
// fixed-base multiplication (SQ_WAIT_ANY, profiles/r03_ed448_prefetch.txt), and the registers to hold the next entry
// during the doublings do not exist (the kernels already fill 256 VGPRs).  global_load_lds_dwordx4 copies memory -> LDS
// without passing through VGPRs: the entry for the next window is requested before the doublings (variable base) /
// before the current addition (fixed base) and read from LDS when it is needed; the compiler's vmcnt wait sits in
// front of that read.  Piece q (16 bytes) of lane l lands at dword (q * 64 + l) * 4 of the wave's staging area.
// `lds` = nullptr (host build, constant-address forms, one-item-per-wave kernels) keeps the direct loads.
constexpr int VB_PF_DWORDS = 16 * 64 * 4;  // one cached point (64 dwords) per lane
constexpr int FB_PF_DWORDS = 12 * 64 * 4;  // one affine cached entry (48 dwords) per lane
#if defined(__HIP_DEVICE_COMPILE__)
template <int QUADS, int SRC_STRIDE_DWORDS = 4>  // piece q of a lane is read from src + q * SRC_STRIDE_DWORDS
__device__ __forceinline__ void lds_prefetch(uint32_t *lds, const uint32_t *src)
{
    // the staging area may still be being read (ds_read of the previous entry): let those reads land first
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
#pragma unroll
    for (int q = 0; q < QUADS; q++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + SRC_STRIDE_DWORDS * q),
                                         (__attribute__((address_space(3))) void *)(lds + q * 256), 16, 0, 0);
}
// the requested entry has arrived.  The compiler puts this wait in front of an LDS read that follows the request in
// straight-line code, but NOT when the request was made in the previous trip of a loop (seen in fb_kernel: ds_read at
// the loop head with no vmcnt wait), so it is explicit
__device__ __forceinline__ void lds_prefetch_wait() { __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0) */ }
__device__ __forceinline__ Fe lds_load_fe(const uint32_t *lds, int f)
{
    const uint32_t lane = threadIdx.x & 63;
    Fe a;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(lds + ((f * 4 + q) * 64 + lane) * 4);
        a.l[4 * q] = v.x;
        a.l[4 * q + 1] = v.y;
        a.l[4 * q + 2] = v.z;
        a.l[4 * q + 3] = v.w;
    }
    return a;
}
#endif

#if defined(__HIPCC__)
// Two items per lane sharing one inversion (pt_pair_to_affine_bytes): a wave takes 128 consecutive items, lane l the
// items base + l and base + 64 + l.  The first result waits in LDS (one column per lane) while the second is computed
// by the same loop body, so the code is not duplicated.  Used when the batch still fills the chip at half the waves.
struct PtXYZ {
    uint32_t w[48][64];  // X, Y, Z limbs x lanes
};
__device__ __forceinline__ void park_xyz(PtXYZ &s, const Pt &p)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        s.w[i][threadIdx.x] = p.X.l[i];
        s.w[16 + i][threadIdx.x] = p.Y.l[i];
        s.w[32 + i][threadIdx.x] = p.Z.l[i];
    }
}
__device__ __forceinline__ Pt unpark_xyz(const PtXYZ &s)
{
    Pt p;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        p.X.l[i] = s.w[i][threadIdx.x];
        p.Y.l[i] = s.w[16 + i][threadIdx.x];
        p.Z.l[i] = s.w[32 + i][threadIdx.x];
        p.T.l[i] = 0;
    }
    return p;
}
#endif

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

// The same with a CONSTANT ADDRESS STREAM: every row of the table is read and the wanted one is kept by masking, so
// that neither control flow nor any address depends on the (secret) digit -- what the reference's curve crate means
// by its fixed-time table lookup (/root/reference/tests/integration_tests.rs:131-134).  Used when
// capy_ed448_set_hardened(1) is in force.  Since every lane reads EVERY row, the table of a wave is interleaved across
// its lanes -- 16-byte quad q of field element f of entry j of lane l sits at ((j*16 + f*4 + q) * nlanes + l) * 4 dwords
// -- so that each load instruction of the wave covers one contiguous KiB (the lane-major layout of the indexed form
// would touch 64 different lines per instruction).  17x the table reads and 1088 more VALU per window.
// The hardened form uses its own, narrower windows: every row is read per window, so fewer rows (9 instead of 17)
// outweigh the extra windows (112 instead of 90): CAPY_ED448_CT_WBITS = 4 measured against 5 in
// profiles/r02_ed448_hardened.txt.
#ifndef CAPY_ED448_CT_SCAN_UNROLL
#define CAPY_ED448_CT_SCAN_UNROLL 1  // rows of the per-item table per trip of the scan loop (2 / 4 measured: 3.27 -> 6.6 / 5.9 ms
                                     // at 65 536 items, the 64 extra registers per row in flight spill)
#endif
#ifndef CAPY_ED448_CT_WBITS
#define CAPY_ED448_CT_WBITS 4
#endif
constexpr int CT_WBITS = CAPY_ED448_CT_WBITS;
using CtWin = Win<CT_WBITS>;
static_assert(CtWin::ENTRIES <= TAB_ENTRIES, "the hardened table lives in the same scratch as the indexed one");

// m with its value hidden from the optimiser: a select written as `hit ? v : x` is turned into a BRANCH around the
// loads (control flow and addresses would then depend on the secret digit); an opaque mask keeps it arithmetic
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) uint32_t *CtConstPtr;
#else
typedef const uint32_t *CtConstPtr;
#endif
// acc | (v & m) in one instruction (v_bitop3_b32, truth table 0xEA; the compiler emits v_and_b32 + v_or_b32 for the C form)
CAPY_HD inline uint32_t ct_take(uint32_t acc, uint32_t v, uint32_t m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(v, m, acc, 0xEA);
#else
    return acc | (v & m);
#endif
}
CAPY_HD inline uint32_t ct_mask(bool hit)
{
    uint32_t m = 0u - (uint32_t)hit;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(m));
#endif
    return m;
}

struct CtTable {
    uint32_t *base;   // the wave's table: CtWin::ENTRIES * 64 dwords per lane, interleaved
    uint32_t lane;    // my lane within the wave (0 on the host)
    uint32_t nlanes;  // 64 on the device, 1 in the host unit test
};
CAPY_HD inline void store_fe_ct(const CtTable &t, int row4, const Fe &a)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint4 v = {a.l[4 * q], a.l[4 * q + 1], a.l[4 * q + 2], a.l[4 * q + 3]};
        *reinterpret_cast<uint4 *>(t.base + ((size_t)(row4 + q) * t.nlanes + t.lane) * 4) = v;
    }
}
CAPY_HD inline void vb_build_table_ct(const CtTable &t, const Pt &P)
{
    const Fe Pd = fe_mul_d(P.T);
    Pt acc = pt_identity();
#pragma unroll 1
    for (int j = 0; j < CtWin::ENTRIES; j++) {
        store_fe_ct(t, j * 16, acc.X);
        store_fe_ct(t, j * 16 + 4, acc.Y);
        store_fe_ct(t, j * 16 + 8, acc.Z);
        store_fe_ct(t, j * 16 + 12, fe_mul_d(acc.T));
        if (j + 1 < CtWin::ENTRIES) acc = pt_add_cached(acc, P.X, P.Y, P.Z, Pd);
    }
}
CAPY_HD inline Pt vb_add_digit_ct(const Pt &acc, const CtTable &t, int digit)
{
    const bool neg = digit < 0;
    const uint32_t idx = (uint32_t)(neg ? -digit : digit);
    Fe sel[4] = {fe_zero(), fe_zero(), fe_zero(), fe_zero()};  // X, Y, Z, dT
    // row 0 (the identity, digit 0) is not read: a digit of 0 matches no row and leaves zeros, which become the cached
    // identity (0, 1, 1, 0) below -- one ninth less to scan (the scan is what the hardened variable base adds: its
    // per-item tables make 2^18 multiplications read ~60 GB)
    CAPY_UNROLL(CAPY_ED448_CT_SCAN_UNROLL)
    for (uint32_t j = 1; j < (uint32_t)CtWin::ENTRIES; j++) {
        const uint32_t m = ct_mask(j == idx);
#pragma unroll
        for (int f = 0; f < 4; f++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(t.base + ((size_t)(j * 16 + f * 4 + q) * t.nlanes + t.lane) * 4);
                sel[f].l[4 * q] = ct_take(sel[f].l[4 * q], v.x, m);
                sel[f].l[4 * q + 1] = ct_take(sel[f].l[4 * q + 1], v.y, m);
                sel[f].l[4 * q + 2] = ct_take(sel[f].l[4 * q + 2], v.z, m);
                sel[f].l[4 * q + 3] = ct_take(sel[f].l[4 * q + 3], v.w, m);
            }
        }
    }
    const uint32_t is0 = ct_mask(idx == 0) & 1u;
    sel[1].l[0] |= is0;
    sel[2].l[0] |= is0;
    sel[0] = fe_select(neg, sel[0], fe_neg_nr(sel[0]));
    sel[3] = fe_select(neg, sel[3], fe_neg_nr(sel[3]));
    return pt_add_cached(acc, sel[0], sel[1], sel[2], sel[3]);
}

// [k]P with constant-address table lookups
CAPY_HD_INLINE Pt vb_scalarmul_ct(const uint8_t *k_be, const Pt &P, const CtTable &t)
{
    vb_build_table_ct(t, P);
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<CT_WBITS>(w, k);
    sc_msb_align<CT_WBITS>(w);
    Pt acc = vb_add_digit_ct(pt_identity(), t, (int)top);
#pragma unroll 1
    for (int i = 0; i < CtWin::NWIN; i++) {
        acc = pt_dbl_n<CT_WBITS, CAPY_ED448_DBL_UNROLL>(acc);
        acc = vb_add_digit_ct(acc, t, sc_next_digit_msb<CT_WBITS>(w));
    }
    return acc;
}

// [k]P, k = 56 big-endian bytes (all 448 bits used), tab = VB_TABLE_DWORDS of scratch for this item; lds = the wave's
// VB_PF_DWORDS of staging (or nullptr: direct table reads).
CAPY_HD_INLINE Pt vb_scalarmul(const uint8_t *k_be, const Pt &P, uint32_t *tab, uint32_t *lds = nullptr)
{
    vb_build_table(tab, P);
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<WBITS>(w, k);
    sc_msb_align<WBITS>(w);
    Pt acc = vb_add_digit(pt_identity(), tab, (int)top);
#if defined(__HIP_DEVICE_COMPILE__)
    if (lds) {
#pragma unroll 1
        for (int i = 0; i < NWIN; i++) {
            const int digit = sc_next_digit_msb<WBITS>(w);
            const bool neg = digit < 0;
            lds_prefetch<16>(lds, tab + (neg ? -digit : digit) * 64);
            acc = pt_dbl_n<WBITS, CAPY_ED448_DBL_UNROLL>(acc);
            lds_prefetch_wait();
            // pt_add_cached with every operand of the entry read from LDS right before its multiplication (16 live
            // registers for the entry instead of 64: the doubling loop around this keeps 256 VGPRs busy)
            {
                const Pt p = acc;
                Fe Td2 = lds_load_fe(lds, 3);
                Td2 = fe_select(neg, Td2, fe_neg_nr(Td2));
                const Fe C = fe_mul(p.T, Td2);
                __builtin_amdgcn_sched_barrier(0);
                const Fe D = fe_mul(p.Z, lds_load_fe(lds, 2));
                __builtin_amdgcn_sched_barrier(0);
                const Fe F = fe_sub_nr(D, C), G = fe_add_nr(D, C);
                Fe X2 = lds_load_fe(lds, 0);
                X2 = fe_select(neg, X2, fe_neg_nr(X2));
                const Fe Y2 = lds_load_fe(lds, 1);
                const Fe A = fe_mul(p.X, X2), B = fe_mul(p.Y, Y2);
                Fe E = fe_mul(fe_add_nr(p.X, p.Y), fe_add_nr(X2, Y2));
                E = fe_sub(fe_sub_nr(E, A), B);
                const Fe H = fe_sub_nr(B, A);
                acc.X = fe_mul(E, F);
                acc.Y = fe_mul(G, H);
                acc.Z = fe_mul(F, G);
                acc.T = fe_mul(E, H);
            }
        }
        return acc;
    }
#endif
#pragma unroll 1
    for (int i = 0; i < NWIN; i++) {
        // one doubling body keeps the loop inside the I-cache; T comes from the last doubling only (pt_dbl_n): 3S + 4M per
        // doubling plus one multiplication per window
        acc = pt_dbl_n<WBITS, CAPY_ED448_DBL_UNROLL>(acc);
        acc = vb_add_digit(acc, tab, sc_next_digit_msb<WBITS>(w));
    }
    return acc;
}

// acc + sign * entry for a fixed-base table entry (e0, e1, e2).  TW = false: the entry is (x, y, d x y) on E, its
// negative (-x, y, -d x y), the addition pt_add_affine_cached (8M).  TW = true: the entry is (y - x, y + x, 2 d' x y)
// on the twisted curve E' (ed448_dev.h), its negative swaps the first two and negates the third, the addition is
// pt_madd_niels_tw (7M).  The whole accumulation then lives on E' and ends with pt_tw_to_affine_bytes.
template <bool TW>
CAPY_HD inline Pt fb_add_entry(const Pt &acc, bool neg, const Fe &e0, const Fe &e1, const Fe &e2)
{
    if constexpr (TW) {
        const Fe ymx = fe_select(neg, e0, e1), ypx = fe_select(neg, e1, e0);
        return pt_madd_niels_tw(acc, ymx, ypx, fe_select(neg, e2, fe_neg_nr(e2)));
    } else {
        return pt_add_affine_cached(acc, fe_select(neg, e0, fe_neg_nr(e0)), e1, fe_select(neg, e2, fe_neg_nr(e2)));
    }
}

// acc += sign(digit) * G16[row][|digit|]
template <bool TW = false>
CAPY_HD inline Pt fb_add_digit(const Pt &acc, const uint32_t *gtab, int row, int digit)
{
    const bool neg = digit < 0;
    const int idx = neg ? -digit : digit;
    const uint32_t *e = gtab + (row * FB_TAB_ENTRIES + idx) * FB_ENTRY_DWORDS;
    const Fe e0 = load_fe(e), e1 = load_fe(e + 16), e2 = load_fe(e + 32);
    return fb_add_entry<TW>(acc, neg, e0, e1, e2);
}

// Hardened fixed base: a second shared table with FBCT_WBITS-bit signed windows -- few enough entries per row to read ALL
// of them per window and keep the wanted one.  The entries are the same for every lane of a wave, so they arrive by
// scalar loads (no vector memory traffic, no VGPRs) and each costs one v_bitop3_b32 per limb: 17 x 48 = 816 VALU per
// window at 5 bits, against ~2650 for the mixed addition that follows.  Windows x (addition + scan) is flat between 5 and
// 6 bits (90 x 3466 vs 75 x 4234) and worse at 4 (113 x 3082) and 7 (64 x 5770); r02 ran 4 bits with vector loads and
// AND/OR masking (2 VALU per limb and entry): 3.0x the indexed kernel, now profiles/r03_ed448_hardened.txt.
#ifndef CAPY_ED448_FBCT_WBITS
#define CAPY_ED448_FBCT_WBITS 5
#endif
constexpr int FBCT_WBITS = CAPY_ED448_FBCT_WBITS;
using FbCtWin = Win<FBCT_WBITS>;
constexpr int FBCT_ROWS = FbCtWin::NWIN + 1;
constexpr int FBCT_ENTRIES = FbCtWin::ENTRIES;
constexpr int FBCT_TABLE_DWORDS = FBCT_ROWS * FBCT_ENTRIES * FB_ENTRY_DWORDS;

CAPY_HD inline Pt fb_add_digit_ct(const Pt &acc, const uint32_t *__restrict__ gtab, int row, int digit)
{
    const bool neg = digit < 0;
    const uint32_t idx = (uint32_t)(neg ? -digit : digit);
    Fe x2 = fe_zero(), y2 = fe_zero(), td2 = fe_zero();
#pragma unroll 1
    for (uint32_t j = 0; j < (uint32_t)FBCT_ENTRIES; j++) {
        const uint32_t m = ct_mask(j == idx);
        // wave-uniform address; read through the constant address space so that the entry arrives by scalar loads
        // (s_load_dwordx16 into SGPRs: the table is never written while a kernel that uses it runs)
        const CtConstPtr e = (CtConstPtr)(gtab + ((size_t)row * FBCT_ENTRIES + j) * FB_ENTRY_DWORDS);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            x2.l[i] = ct_take(x2.l[i], e[i], m);
            y2.l[i] = ct_take(y2.l[i], e[16 + i], m);
            td2.l[i] = ct_take(td2.l[i], e[32 + i], m);
        }
    }
    x2 = fe_select(neg, x2, fe_neg_nr(x2));
    td2 = fe_select(neg, td2, fe_neg_nr(td2));
    return pt_add_affine_cached(acc, x2, y2, td2);
}

// [k]G with constant-address table lookups, from gtab[FBCT_TABLE_DWORDS]
CAPY_HD_INLINE Pt fb_scalarmul_ct(const uint8_t *k_be, const uint32_t *gtab)
{
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<FBCT_WBITS>(w, k);
    Pt acc = fb_add_digit_ct(pt_identity(), gtab, FbCtWin::NWIN, (int)top);
#pragma unroll 1
    for (int i = 0; i < FbCtWin::NWIN; i++) acc = fb_add_digit_ct(acc, gtab, i, sc_next_digit_lsb<FBCT_WBITS>(w));
    return acc;
}

// acc += sum over the FbWin::NWIN windows of the recoded scalar w (the top digit is the caller's business); with lds the
// entry of window i + 1 is on its way while window i is added
template <bool TW = false>
CAPY_HD_INLINE Pt fb_add_windows(Pt acc, uint32_t *w, const uint32_t *gtab, uint32_t *lds)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (lds) {
        int digit = sc_next_digit_lsb<FB_WBITS>(w);
        lds_prefetch<12>(lds, gtab + (0 * FB_TAB_ENTRIES + (digit < 0 ? -digit : digit)) * FB_ENTRY_DWORDS);
#pragma unroll 1
        for (int i = 0; i < FbWin::NWIN; i++) {
            const bool neg = digit < 0;
            lds_prefetch_wait();
            const Fe e0 = lds_load_fe(lds, 0), e1 = lds_load_fe(lds, 1), e2 = lds_load_fe(lds, 2);
            if (i + 1 < FbWin::NWIN) {
                digit = sc_next_digit_lsb<FB_WBITS>(w);
                lds_prefetch<12>(lds, gtab + ((i + 1) * FB_TAB_ENTRIES + (digit < 0 ? -digit : digit)) * FB_ENTRY_DWORDS);
            }
            acc = fb_add_entry<TW>(acc, neg, e0, e1, e2);
        }
        return acc;
    }
#endif
#pragma unroll 1
    for (int i = 0; i < FbWin::NWIN; i++) acc = fb_add_digit<TW>(acc, gtab, i, sc_next_digit_lsb<FB_WBITS>(w));
    return acc;
}

// [k]G from the shared table gtab[FB_TABLE_DWORDS]; lds = the wave's FB_PF_DWORDS of staging (or nullptr).
// TW: gtab is the twisted table and the result a point of E' (see fb_add_entry)
template <bool TW = false>
CAPY_HD_INLINE Pt fb_scalarmul(const uint8_t *k_be, const uint32_t *gtab, uint32_t *lds = nullptr)
{
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<FB_WBITS>(w, k);
    const Pt acc = fb_add_digit<TW>(pt_identity(), gtab, FbWin::NWIN, (int)top);
    return fb_add_windows<TW>(acc, w, gtab, lds);
}

// [a]G + [b]P: the variable-base window loop for [b]P, then [a]G added from the shared fixed-base table (39 mixed
// additions with 12-bit digits; interleaving 5-bit digits of `a` into the doubling chain, Straus style, costs 91).
CAPY_HD_INLINE Pt double_scalarmul(const uint8_t *a_be, const uint8_t *b_be, const Pt &P, uint32_t *tab,
                                   const uint32_t *gtab, uint32_t *lds = nullptr)
{
    Pt acc = vb_scalarmul(b_be, P, tab, lds);
    uint32_t ka[14], wa[15];
    sc_from_be(ka, a_be);
    const uint32_t topa = sc_recode_signed<FB_WBITS>(wa, ka);
    acc = fb_add_digit(acc, gtab, FbWin::NWIN, (int)topa);
    return fb_add_windows(acc, wa, gtab, lds);
}

// ------------------------------------------------------------------ scalars mod r (Schnorr / ECDHIES glue)
// r = 2^446 - 0x8335dc163bb124b65129c96fde933d8d723a70aadc873d6d54a7bb0d, as 14 LE words.
// The curve crate's Scalar ops (`mul_mod`, `*`, `-`; call sites /root/reference/src/ecc/keypair.rs:43,
// signable.rs:42,46,54, encryptable.rs:36,77) are taken as arithmetic mod r with reduced results
// (assumption (iii)).  Statically indexed register arrays throughout.
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

// 2^448 mod r = 4c for r = 2^446 - c: 226 bits, nine 28-bit limbs
CAPY_HD inline uint32_t sc_c4_limb(int i)
{
    constexpr uint32_t C4[9] = {0x29eec34u, 0x1cf5b55u, 0x9c2ab72u, 0xf635c8eu, 0x5bf7a4cu,
                                0xd944a72u, 0x8eec492u, 0x0cd7705u, 0x2u};
    return C4[i];
}

// 14 x 32-bit words -> 16 x 28-bit limbs and back
CAPY_HD inline void sc_to_limbs28(uint32_t l[16], const uint32_t w[14])
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int bit = 28 * i, j = bit >> 5, sh = bit & 31;
        uint64_t v = (uint64_t)w[j] >> sh;
        if (sh > 4 && j + 1 < 14) v |= (uint64_t)w[j + 1] << (32 - sh);
        l[i] = (uint32_t)v & 0x0fffffffu;
    }
}
CAPY_HD inline void sc_from_limbs28(uint32_t w[14], const uint32_t l[16])
{
#pragma unroll
    for (int j = 0; j < 14; j++) {
        const int bit = 32 * j, i = bit / 28, sh = bit - 28 * i;
        uint64_t v = (uint64_t)l[i] >> sh;
        v |= (uint64_t)l[i + 1] << (28 - sh);
        if (56 - sh < 32 && i + 2 < 16) v |= (uint64_t)l[i + 2] << (56 - sh);
        w[j] = (uint32_t)v;
    }
}

// out = a * b mod r (inputs: arbitrary 448-bit values).  Schoolbook product on 28-bit limbs (columns < 2^60), then
// the high half is folded down with 2^448 = 4c (mod r): 896 -> 675 -> 454 -> 449 -> 448 bits, and at most four
// conditional subtractions of r finish (2^448 < 5r).
CAPY_HD inline void sc_mul_mod(uint32_t out[14], const uint32_t a_in[14], const uint32_t b_in[14])
{
    constexpr uint32_t M = 0x0fffffffu;
    uint32_t a[16], b[16];
    sc_to_limbs28(a, a_in);
    sc_to_limbs28(b, b_in);
    uint32_t x[32];
    {
        uint64_t col[31];
#pragma unroll
        for (int k = 0; k < 31; k++) col[k] = 0;
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) col[i + j] += (uint64_t)a[i] * b[j];
        uint64_t c = 0;
#pragma unroll
        for (int k = 0; k < 31; k++) {
            const uint64_t v = col[k] + c;
            x[k] = (uint32_t)v & M;
            c = v >> 28;
        }
        x[31] = (uint32_t)c;  // < 2^28: the product is below 2^896
    }
    // fold 1: limbs 16..31 (448 bits) x 4c -> 25 limbs
    uint32_t y[26];
    {
        uint64_t col[26];
#pragma unroll
        for (int k = 0; k < 26; k++) col[k] = k < 16 ? x[k] : 0;
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) col[i + j] += (uint64_t)x[16 + i] * sc_c4_limb(j);
        uint64_t c = 0;
#pragma unroll
        for (int k = 0; k < 26; k++) {
            const uint64_t v = col[k] + c;
            y[k] = (uint32_t)v & M;
            c = v >> 28;
        }
    }
    // fold 2: limbs 16..25 (< 2^228) x 4c -> 19 limbs, + low 16 limbs
    uint32_t z[20];
    {
        uint64_t col[20];
#pragma unroll
        for (int k = 0; k < 20; k++) col[k] = k < 16 ? y[k] : 0;
#pragma unroll
        for (int i = 0; i < 10; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) col[i + j] += (uint64_t)y[16 + i] * sc_c4_limb(j);
        uint64_t c = 0;
#pragma unroll
        for (int k = 0; k < 20; k++) {
            const uint64_t v = col[k] + c;
            z[k] = (uint32_t)v & M;
            c = v >> 28;
        }
    }
    // folds 3 and 4: what is left above 2^448 is < 2^7, then 0 or 1 (a set bit after fold 3 means the low part
    // wrapped, so adding 4c once more cannot carry out again)
    uint32_t r16[17];
#pragma unroll
    for (int k = 0; k < 17; k++) r16[k] = z[k];
    // z[17..19] are zero: the value after fold 2 is below 2^448 + 2^(228+226)
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t h = r16[16];
        uint64_t c = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint64_t v = (uint64_t)r16[k] + (k < 9 ? (uint64_t)h * sc_c4_limb(k) : 0) + c;
            r16[k] = (uint32_t)v & M;
            c = v >> 28;
        }
        r16[16] = (uint32_t)c;
    }
    sc_from_limbs28(out, r16);
    sc_reduce(out);
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

// ---- Signable::sign's `bytes_to_scalar(k_bytes) * Scalar::from(4)` (/root/reference/src/ecc/signable.rs:46) and the
// `k - h.mul_mod(&s)` that consumes it (:54).  `*` and `-` are operators of the absent curve crate; what they do to a value
// that is not reduced mod r cannot be read off /root/reference (assumption (iii)), so the reading is a
// run-time choice (capy_ed448_set_scalar_star; the default is the one every other call site -- mul_mod -- spells out):
//   0  `*` is the product mod r:  k = 4 kb mod r,            z = (k - h s) mod r
//   1  `*` wraps at 2^448 (the U448 the Scalar wraps) and `-` is crypto-bigint's sub_mod applied to the unreduced value:
//      k = 4 kb mod 2^448,  z = k - hs  (+ r if that borrows);  U = [k]G with the unreduced k
//   2  `*` wraps at 2^448, `-` reduces: k = 4 kb mod 2^448,  z = (k - h s) mod r
// All three give signatures that verify (z G + h V = U either way); they differ in k, hence in U, h and z.
CAPY_HD inline void sc_star4(uint32_t out[14], const uint32_t a[14], int star)
{
    if (star == 0) {
        sc_mul4_mod(out, a);
        return;
    }
#pragma unroll
    for (int i = 13; i > 0; i--) out[i] = (a[i] << 2) | (a[i - 1] >> 30);
    out[0] = a[0] << 2;
}
CAPY_HD inline void sc_sign_z(uint32_t z[14], const uint32_t k[14], const uint32_t h[14], const uint32_t s[14], int star)
{
    uint32_t hs[14];
    sc_mul_mod(hs, h, s);
    if (star != 1) {
        sc_sub_mod(z, k, hs);  // reduces both operands first
        return;
    }
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint64_t v = (uint64_t)k[i] - hs[i] - borrow;
        z[i] = (uint32_t)v;
        borrow = (v >> 63) & 1;
    }
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint64_t v = (uint64_t)z[i] + (borrow ? sc_r_word(i) : 0u) + c;
        z[i] = (uint32_t)v;
        c = v >> 32;
    }
}

}  // namespace capy
