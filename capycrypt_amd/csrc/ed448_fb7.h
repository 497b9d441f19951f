// ed448_fb7.h — hardened fixed-base multiplication [k]G with the table lookups done by the MATRIX cores (device only; r03).
//
// Constant-address lookups mean that every entry of a window's table row is read and the wanted one kept, whatever the
// (secret) digit is.  fb_scalarmul_ct (ed448_algo.h) does the keeping on the VALU: one v_bitop3_b32 per limb and entry,
// 48 x 17 = 816 instructions per 5-bit window on top of a 2650-instruction mixed addition, 90 windows.  Wider windows
// mean fewer additions but a longer scan, and the product is flat (ed448_algo.h) -- as long as the scan is VALU work.
//
// A table row is the same for all 64 items of a wave, and "keep entry idx[n] for item n" is a matrix product with a
// one-hot matrix:  D[m][n] = sum_k T[k][m] * (idx[n] == k)  =  T[idx[n]][m].  With bytes as the elements that is
// v_mfma_i32_16x16x64_i8 -- 64 table entries (k) x 16 bytes (m) x 16 items (n) per instruction, exact in int32, on a
// pipe that runs beside the VALU.  So: 7-bit signed windows (|digit| - 1 = k in 0..63; digit 0 selects nothing, which
// leaves all-zero bytes = the affine cached identity once y's limb 0 is set to 1), 65 windows instead of 90, and per
// window 48 MFMAs (4 groups of 16 items x 12 groups of 16 bytes of the 192-byte entry) instead of 816 VALU selects.
// Register r of lane l of a result holds byte 4 (l / 16) + r of the byte group for item l % 16 (+ 16 per item group) --
// four consecutive bytes = one 28-bit limb, packed with two v_perm_b32 and one v_lshl_or_b32 and handed to the lane
// that owns the item through LDS (row stride 68 dwords: conflict-free 16-byte reads).  Operand layouts checked on the
// hardware by tools/probe_mfma_onehot.hip.  No address, branch or instruction count depends on the scalar: the digits
// only enter as VALU data (the one-hot bytes, the negation mask, the identity fix-up).
//
// Table (built once per device from the ordinary affine cached entries, ed448.hip: build_gtab7): for row r (window r of
// the recoded scalar, row 64 = the recoding carry), byte group mb, lane l = 16 g + c, slot s:
//   gt7[((r * 12 + mb) * 64 + l) * 16 + s] = byte 16 mb + c of entry (16 g + s + 1) * 2^(7 r) * G      (798 720 bytes)
// (with 8-bit windows: ((r * 2 + kb) * 12 + mb) and entry 64 kb + 16 g + s + 1, 1 400 832 bytes)
// which is exactly the A operand of the instruction for (r, mb): one 16-byte load per lane.
#pragma once
#include "ed448_algo.h"

namespace capy {

// CAPY_ED448_FB7_WBITS = 7 (default) or 8: 8-bit windows take two MFMAs per product (128 magnitudes = two K blocks of 64,
// the second accumulating onto the first) for 57 additions instead of 65; measured in profiles/r03_ed448_fb7_mfma.txt
#ifndef CAPY_ED448_FB7_WBITS
#define CAPY_ED448_FB7_WBITS 7
#endif
constexpr int FB7_WBITS = CAPY_ED448_FB7_WBITS;
static_assert(FB7_WBITS == 7 || FB7_WBITS == 8, "one or two K blocks of 64 entries");
using Fb7Win = Win<FB7_WBITS>;
constexpr int FB7_ROWS = Fb7Win::NWIN + 1;    // the windows + the recoding carry
constexpr int FB7_ENTRIES = Fb7Win::ENTRIES;  // 0 .. 2^(W-1) times the row's base point (the linear table build_gtab7 starts from)
constexpr int FB7_KBLOCKS = Fb7Win::HALF / 64;  // MFMAs per product: each sees 64 entries
constexpr int FB7_GROUPS = 12;                // 192 bytes per entry / 16
constexpr size_t FB7_TABLE_BYTES = (size_t)FB7_ROWS * FB7_KBLOCKS * FB7_GROUPS * 64 * 16;
constexpr int FB7_XPOSE_STRIDE = 68;          // dwords per item in the LDS hand-over (48 used)
// the LDS area serves as the staging of the next row (KBLOCKS x 12 KiB) and then as the hand-over (64 x 68 dwords)
constexpr int FB7_LDS_DWORDS = FB7_KBLOCKS * FB7_GROUPS * 256 > 64 * FB7_XPOSE_STRIDE ? FB7_KBLOCKS * FB7_GROUPS * 256 : 64 * FB7_XPOSE_STRIDE;

#if defined(__HIP_DEVICE_COMPILE__)
typedef int fb7_v4i __attribute__((ext_vector_type(4)));

// row `row` of the table on its way into the hand-over area (12 x 1 KiB, no VGPRs; ed448_algo.h: lds_prefetch)
__device__ __forceinline__ void fb7_request_row(const uint8_t *__restrict__ gt7, int row, uint32_t *xpose)
{
    const uint32_t lane = threadIdx.x & 63;
    lds_prefetch<FB7_KBLOCKS * FB7_GROUPS, 256>(xpose, reinterpret_cast<const uint32_t *>(gt7) +
                                                          ((size_t)row * FB7_KBLOCKS * FB7_GROUPS * 64 + lane) * 4);
}

// the affine cached entry  sign(digit) * |digit| * 2^(7 row) * G  for every lane's own digit, selected by the matrix cores
// (the row itself was requested by fb7_request_row; next_row >= 0: request that one before returning)
// TW: the table holds twisted entries (y - x, y + x, 2 d' x y): the identity is (1, 1, 0), a negative swaps the first two
template <bool TW>
__device__ __forceinline__ void fb7_select(const uint8_t *__restrict__ gt7, int next_row, int digit, uint32_t *xpose, Fe &x2, Fe &y2, Fe &td2)
{
    const uint32_t lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t neg = (uint32_t)(digit >> 31);            // all ones for a negative digit
    const uint32_t mag = ((uint32_t)digit ^ neg) - neg;      // |digit|, 0 .. 64
    // A operands: this row's 12 byte groups (the address depends on the row only).  They were requested memory -> LDS
    // (fb7_request_row) before the previous addition and sit in the hand-over area, which is free until the first
    // result is written below
    fb7_v4i a[FB7_KBLOCKS][FB7_GROUPS];
    lds_prefetch_wait();
#pragma unroll
    for (int kb = 0; kb < FB7_KBLOCKS; kb++)
#pragma unroll
        for (int mb = 0; mb < FB7_GROUPS; mb++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(xpose + ((kb * FB7_GROUPS + mb) * 64 + lane) * 4);
            a[kb][mb] = fb7_v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        }
    __syncthreads();  // every lane has its operands before the area is overwritten
    // B operands: one-hot bytes.  Lane (g, c) holds, for the item group ib and the K block kb, the k-slots
    // 64 kb + 16 g .. + 15 of item 16 ib + c
#pragma unroll
    for (int ib = 0; ib < 4; ib++) {
        const uint32_t m = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * (16 * ib + c)), (int)mag);
        fb7_v4i b[FB7_KBLOCKS];
#pragma unroll
        for (int kb = 0; kb < FB7_KBLOCKS; kb++) {
            const uint32_t t = m - 1u - 64u * kb - 16u * g;  // slot within my 16, if < 16 (wraps to a huge value otherwise)
            const uint32_t one = 1u << ((t & 3u) * 8u);
#pragma unroll
            for (int q = 0; q < 4; q++) b[kb][q] = (int)(one & (0u - (uint32_t)((t >> 2) == (uint32_t)q)));
        }
        // four products in flight before the first is packed (a result is readable ~8 cycles after its issue: packed
        // one by one every MFMA would be followed by that many idle cycles)
#pragma unroll
        for (int mb0 = 0; mb0 < FB7_GROUPS; mb0 += 4) {
            fb7_v4i d[4];
#pragma unroll
            for (int j = 0; j < 4; j++) d[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0][mb0 + j], b[0], fb7_v4i{0, 0, 0, 0}, 0, 0, 0);
            if constexpr (FB7_KBLOCKS == 2) {
#pragma unroll
                for (int j = 0; j < 4; j++) d[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1][mb0 + j], b[1], d[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // bytes 4 g .. 4 g + 3 of group mb = limb dword 4 mb + g of item 16 ib + c
                const uint32_t lo = __builtin_amdgcn_perm((uint32_t)d[j][1], (uint32_t)d[j][0], 0x0c0c0400u);
                const uint32_t hi = __builtin_amdgcn_perm((uint32_t)d[j][3], (uint32_t)d[j][2], 0x0c0c0400u);
                xpose[(16 * ib + c) * FB7_XPOSE_STRIDE + 4 * (mb0 + j) + g] = (hi << 16) | lo;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();  // one wave per block: orders the LDS writes above before the reads below
    const uint32_t *mine = xpose + lane * FB7_XPOSE_STRIDE;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint4 vx = *reinterpret_cast<const uint4 *>(mine + 4 * q);
        const uint4 vy = *reinterpret_cast<const uint4 *>(mine + 16 + 4 * q);
        const uint4 vt = *reinterpret_cast<const uint4 *>(mine + 32 + 4 * q);
        x2.l[4 * q] = vx.x, x2.l[4 * q + 1] = vx.y, x2.l[4 * q + 2] = vx.z, x2.l[4 * q + 3] = vx.w;
        y2.l[4 * q] = vy.x, y2.l[4 * q + 1] = vy.y, y2.l[4 * q + 2] = vy.z, y2.l[4 * q + 3] = vy.w;
        td2.l[4 * q] = vt.x, td2.l[4 * q + 1] = vt.y, td2.l[4 * q + 2] = vt.z, td2.l[4 * q + 3] = vt.w;
    }
    __syncthreads();  // the next window's writes come after these reads
    if (next_row >= 0) fb7_request_row(gt7, next_row, xpose);  // wave-uniform; lands while the addition runs
    const uint32_t is0 = (uint32_t)(mag == 0);  // digit 0 selected nothing: all zeros -> the identity entry
    const Fe nt = fe_neg_nr(td2);
    if constexpr (TW) {
        x2.l[0] |= is0;  // (1, 1, 0)
        y2.l[0] |= is0;
        // -(x, y) = (-x, y): y - x and y + x change places, 2 d' x y changes sign (limb-wise selects, no branch)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t a = x2.l[i], b = y2.l[i];
            x2.l[i] = (b & neg) | (a & ~neg);
            y2.l[i] = (a & neg) | (b & ~neg);
            td2.l[i] = (nt.l[i] & neg) | (td2.l[i] & ~neg);
        }
    } else {
        y2.l[0] |= is0;  // (0, 1, 0)
        // -(x, y) = (-x, y): negate x and d x y under the sign mask
        const Fe nx = fe_neg_nr(x2);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            x2.l[i] = (nx.l[i] & neg) | (x2.l[i] & ~neg);
            td2.l[i] = (nt.l[i] & neg) | (td2.l[i] & ~neg);
        }
    }
}

// [k]G: 65 mixed additions, every table byte of every row read by every wave
// TW: on the twisted curve (7M additions; the result is a point of E', to be finished by pt_tw_to_affine_bytes)
template <bool TW>
__device__ __forceinline__ Pt fb7_scalarmul(const uint8_t *k_be, const uint8_t *__restrict__ gt7, uint32_t *xpose)
{
    uint32_t k[14], w[15];
    sc_from_be(k, k_be);
    const uint32_t top = sc_recode_signed<FB7_WBITS>(w, k);
    Fe x2, y2, td2;
    fb7_request_row(gt7, Fb7Win::NWIN, xpose);
    fb7_select<TW>(gt7, 0, (int)top, xpose, x2, y2, td2);
    Pt acc = TW ? pt_madd_niels_tw(pt_identity(), x2, y2, td2) : pt_add_affine_cached(pt_identity(), x2, y2, td2);
#pragma unroll 1
    for (int i = 0; i < Fb7Win::NWIN; i++) {
        fb7_select<TW>(gt7, i + 1 < Fb7Win::NWIN ? i + 1 : -1, sc_next_digit_lsb<FB7_WBITS>(w), xpose, x2, y2, td2);
        acc = TW ? pt_madd_niels_tw(acc, x2, y2, td2) : pt_add_affine_cached(acc, x2, y2, td2);
    }
    return acc;
}
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace capy
