// ed448.hip — Ed448 kernels, their C-ABI entry points, and the src/ecc protocol glue
// (KeyPair::new, Signable, KeyEncryptable) composed from the sponge and curve kernels on device buffers.
// No CPU fallback: every scalar multiplication, KMAC and mod-r operation of the data path runs on the GPU.
#include <string.h>
#include <atomic>
#include <mutex>
#include "common.h"
#include "occupancy.h"
#include "ed448_algo.h"
#include "ed448_wave.h"
#include "ed448_fb7.h"
#include "ed448_quad.h"
#include "ed448_duo.h"
#include "sponge_host.h"

namespace capy {


// ------------------------------------------------------------------ kernels (one item per lane)
#ifndef CAPY_ED448_WAVES
#define CAPY_ED448_WAVES 2
#endif
#ifndef CAPY_ED448_WAVE_MAX_PER_SIMD
#define CAPY_ED448_WAVE_MAX_PER_SIMD 8  // one item per wave up to 8 items per SIMD: 8192 on MI355X
#endif
#ifdef CAPY_ED448_NUMVGPR
__attribute__((amdgpu_num_vgpr(CAPY_ED448_NUMVGPR)))
#endif
// Table entries fetched ahead through LDS (ed448_algo.h: lds_prefetch) in the lane-per-item kernels that run at ONE wave
// per SIMD (the two-items-per-lane kernels run two, which hide each other's waits).  Measured at 65 536 items
// (profiles/r03_ed448_prefetch.txt): fixed base 0.327 -> 0.299 ms, on by default; variable base 2.886 -> 2.984 ms (the
// loop's register allocation gets worse: 96 instead of 44 spilled VGPRs; with the entry's operands read from LDS one by
// one inside the addition 55, and 2.72 ms either way), off by default.
#ifndef CAPY_ED448_PREFETCH_FB
#define CAPY_ED448_PREFETCH_FB 1
#endif
#ifndef CAPY_ED448_PREFETCH_VB
#define CAPY_ED448_PREFETCH_VB 0
#endif
// double_scalarmul (verify): both parts fetch ahead.  With the entry's operands read from LDS one by one inside the
// addition the kernel spills 123 instead of 172 VGPRs: 2^16 verifications 3.65 -> 3.54 ms (the variable-base kernel alone:
// 2.72 ms either way at one wave per SIMD, 4.80 -> 4.91 at two)
#ifndef CAPY_ED448_PREFETCH_DSM
#define CAPY_ED448_PREFETCH_DSM 1
#endif
// One item per lane.  Each of vb_kernel / vb_ct_kernel / dsm_kernel exists twice: the plain form (two waves per SIMD fit) for
// batches beyond one wave per SIMD, and a *_1w form compiled for exactly one wave per SIMD (CAPY_WAVES_PER_SIMD, occupancy.h)
// for batches of up to 64 items per SIMD, whose launch time is one wave's chain and must not double because the dispatcher
// put two waves on one SIMD.
__device__ __forceinline__ void vb_body(uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride, const uint8_t *points_xy,
                                        uint64_t point_stride, uint8_t *out_xy, uint32_t *table_ws, uint32_t *pf)
{
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const Pt P = pt_from_affine_bytes(points_xy + i * point_stride);
    const Pt r = vb_scalarmul(scalars_be + i * scalar_stride, P, table_ws + i * VB_TABLE_DWORDS, pf);
    pt_to_affine_bytes(out_xy + i * 112, r);
}
#if CAPY_ED448_PREFETCH_VB
#define CAPY_VB_PF __shared__ uint32_t pf[VB_PF_DWORDS];
#else
#define CAPY_VB_PF uint32_t *const pf = nullptr;
#endif
__global__ __launch_bounds__(64, CAPY_ED448_WAVES) void vb_kernel(uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride,
                                                const uint8_t *points_xy, uint64_t point_stride, uint8_t *out_xy,
                                                uint32_t *table_ws)
{
    CAPY_VB_PF
    vb_body(n, scalars_be, scalar_stride, points_xy, point_stride, out_xy, table_ws, pf);
}
__global__ __launch_bounds__(64) CAPY_WAVES_PER_SIMD(1) void vb_kernel_1w(
    uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride, const uint8_t *points_xy, uint64_t point_stride, uint8_t *out_xy,
    uint32_t *table_ws)
{
    CAPY_VB_PF
    vb_body(n, scalars_be, scalar_stride, points_xy, point_stride, out_xy, table_ws, pf);
}

// The kernels below are chosen for batches that put at most ONE wave on a SIMD, where the time of the launch is the chain of
// one wave -- if the dispatcher puts two of them on one SIMD (it does when the launch follows a kernel whose waves end
// staggered: 32 768 items in the two-lane form took 3.3 instead of 1.8 ms behind a 196 608-item launch,
// profiles/r04_ed448_remainder.txt) the launch takes twice as long.  amdgpu_waves_per_eu(1, 1) rounds the register
// allocation up so that a second wave of the SAME kernel does not fit on the SIMD.
#define CAPY_ONE_WAVE_PER_SIMD CAPY_WAVES_PER_SIMD(1)

// four lanes per item (ed448_quad.h): batches between the one-item-per-wave and the one-item-per-lane kernels
__global__ __launch_bounds__(64) CAPY_ONE_WAVE_PER_SIMD void vb_quad_kernel(uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride,
                                                        const uint8_t *points_xy, uint64_t point_stride, uint8_t *out_xy,
                                                        uint32_t *table_ws)
{
#if defined(__HIP_DEVICE_COMPILE__)  // (the quad primitives are DPP: device pass only)
    const uint64_t slot = ((uint64_t)blockIdx.x * 64 + threadIdx.x) >> 2;  // the quad's own table slot (n rounded up to 16 of them)
    const uint64_t i = slot < n ? slot : n - 1;                            // quads past the batch redo the last item, write nothing
    const uint32_t q = threadIdx.x & 3;
    const Fe r = quad::scalarmul(scalars_be + i * scalar_stride, points_xy + i * point_stride, table_ws + slot * VB_TABLE_DWORDS, q);
    // affine: every lane inverts Z (a serial chain either way), lanes 0 and 1 write x and y
    const Fe zi = fe_inv_out(quad::fe_perm<2, 2, 2, 2>(r));
    const Fe c = fe_mul(r, zi);
    if (q < 2 && slot < n) fe_to_bytes(out_xy + i * 112 + q * 56, c);
#endif
}

// [a]G + [b]P with four lanes per item (the shape of verify, /root/reference/src/ecc/signable.rs:77)
__global__ __launch_bounds__(64) CAPY_ONE_WAVE_PER_SIMD void dsm_quad_kernel(uint64_t n, const uint8_t *a_be, const uint8_t *b_be, const uint8_t *points_xy,
                                                         uint8_t *out_xy, uint32_t *table_ws, const uint32_t *gtab)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t slot = ((uint64_t)blockIdx.x * 64 + threadIdx.x) >> 2;
    const uint64_t i = slot < n ? slot : n - 1;
    const uint32_t q = threadIdx.x & 3;
    Fe r = quad::scalarmul(b_be + i * 56, points_xy + i * 112, table_ws + slot * VB_TABLE_DWORDS, q);
    r = quad::add_fixed_base(r, a_be + i * 56, gtab, q);
    const Fe zi = fe_inv_out(quad::fe_perm<2, 2, 2, 2>(r));
    const Fe c = fe_mul(r, zi);
    if (q < 2 && slot < n) fe_to_bytes(out_xy + i * 112 + q * 56, c);
#endif
}

// four lanes per item with constant-address lookups, the table in LDS (ed448_quad.h): secret scalars, 4 k .. 32 k items
__global__ __launch_bounds__(64, 1) void vb_quad_ct_kernel(uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride,
                                                           const uint8_t *points_xy, uint64_t point_stride, uint8_t *out_xy)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t tab[quad::QUAD_CT_LDS_DWORDS];
    const uint64_t slot = ((uint64_t)blockIdx.x * 64 + threadIdx.x) >> 2;
    const uint64_t i = slot < n ? slot : n - 1;  // quads past the batch redo the last item, write nothing
    const uint32_t q = threadIdx.x & 3;
    const Fe r = quad::scalarmul_ct(scalars_be + i * scalar_stride, points_xy + i * point_stride, tab, q);
    const Fe zi = fe_inv_out(quad::fe_perm<2, 2, 2, 2>(r));
    const Fe c = fe_mul(r, zi);
    if (q < 2 && slot < n) fe_to_bytes(out_xy + i * 112 + q * 56, c);
#endif
}

// two lanes per item (ed448_duo.h): 16 k .. 32 k items, one wave of 32 items per SIMD at 32 768
__global__ __launch_bounds__(64) CAPY_ONE_WAVE_PER_SIMD void vb_duo_kernel(uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride,
                                                       const uint8_t *points_xy, uint64_t point_stride, uint8_t *out_xy,
                                                       uint32_t *table_ws)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t slot = ((uint64_t)blockIdx.x * 64 + threadIdx.x) >> 1;  // the pair's own table slot (n rounded up to 32 of them)
    const uint64_t i = slot < n ? slot : n - 1;                            // pairs past the batch redo the last item, write nothing
    const bool p = threadIdx.x & 1;
    const duo::Half r = duo::scalarmul(scalars_be + i * scalar_stride, points_xy + i * point_stride, table_ws + slot * VB_TABLE_DWORDS, p);
    duo::store_affine(out_xy + i * 112, r, p, slot < n);
#endif
}

// two lanes per item with constant-address lookups, the table half in registers and half in LDS (ed448_duo.h): secret scalars,
// 16 k .. 32 k items in ONE round of waves
__global__ __launch_bounds__(64) CAPY_ONE_WAVE_PER_SIMD void vb_duo_ct_kernel(uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride,
                                                          const uint8_t *points_xy, uint64_t point_stride, uint8_t *out_xy)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t tab[duo::DUO_CT_LDS_DWORDS];
    const uint64_t slot = ((uint64_t)blockIdx.x * 64 + threadIdx.x) >> 1;
    const uint64_t i = slot < n ? slot : n - 1;  // pairs past the batch redo the last item, write nothing
    const bool p = threadIdx.x & 1;
    const duo::Half r = duo::scalarmul_ct(scalars_be + i * scalar_stride, points_xy + i * point_stride, tab, p);
    duo::store_affine(out_xy + i * 112, r, p, slot < n);
#endif
}

__global__ __launch_bounds__(64) CAPY_ONE_WAVE_PER_SIMD void dsm_duo_kernel(uint64_t n, const uint8_t *a_be, const uint8_t *b_be, const uint8_t *points_xy,
                                                        uint8_t *out_xy, uint32_t *table_ws, const uint32_t *gtab)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t slot = ((uint64_t)blockIdx.x * 64 + threadIdx.x) >> 1;
    const uint64_t i = slot < n ? slot : n - 1;
    const bool p = threadIdx.x & 1;
    duo::Half r = duo::scalarmul(b_be + i * 56, points_xy + i * 112, table_ws + slot * VB_TABLE_DWORDS, p);
    r = duo::add_fixed_base(r, a_be + i * 56, gtab, p);
    duo::store_affine(out_xy + i * 112, r, p, slot < n);
#endif
}

// hardened form: constant-address table lookups (ed448_algo.h: vb_add_digit_ct); also serves [k]G with point_stride 0
__device__ __forceinline__ void vb_ct_body(uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride, const uint8_t *points_xy,
                                           uint64_t point_stride, uint8_t *out_xy, uint32_t *table_ws)
{
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const Pt P = pt_from_affine_bytes(points_xy + i * point_stride);
    // the wave's table, interleaved across its 64 lanes (the scratch is sized in whole waves by the launcher)
    const CtTable t = {table_ws + (uint64_t)blockIdx.x * 64 * VB_TABLE_DWORDS, threadIdx.x, 64};
    const Pt r = vb_scalarmul_ct(scalars_be + i * scalar_stride, P, t);
    pt_to_affine_bytes(out_xy + i * 112, r);
}
__global__ __launch_bounds__(64, CAPY_ED448_WAVES) void vb_ct_kernel(uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride,
                                                   const uint8_t *points_xy, uint64_t point_stride, uint8_t *out_xy,
                                                   uint32_t *table_ws)
{
    vb_ct_body(n, scalars_be, scalar_stride, points_xy, point_stride, out_xy, table_ws);
}
__global__ __launch_bounds__(64) CAPY_WAVES_PER_SIMD(1) void vb_ct_kernel_1w(
    uint64_t n, const uint8_t *scalars_be, uint64_t scalar_stride, const uint8_t *points_xy, uint64_t point_stride, uint8_t *out_xy,
    uint32_t *table_ws)
{
    vb_ct_body(n, scalars_be, scalar_stride, points_xy, point_stride, out_xy, table_ws);
}

__global__ __launch_bounds__(64, CAPY_ED448_WAVES) void fb_ct_kernel(uint64_t n, const uint8_t *scalars_be, uint8_t *out_xy,
                                                   const uint32_t *gtab_ct)
{
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    pt_to_affine_bytes(out_xy + i * 112, fb_scalarmul_ct(scalars_be + i * 56, gtab_ct));
}

// hardened fixed base with the table lookups on the matrix cores (ed448_fb7.h).  All 64 lanes stay in the loop (an MFMA
// wants the whole wave): the lanes past the end of the batch repeat the last item and do not store.
// TW: the table is the twisted one (7M additions on E', back through the dual isogeny: ed448_dev.h)
template <bool TW>
__global__ __launch_bounds__(64, CAPY_ED448_WAVES) void fb_ct7_kernel(uint64_t n, const uint8_t *scalars_be, uint8_t *out_xy,
                                                    const uint8_t *gt7)
{
#if defined(__HIP_DEVICE_COMPILE__)  // (the MFMA builtins exist in the device pass only)
    __shared__ uint32_t xpose[FB7_LDS_DWORDS];
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint64_t src = i < n ? i : n - 1;
    const Pt r = fb7_scalarmul<TW>(scalars_be + src * 56, gt7, xpose);
    if (i < n) {
        if constexpr (TW)
            pt_tw_to_affine_bytes(out_xy + i * 112, r);
        else
            pt_to_affine_bytes(out_xy + i * 112, r);
    }
#endif
}

// the same with two items per lane sharing one inversion (as fb2_kernel): lane l takes the items base + l and
// base + 64 + l of the wave's 128.  The first result waits in global scratch (park: 48 dwords per lane, 16-byte pieces
// interleaved across the wave's lanes) -- the LDS is taken by the hand-over area and 48 more VGPRs would spill.
template <bool TW>
__global__ __launch_bounds__(64, CAPY_ED448_WAVES) void fb_ct7_pair_kernel(uint64_t n, const uint8_t *scalars_be, uint8_t *out_xy,
                                                         const uint8_t *gt7, uint32_t *park)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t xpose[FB7_LDS_DWORDS];
    const uint64_t base = (uint64_t)blockIdx.x * 128 + threadIdx.x;
    const uint64_t i0 = base < n ? base : n - 1, i1 = base + 64 < n ? base + 64 : i0;
    uint4 *mine = reinterpret_cast<uint4 *>(park) + (uint64_t)blockIdx.x * 12 * 64 + threadIdx.x;
    {
        const Pt r0 = fb7_scalarmul<TW>(scalars_be + i0 * 56, gt7, xpose);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            mine[q * 64] = uint4{r0.X.l[4 * q], r0.X.l[4 * q + 1], r0.X.l[4 * q + 2], r0.X.l[4 * q + 3]};
            mine[(4 + q) * 64] = uint4{r0.Y.l[4 * q], r0.Y.l[4 * q + 1], r0.Y.l[4 * q + 2], r0.Y.l[4 * q + 3]};
            mine[(8 + q) * 64] = uint4{r0.Z.l[4 * q], r0.Z.l[4 * q + 1], r0.Z.l[4 * q + 2], r0.Z.l[4 * q + 3]};
        }
    }
    const Pt r1 = fb7_scalarmul<TW>(scalars_be + i1 * 56, gt7, xpose);
    Pt r0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint4 vx = mine[q * 64], vy = mine[(4 + q) * 64], vz = mine[(8 + q) * 64];
        r0.X.l[4 * q] = vx.x, r0.X.l[4 * q + 1] = vx.y, r0.X.l[4 * q + 2] = vx.z, r0.X.l[4 * q + 3] = vx.w;
        r0.Y.l[4 * q] = vy.x, r0.Y.l[4 * q + 1] = vy.y, r0.Y.l[4 * q + 2] = vy.z, r0.Y.l[4 * q + 3] = vy.w;
        r0.Z.l[4 * q] = vz.x, r0.Z.l[4 * q + 1] = vz.y, r0.Z.l[4 * q + 2] = vz.z, r0.Z.l[4 * q + 3] = vz.w;
    }
    r0.T = fe_zero();  // not needed for the conversion
    // No predicate on the stores: lanes past the batch hold the last item (i0) and a missing second item repeats the first
    // (i1 = i0), so they write bytes that are already there.  (With the stores under two nested lane masks the register
    // allocator parks the second store's address in a register that held scalar-derived limbs on the path around the first
    // block: harmless -- those lanes are off -- but the static constant-address check cannot tell, tools/ct_taint.py.)
    if constexpr (TW)
        pt_tw_pair_to_affine_bytes(out_xy + i0 * 112, out_xy + i1 * 112, r0, r1);
    else
        pt_pair_to_affine_bytes(out_xy + i0 * 112, out_xy + i1 * 112, r0, r1);
#endif
}

// the linear table (rows x 65 affine cached entries of 48 limb dwords) re-laid as MFMA A operands (ed448_fb7.h)
__global__ void gtab7_pack_kernel(const uint32_t *lin, uint32_t *gt7_words)
{
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;  // output dword: (((row * KBLOCKS + kb) * 12 + mb) * 64 + lane) * 4 + sq
    if (o >= FB7_TABLE_BYTES / 4) return;
    const uint32_t sq = o & 3, lane = (o >> 2) & 63, rm = o >> 8, mb = rm % FB7_GROUPS, rk = rm / FB7_GROUPS;
    const uint32_t kb = rk % FB7_KBLOCKS, row = rk / FB7_KBLOCKS;
    const uint32_t g = lane >> 4, c = lane & 15, bi = 16 * mb + c;
    uint32_t v = 0;
    for (uint32_t t = 0; t < 4; t++) {
        const uint32_t entry = 64 * kb + 16 * g + 4 * sq + t + 1;
        const uint32_t w = lin[((size_t)row * FB7_ENTRIES + entry) * FB_ENTRY_DWORDS + bi / 4];
        v |= ((w >> (8 * (bi & 3))) & 0xffu) << (8 * t);
    }
    gt7_words[o] = v;
}

template <bool TW>  // TW: gtab is the twisted table (7M additions on E', back through the dual isogeny: ed448_dev.h)
__global__ __launch_bounds__(64, CAPY_ED448_WAVES) void fb_kernel(uint64_t n, const uint8_t *scalars_be, uint8_t *out_xy,
                                                const uint32_t *gtab)
{
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
#if CAPY_ED448_PREFETCH_FB
    __shared__ uint32_t pf[FB_PF_DWORDS];
#else
    uint32_t *const pf = nullptr;
#endif
    const Pt r = fb_scalarmul<TW>(scalars_be + i * 56, gtab, pf);
    if constexpr (TW)
        pt_tw_to_affine_bytes(out_xy + i * 112, r);
    else
        pt_to_affine_bytes(out_xy + i * 112, r);
}

// Two items per lane sharing one inversion: PtXYZ / park_xyz / unpark_xyz (ed448_algo.h).  The variable-base kernel of that
// shape, vb2_kernel, is compiled in its own translation unit (ed448_vb2.hip) with pinned multiply-add chains.
int vb2_launch(size_t n, const uint8_t *scalars, uint64_t scalar_stride, const uint8_t *points, uint64_t point_stride, uint8_t *out,
               uint32_t *table_ws, hipStream_t s);

template <bool CT, bool TW = false>  // CT: gtab is the hardened 5-bit table (VALU scan); TW (indexed only): the twisted table
__global__ __launch_bounds__(64, CAPY_ED448_WAVES) void fb2_kernel(uint64_t n, const uint8_t *scalars_be, uint8_t *out_xy,
                                                 const uint32_t *gtab)
{
    static_assert(!(CT && TW), "the VALU-scan table stays on E");
    __shared__ PtXYZ parked;
    const uint64_t base = (uint64_t)blockIdx.x * 128 + threadIdx.x;
    if (base >= n) return;
    Pt r = pt_identity();
#pragma unroll 1
    for (int j = 0; j < 2; j++) {
        const uint64_t i = (j == 1 && base + 64 < n) ? base + 64 : base;
        if constexpr (CT)
            r = fb_scalarmul_ct(scalars_be + i * 56, gtab);
        else
            r = fb_scalarmul<TW>(scalars_be + i * 56, gtab);
        if (j == 0) park_xyz(parked, r);
    }
    const Pt r0 = unpark_xyz(parked);
    if (base + 64 < n) {
        if constexpr (TW)
            pt_tw_pair_to_affine_bytes(out_xy + base * 112, out_xy + (base + 64) * 112, r0, r);
        else
            pt_pair_to_affine_bytes(out_xy + base * 112, out_xy + (base + 64) * 112, r0, r);
    } else {
        if constexpr (TW)
            pt_tw_to_affine_bytes(out_xy + base * 112, r0);
        else
            pt_to_affine_bytes(out_xy + base * 112, r0);
    }
}

__device__ __forceinline__ void dsm_body(uint64_t n, const uint8_t *a_be, const uint8_t *b_be, const uint8_t *points_xy, uint8_t *out_xy,
                                         uint32_t *table_ws, const uint32_t *gtab, uint32_t *pf)
{
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const Pt P = pt_from_affine_bytes(points_xy + i * 112);
    const Pt r = double_scalarmul(a_be + i * 56, b_be + i * 56, P, table_ws + i * VB_TABLE_DWORDS, gtab, pf);
    pt_to_affine_bytes(out_xy + i * 112, r);
}
#if CAPY_ED448_PREFETCH_DSM
#define CAPY_DSM_PF __shared__ uint32_t pf[VB_PF_DWORDS];  /* serves the variable-base part, then the fixed-base part */
#else
#define CAPY_DSM_PF uint32_t *const pf = nullptr;
#endif
__global__ __launch_bounds__(64, CAPY_ED448_WAVES) void dsm_kernel(uint64_t n, const uint8_t *a_be, const uint8_t *b_be,
                                                 const uint8_t *points_xy, uint8_t *out_xy, uint32_t *table_ws,
                                                 const uint32_t *gtab)
{
    CAPY_DSM_PF
    dsm_body(n, a_be, b_be, points_xy, out_xy, table_ws, gtab, pf);
}
__global__ __launch_bounds__(64) CAPY_WAVES_PER_SIMD(1) void dsm_kernel_1w(
    uint64_t n, const uint8_t *a_be, const uint8_t *b_be, const uint8_t *points_xy, uint8_t *out_xy, uint32_t *table_ws,
    const uint32_t *gtab)
{
    CAPY_DSM_PF
    dsm_body(n, a_be, b_be, points_xy, out_xy, table_ws, gtab, pf);
}

__global__ __launch_bounds__(64) void add_kernel(uint64_t n, const uint8_t *p_xy, const uint8_t *q_xy, uint8_t *out_xy)
{
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    pt_to_affine_bytes(out_xy + i * 112, pt_add(pt_from_affine_bytes(p_xy + i * 112), pt_from_affine_bytes(q_xy + i * 112)));
}

__global__ __launch_bounds__(64) void validate_kernel(uint64_t n, const uint8_t *xy, int32_t *status)
{
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    status[i] = pt_validate_bytes(xy + i * 112) ? CAPY_ITEM_OK : CAPY_ITEM_FAIL;
}

// affine (x, y) -> fixed-base table entry (x, y, d*x*y) in limbs
__global__ void gtab_pack_kernel(uint32_t n, const uint8_t *xy, uint32_t *gtab)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe x = fe_from_bytes(xy + (uint64_t)i * 112), y = fe_from_bytes(xy + (uint64_t)i * 112 + 56);
    uint32_t *e = gtab + (uint64_t)i * FB_ENTRY_DWORDS;
    store_fe(e, x);
    store_fe(e + 16, y);
    store_fe(e + 32, fe_mul_d(fe_mul(x, y)));
}

// the same for the twisted table: phi of the point, stored as (y' - x', y' + x', 2 d' x' y') (ed448_dev.h)
__global__ void gtab_tw_pack_kernel(uint32_t n, const uint8_t *xy, uint32_t *gtab)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe x = fe_from_bytes(xy + (uint64_t)i * 112), y = fe_from_bytes(xy + (uint64_t)i * 112 + 56);
    Fe ymx, ypx, td;
    pt_tw_niels_from_affine(ymx, ypx, td, x, y);
    uint32_t *e = gtab + (uint64_t)i * FB_ENTRY_DWORDS;
    store_fe(e, ymx);
    store_fe(e + 16, ypx);
    store_fe(e + 32, td);
}

// out = 4 * in mod r  (56-byte BE in/out)   — `bytes_to_scalar(..).mul_mod(&Scalar::from(4))`
__global__ void sc_mul4_kernel(uint64_t n, const uint8_t *in, uint8_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t a[14], r[14];
    sc_from_be(a, in + i * 56);
    sc_mul4_mod(r, a);
    sc_to_be(out + i * 56, r);
}

// k = kb `*` 4 (src/ecc/signable.rs:46) under the chosen reading of the crate's `*` (ed448_algo.h: sc_star4)
__global__ void sc_star4_kernel(uint64_t n, const uint8_t *in, uint8_t *out, int star)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t a[14], r[14];
    sc_from_be(a, in + i * 56);
    sc_star4(r, a, star);
    sc_to_be(out + i * 56, r);
}

// z = k - h*s   (src/ecc/signable.rs:54; ed448_algo.h: sc_sign_z)
__global__ void sc_sign_z_kernel(uint64_t n, const uint8_t *k_be, const uint8_t *h_be, const uint8_t *s_be, uint8_t *z_be,
                                 int star)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[14], h[14], s[14], z[14];
    sc_from_be(k, k_be + i * 56);
    sc_from_be(h, h_be + i * 56);
    sc_from_be(s, s_be + i * 56);
    sc_sign_z(z, k, h, s, star);
    sc_to_be(z_be + i * 56, z);
}

// ------------------------------------------------------------------ launchers
static inline dim3 grid64(size_t n) { return dim3((unsigned)((n + 63) / 64)); }

// SIMDs of the current device (4 per compute unit; 1024 on a whole MI355X).  Every batch-size threshold of the kernel-family
// choice below is a multiple of it -- the families differ in lanes per item, so what matters is items per SIMD -- as in the
// sponge launcher (sponge_launch.hip: device_simds); on a partitioned device (CPX: 128 SIMDs) or a smaller part the
// literals of r04 chose every family wrongly by the partition factor.  The CAPY_DEBUG knobs and the
// capy_ed448_set_* setters stay absolute item counts.
static size_t dev_simds()
{
    static std::atomic<unsigned> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1024;
    unsigned v = cached[dev].load();
    if (!v) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        v = 4u * (unsigned)cus;
        cached[dev].store(v);
    }
    return v;
}

// Two items per lane (one shared inversion) once the batch still gives every SIMD two waves at half the wave count:
// 2 waves x 4 SIMDs x CUs x 128 items = 262 144 on MI355X.  CAPY_DEBUG=ed448_pair=0/1 forces it off / on (A/B).
static size_t pair_min_items()
{
    static const double e = debug_knob("ed448_pair", -1);
    if (e == 0) return (size_t)-1;
    if (e == 1) return (size_t)128;
    return dev_simds() * 2 * 128;
}

// Constant-address table lookups (capy_ed448_set_hardened / capy_call_options::hardened): table reads that do not depend
// on the scalar.  Variable base: vb_scalarmul_ct reads every row of the (lane-interleaved) per-item table per window.
// Fixed base: the 12-bit table cannot be read in full per window (2049 entries), so the hardened kernels use tables with
// narrow windows.  CAPY_HARDEN_PROTOCOL (the default): the multiplications by SECRET scalars inside the protocol calls
// (KeyPair::new, sign, key_encrypt's ephemeral k, key_decrypt), as the reference's curve crate advertises fixed-time
// lookups; CAPY_HARDEN_ALL: also the raw capy_ed448_scalarmul / basemul calls, whose scalars the library cannot
// classify; CAPY_HARDEN_OFF: none.  Verification and the table builds work on public data and always take the indexed kernels.
static std::atomic<int> g_hardened{CAPY_HARDEN_PROTOCOL};
// reading of the curve crate's `Scalar * Scalar` at signable.rs:46 (ed448_algo.h: sc_star4); 0 = product mod r
static std::atomic<int> g_scalar_star{0};
static bool harden(bool secret)
{
    const int o = thread_opts().hardened;
    const int m = o >= 0 ? o : g_hardened.load();
    return m == CAPY_HARDEN_ALL || (secret && m == CAPY_HARDEN_PROTOCOL);
}
static int scalar_star_mode()
{
    const int o = thread_opts().scalar_star;
    return o >= 0 ? o : g_scalar_star.load();
}
// test hook (capy_debug_last_curve_kernel): which kernel family the calling thread's last variable-base / fixed-base
// launch took: 1 indexed, 2 constant-address; +16 for the one-item-per-wave kernels
static thread_local int t_last_vb_kernel = 0, t_last_fb_kernel = 0;

// Small batches: one item per WAVE (ed448_wave.h) instead of one per lane -- 7x lower latency for the variable-base
// and 3.6x for the fixed-base multiplication, worth it while the batch is too small to fill the chip's lanes: the
// crossover is at ~10 000 items for variable base / double-scalar and ~5 000 for fixed base (profiles/r02_ed448_wave.txt),
// hence a threshold of 8192 for those.  Fixed base, re-measured with the division-step inversion of r03 (lane kernel
// 0.46 -> 0.30 ms at one wave per SIMD, wave kernel 0.147 -> 0.119 ms; profiles/r03_ed448_gcd_inversion.txt): the
// crossover is ~3000 items indexed and ~6500 with constant-address lookups, hence 5/16 and 3/4 of the threshold; with
// the constant-address lookups on the matrix cores (ed448_fb7.h: 0.52 ms at one wave per SIMD) ~3800, hence 7/16.
// capy_ed448_set_wave_max() / CAPY_DEBUG=ed448_wave_max=N override the threshold (0 = never).
static std::atomic<long> g_wave_max{-1};
static size_t wave_max_items()
{
    const long forced = g_wave_max.load();
    if (forced >= 0) return (size_t)forced;
    static const long env = (long)debug_knob("ed448_wave_max", -1);
    return env >= 0 ? (size_t)env : (size_t)CAPY_ED448_WAVE_MAX_PER_SIMD * dev_simds();
}

// Four lanes per item (ed448_quad.h) for quad_min < n <= quad_max public-scalar multiplications: below, a wave per item is
// faster still; above, every SIMD holds more than two quad waves and the lane-per-item kernels' throughput wins.
// CAPY_DEBUG=ed448_quad_min=A,ed448_quad_max=B override (max = 0: never).
static std::atomic<long> g_quad_min{-1}, g_quad_max{-1};  // capy_ed448_set_quad_range; negative: the defaults
static size_t quad_min_items()
{
    const long f = g_quad_min.load();
    if (f >= 0) return (size_t)f;
    static const long v = (long)debug_knob("ed448_quad_min", -1);
    return v >= 0 ? (size_t)v : 4 * dev_simds();  // 4096
}
static size_t quad_max_items()
{
    const long f = g_quad_max.load();
    if (f >= 0) return (size_t)f;
    static const long v = (long)debug_knob("ed448_quad_max", -1);
    return v >= 0 ? (size_t)v : 32 * dev_simds();  // 32 768
}

// Two lanes per item (ed448_duo.h) for duo_min < n <= duo_max public-scalar multiplications (checked before the quad range):
// one wave of 32 items per SIMD at 32 768 items, where the quad kernels need two.  CAPY_DEBUG=ed448_duo_min=A,ed448_duo_max=B.
static std::atomic<long> g_duo_min{-1}, g_duo_max{-1};  // capy_ed448_set_duo_range; negative: the defaults
static size_t duo_min_items()
{
    const long f = g_duo_min.load();
    if (f >= 0) return (size_t)f;
    static const long v = (long)debug_knob("ed448_duo_min", -1);
    return v >= 0 ? (size_t)v : 16 * dev_simds();  // 16 384
}
static size_t duo_max_items()
{
    const long f = g_duo_max.load();
    if (f >= 0) return (size_t)f;
    static const long v = (long)debug_knob("ed448_duo_max", -1);
    return v >= 0 ? (size_t)v : 32 * dev_simds();  // 32 768
}
// the constant-address quad kernel (secret scalars): quad_min < n <= this.  One round of waves up to 16 384 items (LDS: four
// waves per compute unit); beyond, a second round -- still ahead of the one-item-per-lane hardened kernel up to 32 768.
static size_t quad_ct_max_items()
{
    if (g_quad_max.load() >= 0) return (size_t)g_quad_max.load();  // capy_ed448_set_quad_range moves both
    static const long v = (long)debug_knob("ed448_quad_ct_max", -1);
    return v >= 0 ? (size_t)v : 32 * dev_simds();  // 32 768
}
static bool duo_range(size_t n) { return n > duo_min_items() && n <= duo_max_items(); }
static bool quad_range(size_t n) { return !duo_range(n) && n > quad_min_items() && n <= quad_max_items(); }

// Wave quantisation of the one-item-per-lane kernels: a batch of q x 65 536 + x items puts a further wave on x / 64 SIMDs, and
// the launch takes a whole further chain (2.3-3.3 ms) however small x is -- 81 920 items took 5.03 ms where 65 536 take 2.88
// (profiles/r04_ed448_remainder.txt).  A remainder of up to 32 768 items is therefore peeled off into a launch of its own,
// which takes the kernel family of ITS size (one item per wave, four or two lanes per item: 0.4-1.8 ms), on the same stream.
// CAPY_DEBUG=ed448_peel=0 switches it off.
// batches of at most one wave per SIMD in the one-item-per-lane form take the *_1w kernels
static size_t one_wave_items() { return 64 * dev_simds(); }  // 65 536
static size_t peel_remainder(size_t n)
{
    static const bool on = debug_knob("ed448_peel", 1) != 0;
    const size_t quantum = 64 * dev_simds();  // SIMDs x 64 lanes: 65 536
    if (!on || n <= quantum) return 0;
    const size_t x = n % quantum;
    return x <= quantum / 2 ? x : 0;
}

// secret: the scalars are key material (see harden())
static int vb_launch(size_t n, const uint8_t *scalars, uint64_t scalar_stride, const uint8_t *points,
                     uint64_t point_stride, uint8_t *out, hipStream_t s, bool secret)
{
    if (!n) return CAPY_OK;
    if (const size_t x = peel_remainder(n)) {
        // the remainder FIRST: its waves all start on an empty chip and run equally long, so the big launch behind it starts
        // on an empty chip too.  Behind the big launch (whose waves end staggered) the remainder's waves doubled up on the
        // SIMDs that happened to be free: 32 768 items took 3.7 ms there instead of 1.8
        const int rc = vb_launch(x, scalars + (n - x) * scalar_stride, scalar_stride, points + (n - x) * point_stride, point_stride,
                                 out + (n - x) * 112, s, secret);
        if (rc) return rc;
        return vb_launch(n - x, scalars, scalar_stride, points, point_stride, out, s, secret);
    }
    const bool ct = harden(secret);
    const bool quad_ct = ct && n > quad_min_items() && n <= quad_ct_max_items();
    const bool wave_family = n <= wave_max_items() && !quad_ct && (ct || !(duo_range(n) || quad_range(n)));
    t_last_vb_kernel = (ct ? 2 : 1) + (wave_family ? 16 : 0);
    if (wave_family) {
        if (ct)
            hipLaunchKernelGGL(wave::vb_wave_kernel<true>, dim3((unsigned)n), dim3(64), 0, s, (uint64_t)n, scalars, scalar_stride,
                               points, point_stride, out);
        else
            hipLaunchKernelGGL(wave::vb_wave_kernel<false>, dim3((unsigned)n), dim3(64), 0, s, (uint64_t)n, scalars, scalar_stride,
                               points, point_stride, out);
        CAPY_HIP(hipGetLastError());
        return CAPY_OK;
    }
    // two lanes per item, constant-address lookups, the table half in registers and half in LDS: one round of waves where the
    // quad form needs two (16 S < n <= 32 S).  CAPY_DEBUG=ed448_duo_ct=0 switches it off (A/B).
    static const bool duo_ct_on = debug_knob("ed448_duo_ct", 1) != 0;
    if (quad_ct && duo_ct_on && n > 16 * dev_simds() && n <= 32 * dev_simds() && g_quad_max.load() < 0) {
        t_last_vb_kernel = 2 + 64;
        hipLaunchKernelGGL(vb_duo_ct_kernel, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, s, (uint64_t)n, scalars, scalar_stride, points,
                           point_stride, out);
        CAPY_HIP(hipGetLastError());
        return CAPY_OK;
    }
    // four lanes per item, constant-address lookups in LDS: r04, profiles/r04_ed448_quad_ct.txt
    if (quad_ct) {
        t_last_vb_kernel = 2 + 32;
        hipLaunchKernelGGL(vb_quad_ct_kernel, dim3((unsigned)((n + 15) / 16)), dim3(64), 0, s, (uint64_t)n, scalars, scalar_stride, points,
                           point_stride, out);
        CAPY_HIP(hipGetLastError());
        return CAPY_OK;
    }
    // two lanes per item (indexed lookups only): r04, profiles/r04_ed448_duo.txt
    if (!ct && duo_range(n)) {
        t_last_vb_kernel = 1 + 64;
        const size_t slots = (n + 31) / 32 * 32;
        CAPY_WS(dtab, uint32_t *, s, WS_TABLE, slots * VB_TABLE_DWORDS * 4);
        hipLaunchKernelGGL(vb_duo_kernel, dim3((unsigned)(slots / 32)), dim3(64), 0, s, (uint64_t)n, scalars, scalar_stride, points,
                           point_stride, out, dtab);
        CAPY_HIP(hipGetLastError());
        return CAPY_OK;
    }
    // four lanes per item between the two families (indexed lookups only): r04, profiles/r04_ed448_quad.txt
    if (!ct && quad_range(n)) {
        t_last_vb_kernel = 1 + 32;
        const size_t slots = (n + 15) / 16 * 16;
        CAPY_WS(qtab, uint32_t *, s, WS_TABLE, slots * VB_TABLE_DWORDS * 4);
        hipLaunchKernelGGL(vb_quad_kernel, dim3((unsigned)(slots / 16)), dim3(64), 0, s, (uint64_t)n, scalars, scalar_stride, points,
                           point_stride, out, qtab);
        CAPY_HIP(hipGetLastError());
        return CAPY_OK;
    }
    // whole waves: the constant-address table of a wave is interleaved across its 64 lanes
    CAPY_WS(tab, uint32_t *, s, WS_TABLE, (n + 63) / 64 * 64 * VB_TABLE_DWORDS * 4);
    if (ct) {
        // one item per lane at every size: two items per lane with a shared inversion (vb2_kernel<true>) took 20.7 ms for
        // 2^18 items against 12.9 ms here (profiles/r03_ed448_hardened.txt) -- at two waves per SIMD the 17-row scans
        // of the wave-interleaved table have nothing to hide behind

        if (n <= one_wave_items())
            hipLaunchKernelGGL(vb_ct_kernel_1w, grid64(n), dim3(64), 0, s, (uint64_t)n, scalars, scalar_stride, points, point_stride,
                               out, tab);
        else
            hipLaunchKernelGGL(vb_ct_kernel, grid64(n), dim3(64), 0, s, (uint64_t)n, scalars, scalar_stride, points, point_stride,
                               out, tab);
    } else if (n >= pair_min_items()) {
        return vb2_launch(n, scalars, scalar_stride, points, point_stride, out, tab, s);
    } else {
        if (n <= one_wave_items())
            hipLaunchKernelGGL(vb_kernel_1w, grid64(n), dim3(64), 0, s, (uint64_t)n, scalars, scalar_stride, points, point_stride,
                               out, tab);
        else
            hipLaunchKernelGGL(vb_kernel, grid64(n), dim3(64), 0, s, (uint64_t)n, scalars, scalar_stride, points, point_stride,
                               out, tab);
    }
    CAPY_HIP(hipGetLastError());
    return CAPY_OK;
}

// The shared fixed-base table: row i, entry j = (j * 2^(WBITS i) mod r) * G, affine cached.  Built once per
// device by running the variable-base kernel on G itself, then packed into limbs.
static std::mutex g_gtab_mu;

static const uint8_t G_XY[112] = {
    0x5e, 0xc0, 0x0c, 0xc7, 0x2b, 0xa8, 0x26, 0x26, 0x8e, 0x93, 0x00, 0x8b, 0xe1, 0x80, 0x3b, 0x43, 0x11, 0x65, 0xb6,
    0x2a, 0xf7, 0x1a, 0xae, 0x12, 0x64, 0xa4, 0xd3, 0xa3, 0x24, 0xe3, 0x6d, 0xea, 0x67, 0x17, 0x0f, 0x47, 0x70, 0x65,
    0x14, 0x9e, 0xda, 0x36, 0xbf, 0x22, 0xa6, 0x15, 0x1d, 0x22, 0xed, 0x0d, 0xed, 0x6b, 0xc6, 0x70, 0x19, 0x4f,
    0x14, 0xfa, 0x30, 0xf2, 0x5b, 0x79, 0x08, 0x98, 0xad, 0xc8, 0xd7, 0x4e, 0x2c, 0x13, 0xbd, 0xfd, 0xc4, 0x39, 0x7c,
    0xe6, 0x1c, 0xff, 0xd3, 0x3a, 0xd7, 0xc2, 0xa0, 0x05, 0x1e, 0x9c, 0x78, 0x87, 0x40, 0x98, 0xa3, 0x6c, 0x73, 0x73,
    0xea, 0x4b, 0x62, 0xc7, 0xc9, 0x56, 0x37, 0x20, 0x76, 0x88, 0x24, 0xbc, 0xb6, 0x6e, 0x71, 0x46, 0x3f, 0x69};

// Generators.  Handle 0 is the process generator: by default the RFC 8032 base point above -- what
// `ExtendedPoint::generator()` of the absent curve crate is ASSUMED to be (assumption (i)).
// capy_ed448_set_generator replaces it, so that a maintainer who finds the crate's generator to be a different point of
// the curve aligns the library in one call instead of a rebuild; capy_ed448_generator_create registers further points
// that single calls select through capy_call_options::generator (r04), so that two host threads can work with different
// generators at the same time.  Every context owns its fixed-base tables, built lazily per device.
struct GenCtx {
    uint8_t xy[112];
    uint32_t *gtab[64] = {nullptr};      // indexed 12-bit table on E
    uint32_t *gtab_ct[64] = {nullptr};   // the hardened table (FBCT_WBITS-bit windows), built on first hardened use
    uint32_t *gtab_tw[64] = {nullptr};   // the indexed 12-bit table on the twisted curve (lane-per-item fixed base)
    uint8_t *gtab7[64] = {nullptr};      // the hardened table as matrix-core operands (ed448_fb7.h)
    bool gtab7_twisted[64] = {false};
};
static std::vector<GenCtx *> g_gens;  // guarded by g_gtab_mu; contexts are never freed (handles stay valid)
static GenCtx *gen_ctx(int handle)    // caller holds g_gtab_mu; nullptr for an unknown handle
{
    if (g_gens.empty()) {
        GenCtx *c = new GenCtx();
        memcpy(c->xy, G_XY, 112);
        g_gens.push_back(c);
    }
    return (handle >= 0 && (size_t)handle < g_gens.size()) ? g_gens[handle] : nullptr;
}
static GenCtx *current_gen() { return gen_ctx(thread_opts().generator); }  // caller holds g_gtab_mu
static const uint8_t *current_generator()  // caller holds g_gtab_mu
{
    GenCtx *c = current_gen();
    return c ? c->xy : G_XY;
}


// rows x entries table of j * 2^(wbits row) * G in affine cached form, built by the variable-base kernel on G itself
// twisted: the multiples of G4 = [1/4 mod r] G mapped to the 4-isogenous twisted curve (ed448_dev.h); only for a
// generator of order r (generator_has_order_r)
static int build_gtab(const uint8_t *gen_xy, int rows, int entries, int wbits, uint32_t **slot, bool twisted = false)
{
    const size_t n = (size_t)rows * entries;
    std::vector<uint8_t> sc(n * 56), pts(n * 112);
    // 2^(wbits row) mod r, times 1/4 mod r for the twisted table
    static const uint32_t INV4[14] = {0xaad6113du, 0x48de30a4u, 0xa37163d5u, 0x085b309cu, 0x6bb58da4u, 0x7113b6d2u, 0xdf3288fau,
                                      0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x0fffffffu};
    uint32_t pw[14] = {1};
    if (twisted) memcpy(pw, INV4, sizeof(pw));
    for (int row = 0; row < rows; row++) {
        uint32_t acc[14] = {0};
        for (int j = 0; j < entries; j++) {
            sc_to_be(sc.data() + (size_t)(row * entries + j) * 56, acc);
            memcpy(pts.data() + (size_t)(row * entries + j) * 112, gen_xy, 112);
            sc_add_mod(acc, pw);
        }
        for (int d = 0; d < wbits; d++) sc_dbl_mod(pw);
    }
    DevBuf dsc, dpts, dout, dtab;
    CAPY_HIP(dsc.alloc(sc.size()));
    CAPY_HIP(dpts.alloc(pts.size()));
    CAPY_HIP(dout.alloc(n * 112));
    CAPY_HIP(dtab.alloc(n * VB_TABLE_DWORDS * 4));
    CAPY_HIP(dsc.put(sc.data(), sc.size()));
    CAPY_HIP(dpts.put(pts.data(), pts.size()));
    uint32_t *gt = nullptr;
    CAPY_HIP(hipMalloc((void **)&gt, n * FB_ENTRY_DWORDS * 4));
    hipLaunchKernelGGL(vb_kernel, grid64(n), dim3(64), 0, nullptr, (uint64_t)n, dsc.as<uint8_t>(), (uint64_t)56,
                       dpts.as<uint8_t>(), (uint64_t)112, dout.as<uint8_t>(), dtab.as<uint32_t>());
    if (twisted)
        hipLaunchKernelGGL(gtab_tw_pack_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, (uint32_t)n,
                           dout.as<uint8_t>(), gt);
    else
        hipLaunchKernelGGL(gtab_pack_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, (uint32_t)n,
                           dout.as<uint8_t>(), gt);
    CAPY_HIP(hipGetLastError());
    CAPY_HIP(hipDeviceSynchronize());
    *slot = gt;
    return CAPY_OK;
}

// the hardened table in the form the matrix cores read (ed448_fb7.h), built from a linear 7-bit table on first use
#ifndef CAPY_ED448_FBCT_MFMA
#define CAPY_ED448_FBCT_MFMA 1  // 0: the VALU scan of fb_ct_kernel / fb2_kernel<true> for every lane-per-item hardened fixed base (A/B)
#endif
// The lane-per-item fixed-base kernels accumulate on the 4-isogenous twisted curve (7M instead of 8M per addition,
// ed448_dev.h) when the generator in use has the prime order r -- the RFC 8032 base point does; a configured generator
// that does not keeps tables and additions on E.  CAPY_ED448_FB_TWISTED=0 (compile time) keeps everything on E (A/B).
#ifndef CAPY_ED448_FB_TWISTED
#define CAPY_ED448_FB_TWISTED 1
#endif
// Every generator has passed check_generator (prime order r), so the twisted-curve tables are always available; the
// lane-per-item fixed-base kernels are instantiated for that form only (r04: the E-only fallback for generators of other
// orders could not be reached any more and was removed with its four kernel instances).
constexpr bool FB_TW = CAPY_ED448_FB_TWISTED != 0;

static int ensure_gtab7(const uint8_t **out, bool *twisted)
{
    int dev = 0;
    CAPY_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(CAPY_ERR_ARG, "device index out of range");
    std::lock_guard<std::mutex> lk(g_gtab_mu);
    GenCtx *g = current_gen();
    if (!g) return fail(CAPY_ERR_ARG, "unknown generator handle");
    if (!g->gtab7[dev]) {
        const bool tw = FB_TW;
        g->gtab7_twisted[dev] = tw;
        uint32_t *lin = nullptr;
        const int rc = build_gtab(g->xy, FB7_ROWS, FB7_ENTRIES, FB7_WBITS, &lin, tw);
        if (rc) return rc;
        uint8_t *gt7 = nullptr;
        if (hipMalloc((void **)&gt7, FB7_TABLE_BYTES) != hipSuccess) {
            (void)hipFree(lin);
            return fail(CAPY_ERR_HIP, "hipMalloc of the fixed-base table failed");
        }
        const unsigned words = (unsigned)(FB7_TABLE_BYTES / 4);
        hipLaunchKernelGGL(gtab7_pack_kernel, dim3((words + 255) / 256), dim3(256), 0, nullptr, lin, reinterpret_cast<uint32_t *>(gt7));
        const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
        (void)hipFree(lin);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            (void)hipFree(gt7);
            return fail(CAPY_ERR_HIP, "building the fixed-base table failed");
        }
        g->gtab7[dev] = gt7;
    }
    *out = g->gtab7[dev];
    *twisted = g->gtab7_twisted[dev];
    return CAPY_OK;
}

// twisted (in/out, indexed table only): ask for the twisted table; comes back false when the generator's order is not r
static int ensure_gtab(const uint32_t **out, bool hardened_table = false, bool *twisted = nullptr)
{
    int dev = 0;
    CAPY_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(CAPY_ERR_ARG, "device index out of range");
    std::lock_guard<std::mutex> lk(g_gtab_mu);
    GenCtx *g = current_gen();
    if (!g) return fail(CAPY_ERR_ARG, "unknown generator handle");
    const bool tw = twisted && *twisted && !hardened_table && FB_TW;
    if (twisted) *twisted = tw;
    uint32_t **slot = hardened_table ? &g->gtab_ct[dev] : (tw ? &g->gtab_tw[dev] : &g->gtab[dev]);
    if (!*slot) {
        const int rc = hardened_table ? build_gtab(g->xy, FBCT_ROWS, FBCT_ENTRIES, FBCT_WBITS, slot)
                                      : build_gtab(g->xy, FB_ROWS, FB_TAB_ENTRIES, FB_WBITS, slot, tw);
        if (rc) return rc;
    }
    *out = *slot;
    return CAPY_OK;
}

static int fb_launch(size_t n, const uint8_t *scalars, uint8_t *out, hipStream_t s, bool secret)
{
    if (!n) return CAPY_OK;
    const bool ct = harden(secret);
    const bool small = n <= (ct ? (CAPY_ED448_FBCT_MFMA ? wave_max_items() * 7 / 16 : wave_max_items() * 3 / 4) : wave_max_items() * 5 / 16);
    t_last_fb_kernel = (ct ? 2 : 1) + (small ? 16 : 0);
    if (ct && !small && CAPY_ED448_FBCT_MFMA) {
        // every byte of the window's table row is read per window by every wave and the wanted entry is picked by a
        // one-hot matrix product on the matrix cores: no address depends on the scalar (ed448_fb7.h)
        const uint8_t *gt7 = nullptr;
        bool tw = false;
        const int rc7 = ensure_gtab7(&gt7, &tw);
        if (rc7) return rc7;
        if (tw != FB_TW) return fail(CAPY_ERR_HIP, "internal: fixed-base table form");
        if (n >= pair_min_items()) {
            const size_t blocks = (n + 127) / 128;
            CAPY_WS(park, uint32_t *, s, WS_TABLE, blocks * 64 * 48 * 4);
            hipLaunchKernelGGL(fb_ct7_pair_kernel<FB_TW>, dim3((unsigned)blocks), dim3(64), 0, s, (uint64_t)n, scalars, out, gt7, park);
            // the parked PROJECTIVE results of secret multiples must not outlive the call (a projective representation
            // of [k]G says more about k than the affine point does)
            CAPY_HIP(hipMemsetAsync(park, 0, blocks * 64 * 48 * 4, s));
        } else {
            hipLaunchKernelGGL(fb_ct7_kernel<FB_TW>, grid64(n), dim3(64), 0, s, (uint64_t)n, scalars, out, gt7);
        }
        CAPY_HIP(hipGetLastError());
        return CAPY_OK;
    }
    const uint32_t *gt = nullptr;
    // ct: every entry of the window's row of the 5-bit table is read per window: no address depends on the scalar.
    // The indexed lane-per-item kernels take the twisted table when the generator allows it
    bool tw = !ct && !small;
    int rc = ensure_gtab(&gt, ct, &tw);
    if (rc) return rc;
    const dim3 pair_grid((unsigned)((n + 127) / 128));
    if (small) {
        if (ct)
            hipLaunchKernelGGL(wave::fb_wave_kernel<true>, dim3((unsigned)n), dim3(64), 0, s, (uint64_t)n, scalars, out, gt);
        else
            hipLaunchKernelGGL(wave::fb_wave_kernel<false>, dim3((unsigned)n), dim3(64), 0, s, (uint64_t)n, scalars, out, gt);
    } else if (n >= pair_min_items()) {
        if (ct)
            hipLaunchKernelGGL((fb2_kernel<true, false>), pair_grid, dim3(64), 0, s, (uint64_t)n, scalars, out, gt);
        else
            hipLaunchKernelGGL((fb2_kernel<false, FB_TW>), pair_grid, dim3(64), 0, s, (uint64_t)n, scalars, out, gt);
    } else if (ct) {
        hipLaunchKernelGGL(fb_ct_kernel, grid64(n), dim3(64), 0, s, (uint64_t)n, scalars, out, gt);
    } else {
        hipLaunchKernelGGL(fb_kernel<FB_TW>, grid64(n), dim3(64), 0, s, (uint64_t)n, scalars, out, gt);
    }
    CAPY_HIP(hipGetLastError());
    return CAPY_OK;
}

static int dsm_launch(size_t n, const uint8_t *a, const uint8_t *b, const uint8_t *points, uint8_t *out, hipStream_t s)
{
    if (!n) return CAPY_OK;
    if (const size_t x = peel_remainder(n)) {  // see peel_remainder() and vb_launch(): the remainder first
        const int rc = dsm_launch(x, a + (n - x) * 56, b + (n - x) * 56, points + (n - x) * 112, out + (n - x) * 112, s);
        if (rc) return rc;
        return dsm_launch(n - x, a, b, points, out, s);
    }
    const uint32_t *gt = nullptr;
    int rc = ensure_gtab(&gt);
    if (rc) return rc;
    // the test hook reports the family of the double multiplication's variable-base part like a variable-base launch
    t_last_vb_kernel = duo_range(n) ? 1 + 64 : (quad_range(n) ? 1 + 32 : (n <= wave_max_items() ? 1 + 16 : 1));
    if (duo_range(n)) {  // two lanes per item (ed448_duo.h)
        const size_t slots = (n + 31) / 32 * 32;
        CAPY_WS(dtab, uint32_t *, s, WS_TABLE, slots * VB_TABLE_DWORDS * 4);
        hipLaunchKernelGGL(dsm_duo_kernel, dim3((unsigned)(slots / 32)), dim3(64), 0, s, (uint64_t)n, a, b, points, out, dtab, gt);
        CAPY_HIP(hipGetLastError());
        return CAPY_OK;
    }
    if (quad_range(n)) {  // four lanes per item (ed448_quad.h)
        const size_t slots = (n + 15) / 16 * 16;
        CAPY_WS(qtab, uint32_t *, s, WS_TABLE, slots * VB_TABLE_DWORDS * 4);
        hipLaunchKernelGGL(dsm_quad_kernel, dim3((unsigned)(slots / 16)), dim3(64), 0, s, (uint64_t)n, a, b, points, out, qtab, gt);
        CAPY_HIP(hipGetLastError());
        return CAPY_OK;
    }
    if (n <= wave_max_items()) {
        hipLaunchKernelGGL(wave::dsm_wave_kernel, dim3((unsigned)n), dim3(64), 0, s, (uint64_t)n, a, b, points, out, gt);
        CAPY_HIP(hipGetLastError());
        return CAPY_OK;
    }
    CAPY_WS(tab, uint32_t *, s, WS_TABLE, n * VB_TABLE_DWORDS * 4);
    if (n <= one_wave_items())
        hipLaunchKernelGGL(dsm_kernel_1w, grid64(n), dim3(64), 0, s, (uint64_t)n, a, b, points, out, tab, gt);
    else
        hipLaunchKernelGGL(dsm_kernel, grid64(n), dim3(64), 0, s, (uint64_t)n, a, b, points, out, tab, gt);
    CAPY_HIP(hipGetLastError());
    return CAPY_OK;
}

static int sc_mul4_launch(size_t n, const uint8_t *in, uint8_t *out, hipStream_t s)
{
    if (!n) return CAPY_OK;
    hipLaunchKernelGGL(sc_mul4_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, (uint64_t)n, in, out);
    CAPY_HIP(hipGetLastError());
    return CAPY_OK;
}

// s_i = 4 * KMAC(pw_i, "", 448, "SK", d) mod r   (keypair.rs:42-43, signable.rs:41-43, ecc/encryptable.rs:76-77)
static int derive_s_dev(int d, size_t n, const KeyView &pw, uint8_t *s_be, hipStream_t st)
{
    MsgView none;
    int rc = kmac_launch(d, n, pw, none, true, (const uint8_t *)"SK", 2, 0, s_be, 56, 56, nullptr, st);
    if (rc) return rc;
    return sc_mul4_launch(n, s_be, s_be, st);
}

// ------------------------------------------------------------------ protocol glue on device buffers
// Signable::sign, src/ecc/signable.rs:40-57
static int sign_dev(int d, size_t n, const KeyView &pw, const MsgView &m, uint8_t *h, uint8_t *z, hipStream_t st)
{
    WsScrubGuard scrub(st);  // on every return path
    CAPY_WS(s_be, uint8_t *, st, WS_A, n * 56);
    scrub.add(WS_A, n * 56);  // the secret scalar s
    CAPY_WS(k_be, uint8_t *, st, WS_B, n * 56);
    scrub.add(WS_B, n * 56);  // the nonce k
    CAPY_WS(U, uint8_t *, st, WS_C, n * 112);
    int rc = derive_s_dev(d, n, pw, s_be, st);
    if (rc) return rc;
    // k = 4 * KMAC(s_bytes, msg, 448, "N")  (`*` taken as arithmetic mod r)
    rc = kmac_launch(d, n, fixed_keys(s_be, 56, 56), m, true, (const uint8_t *)"N", 1, 0, k_be, 56, 56, nullptr, st);
    if (rc) return rc;
    const int star = scalar_star_mode();
    hipLaunchKernelGGL(sc_star4_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (uint64_t)n, k_be, k_be, star);
    CAPY_HIP(hipGetLastError());
    rc = fb_launch(n, k_be, U, st, true);  // U = k*G, affine; k is the secret nonce
    if (rc) return rc;
    // h = KMAC(U.x bytes, msg, 448, "T")
    rc = kmac_launch(d, n, fixed_keys(U, 56, 112), m, true, (const uint8_t *)"T", 1, 0, h, 56, 56, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(sc_sign_z_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (uint64_t)n, k_be, h, s_be, z, star);
    CAPY_HIP(hipGetLastError());
    return CAPY_OK;
}

// Signable::verify, src/ecc/signable.rs:72-86
// Host-buffer entry points whose curve work does not depend on the messages (verify: U = z*G + h*V; key_encrypt:
// W = k*V, Z = k*G; key_decrypt: W = s*Z) launch that work first, on a non-blocking side stream, and only then copy the
// messages to the device: the blocking host-to-device copy runs while the scalar multiplications do (2^16 x 1 KiB
// verify: 64 MiB over PCIe under a 3 ms kernel).  LateMsgs carries the batch that is still on the host.
struct LateMsgs {
    PackedBatch *b;
    size_t n;
    const uint8_t *msgs;
    const uint64_t *offsets;
    int upload(MsgView &out)
    {
        const int rc = b->upload(n, msgs, offsets);
        if (rc) return rc;
        out = view_of(*b);
        return CAPY_OK;
    }
};
struct SideStreams {  // one per host thread and device, destroyed with the thread
    hipStream_t s[64] = {nullptr};
    ~SideStreams()
    {
        for (hipStream_t x : s)
            if (x) (void)hipStreamDestroy(x);
    }
};
static hipStream_t side_stream()
{
    thread_local SideStreams holder;
    hipStream_t *streams = holder.s;
    // CAPY_DEBUG=host_overlap=0: everything on the default stream (A/B: copy and kernels serialise)
    static const bool off = debug_knob("host_overlap", 1) == 0;
    if (off) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!streams[dev] && hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        streams[dev] = nullptr;  // fall back to the default stream: correct, no overlap
    }
    return streams[dev];
}

static int verify_dev(int d, size_t n, const uint8_t *pubs, const MsgView &m_in, const uint8_t *h, const uint8_t *z,
                      int32_t *status, hipStream_t st, LateMsgs *late = nullptr)
{
    CAPY_WS(U, uint8_t *, st, WS_C, n * 112);
    CAPY_WS(h2, uint8_t *, st, WS_A, n * 56);
    int rc = dsm_launch(n, z, h, pubs, U, st);  // U = z*G + h*V
    if (rc) return rc;
    MsgView m = m_in;
    if (late && (rc = late->upload(m))) return rc;
    rc = kmac_launch(d, n, fixed_keys(U, 56, 112), m, true, (const uint8_t *)"T", 1, 0, h2, 56, 56, nullptr, st);
    if (rc) return rc;
    tag_compare_launch(h, 56, h2, 56, 56, status, n, st);
    CAPY_HIP(hipGetLastError());
    return CAPY_OK;
}

// the symmetric half shared by key_encrypt / key_decrypt: (ke || ka) = KMAC(W.x, "", 896, "PK")
static int pk_keys_dev(int d, size_t n, const uint8_t *W, uint8_t *keka, hipStream_t st)
{
    MsgView none;
    return kmac_launch(d, n, fixed_keys(W, 56, 112), none, true, (const uint8_t *)"PK", 2, 0, keka, 112, 112, nullptr, st);
}
// KeyEncryptable::key_encrypt, src/ecc/encryptable.rs:34-50
static int key_encrypt_dev(int d, size_t n, const uint8_t *pubs, const uint8_t *k_rand, const MsgView &m_in, uint8_t *z_xy,
                           uint8_t *tags, hipStream_t st, LateMsgs *late = nullptr)
{
    WsScrubGuard scrub(st);  // on every return path
    CAPY_WS(k_be, uint8_t *, st, WS_B, n * 56);
    scrub.add(WS_B, n * 56);  // the ephemeral scalar k
    CAPY_WS(W, uint8_t *, st, WS_C, n * 112);
    scrub.add(WS_C, n * 112);  // the shared point W
    CAPY_WS(keka, uint8_t *, st, WS_D, n * 112);
    scrub.add(WS_D, n * 112);  // ke || ka
    int rc = sc_mul4_launch(n, k_rand, k_be, st);
    if (rc) return rc;
    rc = vb_launch(n, k_be, 56, pubs, 112, W, st, true);  // W = k*V; k is the ephemeral secret
    if (rc) return rc;
    rc = fb_launch(n, k_be, z_xy, st, true);  // Z = k*G
    if (rc) return rc;
    rc = pk_keys_dev(d, n, W, keka, st);
    if (rc) return rc;
    MsgView m = m_in;
    if (late && (rc = late->upload(m))) return rc;
    // t = kmac_xof(ka, m, 448, "PKA") over the plaintext (:43), then m ^= kmac_xof(ke, "", |m|, "PKE") (:45-46)
    return symmetric_crypt_dev(true, d, n, keka, 56, 112, m, tags, 56, "PKE", "PKA", nullptr, st);
}

// KeyEncryptable::key_decrypt, src/ecc/encryptable.rs:72-94
static int key_decrypt_dev(int d, size_t n, const KeyView &pw, const uint8_t *z_xy, const MsgView &m_in,
                           const uint8_t *tags, int32_t *status, hipStream_t st, LateMsgs *late = nullptr)
{
    WsScrubGuard scrub(st);  // on every return path
    CAPY_WS(s_be, uint8_t *, st, WS_A, n * 56);
    scrub.add(WS_A, n * 56);  // the secret scalar s
    CAPY_WS(W, uint8_t *, st, WS_C, n * 112);
    scrub.add(WS_C, n * 112);  // the shared point W
    CAPY_WS(keka, uint8_t *, st, WS_D, n * 112);
    scrub.add(WS_D, n * 112);  // ke || ka
    int rc = derive_s_dev(d, n, pw, s_be, st);
    if (rc) return rc;
    rc = vb_launch(n, s_be, 56, z_xy, 112, W, st, true);  // W = s*Z; s is the private scalar
    if (rc) return rc;
    rc = pk_keys_dev(d, n, W, keka, st);
    if (rc) return rc;
    MsgView m = m_in;
    if (late && (rc = late->upload(m))) return rc;
    // candidate plaintext, tag check, restore the ciphertext where the tag failed (:82-93)
    return symmetric_crypt_dev(false, d, n, keka, 56, 112, m, const_cast<uint8_t *>(tags), 56, "PKE", "PKA", status, st);
}

static int up(DevBuf &b, const void *src, size_t bytes)
{
    CAPY_HIP(b.alloc(bytes));
    CAPY_HIP(b.put(src, bytes));
    return CAPY_OK;
}
static int down(void *dst, const DevBuf &b, size_t bytes)
{
    CAPY_HIP(b.get(dst, bytes));
    return CAPY_OK;
}

}  // namespace capy

using namespace capy;

#define TRY(x)            \
    do {                  \
        int _rc = (x);    \
        if (_rc) return _rc; \
    } while (0)

extern "C" {

// ---------------------------------------------------------------- raw curve operations
int capy_ed448_scalarmul_batch_dev(size_t n, const uint8_t *scalars_be, const uint8_t *points_xy, uint8_t *out_xy,
                                   void *stream)
{
    if (n) CAPY_REQUIRE(scalars_be && points_xy && out_xy, "scalars / points / out");
    return vb_launch(n, scalars_be, 56, points_xy, 112, out_xy, (hipStream_t)stream, false);
}

int capy_ed448_scalarmul_batch(size_t n, const uint8_t *scalars_be, const uint8_t *points_xy, uint8_t *out_xy)
{
    if (!n) return CAPY_OK;
    if (!scalars_be || !points_xy || !out_xy) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, nullptr, capy_ed448_scalarmul_batch(count, scalars_be + first * 56, points_xy + first * 112, out_xy + first * 112));
    DevBuf s, p, o;
    TRY(up(s, scalars_be, n * 56));
    TRY(up(p, points_xy, n * 112));
    CAPY_HIP(o.alloc(n * 112));
    TRY(vb_launch(n, s.as<uint8_t>(), 56, p.as<uint8_t>(), 112, o.as<uint8_t>(), nullptr, false));
    return down(out_xy, o, n * 112);
}

int capy_ed448_basemul_batch_dev(size_t n, const uint8_t *scalars_be, uint8_t *out_xy, void *stream)
{
    if (n) CAPY_REQUIRE(scalars_be && out_xy, "scalars / out");
    return fb_launch(n, scalars_be, out_xy, (hipStream_t)stream, false);
}

int capy_ed448_basemul_batch(size_t n, const uint8_t *scalars_be, uint8_t *out_xy)
{
    if (!n) return CAPY_OK;
    if (!scalars_be || !out_xy) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, nullptr, capy_ed448_basemul_batch(count, scalars_be + first * 56, out_xy + first * 112));
    DevBuf s, o;
    TRY(up(s, scalars_be, n * 56));
    CAPY_HIP(o.alloc(n * 112));
    TRY(fb_launch(n, s.as<uint8_t>(), o.as<uint8_t>(), nullptr, false));
    return down(out_xy, o, n * 112);
}

int capy_ed448_add_batch_dev(size_t n, const uint8_t *p_xy, const uint8_t *q_xy, uint8_t *out_xy, void *stream)
{
    if (!n) return CAPY_OK;
    CAPY_REQUIRE(p_xy && q_xy && out_xy, "p / q / out");
    hipLaunchKernelGGL(add_kernel, grid64(n), dim3(64), 0, (hipStream_t)stream, (uint64_t)n, p_xy, q_xy, out_xy);
    CAPY_HIP(hipGetLastError());
    return CAPY_OK;
}

int capy_ed448_add_batch(size_t n, const uint8_t *p_xy, const uint8_t *q_xy, uint8_t *out_xy)
{
    if (!n) return CAPY_OK;
    if (!p_xy || !q_xy || !out_xy) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, nullptr, capy_ed448_add_batch(count, p_xy + first * 112, q_xy + first * 112, out_xy + first * 112));
    DevBuf p, q, o;
    TRY(up(p, p_xy, n * 112));
    TRY(up(q, q_xy, n * 112));
    CAPY_HIP(o.alloc(n * 112));
    TRY(capy_ed448_add_batch_dev(n, p.as<uint8_t>(), q.as<uint8_t>(), o.as<uint8_t>(), nullptr));
    return down(out_xy, o, n * 112);
}

int capy_ed448_set_wave_max(long max_items)
{
    g_wave_max.store(max_items < 0 ? -1 : max_items);
    return CAPY_OK;
}

int capy_ed448_set_duo_range(long min_items, long max_items)
{
    g_duo_min.store(min_items < 0 ? -1 : min_items);
    g_duo_max.store(max_items < 0 ? -1 : max_items);
    return CAPY_OK;
}
int capy_ed448_set_quad_range(long min_items, long max_items)
{
    g_quad_min.store(min_items < 0 ? -1 : min_items);
    g_quad_max.store(max_items < 0 ? -1 : max_items);
    return CAPY_OK;
}

int capy_ed448_set_hardened(int mode)
{
    // 2 and 3 were "raw calls only" / "everything" in the r03 library, where 1 had come to mean "protocol calls only" after
    // meaning "everything" in r02: the ambiguous values are refused rather than reinterpreted
    if (mode != CAPY_HARDEN_OFF && mode != CAPY_HARDEN_ALL && mode != CAPY_HARDEN_PROTOCOL)
        return fail(CAPY_ERR_ARG, "mode must be CAPY_HARDEN_OFF (0), CAPY_HARDEN_ALL (1) or CAPY_HARDEN_PROTOCOL (4, the default)");
    g_hardened.store(mode);
    return CAPY_OK;
}

int capy_debug_last_curve_kernel(int *variable_base, int *fixed_base)
{
    if (variable_base) *variable_base = t_last_vb_kernel;
    if (fixed_base) *fixed_base = t_last_fb_kernel;
    return CAPY_OK;
}

int capy_ed448_set_scalar_star(int mode)
{
    if (mode < 0 || mode > 2) return fail(CAPY_ERR_ARG, "mode must be 0 (product mod r), 1 or 2 (see capyhip.h)");
    g_scalar_star.store(mode);
    return CAPY_OK;
}

int capy_ed448_get_generator(uint8_t *xy)
{
    if (!xy) return fail(CAPY_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> lk(g_gtab_mu);
    memcpy(xy, current_generator(), 112);
    return CAPY_OK;
}

// [4]P = (0, 1): the identity and the points of order 2 and 4 -- useless (and dangerous) as a generator
static bool pt_order_divides_4(const uint8_t *xy)
{
    const Pt q = pt_dbl<true>(pt_dbl<true>(pt_from_affine_bytes(xy)));
    return fe_is_zero(q.X) && fe_is_zero(fe_sub(q.Y, q.Z));
}

// [r]P = (0, 1), computed on the host with the device code's variable-base algorithm (no GPU needed to refuse a point)
static bool pt_has_order_r(const uint8_t *xy)
{
    static const uint32_t R_WORDS[14] = {0xab5844f3u, 0x2378c292u, 0x8dc58f55u, 0x216cc272u, 0xaed63690u, 0xc44edb49u, 0x7cca23e9u,
                                         0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
    uint8_t r_be[56];
    sc_to_be(r_be, R_WORDS);
    std::vector<uint32_t> tab(VB_TABLE_DWORDS);
    const Pt q = vb_scalarmul(r_be, pt_from_affine_bytes(xy), tab.data());
    return fe_is_zero(q.X) && fe_is_zero(fe_sub(q.Y, q.Z));
}

static int check_generator(const uint8_t *xy)
{
    if (!pt_validate_bytes(xy)) return fail(CAPY_ERR_ARG, "generator is not a canonical point of the curve");
    if (pt_order_divides_4(xy)) return fail(CAPY_ERR_ARG, "generator has order 1, 2 or 4");
    // the fixed-base tables are built from scalars reduced mod r (and, on the twisted curve, from [1/4 mod r] G): both
    // need a generator of the prime order r, as ExtendedPoint::generator() of any Ed448 library is
    if (!pt_has_order_r(xy)) return fail(CAPY_ERR_ARG, "generator does not have the prime order r (cofactor component)");
    return CAPY_OK;
}

int capy_ed448_set_generator(const uint8_t *xy)
{
    if (xy) TRY(check_generator(xy));
    std::lock_guard<std::mutex> lk(g_gtab_mu);
    const uint8_t *want = xy ? xy : G_XY;
    GenCtx *old = gen_ctx(0);
    if (memcmp(old->xy, want, 112) == 0) return CAPY_OK;
    // A fresh context takes handle 0; its tables are built lazily per device.  The old context is RETIRED, not freed: a
    // concurrent call that took a table pointer before this one (ensure_gtab releases the mutex before it launches) may
    // still have kernels in flight on it.  A retired set of tables (15 MB per device used) stays allocated until the
    // process ends; the process generator is meant to be set once, at start-up.
    GenCtx *fresh = new GenCtx();
    memcpy(fresh->xy, want, 112);
    g_gens[0] = fresh;
    return CAPY_OK;
}

int capy_ed448_generator_create(const uint8_t *xy, int *handle)
{
    if (!xy || !handle) return fail(CAPY_ERR_ARG, "null argument");
    TRY(check_generator(xy));
    std::lock_guard<std::mutex> lk(g_gtab_mu);
    (void)gen_ctx(0);
    for (size_t i = 1; i < g_gens.size(); i++)
        if (memcmp(g_gens[i]->xy, xy, 112) == 0) {
            *handle = (int)i;
            return CAPY_OK;
        }
    if (g_gens.size() >= 64) return fail(CAPY_ERR_ARG, "too many generators (64)");
    GenCtx *c = new GenCtx();
    memcpy(c->xy, xy, 112);
    g_gens.push_back(c);
    *handle = (int)g_gens.size() - 1;
    return CAPY_OK;
}

int capy_ed448_validate_batch_dev(size_t n, const uint8_t *points_xy, int32_t *status, void *stream)
{
    if (!n) return CAPY_OK;
    if (!points_xy || !status) return fail(CAPY_ERR_ARG, "null argument");
    hipLaunchKernelGGL(validate_kernel, grid64(n), dim3(64), 0, (hipStream_t)stream, (uint64_t)n, points_xy, status);
    CAPY_HIP(hipGetLastError());
    return CAPY_OK;
}

int capy_ed448_validate_batch(size_t n, const uint8_t *points_xy, int32_t *status)
{
    if (!n) return CAPY_OK;
    if (!points_xy || !status) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, nullptr, capy_ed448_validate_batch(count, points_xy + first * 112, status + first));
    DevBuf p, st;
    TRY(up(p, points_xy, n * 112));
    CAPY_HIP(st.alloc(n * 4));
    TRY(capy_ed448_validate_batch_dev(n, p.as<uint8_t>(), st.as<int32_t>(), nullptr));
    return down(status, st, n * 4);
}

int capy_ed448_double_scalarmul_batch_dev(size_t n, const uint8_t *a_be, const uint8_t *b_be, const uint8_t *points_xy,
                                          uint8_t *out_xy, void *stream)
{
    if (n) CAPY_REQUIRE(a_be && b_be && points_xy && out_xy, "a / b / points / out");
    return dsm_launch(n, a_be, b_be, points_xy, out_xy, (hipStream_t)stream);
}

int capy_ed448_double_scalarmul_batch(size_t n, const uint8_t *a_be, const uint8_t *b_be, const uint8_t *points_xy,
                                      uint8_t *out_xy)
{
    if (!n) return CAPY_OK;
    if (!a_be || !b_be || !points_xy || !out_xy) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, nullptr, capy_ed448_double_scalarmul_batch(count, a_be + first * 56, b_be + first * 56, points_xy + first * 112,
                                                             out_xy + first * 112));
    DevBuf a, b, p, o;
    TRY(up(a, a_be, n * 56));
    TRY(up(b, b_be, n * 56));
    TRY(up(p, points_xy, n * 112));
    CAPY_HIP(o.alloc(n * 112));
    TRY(dsm_launch(n, a.as<uint8_t>(), b.as<uint8_t>(), p.as<uint8_t>(), o.as<uint8_t>(), nullptr));
    return down(out_xy, o, n * 112);
}

// ---------------------------------------------------------------- src/ecc protocols (device buffers, stream ordered)
static KeyView dev_keys(const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets)
{
    KeyView kv = fixed_keys(pws, pw_len, pw_len);
    kv.key_offsets = pw_offsets;
    return kv;
}

int capy_keypair_batch_dev(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets, uint8_t *pub_xy,
                           void *stream)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (!n) return CAPY_OK;
    CAPY_REQUIRE(pub_xy, "pub_xy");
    CAPY_REQUIRE(keys_ok(pws, pw_len, pw_offsets), "pws");
    hipStream_t st = (hipStream_t)stream;
    WsScrubGuard scrub(st);  // on every return path
    CAPY_WS(s_be, uint8_t *, st, WS_A, n * 56);
    scrub.add(WS_A, n * 56);  // the secret scalar s
    TRY(derive_s_dev(d, n, dev_keys(pws, pw_len, pw_offsets), s_be, st));
    TRY(fb_launch(n, s_be, pub_xy, st, true));  // the private scalar
    return CAPY_OK;
}

int capy_schnorr_sign_batch_dev(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                                const uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride,
                                uint8_t *h, uint8_t *z_be, void *stream)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (!n) return CAPY_OK;
    CAPY_REQUIRE(h && z_be, "h / z_be");
    CAPY_REQUIRE(keys_ok(pws, pw_len, pw_offsets), "pws");
    CAPY_REQUIRE(msgs_ok(msgs, offsets, uniform_len), "msgs");
    return sign_dev(d, n, dev_keys(pws, pw_len, pw_offsets), view_dev(msgs, offsets, uniform_len, msg_stride), h, z_be,
                    (hipStream_t)stream);
}

int capy_schnorr_verify_batch_dev(int d, size_t n, const uint8_t *pub_xy, const uint8_t *msgs, const uint64_t *offsets,
                                  uint64_t uniform_len, uint64_t msg_stride, const uint8_t *h, const uint8_t *z_be,
                                  int32_t *status, void *stream)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (!n) return CAPY_OK;
    CAPY_REQUIRE(pub_xy && h && z_be && status, "pub_xy / h / z_be / status");
    CAPY_REQUIRE(msgs_ok(msgs, offsets, uniform_len), "msgs");
    return verify_dev(d, n, pub_xy, view_dev(msgs, offsets, uniform_len, msg_stride), h, z_be, status,
                      (hipStream_t)stream);
}

int capy_key_encrypt_batch_dev(int d, size_t n, const uint8_t *pub_xy, const uint8_t *k_rand, uint8_t *msgs,
                               const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride, uint8_t *z_xy,
                               uint8_t *tags, void *stream)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (!n) return CAPY_OK;
    CAPY_REQUIRE(pub_xy && k_rand && z_xy && tags, "pub_xy / k_rand / z_xy / tags");
    CAPY_REQUIRE(msgs_ok(msgs, offsets, uniform_len), "msgs");
    return key_encrypt_dev(d, n, pub_xy, k_rand, view_dev(msgs, offsets, uniform_len, msg_stride), z_xy, tags,
                           (hipStream_t)stream);
}

int capy_key_decrypt_batch_dev(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                               const uint8_t *z_xy, uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len,
                               uint64_t msg_stride, const uint8_t *tags, int32_t *status, void *stream)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (!n) return CAPY_OK;
    CAPY_REQUIRE(z_xy && tags && status, "z_xy / tags / status");
    CAPY_REQUIRE(keys_ok(pws, pw_len, pw_offsets), "pws");
    CAPY_REQUIRE(msgs_ok(msgs, offsets, uniform_len), "msgs");
    return key_decrypt_dev(d, n, dev_keys(pws, pw_len, pw_offsets), z_xy, view_dev(msgs, offsets, uniform_len, msg_stride),
                           tags, status, (hipStream_t)stream);
}

// ---------------------------------------------------------------- src/ecc protocols (host buffers)
int capy_keypair_batch(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets, uint8_t *pub_xy)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (!n) return CAPY_OK;
    if (!pub_xy) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, pw_offsets, capy_keypair_batch(d, count, pw_offsets ? pws : pws + first * pw_len, pw_len,
                                                 pw_offsets ? pw_offsets + first : nullptr, pub_xy + first * 112));
    PackedKeys pw;
    TRY(pw.upload(n, pws, pw_len, pw_offsets));
    DevBuf s, o;
    s.secret = true;  // the secret scalars: zeroed before the buffer is freed
    CAPY_HIP(s.alloc(n * 56));
    CAPY_HIP(o.alloc(n * 112));
    TRY(derive_s_dev(d, n, pw.view, s.as<uint8_t>(), nullptr));
    TRY(fb_launch(n, s.as<uint8_t>(), o.as<uint8_t>(), nullptr, true));  // the private scalar
    return down(pub_xy, o, n * 112);
}

int capy_schnorr_sign_batch(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                            const uint8_t *msgs, const uint64_t *offsets, uint8_t *h, uint8_t *z_be)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (!n) return CAPY_OK;
    if (!offsets || !h || !z_be) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, offsets, capy_schnorr_sign_batch(d, count, pw_offsets ? pws : pws + first * pw_len, pw_len,
                                                   pw_offsets ? pw_offsets + first : nullptr, msgs, offsets + first,
                                                   h + first * 56, z_be + first * 56));
    PackedBatch b;
    TRY(b.upload(n, msgs, offsets));
    PackedKeys pw;
    TRY(pw.upload(n, pws, pw_len, pw_offsets));
    DevBuf dh, dz;
    CAPY_HIP(dh.alloc(n * 56));
    CAPY_HIP(dz.alloc(n * 56));
    TRY(sign_dev(d, n, pw.view, view_of(b), dh.as<uint8_t>(), dz.as<uint8_t>(), nullptr));
    TRY(down(h, dh, n * 56));
    return down(z_be, dz, n * 56);
}

// Declared after the DevBufs of a host-buffer entry point that launches on the side stream: whatever path the function
// leaves by, the side stream has drained before the buffers go back to the per-thread cache (common.h: DevBuf).
struct SideStreamDrain {
    hipStream_t s;
    ~SideStreamDrain() { (void)hipStreamSynchronize(s); }
};

int capy_schnorr_verify_batch(int d, size_t n, const uint8_t *pub_xy, const uint8_t *msgs, const uint64_t *offsets,
                              const uint8_t *h, const uint8_t *z_be, int32_t *status)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (!n) return CAPY_OK;
    if (!offsets || !h || !z_be || !status || !pub_xy) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, offsets, capy_schnorr_verify_batch(d, count, pub_xy + first * 112, msgs, offsets + first, h + first * 56,
                                                     z_be + first * 56, status + first));
    PackedBatch b;
    LateMsgs late = {&b, n, msgs, offsets};
    DevBuf pk, dh, dz, st;
    TRY(up(pk, pub_xy, n * 112));
    TRY(up(dh, h, n * 56));
    TRY(up(dz, z_be, n * 56));
    CAPY_HIP(st.alloc(n * 4));
    hipStream_t side = side_stream();
    SideStreamDrain drain{side};
    TRY(verify_dev(d, n, pk.as<uint8_t>(), MsgView(), dh.as<uint8_t>(), dz.as<uint8_t>(), st.as<int32_t>(), side, &late));
    CAPY_HIP(hipStreamSynchronize(side));
    return down(status, st, n * 4);
}

int capy_key_encrypt_batch(int d, size_t n, const uint8_t *pub_xy, const uint8_t *k_rand, uint8_t *msgs,
                           const uint64_t *offsets, uint8_t *z_xy, uint8_t *tags)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (!n) return CAPY_OK;
    if (!offsets || !pub_xy || !k_rand || !z_xy || !tags) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, offsets, capy_key_encrypt_batch(d, count, pub_xy + first * 112, k_rand + first * 56, msgs, offsets + first,
                                                  z_xy + first * 112, tags + first * 56));
    PackedBatch b;
    LateMsgs late = {&b, n, msgs, offsets};
    DevBuf pk, kr, dz, dt;
    kr.secret = true;  // the ephemeral scalars k: zeroed before the staging block (device cache or pinned arena) is reused
    TRY(up(pk, pub_xy, n * 112));
    TRY(up(kr, k_rand, n * 56));
    CAPY_HIP(dz.alloc(n * 112));
    CAPY_HIP(dt.alloc(n * 56));
    hipStream_t side = side_stream();
    SideStreamDrain drain{side};
    TRY(key_encrypt_dev(d, n, pk.as<uint8_t>(), kr.as<uint8_t>(), MsgView(), dz.as<uint8_t>(), dt.as<uint8_t>(), side, &late));
    CAPY_HIP(hipStreamSynchronize(side));
    TRY(b.download(n, msgs, offsets));
    TRY(down(z_xy, dz, n * 112));
    return down(tags, dt, n * 56);
}

int capy_key_decrypt_batch(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                           const uint8_t *z_xy, uint8_t *msgs, const uint64_t *offsets, const uint8_t *tags, int32_t *status)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (!n) return CAPY_OK;
    if (!offsets || !z_xy || !tags || !status) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, offsets, capy_key_decrypt_batch(d, count, pw_offsets ? pws : pws + first * pw_len, pw_len,
                                                  pw_offsets ? pw_offsets + first : nullptr, z_xy + first * 112, msgs,
                                                  offsets + first, tags + first * 56, status + first));
    PackedBatch b;
    b.msgs.secret = true;  // holds the decrypted plaintext: zeroed before the staging block is reused
    LateMsgs late = {&b, n, msgs, offsets};
    PackedKeys pw;
    TRY(pw.upload(n, pws, pw_len, pw_offsets));
    DevBuf dz, dt, st;
    TRY(up(dz, z_xy, n * 112));
    TRY(up(dt, tags, n * 56));
    CAPY_HIP(st.alloc(n * 4));
    hipStream_t side = side_stream();
    SideStreamDrain drain{side};
    TRY(key_decrypt_dev(d, n, pw.view, dz.as<uint8_t>(), MsgView(), dt.as<uint8_t>(), st.as<int32_t>(), side, &late));
    CAPY_HIP(hipStreamSynchronize(side));
    TRY(b.download(n, msgs, offsets));
    return down(status, st, n * 4);
}

// ---------------------------------------------------------------- the same calls with per-call options (capy_call_options)
#define CAPY_WITH_OPTIONS(opt, call)         \
    do {                                     \
        CallOpts _o;                         \
        TRY(parse_call_options((opt), _o));  \
        OptScope _scope(_o);                 \
        return (call);                       \
    } while (0)
#define CAPY_OPT_STREAM(opt) ((opt) ? (opt)->stream : nullptr)

int capy_ed448_scalarmul_batch_ex(size_t n, const uint8_t *scalars_be, const uint8_t *points_xy, uint8_t *out_xy,
                                  const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_ed448_scalarmul_batch(n, scalars_be, points_xy, out_xy));
}
int capy_ed448_scalarmul_batch_dev_ex(size_t n, const uint8_t *scalars_be, const uint8_t *points_xy, uint8_t *out_xy,
                                      const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_ed448_scalarmul_batch_dev(n, scalars_be, points_xy, out_xy, CAPY_OPT_STREAM(opt)));
}
int capy_ed448_basemul_batch_ex(size_t n, const uint8_t *scalars_be, uint8_t *out_xy, const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_ed448_basemul_batch(n, scalars_be, out_xy));
}
int capy_ed448_basemul_batch_dev_ex(size_t n, const uint8_t *scalars_be, uint8_t *out_xy, const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_ed448_basemul_batch_dev(n, scalars_be, out_xy, CAPY_OPT_STREAM(opt)));
}
int capy_keypair_batch_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets, uint8_t *pub_xy,
                          const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_keypair_batch(d, n, pws, pw_len, pw_offsets, pub_xy));
}
int capy_schnorr_sign_batch_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                               const uint8_t *msgs, const uint64_t *offsets, uint8_t *h, uint8_t *z_be,
                               const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_schnorr_sign_batch(d, n, pws, pw_len, pw_offsets, msgs, offsets, h, z_be));
}
int capy_schnorr_verify_batch_ex(int d, size_t n, const uint8_t *pub_xy, const uint8_t *msgs, const uint64_t *offsets,
                                 const uint8_t *h, const uint8_t *z_be, int32_t *status, const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_schnorr_verify_batch(d, n, pub_xy, msgs, offsets, h, z_be, status));
}
int capy_key_encrypt_batch_ex(int d, size_t n, const uint8_t *pub_xy, const uint8_t *k_rand, uint8_t *msgs,
                              const uint64_t *offsets, uint8_t *z_xy, uint8_t *tags, const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_key_encrypt_batch(d, n, pub_xy, k_rand, msgs, offsets, z_xy, tags));
}
int capy_key_decrypt_batch_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                              const uint8_t *z_xy, uint8_t *msgs, const uint64_t *offsets, const uint8_t *tags,
                              int32_t *status, const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_key_decrypt_batch(d, n, pws, pw_len, pw_offsets, z_xy, msgs, offsets, tags, status));
}
int capy_keypair_batch_dev_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                              uint8_t *pub_xy, const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_keypair_batch_dev(d, n, pws, pw_len, pw_offsets, pub_xy, CAPY_OPT_STREAM(opt)));
}
int capy_schnorr_sign_batch_dev_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                                   const uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride,
                                   uint8_t *h, uint8_t *z_be, const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_schnorr_sign_batch_dev(d, n, pws, pw_len, pw_offsets, msgs, offsets, uniform_len, msg_stride, h, z_be,
                                                       CAPY_OPT_STREAM(opt)));
}
int capy_schnorr_verify_batch_dev_ex(int d, size_t n, const uint8_t *pub_xy, const uint8_t *msgs, const uint64_t *offsets,
                                     uint64_t uniform_len, uint64_t msg_stride, const uint8_t *h, const uint8_t *z_be,
                                     int32_t *status, const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_schnorr_verify_batch_dev(d, n, pub_xy, msgs, offsets, uniform_len, msg_stride, h, z_be, status,
                                                         CAPY_OPT_STREAM(opt)));
}
int capy_key_encrypt_batch_dev_ex(int d, size_t n, const uint8_t *pub_xy, const uint8_t *k_rand, uint8_t *msgs,
                                  const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride, uint8_t *z_xy,
                                  uint8_t *tags, const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_key_encrypt_batch_dev(d, n, pub_xy, k_rand, msgs, offsets, uniform_len, msg_stride, z_xy, tags,
                                                      CAPY_OPT_STREAM(opt)));
}
int capy_key_decrypt_batch_dev_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                                  const uint8_t *z_xy, uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len,
                                  uint64_t msg_stride, const uint8_t *tags, int32_t *status, const capy_call_options *opt)
{
    CAPY_WITH_OPTIONS(opt, capy_key_decrypt_batch_dev(d, n, pws, pw_len, pw_offsets, z_xy, msgs, offsets, uniform_len, msg_stride, tags,
                                                      status, CAPY_OPT_STREAM(opt)));
}

}  // extern "C"
