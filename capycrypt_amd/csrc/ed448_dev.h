// ed448_dev.h — Ed448-Goldilocks field and point arithmetic for one (scalar, point) pair per lane.
//
// Stands in for the un-vendored crate tiny_ed448_goldilocks 0.1.8 at its call sites in
// /root/reference/src/ecc: untwisted Edwards curve x^2+y^2 = 1 + d x^2 y^2,
// d = -39081, p = 2^448 - 2^224 - 1 (RFC 7748 §4.2 / RFC 8032 §5.2).
//
// Field element: 16 limbs of 28 bits in 16 VGPRs (radix 2^28, little endian).  Products are
// accumulated with v_mad_u64_u32 (32x32+64 -> 64 in one instruction; measured 1.2x the cost of
// v_mul_lo_u32 on gfx950, tools/microbench.hip) over the Goldilocks split a = a0 + a1*phi, phi = 2^224,
// phi^2 = phi + 1, in Karatsuba form by default (192 MADs, fe_mul below; the plain 256-MAD form is kept behind
// CAPY_ED448_KARATSUBA=0).  north_star suggests u64 limbs with __umul64hi; on gfx950 a 64x64 product is 4 MADs plus carry
// adds, so 28-bit limbs with lazy carries do the same multiplication in fewer, cheaper instructions.
//
// Everything is __host__ __device__ so the same code is unit-tested on the CPU (tests/test_ed448_host.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CAPY_HD __host__ __device__
// whole algorithms are inlined into every kernel that uses them: as shared out-of-line functions they lose ~8 %
// (calling convention, callee-saved registers)
#define CAPY_HD_INLINE __host__ __device__ __attribute__((always_inline)) inline

namespace capy {

struct Fe {
    uint32_t l[16];
};
struct Pt {  // extended homogeneous coordinates, x = X/Z, y = Y/Z, T = XY/Z
    Fe X, Y, Z, T;
};

constexpr uint32_t M28 = 0x0fffffffu;
constexpr uint32_t ED448_D_ABS = 39081;  // d = -39081

CAPY_HD inline Fe fe_zero()
{
    Fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.l[i] = 0;
    return r;
}
CAPY_HD inline Fe fe_one()
{
    Fe r = fe_zero();
    r.l[0] = 1;
    return r;
}

// carry-save normalisation: limbs < 2^31 in, limbs <= 2^28 + 8 out (value unchanged mod p)
CAPY_HD inline void fe_weak_reduce(Fe &r)
{
    uint32_t c[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        c[i] = r.l[i] >> 28;
        r.l[i] &= M28;
    }
#pragma unroll
    for (int i = 1; i < 16; i++) r.l[i] += c[i - 1];
    r.l[0] += c[15];  // 2^448 = 2^224 + 1
    r.l[8] += c[15];
}

CAPY_HD inline Fe fe_add(const Fe &a, const Fe &b)
{
    Fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.l[i] = a.l[i] + b.l[i];
    fe_weak_reduce(r);
    return r;
}

// a - b + 2p, limbs of b must be <= 2^28 + 8
CAPY_HD inline Fe fe_sub(const Fe &a, const Fe &b)
{
    Fe r;
#pragma unroll
    // limbs of 2p: 2*(2^28-1), except limb 8 (weight 2^224) = 2*(2^28-2)
    for (int i = 0; i < 16; i++) r.l[i] = a.l[i] + (i == 8 ? 2 * (M28 - 1) : 2 * M28) - b.l[i];
    fe_weak_reduce(r);
    return r;
}

CAPY_HD inline Fe fe_neg(const Fe &a) { return fe_sub(fe_zero(), a); }

// ---- lazily reduced forms.  Notation: R = a limb bound of 2^28 + 2^10 (every fe_mul / fe_sqr / weak-reduced
// output).  fe_mul(a, b) is exact while 38 * max_limb(a) * max_limb(b) < 2^64 and fe_sqr(a) while
// 40 * max_limb(a)^2 < 2^64 (column-sum bounds, see fe_mul), i.e. products of limb bounds up to 2^58.7.  The point
// formulas below use these unreduced sums / differences only where that holds; each use states its bound.
//   fe_add_nr(R, R)        <= 2^29 + 2^11
//   fe_sub_nr(a, b<=2p_l)  <= max_limb(a) + 2^29      (b must be R so that 2p - b >= 0 limb-wise)
//   fe_neg_nr(R)           <= 2^29
CAPY_HD inline Fe fe_add_nr(const Fe &a, const Fe &b)
{
    Fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
CAPY_HD inline Fe fe_sub_nr(const Fe &a, const Fe &b)
{
    Fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.l[i] = a.l[i] + (i == 8 ? 2 * (M28 - 1) : 2 * M28) - b.l[i];
    return r;
}
CAPY_HD inline Fe fe_neg_nr(const Fe &a)
{
    Fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.l[i] = (i == 8 ? 2 * (M28 - 1) : 2 * M28) - a.l[i];
    return r;
}
// a - b + 4p, reduced; limbs of b may be up to 2^30 - 8
CAPY_HD inline Fe fe_sub4(const Fe &a, const Fe &b)
{
    Fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.l[i] = a.l[i] + (i == 8 ? 4 * (M28 - 1) : 4 * M28) - b.l[i];
    fe_weak_reduce(r);
    return r;
}

// CAPY_ED448_SETPRIO (r04): raise the wave's priority around the 4-cycle instructions of fe_mul / fe_sqr (multiply-add chains,
// 64-bit subtractions, carry chain) so that the other wave of the SIMD can issue its SIMPLE instructions (limb sums, masks, the
// limb arithmetic between products) in the half windows they leave (profiles/r03_valu_issue_bisect.txt, finding 4).
//   1: priority 1 from the first multiply-add to the last;  2: priority 1 up to the end of the carry chain, the 16 masks of the
//   result deferred into one block at priority 0 (so that a wave at priority 0 holds simple instructions only).
// Pays in vb2_kernel only (two waves per SIMD, pinned chains): profiles/r04_ed448_setprio.txt.
#ifndef CAPY_ED448_SETPRIO
#define CAPY_ED448_SETPRIO 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && CAPY_ED448_SETPRIO
#define CAPY_ED448_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define CAPY_ED448_PRIO(x)
#endif
#if defined(__HIP_DEVICE_COMPILE__) && CAPY_ED448_SETPRIO == 2
#define CAPY_ED448_PRIO_TAIL(x) __builtin_amdgcn_s_setprio(x)
#define CAPY_ED448_PRIO_MID(x)
#else
#define CAPY_ED448_PRIO_TAIL(x)
#define CAPY_ED448_PRIO_MID(x) CAPY_ED448_PRIO(x)
#endif

// 16 column sums (lo = columns 0..7, hi = 8..15), each < 2^63, to 28-bit limbs
CAPY_HD inline Fe fe_from_columns(uint64_t lo[8], uint64_t hi[8])
{
    Fe r;
    uint64_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__) && CAPY_ED448_SETPRIO == 2
    // the carry chain at priority 1 with the low dwords kept unmasked, then all masks and the wrap in one block at priority 0
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t v = lo[k] + c;
        r.l[k] = (uint32_t)v;
        c = v >> 28;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t v = hi[k] + c;
        r.l[8 + k] = (uint32_t)v;
        c = v >> 28;
    }
    asm volatile("" ::: "memory");
    CAPY_ED448_PRIO_TAIL(0);
#pragma unroll
    for (int k = 0; k < 16; k++) r.l[k] &= M28;
#else
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t v = lo[k] + c;
        r.l[k] = (uint32_t)v & M28;
        c = v >> 28;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t v = hi[k] + c;
        r.l[8 + k] = (uint32_t)v & M28;
        c = v >> 28;
    }
#endif
    // c * 2^448 = c * (2^224 + 1), c < 2^36
    uint64_t v = r.l[0] + c;
    r.l[0] = (uint32_t)v & M28;
    v = r.l[1] + (v >> 28);
    r.l[1] = (uint32_t)v & M28;
    r.l[2] += (uint32_t)(v >> 28);
    v = r.l[8] + c;
    r.l[8] = (uint32_t)v & M28;
    v = r.l[9] + (v >> 28);
    r.l[9] = (uint32_t)v & M28;
    r.l[10] += (uint32_t)(v >> 28);
    return r;
}

// acc + x * y as ONE v_mad_u64_u32, in the order written.  The optimiser re-associates a column's sum of products and
// addends freely and then needs a separate 64-bit addition wherever a value other than the running sum should have been
// the first multiply-add's addend (r04: 33 v_lshl_add_u64 per fe_mul where 26 are needed); the asm pins the chain.
// CAPY_ED448_ASM_MAD=0 keeps the plain C form (also what the host build runs).  Pinned chains pay in ONE kernel, vb2_kernel
// (+1.1 % at 2^18 items; ed448_vb2.hip defines the macro to 1); everywhere else they cost -- 2.7 % (one item per lane at
// 65 536 items) to 2.1x (vb_ct_kernel: its table scan no longer overlaps the arithmetic) -- so the default is 0
// (profiles/r04_ed448_pinned_chains.txt).
#ifndef CAPY_ED448_ASM_MAD
#define CAPY_ED448_ASM_MAD 0
#endif
CAPY_HD inline uint64_t mad64(uint32_t x, uint32_t y, uint64_t acc)
{
#if defined(__HIP_DEVICE_COMPILE__) && CAPY_ED448_ASM_MAD
    // (a literal v_mad_u64_u32 in the asm would do the same, but the hazard recogniser then puts an s_nop after every one
    // of them and the optimiser no longer sinks dead products out of loops: 12 % slower, measured)
    uint64_t d = (uint64_t)x * y + acc;
    asm("" : "+v"(d));
    return d;
#else
    return (uint64_t)x * y + acc;
#endif
}

// CAPY_ED448_INLINE=1: inline fe_mul / fe_sqr at every call site (no call, no stack traffic, larger code).
#ifndef CAPY_ED448_INLINE
#define CAPY_ED448_INLINE 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !CAPY_ED448_INLINE
#define CAPY_NOINLINE __noinline__
#else
#define CAPY_NOINLINE
#endif

#ifdef CAPY_FE_CHECK_BOUNDS
// host-only audit of the lazy-reduction bounds (tests/native/ed448_host_test.cpp builds with this)
extern "C" void capy_fe_bound_violation(const char *what, double bits);
inline void fe_check_mul(const Fe &a, const Fe &b)
{
    uint64_t ma = 0, mb = 0;
    for (int i = 0; i < 16; i++) {
        ma = a.l[i] > ma ? a.l[i] : ma;
        mb = b.l[i] > mb ? b.l[i] : mb;
    }
    const long double prod = 38.0L * (long double)ma * (long double)mb;
    if (prod >= 18446744073709551616.0L) capy_fe_bound_violation("fe_mul", (double)prod);
    // Karatsuba adds the halves of BOTH operands in 32 bits: a0 + a1 < 2^32 needs limbs < 2^31
    if (ma >= (1ull << 31) || mb >= (1ull << 31)) capy_fe_bound_violation("fe_mul operand sum", (double)(ma > mb ? ma : mb));
}
inline void fe_check_sqr(const Fe &a)
{
    uint64_t ma = 0;
    for (int i = 0; i < 16; i++) ma = a.l[i] > ma ? a.l[i] : ma;
    const long double prod = 40.0L * (long double)ma * (long double)ma;
    if (prod >= 18446744073709551616.0L) capy_fe_bound_violation("fe_sqr", (double)prod);
    // Karatsuba squaring doubles a0 + a1 in 32 bits: 2 (a0 + a1) < 2^32 needs limbs < 2^30
    if (ma >= (1ull << 30)) capy_fe_bound_violation("fe_sqr operand sum", (double)ma);
}
#define CAPY_FE_CHECK_MUL(a, b) fe_check_mul(a, b)
#define CAPY_FE_CHECK_SQR(a) fe_check_sqr(a)
#else
#define CAPY_FE_CHECK_MUL(a, b)
#define CAPY_FE_CHECK_SQR(a)
#endif

// r = a * b mod p.  256 MADs.  Exact while 38 * max_limb(a) * max_limb(b) < 2^64: with bs = b0 + b1 <= 2 Lb, column
// hi[k] collects (k+1) pairs of (a0 b1 + a1 bs) <= 3 La Lb, (7-k) pairs of (a0 b0 + a1 b1) <= 2 La Lb and qh[k] of
// (7-k) pairs <= 3 La Lb: at most (38 - 2k) La Lb.  Output limbs <= 2^28 + 2^9.
// 1: Karatsuba multiplication / squaring (default: +5 % variable base, +10 % fixed base on MI355X, because the Ed448
// kernels are power-limited and a 32x32->64 multiply-add costs more energy than the additions that replace it);
// 0: the plain 256 / 136-MAD forms (same instruction count within 10 %).
// r03, tried and dropped (profiles/r03_ed448_forms.txt): the -AA columns accumulated as (-a0[i]) * b0[j] by
// v_mad_i64_i32 instead of 15 64-bit subtractions (each a v_sub_co / v_subb_co pair plus the s_nop their vcc hazard
// costs on gfx950).  It needs 36 (multiplication) / 20 (squaring) more multiply-adds and 8-16 negations: 5497 VALU in the
// doubling loop of vb2_kernel against 5266 + 97 s_nop here, and the optimiser only emits the signed multiply-add when
// the operands' value ranges are hidden from it (otherwise four instructions each: 23.5 instead of 26.6 M/s).
#ifndef CAPY_ED448_KARATSUBA
#define CAPY_ED448_KARATSUBA 1
#endif
#if CAPY_ED448_KARATSUBA
// Karatsuba over the Goldilocks split: AA = a0 b0, BB = a1 b1, CC = (a0 + a1)(b0 + b1);
//   lo[k] = AA[k] + BB[k] + CC[k+8] - AA[k+8]      hi[k] = BB[k+8] + CC[k] + CC[k+8] - AA[k]
// 192 MADs instead of 256, for 16 limb sums, 30 64-bit subtractions and 14 additions: about the same instruction count,
// but a quarter fewer multiplies, which is what the power-limited Ed448 kernels pay for.  Everything is arithmetic
// mod 2^64; the final columns are the same sums as in the plain form, so the same operand bound makes them exact.
CAPY_HD CAPY_NOINLINE inline Fe fe_mul(const Fe a, const Fe b)
{
    CAPY_FE_CHECK_MUL(a, b);
    uint32_t as[8], bs[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        as[j] = a.l[j] + a.l[8 + j];
        bs[j] = b.l[j] + b.l[8 + j];
    }
    uint64_t aa[15], lo[8], hi[8], cch[7];
#pragma unroll
    for (int k = 0; k < 15; k++) aa[k] = 0;
    CAPY_ED448_PRIO(1);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) aa[i + j] = mad64(a.l[i], b.l[j], aa[i + j]);
    // the upper columns of CC first: cch[k] is an addend of lo[k] AND the start of hi[k]'s multiply-add chain (r04: a
    // chain's first multiply-add takes any 64-bit addend for free, so `hi[k] += cch[k]` costs nothing this way)
#pragma unroll
    for (int k = 0; k < 7; k++) cch[k] = 0;
#pragma unroll
    for (int i = 1; i < 8; i++)
#pragma unroll
        for (int j = 8 - i; j < 8; j++) cch[i + j - 8] = mad64(as[i], bs[j], cch[i + j - 8]);
    // lo starts from AA[k] and takes BB[k]; hi starts from CC[k + 8] and takes CC[k] and BB[k + 8]
#pragma unroll
    for (int k = 0; k < 8; k++) {
        lo[k] = aa[k];
        hi[k] = k < 7 ? cch[k] : 0;
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = i + j;
            if (k < 8) {
                lo[k] = mad64(a.l[8 + i], b.l[8 + j], lo[k]);
                hi[k] = mad64(as[i], bs[j], hi[k]);
            } else {
                hi[k - 8] = mad64(a.l[8 + i], b.l[8 + j], hi[k - 8]);
            }
        }
    CAPY_ED448_PRIO_MID(0);
#pragma unroll
    for (int k = 0; k < 8; k++) hi[k] -= aa[k];
#pragma unroll
    for (int k = 0; k < 7; k++) lo[k] += cch[k] - aa[k + 8];
    return fe_from_columns(lo, hi);
}
#else
CAPY_HD CAPY_NOINLINE inline Fe fe_mul(const Fe a, const Fe b)
{
    CAPY_FE_CHECK_MUL(a, b);
    uint64_t lo[8], hi[8], qh[7];
    uint32_t bs[8];
#pragma unroll
    for (int j = 0; j < 8; j++) bs[j] = b.l[j] + b.l[8 + j];
#pragma unroll
    for (int k = 0; k < 8; k++) lo[k] = hi[k] = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) qh[k] = 0;
    // P = a0 b0 + a1 b1 ; Q = a0 b1 + a1 (b0 + b1)
    // result_lo[k] = P[k] + Q[k+8] ; result_hi[k] = P[k+8] + Q[k] + Q[k+8]
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = i + j;
            const uint64_t p1 = (uint64_t)a.l[i] * b.l[j], p2 = (uint64_t)a.l[8 + i] * b.l[8 + j];
            const uint64_t q1 = (uint64_t)a.l[i] * b.l[8 + j], q2 = (uint64_t)a.l[8 + i] * bs[j];
            if (k < 8) {
                lo[k] += p1;
                lo[k] += p2;
                hi[k] += q1;
                hi[k] += q2;
            } else {
                hi[k - 8] += p1;
                hi[k - 8] += p2;
                qh[k - 8] += q1;
                qh[k - 8] += q2;
            }
        }
#pragma unroll
    for (int k = 0; k < 7; k++) {
        lo[k] += qh[k];
        hi[k] += qh[k];
    }
    return fe_from_columns(lo, hi);
}

#endif

#if CAPY_ED448_KARATSUBA
// r = a^2 mod p, Karatsuba as in fe_mul with AA = a0^2, BB = a1^2, CC = (a0 + a1)^2: 3 x 36 = 108 MADs instead of 136.
// r04: only AA (the product that is SUBTRACTED) is formed with pre-doubled operands.  BB and CC are kept as
//   off[k]  = sum over i < j, i + j = k of x_i x_j        (the off-diagonal HALF sum)      and
//   diag[k] = x_(k/2)^2 for even k,
// the off-diagonal sums of a result column share one multiply-add chain, and the column is
//   (off << 1) + (carry + diag terms + AA terms)
// -- one v_lshl_add_u64, which doubles, adds and takes the place of the carry addition of fe_from_columns; the diagonal
// squares ride on a chain that starts from the carry.  16 operand doublings, 7 `hi += cch` and 8 carry additions go.
CAPY_HD CAPY_NOINLINE inline Fe fe_sqr(const Fe a)
{
    CAPY_FE_CHECK_SQR(a);
    uint32_t as[8], d0[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        as[j] = a.l[j] + a.l[8 + j];  // < 2^31 under the operand bound
        d0[j] = 2 * a.l[j];
    }
    uint64_t aa[15];
#pragma unroll
    for (int k = 0; k < 15; k++) aa[k] = 0;
    CAPY_ED448_PRIO(1);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = i; j < 8; j++) aa[i + j] = mad64(i < j ? d0[i] : a.l[i], a.l[j], aa[i + j]);
    // off-diagonal half sums: och[k] of CC column k + 8 (shared by lo[k] and hi[k]); olo[k] = och[k] + BB[k];
    // ohi[k] = och[k] + CC[k] + BB[k + 8]
    uint64_t och[7], olo[8], ohi[8];
#pragma unroll
    for (int k = 0; k < 7; k++) och[k] = 0;
#pragma unroll
    for (int i = 1; i < 8; i++)
#pragma unroll
        for (int j = i + 1; j < 8; j++)
            if (i + j >= 8) och[i + j - 8] = mad64(as[i], as[j], och[i + j - 8]);
#pragma unroll
    for (int k = 0; k < 8; k++) olo[k] = ohi[k] = k < 7 ? och[k] : 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = i + 1; j < 8; j++) {
            const int k = i + j;
            if (k < 8) {
                olo[k] = mad64(a.l[8 + i], a.l[8 + j], olo[k]);
                ohi[k] = mad64(as[i], as[j], ohi[k]);
            } else {
                ohi[k - 8] = mad64(a.l[8 + i], a.l[8 + j], ohi[k - 8]);
            }
        }
    CAPY_ED448_PRIO_MID(0);
    // columns in carry order; the diagonal squares of column k sit at index k / 2 (even k only)
    Fe r;
    uint64_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__) && CAPY_ED448_SETPRIO == 2
    constexpr uint32_t MASK_NOW = 0xffffffffu;  // masks deferred into one block behind the carry chain (see fe_from_columns)
#else
    constexpr uint32_t MASK_NOW = M28;
#endif
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t d = c;
        if (k % 2 == 0) {
            d = mad64(a.l[8 + k / 2], a.l[8 + k / 2], d);                // BB[k]
            if (k < 7) d = mad64(as[(k + 8) / 2], as[(k + 8) / 2], d);  // CC[k + 8]
        }
        d += aa[k];
        if (k < 7) d -= aa[k + 8];
        const uint64_t v = (olo[k] << 1) + d;
        r.l[k] = (uint32_t)v & MASK_NOW;
        c = v >> 28;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t d = c;
        if (k % 2 == 0) {
            d = mad64(as[k / 2], as[k / 2], d);  // CC[k]
            if (k < 7) {
                d = mad64(as[(k + 8) / 2], as[(k + 8) / 2], d);            // CC[k + 8]
                d = mad64(a.l[8 + (k + 8) / 2], a.l[8 + (k + 8) / 2], d);  // BB[k + 8]
            }
        }
        d -= aa[k];
        const uint64_t v = (ohi[k] << 1) + d;
        r.l[8 + k] = (uint32_t)v & MASK_NOW;
        c = v >> 28;
    }
#if defined(__HIP_DEVICE_COMPILE__) && CAPY_ED448_SETPRIO == 2
    asm volatile("" ::: "memory");
    CAPY_ED448_PRIO_TAIL(0);
#pragma unroll
    for (int k = 0; k < 16; k++) r.l[k] &= M28;
#endif
    // c * 2^448 = c * (2^224 + 1), c < 2^36
    uint64_t v = r.l[0] + c;
    r.l[0] = (uint32_t)v & M28;
    v = r.l[1] + (v >> 28);
    r.l[1] = (uint32_t)v & M28;
    r.l[2] += (uint32_t)(v >> 28);
    v = r.l[8] + c;
    r.l[8] = (uint32_t)v & M28;
    v = r.l[9] + (v >> 28);
    r.l[9] = (uint32_t)v & M28;
    r.l[10] += (uint32_t)(v >> 28);
    return r;
}
#else
// r = a^2 mod p.  P = a0^2 + a1^2 ; Q = a1 (2 a0 + a1).  136 MADs.
CAPY_HD CAPY_NOINLINE inline Fe fe_sqr(const Fe a)
{
    CAPY_FE_CHECK_SQR(a);
    uint64_t lo[8], hi[8], qh[7];
    uint32_t s[8], d0[8], d1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        s[j] = 2 * a.l[j] + a.l[8 + j];  // < 2^30
        d0[j] = 2 * a.l[j];
        d1[j] = 2 * a.l[8 + j];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) lo[k] = hi[k] = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) qh[k] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = i + j;
            // P: symmetric, take i <= j with the doubled operand for i < j
            if (i <= j) {
                const uint64_t p1 = (uint64_t)(i < j ? d0[i] : a.l[i]) * a.l[j];
                const uint64_t p2 = (uint64_t)(i < j ? d1[i] : a.l[8 + i]) * a.l[8 + j];
                if (k < 8) {
                    lo[k] += p1;
                    lo[k] += p2;
                } else {
                    hi[k - 8] += p1;
                    hi[k - 8] += p2;
                }
            }
            const uint64_t q = (uint64_t)a.l[8 + i] * s[j];
            if (k < 8)
                hi[k] += q;
            else
                qh[k - 8] += q;
        }
#pragma unroll
    for (int k = 0; k < 7; k++) {
        lo[k] += qh[k];
        hi[k] += qh[k];
    }
    return fe_from_columns(lo, hi);
}

#endif

// a * k for a small constant k < 2^17
CAPY_HD inline Fe fe_mul_small(const Fe &a, uint32_t k)
{
    Fe r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint64_t v = (uint64_t)a.l[i] * k + c;
        r.l[i] = (uint32_t)v & M28;
        c = v >> 28;
    }
    r.l[0] += (uint32_t)c;
    r.l[8] += (uint32_t)c;
    fe_weak_reduce(r);
    return r;
}

CAPY_HD inline Fe fe_sqrn(Fe a, int n)
{
#pragma unroll 1
    for (int i = 0; i < n; i++) a = fe_sqr(a);
    return a;
}

// a^(p-2); p-2 = [223 ones][0][222 ones][0][1] in binary
CAPY_HD inline Fe fe_inv(const Fe &a)
{
    Fe x2 = fe_mul(fe_sqr(a), a);
    Fe x3 = fe_mul(fe_sqr(x2), a);
    Fe x6 = fe_mul(fe_sqrn(x3, 3), x3);
    Fe x9 = fe_mul(fe_sqrn(x6, 3), x3);
    Fe x18 = fe_mul(fe_sqrn(x9, 9), x9);
    Fe x19 = fe_mul(fe_sqr(x18), a);
    Fe x37 = fe_mul(fe_sqrn(x19, 18), x18);
    Fe x74 = fe_mul(fe_sqrn(x37, 37), x37);
    Fe x111 = fe_mul(fe_sqrn(x74, 37), x37);
    Fe x222 = fe_mul(fe_sqrn(x111, 111), x111);
    Fe x223 = fe_mul(fe_sqr(x222), a);
    Fe t = fe_mul(fe_sqrn(x223, 223), x222);
    return fe_mul(fe_sqrn(t, 2), a);
}

// full reduction to the canonical representative in [0, p), limbs < 2^28
CAPY_HD inline void fe_canon(Fe &r)
{
    // sequential carries (twice: the wrap of the top carry can ripple once more)
#pragma unroll
    for (int pass = 0; pass < 3; pass++) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint32_t v = r.l[i] + c;
            r.l[i] = v & M28;
            c = v >> 28;
        }
        r.l[0] += c;
        r.l[8] += c;
    }
    // now 0 <= r < 2^448 (+ at most a limb overflow of 1 at l[0]/l[8], absorbed below); subtract p if r >= p:
    // r >= p  <=>  r + 2^224 + 1 >= 2^448
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        uint32_t t[16], c = 1;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint32_t v = r.l[i] + c + (i == 8 ? 1u : 0u);
            t[i] = v & M28;
            c = v >> 28;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) r.l[i] = c ? t[i] : r.l[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fe_inv_gcd: 1/a mod p by Bernstein-Yang division steps ("Fast constant-time gcd computation and modular inversion",
// TCHES 2019(3)) instead of the 445 squarings + 13 multiplications of a^(p-2) above (~133 000 VALU instructions per
// lane; this form: ~40 000).  The result is the same field element (0 for a = 0), so nothing observable changes.
//   divstep(delta, f, g) = (1 - delta, g, (g - f) / 2)          if delta > 0 and g odd
//                          (1 + delta, f, (g + (g mod 2) f) / 2) otherwise
// started at (1, p, a).  Theorem 11.2 of the paper: for 0 <= g < f < 2^d, f odd, d >= 46, floor((49 d + 57) / 17)
// steps reach g = 0 with f = +-gcd; d = 448 gives 1294, this code runs 44 x 30 = 1320.  Every step is executed for every
// input (no data-dependent branch or address: a uniform instruction stream is also what the lanes of a wave need).
// Thirty steps at a time on the low words of f and g give a 2 x 2 transition matrix with entries of at most 30 bits
// (t.u f + t.v g, t.q f + t.r g are the new 2^30 f, 2^30 g); the matrix is then applied to the full f, g (exact division
// by 2^30) and, modulo p, to the cofactors d, e that satisfy f = d a, g = e a (mod p): at the end 1/a = sign(f) d.
// Numbers are 15 signed limbs of 30 bits (the top limb carries the sign); p = 2^448 - 2^224 - 1 has three non-zero
// SIGNED limbs: -1 at limb 0, -2^14 at limb 7 (224 = 7 x 30 + 14), +2^28 at limb 14; p = -1 mod 2^30.
struct GcdNum {
    int32_t v[15];
};
struct GcdMat {
    int32_t u, v, q, r;
};
constexpr int32_t GCD_M30 = 0x3fffffff;
// hides a value's known range from the optimiser: with the limbs known to be non-negative it turns a signed
// 32 x 32 + 64 multiply-add into an unsigned one plus fix-ups (four instructions instead of one v_mad_i64_i32)
CAPY_HD inline int32_t gcd_opaque(int32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}

// 30 division steps on the low words.  eta = -delta.  Masks instead of branches: c_odd = g is odd, c_swap = c_odd and
// delta > 0.  g += c_odd & (c_swap ? -f : f), then f += c_swap & g makes f the old g; u, v / q, r follow f / g.
CAPY_HD inline int32_t gcd_divsteps_30(int32_t eta, uint32_t f, uint32_t g, GcdMat &t)
{
    uint32_t u = 1, v = 0, q = 0, r = 1;
#pragma unroll
    for (int i = 0; i < 30; i++) {
        const uint32_t c_neg = (uint32_t)(eta >> 31);            // all ones iff delta > 0
        const uint32_t c_odd = 0u - (g & 1u);
        const uint32_t c_swap = c_neg & c_odd;
        const uint32_t one = c_swap >> 31;                        // -x = ~x + 1
        g += ((f ^ c_neg) & c_odd) + one;                         // (c_neg ? -f : f) & c_odd: the +1 of a negation only counts
        q += ((u ^ c_neg) & c_odd) + one;                         // when c_odd is set too, i.e. exactly when c_swap is
        r += ((v ^ c_neg) & c_odd) + one;
        eta = (int32_t)(((uint32_t)eta ^ c_swap) + one) - 1;      // swap: -eta - 1 (= ~eta); otherwise eta - 1
        f += g & c_swap;
        u += q & c_swap;
        v += r & c_swap;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return eta;
}

// (f, g) <- (t.u f + t.v g, t.q f + t.r g) / 2^30, exact
CAPY_HD inline void gcd_update_fg(GcdNum &f, GcdNum &g, const GcdMat &t)
{
    int32_t fi = gcd_opaque(f.v[0]), gi = gcd_opaque(g.v[0]);
    int64_t cf = (int64_t)t.u * fi + (int64_t)t.v * gi;
    int64_t cg = (int64_t)t.q * fi + (int64_t)t.r * gi;
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 15; i++) {
        fi = gcd_opaque(f.v[i]);
        gi = gcd_opaque(g.v[i]);
        cf += (int64_t)t.u * fi + (int64_t)t.v * gi;
        cg += (int64_t)t.q * fi + (int64_t)t.r * gi;
        f.v[i - 1] = (int32_t)cf & GCD_M30;
        g.v[i - 1] = (int32_t)cg & GCD_M30;
        cf >>= 30;
        cg >>= 30;
    }
    f.v[14] = (int32_t)cf;
    g.v[14] = (int32_t)cg;
}

// (d, e) <- (t.u d + t.v e, t.q d + t.r e) / 2^30 mod p.  d, e stay in (-2p, p): a negative input is first taken as
// its value + p (md, me start from the matrix entries of the negative operands), then a multiple of p is added that
// clears the low 30 bits (p^-1 = -1 mod 2^30: md has to end up = cd mod 2^30) so that the division is exact.
CAPY_HD inline void gcd_update_de(GcdNum &d, GcdNum &e, const GcdMat &t)
{
    const int32_t sd = d.v[14] >> 31, se = e.v[14] >> 31;
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    int32_t di = gcd_opaque(d.v[0]), ei = gcd_opaque(e.v[0]);
    int64_t cd = (int64_t)t.u * di + (int64_t)t.v * ei;
    int64_t ce = (int64_t)t.q * di + (int64_t)t.r * ei;
    md -= (int32_t)(((uint32_t)md - (uint32_t)cd) & (uint32_t)GCD_M30);
    me -= (int32_t)(((uint32_t)me - (uint32_t)ce) & (uint32_t)GCD_M30);
    cd -= md;  // + p.limb[0] * md, p.limb[0] = -1
    ce -= me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 15; i++) {
        di = gcd_opaque(d.v[i]);
        ei = gcd_opaque(e.v[i]);
        cd += (int64_t)t.u * di + (int64_t)t.v * ei;
        ce += (int64_t)t.q * di + (int64_t)t.r * ei;
        if (i == 7) {  // p.limb[7] = -2^14
            cd -= (int64_t)md * (1 << 14);
            ce -= (int64_t)me * (1 << 14);
        }
        if (i == 14) {  // p.limb[14] = +2^28
            cd += (int64_t)md * (1 << 28);
            ce += (int64_t)me * (1 << 28);
        }
        d.v[i - 1] = (int32_t)cd & GCD_M30;
        e.v[i - 1] = (int32_t)ce & GCD_M30;
        cd >>= 30;
        ce >>= 30;
    }
    d.v[14] = (int32_t)cd;
    e.v[14] = (int32_t)ce;
}

CAPY_HD inline Fe fe_inv_gcd(Fe a)
{
    fe_canon(a);
    // 16 x 28 bits -> 15 x 30 bits
    GcdNum f, g, d, e;
    {
        uint32_t w[15];
#pragma unroll
        for (int i = 0; i < 15; i++) w[i] = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int bit = 28 * k, j = bit >> 5, sh = bit & 31;
            const uint64_t x = (uint64_t)a.l[k] << sh;
            w[j] |= (uint32_t)x;
            w[j + 1] |= (uint32_t)(x >> 32);
        }
#pragma unroll
        for (int k = 0; k < 15; k++) {
            const int bit = 30 * k, j = bit >> 5, sh = bit & 31;
            const uint64_t x = ((uint64_t)(j + 1 < 15 ? w[j + 1] : 0u) << 32) | w[j];
            g.v[k] = (int32_t)((uint32_t)(x >> sh) & (uint32_t)GCD_M30);
        }
    }
#pragma unroll
    for (int k = 0; k < 15; k++) {
        f.v[k] = GCD_M30;  // p in unsigned limbs: all ones up to bit 447, bit 224 clear
        d.v[k] = 0;
        e.v[k] = 0;
    }
    f.v[7] = GCD_M30 - (1 << 14);
    f.v[14] = (1 << 28) - 1;
    e.v[0] = 1;
    int32_t eta = -1;
#pragma unroll 1
    for (int it = 0; it < 44; it++) {
        GcdMat t;
        eta = gcd_divsteps_30(eta, (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30), (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30), t);
        gcd_update_de(d, e, t);
        gcd_update_fg(f, g, t);
    }
    // g = 0 now and f = +-1 (or +-p for a = 0, where d = 0): the inverse is sign(f) d, brought into [0, p)
    const int32_t sf = f.v[14] >> 31;
    {
        int32_t c = 0;
#pragma unroll
        for (int k = 0; k < 15; k++) {  // d <- sf ? -d : d   (limb-wise negation, then carries)
            const int32_t x = ((d.v[k] ^ sf) - sf) + c;
            d.v[k] = k < 14 ? (x & GCD_M30) : x;
            c = k < 14 ? (x >> 30) : 0;
        }
    }
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {  // d in (-2p, 2p) -> [0, p): add p while negative ...
        const int32_t neg = d.v[14] >> 31;
        int32_t c = 0;
#pragma unroll
        for (int k = 0; k < 15; k++) {
            const int32_t pk = k == 0 ? -1 : (k == 7 ? -(1 << 14) : (k == 14 ? (1 << 28) : 0));
            const int32_t x = d.v[k] + (pk & neg) + c;
            d.v[k] = k < 14 ? (x & GCD_M30) : x;
            c = k < 14 ? (x >> 30) : 0;
        }
    }
    // ... and the representative may still be >= p (only when it was in [p, 2p)): fe_canon below takes care of that
    Fe r;
    {
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = 0;
#pragma unroll
        for (int k = 0; k < 15; k++) {
            const int bit = 30 * k, j = bit >> 5, sh = bit & 31;
            const uint64_t x = (uint64_t)(uint32_t)d.v[k] << sh;
            w[j] |= (uint32_t)x;
            w[j + 1] |= (uint32_t)(x >> 32);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int bit = 28 * k, j = bit >> 5, sh = bit & 31;
            const uint64_t x = ((uint64_t)(j + 1 < 16 ? w[j + 1] : 0u) << 32) | w[j];
            r.l[k] = (uint32_t)(x >> sh) & (k < 15 ? M28 : 0xffffffffu);  // the value is < 2p < 2^449: the top limb keeps bit 448
        }
    }
    fe_weak_reduce(r);
    return r;
}

// the inversion behind every projective -> affine conversion.  1 (default): division steps; 0: the a^(p-2) chain (A/B)
#ifndef CAPY_ED448_GCD_INV
#define CAPY_ED448_GCD_INV 1
#endif
CAPY_HD inline Fe fe_inv_out(const Fe &a)
{
#if CAPY_ED448_GCD_INV
    return fe_inv_gcd(a);
#else
    return fe_inv(a);
#endif
}

// 56 little-endian bytes <-> limbs (input need not be < p)
CAPY_HD inline Fe fe_from_bytes(const uint8_t *in)
{
    uint32_t w[15];
#pragma unroll
    for (int i = 0; i < 14; i++)
        w[i] = (uint32_t)in[4 * i] | ((uint32_t)in[4 * i + 1] << 8) | ((uint32_t)in[4 * i + 2] << 16) |
               ((uint32_t)in[4 * i + 3] << 24);
    w[14] = 0;
    Fe r;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int bit = 28 * k, j = bit >> 5, s = bit & 31;
        uint64_t v = ((uint64_t)w[j + 1] << 32) | w[j];
        r.l[k] = (uint32_t)(v >> s) & M28;
    }
    return r;
}

CAPY_HD inline void fe_to_bytes(uint8_t *out, Fe a)
{
    fe_canon(a);
    uint32_t w[14];
#pragma unroll
    for (int i = 0; i < 14; i++) w[i] = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int bit = 28 * k, j = bit >> 5, s = bit & 31;
        uint64_t v = (uint64_t)a.l[k] << s;
        w[j] |= (uint32_t)v;
        if (j + 1 < 14) w[j + 1] |= (uint32_t)(v >> 32);
    }
#pragma unroll
    for (int i = 0; i < 14; i++) {
        out[4 * i] = (uint8_t)w[i];
        out[4 * i + 1] = (uint8_t)(w[i] >> 8);
        out[4 * i + 2] = (uint8_t)(w[i] >> 16);
        out[4 * i + 3] = (uint8_t)(w[i] >> 24);
    }
}

CAPY_HD inline Fe fe_select(bool take_b, const Fe &a, const Fe &b)
{
    Fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.l[i] = take_b ? b.l[i] : a.l[i];
    return r;
}

// ------------------------------------------------------------------ group law, a = 1, d = -39081
CAPY_HD inline Pt pt_identity()
{
    Pt r;
    r.X = fe_zero();
    r.Y = fe_one();
    r.Z = fe_one();
    r.T = fe_zero();
    return r;
}

// d * t = -39081 t
CAPY_HD inline Fe fe_mul_d(const Fe &t) { return fe_neg(fe_mul_small(t, ED448_D_ABS)); }

// Unified, complete addition (add-2008-hwcd).  q is given in "cached" form: (X2, Y2, Z2, d*T2).
// Limb bounds: p.* are R (products); X2, Td2 <= 2^29 (possibly fe_neg_nr'ed table entries), Y2, Z2 are R.
CAPY_HD inline Pt pt_add_cached(const Pt &p, const Fe &X2, const Fe &Y2, const Fe &Z2, const Fe &Td2)
{
    Fe A = fe_mul(p.X, X2);                                   // R x 2^29
    Fe B = fe_mul(p.Y, Y2);
    Fe C = fe_mul(p.T, Td2);                                  // R x 2^29
    Fe D = fe_mul(p.Z, Z2);
    Fe E = fe_mul(fe_add_nr(p.X, p.Y), fe_add_nr(X2, Y2));    // 2^29 x 2^29.58
    E = fe_sub(fe_sub_nr(E, A), B);                           // reduced
    Fe F = fe_sub_nr(D, C);                                   // <= 2^29.58
    Fe G = fe_add_nr(D, C);                                   // <= 2^29
    Fe H = fe_sub_nr(B, A);                                   // <= 2^29.58
    Pt r;
    r.X = fe_mul(E, F);                                       // R x 2^29.58
    r.Y = fe_mul(G, H);                                       // 2^29 x 2^29.58 = 2^58.58
    r.Z = fe_mul(F, G);
    r.T = fe_mul(E, H);
    return r;
}

CAPY_HD inline Pt pt_add(const Pt &p, const Pt &q) { return pt_add_cached(p, q.X, q.Y, q.Z, fe_mul_d(q.T)); }

// Mixed addition with an affine precomputed point (x2, y2, d*x2*y2), Z2 = 1: 8 multiplications.
// Limb bounds as in pt_add_cached (x2, td2 <= 2^29; y2 is R).
CAPY_HD inline Pt pt_add_affine_cached(const Pt &p, const Fe &x2, const Fe &y2, const Fe &td2)
{
    Fe A = fe_mul(p.X, x2);
    Fe B = fe_mul(p.Y, y2);
    Fe C = fe_mul(p.T, td2);
    Fe E = fe_mul(fe_add_nr(p.X, p.Y), fe_add_nr(x2, y2));
    E = fe_sub(fe_sub_nr(E, A), B);
    Fe F = fe_sub_nr(p.Z, C);
    Fe G = fe_add_nr(p.Z, C);
    Fe H = fe_sub_nr(B, A);
    Pt r;
    r.X = fe_mul(E, F);
    r.Y = fe_mul(G, H);
    r.Z = fe_mul(F, G);
    r.T = fe_mul(E, H);
    return r;
}

// Doubling (dbl-2008-hwcd, a = 1): 4 squarings + 3 multiplications (4 with T).
// CAPY_ED448_DBL_XY=1 forms E = 2XY as ONE multiplication X * Y instead of (X + Y)^2 - X^2 - Y^2 (3S + 4M): a squaring is
// cheaper than a multiplication (237 against 310 VALU), but extracting the cross term from it costs a limb sum, two
// limb-wise subtractions and a carry pass (129), so that form has 5 % FEWER instructions in the doubling loop (2036
// against 2144) -- and 8 % MORE v_mad_u64_u32 (1104 against 1020), and runs at the same speed to within the noise of a
// box (27.3 against 27.5 M/s, profiles/r04_ed448_fe_trim.txt): these kernels are bound by the ENERGY of their multiply-adds
// (the clock follows the power limit), not by issue slots.  The default keeps the form with fewer multiply-adds.
// pt_dbl_core leaves E and H to the caller: T3 = E H is only needed by an addition that follows, so a run of
// doublings computes it once, after the last one (the optimiser used to sink that product out of the loop on its own;
// with the pinned multiply-add chains of fe_mul it no longer does).
// Bounds: A, B, C', XY are R (2^28 + 2^10); E, C, G <= 2^29 + 2^11; H <= 2^29.58; F reduced.
#ifndef CAPY_ED448_DBL_XY
#define CAPY_ED448_DBL_XY 0
#endif
CAPY_HD inline void pt_dbl_core(Pt &r, Fe &E, Fe &H, const Pt &p)
{
    Fe A = fe_sqr(p.X);
    Fe B = fe_sqr(p.Y);
    Fe C = fe_sqr(p.Z);
    C = fe_add_nr(C, C);                          // <= 2^29
#if CAPY_ED448_DBL_XY
    E = fe_mul(p.X, p.Y);
    E = fe_add_nr(E, E);                          // <= 2^29 + 2^11
#else
    E = fe_sqr(fe_add_nr(p.X, p.Y));              // sqr of 2^29: 40 * 2^58 < 2^64
    E = fe_sub(fe_sub_nr(E, A), B);               // reduced
#endif
    Fe G = fe_add_nr(A, B);                       // <= 2^29
    Fe F = fe_sub4(G, C);                         // reduced (C exceeds the 2p bias)
    H = fe_sub_nr(A, B);                          // <= 2^29.58
    r.Z = fe_mul(F, G);
    r.X = fe_mul(E, F);
    r.Y = fe_mul(G, H);                           // 2^29 x 2^29.58
}
template <bool WANT_T>
CAPY_HD inline Pt pt_dbl(const Pt &p)
{
    Pt r;
    Fe E, H;
    pt_dbl_core(r, E, H, p);
    if (WANT_T)
        r.T = fe_mul(E, H);                       // (2^29 + 2^11) x 2^29.58
    else
        r.T = p.T;
    return r;
}
// N doublings in a row, T from the last one only
#define CAPY_PRAGMA_DEV_(x) _Pragma(#x)
template <int N, int UNROLL = 1>
CAPY_HD inline Pt pt_dbl_n(Pt acc)
{
    Fe E, H;
    CAPY_PRAGMA_DEV_(unroll UNROLL)
    for (int j = 0; j < N; j++) {
        Pt r;
        pt_dbl_core(r, E, H, acc);
        acc.X = r.X;
        acc.Y = r.Y;
        acc.Z = r.Z;
    }
    acc.T = fe_mul(E, H);
    return acc;
}

CAPY_HD inline Pt pt_from_affine_bytes(const uint8_t *xy)
{
    Pt r;
    r.X = fe_from_bytes(xy);
    r.Y = fe_from_bytes(xy + 56);
    r.Z = fe_one();
    r.T = fe_mul(r.X, r.Y);
    return r;
}

CAPY_HD inline void pt_to_affine_bytes(uint8_t *xy, const Pt &p)
{
    Fe zi = fe_inv_out(p.Z);
    fe_to_bytes(xy, fe_mul(p.X, zi));
    fe_to_bytes(xy + 56, fe_mul(p.Y, zi));
}

CAPY_HD inline bool fe_is_zero(Fe a)
{
    fe_canon(a);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc |= a.l[i];
    return acc == 0;
}

// Two projective points to affine with ONE inversion (Montgomery's trick): 1/Z0 = Z1 / (Z0 Z1), 1/Z1 = Z0 / (Z0 Z1).
// The inversion is 446 squarings -- 7 % of a variable-base and 44 % of a fixed-base scalar multiplication.
// A zero Z (possible only for inputs that are not curve points) is replaced by 1 in the shared product, so it cannot
// spoil its partner; that item still gets (0, 0), exactly what pt_to_affine_bytes writes for it.
// Input validation for points that come from outside (public keys, Z of a ciphertext): both coordinates canonical
// (< p, i.e. re-encoding reproduces the input bytes) and x^2 + y^2 = 1 + d x^2 y^2.  The multiplication kernels do
// not check this themselves (the group law is only meaningful on the curve); callers that accept untrusted points
// run capy_ed448_validate_batch first.
CAPY_HD inline bool pt_validate_bytes(const uint8_t *xy)
{
    const Fe x = fe_from_bytes(xy), y = fe_from_bytes(xy + 56);
    uint8_t re[112];
    fe_to_bytes(re, x);
    fe_to_bytes(re + 56, y);
    uint32_t diff = 0;
    for (int i = 0; i < 112; i++) diff |= (uint32_t)(re[i] ^ xy[i]);
    const Fe xx = fe_sqr(x), yy = fe_sqr(y);
    const Fe lhs = fe_add(xx, yy), rhs = fe_add(fe_one(), fe_mul_d(fe_mul(xx, yy)));
    return diff == 0 && fe_is_zero(fe_sub(lhs, rhs));
}

CAPY_HD inline void pt_pair_to_affine_bytes(uint8_t *xy0, uint8_t *xy1, const Pt &p0, const Pt &p1)
{
    const bool z0_bad = fe_is_zero(p0.Z), z1_bad = fe_is_zero(p1.Z);
    const Fe z0 = fe_select(z0_bad, p0.Z, fe_one()), z1 = fe_select(z1_bad, p1.Z, fe_one());
    const Fe ti = fe_inv_out(fe_mul(z0, z1));
    const Fe zi0 = fe_select(z0_bad, fe_mul(ti, z1), fe_zero());
    const Fe zi1 = fe_select(z1_bad, fe_mul(ti, z0), fe_zero());
    fe_to_bytes(xy0, fe_mul(p0.X, zi0));
    fe_to_bytes(xy0 + 56, fe_mul(p0.Y, zi0));
    fe_to_bytes(xy1, fe_mul(p1.X, zi1));
    fe_to_bytes(xy1 + 56, fe_mul(p1.Y, zi1));
}

// ------------------------------------------------------------------ the fixed base on the 4-isogenous TWISTED curve (r03)
// E : x^2 + y^2 = 1 + d x^2 y^2 (a = +1, d = -39081) has no 7-multiplication mixed addition: its numerators
// y1 y2 - x1 x2 and x1 y2 + y1 x2 are a complex-number product (three multiplications at the least), so 8M with Z2 = 1.
// E': -x^2 + y^2 = 1 + (d - 1) x^2 y^2 (a = -1) has one (madd-2008-hwcd-3, table entries in the form
// (y - x, y + x, 2 d' x y)), and the two curves are 4-isogenous (Hamburg, "Decaf", section on isogenies):
//   phi    : E  -> E',  (x, y) -> ( 2xy / (y^2 - x^2),  (y^2 + x^2) / (2 - y^2 - x^2) )
//   phi^   : E' -> E,   (x, y) -> ( 2xy / (y^2 + x^2),  (y^2 - x^2) / (2 - y^2 + x^2) ),     phi^ o phi = [4]
// (checked numerically against the oracle's group law; tests/test_ed448_host.py).  For a generator G of the prime order
// r take G4 = [1/4 mod r] G: the table holds phi(j 2^(w i) G4), the digits are accumulated on E' with 7M additions, and
// phi^ of the sum is [4][k] G4 = [k] G -- the same affine point, so the same bytes as the kernels that stay on E.
// (A configured generator whose order is not r keeps the tables and additions on E.)
constexpr uint32_t ED448_TW_D_ABS = 39082;  // d' = d - 1 = -39082

// one entry of a twisted table from an affine point of E: phi, then (y' - x', y' + x', 2 d' x' y'), canonical limbs
CAPY_HD inline void pt_tw_niels_from_affine(Fe &ymx, Fe &ypx, Fe &td, const Fe &x, const Fe &y)
{
    const Fe xx = fe_sqr(x), yy = fe_sqr(y);
    const Fe dx = fe_sub(yy, xx);                                  // y^2 - x^2
    const Fe dy = fe_sub(fe_sub(fe_add(fe_one(), fe_one()), yy), xx);  // 2 - y^2 - x^2
    const Fe inv = fe_inv_out(fe_mul(dx, dy));
    const Fe xy = fe_mul(x, y);
    Fe xt = fe_mul(fe_add(xy, xy), fe_mul(inv, dy));               // 2xy / dx
    Fe yt = fe_mul(fe_add(yy, xx), fe_mul(inv, dx));               // (y^2 + x^2) / dy
    ymx = fe_sub(yt, xt);
    ypx = fe_add(yt, xt);
    const Fe t = fe_mul(xt, yt);
    td = fe_neg(fe_mul_small(fe_add(t, t), ED448_TW_D_ABS));       // 2 d' x' y'
    fe_canon(ymx);
    fe_canon(ypx);
    fe_canon(td);
}

// p + q on E', q given as (y2 - x2, y2 + x2, 2 d' x2 y2) with limbs <= 2^29 (table entries, the third possibly negated).
// Bounds: E and F are weak-reduced so that every product stays below 2^58.7 (fe_mul).
CAPY_HD inline Pt pt_madd_niels_tw(const Pt &p, const Fe &ymx2, const Fe &ypx2, const Fe &td2)
{
    const Fe A = fe_mul(fe_sub_nr(p.Y, p.X), ymx2);   // <= 2^29.6 x 2^28
    const Fe B = fe_mul(fe_add_nr(p.Y, p.X), ypx2);
    const Fe C = fe_mul(p.T, td2);                    // R x 2^29
    const Fe D = fe_add_nr(p.Z, p.Z);                 // <= 2^29.01
    const Fe E = fe_sub(B, A);                        // reduced
    const Fe F = fe_sub(D, C);                        // reduced
    const Fe G = fe_add_nr(D, C);                     // <= 2^29.6
    const Fe H = fe_add_nr(B, A);                     // <= 2^29.01
    Pt r;
    r.X = fe_mul(E, F);
    r.Y = fe_mul(G, H);                               // 2^29.6 x 2^29.01
    r.Z = fe_mul(F, G);
    r.T = fe_mul(E, H);
    return r;
}

// numerators and denominators of phi^ for a point of E' in extended coordinates: x = nx / dx, y = ny / dy on E
CAPY_HD inline void pt_tw_dual_fractions(const Pt &p, Fe &nx, Fe &dx, Fe &ny, Fe &dy)
{
    const Fe xx = fe_sqr(p.X), yy = fe_sqr(p.Y), zz = fe_sqr(p.Z);
    const Fe xy = fe_mul(p.X, p.Y);
    nx = fe_add(xy, xy);
    dx = fe_add(yy, xx);
    ny = fe_sub(yy, xx);
    dy = fe_add(fe_sub(fe_add(zz, zz), yy), xx);
}
CAPY_HD inline void pt_tw_to_affine_bytes(uint8_t *xy, const Pt &p)
{
    Fe nx, dx, ny, dy;
    pt_tw_dual_fractions(p, nx, dx, ny, dy);
    const Fe inv = fe_inv_out(fe_mul(dx, dy));
    fe_to_bytes(xy, fe_mul(nx, fe_mul(inv, dy)));
    fe_to_bytes(xy + 56, fe_mul(ny, fe_mul(inv, dx)));
}
// two of them with one inversion
CAPY_HD inline void pt_tw_pair_to_affine_bytes(uint8_t *xy0, uint8_t *xy1, const Pt &p0, const Pt &p1)
{
    Fe nx0, dx0, ny0, dy0, nx1, dx1, ny1, dy1;
    pt_tw_dual_fractions(p0, nx0, dx0, ny0, dy0);
    pt_tw_dual_fractions(p1, nx1, dx1, ny1, dy1);
    const Fe m0 = fe_mul(dx0, dy0), m1 = fe_mul(dx1, dy1);
    const Fe ti = fe_inv_out(fe_mul(m0, m1));
    const Fe i0 = fe_mul(ti, m1), i1 = fe_mul(ti, m0);  // 1 / (dx0 dy0), 1 / (dx1 dy1)
    fe_to_bytes(xy0, fe_mul(nx0, fe_mul(i0, dy0)));
    fe_to_bytes(xy0 + 56, fe_mul(ny0, fe_mul(i0, dx0)));
    fe_to_bytes(xy1, fe_mul(nx1, fe_mul(i1, dy1)));
    fe_to_bytes(xy1 + 56, fe_mul(ny1, fe_mul(i1, dx1)));
}

// ------------------------------------------------------------------ scalars
// 56 big-endian bytes -> 14 little-endian 32-bit words
CAPY_HD inline void sc_from_be(uint32_t w[14], const uint8_t *in)
{
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint8_t *b = in + 52 - 4 * i;
        w[i] = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
    }
}
CAPY_HD inline void sc_to_be(uint8_t *out, const uint32_t w[14])
{
#pragma unroll
    for (int i = 0; i < 14; i++) {
        uint8_t *b = out + 52 - 4 * i;
        b[0] = (uint8_t)(w[i] >> 24);
        b[1] = (uint8_t)(w[i] >> 16);
        b[2] = (uint8_t)(w[i] >> 8);
        b[3] = (uint8_t)w[i];
    }
}

// Signed fixed-window recoding, window width W (radix 2^W), NWIN windows cover the 448 scalar bits:
//     k = sum_{i<NWIN} (dig_i - HALF) 2^(W i) + top 2^(W NWIN),   dig_i = W-bit digits of k' = k + OFFSET,
// OFFSET = sum_i HALF 2^(W i), HALF = 2^(W-1), top = bit W*NWIN of k'.  Digits lie in [-HALF, HALF), so a
// table of {0..HALF} P plus a sign serves every window with the same control flow.
// Two widths are in use: WBITS for the per-item tables of the variable-base path (table build cost grows with
// 2^W), FB_WBITS for the shared fixed-base table (built once per device, so as wide as the caches comfortably hold).
#ifndef CAPY_ED448_WBITS
#define CAPY_ED448_WBITS 5
#endif
#ifndef CAPY_ED448_FB_WBITS
#define CAPY_ED448_FB_WBITS 12
#endif
template <int W>
struct Win {
    static_assert(W >= 2 && W <= 12, "window width");
    static constexpr int BITS = W;
    static constexpr int NWIN = (448 + W - 1) / W;
    static constexpr int HALF = 1 << (W - 1);
    static constexpr int ENTRIES = HALF + 1;
    static constexpr int TOP_BIT = W * NWIN;  // >= 448, < 480
    static_assert(TOP_BIT < 480, "the recoded scalar must fit 15 words");
};
constexpr int WBITS = CAPY_ED448_WBITS;
constexpr int FB_WBITS = CAPY_ED448_FB_WBITS;
static_assert(FB_WBITS >= WBITS, "row 0 of the fixed-base table also serves the variable-base digits (Straus)");
using VbWin = Win<WBITS>;
using FbWin = Win<FB_WBITS>;
constexpr int NWIN = VbWin::NWIN;
constexpr int WHALF = VbWin::HALF;
constexpr int TAB_ENTRIES = VbWin::ENTRIES;

template <int W>
CAPY_HD constexpr uint32_t sc_recode_offset_word(int j)
{
    uint32_t w = 0;
    for (int i = 0; i < Win<W>::NWIN; i++) {
        const int pos = W * i + W - 1;
        if (pos / 32 == j) w |= 1u << (pos % 32);
    }
    return w;
}

// in: 14 words of k (LE).  out: 15 words of k' (LE).  returns top.
template <int W>
CAPY_HD inline uint32_t sc_recode_signed(uint32_t kp[15], const uint32_t k[14])
{
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 15; i++) {
        const uint64_t v = (uint64_t)(i < 14 ? k[i] : 0u) + sc_recode_offset_word<W>(i) + c;
        kp[i] = (uint32_t)v;
        c = v >> 32;
    }
    return (kp[14] >> (Win<W>::TOP_BIT - 448)) & 1u;
}

// most-significant digit first: after sc_msb_align the current digit is the top W bits of kp[14]
CAPY_HD inline void sc_shl(uint32_t kp[15], int s)  // 1 <= s <= 32
{
    if (s == 32) {
#pragma unroll
        for (int t = 14; t > 0; t--) kp[t] = kp[t - 1];
        kp[0] = 0;
    } else {
#pragma unroll
        for (int t = 14; t > 0; t--) kp[t] = (kp[t] << s) | (kp[t - 1] >> (32 - s));
        kp[0] <<= s;
    }
}
template <int W>
CAPY_HD inline void sc_msb_align(uint32_t kp[15])
{
    sc_shl(kp, 480 - Win<W>::TOP_BIT);
}
template <int W>
CAPY_HD inline int sc_next_digit_msb(uint32_t kp[15])
{
    const int dig = (int)(kp[14] >> (32 - W)) - Win<W>::HALF;
    sc_shl(kp, W);
    return dig;
}
// least-significant digit first
template <int W>
CAPY_HD inline int sc_next_digit_lsb(uint32_t kp[15])
{
    const int dig = (int)(kp[0] & ((1u << W) - 1u)) - Win<W>::HALF;
#pragma unroll
    for (int t = 0; t < 14; t++) kp[t] = (kp[t] >> W) | (kp[t + 1] << (32 - W));
    kp[14] >>= W;
    return dig;
}

}  // namespace capy
