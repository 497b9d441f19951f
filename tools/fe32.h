// fe32.h — PROBE, not product: GF(2^448 - 2^224 - 1) multiplication and squaring on 14 SATURATED 32-bit limbs, written to
// price VERDICT r5 item 1 ("Ed448 field arithmetic at 14 x 32-bit limbs: 147 / 84 multiply-adds per multiplication /
// squaring instead of 192 / 108").  Real arithmetic (tools/microbench_fe32.hip checks every result against the library's
// 16 x 28-bit form), with the whole carry handling a saturated radix needs on gfx950:
//   * a 32 x 32 product fills 64 bits, so a column of up to 7 products needs a THIRD accumulator word: every
//     v_mad_u64_u32 is followed by a v_addc_co_u32 that collects its carry-out (one asm statement per column, so that the
//     compiler's hazard recogniser puts one s_nop behind a column, not behind every pair);
//   * Karatsuba over the Goldilocks split needs a0 + a1 and b0 + b1, which are 225-bit: the two carry bits select a masked
//     addend u = alpha * sb + beta * sa (+ alpha * beta * 2^224) for the upper half of CC;
//   * nothing can stay lazy: each of the three half products is normalised to limbs, and the recombination
//     (AA + BB) + (CC - AA) * phi and the wrap of the top word (2^448 = 2^224 + 1) are v_addc / v_subb chains over limbs
//     (one asm statement per chain: the carry lives in vcc).
// The host build (plain C) is the same algorithm step for step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace capy32 {

#define F32_HD __host__ __device__ __attribute__((always_inline)) inline

struct Fe32 {
    uint32_t l[14];
};
struct Acc {  // 96-bit column accumulator
    uint64_t lo;
    uint32_t hi;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define F32_PAIR(x, y) "v_mad_u64_u32 %0, vcc, %" #x ", %" #y ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define F32_OUT(t) "+v"((t).lo), "+v"((t).hi)
#endif

// t += sum_{i < N} x[i] * y[N - 1 - i]: one column of a schoolbook product, N = 1 .. 7
template <int N>
F32_HD void column(Acc &t, const uint32_t *x, const uint32_t *y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (N == 1)
        asm(F32_PAIR(2, 3) : F32_OUT(t) : "v"(x[0]), "v"(y[0]) : "vcc");
    else if constexpr (N == 2)
        asm(F32_PAIR(2, 3) F32_PAIR(4, 5) : F32_OUT(t) : "v"(x[0]), "v"(y[1]), "v"(x[1]), "v"(y[0]) : "vcc");
    else if constexpr (N == 3)
        asm(F32_PAIR(2, 3) F32_PAIR(4, 5) F32_PAIR(6, 7)
            : F32_OUT(t)
            : "v"(x[0]), "v"(y[2]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[0])
            : "vcc");
    else if constexpr (N == 4)
        asm(F32_PAIR(2, 3) F32_PAIR(4, 5) F32_PAIR(6, 7) F32_PAIR(8, 9)
            : F32_OUT(t)
            : "v"(x[0]), "v"(y[3]), "v"(x[1]), "v"(y[2]), "v"(x[2]), "v"(y[1]), "v"(x[3]), "v"(y[0])
            : "vcc");
    else if constexpr (N == 5)
        asm(F32_PAIR(2, 3) F32_PAIR(4, 5) F32_PAIR(6, 7) F32_PAIR(8, 9) F32_PAIR(10, 11)
            : F32_OUT(t)
            : "v"(x[0]), "v"(y[4]), "v"(x[1]), "v"(y[3]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[1]), "v"(x[4]), "v"(y[0])
            : "vcc");
    else if constexpr (N == 6)
        asm(F32_PAIR(2, 3) F32_PAIR(4, 5) F32_PAIR(6, 7) F32_PAIR(8, 9) F32_PAIR(10, 11) F32_PAIR(12, 13)
            : F32_OUT(t)
            : "v"(x[0]), "v"(y[5]), "v"(x[1]), "v"(y[4]), "v"(x[2]), "v"(y[3]), "v"(x[3]), "v"(y[2]), "v"(x[4]), "v"(y[1]),
              "v"(x[5]), "v"(y[0])
            : "vcc");
    else
        asm(F32_PAIR(2, 3) F32_PAIR(4, 5) F32_PAIR(6, 7) F32_PAIR(8, 9) F32_PAIR(10, 11) F32_PAIR(12, 13) F32_PAIR(14, 15)
            : F32_OUT(t)
            : "v"(x[0]), "v"(y[6]), "v"(x[1]), "v"(y[5]), "v"(x[2]), "v"(y[4]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[2]),
              "v"(x[5]), "v"(y[1]), "v"(x[6]), "v"(y[0])
            : "vcc");
#else
    for (int i = 0; i < N; i++) {
        const uint64_t p = (uint64_t)x[i] * y[N - 1 - i];
        const uint64_t s = t.lo + p;
        t.hi += s < p;
        t.lo = s;
    }
#endif
}

// the low limb leaves, the accumulator moves down one limb
F32_HD uint32_t next_limb(Acc &t)
{
    const uint32_t r = (uint32_t)t.lo;
    t.lo = (t.lo >> 32) | ((uint64_t)t.hi << 32);
    t.hi = 0;
    return r;
}

// out[0 .. 13] = x[0 .. 6] * y[0 .. 6], product scanning: 49 multiply-add / add-carry pairs
F32_HD void mul7(uint32_t *out, const uint32_t *x, const uint32_t *y)
{
    Acc t{0, 0};
    column<1>(t, x, y);
    out[0] = next_limb(t);
    column<2>(t, x, y);
    out[1] = next_limb(t);
    column<3>(t, x, y);
    out[2] = next_limb(t);
    column<4>(t, x, y);
    out[3] = next_limb(t);
    column<5>(t, x, y);
    out[4] = next_limb(t);
    column<6>(t, x, y);
    out[5] = next_limb(t);
    column<7>(t, x, y);
    out[6] = next_limb(t);
    column<6>(t, x + 1, y + 1);
    out[7] = next_limb(t);
    column<5>(t, x + 2, y + 2);
    out[8] = next_limb(t);
    column<4>(t, x + 3, y + 3);
    out[9] = next_limb(t);
    column<3>(t, x + 4, y + 4);
    out[10] = next_limb(t);
    column<2>(t, x + 5, y + 5);
    out[11] = next_limb(t);
    column<1>(t, x + 6, y + 6);
    out[12] = next_limb(t);
    out[13] = (uint32_t)t.lo;
}

// ---- limb chains (in place; the carry lives in vcc inside ONE asm statement) ----
// x[0 .. 6] += y[0 .. 6]; returns the carry (0 / 1)
F32_HD uint32_t add7(uint32_t *x, const uint32_t *y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t c;
    asm("v_add_co_u32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %10, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %11, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %4, %12, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %5, %13, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %14, vcc\n\t"
        "v_addc_co_u32 %7, vcc, 0, 0, vcc"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "=&v"(c)
        : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6])
        : "vcc");
    return c;
#else
    uint64_t c = 0;
    for (int i = 0; i < 7; i++) {
        c += (uint64_t)x[i] + y[i];
        x[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
#endif
}
// r[0 .. 6] = x + y (three-operand form for the operand sums); returns the carry
F32_HD uint32_t sum7(uint32_t *r, const uint32_t *x, const uint32_t *y)
{
#pragma unroll
    for (int i = 0; i < 7; i++) r[i] = x[i];
    return add7(r, y);
}
// x[0 .. 13] += y[0 .. 13]; returns the carry
F32_HD uint32_t add14(uint32_t *x, const uint32_t *y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t c;
    asm("v_add_co_u32 %0, vcc, %0, %15\n\t"
        "v_addc_co_u32 %1, vcc, %1, %16, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %17, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %18, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %4, %19, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %5, %20, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %21, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %22, vcc\n\t"
        "v_addc_co_u32 %8, vcc, %8, %23, vcc\n\t"
        "v_addc_co_u32 %9, vcc, %9, %24, vcc\n\t"
        "v_addc_co_u32 %10, vcc, %10, %25, vcc\n\t"
        "v_addc_co_u32 %11, vcc, %11, %26, vcc\n\t"
        "v_addc_co_u32 %12, vcc, %12, %27, vcc\n\t"
        "v_addc_co_u32 %13, vcc, %13, %28, vcc\n\t"
        "v_addc_co_u32 %14, vcc, 0, 0, vcc"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
          "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "=&v"(c)
        : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]),
          "v"(y[10]), "v"(y[11]), "v"(y[12]), "v"(y[13])
        : "vcc");
    return c;
#else
    uint64_t c = 0;
    for (int i = 0; i < 14; i++) {
        c += (uint64_t)x[i] + y[i];
        x[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
#endif
}
// x[0 .. 14] -= y[0 .. 13]  (the caller knows that x >= y)
F32_HD void sub15_14(uint32_t *x, const uint32_t *y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_sub_co_u32 %0, vcc, %0, %15\n\t"
        "v_subb_co_u32 %1, vcc, %1, %16, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %2, %17, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %3, %18, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %4, %19, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %5, %20, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %6, %21, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %7, %22, vcc\n\t"
        "v_subb_co_u32 %8, vcc, %8, %23, vcc\n\t"
        "v_subb_co_u32 %9, vcc, %9, %24, vcc\n\t"
        "v_subb_co_u32 %10, vcc, %10, %25, vcc\n\t"
        "v_subb_co_u32 %11, vcc, %11, %26, vcc\n\t"
        "v_subb_co_u32 %12, vcc, %12, %27, vcc\n\t"
        "v_subb_co_u32 %13, vcc, %13, %28, vcc\n\t"
        "v_subb_co_u32 %14, vcc, %14, 0, vcc"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
          "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14])
        : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]),
          "v"(y[10]), "v"(y[11]), "v"(y[12]), "v"(y[13])
        : "vcc");
#else
    int64_t c = 0;
    for (int i = 0; i < 15; i++) {
        c += (int64_t)x[i] - (i < 14 ? (int64_t)y[i] : 0);
        x[i] = (uint32_t)c;
        c >>= 32;
    }
#endif
}
// l[0] += t0 and l[7] += t7 in one pass over all 14 limbs (t0, t7 small); returns the carry out of limb 13
F32_HD uint32_t fold14(uint32_t *l, uint32_t t0, uint32_t t7)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t c;
    asm("v_add_co_u32 %0, vcc, %0, %15\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
        "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
        "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %16, vcc\n\t"
        "v_addc_co_u32 %8, vcc, 0, %8, vcc\n\t"
        "v_addc_co_u32 %9, vcc, 0, %9, vcc\n\t"
        "v_addc_co_u32 %10, vcc, 0, %10, vcc\n\t"
        "v_addc_co_u32 %11, vcc, 0, %11, vcc\n\t"
        "v_addc_co_u32 %12, vcc, 0, %12, vcc\n\t"
        "v_addc_co_u32 %13, vcc, 0, %13, vcc\n\t"
        "v_addc_co_u32 %14, vcc, 0, 0, vcc"
        : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(l[4]), "+v"(l[5]), "+v"(l[6]), "+v"(l[7]), "+v"(l[8]),
          "+v"(l[9]), "+v"(l[10]), "+v"(l[11]), "+v"(l[12]), "+v"(l[13]), "=&v"(c)
        : "v"(t0), "v"(t7)
        : "vcc");
    return c;
#else
    uint64_t c = t0;
    for (int i = 0; i < 14; i++) {
        c += (uint64_t)l[i] + (i == 7 ? t7 : 0);
        l[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
#endif
}
// the second wrap: g in {0, 1}; if g = 1 the value is < 16 phi, so the carry cannot leave limb 7
F32_HD void fold8(uint32_t *l, uint32_t g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_add_co_u32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
        "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
        "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %8, vcc"
        : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(l[4]), "+v"(l[5]), "+v"(l[6]), "+v"(l[7])
        : "v"(g)
        : "vcc");
#else
    uint64_t c = g;
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)l[i] + (i == 7 ? g : 0);
        l[i] = (uint32_t)c;
        c >>= 32;
    }
#endif
}

// (AA + BB) + (CC - AA) phi mod p, from the three normalised half products; AA, BB: 14 limbs; CC: 15 limbs (incl. the
// operand sums' carries); AA and CC are consumed.
F32_HD Fe32 recombine(uint32_t *A, const uint32_t *B, uint32_t *C)
{
    sub15_14(C, A);                     // Y = CC - AA >= 0, 15 limbs
    const uint32_t x14 = add14(A, B);   // X = AA + BB, 14 limbs + x14
    // r = [X_L + Y_H7] + [X_H7 + Y_L + Y_H7 + y14] phi + [x14 + y14] phi^2,  Y_H7 = Y[7 .. 13], y14 = Y[14]
    const uint32_t c1 = add7(A, C + 7);
    const uint32_t c2 = add7(A + 7, C);
    const uint32_t c3 = add7(A + 7, C + 7);
    const uint32_t t = x14 + C[14] + c2 + c3;  // weight phi^2 = phi + 1
    const uint32_t g = fold14(A, t, t + c1 + C[14]);
    fold8(A, g);
    Fe32 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = A[i];
    return r;
}

// r = a * b mod p: 147 multiply-adds
F32_HD Fe32 fe32_mul(const Fe32 &a, const Fe32 &b)
{
    uint32_t sa[7], sb[7], u[8];
    const uint32_t alpha = sum7(sa, a.l, a.l + 7), beta = sum7(sb, b.l, b.l + 7);
    // (sa + alpha phi)(sb + beta phi) = sa sb + (alpha sb + beta sa) phi + alpha beta phi^2
    const uint32_t ma = 0u - alpha, mb = 0u - beta;
    uint32_t v[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        u[i] = ma & sb[i];
        v[i] = mb & sa[i];
    }
    u[7] = add7(u, v) + (alpha & beta);
    uint32_t A[14], B[14], C[15];
    mul7(A, a.l, b.l);
    mul7(B, a.l + 7, b.l + 7);
    mul7(C, sa, sb);
    C[14] = u[7] + add7(C + 7, u);
    return recombine(A, B, C);
}

// out[0 .. 13] = x[0 .. 6]^2: the 21 off-diagonal products as a normalised number, doubled by limb shifts, plus the 7
// diagonal squares (each fills its own two limbs: no accumulation)
F32_HD void sqr7(uint32_t *out, const uint32_t *x)
{
    uint32_t o[14];
    Acc t{0, 0};
    o[0] = 0;
    // column k: pairs (i, k - i), max(0, k - 6) <= i < k - i
    column<1>(t, x, x + 1);  // k = 1: (0,1)
    o[1] = next_limb(t);
    column<1>(t, x, x + 2);  // k = 2: (0,2)
    o[2] = next_limb(t);
    column<2>(t, x, x + 2);  // k = 3: (0,3) (1,2)
    o[3] = next_limb(t);
    column<2>(t, x, x + 3);  // k = 4: (0,4) (1,3)
    o[4] = next_limb(t);
    column<3>(t, x, x + 3);  // k = 5: (0,5) (1,4) (2,3)
    o[5] = next_limb(t);
    column<3>(t, x, x + 4);  // k = 6: (0,6) (1,5) (2,4)
    o[6] = next_limb(t);
    column<3>(t, x + 1, x + 4);  // k = 7: (1,6) (2,5) (3,4)
    o[7] = next_limb(t);
    column<2>(t, x + 2, x + 5);  // k = 8: (2,6) (3,5)
    o[8] = next_limb(t);
    column<2>(t, x + 3, x + 5);  // k = 9: (3,6) (4,5)
    o[9] = next_limb(t);
    column<1>(t, x + 4, x + 6);  // k = 10: (4,6)
    o[10] = next_limb(t);
    column<1>(t, x + 5, x + 6);  // k = 11: (5,6)
    o[11] = next_limb(t);
    o[12] = next_limb(t);
    o[13] = (uint32_t)t.lo;
    // 2 * off-diagonal (< 2^448)
#pragma unroll
    for (int i = 13; i > 0; i--) o[i] = (o[i] << 1) | (o[i - 1] >> 31);
    o[0] = 0;  // o[0] was 0
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const uint64_t d = (uint64_t)x[i] * x[i];
        out[2 * i] = (uint32_t)d;
        out[2 * i + 1] = (uint32_t)(d >> 32);
    }
    add14(out, o);  // the square fits 14 limbs
}

// r = a^2 mod p: 84 multiplies
F32_HD Fe32 fe32_sqr(const Fe32 &a)
{
    uint32_t sa[7], u[8];
    const uint32_t alpha = sum7(sa, a.l, a.l + 7);
    // (sa + alpha phi)^2 = sa^2 + 2 alpha sa phi + alpha phi^2
    const uint32_t ma = 0u - alpha;
    u[0] = (sa[0] << 1) & ma;
#pragma unroll
    for (int i = 1; i < 7; i++) u[i] = ((sa[i] << 1) | (sa[i - 1] >> 31)) & ma;
    u[7] = ((sa[6] >> 31) & ma) + alpha;
    uint32_t A[14], B[14], C[15];
    sqr7(A, a.l);
    sqr7(B, a.l + 7);
    sqr7(C, sa);
    C[14] = u[7] + add7(C + 7, u);
    return recombine(A, B, C);
}

// 56 little-endian bytes <-> limbs; the value of a Fe32 is any representative below 2^448
F32_HD Fe32 fe32_from_bytes(const uint8_t *in)
{
    Fe32 r;
#pragma unroll
    for (int i = 0; i < 14; i++)
        r.l[i] = (uint32_t)in[4 * i] | ((uint32_t)in[4 * i + 1] << 8) | ((uint32_t)in[4 * i + 2] << 16) | ((uint32_t)in[4 * i + 3] << 24);
    return r;
}
// canonical representative: subtract p once if the value is >= p  (value + 2^224 + 1 >= 2^448)
F32_HD void fe32_to_bytes(uint8_t *out, Fe32 a)
{
    uint32_t t[14];
    uint64_t c = 1;
    for (int i = 0; i < 14; i++) {
        c += (uint64_t)a.l[i] + (i == 7 ? 1u : 0u);
        t[i] = (uint32_t)c;
        c >>= 32;
    }
    for (int i = 0; i < 14; i++) {
        const uint32_t w = c ? t[i] : a.l[i];
        out[4 * i] = (uint8_t)w;
        out[4 * i + 1] = (uint8_t)(w >> 8);
        out[4 * i + 2] = (uint8_t)(w >> 16);
        out[4 * i + 3] = (uint8_t)(w >> 24);
    }
}

}  // namespace capy32
