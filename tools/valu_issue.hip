// valu_issue.hip — ONE instrument for the VALU issue question (profiles/r03_valu_issue_bisect.txt).
//
// Every stream below -- the library's real keccak rounds (one-lane, two-lane, rolled, unrolled), the real Ed448 field
// multiplication, and synthetic asm streams of the same opcode mixes -- runs in the SAME harness at W = 1, 2, 4, 8 waves
// per SIMD, and is meant to be run under the SAME rocprofv3 --pmc pass (tools/valu_issue.sh):
//   cycles per VALU instruction per SIMD = (GRBM_GUI_ACTIVE / 8) / (SQ_INSTS_VALU / 1024)       [PMC only]
//   clock                               = (GRBM_GUI_ACTIVE / 8) / dispatch duration            [PMC only]
// In-kernel cross-checks written per wave: s_memtime (shader clock), s_memrealtime (100 MHz), HW_ID and XCC_ID, from
// which the host proves the occupancy claim: every SIMD that ran waves held exactly W of them, and their lifetimes
// overlapped (r02 compared s_memtime ticks of possibly non-co-resident waves with wall clock of the whole launch).
// DATA = 0 runs the same instructions on all-zero registers (the keccak rounds get their round constants masked to
// zero): a power limit shows as zero-data running faster at a higher clock, an issue limit as no difference.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I capycrypt_amd/csrc -o tools/valu_issue tools/valu_issue.hip
// Run:   tools/valu_issue [filter-substring]        (plain timings + occupancy proof)
//        rocprofv3 --pmc ... -- tools/valu_issue    (counters; tools/summarize_valu_issue.py joins them by kernel + grid)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include "sponge_kernels_k2.h"
#include "ed448_dev.h"
using namespace capy;

struct Rec {
    unsigned long long t0, t1, r0, r1;
    unsigned hwid, xcc, pad0, pad1;
};

#define STAMP_BEGIN()                                                                                              \
    unsigned long long t0, t1, r0, r1;                                                                             \
    unsigned hwid, xcc;                                                                                            \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc)); \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)"      \
                 : "=s"(r0), "=s"(t0)::"memory");
#define STAMP_END()                                                                                                \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");       \
    if ((threadIdx.x & 63) == 0) {                                                                                 \
        Rec &o = rec[blockIdx.x * 4 + threadIdx.x / 64];                                                           \
        o.t0 = t0; o.t1 = t1; o.r0 = r0; o.r1 = r1; o.hwid = hwid; o.xcc = xcc;                                    \
    }

// ---------------------------------------------------------------- real streams (compiler-scheduled library code)
enum { K1_UNROLLED = 0, K1_ROLLED = 1, K2_UNROLLED = 2, K2_ROLLED = 3, FE_MUL = 4, FE_SQR = 5, K1_NOALIGN = 6,
       K1_THETA = 7, K1_RHO = 8, K1_CHI = 9, K1_BLOCKED = 10, K1_PAIRED = 11, K1_PAIRED_UNROLLED = 12 };

template <int R>
__device__ __forceinline__ void k1_round_masked(KState &a, uint32_t m)
{
    keccak_round_aligned(a, (uint32_t)keccak_rc64(R) & m, (uint32_t)(keccak_rc64(R) >> 32) & m);
}
template <int... Rs>
__device__ __forceinline__ void k1_perm_masked(KState &a, uint32_t m, std::integer_sequence<int, Rs...>)
{
    (k1_round_masked<Rs>(a, m), ...);
}

// the one-lane round with every rotation replaced by a bitop3 of the same two inputs: same dependency graph and
// register pressure, no v_alignbit_b32 (results are not keccak)
__device__ __forceinline__ void k1_round_noalign(KState &a, uint32_t rc_lo, uint32_t rc_hi)
{
    uint32_t cl[5], ch[5], rl[5], rh[5];
#pragma unroll
    for (int x = 0; x < 5; x++) {
        cl[x] = xor3(xor3(a.lo[x], a.lo[x + 5], a.lo[x + 10]), a.lo[x + 15], a.lo[x + 20]);
        ch[x] = xor3(xor3(a.hi[x], a.hi[x + 5], a.hi[x + 10]), a.hi[x + 15], a.hi[x + 20]);
    }
#pragma unroll
    for (int x = 0; x < 5; x++) {
        rl[x] = __builtin_amdgcn_bitop3_b32(cl[x], ch[x], rc_lo, 0x1E);
        rh[x] = __builtin_amdgcn_bitop3_b32(ch[x], cl[x], rc_hi, 0x2D);
    }
    KState e, b;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        e.lo[i] = xor3(a.lo[i], cl[(i % 5 + 4) % 5], rl[(i % 5 + 1) % 5]);
        e.hi[i] = xor3(a.hi[i], ch[(i % 5 + 4) % 5], rh[(i % 5 + 1) % 5]);
    }
#pragma unroll
    for (int i = 0; i < 25; i++) {
        const int x = i % 5, y = i / 5, dst = y + 5 * ((2 * x + 3 * y) % 5);
        if (i == 0) {
            b.lo[dst] = e.lo[i];
            b.hi[dst] = e.hi[i];
        } else {
            b.lo[dst] = __builtin_amdgcn_bitop3_b32(e.lo[i], e.hi[i], rc_lo, 0x1E);
            b.hi[dst] = __builtin_amdgcn_bitop3_b32(e.hi[i], e.lo[i], rc_hi, 0x2D);
        }
    }
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) {
            a.lo[x + 5 * y] = chi3(b.lo[x + 5 * y], b.lo[(x + 1) % 5 + 5 * y], b.lo[(x + 2) % 5 + 5 * y]);
            a.hi[x + 5 * y] = chi3(b.hi[x + 5 * y], b.hi[(x + 1) % 5 + 5 * y], b.hi[(x + 2) % 5 + 5 * y]);
        }
    a.lo[0] ^= rc_lo;
    a.hi[0] ^= rc_hi;
    asm volatile(".p2align 3" : "+v"(a.lo[0]), "+v"(a.hi[0]));
}

// theta / rho / chi alone, repeated so that one call is about one round's worth of instructions
__device__ __forceinline__ void k1_theta_only(KState &a)
{
    uint32_t cl[5], ch[5], rl[5], rh[5];
#pragma unroll
    for (int x = 0; x < 5; x++) {
        cl[x] = xor3(xor3(a.lo[x], a.lo[x + 5], a.lo[x + 10]), a.lo[x + 15], a.lo[x + 20]);
        ch[x] = xor3(xor3(a.hi[x], a.hi[x + 5], a.hi[x + 10]), a.hi[x + 15], a.hi[x + 20]);
    }
#pragma unroll
    for (int x = 0; x < 5; x++) rol64c<1>(cl[x], ch[x], rl[x], rh[x]);
#pragma unroll
    for (int i = 0; i < 25; i++) {
        a.lo[i] = xor3(a.lo[i], cl[(i % 5 + 4) % 5], rl[(i % 5 + 1) % 5]);
        a.hi[i] = xor3(a.hi[i], ch[(i % 5 + 4) % 5], rh[(i % 5 + 1) % 5]);
    }
    asm volatile(".p2align 3" : "+v"(a.lo[0]), "+v"(a.hi[0]));
}
__device__ __forceinline__ void k1_rho_only(KState &a)
{
    KState b;
    rho_pi_all(a, b, std::make_integer_sequence<int, 25>{});
    a = b;
    asm volatile(".p2align 3" : "+v"(a.lo[0]), "+v"(a.hi[0]));
}
__device__ __forceinline__ void k1_chi_only(KState &a)
{
    KState b = a;
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) {
            a.lo[x + 5 * y] = chi3(b.lo[x + 5 * y], b.lo[(x + 1) % 5 + 5 * y], b.lo[(x + 2) % 5 + 5 * y]);
            a.hi[x + 5 * y] = chi3(b.hi[x + 5 * y], b.hi[(x + 1) % 5 + 5 * y], b.hi[(x + 2) % 5 + 5 * y]);
        }
    asm volatile(".p2align 3" : "+v"(a.lo[0]), "+v"(a.hi[0]));
}

template <int STREAM, int DATA>
__global__ __launch_bounds__(256) void vi_real(Rec *rec, uint32_t iters, uint32_t m, uint64_t *sink)
{
    const uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t x = 0;
    if constexpr (STREAM == K1_UNROLLED || STREAM == K1_ROLLED || STREAM >= K1_NOALIGN) {
        KState a;
#pragma unroll
        for (int i = 0; i < 25; i++) {
            a.lo[i] = (uint32_t)(id * 25 + i + 1) * 0x85EBCA6Bu & m;
            a.hi[i] = (uint32_t)((id * 25 + i + 1) * 0x9E3779B9u) & m;
        }
        STAMP_BEGIN();
        for (uint32_t it = 0; it < iters; it++) {
            if constexpr (STREAM == K1_UNROLLED)
                k1_perm_masked(a, m, std::make_integer_sequence<int, 24>{});
            else if constexpr (STREAM == K1_ROLLED) {
                // keccakf1600_pipelined with masked constants
                uint32_t c0 = KECCAK_RC32[0] & m, c1 = KECCAK_RC32[1] & m, c2 = KECCAK_RC32[2] & m, c3 = KECCAK_RC32[3] & m;
#pragma unroll 1
                for (int r = 0; r < 24; r += 2) {
                    const int nx = (r + 2 < 24) ? r + 2 : 0;
                    const uint32_t n0 = KECCAK_RC32[2 * nx] & m, n1 = KECCAK_RC32[2 * nx + 1] & m,
                                   n2 = KECCAK_RC32[2 * nx + 2] & m, n3 = KECCAK_RC32[2 * nx + 3] & m;
                    keccak_round(a, c0, c1);
                    keccak_round(a, c2, c3);
                    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
                }
            } else if constexpr (STREAM == K1_BLOCKED) {
                keccakf1600_paired<false>(a);
            } else if constexpr (STREAM == K1_PAIRED) {
                keccakf1600_paired<true>(a);
            } else if constexpr (STREAM == K1_PAIRED_UNROLLED) {
#pragma unroll
                for (int r = 0; r < 24; r++)  // literal round constants: an SGPR operand would make the iota XOR a 4-cycle op
                    keccak_round_blocked<true>(a, (uint32_t)keccak_rc64(r), (uint32_t)(keccak_rc64(r) >> 32));
            } else if constexpr (STREAM == K1_NOALIGN) {
#pragma unroll
                for (int r = 0; r < 24; r++) k1_round_noalign(a, (0x1234567u + r) & m, (0x89abcdeu + r) & m);
            } else if constexpr (STREAM == K1_THETA) {
#pragma unroll
                for (int r = 0; r < 48; r++) k1_theta_only(a);
            } else if constexpr (STREAM == K1_RHO) {
#pragma unroll
                for (int r = 0; r < 96; r++) k1_rho_only(a);
            } else {
#pragma unroll
                for (int r = 0; r < 96; r++) k1_chi_only(a);
            }
        }
        STAMP_END();
#pragma unroll
        for (int i = 0; i < 25; i++) x ^= a.lo[i] ^ a.hi[i];
    } else if constexpr (STREAM == K2_UNROLLED || STREAM == K2_ROLLED) {
        KHalf a;
        const uint32_t hmask = 0u - (threadIdx.x & 1);
#pragma unroll
        for (int i = 0; i < 25; i++) a.a[i] = (uint32_t)((id * 25 + i + 1) * 0x9E3779B9u) & m;
        STAMP_BEGIN();
        for (uint32_t it = 0; it < iters; it++) {
            if constexpr (STREAM == K2_UNROLLED)
                keccakf1600_k2_unrolled(a, hmask);
            else
                keccakf1600_k2_pipelined(a, hmask);
        }
        STAMP_END();
#pragma unroll
        for (int i = 0; i < 25; i++) x ^= a.a[i];
    } else {
        Fe a, b;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            a.l[i] = ((uint32_t)((id * 16 + i + 1) * 0x9E3779B9u) & M28) & m;
            b.l[i] = ((uint32_t)((id * 16 + i + 7) * 0x85EBCA6Bu) & M28) & m;
        }
        STAMP_BEGIN();
#pragma unroll 1
        for (uint32_t it = 0; it < iters; it++) {
            if constexpr (STREAM == FE_MUL) {
                a = fe_mul(a, b);
                b = fe_mul(b, a);
            } else {
                a = fe_sqr(a);
                a = fe_sqr(a);
            }
        }
        STAMP_END();
#pragma unroll
        for (int i = 0; i < 16; i++) x ^= a.l[i] ^ b.l[i];
    }
    if (x == 0x12345678u && m == 0x5a5a5a5au) atomicXor((unsigned long long *)sink, (unsigned long long)x);
}

// ---------------------------------------------------------------- synthetic streams on hard-coded registers v8..v63
#define CLOB                                                                                                      \
    "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
        "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38",  \
        "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",  \
        "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "vcc"

#define BITOP3(d, a, b, c) "v_bitop3_b32 v" #d ", v" #a ", v" #b ", v" #c " bitop3:0x96\n\t"
#define ALIGN(d, a, b, c) "v_alignbit_b32 v" #d ", v" #a ", v" #b ", 7\n\t"
#define XOR2(d, a, b, c) "v_xor_b32 v" #d ", v" #a ", v" #b "\n\t"
#define ADD3(d, a, b, c) "v_add3_u32 v" #d ", v" #a ", v" #b ", v" #c "\n\t"
#define DPPM(d, a, b, c) "v_mov_b32_dpp v" #d ", v" #a " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// 64-bit accumulators live in even-aligned pairs v48:49 .. v62:63 (d is the low register of the pair)
#define MAD64(d, a, b, c) "v_mad_u64_u32 v[" #d ":" #c "], vcc, v" #a ", v" #b ", v[" #d ":" #c "]\n\t"
// 64-bit shift / 64-bit add / carry pair on the accumulator pairs (the carry chain of fe_from_columns)
#define SHR64(d, a, b, c) "v_lshrrev_b64 v[" #d ":" #c "], 28, v[" #d ":" #c "]\n\t"
#define ADD64(d, a, b, c) "v_lshl_add_u64 v[" #d ":" #c "], v[" #d ":" #c "], 0, v[12:13]\n\t"
#define SUBP(d, a, b, c) "v_sub_co_u32 v" #d ", vcc, v" #d ", v" #a "\n\tv_subb_co_u32 v" #c ", vcc, v" #c ", v" #b ", vcc\n\t"
#define MULLO(d, a, b, c) "v_mul_lo_u32 v" #d ", v" #a ", v" #b "\n\t"

// like a keccak round: 16 destinations, sources wander over v8..v33, three different banks
#define PAT_A(I) I(48, 8, 13, 18) I(49, 9, 14, 19) I(50, 10, 15, 20) I(51, 11, 16, 21) I(52, 12, 17, 22) I(53, 13, 18, 23) I(54, 14, 19, 24) I(55, 15, 20, 25)
#define PAT_B(I) I(56, 16, 21, 26) I(57, 17, 22, 27) I(58, 18, 23, 28) I(59, 19, 24, 29) I(60, 20, 25, 30) I(61, 21, 26, 31) I(62, 22, 27, 32) I(63, 23, 28, 33)
// feedback variants: destinations are later read as sources (v34..v47 <- v48.., and back), so values keep changing
#define PAT_C(I) I(34, 48, 53, 58) I(35, 49, 54, 59) I(36, 50, 55, 60) I(37, 51, 56, 61) I(38, 52, 57, 62) I(39, 53, 58, 63) I(40, 54, 59, 48) I(41, 55, 60, 49)
#define PAT_D(I) I(8, 34, 39, 44) I(9, 35, 40, 45) I(10, 36, 41, 46) I(11, 37, 42, 47) I(12, 38, 43, 34) I(13, 39, 44, 35) I(14, 40, 45, 36) I(15, 41, 46, 37)
// 8 MADs into 8 accumulator pairs
#define PAT_M(I) I(48, 8, 13, 49) I(50, 9, 14, 51) I(52, 10, 15, 53) I(54, 11, 16, 55) I(56, 12, 17, 57) I(58, 13, 18, 59) I(60, 14, 19, 61) I(62, 15, 20, 63)
// 8 simple ops that do not touch the accumulators
#define PAT_S(I) I(34, 21, 26, 31) I(35, 22, 27, 32) I(36, 23, 28, 33) I(37, 24, 29, 21) I(38, 25, 30, 22) I(39, 26, 31, 23) I(40, 27, 32, 24) I(41, 28, 33, 25)

#define X4(P) P P P P
#define X2(P) P P

enum { S_XOR = 0, S_BITOP3 = 1, S_ALIGN = 2, S_DPP = 3, S_K1MIX = 4, S_K2MIX = 5, S_MAD = 6, S_MADMIX = 7, S_ADD3 = 8,
       S_K1MIX_FB = 9, S_SHR64 = 10, S_ADD64 = 11, S_SUBP = 12, S_MULLO = 13 };

template <int S>
__device__ __forceinline__ void syn_body()
{
    // 128 instructions per call; every form 8-byte aligned at entry
    if constexpr (S == S_XOR)
        asm volatile(".p2align 3\n\t" X4(X2(PAT_A(XOR2) PAT_B(XOR2))) ::: CLOB);
    else if constexpr (S == S_BITOP3)
        asm volatile(".p2align 3\n\t" X4(X2(PAT_A(BITOP3) PAT_B(BITOP3))) ::: CLOB);
    else if constexpr (S == S_ALIGN)
        asm volatile(".p2align 3\n\t" X4(X2(PAT_A(ALIGN) PAT_B(ALIGN))) ::: CLOB);
    else if constexpr (S == S_DPP)
        asm volatile(".p2align 3\n\t" X4(X2(PAT_A(DPPM) PAT_B(DPPM))) ::: CLOB);
    else if constexpr (S == S_ADD3)
        asm volatile(".p2align 3\n\t" X4(X2(PAT_A(ADD3) PAT_B(ADD3))) ::: CLOB);
    else if constexpr (S == S_K1MIX)  // 2 bitop3 : 1 alignbit (the one-lane round is 120 : 58 : 2)
        asm volatile(".p2align 3\n\t" X4(PAT_A(BITOP3) PAT_B(ALIGN) PAT_C(BITOP3)) X2(PAT_A(BITOP3) PAT_B(ALIGN)) ::: CLOB);
    else if constexpr (S == S_K1MIX_FB)  // the same with results fed back into the sources (values change every trip)
        asm volatile(".p2align 3\n\t" X4(PAT_A(BITOP3) PAT_C(ALIGN) PAT_D(BITOP3)) X2(PAT_B(BITOP3) PAT_C(ALIGN)) ::: CLOB);
    else if constexpr (S == S_K2MIX)  // 2 bitop3 : 1 alignbit : 1 dpp
        asm volatile(".p2align 3\n\t" X4(PAT_A(BITOP3) PAT_B(ALIGN) PAT_A(BITOP3) PAT_B(DPPM)) ::: CLOB);
    else if constexpr (S == S_MAD)
        asm volatile(".p2align 3\n\t" X4(X4(PAT_M(MAD64))) ::: CLOB);
    else if constexpr (S == S_SHR64)
        asm volatile(".p2align 3\n\t" X4(X4(PAT_M(SHR64))) ::: CLOB);
    else if constexpr (S == S_ADD64)
        asm volatile(".p2align 3\n\t" X4(X4(PAT_M(ADD64))) ::: CLOB);
    else if constexpr (S == S_SUBP)  // 64 pairs = 128 instructions
        asm volatile(".p2align 3\n\t" X4(X2(PAT_M(SUBP))) ::: CLOB);
    else if constexpr (S == S_MULLO)
        asm volatile(".p2align 3\n\t" X4(X2(PAT_A(MULLO) PAT_B(MULLO))) ::: CLOB);
    else  // the Ed448 multiplication's mix: 58 % v_mad_u64_u32, the rest simple ops
        asm volatile(".p2align 3\n\t" X4(PAT_M(MAD64) PAT_S(ADD3) PAT_M(MAD64) PAT_S(ALIGN)) PAT_M(MAD64) PAT_M(MAD64) ::: CLOB);
}
template <int S>
constexpr int syn_count()
{
    return S == S_MADMIX ? 144 : (S == S_K1MIX || S == S_K1MIX_FB) ? 128 : 128;
}

#define INITR(r, p) "v_mul_lo_u32 v" #r ", v" #p ", %1\n\t"
template <int S, int DATA>
__global__ __launch_bounds__(256) void vi_syn(Rec *rec, uint32_t iters, uint32_t m, uint64_t *sink)
{
    const uint32_t seed = ((blockIdx.x * 256 + threadIdx.x) * 2654435761u | 1u) & m;
    const uint32_t k = 0x9E3779B1u;
    asm volatile("v_mov_b32 v8, %0\n\t" INITR(9, 8) INITR(10, 9) INITR(11, 10) INITR(12, 11) INITR(13, 12) INITR(14, 13)
                 INITR(15, 14) INITR(16, 15) INITR(17, 16) INITR(18, 17) INITR(19, 18) INITR(20, 19) INITR(21, 20)
                 INITR(22, 21) INITR(23, 22) INITR(24, 23) INITR(25, 24) INITR(26, 25) INITR(27, 26) INITR(28, 27)
                 INITR(29, 28) INITR(30, 29) INITR(31, 30) INITR(32, 31) INITR(33, 32) INITR(34, 33) INITR(35, 34)
                 INITR(36, 35) INITR(37, 36) INITR(38, 37) INITR(39, 38) INITR(40, 39) INITR(41, 40) INITR(42, 41)
                 INITR(43, 42) INITR(44, 43) INITR(45, 44) INITR(46, 45) INITR(47, 46) INITR(48, 47) INITR(49, 48)
                 INITR(50, 49) INITR(51, 50) INITR(52, 51) INITR(53, 52) INITR(54, 53) INITR(55, 54) INITR(56, 55)
                 INITR(57, 56) INITR(58, 57) INITR(59, 58) INITR(60, 59) INITR(61, 60) INITR(62, 61) INITR(63, 62)
                 :
                 : "v"(seed), "s"(k)
                 : CLOB);
    STAMP_BEGIN();
#pragma unroll 1
    for (uint32_t it = 0; it < iters; it++) {
        syn_body<S>();
        syn_body<S>();
        syn_body<S>();
        syn_body<S>();
    }
    STAMP_END();
    uint32_t x;
    asm volatile("v_xor_b32 %0, v48, v63\n\tv_xor_b32 %0, %0, v34" : "=v"(x)::CLOB);
    if (x == 0x12345678u && m == 0x5a5a5a5au) atomicXor((unsigned long long *)sink, (unsigned long long)x);
}

// ---------------------------------------------------------------- host
typedef void (*kfn)(Rec *, uint32_t, uint32_t, uint64_t *);
struct Ent {
    const char *name;
    kfn f;
    int data;          // 1 random, 0 zero
    double insts;      // VALU instructions per iteration per wave if known analytically, else 0 (PMC tells)
    uint32_t iters;
    double units;      // work units per wave per iteration (permutations x sponges, field multiplications x lanes)
    const char *unit;
};

static void analyse(const std::vector<Rec> &h, int W, double &ghz, int &simds, int &minw, int &maxw, double &overlap)
{
    std::map<unsigned, std::vector<const Rec *>> by;
    double tick = 0, real = 0;
    for (auto &r : h) {
        // HW_ID (gfx9): simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
        const unsigned key = (r.xcc & 0xf) << 16 | (r.hwid >> 4 & 3) | (r.hwid >> 8 & 0xf) << 2 | (r.hwid >> 12 & 0xf) << 6;
        by[key].push_back(&r);
        tick += (double)(r.t1 - r.t0);
        real += (double)(r.r1 - r.r0);
    }
    ghz = tick / (real * 10.0);  // s_memrealtime = 100 MHz -> 10 ns per tick
    simds = (int)by.size();
    minw = 1 << 30;
    maxw = 0;
    double ov = 0;
    for (auto &kv : by) {
        const int n = (int)kv.second.size();
        minw = std::min(minw, n);
        maxw = std::max(maxw, n);
        unsigned long long s = 0, e = ~0ull, s0 = ~0ull, e1 = 0;
        for (auto *r : kv.second) {
            s = std::max(s, r->r0);
            e = std::min(e, r->r1);
            s0 = std::min(s0, r->r0);
            e1 = std::max(e1, r->r1);
        }
        ov += e > s ? (double)(e - s) / (double)(e1 - s0) : 0.0;
    }
    overlap = ov / by.size();
}

int main(int argc, char **argv)
{
    const char *filter = argc > 1 ? argv[1] : "";
    const int wmax_arg = argc > 2 ? atoi(argv[2]) : 8;
#define REAL(S, D) vi_real<S, D>
#define SYN(S, D) vi_syn<S, D>
    std::vector<Ent> ents = {
        {"k1 unrolled rand", REAL(K1_UNROLLED, 1), 1, 0, 1500, 64, "perm"},
        {"k1 unrolled zero", REAL(K1_UNROLLED, 0), 0, 0, 1500, 64, "perm"},
        {"k1 rolled rand", REAL(K1_ROLLED, 1), 1, 0, 1500, 64, "perm"},
        {"k1 rolled zero", REAL(K1_ROLLED, 0), 0, 0, 1500, 64, "perm"},
        {"k1 blocked rolled rand", REAL(K1_BLOCKED, 1), 1, 0, 1500, 64, "perm"},
        {"k1 paired rolled rand", REAL(K1_PAIRED, 1), 1, 0, 1500, 64, "perm"},
        {"k1 paired rolled zero", REAL(K1_PAIRED, 0), 0, 0, 1500, 64, "perm"},
        {"k1 paired unrolled rand", REAL(K1_PAIRED_UNROLLED, 1), 1, 0, 1500, 64, "perm"},
        {"k2 unrolled rand", REAL(K2_UNROLLED, 1), 1, 0, 2000, 32, "perm"},
        {"k2 rolled rand", REAL(K2_ROLLED, 1), 1, 0, 2000, 32, "perm"},
        {"k1 noalign rand", REAL(K1_NOALIGN, 1), 1, 0, 1500, 64, "perm"},
        {"k1 theta-only rand", REAL(K1_THETA, 1), 1, 0, 1500, 64, "perm-eq"},
        {"k1 rho-only rand", REAL(K1_RHO, 1), 1, 0, 1500, 64, "perm-eq"},
        {"k1 chi-only rand", REAL(K1_CHI, 1), 1, 0, 1500, 64, "perm-eq"},
        {"fe_mul x2 rand", REAL(FE_MUL, 1), 1, 0, 10000, 128, "fmul"},
        {"fe_mul x2 zero", REAL(FE_MUL, 0), 0, 0, 10000, 128, "fmul"},
        {"fe_sqr x2 rand", REAL(FE_SQR, 1), 1, 0, 16000, 128, "fsqr"},
        {"syn xor rand", SYN(S_XOR, 1), 1, 512, 12000, 0, ""},
        {"syn xor zero", SYN(S_XOR, 0), 0, 512, 12000, 0, ""},
        {"syn bitop3 rand", SYN(S_BITOP3, 1), 1, 512, 12000, 0, ""},
        {"syn bitop3 zero", SYN(S_BITOP3, 0), 0, 512, 12000, 0, ""},
        {"syn add3 rand", SYN(S_ADD3, 1), 1, 512, 12000, 0, ""},
        {"syn alignbit rand", SYN(S_ALIGN, 1), 1, 512, 12000, 0, ""},
        {"syn alignbit zero", SYN(S_ALIGN, 0), 0, 512, 12000, 0, ""},
        {"syn dpp rand", SYN(S_DPP, 1), 1, 512, 12000, 0, ""},
        {"syn k1mix rand", SYN(S_K1MIX, 1), 1, 512, 12000, 0, ""},
        {"syn k1mix zero", SYN(S_K1MIX, 0), 0, 512, 12000, 0, ""},
        {"syn k1mix feedback rand", SYN(S_K1MIX_FB, 1), 1, 512, 12000, 0, ""},
        {"syn k2mix rand", SYN(S_K2MIX, 1), 1, 512, 12000, 0, ""},
        {"syn mad64 rand", SYN(S_MAD, 1), 1, 512, 12000, 0, ""},
        {"syn mad64 zero", SYN(S_MAD, 0), 0, 512, 12000, 0, ""},
        {"syn madmix rand", SYN(S_MADMIX, 1), 1, 576, 10000, 0, ""},
        {"syn lshrrev_b64 rand", SYN(S_SHR64, 1), 1, 512, 12000, 0, ""},
        {"syn lshl_add_u64 rand", SYN(S_ADD64, 1), 1, 512, 12000, 0, ""},
        {"syn sub_co+subb rand", SYN(S_SUBP, 1), 1, 512, 12000, 0, ""},
        {"syn mul_lo_u32 rand", SYN(S_MULLO, 1), 1, 512, 12000, 0, ""},
    };
    Rec *rec;
    uint64_t *sink;
    const int maxwaves = 1024 * 8;
    (void)hipMalloc(&rec, sizeof(Rec) * maxwaves);
    (void)hipMalloc(&sink, 8);
    (void)hipMemset(sink, 0, 8);
    std::vector<Rec> h;
    printf("# stream                     data  vgpr  W  grid   wall_ms  ns/inst/SIMD  cyc/inst@2.4  memtime_GHz  tick_cyc/inst  SIMDs  waves/SIMD(min..max)  overlap  units/s\n");
    for (auto &e : ents) {
        if (*filter && !strstr(e.name, filter)) continue;
        hipFuncAttributes at;
        (void)hipFuncGetAttributes(&at, (const void *)e.f);
        const int vg = (at.numRegs + 7) / 8 * 8;
        const int wcap = std::min(8, 512 / std::max(vg, 1));
        for (int W : {1, 2, 3, 4, 5, 6, 7, 8}) {
            if (W > wcap || W > wmax_arg) continue;
            if (W != 1 && W != 2 && W != 4 && W != 8 && W != wcap) continue;  // 1, 2, 4, 8 and the kernel's own maximum
            const int blocks = 256 * W, waves = blocks * 4;
            const uint32_t m = e.data ? 0xffffffffu : 0u;
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, rec, 20u, m, sink);
            (void)hipDeviceSynchronize();
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, rec, e.iters, m, sink);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            h.resize(waves);
            (void)hipMemcpy(h.data(), rec, sizeof(Rec) * waves, hipMemcpyDeviceToHost);
            double ghz, overlap;
            int simds, minw, maxw;
            analyse(h, W, ghz, simds, minw, maxw, overlap);
            double tick = 0;
            for (auto &r : h) tick += (double)(r.t1 - r.t0);
            tick /= waves;
            const double insts = e.insts * e.iters;  // per wave
            char a[32] = "      -", b[32] = "      -", c[32] = "      -";
            if (insts > 0) {
                snprintf(a, sizeof a, "%7.3f", ms * 1e6 / (insts * W));
                snprintf(b, sizeof b, "%7.2f", ms * 1e-3 * 2.4e9 / (insts * W));
                snprintf(c, sizeof c, "%7.2f", tick / (insts * W));
            }
            printf("%-28s %4s  %4d  %d  %5d  %8.3f  %s       %s       %6.3f        %s   %5d   %d..%d                 %5.3f   %.4g %s/s\n",
                   e.name, e.data ? "rand" : "zero", at.numRegs, W, blocks, ms, a, b, ghz, c, simds, minw, maxw, overlap,
                   e.units > 0 ? e.units * waves * e.iters / (ms * 1e-3) : 0.0, e.unit);
            fflush(stdout);
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
    }
    return 0;
}
