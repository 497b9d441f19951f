// microbench_fmul.hip — VERDICT r1 item 4: would a double-precision-FMA field multiplication beat the
// v_mad_u64_u32 one on MI355X?  Both forms are priced as INSTRUCTION STREAMS of one 448-bit Goldilocks multiplication
// (independent instructions on hard-coded registers: an upper bound on what either form can issue, no correctness
// involved), at one and two waves per SIMD, in wall-clock time over ~10 ms launches.
//
//  MAD form (ed448_dev.h, Karatsuba over phi = 2^224, 16 x 28-bit limbs):
//      192 v_mad_u64_u32 | 16 v_add_u32 (limb sums) | 30 64-bit subtractions (v_sub_co_u32 + v_subb_co_u32)
//      | 14 v_lshl_add_u64 | carry propagation 16 x (v_lshrrev_b64, v_and_b32, v_lshl_add_u64)            = 330
//  DFMA form (9 x 52-bit limbs, the exact-product trick of Emmart et al.: hi = fma_rz(a, b, 2^104), lo = fma_rz(a, b, -hi')):
//      162 v_fma_f64 (two per limb product, 81 products) | 162 v_lshl_add_u64 (hi and lo parts into integer columns)
//      | 17 column corrections (v_lshl_add_u64) | carry propagation 17 x 3 | reduction mod 2^448 - 2^224 - 1 with limbs
//      that do not align to 224 bits: 9 x (2 v_lshlrev_b64 + v_lshrrev_b64 + 2 v_lshl_add_u64)                = 437
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/microbench_fmul tools/microbench_fmul.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CLOB                                                                                                      \
    "vcc", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", \
        "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37",  \
        "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"

// eight independent instances of each instruction (different destination registers)
#define MAD8                                                                                                     \
    "v_mad_u64_u32 v[8:9], vcc, v40, v41, v[8:9]\n\tv_mad_u64_u32 v[10:11], vcc, v42, v43, v[10:11]\n\t"         \
    "v_mad_u64_u32 v[12:13], vcc, v40, v43, v[12:13]\n\tv_mad_u64_u32 v[14:15], vcc, v41, v42, v[14:15]\n\t"     \
    "v_mad_u64_u32 v[16:17], vcc, v44, v45, v[16:17]\n\tv_mad_u64_u32 v[18:19], vcc, v46, v47, v[18:19]\n\t"     \
    "v_mad_u64_u32 v[20:21], vcc, v44, v47, v[20:21]\n\tv_mad_u64_u32 v[22:23], vcc, v45, v46, v[22:23]\n\t"
#define FMA8                                                                                                     \
    "v_fma_f64 v[8:9], v[40:41], v[42:43], v[8:9]\n\tv_fma_f64 v[10:11], v[44:45], v[46:47], v[10:11]\n\t"       \
    "v_fma_f64 v[12:13], v[40:41], v[46:47], v[12:13]\n\tv_fma_f64 v[14:15], v[42:43], v[44:45], v[14:15]\n\t"   \
    "v_fma_f64 v[16:17], v[40:41], v[44:45], v[16:17]\n\tv_fma_f64 v[18:19], v[42:43], v[46:47], v[18:19]\n\t"   \
    "v_fma_f64 v[20:21], v[40:41], v[42:43], v[20:21]\n\tv_fma_f64 v[22:23], v[44:45], v[46:47], v[22:23]\n\t"
#define ADD64_8                                                                                                  \
    "v_lshl_add_u64 v[24:25], v[8:9], 0, v[24:25]\n\tv_lshl_add_u64 v[26:27], v[10:11], 0, v[26:27]\n\t"         \
    "v_lshl_add_u64 v[28:29], v[12:13], 0, v[28:29]\n\tv_lshl_add_u64 v[30:31], v[14:15], 0, v[30:31]\n\t"       \
    "v_lshl_add_u64 v[32:33], v[16:17], 0, v[32:33]\n\tv_lshl_add_u64 v[34:35], v[18:19], 0, v[34:35]\n\t"       \
    "v_lshl_add_u64 v[36:37], v[20:21], 0, v[36:37]\n\tv_lshl_add_u64 v[38:39], v[22:23], 0, v[38:39]\n\t"
#define SUB64_2 "v_sub_co_u32 v24, vcc, v24, v8\n\tv_subb_co_u32 v25, vcc, v25, v9, vcc\n\tv_sub_co_u32 v26, vcc, v26, v10\n\tv_subb_co_u32 v27, vcc, v27, v11, vcc\n\t"
#define ADD32_8                                                                                                  \
    "v_add_u32 v24, v8, v40\n\tv_add_u32 v25, v9, v41\n\tv_add_u32 v26, v10, v42\n\tv_add_u32 v27, v11, v43\n\t" \
    "v_add_u32 v28, v12, v44\n\tv_add_u32 v29, v13, v45\n\tv_add_u32 v30, v14, v46\n\tv_add_u32 v31, v15, v47\n\t"
#define CARRY1 "v_lshrrev_b64 v[32:33], 28, v[8:9]\n\tv_and_b32 v34, 0xfffffff, v8\n\tv_lshl_add_u64 v[36:37], v[32:33], 0, v[10:11]\n\t"
#define SHIFT5 "v_lshlrev_b64 v[32:33], 16, v[8:9]\n\tv_lshlrev_b64 v[34:35], 16, v[10:11]\n\tv_lshrrev_b64 v[36:37], 36, v[12:13]\n\tv_lshl_add_u64 v[24:25], v[32:33], 0, v[24:25]\n\tv_lshl_add_u64 v[26:27], v[34:35], 0, v[26:27]\n\t"

#define X2(P) P P
#define X3(P) P P P
#define X4(P) P P P P
#define X8(P) X4(P) X4(P)
#define X12(P) X8(P) X4(P)
#define X15(P) X8(P) X4(P) X3(P)
#define X16(P) X8(P) X8(P)
#define X17(P) X16(P) P
#define X20(P) X16(P) X4(P)
#define X24(P) X16(P) X8(P)

// one field multiplication each
#define FMUL_MAD X24(MAD8) X2(ADD32_8) X15(SUB64_2) ADD64_8 X4("v_lshl_add_u64 v[24:25], v[8:9], 0, v[24:25]\n\t") X2("v_lshl_add_u64 v[26:27], v[10:11], 0, v[26:27]\n\t") X16(CARRY1)
#define FMUL_DFMA X20(FMA8) X2("v_fma_f64 v[8:9], v[40:41], v[42:43], v[8:9]\n\t") X20(ADD64_8) X2("v_lshl_add_u64 v[24:25], v[8:9], 0, v[24:25]\n\t") X17("v_lshl_add_u64 v[26:27], v[10:11], 0, v[26:27]\n\t") X17(CARRY1) X8(SHIFT5) SHIFT5

// r04: timing skeleton of a field multiplication spread over TWO lanes on the Goldilocks split (lane 0
// holds a0, b0, lane 1 holds a1, b1; each lane computes its own half product and one triangle of (a0 + a1)(b0 + b1)), per
// lane: 100 v_mad_u64_u32 (64 + 36) | 16 v_add_u32_dpp (limb sums from the partner) | 32 v_cndmask_b32 (the reversed operand
// copy of the lane that takes the upper triangle; role-dependent operands of the recombination) | 16 64-bit subtractions
// | 24 64-bit additions | 20 v_mov_b32_dpp (the 8-column exchange and the two carries) | carry propagation 8 x 3 | 10 limb
// additions (cross-lane carries, wrap)                                                                                   = 258
#define DPPADD8                                                                                                          \
    "v_add_u32_dpp v24, v40, v41 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v25, v42, v43 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_u32_dpp v26, v44, v45 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v27, v46, v47 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_u32_dpp v28, v40, v43 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v29, v42, v45 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_u32_dpp v30, v44, v47 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v31, v46, v41 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define DPPMOV4                                                                                                          \
    "v_mov_b32_dpp v32, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v33, v9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_mov_b32_dpp v34, v10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v35, v11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define CND8                                                                                                             \
    "v_cndmask_b32 v24, v8, v9, vcc\n\tv_cndmask_b32 v25, v10, v11, vcc\n\tv_cndmask_b32 v26, v12, v13, vcc\n\tv_cndmask_b32 v27, v14, v15, vcc\n\t" \
    "v_cndmask_b32 v28, v16, v17, vcc\n\tv_cndmask_b32 v29, v18, v19, vcc\n\tv_cndmask_b32 v30, v20, v21, vcc\n\tv_cndmask_b32 v31, v22, v23, vcc\n\t"
#define FMUL_2LANE X12(MAD8) X4("v_mad_u64_u32 v[8:9], vcc, v40, v41, v[8:9]\n\t") X2(DPPADD8) X4(CND8) X8(SUB64_2) X3(ADD64_8) X4(DPPMOV4) DPPMOV4 X8(CARRY1) ADD32_8 X2("v_add_u32 v24, v8, v40\n\t")

#define DEF(NAME, BODY)                                                                  \
    __global__ __launch_bounds__(256) void k_##NAME(int loops)                           \
    {                                                                                    \
        for (int i = 0; i < loops; i++) asm volatile(BODY ::: CLOB);                     \
    }
DEF(mad, X4(FMUL_MAD))
DEF(dfma, X4(FMUL_DFMA))
DEF(two_lane, X4(FMUL_2LANE))
DEF(mad_only, X24(MAD8))
DEF(fma_only, X20(FMA8) X2("v_fma_f64 v[8:9], v[40:41], v[42:43], v[8:9]\n\t"))

typedef void (*kfn)(int);
struct Ent { const char *name; kfn f; double fmuls_per_trip; int insts; };

int main()
{
    Ent ents[] = {{"MAD form, whole multiplication (330 instr)", k_mad, 4, 330},
                  {"DFMA form, whole multiplication (437 instr)", k_dfma, 4, 437},
                  {"two lanes per multiplication, per lane (258 instr)", k_two_lane, 4, 258},
                  {"192 v_mad_u64_u32 alone", k_mad_only, 1, 192},
                  {"162 v_fma_f64 alone", k_fma_only, 1, 162}};
    printf("%-46s %6s %12s %22s %14s\n", "stream", "W", "ms", "G field-mults/s (chip)", "ns/instr/SIMD");
    for (auto &e : ents) {
        for (int W : {1, 2, 4}) {
            const int blocks = 256 * W, loops = 6000;
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, 200);
            (void)hipDeviceSynchronize();
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, loops);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double fm = (double)blocks * 4 * 64 * loops * e.fmuls_per_trip;  // lane-multiplications
            const double insts_per_simd = (double)W * loops * e.fmuls_per_trip * e.insts;
            printf("%-46s %6d %12.2f %22.2f %14.3f\n", e.name, W, ms, fm / (ms * 1e-3) / 1e9, ms * 1e6 / insts_per_simd);
        }
    }
    return 0;
}
