// microbench.hip — per-instruction VALU issue cost on gfx950 (cycles per wave64 instruction per SIMD).
// Each wave runs LOOPS x 64 independent instances of one instruction (8 register chains) between two
// s_memtime stamps; reported = SIMD cycles per instruction when W waves share a SIMD
// ( = stamp delta / instructions issued by ONE wave / ... see main ).  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define DEF_KERNEL(NAME, ASMLINE)                                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned long long *out, int loops)                        \
    {                                                                                                          \
        unsigned r0 = threadIdx.x, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 * 11, r5 = r0 ^ 77, \
                 r6 = r0 + 9, r7 = r0 * 13;                                                                    \
        unsigned a = r0 * 17 + 5, b = r0 * 19 + 3;                                                             \
        unsigned long long q0 = r0, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5, q6 = r6, q7 = r7;             \
        double d0 = r0, d1 = r1, d2 = r2, d3 = r3, d4 = r4, d5 = r5, d6 = r6, d7 = r7, da = 1.0000001;         \
        unsigned long long t0, t1;                                                                             \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); \
        for (int i = 0; i < loops; i++) {                                                                      \
            BODY64(ASMLINE)                                                                                    \
        }                                                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                             \
        unsigned acc = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (unsigned)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ \
                       (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);                                      \
        if (acc == 0x13572468u) out[1] = acc;                                                                  \
        if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;                     \
    }

#define X_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r##i) : "v"(a));
#define X_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r##i) : "v"(a), "v"(b));
#define X_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(r##i) : "v"(a));
#define X_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define X_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r##i) : "v"(a));
#define X_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define X_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r##i) : "v"(a));
#define X_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r##i) : "v"(a));
#define X_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define X_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##i) : "v"(a), "v"(b) : "vcc");
#define X_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q##i));
#define X_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc" : "+v"(r##i) : "v"(a), "v"(b) : "vcc");
#define X_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##i) : "v"(a) : "vcc");
#define X_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define X_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r##i));
#define X_XORDPP(i) asm volatile("v_xor_b32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r##i) : "v"(a));
#define X_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
#define X_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q##i) : "v"(q7));
#define X_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d##i) : "v"(da));
#define X_SWAP32(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r##i), "+v"(a));
#define X_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r##i) : "v"(a));
#define X_MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r##i) : "v"(a));
#define X_DEPXOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r0) : "v"(a));
#define X_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(r##i) : "v"(a));

DEF_KERNEL(xor, X_XOR)
DEF_KERNEL(bitop3, X_BITOP3)
DEF_KERNEL(alignbit, X_ALIGNBIT)
DEF_KERNEL(and_or, X_ANDOR)
DEF_KERNEL(add, X_ADD)
DEF_KERNEL(add3, X_ADD3)
DEF_KERNEL(mul_lo, X_MULLO)
DEF_KERNEL(mul_hi, X_MULHI)
DEF_KERNEL(mad_u32_u24, X_MAD24)
DEF_KERNEL(mul_u32_u24, X_MUL24)
DEF_KERNEL(mul_hi_u32_u24, X_MULHI24)
DEF_KERNEL(mad_u64_u32, X_MAD64)
DEF_KERNEL(lshl64, X_LSHL64)
DEF_KERNEL(add_co_addc, X_ADDCO)
DEF_KERNEL(cndmask, X_CNDMASK)
DEF_KERNEL(perm_b32, X_PERM)
DEF_KERNEL(mov_dpp, X_DPP)
DEF_KERNEL(xor_dpp, X_XORDPP)
DEF_KERNEL(fma_f32, X_FMA32)
DEF_KERNEL(pk_fma_f32, X_PKFMA)
DEF_KERNEL(fma_f64, X_FMA64)
DEF_KERNEL(permlane32_swap, X_SWAP32)
DEF_KERNEL(dep_xor, X_DEPXOR)
DEF_KERNEL(bpermute_wait, X_BPERM)

typedef void (*kfn)(unsigned long long *, int);
struct Ent { const char *name; kfn f; int per; };

int main()
{
    Ent ents[] = {{"v_xor_b32", k_xor, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_alignbit_b32", k_alignbit, 1},
                  {"v_and_or_b32", k_and_or, 1}, {"v_add_u32", k_add, 1}, {"v_add3_u32", k_add3, 1},
                  {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1}, {"v_mad_u32_u24", k_mad_u32_u24, 1},
                  {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1},
                  {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_lshlrev_b64", k_lshl64, 1},
                  {"v_add_co+v_addc_co (pair)", k_add_co_addc, 1}, {"v_cndmask_b32", k_cndmask, 1},
                  {"v_perm_b32", k_perm_b32, 1}, {"v_mov_b32_dpp", k_mov_dpp, 1}, {"v_xor_b32_dpp", k_xor_dpp, 1},
                  {"v_fma_f32", k_fma_f32, 1}, {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_fma_f64", k_fma_f64, 1},
                  {"v_permlane32_swap", k_permlane32_swap, 1}, {"v_xor_b32 dependent chain", k_dep_xor, 1},
                  {"ds_bpermute_b32+wait", k_bpermute_wait, 1}};
    const int loops = 2000;
    unsigned long long *out;
    hipMalloc(&out, 8 * (2 + 4096 * 4));
    std::vector<unsigned long long> h(2 + 4096 * 4);
    printf("%-28s %10s %10s %10s %10s   (SIMD cycles per wave64 instruction; W = waves per SIMD; all 256 CUs busy)\n", "instruction",
           "W=1", "W=2", "W=4", "wall W=4");
    for (auto &e : ents) {
        printf("%-28s", e.name);
        for (int W : {1, 2, 4}) {
            // blocks of 256 threads = 4 waves = one per SIMD; W blocks per CU
            int blocks = 256 * W;
            hipMemset(out, 0, 8 * (2 + 4096 * 4));
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, out, loops);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), out, 8 * (2 + blocks * 4), hipMemcpyDeviceToHost);
            double sum = 0;
            for (int i = 0; i < blocks * 4; i++) sum += (double)h[2 + i];
            double cyc_per_wave = sum / (blocks * 4);
            // W waves interleave on the SIMD: SIMD cycles per instruction = per-wave stamp delta / (insts per wave * W)
            double per_inst = cyc_per_wave / ((double)loops * 64) / W;
            printf(" %10.2f", per_inst);
            if (W == 4) printf(" %10.2f", ms * 1e-3 * 2.4e9 / ((double)loops * 64) / W);
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
        printf("\n");
    }
    return 0;
}
