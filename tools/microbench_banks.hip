// microbench_banks.hip — does the VGPR operand pattern decide whether a second wave on the SIMD gains anything?
// (follow-up to profiles/r02_second_issue_slot.txt: two waves of the keccak kernels share a SIMD at exactly half
// rate each although tools/microbench.hip shows v_bitop3_b32 doubling from W=1 to W=2 -- there every instruction
// re-reads the same two source registers.)  Each kernel is one asm block over hard-coded physical registers
// v8..v71, 64 instructions per trip, same layout as microbench.hip: SIMD cycles per wave64 instruction at W waves/SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/microbench_banks tools/microbench_banks.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include <vector>

#define CLOB                                                                                                      \
    "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
        "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38",  \
        "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",  \
        "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68",  \
        "v69", "v70", "v71"

// 8 independent chains; chain c owns registers whose index pattern is chosen per kernel.
// I(d, a, b, c) expands to one instruction.
#define BITOP3(d, a, b, c) "v_bitop3_b32 v" #d ", v" #a ", v" #b ", v" #c " bitop3:0x96\n\t"
#define ALIGN(d, a, b, c) "v_alignbit_b32 v" #d ", v" #a ", v" #b ", 7\n\t"
#define XOR2(d, a, b, c) "v_xor_b32 v" #d ", v" #a ", v" #b "\n\t"
#define DPPM(d, a, b, c) "v_mov_b32_dpp v" #d ", v" #a " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"

// pattern SAME: the three sources of every instruction share one bank (index mod 4 equal), dst = src0
#define PAT_SAME(I) I(8, 8, 12, 16) I(9, 9, 13, 17) I(10, 10, 14, 18) I(11, 11, 15, 19) I(20, 20, 24, 28) I(21, 21, 25, 29) I(22, 22, 26, 30) I(23, 23, 27, 31)
// pattern DIFF: three sources in three different banks
#define PAT_DIFF(I) I(8, 8, 13, 18) I(9, 9, 14, 19) I(10, 10, 15, 16) I(11, 11, 12, 17) I(20, 20, 25, 30) I(21, 21, 26, 31) I(22, 22, 27, 28) I(23, 23, 24, 29)
// pattern FIX: dst = src0, the other two sources are the same two registers for every instruction (microbench.hip)
#define PAT_FIX(I) I(8, 8, 40, 41) I(9, 9, 40, 41) I(10, 10, 40, 41) I(11, 11, 40, 41) I(20, 20, 40, 41) I(21, 21, 40, 41) I(22, 22, 40, 41) I(23, 23, 40, 41)
// pattern WIDE: like a keccak round -- 24 different destinations, sources wander over 48 registers, banks differ
#define PAT_WIDE(I)                                                                                               \
    I(48, 8, 13, 18) I(49, 9, 14, 19) I(50, 10, 15, 20) I(51, 11, 16, 21) I(52, 12, 17, 22) I(53, 13, 18, 23)     \
    I(54, 14, 19, 24) I(55, 15, 20, 25)
#define PAT_WIDE2(I)                                                                                              \
    I(56, 16, 21, 26) I(57, 17, 22, 27) I(58, 18, 23, 28) I(59, 19, 24, 29) I(60, 20, 25, 30) I(61, 21, 26, 31)   \
    I(62, 22, 27, 32) I(63, 23, 28, 33)
// WIDE with all three sources in ONE bank
#define PAT_WSAME(I)                                                                                              \
    I(48, 8, 12, 16) I(49, 9, 13, 17) I(50, 10, 14, 18) I(51, 11, 15, 19) I(52, 12, 16, 20) I(53, 13, 17, 21)     \
    I(54, 14, 18, 22) I(55, 15, 19, 23)
#define PAT_WSAME2(I)                                                                                             \
    I(56, 16, 20, 24) I(57, 17, 21, 25) I(58, 18, 22, 26) I(59, 19, 23, 27) I(60, 20, 24, 28) I(61, 21, 25, 29)   \
    I(62, 22, 26, 30) I(63, 23, 27, 31)

#define X8(P) P P P P P P P P
#define X4(P) P P P P

#define DEF(NAME, BODY)                                                                                           \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned long long *out, int loops)                           \
    {                                                                                                             \
        unsigned long long t0, t1;                                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); \
        for (int i = 0; i < loops; i++) asm volatile(BODY ::: CLOB);                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                                \
        if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;                        \
    }

DEF(bitop3_fix, X8(PAT_FIX(BITOP3)))
DEF(bitop3_same, X8(PAT_SAME(BITOP3)))
DEF(bitop3_diff, X8(PAT_DIFF(BITOP3)))
DEF(bitop3_wide, X4(PAT_WIDE(BITOP3) PAT_WIDE2(BITOP3)))
DEF(bitop3_wsame, X4(PAT_WSAME(BITOP3) PAT_WSAME2(BITOP3)))
DEF(xor_fix, X8(PAT_FIX(XOR2)))
DEF(xor_same, X8(PAT_SAME(XOR2)))
DEF(xor_diff, X8(PAT_DIFF(XOR2)))
DEF(xor_wide, X4(PAT_WIDE(XOR2) PAT_WIDE2(XOR2)))
DEF(align_fix, X8(PAT_FIX(ALIGN)))
DEF(align_same, X8(PAT_SAME(ALIGN)))
DEF(align_diff, X8(PAT_DIFF(ALIGN)))
DEF(align_wide, X4(PAT_WIDE(ALIGN) PAT_WIDE2(ALIGN)))
DEF(dpp_wide, X4(PAT_WIDE(DPPM) PAT_WIDE2(DPPM)))
// the two-lane round's mix: 60 bitop3 : 29 alignbit : 29 dpp  ~ 2 : 1 : 1
DEF(mix_wide, X4(PAT_WIDE(BITOP3) PAT_WIDE2(ALIGN)) X4(PAT_WIDE(BITOP3) PAT_WIDE2(DPPM)))

// Long straight-line bodies (4096 instructions = 32 KB per trip, 16 trips) with an optional per-wave start skew, so
// that the waves of a SIMD / CU do NOT walk the code in lockstep (in the short loops above they do: all waves
// start together and fetch the same 512 bytes).  `loops` < 0 selects the skew.
#define DEFLONG(NAME, BODY)                                                                                       \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned long long *out, int loops)                           \
    {                                                                                                             \
        unsigned long long t0, t1;                                                                                \
        if (loops < 0) {                                                                                          \
            loops = -loops;                                                                                       \
            const int skew = (int)((blockIdx.x * 4 + threadIdx.x / 64) * 37 % 61);                                \
            for (int i = 0; i < skew; i++) asm volatile("s_sleep 3" ::: "memory");                                \
        }                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); \
        for (int i = 0; i < loops; i++) asm volatile(BODY ::: CLOB);                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                                \
        if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;                        \
    }
#define X64(P) X8(X8(P))
DEFLONG(long_bitop3, X64(X4(PAT_WIDE(BITOP3) PAT_WIDE2(BITOP3))))
DEFLONG(long_xor, X64(X4(PAT_WIDE(XOR2) PAT_WIDE2(XOR2))))
DEFLONG(long_mix, X64(X4(PAT_WIDE(BITOP3) PAT_WIDE2(ALIGN))) X64(X4(PAT_WIDE(BITOP3) PAT_WIDE2(DPPM))))

typedef void (*kfn)(unsigned long long *, int);
struct Ent { const char *name; kfn f; int per_trip; };

int main()
{
    Ent ents[] = {{"bitop3  fixed b,c", k_bitop3_fix, 64},      {"bitop3  3 srcs one bank", k_bitop3_same, 64},
                  {"bitop3  3 srcs 3 banks", k_bitop3_diff, 64}, {"bitop3  wide, 3 banks", k_bitop3_wide, 64},
                  {"bitop3  wide, one bank", k_bitop3_wsame, 64}, {"xor     fixed b", k_xor_fix, 64},
                  {"xor     2 srcs one bank", k_xor_same, 64},    {"xor     2 srcs 2 banks", k_xor_diff, 64},
                  {"xor     wide", k_xor_wide, 64},               {"alignbit fixed b", k_align_fix, 64},
                  {"alignbit one bank", k_align_same, 64},        {"alignbit 2 banks", k_align_diff, 64},
                  {"alignbit wide", k_align_wide, 64},            {"mov_dpp wide", k_dpp_wide, 64},
                  {"k2 mix 2:1:1 wide", k_mix_wide, 128},
                  {"LONG bitop3 lockstep", k_long_bitop3, 4096},  {"LONG bitop3 skewed", k_long_bitop3, -4096},
                  {"LONG xor(4B) lockstep", k_long_xor, 4096},    {"LONG xor(4B) skewed", k_long_xor, -4096},
                  {"LONG k2 mix lockstep", k_long_mix, 8192},     {"LONG k2 mix skewed", k_long_mix, -8192}};
    const int loops_short = 2000;
    unsigned long long *out;
    hipMalloc(&out, 8 * (2 + 8192 * 4));
    std::vector<unsigned long long> h(2 + 8192 * 4);
    printf("%-28s %9s %9s %9s %9s | %9s %9s %9s %9s  (left: SIMD ticks per wave64 instruction from s_memtime; right: wall ns*2.4 per instruction per SIMD)\n",
           "pattern", "W=1", "W=2", "W=4", "W=8", "W=1", "W=2", "W=4", "W=8");
    for (auto &e : ents) {
        printf("%-28s", e.name);
        std::string wall;
        const bool skew = e.per_trip < 0;
        if (skew) e.per_trip = -e.per_trip;
        const int loops = e.per_trip >= 4096 ? 64 : loops_short;
        for (int W : {1, 2, 4, 8}) {
            int blocks = 256 * W;
            hipMemset(out, 0, 8 * (2 + 8192 * 4));
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, out, skew ? -loops : loops);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), out, 8 * (2 + blocks * 4), hipMemcpyDeviceToHost);
            double sum = 0;
            for (int i = 0; i < blocks * 4; i++) sum += (double)h[2 + i];
            double per_inst = sum / (blocks * 4) / ((double)loops * e.per_trip) / W;
            printf(" %9.2f", per_inst);
            char buf[32];
            snprintf(buf, sizeof buf, " %9.2f", ms * 1e-3 * 2.4e9 / ((double)loops * e.per_trip) / W);
            wall += buf;
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
        printf(" |%s\n", wall.c_str());
    }
    return 0;
}
