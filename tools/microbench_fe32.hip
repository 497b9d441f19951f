// microbench_fe32.hip — VERDICT r5 item 1: does Ed448 field arithmetic on 14 SATURATED 32-bit limbs (147 / 84 multiply-adds
// per multiplication / squaring; tools/fe32.h) beat the library's 16 x 28-bit lazy-carry form (192 / 108; ed448_dev.h)?
// Both forms are REAL arithmetic here: every lane runs the same dependent chain of multiplications and squarings on its own
// random operands in both radices, the two results are compared byte for byte (and the CPU build of both is compared before
// anything is timed), and the chains are timed at one and two waves per SIMD with the shader clock read per wave
// (s_memtime / s_memrealtime).  The 28-bit form is compiled the way ed448_vb2.hip compiles it (pinned multiply-add chains,
// raised priority around the 4-cycle blocks) AND the way ed448.hip does (neither).
//   mix  = 3 multiplications + 4 squarings per trip, the mix of a point doubling;  mul / sqr = that operation alone.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I capycrypt_amd/csrc [-DFE28_VB2=1] -o tools/microbench_fe32 tools/microbench_fe32.hip
#ifdef FE28_VB2
#define CAPY_ED448_ASM_MAD 1
#define CAPY_ED448_SETPRIO 2
#endif
#include "ed448_dev.h"
#include "fe32.h"
#include <stdio.h>
#include <string.h>
#include <vector>

using capy::Fe;
using capy32::Fe32;

#define HIPCHECK(x)                                                                   \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)

template <int MODE, typename F, typename MUL, typename SQR>
__host__ __device__ inline void chain(F &x, F &y, int trips, MUL mul, SQR sqr)
{
#pragma unroll 1
    for (int t = 0; t < trips; t++) {
        if (MODE == 0) {
            x = mul(x, y);
            y = sqr(y);
            x = sqr(x);
            y = mul(y, x);
            x = sqr(x);
            y = sqr(y);
            x = mul(x, y);
        } else if (MODE == 1) {
            x = mul(x, y);
            y = mul(y, x);
        } else {
            x = sqr(x);
            y = sqr(y);
        }
    }
}

struct Stamp {
    uint64_t t0, r0, t1, r1;
};
#define STAMP(t, r) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r)::"memory")

template <int MODE>
__global__ __launch_bounds__(256) void k28(const uint8_t *in, uint8_t *out, Stamp *st, int trips)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fe x = capy::fe_from_bytes(in + i * 112), y = capy::fe_from_bytes(in + i * 112 + 56);
    uint64_t t0, r0, t1, r1;
    STAMP(t0, r0);
    chain<MODE>(x, y, trips, [](const Fe &a, const Fe &b) { return capy::fe_mul(a, b); }, [](const Fe &a) { return capy::fe_sqr(a); });
    STAMP(t1, r1);
    capy::fe_to_bytes(out + i * 112, x);
    capy::fe_to_bytes(out + i * 112 + 56, y);
    if ((threadIdx.x & 63) == 0) st[i >> 6] = Stamp{t0, r0, t1, r1};
}
template <int MODE>
__global__ __launch_bounds__(256) void k32(const uint8_t *in, uint8_t *out, Stamp *st, int trips)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fe32 x = capy32::fe32_from_bytes(in + i * 112), y = capy32::fe32_from_bytes(in + i * 112 + 56);
    uint64_t t0, r0, t1, r1;
    STAMP(t0, r0);
    chain<MODE>(x, y, trips, [](const Fe32 &a, const Fe32 &b) { return capy32::fe32_mul(a, b); },
                [](const Fe32 &a) { return capy32::fe32_sqr(a); });
    STAMP(t1, r1);
    capy32::fe32_to_bytes(out + i * 112, x);
    capy32::fe32_to_bytes(out + i * 112 + 56, y);
    if ((threadIdx.x & 63) == 0) st[i >> 6] = Stamp{t0, r0, t1, r1};
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// CPU build of both forms on the same chains (also covers all-ones / near-p operands that make every carry path fire)
static int cpu_selftest()
{
    int bad = 0;
    for (int it = 0; it < 2000; it++) {
        uint8_t in[112], o28[112], o32[112];
        for (int j = 0; j < 112; j++) in[j] = (uint8_t)rnd();
        if (it % 8 == 1) memset(in, 0xff, 112);
        if (it % 8 == 2) memset(in, 0xff, 56);
        if (it % 8 == 3) {
            memset(in, 0xff, 112);
            in[28] = 0xfe;  // p itself
        }
        if (it % 8 == 4) memset(in + 28, 0, 28);
        Fe x = capy::fe_from_bytes(in), y = capy::fe_from_bytes(in + 56);
        Fe32 x2 = capy32::fe32_from_bytes(in), y2 = capy32::fe32_from_bytes(in + 56);
        chain<0>(x, y, 3, [](const Fe &a, const Fe &b) { return capy::fe_mul(a, b); }, [](const Fe &a) { return capy::fe_sqr(a); });
        chain<0>(x2, y2, 3, [](const Fe32 &a, const Fe32 &b) { return capy32::fe32_mul(a, b); },
                 [](const Fe32 &a) { return capy32::fe32_sqr(a); });
        capy::fe_to_bytes(o28, x);
        capy::fe_to_bytes(o28 + 56, y);
        capy32::fe32_to_bytes(o32, x2);
        capy32::fe32_to_bytes(o32 + 56, y2);
        if (memcmp(o28, o32, 112)) bad++;
    }
    printf("# CPU build: 2000 chains of 9 multiplications + 12 squarings, radix 2^28 vs radix 2^32: %d mismatches\n", bad);
    return bad;
}

typedef void (*kfn)(const uint8_t *, uint8_t *, Stamp *, int);

int main()
{
    if (cpu_selftest()) return 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        printf("# no GPU: CPU self-test only\n");
        return 0;
    }
    const int maxW = 2, lanes = 256 * 256 * maxW;
    std::vector<uint8_t> h_in((size_t)lanes * 112), h28((size_t)lanes * 112), h32((size_t)lanes * 112);
    for (auto &b : h_in) b = (uint8_t)rnd();
    uint8_t *d_in, *d_out;
    Stamp *d_st;
    HIPCHECK(hipMalloc(&d_in, h_in.size()));
    HIPCHECK(hipMalloc(&d_out, h_in.size()));
    HIPCHECK(hipMalloc(&d_st, sizeof(Stamp) * lanes / 64));
    HIPCHECK(hipMemcpy(d_in, h_in.data(), h_in.size(), hipMemcpyHostToDevice));
    struct Ent {
        const char *name;
        kfn f28, f32;
        double muls, sqrs;
    } ents[] = {{"mix (3 mul + 4 sqr)", k28<0>, k32<0>, 3, 4}, {"mul", k28<1>, k32<1>, 2, 0}, {"sqr", k28<2>, k32<2>, 0, 2}};
#ifdef FE28_VB2
    printf("# radix 2^28 compiled as ed448_vb2.hip compiles it: CAPY_ED448_ASM_MAD=1 CAPY_ED448_SETPRIO=2\n");
#else
    printf("# radix 2^28 compiled as ed448.hip compiles it (no pinned chains, no s_setprio)\n");
#endif
    printf("%-22s %-6s %2s %9s %14s %9s %18s\n", "chain", "radix", "W", "ms", "ns/op/SIMD", "GHz", "G field-ops/s chip");
    std::vector<Stamp> st(lanes / 64);
    for (auto &e : ents) {
        double ms_of[2][3] = {{0}};
        for (int W = 1; W <= maxW; W++) {
            const int blocks = 256 * W, trips = 1500;
            for (int radix = 0; radix < 2; radix++) {
                kfn f = radix ? e.f32 : e.f28;
                hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d_in, d_out, d_st, 20);
                HIPCHECK(hipDeviceSynchronize());
                hipEvent_t e0, e1;
                HIPCHECK(hipEventCreate(&e0));
                HIPCHECK(hipEventCreate(&e1));
                float best = 1e30f;
                for (int rep = 0; rep < 3; rep++) {
                    HIPCHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d_in, d_out, d_st, trips);
                    HIPCHECK(hipEventRecord(e1));
                    HIPCHECK(hipDeviceSynchronize());
                    float ms = 0;
                    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                HIPCHECK(hipMemcpy(st.data(), d_st, sizeof(Stamp) * blocks * 4, hipMemcpyDeviceToHost));
                double ghz = 0;
                for (int w = 0; w < blocks * 4; w++) ghz += (double)(st[w].t1 - st[w].t0) / ((double)(st[w].r1 - st[w].r0) * 10.0);
                ghz /= blocks * 4;
                HIPCHECK(hipMemcpy((radix ? h32 : h28).data(), d_out, (size_t)blocks * 256 * 112, hipMemcpyDeviceToHost));
                const double ops = (e.muls + e.sqrs) * trips;  // per lane
                printf("%-22s %-6s %2d %9.3f %14.2f %9.3f %18.2f\n", e.name, radix ? "2^32" : "2^28", W, best,
                       best * 1e6 / (ops * W), ghz, ops * blocks * 256 / (best * 1e-3) / 1e9);
                ms_of[radix][W] = best;
            }
            const size_t nb = (size_t)blocks * 256 * 112;
            printf("#   results of the two radices, %d lanes: %s;  time 2^32 / 2^28 = %.3f\n", blocks * 256,
                   memcmp(h28.data(), h32.data(), nb) ? "DIFFER" : "byte-identical", ms_of[1][W] / ms_of[0][W]);
            if (memcmp(h28.data(), h32.data(), nb)) return 1;
        }
    }
    return 0;
}
