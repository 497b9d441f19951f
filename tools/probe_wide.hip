// probe_wide.hip — VERDICT r1 item 6: "measure the wide-lane form instead of estimating it".
// keccak-f[1600] with ONE 64-bit Keccak lane per GPU lane (25 lanes per sponge, two sponges per wave at lane offsets
// 0 and 32), register resident, cross-lane traffic through ds_bpermute_b32:
//   theta  4 + 4 gathers for the column parity (rows y+1..y+4), 2 + 2 for C[x-1] and C[x+1]           12 bpermute
//   rho    per-lane rotation amount: v_alignbit_b32 with a VGPR shift + selects for the >= 32 and the 0 case
//   pi+chi one gather each of B[x], B[x+1], B[x+2] straight from the rho output (pi folded into the index)   6 bpermute
// and the question whether a lone sponge chain advances faster than in the two-lane form (120 VALU per round, no LDS):
// BASELINE config 3 as specified leaves 128 messages (256 sponges) per GPU, one chain each.
// Prints permutations/s PER SPONGE for both forms at one wave per SIMD and checks the wide form against the one-lane
// permutation bit for bit.   Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I capycrypt_amd/csrc -o tools/probe_wide tools/probe_wide.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "sponge_kernels_k2.h"
using namespace capy;

__device__ __forceinline__ uint32_t bperm(uint32_t byte_index, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_index, (int)v);
}
// a ^ (b & c)
__device__ __forceinline__ uint32_t xor_and(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x78); }

struct WideIdx {
    uint32_t up[4];            // (x, y+k)
    uint32_t xm1, xp1;         // (x-1, y), (x+1, y)
    uint32_t b0, b1, b2;       // pi sources of B[X], B[X+1], B[X+2] in row Y
    uint32_t sh;               // 32 - (r & 31), 0 when r & 31 == 0
    uint32_t m_swap, m_zero, m_l0;  // all-ones masks: r >= 32, r % 32 == 0, lane is Keccak lane 0
};

__device__ __forceinline__ WideIdx wide_setup()
{
    static const uint8_t RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    const uint32_t lane = threadIdx.x & 63, base = lane & 32;
    uint32_t i = lane & 31;
    if (i > 24) i = 24;  // idle lanes mirror lane 24 (their values are never read)
    const uint32_t x = i % 5, y = i / 5;
    auto at = [&](uint32_t xx, uint32_t yy) { return 4 * (base + (xx % 5) + 5 * (yy % 5)); };
    WideIdx w;
    for (int k = 0; k < 4; k++) w.up[k] = at(x, y + 1 + k);
    w.xm1 = at(x + 4, y);
    w.xp1 = at(x + 1, y);
    auto src = [&](uint32_t X, uint32_t Y) { return at((X + 3 * Y) % 5, X % 5); };  // B[X,Y] = rho(e)[(X+3Y)%5, X]
    w.b0 = src(x, y);
    w.b1 = src(x + 1, y);
    w.b2 = src(x + 2, y);
    const uint32_t r = RHO[i];
    w.sh = (32 - (r & 31)) & 31;
    w.m_swap = r >= 32 ? ~0u : 0u;
    w.m_zero = (r & 31) == 0 ? ~0u : 0u;
    w.m_l0 = i == 0 && (lane & 31) == 0 ? ~0u : 0u;
    return w;
}

__device__ __forceinline__ void wide_round(uint32_t &lo, uint32_t &hi, const WideIdx &w, uint32_t rc_lo, uint32_t rc_hi)
{
    // theta: column parity in every lane of the column
    uint32_t g0 = bperm(w.up[0], lo), g1 = bperm(w.up[1], lo), g2 = bperm(w.up[2], lo), g3 = bperm(w.up[3], lo);
    uint32_t h0 = bperm(w.up[0], hi), h1 = bperm(w.up[1], hi), h2 = bperm(w.up[2], hi), h3 = bperm(w.up[3], hi);
    const uint32_t cl = xor3(xor3(lo, g0, g1), g2, g3), ch = xor3(xor3(hi, h0, h1), h2, h3);
    const uint32_t ml = bperm(w.xm1, cl), mh = bperm(w.xm1, ch), pl = bperm(w.xp1, cl), ph = bperm(w.xp1, ch);
    const uint32_t rl = __builtin_amdgcn_alignbit(pl, ph, 31), rh = __builtin_amdgcn_alignbit(ph, pl, 31);  // rol1
    uint32_t el = xor3(lo, ml, rl), eh = xor3(hi, mh, rh);
    // rho: rotate left by the lane's own amount
    const uint32_t a = (eh & w.m_swap) | (el & ~w.m_swap), b = (el & w.m_swap) | (eh & ~w.m_swap);
    const uint32_t ra = __builtin_amdgcn_alignbit(a, b, w.sh), rb = __builtin_amdgcn_alignbit(b, a, w.sh);
    el = (a & w.m_zero) | (ra & ~w.m_zero);
    eh = (b & w.m_zero) | (rb & ~w.m_zero);
    // pi + chi
    const uint32_t b0l = bperm(w.b0, el), b1l = bperm(w.b1, el), b2l = bperm(w.b2, el);
    const uint32_t b0h = bperm(w.b0, eh), b1h = bperm(w.b1, eh), b2h = bperm(w.b2, eh);
    lo = xor_and(chi3(b0l, b1l, b2l), rc_lo, w.m_l0);
    hi = xor_and(chi3(b0h, b1h, b2h), rc_hi, w.m_l0);
}

// variant: theta in ONE LDS round trip -- gather the five lanes of column x-1 and the five of column x+1 directly
// (20 bpermutes instead of 8 + 4, no dependent second trip)
struct WideIdx2 {
    uint32_t cm[5], cp[5];
};
__device__ __forceinline__ WideIdx2 wide_setup2()
{
    const uint32_t lane = threadIdx.x & 63, base = lane & 32;
    uint32_t i = lane & 31;
    if (i > 24) i = 24;
    const uint32_t x = i % 5;
    WideIdx2 w;
    for (int k = 0; k < 5; k++) {
        w.cm[k] = 4 * (base + (x + 4) % 5 + 5 * k);
        w.cp[k] = 4 * (base + (x + 1) % 5 + 5 * k);
    }
    return w;
}
__device__ __forceinline__ void wide_round2(uint32_t &lo, uint32_t &hi, const WideIdx &w, const WideIdx2 &w2, uint32_t rc_lo,
                                            uint32_t rc_hi)
{
    uint32_t ml[5], mh[5], pl[5], ph[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        ml[k] = bperm(w2.cm[k], lo);
        mh[k] = bperm(w2.cm[k], hi);
        pl[k] = bperm(w2.cp[k], lo);
        ph[k] = bperm(w2.cp[k], hi);
    }
    const uint32_t cml = xor3(xor3(ml[0], ml[1], ml[2]), ml[3], ml[4]), cmh = xor3(xor3(mh[0], mh[1], mh[2]), mh[3], mh[4]);
    const uint32_t cpl = xor3(xor3(pl[0], pl[1], pl[2]), pl[3], pl[4]), cph = xor3(xor3(ph[0], ph[1], ph[2]), ph[3], ph[4]);
    const uint32_t rl = __builtin_amdgcn_alignbit(cpl, cph, 31), rh = __builtin_amdgcn_alignbit(cph, cpl, 31);
    uint32_t el = xor3(lo, cml, rl), eh = xor3(hi, cmh, rh);
    const uint32_t a = (eh & w.m_swap) | (el & ~w.m_swap), b = (el & w.m_swap) | (eh & ~w.m_swap);
    const uint32_t ra = __builtin_amdgcn_alignbit(a, b, w.sh), rb = __builtin_amdgcn_alignbit(b, a, w.sh);
    el = (a & w.m_zero) | (ra & ~w.m_zero);
    eh = (b & w.m_zero) | (rb & ~w.m_zero);
    const uint32_t b0l = bperm(w.b0, el), b1l = bperm(w.b1, el), b2l = bperm(w.b2, el);
    const uint32_t b0h = bperm(w.b0, eh), b1h = bperm(w.b1, eh), b2h = bperm(w.b2, eh);
    lo = xor_and(chi3(b0l, b1l, b2l), rc_lo, w.m_l0);
    hi = xor_and(chi3(b0h, b1h, b2h), rc_hi, w.m_l0);
}
__device__ __forceinline__ void wide_permute2(uint32_t &lo, uint32_t &hi, const WideIdx &w, const WideIdx2 &w2)
{
#pragma unroll
    for (int r = 0; r < 24; r++) wide_round2(lo, hi, w, w2, (uint32_t)keccak_rc64(r), (uint32_t)(keccak_rc64(r) >> 32));
}
__global__ __launch_bounds__(64) void wide2_kernel(uint64_t *state, uint32_t iters)
{
    const WideIdx w = wide_setup();
    const WideIdx2 w2 = wide_setup2();
    const uint32_t lane = threadIdx.x, i = lane & 31;
    const uint64_t sponge = (uint64_t)blockIdx.x * 2 + (lane >> 5);
    const uint64_t v = i < 25 ? state[sponge * 25 + i] : 0;
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    for (uint32_t t = 0; t < iters; t++) wide_permute2(lo, hi, w, w2);
    if (i < 25) state[sponge * 25 + i] = ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ void wide_permute(uint32_t &lo, uint32_t &hi, const WideIdx &w)
{
#pragma unroll
    for (int r = 0; r < 24; r++) wide_round(lo, hi, w, (uint32_t)keccak_rc64(r), (uint32_t)(keccak_rc64(r) >> 32));
}

// state in/out: [sponge][25] u64
__global__ __launch_bounds__(64) void wide_kernel(uint64_t *state, uint32_t iters)
{
    const WideIdx w = wide_setup();
    const uint32_t lane = threadIdx.x, i = lane & 31;
    const uint64_t sponge = (uint64_t)blockIdx.x * 2 + (lane >> 5);
    const uint64_t v = i < 25 ? state[sponge * 25 + i] : 0;
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    for (uint32_t t = 0; t < iters; t++) wide_permute(lo, hi, w);
    if (i < 25) state[sponge * 25 + i] = ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void k2_kernel(uint64_t *state, uint32_t iters)
{
    const uint32_t lane = threadIdx.x, h = lane & 1;
    const uint64_t sponge = (uint64_t)blockIdx.x * 32 + (lane >> 1);
    KHalf a;
#pragma unroll
    for (int i = 0; i < 25; i++) a.a[i] = (uint32_t)(state[sponge * 25 + i] >> (32 * h));
    for (uint32_t t = 0; t < iters; t++) keccakf1600_k2_unrolled(a, 0u - h);
    uint32_t *s32 = reinterpret_cast<uint32_t *>(state);
#pragma unroll
    for (int i = 0; i < 25; i++) s32[(sponge * 25 + i) * 2 + h] = a.a[i];
}

__global__ __launch_bounds__(64) void k1_kernel(uint64_t *state, uint32_t iters)
{
    const uint64_t sponge = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    KState a;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        a.lo[i] = (uint32_t)state[sponge * 25 + i];
        a.hi[i] = (uint32_t)(state[sponge * 25 + i] >> 32);
    }
    for (uint32_t t = 0; t < iters; t++) keccakf1600_unrolled(a);
#pragma unroll
    for (int i = 0; i < 25; i++) state[sponge * 25 + i] = ((uint64_t)a.hi[i] << 32) | a.lo[i];
}

template <class K>
static double run(K kern, unsigned waves, unsigned sponges_per_wave, uint64_t *d, const std::vector<uint64_t> &init, uint32_t iters,
                  std::vector<uint64_t> *out)
{
    const size_t n = (size_t)waves * sponges_per_wave * 25;
    (void)hipMemcpy(d, init.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, d, 3u);  // warm
    (void)hipMemcpy(d, init.data(), n * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (out) {
        out->resize(n);
        (void)hipMemcpy(out->data(), d, n * 8, hipMemcpyDeviceToHost);
    }
    return ms;
}

int main()
{
    const unsigned maxw = 2048;
    std::vector<uint64_t> init((size_t)maxw * 64 * 25);
    uint64_t z = 12345;
    for (auto &v : init) {
        z += 0x9E3779B97F4A7C15ULL;
        uint64_t t = z;
        t = (t ^ (t >> 30)) * 0xBF58476D1CE4E5B9ULL;
        t = (t ^ (t >> 27)) * 0x94D049BB133111EBULL;
        v = t ^ (t >> 31);
    }
    uint64_t *d;
    (void)hipMalloc(&d, init.size() * 8);
    // correctness: 5 permutations of the same 128 states through the wide form and the one-lane form
    std::vector<uint64_t> a, b, a2;
    run(wide2_kernel, 64, 2, d, init, 5, &a2);
    run(wide_kernel, 64, 2, d, init, 5, &a);
    run(k1_kernel, 2, 64, d, init, 5, &b);
    int bad = 0;
    for (size_t i = 0; i < 128 * 25; i++) bad += (a[i] != b[i]) + (a2[i] != b[i]);
    printf("wide form vs one-lane form, 128 states x 5 permutations: %s\n", bad ? "MISMATCH" : "bit-exact");
    if (bad) return 1;
    const uint32_t iters = 3000;
    printf("%-34s %8s %10s %16s %18s\n", "form", "waves", "ms", "us/perm/sponge", "rel. to two-lane");
    double k2_us = 0;
    for (unsigned waves : {128u, 512u, 1024u}) {
        const double m2 = run(k2_kernel, waves, 32, d, init, iters, nullptr);
        const double m1 = run(k1_kernel, waves, 64, d, init, iters, nullptr);
        const double mw = run(wide_kernel, waves, 2, d, init, iters, nullptr);
        k2_us = m2 * 1e3 / iters;
        printf("%-34s %8u %10.2f %16.3f %18.2f\n", "one-lane (180 VALU/round)", waves, m1, m1 * 1e3 / iters, k2_us / (m1 * 1e3 / iters));
        printf("%-34s %8u %10.2f %16.3f %18.2f\n", "two-lane (120 VALU/round)", waves, m2, k2_us, 1.0);
        printf("%-34s %8u %10.2f %16.3f %18.2f\n", "25-lane (18 bpermute + ~22 VALU)", waves, mw, mw * 1e3 / iters, k2_us / (mw * 1e3 / iters));
        const double mw2 = run(wide2_kernel, waves, 2, d, init, iters, nullptr);
        printf("%-34s %8u %10.2f %16.3f %18.2f\n", "25-lane, one-trip theta (26 bperm)", waves, mw2, mw2 * 1e3 / iters, k2_us / (mw2 * 1e3 / iters));
    }
    return 0;
}
