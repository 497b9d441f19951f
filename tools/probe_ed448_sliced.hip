// probe_ed448_sliced.hip — latency of ONE Goldilocks field multiplication when a field element is spread over 16 lanes
// (one 28-bit limb per lane, four independent elements per wave64) against the in-lane form of ed448_dev.h (one element
// per lane, 16 limbs in 16 VGPRs).  The in-lane form is what the batched kernels use: ~340 VALU instructions per
// multiplication, all in one lane's dependency chains -- a single scalar multiplication is ~1.35 M serial instructions
// (3 ms).  The sliced form broadcasts a_i within each 16-lane row (ds_swizzle), keeps b * x^i mod p rotating through the
// row (two DPP moves and an add per step: x^16 = x^8 + 1) and accumulates one 64-bit column per lane.
//
// Checks the sliced product against fe_mul bit for bit (canonical form) on random operands, then times a dependent chain
// x <- x * y of both forms at one wave per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I capycrypt_amd/csrc -I include -o tools/probe_ed448_sliced tools/probe_ed448_sliced.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#include "ed448_dev.h"

using namespace capy;

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                           \
        }                                                                       \
    } while (0)

// ---- row primitives (16-lane rows)
template <int N>
__device__ __forceinline__ uint32_t row_ror(uint32_t v)  // lane k <- lane (k - N) mod 16
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xF, 0xF, true);
}
template <int I>
__device__ __forceinline__ uint32_t row_bcast(uint32_t v)  // every lane of a row <- lane I of that row
{
    // ds_swizzle bit mode: lane' = ((lane & and) | or) ^ xor inside each group of 32; and = 0x10 keeps the row
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x10 | (I << 5));
}

// Per-lane masks for B_i = b * x^i mod p (x = 2^28, x^16 = x^8 + 1), all taken from b directly so that the 16 steps do
// not depend on each other.  With R_n = b rotated up by n limbs (lane k <- lane (k - n) mod 16), limb j of b sits at
// e = i + j:   e < 16: position e (that is R_i);   16 <= e < 24: positions e - 16 (R_i again) and e - 8;
// 24 <= e: x^(e-8) wraps a second time, positions e - 16 twice and e - 24.  In lane terms:
//   i <= 8:  B_i = R_i + (R_(i+8) & lanes [8, 8+i))
//   i >  8:  B_i = R_i + (R_(i-8) & (lanes [8,16) | lanes [0, i-8))) + (R_i & lanes [8, i))
struct RowConst {
    uint32_t m8, m89, hi8;   // lane 8 / lanes 8,9 / lanes 8..15
    uint32_t wa[16];         // mask of the R_(i+8) term
    uint32_t wb[16];         // mask of the doubled R_i term (i > 8)
};
__device__ __forceinline__ RowConst row_consts()
{
    const uint32_t l = threadIdx.x & 15;
    RowConst c;
    c.m8 = l == 8 ? 0xffffffffu : 0u;
    c.m89 = (l == 8 || l == 9) ? 0xffffffffu : 0u;
    c.hi8 = l >= 8 ? 0xffffffffu : 0u;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        const bool a = i <= 8 ? (l >= 8 && l < 8 + i) : (l >= 8 || l < i - 8);
        const bool b = i > 8 && l >= 8 && l < i;
        c.wa[i] = a ? 0xffffffffu : 0u;
        c.wb[i] = b ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(c.wa[i]), "+v"(c.wb[i]));  // keep them as AND operands (not v_cndmask on re-derived conditions)
    }
    asm volatile("" : "+v"(c.m8), "+v"(c.m89), "+v"(c.hi8));
    return c;
}

template <int I>
__device__ __forceinline__ uint32_t b_shift(uint32_t b, const RowConst &rc)  // (b * x^I mod p), straight from b
{
    if constexpr (I == 0)
        return b;
    else if constexpr (I < 8)
        return row_ror<I>(b) + (row_ror<I + 8>(b) & rc.wa[I]);
    else if constexpr (I == 8)
        return row_ror<8>(b) + (b & rc.wa[8]);
    else {
        const uint32_t r = row_ror<I>(b);
        return r + (row_ror<I - 8>(b) & rc.wa[I]) + (r & rc.wb[I]);
    }
}

// limbs of a, b <= 2^29 in (B_i limbs <= 4 x, 16 terms per column: 64 La Lb < 2^64), limbs <= 2^28 + 6 out
// CHAIN = true: B_(i+1) from B_i (3 instructions per step, one dependency chain); false: every B_i from b (more
// instructions, no chain)
template <bool CHAIN>
__device__ __forceinline__ uint32_t sl_mul(uint32_t a, uint32_t b, const RowConst &rc)
{
    // all sixteen broadcasts first: they are LDS-pipe operations whose latencies overlap
    uint32_t ai[16];
    ai[0] = row_bcast<0>(a);
    ai[1] = row_bcast<1>(a);
    ai[2] = row_bcast<2>(a);
    ai[3] = row_bcast<3>(a);
    ai[4] = row_bcast<4>(a);
    ai[5] = row_bcast<5>(a);
    ai[6] = row_bcast<6>(a);
    ai[7] = row_bcast<7>(a);
    ai[8] = row_bcast<8>(a);
    ai[9] = row_bcast<9>(a);
    ai[10] = row_bcast<10>(a);
    ai[11] = row_bcast<11>(a);
    ai[12] = row_bcast<12>(a);
    ai[13] = row_bcast<13>(a);
    ai[14] = row_bcast<14>(a);
    ai[15] = row_bcast<15>(a);
    uint32_t B[16];
    if constexpr (CHAIN) {
        B[0] = b;
#pragma unroll
        for (int i = 1; i < 16; i++) B[i] = row_ror<1>(B[i - 1]) + (row_ror<9>(B[i - 1]) & rc.m8);
    } else {
        B[0] = b_shift<0>(b, rc);
        B[1] = b_shift<1>(b, rc);
        B[2] = b_shift<2>(b, rc);
        B[3] = b_shift<3>(b, rc);
        B[4] = b_shift<4>(b, rc);
        B[5] = b_shift<5>(b, rc);
        B[6] = b_shift<6>(b, rc);
        B[7] = b_shift<7>(b, rc);
        B[8] = b_shift<8>(b, rc);
        B[9] = b_shift<9>(b, rc);
        B[10] = b_shift<10>(b, rc);
        B[11] = b_shift<11>(b, rc);
        B[12] = b_shift<12>(b, rc);
        B[13] = b_shift<13>(b, rc);
        B[14] = b_shift<14>(b, rc);
        B[15] = b_shift<15>(b, rc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint64_t acc0 = 0, acc1 = 0;  // two chains of eight
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        acc0 += (uint64_t)ai[i] * B[i];
        acc1 += (uint64_t)ai[i + 1] * B[i + 1];
    }
    const uint64_t acc = acc0 + acc1;
    const uint32_t lo = (uint32_t)acc & M28, mid = (uint32_t)(acc >> 28) & M28, hi = (uint32_t)(acc >> 56);
    uint32_t x = lo + row_ror<1>(mid) + row_ror<2>(hi) + (row_ror<9>(mid) & rc.m8) + (row_ror<10>(hi) & rc.m89);
    const uint32_t c = x >> 28;
    x = (x & M28) + row_ror<1>(c) + (row_ror<9>(c) & rc.m8);
    return x;
}

// ---- kernels
__global__ void lane_id_kernel(uint32_t *out)
{
    const uint32_t v = threadIdx.x;
    out[threadIdx.x] = row_ror<1>(v);
    out[64 + threadIdx.x] = row_ror<9>(v);
    out[128 + threadIdx.x] = row_bcast<5>(v);
}

// items: 4 per wave; operands as 16 limbs each
template <bool CHAIN>
__global__ __launch_bounds__(64) void sliced_chain_kernel(const uint32_t *a, const uint32_t *b, uint32_t *out, int iters)
{
    const uint64_t idx = (uint64_t)blockIdx.x * 64 + threadIdx.x;  // = item * 16 + limb
    const RowConst rc = row_consts();
    uint32_t x = a[idx];
    const uint32_t y = b[idx];
    for (int i = 0; i < iters; i++) x = sl_mul<CHAIN>(x, y, rc);
    out[idx] = x;
}

__global__ __launch_bounds__(64) void inlane_chain_kernel(const uint32_t *a, const uint32_t *b, uint32_t *out, int iters, uint64_t n)
{
    const uint64_t item = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (item >= n) return;
    Fe x, y;
    for (int i = 0; i < 16; i++) {
        x.l[i] = a[item * 16 + i];
        y.l[i] = b[item * 16 + i];
    }
    for (int i = 0; i < iters; i++) x = fe_mul(x, y);
    fe_canon(x);
    for (int i = 0; i < 16; i++) out[item * 16 + i] = x.l[i];
}

__global__ void canon_kernel(uint32_t *v, uint64_t n)
{
    const uint64_t item = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (item >= n) return;
    Fe x;
    for (int i = 0; i < 16; i++) x.l[i] = v[item * 16 + i];
    fe_canon(x);
    for (int i = 0; i < 16; i++) v[item * 16 + i] = x.l[i];
}

int main()
{
    uint32_t *d_ids;
    CK(hipMalloc(&d_ids, 192 * 4));
    hipLaunchKernelGGL(lane_id_kernel, dim3(1), dim3(64), 0, 0, d_ids);
    std::vector<uint32_t> ids(192);
    CK(hipMemcpy(ids.data(), d_ids, 192 * 4, hipMemcpyDeviceToHost));
    printf("row_ror<1>  lanes 0..17: ");
    for (int i = 0; i < 18; i++) printf("%u ", ids[i]);
    printf("\nrow_ror<9>  lanes 0..17: ");
    for (int i = 0; i < 18; i++) printf("%u ", ids[64 + i]);
    printf("\nrow_bcast<5> lanes 0,15,16,31,32,63: %u %u %u %u %u %u\n", ids[128], ids[143], ids[144], ids[159], ids[160], ids[191]);

    const uint64_t n = 4096;  // items
    std::vector<uint32_t> ha(n * 16), hb(n * 16);
    uint64_t s = 0x9E3779B97F4A7C15ULL;
    auto rnd = [&]() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return (uint32_t)(s >> 20);
    };
    for (auto &v : ha) v = rnd() & M28;
    for (auto &v : hb) v = rnd() & M28;
    // a few extreme operands: all limbs at the lazy bound
    for (int i = 0; i < 16; i++) {
        ha[i] = (1u << 29) - 12345;
        hb[i] = (1u << 29) - 54321;
        ha[16 + i] = M28;
        hb[16 + i] = M28;
    }
    uint32_t *da, *db, *o1, *o2;
    CK(hipMalloc(&da, n * 64));
    CK(hipMalloc(&db, n * 64));
    CK(hipMalloc(&o1, n * 64));
    CK(hipMalloc(&o2, n * 64));
    CK(hipMemcpy(da, ha.data(), n * 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), n * 64, hipMemcpyHostToDevice));
    for (int iters : {1, 2, 7}) {
        if (iters == 2)
            hipLaunchKernelGGL(sliced_chain_kernel<true>, dim3(n / 4), dim3(64), 0, 0, da, db, o1, iters);
        else
            hipLaunchKernelGGL(sliced_chain_kernel<false>, dim3(n / 4), dim3(64), 0, 0, da, db, o1, iters);
        hipLaunchKernelGGL(canon_kernel, dim3(n / 64), dim3(64), 0, 0, o1, n);
        hipLaunchKernelGGL(inlane_chain_kernel, dim3(n / 64), dim3(64), 0, 0, da, db, o2, iters, n);
        std::vector<uint32_t> r1(n * 16), r2(n * 16);
        CK(hipMemcpy(r1.data(), o1, n * 64, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r2.data(), o2, n * 64, hipMemcpyDeviceToHost));
        uint64_t bad = 0;
        for (uint64_t i = 0; i < n * 16; i++) bad += r1[i] != r2[i];
        printf("chain of %d multiplications, %llu items: %llu limb mismatches\n", iters, (unsigned long long)n, (unsigned long long)bad);
        if (bad) return 2;
    }
    // latency: one wave per SIMD (1024 waves), long dependent chains
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int rep = 0; rep < 2; rep++) {
        float ms;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(sliced_chain_kernel<false>, dim3(1024), dim3(64), 0, 0, da, db, o1, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("sliced, B_i from b   : %d dependent multiplications in %.3f ms = %.1f ns each (4 per wave side by side, 1024 waves)\n", iters, ms,
               ms * 1e6 / iters);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(sliced_chain_kernel<true>, dim3(1024), dim3(64), 0, 0, da, db, o1, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("sliced, B_i from B_i-1: %d dependent multiplications in %.3f ms = %.1f ns each\n", iters, ms, ms * 1e6 / iters);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(inlane_chain_kernel, dim3(1024), dim3(64), 0, 0, da, db, o2, iters / 10, (uint64_t)4096);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("in-lane: %d dependent multiplications in %.3f ms = %.1f ns each (64 per wave, 64 waves)\n", iters / 10, ms,
               ms * 1e6 / (iters / 10));
    }
    return 0;
}
