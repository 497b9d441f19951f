// One-hot selection as a matrix product on the matrix cores (gfx950 v_mfma_i32_16x16x64_i8): D = A x B with
// A[m][k] = T[k][m] (byte m of table entry k), B[k][n] = (idx[n] == k)  ->  D[m][n] = T[idx[n]][m].
// Checks the operand layouts this relies on:  A: lane l holds row m = l % 16, k-slots (g = l / 16, b = 0..15);
// B: lane l holds column n = l % 16, the same k-slots; D: lane l, register r holds D[4 (l / 16) + r][l % 16].
// Any bijection between k-slots and k works as long as A and B use the same one (here k = 16 g + b).
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/probe_mfma_onehot tools/probe_mfma_onehot.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(const uint8_t *T /* [64][16] */, const uint8_t *idx /* [16] */, int *out /* [64][4] */)
{
    const int l = threadIdx.x, g = l / 16, c = l % 16;
    uint32_t a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    for (int s = 0; s < 16; s++) {
        a[s / 4] |= (uint32_t)T[(16 * g + s) * 16 + c] << (8 * (s % 4));
        b[s / 4] |= (uint32_t)(idx[c] == 16 * g + s) << (8 * (s % 4));
    }
    v4i av = {(int)a[0], (int)a[1], (int)a[2], (int)a[3]}, bv = {(int)b[0], (int)b[1], (int)b[2], (int)b[3]}, acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[l * 4 + r] = acc[r];
}
int main()
{
    uint8_t hT[64 * 16], hi[16];
    srand(7);
    for (auto &v : hT) v = (uint8_t)rand();
    for (auto &v : hi) v = (uint8_t)(rand() % 64);
    uint8_t *dT, *di;
    int *dout, ho[256];
    (void)hipMalloc(&dT, sizeof(hT));
    (void)hipMalloc(&di, sizeof(hi));
    (void)hipMalloc(&dout, sizeof(ho));
    (void)hipMemcpy(dT, hT, sizeof(hT), hipMemcpyHostToDevice);
    (void)hipMemcpy(di, hi, sizeof(hi), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dT, di, dout);
    (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            const int want = (int8_t)hT[hi[l % 16] * 16 + 4 * (l / 16) + r];
            if (ho[l * 4 + r] != want) {
                if (bad < 8) printf("lane %d reg %d: got %d want %d\n", l, r, ho[l * 4 + r], want);
                bad++;
            }
        }
    printf(bad ? "MISMATCH in %d of 256 outputs\n" : "one-hot MFMA select: all 256 outputs as expected (%d bad)\n", bad);
    return bad != 0;
}
