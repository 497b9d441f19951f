// Where does global_load_lds_dwordx4 put the 16 bytes of lane l?  (gfx950; used by ed448_algo.h lds_prefetch)
// Each lane loads the uint4 {l, 100 + l, 200 + l, 300 + l} from its own address into LDS at base 0 and base 1024;
// the kernel then copies the first 2 KiB of LDS out dword by dword.  Build: hipcc -O2 --offload-arch=gfx950 -o tools/probe_lds_dma tools/probe_lds_dma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void k(const uint4 *src, uint32_t *out)
{
    __shared__ uint32_t buf[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) buf[i] = 0xdeadbeef;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane), (__attribute__((address_space(3))) void *)&buf[0], 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 64 + lane), (__attribute__((address_space(3))) void *)&buf[256], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = buf[i];
}
int main()
{
    uint4 h[128];
    for (unsigned l = 0; l < 128; l++) h[l] = {l, 1000 + l, 2000 + l, 3000 + l};
    uint4 *d;
    uint32_t *o, ho[1024];
    (void)hipMalloc(&d, sizeof(h));
    (void)hipMalloc(&o, sizeof(ho));
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    (void)hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    for (int i = 0; i < 1024; i += 8) {
        printf("%4d:", i);
        for (int j = 0; j < 8; j++) printf(" %8x", ho[i + j]);
        printf("\n");
    }
    return 0;
}
