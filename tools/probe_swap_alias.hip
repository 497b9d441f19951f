// probe_swap_alias.hip -- does v_permlane32_swap_b32 with ONE register for both operands exchange the halves of the wave?
// (the builtin models two tied operands and copies first; an aliased swap would save that copy and the select behind it)
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe_swap_alias tools/probe_swap_alias.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out)
{
    unsigned x = threadIdx.x * 2654435761u + 17u;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %0\n\ts_nop 1" : "+v"(x));
    out[threadIdx.x] = x;
}
int main()
{
    unsigned *d, h[64];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int ok = 1, same = 1;
    for (unsigned l = 0; l < 64; l++) {
        ok &= h[l] == (l ^ 32) * 2654435761u + 17u;
        same &= h[l] == l * 2654435761u + 17u;
    }
    printf("aliased v_permlane32_swap_b32: %s\n", ok ? "exchanges the halves" : same ? "leaves the register unchanged" : "something else");
    for (unsigned l = 0; l < 64; l += 9) printf("lane %2u: got %08x  own %08x  partner %08x\n", l, h[l], l * 2654435761u + 17u, (l ^ 32) * 2654435761u + 17u);
    return ok ? 0 : 1;
}
