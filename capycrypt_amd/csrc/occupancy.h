// occupancy.h — one macro shared by the sponge and the Ed448 kernels.
//
// Kernels that the launchers take only for batches of AT MOST ONE WAVE PER SIMD are latency-bound: the launch takes one wave's
// chain, and twice that if the dispatcher puts two of the kernel's waves on one SIMD -- which it does behind a launch whose
// waves end staggered (the SIMDs that free first are filled first; profiles/r04_placement.txt, r04_ed448_remainder.txt).
// amdgpu_waves_per_eu(1, MAXW) makes the compiler round the register count in the kernel descriptor up so that at most MAXW
// waves of THIS kernel fit on a SIMD (MAXW = 1: 257+ VGPRs); waves of other kernels still share it if their registers fit.
#pragma once
#define CAPY_WAVES_PER_SIMD(MAXW) __attribute__((amdgpu_waves_per_eu(1, (MAXW))))
