// sponge_launch.h — the sponge kernel instances live in several translation units (built in parallel);
// each exports one launcher.  mode: 0 digest, 1 keystream XOR.
#pragma once
#include <hip/hip_runtime.h>
#include "sponge_params.h"

namespace capy {
// one lane per sponge, latency-tuned instance (launches that cannot fill the chip)
hipError_t launch_sponge_k1_lat(int rw, int mode, const SpongeParams &p, hipStream_t s);
// the same for launches with two waves per SIMD (blocked round, raised priority around the rotation blocks)
hipError_t launch_sponge_k1_lat_paired(int rw, int mode, const SpongeParams &p, hipStream_t s);
// one lane per sponge, issue-tuned instance (many waves per SIMD)
hipError_t launch_sponge_k1_full(int rw, int mode, const SpongeParams &p, hipStream_t s);
// two lanes per sponge (small batches of long messages)
hipError_t launch_sponge_k2(int rw, int mode, const SpongeParams &p, hipStream_t s);
// tag + keystream sponges of sha3_encrypt / sha3_decrypt in one pass (sponge_fused.h); rw in {17, 19, 21}
// one- and two-lane waves side by side, one phase of the rotating schedule (sponge_mixed.h); rw in {9,13,17,18,19,21}
struct MixedParams;
hipError_t launch_sponge_mixed(int rw, const MixedParams &q, unsigned waves, hipStream_t s);
// one phase of the rotating-occupancy schedule for 64 S < n < 128 S sponges (sponge_rot.h); rw in {9,13,17,18,19,21}
struct RotParams;
hipError_t launch_sponge_rot(int rw, const RotParams &q, unsigned cus, size_t lds_bytes, hipStream_t s);
struct FusedParams;
hipError_t launch_sponge_fused(int rw, const FusedParams &fp, hipStream_t s);
// the same with ONE lane per sponge, two lanes per item, for chip-filling batches (sponge_fused1.h); fp.one_lane = the instance
hipError_t launch_sponge_fused1(int rw, const FusedParams &fp, hipStream_t s);
hipError_t launch_sponge_fused1_rot(int rw, const FusedParams &fp, unsigned cus, hipStream_t s);
// very small batches: one item per wave, the sponge spread over the wave's lanes with bit-interleaved Keccak lanes
// (sponge_wide_il.h).  Digests: rw in {9, 13, 17, 18, 19, 21}, digest mode only, no raw prefix bytes; the crypt form takes two
// waves per item (rw in {17, 19, 21}, fp.wide)
hipError_t launch_sponge_il_digest(int rw, const SpongeParams &p, hipStream_t s);
hipError_t launch_sponge_il_crypt(int rw, const FusedParams &fp, hipStream_t s);
// chip-full digest / XOF launches with wave-uniform framing (sponge_uniform.h); rw in {9, 13, 17, 18, 19, 21};
// waves = 1..3: occupancy cap in waves per SIMD (A/B), else none
hipError_t launch_sponge_uniform(int rw, const SpongeParams &p, int waves, hipStream_t s, unsigned sliced_grid = 0);
}  // namespace capy
