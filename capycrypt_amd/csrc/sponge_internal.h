// sponge_internal.h — what sponge_launch.hip (framing, kernel choice and schedules of the digest / XOF / keystream launches) and
// sponge_crypt.hip (sha3_encrypt / sha3_decrypt composition, the fused kernels' schedules) share.  Host code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "sponge_params.h"

namespace capy {

struct Framing {
    int rw;             // absorb words per block
    uint32_t stride;    // reference `r`
    uint32_t sq_words;  // squeeze words per block
};
Framing cshake_framing(int d);
// fold the batch-shared cSHAKE prefix bytepad(encode_string(N) || encode_string(S), w) into p.init_state (or pre_host at D224)
void cshake_prefix(int d, const uint8_t *fn, size_t fn_len, const uint8_t *cs, size_t cs_len, const Framing &f, SpongeParams &p,
                   std::vector<uint8_t> &pre_host);
// per-item KMAC head = bytepad(encode_string(K), w): hdr bytes, head length
void kmac_head(int d, size_t key_len, SpongeParams &p);

unsigned device_simds();      // SIMDs of the current device (4 per compute unit)
unsigned sponge_debug_flags();
bool fused_enabled();          // capy_set_sponge_lanes bit 16 clear
size_t wide_max_items();
       // largest batch of the one-wave-per-item kernels

// test hook (capy_debug_last_sponge_kernel): what the calling thread's last launch took
void note_kernel(int kind, int launches);
void last_kernel(int *kind, int *launches);

// longest-first processing order for ragged device batches
bool wants_device_order(const uint64_t *offsets, const uint32_t *order, uint64_t n);
int device_order(const uint64_t *offsets, const uint64_t *lens, size_t n, hipStream_t s, const uint32_t **out);

// plan of a rotating-occupancy schedule (sponge_rot.h; sponge_fused1.h): C compute units, Cp of them doubled up per phase,
// P phases, every group doubled up in `a` of them, nb1 / nb2 blocks per phase of a group on its own / doubled up
struct RotPlan {
    uint32_t P, a, C, Cp, G, nb1, nb2;
    uint64_t nf;
};

}  // namespace capy
