// sponge_host.h — internal launcher interface of the sponge path, shared with the Ed448 protocol glue.
#pragma once
#include "common.h"

namespace capy {

// where the per-item message bytes live on the device
struct MsgView {
    const uint8_t *msgs = nullptr;
    const uint64_t *offsets = nullptr;  // n+1 starts (or null: uniform)
    const uint64_t *lens = nullptr;     // optional n lengths (re-packed batches)
    uint64_t uniform_len = 0, msg_stride = 0;
    const uint32_t *order = nullptr;  // optional processing order (SpongeParams::order)
    bool aligned8 = false;  // every message start is known to be 8-byte aligned
};

// where the per-item keys / passwords live on the device: n fixed-length keys (key_len bytes, key_stride apart) or,
// with key_offsets (n+1 device offsets into keys), one length per item
struct KeyView {
    const uint8_t *keys = nullptr;
    size_t key_len = 0;
    uint64_t key_stride = 0;
    const uint64_t *key_offsets = nullptr;
};
inline KeyView fixed_keys(const uint8_t *keys, size_t key_len, uint64_t key_stride)
{
    KeyView k;
    k.keys = keys;
    k.key_len = key_len;
    k.key_stride = key_stride;
    return k;
}

// Keys / passwords of a host batch on the device.  offsets == nullptr: n keys of key_len bytes; else n+1 host offsets
// (non-decreasing), re-based to the first key on upload.
struct PackedKeys {
    DevBuf data, offs;
    KeyView view;
    uint64_t total = 0;  // key bytes of the batch
    int upload(size_t n, const uint8_t *keys, size_t key_len, const uint64_t *offsets);
};

// ---- sponge_launch.hip
// SHA3-d / cSHAKE over device buffers; sha3_encrypt / sha3_decrypt composition (ke_custom / ka_custom: "SKE" / "SKA",
// "KEMKE" / "KEMKA" for the KEM sponge half)
int sha3_launch(int d, size_t n, const MsgView &m, uint8_t *digests, uint64_t out_stride, hipStream_t s);
int cshake_launch(int d, size_t n, const MsgView &m, size_t l_bits, const uint8_t *fn, size_t fn_len, const uint8_t *cs,
                  size_t cs_len, uint8_t *outs, uint64_t out_stride, hipStream_t s, bool body_has_trailer = false);
int sha3_crypt_dev(bool encrypt, int d, size_t n, const KeyView &pw, uint64_t pws_bytes, const uint8_t *zs, const MsgView &m,
                   uint8_t *tags, int32_t *status, hipStream_t s, const char *ke_custom = "SKE", const char *ka_custom = "SKA");
unsigned sponge_debug_flags();  // the A/B bits of capy_set_sponge_lanes
// Dispatch order of the 64-item groups of a length-sorted ragged batch: rank-major over the full neighbourhoods of
// 4096 items -- first every neighbourhood's longest group, then every second-longest, ... (longest-processing-time first
// for the grid as a whole, while each group still reads from one neighbourhood).  Shared by the device sort
// (sponge_launch.hip) and the host sort of PackedBatch::upload (workspace.hip).
constexpr int ORDER_CHUNK_SHIFT = 12;
__host__ __device__ __forceinline__ uint32_t order_spread(uint32_t pos, uint64_t n)
{
    const uint32_t full = (uint32_t)(n >> ORDER_CHUNK_SHIFT);  // complete neighbourhoods
    const uint32_t c = pos >> ORDER_CHUNK_SHIFT;
    if (c >= full) return pos;  // partial last neighbourhood: plain sorted order, at the end
    const uint32_t r = (pos >> 6) & 63u;
    return ((r * full + c) << 6) + (pos & 63u);
}
// ---- workspace.hip
hipError_t copy_rows_out(uint8_t *dst, size_t row, const DevBuf &b, size_t stride, size_t n);

MsgView view_of(const PackedBatch &b);
MsgView view_dev(const uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride);

// kmac_xof over device buffers (see sponge.hip). out_mode 0: out_len bytes per item at outs + i*out_stride;
// out_mode 1: keystream XOR into the message buffer (absorb_body must be false). mask: optional per-item enable.
int kmac_launch(int d, size_t n, const KeyView &kv, const MsgView &m,
                bool absorb_body, const uint8_t *custom, size_t custom_len, int out_mode, uint8_t *outs,
                uint64_t out_stride, size_t out_len, const int32_t *mask, hipStream_t s);

// tag + keystream XOR of the encryptable traits (see sponge.hip); ke at keka + i*stride, ka right behind it
int symmetric_crypt_dev(bool encrypt, int d, size_t n, const uint8_t *keka, size_t key_len, uint64_t keka_stride,
                        const MsgView &m, uint8_t *tags, size_t tag_len, const char *ke_custom, const char *ka_custom,
                        int32_t *status, hipStream_t s);

// status[i] = (a_i == b_i) ? CAPY_ITEM_OK : CAPY_ITEM_FAIL
void tag_compare_launch(const uint8_t *a, uint64_t a_stride, const uint8_t *b, uint64_t b_stride, uint32_t tag_len,
                        int32_t *status, size_t n, hipStream_t s);

}  // namespace capy
