// abi_sponge.hip — the C-ABI entry points of the sponge path (include/capyhip.h: SHA3 / cSHAKE / KMACXOF / sha3_encrypt /
// sha3_decrypt / KEM sponge half): argument checks, sharding, host <-> device staging, then sponge_launch.hip.
#include <string.h>
#include <algorithm>
#include <atomic>
#include <string>
#include <vector>
#include "common.h"
#include "sponge_host.h"

namespace capy {

// cshake(x, l, "", "", d) on device buffers (shake_functions.rs:59-61, see capy_cshake_batch): y_i = x_i || 04 || sfx || pad,
// where sfx is the SHA3 suffix the dropped shake() call left behind (86 when the framed length is 135 mod 136, else 06)
// and pad is its pad10*1 up to the SHA3-d rate r1 (nothing when already a multiple).  One wave per item: the bytes are
// copied, lane 0 writes the trailer.  dst_off == nullptr: y_i at dst + i * dst_stride (uniform lengths).
__host__ __device__ inline uint64_t cshake_empty_trailer(uint64_t len, uint64_t w, uint64_t r1, uint8_t *sfx)
{
    uint64_t L = w + len + 1;  // bytepad(encode_string("") || encode_string(""), w) is exactly one block of w bytes
    *sfx = (136 - L % 136) == 1 ? 0x86 : 0x06;
    L += 1;
    return 2 + (L % r1 ? r1 - L % r1 : 0);
}
__global__ __launch_bounds__(64) void cshake_empty_trailer_kernel(uint8_t *dst, const uint64_t *dst_off, uint64_t dst_stride,
                                                                  const uint8_t *src, const uint64_t *src_off, uint64_t uniform_len,
                                                                  uint64_t src_stride, uint64_t n, uint32_t w, uint32_t r1)
{
    const uint64_t i = blockIdx.x;
    if (i >= n) return;
    const uint64_t s0 = src_off ? src_off[i] : i * src_stride, len = src_off ? src_off[i + 1] - s0 : uniform_len;
    uint8_t *y = dst + (dst_off ? dst_off[i] : i * dst_stride);
    for (uint64_t j = threadIdx.x; j < len; j += 64) y[j] = src[s0 + j];
    if (threadIdx.x == 0) {
        uint8_t sfx;
        const uint64_t t = cshake_empty_trailer(len, w, r1, &sfx);
        y[len] = 0x04;
        y[len + 1] = sfx;
        for (uint64_t j = 2; j < t; j++) y[len + j] = 0;
        if (t > 2) y[len + t - 1] = 0x80;
    }
}

}  // namespace capy

using namespace capy;

extern "C" {

// ---------------------------------------------------------------- SHA3
int capy_sha3_batch_dev(int d, size_t n, const uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len,
                        uint64_t msg_stride, uint8_t *digests, void *stream)
{
    if (n) {
        CAPY_REQUIRE(digests, "digests");
        CAPY_REQUIRE(msgs_ok(msgs, offsets, uniform_len), "msgs");
    }
    return sha3_launch(d, n, view_dev(msgs, offsets, uniform_len, msg_stride), digests, (uint64_t)(d / 8),
                       (hipStream_t)stream);
}

int capy_sha3_batch(int d, size_t n, const uint8_t *msgs, const uint64_t *offsets, uint8_t *digests)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (n == 0) return CAPY_OK;
    if (!offsets || !digests) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, offsets, capy_sha3_batch(d, count, msgs, offsets + first, digests + first * (size_t)(d / 8)));
    PackedBatch b;
    int rc = b.upload(n, msgs, offsets);
    if (rc) return rc;
    DevBuf out;
    const size_t dl = d / 8;
    CAPY_HIP(out.alloc(n * dl));
    rc = sha3_launch(d, n, view_of(b), out.as<uint8_t>(), dl, nullptr);
    if (rc) return rc;
    CAPY_HIP(out.get(digests, n * dl));
    return CAPY_OK;
}

// ---------------------------------------------------------------- cSHAKE / KMACXOF
int capy_cshake_batch(int d, size_t n, const uint8_t *xs, const uint64_t *offsets, size_t l_bits,
                      const uint8_t *fn_name, size_t fn_len, const uint8_t *custom, size_t custom_len, uint8_t *outs)
{
    if (l_bits / 8 > CAPY_MAX_OUT_LEN) return fail(CAPY_ERR_ARG, "output longer than 2^32 - 1 bytes per item (l_bits >= 2^35)");
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (n == 0) return CAPY_OK;
    if (!offsets || !outs) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, offsets, capy_cshake_batch(d, count, xs, offsets + first, l_bits, fn_name, fn_len, custom, custom_len,
                                             outs + first * (l_bits / 8)));
    PackedBatch b;
    int rc;
    const bool empty_ns = fn_len == 0 && custom_len == 0;
    if (empty_ns) {
        // cshake(x, l, "", "", d), shake_functions.rs:59-61: the reference runs shake() on the framed buffer, drops its
        // digest and KEEPS its mutation -- the SHA3 suffix (06, or 86 when the length is 135 mod 136) and, when the
        // result is not a multiple of the SHA3-d rate (1600 - 2d)/8, pad10*1 up to it -- and then absorbs that buffer
        // at capacity d.  Unreachable through the public API (kmac_xof passes N = "KMAC"); reproduced here by giving
        // every message its trailer on the host and absorbing it without a further suffix.
        const uint64_t w = (1600 - (uint64_t)d) / 8, r1 = (1600 - 2 * (uint64_t)d) / 8;
        std::vector<uint8_t> ys;
        std::vector<uint64_t> yoff(n + 1, 0);
        for (size_t i = 0; i < n; i++) {
            if (offsets[i + 1] < offsets[i]) return fail(CAPY_ERR_ARG, "offsets must be non-decreasing");
            const uint64_t len = offsets[i + 1] - offsets[i];
            if (len) ys.insert(ys.end(), xs + offsets[i], xs + offsets[i + 1]);
            ys.push_back(0x04);
            uint64_t L = w + len + 1;  // bytepad(encode_string("") || encode_string(""), w) is exactly one block of w bytes
            ys.push_back((136 - L % 136) == 1 ? 0x86 : 0x06);
            L += 1;
            if (L % r1) {
                const uint64_t q = r1 - L % r1;
                ys.insert(ys.end(), q, 0);
                ys.back() = 0x80;
            }
            yoff[i + 1] = ys.size();
        }
        rc = b.upload(n, ys.data(), yoff.data());
    } else {
        rc = b.upload(n, xs, offsets);
    }
    if (rc) return rc;
    const size_t ol = l_bits / 8, os = (ol + 7) & ~(size_t)7;
    DevBuf out;
    CAPY_HIP(out.alloc(n * os));
    rc = cshake_launch(d, n, view_of(b), l_bits, fn_name, fn_len, custom, custom_len, out.as<uint8_t>(), os, nullptr,
                       empty_ns);
    if (rc) return rc;
    if (ol) CAPY_HIP(copy_rows_out(outs, ol, out, os, n));
    return CAPY_OK;
}

int capy_cshake_batch_dev(int d, size_t n, const uint8_t *xs, const uint64_t *offsets, uint64_t uniform_len,
                          uint64_t msg_stride, size_t l_bits, const uint8_t *fn_name, size_t fn_len,
                          const uint8_t *custom, size_t custom_len, uint8_t *outs, uint64_t out_stride, void *stream)
{
    if (l_bits / 8 > CAPY_MAX_OUT_LEN) return fail(CAPY_ERR_ARG, "output longer than 2^32 - 1 bytes per item (l_bits >= 2^35)");
    if (n && !outs) return fail(CAPY_ERR_ARG, "null argument");
    if (n) CAPY_REQUIRE(msgs_ok(xs, offsets, uniform_len), "xs");
    if (out_stride < l_bits / 8) return fail(CAPY_ERR_ARG, "out_stride shorter than the output");
    if (fn_len == 0 && custom_len == 0 && n && valid_d(d)) {
        // the N = S = "" corner: every message gets its trailer in a scratch copy, absorbed without a further suffix.
        // Synchronous (ragged batches read their offsets back to size the copy, and the scratch copy must outlive the
        // launch): crate-internal and unreachable through kmac_xof, so not a path worth a stream-ordered allocator
        hipStream_t s = (hipStream_t)stream;
        // this corner synchronises (twice for ragged batches): not capturable -- say so instead of breaking the capture
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            return fail(CAPY_ERR_UNSUPPORTED, "cshake with empty N and S synchronises the stream: not available under stream capture");
        (void)hipGetLastError();
        const uint64_t w = (1600 - (uint64_t)d) / 8, r1 = (1600 - 2 * (uint64_t)d) / 8;
        DevBuf y, meta;  // meta: n + 1 starts of the copies, then their n lengths
        // Kernels on the CALLER's stream read y / meta; ~DevBuf only waits for the thread's default stream before the block goes
        // back to the cache.  Declared after the buffers, so it runs before their destructors on EVERY return path.
        struct SyncOnExit {
            hipStream_t s;
            ~SyncOnExit() { (void)hipStreamSynchronize(s); }
        } sync_on_exit{s};
        uint64_t ylen = 0, ystride = 0;
        uint8_t sfx;
        MsgView v;
        if (offsets) {
            std::vector<uint64_t> off(n + 1), m(2 * n + 1, 0);
            CAPY_HIP(hipMemcpyAsync(off.data(), offsets, (n + 1) * 8, hipMemcpyDeviceToHost, s));
            CAPY_HIP(hipStreamSynchronize(s));
            for (size_t i = 0; i < n; i++) {
                if (off[i + 1] < off[i]) return fail(CAPY_ERR_ARG, "offsets must be non-decreasing");
                const uint64_t len = off[i + 1] - off[i];
                m[n + 1 + i] = len + cshake_empty_trailer(len, w, r1, &sfx);
                m[i + 1] = m[i] + ((m[n + 1 + i] + 7) & ~7ull);  // 8-byte aligned starts
            }
            CAPY_HIP(meta.alloc((2 * n + 1) * 8));
            CAPY_HIP(meta.put(m.data(), (2 * n + 1) * 8));  // (a small buffer may live in the pinned arena: plain memory)
            CAPY_HIP(y.alloc(m[n] + 8));
            v = view_dev(y.as<uint8_t>(), meta.as<uint64_t>(), 0, 0);
            v.lens = meta.as<uint64_t>() + n + 1;
            v.aligned8 = true;
        } else {
            ylen = uniform_len + cshake_empty_trailer(uniform_len, w, r1, &sfx);
            ystride = (ylen + 7) & ~7ull;
            CAPY_HIP(y.alloc(n * ystride + 8));
            v = view_dev(y.as<uint8_t>(), nullptr, ylen, ystride);
        }
        hipLaunchKernelGGL(cshake_empty_trailer_kernel, dim3((unsigned)n), dim3(64), 0, s, y.as<uint8_t>(),
                           offsets ? meta.as<uint64_t>() : nullptr, ystride, xs, offsets, uniform_len,
                           msg_stride ? msg_stride : uniform_len, (uint64_t)n, (uint32_t)w, (uint32_t)r1);
        CAPY_HIP(hipGetLastError());
        return cshake_launch(d, n, v, l_bits, fn_name, 0, custom, 0, outs, out_stride, s, true);  // (synchronised on exit)
    }
    return cshake_launch(d, n, view_dev(xs, offsets, uniform_len, msg_stride), l_bits, fn_name, fn_len, custom,
                         custom_len, outs, out_stride, (hipStream_t)stream);
}

int capy_kmac_xof_batch_dev(int d, size_t n, const uint8_t *keys, size_t key_len, uint64_t key_stride,
                            const uint64_t *key_offsets, const uint8_t *xs, const uint64_t *offsets, uint64_t uniform_len,
                            uint64_t msg_stride, size_t l_bits, const uint8_t *custom, size_t custom_len, uint8_t *outs,
                            uint64_t out_stride, void *stream)
{
    if (l_bits / 8 > CAPY_MAX_OUT_LEN) return fail(CAPY_ERR_ARG, "output longer than 2^32 - 1 bytes per item (l_bits >= 2^35)");
    if (n) {
        CAPY_REQUIRE(outs, "outs");
        CAPY_REQUIRE(out_stride >= l_bits / 8, "out_stride shorter than the output");
        CAPY_REQUIRE(keys_ok(keys, key_len, key_offsets), "keys");
        CAPY_REQUIRE(msgs_ok(xs, offsets, uniform_len), "xs");
    }
    KeyView kv = fixed_keys(keys, key_len, key_stride);
    kv.key_offsets = key_offsets;
    return kmac_launch(d, n, kv, view_dev(xs, offsets, uniform_len, msg_stride), true, custom, custom_len, 0, outs,
                       out_stride, l_bits / 8, nullptr, (hipStream_t)stream);
}

int capy_kmac_xof_batch(int d, size_t n, const uint8_t *keys, size_t key_len, const uint64_t *key_offsets,
                        const uint8_t *xs, const uint64_t *offsets, size_t l_bits, const uint8_t *custom, size_t custom_len,
                        uint8_t *outs)
{
    if (l_bits / 8 > CAPY_MAX_OUT_LEN) return fail(CAPY_ERR_ARG, "output longer than 2^32 - 1 bytes per item (l_bits >= 2^35)");
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (n == 0) return CAPY_OK;
    if (!outs) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, offsets,
               capy_kmac_xof_batch(d, count, key_offsets ? keys : keys + first * key_len, key_len,
                                   key_offsets ? key_offsets + first : nullptr, xs, offsets ? offsets + first : nullptr,
                                   l_bits, custom, custom_len, outs + first * (l_bits / 8)));
    std::vector<uint64_t> zero_off;
    if (!offsets) {  // every x_i empty
        zero_off.assign(n + 1, 0);
        offsets = zero_off.data();
    }
    PackedBatch b;
    int rc = b.upload(n, xs, offsets);
    if (rc) return rc;
    PackedKeys k;
    rc = k.upload(n, keys, key_len, key_offsets);
    if (rc) return rc;
    DevBuf out;
    const size_t ol = l_bits / 8, os = (ol + 7) & ~(size_t)7;
    CAPY_HIP(out.alloc(n * os));
    rc = kmac_launch(d, n, k.view, view_of(b), true, custom, custom_len, 0, out.as<uint8_t>(), os, ol, nullptr, nullptr);
    if (rc) return rc;
    if (ol) CAPY_HIP(copy_rows_out(outs, ol, out, os, n));
    return CAPY_OK;
}

// ---------------------------------------------------------------- sha3_encrypt / sha3_decrypt
static KeyView dev_keys(const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets)
{
    KeyView kv = fixed_keys(pws, pw_len, pw_len);
    kv.key_offsets = pw_offsets;
    return kv;
}

int capy_sha3_encrypt_batch_dev(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                                uint64_t pws_bytes, const uint8_t *zs, uint8_t *msgs, const uint64_t *offsets,
                                uint64_t uniform_len, uint64_t msg_stride, uint8_t *tags, void *stream)
{
    if (n) {
        CAPY_REQUIRE(zs && tags, "zs / tags");
        CAPY_REQUIRE(keys_ok(pws, pw_len, pw_offsets), "pws");
        CAPY_REQUIRE(msgs_ok(msgs, offsets, uniform_len), "msgs");
    }
    return sha3_crypt_dev(true, d, n, dev_keys(pws, pw_len, pw_offsets), pws_bytes, zs,
                          view_dev(msgs, offsets, uniform_len, msg_stride), tags, nullptr, (hipStream_t)stream);
}

int capy_sha3_decrypt_batch_dev(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                                uint64_t pws_bytes, const uint8_t *zs, uint8_t *msgs, const uint64_t *offsets,
                                uint64_t uniform_len, uint64_t msg_stride, const uint8_t *tags, int32_t *status,
                                void *stream)
{
    if (n) {
        CAPY_REQUIRE(zs && tags && status, "zs / tags / status");
        CAPY_REQUIRE(keys_ok(pws, pw_len, pw_offsets), "pws");
        CAPY_REQUIRE(msgs_ok(msgs, offsets, uniform_len), "msgs");
    }
    return sha3_crypt_dev(false, d, n, dev_keys(pws, pw_len, pw_offsets), pws_bytes, zs,
                          view_dev(msgs, offsets, uniform_len, msg_stride), const_cast<uint8_t *>(tags), status,
                          (hipStream_t)stream);
}

static int sha3_crypt_host(bool encrypt, int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                           const uint8_t *zs, uint8_t *msgs, const uint64_t *offsets, uint8_t *tags, int32_t *status,
                           const char *ke_custom = "SKE", const char *ka_custom = "SKA")
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (n == 0) return CAPY_OK;
    if (!offsets || !zs || !tags) return fail(CAPY_ERR_ARG, "null argument");
    CAPY_SHARD(n, offsets,
               sha3_crypt_host(encrypt, d, count, pw_offsets ? pws : pws + first * pw_len, pw_len,
                               pw_offsets ? pw_offsets + first : nullptr, zs + first * 512, msgs, offsets + first,
                               tags + first * 64, status ? status + first : nullptr, ke_custom, ka_custom));
    PackedBatch b;
    b.msgs.secret = !encrypt;  // decryption leaves plaintext in the staging block: zeroed before it is reused
    int rc = b.upload(n, msgs, offsets);
    if (rc) return rc;
    PackedKeys dpw;
    rc = dpw.upload(n, pws, pw_len, pw_offsets);
    if (rc) return rc;
    DevBuf dz, dtag, dst;
    CAPY_HIP(dz.alloc(n * 512));
    CAPY_HIP(dtag.alloc(n * 64));
    CAPY_HIP(dst.alloc(n * 4));
    CAPY_HIP(dz.put(zs, n * 512));
    if (!encrypt) CAPY_HIP(dtag.put(tags, n * 64));
    rc = sha3_crypt_dev(encrypt, d, n, dpw.view, dpw.total, dz.as<uint8_t>(), view_of(b), dtag.as<uint8_t>(),
                        dst.as<int32_t>(), nullptr, ke_custom, ka_custom);
    if (rc) return rc;
    CAPY_HIP(hipStreamSynchronize(nullptr));
    rc = b.download(n, msgs, offsets);
    if (rc) return rc;
    if (encrypt)
        CAPY_HIP(dtag.get(tags, n * 64));
    else
        CAPY_HIP(dst.get(status, n * 4));
    return CAPY_OK;
}

int capy_sha3_encrypt_batch(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                            const uint8_t *zs, uint8_t *msgs, const uint64_t *offsets, uint8_t *tags)
{
    return sha3_crypt_host(true, d, n, pws, pw_len, pw_offsets, zs, msgs, offsets, tags, nullptr);
}

int capy_sha3_decrypt_batch(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                            const uint8_t *zs, uint8_t *msgs, const uint64_t *offsets, const uint8_t *tags, int32_t *status)
{
    if (!status) return fail(CAPY_ERR_ARG, "null status");
    return sha3_crypt_host(false, d, n, pws, pw_len, pw_offsets, zs, msgs, offsets, const_cast<uint8_t *>(tags), status);
}

// ---------------------------------------------------------------- KEMEncryptable, sponge half
int capy_kem_sponge_encrypt_batch(int d, size_t n, const uint8_t *secrets, size_t secret_len, const uint8_t *zs,
                                  uint8_t *msgs, const uint64_t *offsets, uint8_t *tags)
{
    return sha3_crypt_host(true, d, n, secrets, secret_len, nullptr, zs, msgs, offsets, tags, nullptr, "KEMKE", "KEMKA");
}

int capy_kem_sponge_decrypt_batch(int d, size_t n, const uint8_t *secrets, size_t secret_len, const uint8_t *zs,
                                  uint8_t *msgs, const uint64_t *offsets, const uint8_t *tags, int32_t *status)
{
    if (!status) return fail(CAPY_ERR_ARG, "null status");
    return sha3_crypt_host(false, d, n, secrets, secret_len, nullptr, zs, msgs, offsets, const_cast<uint8_t *>(tags),
                           status, "KEMKE", "KEMKA");
}

int capy_kem_sponge_encrypt_batch_dev(int d, size_t n, const uint8_t *secrets, size_t secret_len, const uint8_t *zs,
                                      uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride,
                                      uint8_t *tags, void *stream)
{
    if (n) {
        CAPY_REQUIRE(zs && tags, "zs / tags");
        CAPY_REQUIRE(keys_ok(secrets, secret_len, nullptr), "secrets");
        CAPY_REQUIRE(msgs_ok(msgs, offsets, uniform_len), "msgs");
    }
    return sha3_crypt_dev(true, d, n, dev_keys(secrets, secret_len, nullptr), 0, zs, view_dev(msgs, offsets, uniform_len, msg_stride),
                          tags, nullptr, (hipStream_t)stream, "KEMKE", "KEMKA");
}

int capy_kem_sponge_decrypt_batch_dev(int d, size_t n, const uint8_t *secrets, size_t secret_len, const uint8_t *zs,
                                      uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride,
                                      const uint8_t *tags, int32_t *status, void *stream)
{
    if (n) {
        CAPY_REQUIRE(zs && tags && status, "zs / tags / status");
        CAPY_REQUIRE(keys_ok(secrets, secret_len, nullptr), "secrets");
        CAPY_REQUIRE(msgs_ok(msgs, offsets, uniform_len), "msgs");
    }
    return sha3_crypt_dev(false, d, n, dev_keys(secrets, secret_len, nullptr), 0, zs, view_dev(msgs, offsets, uniform_len, msg_stride),
                          const_cast<uint8_t *>(tags), status, (hipStream_t)stream, "KEMKE", "KEMKA");
}

}  // extern "C"
