// sponge_crypt.hip — the symmetric half of the encryptable traits on device buffers: sha3_encrypt / sha3_decrypt
// (/root/reference/src/sha3/encryptable.rs:29-83), the sponge half of key_encrypt / key_decrypt (src/ecc/encryptable.rs:43-46,
// 82-93) and of kem_encrypt / kem_decrypt (src/kem/encryptable.rs:55-57, 96-103), composed from the KMAC launches of
// sponge_launch.hip or run in ONE pass by the fused kernels (sponge_fused.h: four lanes per item; sponge_wide_il.h: two waves
// per item; sponge_fused1.h: one lane per sponge), with those kernels' schedules.  NO CPU fallback for the data path.
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "common.h"
#include "sponge_launch.h"
#include "sponge_fused.h"
#include "sponge_host.h"
#include "sponge_internal.h"

namespace capy {

// 16 items per wave x one wave per SIMD with the plain round; beyond that the blocked round at raised priority pairs the
// waves of a SIMD (r03, profiles/r03_chipfull.txt: 32 768 x 5 MiB 353 -> 451 GiB/s, 49 152 x 4 MiB 405 -> 472, 98 304 x
// 1 MiB 432 -> 515 against the two-pass form; at 131 072 x 1 MiB the two passes win again, 541 vs 527).
// CAPY_DEBUG=fused_max=N overrides for A/B.

static size_t fused_one_wave_items() { return 16 * (size_t)device_simds(); }  // 16 384 on MI355X
static size_t fused_max_items()
{
    static const double v = debug_knob("fused_max", 0);
    return v > 0 ? (size_t)v : 96 * (size_t)device_simds();  // 98 304
}

// device-side tag compare for decrypt: status[i] = tags match ? OK : FAIL
__global__ void tag_compare_kernel_(const uint8_t *a, const uint8_t *b, uint32_t tag_len, uint64_t a_stride,
                                   uint64_t b_stride, int32_t *status, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t diff = 0;
    for (uint32_t j = 0; j < tag_len; j++) diff |= a[i * a_stride + j] ^ b[i * b_stride + j];
    status[i] = diff ? CAPY_ITEM_FAIL : CAPY_ITEM_OK;
}

// dst[i] = a[i] || b[i]  (z || pw of sha3_encrypt, encryptable.rs:33-34)
__global__ void concat_rows_kernel(uint8_t *dst, const uint8_t *a, uint32_t a_len, const uint8_t *b, uint32_t b_len,
                                   uint64_t n)
{
    const uint64_t row = a_len + b_len;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n * row; i += stride) {
        uint64_t r = i / row, c = i - r * row;
        dst[i] = c < a_len ? a[r * a_len + c] : b[r * b_len + (c - a_len)];
    }
}

// the same with one password length per item: row i = z_i (512 bytes) || pw_i, rows packed back to back;
// row_off[i] = 512 i + (pw_off[i] - pw_off[0]).  One wave per item.
__global__ __launch_bounds__(256) void concat_var_kernel(uint8_t *dst, uint64_t *row_off, const uint8_t *zs, const uint8_t *pws,
                                                         const uint64_t *pw_off, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 4 + threadIdx.x / 64;
    const uint32_t lane = threadIdx.x & 63;
    if (i > n) return;
    const uint64_t base = pw_off[0];
    const uint64_t o = pw_off[i] - base, row = 512 * i + o;
    if (lane == 0) row_off[i] = row;
    if (i == n) return;
    const uint64_t len = pw_off[i + 1] - pw_off[i];
    for (uint32_t c = lane; c < 512; c += 64) dst[row + c] = zs[512 * i + c];
    for (uint64_t c = lane; c < len; c += 64) dst[row + 512 + c] = pws[base + o + c];
}

void tag_compare_launch(const uint8_t *a, uint64_t a_stride, const uint8_t *b, uint64_t b_stride, uint32_t tag_len,
                        int32_t *status, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(tag_compare_kernel_, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, tag_len, a_stride,
                       b_stride, status, (uint64_t)n);
}

// first batch size that takes the one-lane-per-sponge fused kernel.  Beyond 32 items per SIMD there is a sponge for every lane of
// the chip (33 000 x 1 MiB: 513 GiB/s on the rotating schedule against 500 for the four-lane form in time slices,
// profiles/r05_fused_one_lane.txt).  r06: UNIFORM batches switch at 24 items per SIMD already -- up to 32 S items the one-lane
// form is one lone wave per SIMD (on 3/4 of the SIMDs at 24 S) whose launch takes the same 59.8 ms per MiB of message
// whatever n is, and same-box alternating runs (profiles/r06_fused_32s_ab.txt) put it ahead of the four-lane form at every
// size from there: 24 576 x 1 MiB 403 against 386 GiB/s, 28 672: 468 / 436, 32 768: 535 / 484 (x 5 MiB 533 / 488; D256
// 653 / 571), and the four-lane form's dip at 29 696-30 720 items (348-359 GiB/s: two waves on some SIMDs) is gone.  Ragged
// batches keep 32 S: half as many items per wave wait for a wave's longest message in the four-lane form.
static size_t fused1_min_items(bool uniform)
{
    static const long forced = (long)debug_knob("fused1_min", -1);
    if (forced >= 0) return (size_t)forced;
    return uniform ? 24 * (size_t)device_simds() : 32 * (size_t)device_simds() + 1;
}

// The plan of the rotating-occupancy schedule for the one-lane fused kernel: bundles of 128 items (four waves), C compute units,
// Cp of them doubled up per phase, P phases, every bundle doubled up in `a` of them (the arithmetic of rot_plan above);
// nb1 / nb2 = the speed of a lone wave (unrolled plain round) over that of a wave that shares its SIMD (rolled blocked round),
// swept over 33 000 .. 61 440 x 1 MiB (profiles/r05_fused_one_lane.txt): 1.45 with the lone role on the line stores, 1.6 with
// per-lane stores in the lone role (lone_direct: where lone waves dominate the schedule, fused1_launch).
// CAPY_DEBUG=fused1_ratio=R for A/B.
static bool fused1_rot_plan(uint64_t n, uint64_t nf, size_t simds, bool lone_direct, RotPlan &m)
{
    static const double forced_ratio = debug_knob("fused1_ratio", 0.0);
    const double ratio = (forced_ratio >= 1.0 && forced_ratio <= 2.5) ? forced_ratio : (lone_direct ? 1.6 : 1.45);
    if (n <= 32 * simds || n >= 64 * simds) return false;
    m.nf = nf;
    m.C = (uint32_t)(simds / 4);
    const uint64_t bundles = (n + 127) / 128;
    if (bundles <= m.C) return false;
    const uint32_t cp0 = (uint32_t)(bundles - m.C);
    double best = 2.0;
    m.P = 0;
    for (uint32_t a = 1; a <= 24; a++)
        for (uint32_t P = a + 1; P <= 24; P++) {
            if ((a * m.C) % (2 * P - a)) continue;
            const uint32_t cp = a * m.C / (2 * P - a);
            if (cp < cp0 || cp > m.C) continue;
            const double f = (double)a / P;
            if (f < best - 1e-12 || (f < best + 1e-12 && P < m.P)) {
                best = f;
                m.P = P;
                m.a = a;
                m.Cp = cp;
            }
        }
    if (!m.P) return false;
    m.G = m.C + m.Cp;
    m.nb2 = (uint32_t)((double)nf / ((double)m.a + ratio * (double)(m.P - m.a)));
    m.nb1 = (uint32_t)((nf - (uint64_t)m.a * m.nb2) / (m.P - m.a));
    return m.nb2 != 0 && m.nb1 != 0;
}

// The one-lane-per-sponge fused pass (sponge_fused1.h) over a batch that fills the chip; fp is complete but for the schedule.
static int fused1_launch(int rw, FusedParams &fp, const MsgView &m, hipStream_t s)
{
    const size_t simds = device_simds();
    const size_t groups = (fp.n + 31) / 32;  // waves
    static const bool direct = debug_knob("fused1_direct", 0) != 0;
    fp.direct_stores = direct ? 1 : 0;
    // line stores with sc1: written through and dropped from the XCD's L2.  Plain stores keep the written lines there, where
    // they crowd out the message lines that the next block step needs again (a 136-byte block shares a line with its successor):
    // HBM reads 1.26x the message bytes at four waves per SIMD and 1.09x at two against 1.015x / 1.003x with sc1, at the same
    // speed (profiles/r05_fused_one_lane.txt); CAPY_DEBUG=fused1_store=0 plain, 2 nt (1.19x / 1.02x)
    static const uint32_t store_policy = (uint32_t)debug_knob("fused1_store", 1);
    fp.store_policy = store_policy;
    // up to two waves per SIMD: the unrolled blocked round with the next block prefetched; beyond: the rolled round at three
    // or four waves per SIMD
    static const int forced_form = (int)debug_knob("fused1_form", 0);
    // at most one wave per SIMD (r06: the batches just below and at 32 items per SIMD): the lone-wave instance -- compiled so that
    // a second wave does not fit, plain unrolled round, per-lane stores (lone_direct below)
    fp.one_lane = groups <= simds ? 1 : (groups <= 2 * simds ? 2 : 4);
    if (forced_form == 1 || forced_form == 2 || forced_form == 4) fp.one_lane = (uint32_t)forced_form;
    fp.cap_waves = 0;
    if (fp.one_lane == 4) {
        const size_t w = (groups + simds - 1) / simds;
        static const int forced_cap = (int)debug_knob("fused1_waves", 0);
        fp.cap_waves = forced_cap ? (uint32_t)forced_cap : (w <= 3 ? (uint32_t)w : 0u);
    }
    const uint64_t nfull = m.offsets ? 0 : m.uniform_len / ((uint64_t)rw * 8);
    const bool long_uniform = !forced_form && !m.offsets && !m.order && nfull >= 512 && nfull < 0xfffffff0u;
    // Between one and two waves per SIMD: the rotating-occupancy schedule (sponge_fused1.h: sponge_fused1_rot_kernel).
    // CAPY_DEBUG=fused1_rot=0 switches it off.
    static const bool rot_on = debug_knob("fused1_rot", 1) != 0;
    RotPlan rp;
    // below 44 items per SIMD most of the blocks are absorbed by lone waves: those store per lane (sponge_fused1.h: lone_direct;
    // 36 864 / 40 960 x 1 MiB 511 / 533 -> 551 / 562 GiB/s at 1.08-1.16 x the bytes written; from 45 056 items on the two forms
    // are equal and the line stores keep the traffic at 1.00 x).  CAPY_DEBUG=fused1_lone_direct=0 / 1 forces it (A/B).
    static const int forced_lone = (int)debug_knob("fused1_lone_direct", -1);
    fp.lone_direct = forced_lone >= 0 ? (uint32_t)(forced_lone != 0) : (fp.n < 44 * simds ? 1u : 0u);
    if (rot_on && long_uniform && fused1_rot_plan(fp.n, nfull, simds, fp.lone_direct != 0, rp)) {
        const size_t done_bytes = (groups * 4 + 255) & ~(size_t)255, state_bytes = groups * 50 * 64 * 4;
        WsScrubGuard scrub(s);  // keyed sponge states: zeroed however this returns
        CAPY_WS(slws, uint8_t *, s, WS_STATE, done_bytes + state_bytes);
        scrub.add(WS_STATE, done_bytes + state_bytes);
        CAPY_HIP(hipMemsetAsync(slws, 0xff, done_bytes, s));  // SLICE_FRESH
        fp.sl_done = reinterpret_cast<uint32_t *>(slws);
        fp.sl_state = reinterpret_cast<uint32_t *>(slws + done_bytes);
        fp.rot_Cp = rp.Cp;
        fp.rot_G = rp.G;
        fp.rot_nb1 = rp.nb1;
        fp.rot_nb2 = rp.nb2;
        for (uint32_t ph = 0; ph < rp.P; ph++) {
            fp.rot_phase = ph;
            hipError_t e = launch_sponge_fused1_rot(rw, fp, rp.C, s);
            if (e == hipErrorInvalidValue) return fail(CAPY_ERR_ARG, "internal: no one-lane fused kernel instance for this rate");
            CAPY_HIP(e);
        }
        // what the phases left of the full blocks (fewer than P - a), tail and tag: one launch over all wave-groups
        fp.rot_G = 0;
        fp.one_lane = 2;
        fp.cap_waves = 0;
        fp.sl_groups = (uint32_t)groups;
        fp.sl_grid = (uint32_t)groups;
        fp.sl_launch = 0;
        fp.sl_blocks = 0xffffffffu;
        CAPY_HIP(launch_sponge_fused1(rw, fp, s));
        fp.sl_groups = 0;
        note_kernel(25, (int)rp.P + 1);
        return CAPY_OK;
    }
    // TIME SLICES for uniform batches of long messages whose wave count is not a whole number per SIMD (as for the four-lane
    // kernel and the uniform-framing digest kernel): one launch of the whole batch puts a further wave on some SIMDs and takes
    // that many waves' time however few they are.  Instead every launch holds exactly `level` waves per SIMD, launch k works on
    // the wave-groups (k level S + w) mod G for `bp` full blocks, the states cross launches through WS_STATE (12.8 KB per group
    // of 32 items), heads in a group's first turn, tail and tag in its last.  CAPY_DEBUG=fused1_slices=0 switches it off.
    static const bool slices_on = debug_knob("fused1_slices", 1) != 0;
    // level = waves per SIMD and launch.  Measured (1 MiB messages, profiles/r05_fused_one_lane.txt): slices of two waves per
    // SIMD on the unrolled instance run at a flat 645-650 GiB/s for every batch size, the rolled instance at three waves per SIMD
    // reaches 620-645 and at four 665-670 -- so beyond two waves per SIMD every batch takes level 2 unless it is within a fifth
    // of a quantum below a whole number q >= 4 of waves per SIMD, where one launch of the four-waves instance is at its best.
    // Level 1 (lone waves, 477 GiB/s) only when the rotating schedule above is switched off.
    uint32_t level = 0;
    if (groups > simds && groups * 100 <= simds * 149) level = 1;
    else if (groups > 2 * simds) {
        const size_t q = (groups + simds - 1) / simds;
        if (!(q >= 4 && groups * 5 > (5 * q - 1) * simds)) level = 2;
    }
    static const uint32_t forced_level = (uint32_t)debug_knob("fused1_level", 0);  // A/B: slices of this many waves per SIMD
    if (forced_level >= 1 && forced_level <= 3 && groups > forced_level * simds) level = forced_level;
    if (slices_on && level && long_uniform) {
        // turns per group: at most 64 (fewer leave a coarser last launch: 16 turns 586-604 GiB/s where 64 give 637-645; 128 and 256
        // only add state traffic; choosing the count whose last launch is fullest moved nothing beyond the noise), at least 8.
        // CAPY_DEBUG=fused1_turns=T forces it.
        static const uint64_t max_turns = (uint64_t)debug_knob("fused1_turns", 64);
        const uint32_t turns = (uint32_t)std::min<uint64_t>(max_turns ? max_turns : 64, nfull / 64);
        const uint32_t bp = (uint32_t)((nfull + turns - 1) / turns);
        const uint32_t need0 = (uint32_t)((nfull + bp - 1) / bp);
        const size_t done_bytes = (groups * 4 + 255) & ~(size_t)255, state_bytes = groups * 50 * 64 * 4;
        WsScrubGuard scrub(s);  // keyed sponge states: zeroed however this returns
        CAPY_WS(slws, uint8_t *, s, WS_STATE, done_bytes + state_bytes);
        scrub.add(WS_STATE, done_bytes + state_bytes);
        CAPY_HIP(hipMemsetAsync(slws, 0xff, done_bytes, s));  // SLICE_FRESH
        fp.one_lane = level == 1 ? 1 : 2;
        fp.cap_waves = 0;
        fp.sl_groups = (uint32_t)groups;
        fp.sl_grid = (uint32_t)(level * simds);
        fp.sl_blocks = bp;
        fp.sl_done = reinterpret_cast<uint32_t *>(slws);
        fp.sl_state = reinterpret_cast<uint32_t *>(slws + done_bytes);
        std::vector<uint32_t> need(groups, need0);
        size_t open_groups = groups;
        uint32_t k = 0;
        for (; open_groups; k++) {
            fp.sl_launch = k;
            CAPY_HIP(launch_sponge_fused1(rw, fp, s));
            for (size_t w = 0; w < fp.sl_grid; w++) {
                uint32_t &left = need[((size_t)k * fp.sl_grid + w) % groups];
                if (left && --left == 0) open_groups--;
            }
        }
        fp.sl_groups = 0;
        note_kernel(24, (int)k);
        return CAPY_OK;
    }
    fp.sl_groups = 0;
    hipError_t e = launch_sponge_fused1(rw, fp, s);
    if (e == hipErrorInvalidValue) return fail(CAPY_ERR_ARG, "internal: no one-lane fused kernel instance for this rate");
    CAPY_HIP(e);
    note_kernel(23, 1);
    return CAPY_OK;
}

// The symmetric half shared by sha3_encrypt/decrypt (src/sha3/encryptable.rs:39-42, 71-82), key_encrypt/decrypt
// (src/ecc/encryptable.rs:43-46, 82-93) and kem_encrypt/decrypt (src/kem/encryptable.rs:55-57, 96-103):
//     tag = kmac_xof(ka, m, 8*tag_len, ka_custom) ;  m ^= kmac_xof(ke, "", |m|, ke_custom)
// with ke at keka + i*keka_stride and ka right behind it (key_len bytes each).  Encrypt tags the plaintext first;
// decrypt XORs first, tags the candidate plaintext, writes status and restores the ciphertext of failed items.
// Small batches run both sponges of an item in lock-step in one pass (sponge_fused.h); that needs rate-aligned
// framing (not D224) and 8-byte aligned messages, otherwise the two-pass form is used.
int symmetric_crypt_dev(bool encrypt, int d, size_t n, const uint8_t *keka, size_t key_len, uint64_t keka_stride,
                        const MsgView &m, uint8_t *tags, size_t tag_len, const char *ke_custom, const char *ka_custom,
                        int32_t *status, hipStream_t s)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (n == 0) return CAPY_OK;
    auto keystream = [&](const int32_t *mask) {
        int kind = 0, launches = 0;
        last_kernel(&kind, &launches);  // the masked restore pass of decrypt is not what the hook reports
        const int rc = kmac_launch(d, n, fixed_keys(keka, key_len, keka_stride), m, false, (const uint8_t *)ke_custom,
                                   strlen(ke_custom), 1, nullptr, 0, 0, mask, s);
        if (mask) note_kernel(kind, launches);
        return rc;
    };
    auto tag = [&](uint8_t *out) {
        return kmac_launch(d, n, fixed_keys(keka + key_len, key_len, keka_stride), m, true, (const uint8_t *)ka_custom,
                           strlen(ka_custom), 0, out, tag_len, tag_len, nullptr, s);
    };
    uint8_t *tag2 = nullptr;
    if (!encrypt) {
        tag2 = reinterpret_cast<uint8_t *>(workspace(s, WS_TAG2, n * tag_len));
        if (!tag2) return fail(CAPY_ERR_HIP, "workspace allocation failed");
    }
    const Framing ff = cshake_framing(d);
    const bool fused_shape = fused_enabled() && ff.stride == (uint32_t)ff.rw * 8 && m.aligned8 && m.msgs != nullptr &&
                             tag_len <= 64 && (tag_len & 3) == 0;
    // From 24 (uniform) / 32 (ragged) items per SIMD: one lane per sponge (sponge_fused1.h), at every larger batch size; up to
    // there, four lanes per item (sponge_fused.h).  CAPY_DEBUG=fused1_min=N moves the boundary (A/B, tests).
    const bool one_lane = fused_shape && n >= fused1_min_items(!m.offsets && !m.order) && (key_len & 7) == 0 &&
                          (((uintptr_t)keka | keka_stride) & 7) == 0;
    const bool fused_ok = fused_shape && (one_lane || n <= fused_max_items());
    if (fused_ok) {
        FusedParams fp;
        memset(&fp, 0, sizeof fp);
        SpongeParams t;
        std::vector<uint8_t> unused;
        memset(&t, 0, sizeof t);
        cshake_prefix(d, (const uint8_t *)"KMAC", 4, (const uint8_t *)ka_custom, strlen(ka_custom), ff, t, unused);
        memcpy(fp.init_tag, t.init_state, sizeof fp.init_tag);
        cshake_prefix(d, (const uint8_t *)"KMAC", 4, (const uint8_t *)ke_custom, strlen(ke_custom), ff, t, unused);
        memcpy(fp.init_ks, t.init_state, sizeof fp.init_ks);
        kmac_head(d, key_len, t);
        fp.keka = keka;
        fp.keka_stride = keka_stride;
        fp.ka_offset = (uint32_t)key_len;
        fp.key_len = (uint32_t)key_len;
        fp.hdr_len = t.hdr_len;
        fp.hdr0 = t.hdr0;
        fp.hdr1 = t.hdr1;
        fp.head_len = t.head_len;
        fp.msgs = const_cast<uint8_t *>(m.msgs);
        fp.offsets = m.offsets;
        fp.lens = m.lens;
        fp.order = m.order;
        if (wants_device_order(m.offsets, m.order, n)) {
            const int orc = device_order(m.offsets, m.lens, n, s, &fp.order);
            if (orc) return orc;
        }
        fp.msg_stride = m.msg_stride;
        fp.uniform_len = m.uniform_len;
        fp.tag_stride = tag_len;
        fp.tag_len = (uint32_t)tag_len;
        fp.decrypt = encrypt ? 0 : 1;
        fp.staged = (sponge_debug_flags() & 64) ? 1 : 0;  // A/B switch (debug bit 6)
        fp.paired = (n > fused_one_wave_items() && !fp.staged) ? 1 : 0;
        fp.n = n;
        // Two waves per item (sponge_wide_il.h: a sponge spread over the lanes of a wave) while the batch leaves SIMDs idle
        // anyway: 2.6 us per block instead of 4.8, at any message length; break-even with the four-lane kernel near two waves
        // per SIMD (n = SIMDs).  Debug bits 4 / 5: never / for up to 4096 items (A/B and tests).
        {
            const unsigned dbg = sponge_debug_flags();
            fp.wide = ((dbg & 32) && n <= 4096) || (!(dbg & 16) && n <= wide_max_items()) ? 1 : 0;
        }
        fp.tags = encrypt ? tags : tag2;
        if (one_lane) {
            const int rc1 = fused1_launch(ff.rw, fp, m, s);
            if (rc1) return rc1;
            if (encrypt) return CAPY_OK;
            tag_compare_launch(tags, tag_len, tag2, tag_len, (uint32_t)tag_len, status, n, s);
            return keystream(status);
        }
        // Just above one wave per SIMD (16 384 < n <= 22 528 uniform long messages): TIME SLICES instead of a second wave on some SIMDs.
        // One launch of the whole batch puts a second wave on (n - 16 384) / 16 SIMDs, those run the paired round at 1 / 1.52 of
        // a lone wave's rate and the launch takes the two-waves time (0.30 s for 5 MiB messages) however few they are.  Here
        // every launch holds exactly one wave per SIMD: launch k works on the wave-groups (k C + w) mod G for `bp` full blocks,
        // the states cross launches through WS_STATE (6.4 KB per group of 16 items), so that all groups advance in turn at the
        // lone-wave rate and the batch takes G / C of the one-wave time: 16 400 / 18 432 / 20 480 items 0.30 -> 0.22 / 0.24 / 0.27 s
        // (profiles/r04_fused_slices.txt).  CAPY_DEBUG=fused_slices=0 switches it off.
        {
            const size_t simds = device_simds();
            const uint64_t nfull = m.offsets ? 0 : m.uniform_len / ((uint64_t)ff.rw * 8);
            const size_t groups = (n + 15) / 16;
            static const bool slices_on = debug_knob("fused_slices", 1) != 0;
            // level 1 = one wave per SIMD and launch, for 16 384 < n <= 22 528 (beyond, the single launch with two waves on some
            // SIMDs is as fast).  Beyond 32 768 items the one-lane-per-sponge kernel takes the batch: r04's levels 2 and 3 are gone.
            const uint32_t level = (groups > simds && groups * 16 <= simds * 22) ? 1 : 0;
            if (slices_on && !fp.staged && !fp.wide && !m.offsets && !m.order && level && nfull >= 512 && nfull < 0xfffffff0u) {
                const uint32_t turns = (uint32_t)std::min<uint64_t>(64, nfull / 64);  // >= 8 turns per group
                const uint32_t bp = (uint32_t)((nfull + turns - 1) / turns);
                const uint32_t need0 = (uint32_t)((nfull + bp - 1) / bp);
                const size_t done_bytes = (groups * 4 + 255) & ~(size_t)255, state_bytes = groups * 25 * 64 * 4;
                WsScrubGuard slscrub(s);  // keyed sponge states: zeroed however this block is left
                CAPY_WS(slws, uint8_t *, s, WS_STATE, done_bytes + state_bytes);
                slscrub.add(WS_STATE, done_bytes + state_bytes);
                CAPY_HIP(hipMemsetAsync(slws, 0xff, done_bytes, s));  // SLICE_FRESH
                fp.paired = 0;  // the instance compiled for exactly one wave per SIMD
                fp.sl_groups = (uint32_t)groups;
                fp.sl_grid = (uint32_t)(level * simds);
                fp.sl_blocks = bp;
                fp.sl_done = reinterpret_cast<uint32_t *>(slws);
                fp.sl_state = reinterpret_cast<uint32_t *>(slws + done_bytes);
                std::vector<uint32_t> need(groups, need0);
                size_t open_groups = groups;
                for (uint32_t k = 0; open_groups; k++) {
                    fp.sl_launch = k;
                    CAPY_HIP(launch_sponge_fused(ff.rw, fp, s));
                    for (size_t w = 0; w < fp.sl_grid; w++) {
                        uint32_t &left = need[((size_t)k * fp.sl_grid + w) % groups];
                        if (left && --left == 0) open_groups--;
                    }
                }
                note_kernel(22, (int)fp.sl_launch + 1);
                fp.sl_groups = 0;
            } else {
                CAPY_HIP(launch_sponge_fused(ff.rw, fp, s));
                note_kernel(fp.wide ? 27 : 20, 1);
            }
        }
        if (encrypt) return CAPY_OK;
        tag_compare_launch(tags, tag_len, tag2, tag_len, (uint32_t)tag_len, status, n, s);
        return keystream(status);
    }
    int rc;
    if (encrypt) {
        rc = tag(tags);
        if (rc == CAPY_OK) rc = keystream(nullptr);
        note_kernel(26, 2);
        return rc;
    }
    rc = keystream(nullptr);
    if (rc == CAPY_OK) rc = tag(tag2);
    if (rc) return rc;
    tag_compare_launch(tags, tag_len, tag2, tag_len, (uint32_t)tag_len, status, n, s);
    rc = keystream(status);
    note_kernel(26, 2);
    return rc;
}

// sha3_encrypt / sha3_decrypt on device buffers (src/sha3/encryptable.rs:29-83)
// (and the sponge half of KEMEncryptable, src/kem/encryptable.rs:47-59,84-104: same flow, tags "KEMKE"/"KEMKA")
// pw: n passwords, fixed length or per item (KeyView); pws_bytes = total password bytes (sizes the scratch of the
// per-item form without reading device memory)
int sha3_crypt_dev(bool encrypt, int d, size_t n, const KeyView &pw, uint64_t pws_bytes, const uint8_t *zs,
                   const MsgView &m, uint8_t *tags, int32_t *status, hipStream_t s, const char *ke_custom, const char *ka_custom)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (n == 0) return CAPY_OK;
    // z || pw per item (:33-34), then ke||ka = kmac_xof(z||pw, "", 1024, "S") (:36-37)
    WsScrubGuard scrub(s);  // z || pw and ke || ka are zeroed on the stream however this function returns
    CAPY_WS(keka, uint8_t *, s, WS_KEKA, n * 128);
    scrub.add(WS_KEKA, n * 128);
    scrub.add(WS_ZPW, n * 512 + (pw.key_offsets ? pws_bytes : n * pw.key_len));
    MsgView none;
    int rc;
    if (pw.key_offsets) {
        CAPY_WS(zpw, uint8_t *, s, WS_ZPW, n * 512 + pws_bytes);
        CAPY_WS(zoff, uint64_t *, s, WS_ZOFF, (n + 1) * 8);
        hipLaunchKernelGGL(concat_var_kernel, dim3((unsigned)((n + 1 + 3) / 4)), dim3(256), 0, s, zpw, zoff, zs, pw.keys,
                           pw.key_offsets, (uint64_t)n);
        CAPY_HIP(hipGetLastError());
        KeyView kv;
        kv.keys = zpw;
        kv.key_offsets = zoff;
        rc = kmac_launch(d, n, kv, none, true, (const uint8_t *)"S", 1, 0, keka, 128, 128, nullptr, s);
    } else {
        const size_t zk = 512 + pw.key_len;
        CAPY_WS(zpw, uint8_t *, s, WS_ZPW, n * zk);
        uint64_t tot = (uint64_t)n * zk;
        unsigned blocks = (unsigned)std::min<uint64_t>((tot + 255) / 256, 8192);
        hipLaunchKernelGGL(concat_rows_kernel, dim3(blocks), dim3(256), 0, s, zpw, zs, 512u, pw.keys, (uint32_t)pw.key_len,
                           (uint64_t)n);
        CAPY_HIP(hipGetLastError());
        rc = kmac_launch(d, n, fixed_keys(zpw, zk, zk), none, true, (const uint8_t *)"S", 1, 0, keka, 128, 128, nullptr, s);
    }
    if (rc) return rc;
    return symmetric_crypt_dev(encrypt, d, n, keka, 64, 128, m, tags, 64, ke_custom, ka_custom, status, s);
}

}  // namespace capy
