// sponge_launch.hip — the launch side of the batched sponge path: SP 800-185 framing exactly as the reference builds it
// (prefix blocks folded into the initial state on the host), kernel choice by batch shape, the phase schedules
// (sponge_mixed.h, sponge_rot.h) of the digest / XOF / keystream launches.  The encrypt / decrypt composition and the fused
// kernels' schedules: sponge_crypt.hip.  NO CPU fallback for the data path.
#include <string.h>
#include <algorithm>
#include <atomic>
#include <string>
#include <vector>
#include "common.h"
#include "sponge_launch.h"
#include "sponge_mixed.h"
#include "sponge_rot.h"
#include "sponge_host.h"
#include "sponge_internal.h"

namespace capy {

// ------------------------------------------------------------------ host keccak for the shared prefix block(s)
// (one or two permutations per API call: the batch-shared bytepad(encode_string(N)||encode_string(S), w))
static void host_keccakf(uint64_t a[25])
{
    static const uint64_t rc[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
        0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int rot[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int r = 0; r < 24; r++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int i = 0; i < 25; i++) {
            int x = i % 5, y = i / 5;
            uint64_t c1 = c[(x + 1) % 5];
            uint64_t e = a[i] ^ c[(x + 4) % 5] ^ ((c1 << 1) | (c1 >> 63));
            int s = rot[i];
            b[y + 5 * ((2 * x + 3 * y) % 5)] = s ? (e << s) | (e >> (64 - s)) : e;
        }
        for (int i = 0; i < 25; i++) {
            int x = i % 5, y5 = i - x;
            a[i] = b[i] ^ (~b[(x + 1) % 5 + y5] & b[(x + 2) % 5 + y5]);
        }
        a[0] ^= rc[r];
    }
}

// ------------------------------------------------------------------ SP 800-185 framing (reference forms)
static void left_encode(uint64_t v, std::vector<uint8_t> &out)
{ // src/sha3/aux_functions.rs:34-49
    if (v == 0) {
        out.push_back(1);
        out.push_back(0);
        return;
    }
    int nbytes = 0;
    for (uint64_t t = v; t; t >>= 8) nbytes++;
    out.push_back((uint8_t)nbytes);
    for (int i = nbytes - 1; i >= 0; i--) out.push_back((uint8_t)(v >> (8 * i)));
}

static void encode_string(const uint8_t *s, size_t len, std::vector<uint8_t> &out)
{ // src/sha3/aux_functions.rs:24-28
    left_encode((uint64_t)len * 8, out);
    out.insert(out.end(), s, s + len);
}

// bytepad as the reference writes it: always appends w - len%w zeros (a full w when aligned),
// src/sha3/aux_functions.rs:11-18
static std::vector<uint8_t> byte_pad(const std::vector<uint8_t> &x, uint32_t w)
{
    std::vector<uint8_t> z;
    left_encode(w, z);
    z.insert(z.end(), x.begin(), x.end());
    size_t padlen = w - (z.size() % w);
    z.insert(z.end(), padlen, 0);
    return z;
}


static Framing sha3_framing(int d)
{ // Capacity::from_bit_length(d) = 2d, src/sha3/constants.rs:38-45 ; Rate::from(&d), shake_functions.rs:31
    uint32_t r = (1600 - 2 * d) / 8;
    return {(int)(r / 8), r, (uint32_t)((1600 - d) / 64)};
}
Framing cshake_framing(int d)
{ // capacity = d, shake_functions.rs:63 ; bytes_to_state takes (r*8)/64 words per block, sponge.rs:52
    uint32_t r = (1600 - d) / 8;
    return {(int)(r / 8), r, (uint32_t)((1600 - d) / 64)};
}

// Fold the batch-shared cSHAKE prefix bytepad(encode_string(N) || encode_string(S), w) into p:
// either as init_state (whole blocks) or as raw prefix bytes `pre_host` (D224: r = 172, 168 consumed).
void cshake_prefix(int d, const uint8_t *fn, size_t fn_len, const uint8_t *cs, size_t cs_len,
                          const Framing &f, SpongeParams &p, std::vector<uint8_t> &pre_host)
{
    std::vector<uint8_t> enc;
    encode_string(fn, fn_len, enc);
    encode_string(cs, cs_len, enc);
    std::vector<uint8_t> pre = byte_pad(enc, (uint32_t)((1600 - d) / 8));
    memset(p.init_state, 0, sizeof p.init_state);
    const uint32_t rb = f.rw * 8;
    if (f.stride == rb) {
        for (size_t off = 0; off < pre.size(); off += rb) {
            for (int w = 0; w < f.rw; w++) {
                uint64_t v = 0;
                for (int j = 0; j < 8; j++) v |= (uint64_t)pre[off + 8 * w + j] << (8 * j);
                p.init_state[w] ^= v;
            }
            host_keccakf(p.init_state);
        }
        p.pre = nullptr;
        p.pre_len = 0;
    } else {
        pre_host = pre;
        p.pre_len = (uint32_t)pre.size();
    }
}

// per-item KMAC head = bytepad(encode_string(K), w) = left_encode(w) || left_encode(8|K|) || K || zeros
void kmac_head(int d, size_t key_len, SpongeParams &p)
{
    const uint32_t w = (1600 - d) / 8;
    std::vector<uint8_t> hdr;
    left_encode(w, hdr);
    left_encode((uint64_t)key_len * 8, hdr);
    p.hdr_len = (uint32_t)hdr.size();
    hdr.resize(16, 0);
    p.hdr0 = p.hdr1 = 0;
    for (int j = 0; j < 8; j++) {
        p.hdr0 |= (uint64_t)hdr[j] << (8 * j);
        p.hdr1 |= (uint64_t)hdr[8 + j] << (8 * j);
    }
    size_t z = p.hdr_len + key_len;
    p.head_len = (uint32_t)(z + (w - z % w));
    p.key_len = (uint32_t)key_len;
}

// Lanes per sponge: 1 fills the chip once there are >= ~64k independent sponges; below that the
// two-lane kernel is 1.48x faster per sponge (sponge_kernels_k2.h).  0 = choose by batch size, 3 = rotating schedule.
static std::atomic<int> g_lanes_per_sponge{0};
static std::atomic<unsigned> g_debug_flags{0};
unsigned sponge_debug_flags() { return g_debug_flags.load(); }
static std::atomic<bool> g_fused_enabled{true};
bool fused_enabled() { return g_fused_enabled.load(); }
// what the last symmetric_crypt_dev / launch_sponge call of this thread ran (capy_debug_last_sponge_kernel): kind =
//   digest launches  1 one-lane latency-tuned, 2 two-lane, 3 rotating one-/two-lane schedule, 4 one-lane issue-tuned, 5 wave-quantisation
//                    split, 7 uniform-framing kernel, 8 rotating-occupancy schedule, 9 uniform-framing kernel in time slices, 10 one wave
//                    per item (bit-interleaved Keccak lanes); 6 was the two-items-per-wave kernel of r02-r04
//   encrypt/decrypt  20 four lanes per item, 22 four lanes per item in time slices, 23 one lane per sponge, 24 one lane per sponge
//                    in time slices, 25 one lane per sponge on the rotating-occupancy schedule, 26 two passes, 27 two waves per
//                    item (bit-interleaved Keccak lanes); 21 was the one-wave-per-item kernel of r02-r04
// launches = kernel launches of the data pass (phases / slices)
static thread_local int t_last_kind = 0, t_last_launches = 0;
void last_kernel(int *kind, int *launches)
{
    *kind = t_last_kind;
    *launches = t_last_launches;
}
void note_kernel(int kind, int launches)
{
    t_last_kind = kind;
    t_last_launches = launches;
}


// Kernel choice by batch size relative to the device's SIMD count S (1024 on MI355X; measured crossovers, profiles/):
//   n <= 2 S         one wave per sponge, bit-interleaved Keccak lanes (sponge_wide_il.h)
//   n <= 32 S        two lanes per sponge, at most one wave per SIMD
//   32 S < n < 64 S  rotating one-lane / two-lane schedule when eligible (sponge_mixed.h), else one lane
//   n <= 128 S       one lane per sponge, latency-tuned instance; uniform batches above 64 S are launched as a
//                    head of 64 S + a remainder that follows the rules above (wave quantisation)
//   above            one lane per sponge, issue-tuned instance (> 2 waves per SIMD); ragged batches stay on the
//                    latency-tuned instance

static std::atomic<bool> g_mixed_enabled{true};

// largest batch that takes the wave-per-item encrypt kernel (sponge_wide_il.h, two waves per item): one item per SIMD; the
// digest kernel (one wave per item) takes twice as many -- two waves per SIMD either way.  profiles/r05_wide_interleaved.txt:
// 1024 x 5 MiB encrypt 0.156 s against 0.207 for the four-lane kernel (1280: 0.198, 1536: 0.237); 2048 x 5 MiB SHA3-256 0.149
// against 0.197 for the two-lane kernel (3072: 0.228).  CAPY_DEBUG=wide_max=N overrides
// SIMDs of the current device (4 per CU)
unsigned device_simds()
{
    static std::atomic<unsigned> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1024;
    unsigned v = cached[dev].load();
    if (!v) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        v = 4u * (unsigned)cus;
        cached[dev].store(v);
    }
    return v;
}

size_t wide_max_items()
{
    static const long forced = (long)debug_knob("wide_max", -1);
    return forced >= 0 ? (size_t)forced : device_simds();
}

// speed of the two-lane form relative to the one-lane form, per sponge, when both share the chip at one wave per SIMD.
// Re-measured with the per-lane-load kernels for P = 2..6 phases (profiles/r02_mixed_ratio_sweep.txt): best at 1.46-1.47
// for every P (+1.8 % over the 1.50 of round 1 at the headline batch); CAPY_DEBUG=mixed_ratio=R overrides
static double mixed_ratio()
{
    static const double r = [] {
        const double v = debug_knob("mixed_ratio", 0.0);
        return (v >= 1.0 && v <= 2.0) ? v : 1.47;
    }();
    return r;
}

// The rotating one-lane / two-lane schedule of sponge_mixed.h for batches between half a chip and a full chip of
// one-lane sponges: P groups of gs sponges, P phases; each sponge gets one two-lane phase of nb2 blocks and P-1
// one-lane phases of nb1 blocks.
struct MixedPlan {
    uint64_t P, gs, nf;
    uint32_t nb1, nb2;
};
static bool mixed_plan(int rw, const SpongeParams &p, bool forced, MixedPlan &m)
{
    const uint32_t rb = (uint32_t)rw * 8;
    if (p.out_mode != 0 || !p.absorb_body || p.offsets || p.mask || p.pre_len || p.head_len % rb || p.stride_bytes != rb) return false;
    if (p.key_offsets) return false;  // per-item key lengths: the head block count differs per sponge
    if ((((uintptr_t)p.msgs | p.msg_stride) & 7) || p.msg_stride * 64 >= 0xfff00000ULL || p.msg_stride < p.uniform_len) return false;
    const uint64_t S = device_simds(), n = p.n;
    m.nf = p.uniform_len / rb;
    if (n <= 32 * S || n >= 64 * S || m.nf < 256) return false;  // below ~35 KB per message the phase launches eat the gain
    const uint64_t spare = 64 * S - n;        // sponges' worth of idle lanes under the one-lane kernel
    uint64_t P = (n + spare - 1) / spare;     // phases = groups
    if (P < 2) P = 2;
    if (P > 6 && !forced) return false;       // gain (ratio + P - 1) / P would be below 8 %
    if (P > 16) return false;
    m.gs = ((n + P - 1) / P + 63) / 64 * 64;  // group size: whole one-lane waves
    m.P = (n + m.gs - 1) / m.gs;
    if (m.P < 2) return false;
    m.nb1 = (uint32_t)((double)m.nf / (mixed_ratio() + (double)(m.P - 1)));
    m.nb2 = (uint32_t)(m.nf - (m.P - 1) * (uint64_t)m.nb1);
    return m.nb1 != 0;
}

// Returns 1 if it handled the launch, 0 if the launch is not eligible, < 0 on error.
static int try_launch_mixed(int rw, const SpongeParams &p, bool forced, hipStream_t s)
{
    MixedPlan m;
    if (!mixed_plan(rw, p, forced, m)) return 0;
    const uint64_t n = p.n, n_pad = (n + 63) / 64 * 64;
    WsScrubGuard scrub(s);  // keyed sponge states when the launch has per-item head blocks: zeroed on every return
    CAPY_WS(state, uint64_t *, s, WS_STATE, 25 * n_pad * sizeof(uint64_t));
    if (p.head_len) scrub.add(WS_STATE, 25 * n_pad * sizeof(uint64_t));
    MixedParams q;
    memset(&q, 0, sizeof q);
    q.msgs = p.msgs;
    q.msg_stride = p.msg_stride;
    q.n = n;
    q.state = state;
    q.n_pad = n_pad;
    memcpy(q.init_state, p.init_state, sizeof q.init_state);
    q.k1_count = m.nb1;
    q.k2_count = m.nb2;
    q.staged = (g_debug_flags.load() & 64) ? 1 : 0;  // A/B switch (debug bit 6): LDS-staged loads
    const uint32_t hb = p.head_len / ((uint32_t)rw * 8);
    if (hb) {
        // per-item head blocks (KMAC keys) first: a head-only launch of the one-lane kernel seeds the state buffer
        SpongeParams h = p;
        h.debug_flags = g_debug_flags.load();
        h.head_state = state;
        h.resume_pad = n_pad;
        CAPY_HIP(launch_sponge_k1_lat(rw, 0, h, s));
    }
    for (uint64_t ph = 0; ph < m.P; ph++) {
        q.load_state = (ph || hb) ? 1 : 0;
        q.k2_begin = ph * m.gs;
        q.k2_end = std::min(n, (ph + 1) * m.gs);
        q.k2_waves = (uint32_t)((q.k2_end - q.k2_begin + 31) / 32);
        q.k2_first = (uint32_t)(ph * m.nb1);
        q.k1_first_lo = (uint32_t)((ph ? ph - 1 : 0) * m.nb1 + m.nb2);  // groups below ph have had their fast phase
        q.k1_first_hi = (uint32_t)(ph * m.nb1);
        const uint64_t k1_items = n - (q.k2_end - q.k2_begin);
        const unsigned waves = q.k2_waves + (unsigned)((k1_items + 63) / 64);
        hipError_t e = launch_sponge_mixed(rw, q, waves, s);
        if (e == hipErrorInvalidValue) return fail(CAPY_ERR_ARG, "internal: no mixed kernel instance for this rate");
        CAPY_HIP(e);
    }
    // tail blocks, padding and squeeze from the saved states
    SpongeParams r = p;
    r.debug_flags = g_debug_flags.load();
    r.resume_state = state;
    r.resume_pad = n_pad;
    r.resume_blocks = hb + (uint32_t)m.nf;
    CAPY_HIP(launch_sponge_k1_lat(rw, 0, r, s));
    note_kernel(3, (int)m.P + 1 + (hb ? 1 : 0));
    return 1;
}

// The rotating-OCCUPANCY schedule of sponge_rot.h for uniform digest batches between one and two one-lane waves per SIMD:
// groups of 256 sponges, C compute units, Cp of them doubled up per phase, P phases, every group doubled up in `a` of them.
// A rotation that returns every group to the same count needs 2 Cp P = a G with G = C + Cp groups, i.e.
// Cp = a C / (2 P - a): the plan takes the (P, a) with the smallest a / P (= the shortest time) whose Cp covers the batch.
static double rot_ratio()
{
    // blocks per phase of a group on its own / of a doubled-up group: the speed of a lone wave (unrolled plain round)
    // over that of a wave that shares its SIMD (unrolled blocked round with priority); CAPY_DEBUG=rot_ratio=R for A/B
    static const double r = [] {
        const double v = debug_knob("rot_ratio", 0.0);
        return (v >= 1.0 && v <= 2.5) ? v : 1.5;  // measured 1.4 / 1.5 / 1.6 / 1.75 over 66 048 .. 122 880 x 1 MiB: profiles/r04_chipfull.txt
    }();
    return r;
}
static bool rot_plan(int rw, const SpongeParams &p, RotPlan &m)
{
    const uint32_t rb = (uint32_t)rw * 8;
    if (debug_knob("rot", 1) == 0) return false;
    if (p.out_mode != 0 || !p.absorb_body || p.offsets || p.mask || p.order || p.pre_len || p.head_len % rb || p.stride_bytes != rb) return false;
    if (p.key_offsets || p.resume_state || p.head_state) return false;
    if ((((uintptr_t)p.msgs | p.msg_stride) & 7) || p.msg_stride < p.uniform_len) return false;
    const uint64_t S = device_simds(), n = p.n;
    if (n <= 64 * S || n >= 128 * S) return false;
    m.nf = p.uniform_len / rb;
    if (m.nf < 512) return false;  // short messages: P launches and 2 P state transfers per sponge eat the gain
    m.C = (uint32_t)(S / 4);
    const uint64_t groups = (n + 255) / 256;
    if (groups <= m.C) return false;
    const uint32_t cp0 = (uint32_t)(groups - m.C);
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
    m.nb2 = (uint32_t)((double)m.nf / ((double)m.a + rot_ratio() * (double)(m.P - m.a)));
    m.nb1 = (uint32_t)((m.nf - (uint64_t)m.a * m.nb2) / (m.P - m.a));  // what is left (< P - a blocks) goes to the resume launch
    return m.nb2 != 0 && m.nb1 != 0;
}

// Returns 1 if it handled the launch, 0 if the launch is not eligible, < 0 on error.
static int try_launch_rot(int rw, const SpongeParams &p, hipStream_t s)
{
    RotPlan m;
    if (!rot_plan(rw, p, m)) return 0;
    const uint64_t n = p.n, n_pad = (n + 63) / 64 * 64;
    WsScrubGuard scrub(s);  // with per-item head blocks these are KEYED sponge states (e.g. by the Schnorr secret): zeroed on every return
    CAPY_WS(state, uint64_t *, s, WS_STATE, 25 * n_pad * sizeof(uint64_t));
    if (p.head_len) scrub.add(WS_STATE, 25 * n_pad * sizeof(uint64_t));
    RotParams q;
    memset(&q, 0, sizeof q);
    q.msgs = p.msgs;
    q.msg_stride = p.msg_stride;
    q.n = n;
    q.state = state;
    q.n_pad = n_pad;
    memcpy(q.init_state, p.init_state, sizeof q.init_state);
    q.Cp = m.Cp;
    q.G = m.G;
    q.nb1 = m.nb1;
    q.nb2 = m.nb2;
    const uint32_t hb = p.head_len / ((uint32_t)rw * 8);
    if (hb) {
        // per-item head blocks (KMAC keys) first: a head-only launch of the one-lane kernel seeds the state buffer
        SpongeParams h = p;
        h.debug_flags = g_debug_flags.load();
        h.head_state = state;
        h.resume_pad = n_pad;
        CAPY_HIP(launch_sponge_k1_lat_paired(rw, 0, h, s));
        q.msgs = p.msgs;  // the body starts at the message's first byte; the head lives in the key buffer
    }
    for (uint32_t ph = 0; ph < m.P; ph++) {
        q.load_state = (ph || hb) ? 1 : 0;
        q.phase = ph;
        hipError_t e = launch_sponge_rot(rw, q, m.C, 0, s);
        if (e == hipErrorInvalidValue) return fail(CAPY_ERR_ARG, "internal: no rotating-occupancy kernel instance for this rate");
        CAPY_HIP(e);
    }
    // what the phases left of the body, tail blocks, padding and squeeze from the saved states
    SpongeParams r = p;
    r.debug_flags = g_debug_flags.load() | SPONGE_DIRECT_LOADS;
    r.resume_state = state;
    r.resume_pad = n_pad;
    r.resume_blocks = hb + m.a * m.nb2 + (m.P - m.a) * m.nb1;
    CAPY_HIP(launch_sponge_k1_lat_paired(rw, 0, r, s));
    note_kernel(8, (int)m.P + 1 + (hb ? 1 : 0));
    return 1;
}

// which kernel launch_sponge() picks: 1 one-lane latency-tuned, 2 two-lane, 3 rotating schedule, 4 one-lane issue-tuned,
// 5 wave-quantisation split, 6 wave-per-item digest, 7 uniform-framing kernel, 8 rotating-occupancy schedule
// occupancy cap of the uniform-framing kernel in waves per SIMD (0: none, four fit; CAPY_DEBUG=uniform_waves=1..4 for A/B)
static int uniform_waves()
{
    static const int w = [] {
        const int v = (int)debug_knob("uniform_waves", 0);
        return (v >= 1 && v <= 4) ? v : 0;
    }();
    return w;
}

// The conditions under which launch_sponge() takes the uniform-framing kernel (sponge_uniform.h) / the wave-per-item
// digest kernel (sponge_wide_il.h) -- shared with sponge_plan(), so that capy_sha3_launch_plan reports the kernel that
// really runs.  p.order must already hold the device-side processing order if one is used.  Debug bit 7: never.
static bool uniform_kernel_ok(int rw, const SpongeParams &p, int forced, unsigned dbg, size_t simds)
{
    const uint32_t rb = (uint32_t)rw * 8;
    if (forced != 0 || (dbg & 128) || p.n <= 128 * simds) return false;
    if (p.out_mode != 0 || p.pre_len || p.key_offsets || p.offsets || p.mask || p.order || p.resume_state || p.head_state ||
        p.stride_bytes != rb || p.head_len % rb)
        return false;
    if (p.head_len && ((((uintptr_t)p.keys | p.key_stride) & 7) || (p.key_len & 7) || p.hdr_len > 16)) return false;
    if (p.absorb_body && p.uniform_len && ((((uintptr_t)p.msgs | p.msg_stride) & 7) || p.msg_stride < p.uniform_len)) return false;
    if (p.out_len <= 8 * p.sq_words) return (((uintptr_t)p.out | p.out_stride) & 7) == 0;  // one squeeze block, per lane
    // longer outputs leave as whole 128-byte lines: 16-byte chunks of 16-word rows
    return rw >= 16 && p.sq_words == (uint32_t)rw && (p.out_len & 15) == 0 && (((uintptr_t)p.out | p.out_stride) & 15) == 0 &&
           p.out_stride >= p.out_len && p.out_stride * 64 < 0xfff00000ULL;
}
static bool wide_digest_ok(int rw, const SpongeParams &p, int forced, unsigned dbg)
{
    const bool shape_ok = p.out_mode == 0 && p.pre_len == 0 && p.stride_bytes == (uint32_t)rw * 8 && !p.resume_state && !p.head_state;
    // any message length: measured r03 with that round's wave-per-item kernel (profiles/r03_small_calls.txt), KMACXOF256 of
    // 64 B / 1 KiB / 16 KiB messages at n <= 2048: 0.029 -> 0.015, 0.066 -> 0.038, 0.645 -> 0.394 ms against the two-lane kernel
    return shape_ok && (((dbg & 32) && p.n <= 4096) ||
                        (forced == 0 && !(dbg & 16) && p.n <= 2 * wide_max_items()));
}

// Uniform digest batches JUST ABOVE a whole number k of waves per SIMD (k = 2, 3, 4): one launch puts a further wave on a few
// SIMDs and takes a whole further chain -- SHA3-256 over 256 KiB messages: 1339 GB/s at 131 072 items, 1035 at 133 120; 1399 at
// 196 608, 1160 at 200 704; 1476 at 262 144, 1223 at 266 240 (profiles/r04_uniform_slices.txt).  Instead: a sequence of launches of
// the uniform-framing kernel's SLICED instance that each hold exactly k waves per SIMD; launch j works on the groups of 64
// items (j k S + w) mod G for nfull / 64 full blocks (at least 8 turns per group), states through WS_STATE, heads in a group's
// first turn, trailer and squeeze in its last.  Returns 1 if it handled the launch, 0 if not eligible, < 0 on error.
// CAPY_DEBUG=uniform_slices=0 switches it off.
static int try_launch_uniform_sliced(int rw, const SpongeParams &p, int forced, unsigned dbg, size_t simds, hipStream_t s)
{
    static const bool on = debug_knob("uniform_slices", 1) != 0;
    if (!on || !p.absorb_body || !uniform_kernel_ok(rw, p, forced, dbg, simds)) return 0;
    const uint64_t nfull = p.uniform_len / ((uint64_t)rw * 8);
    if (nfull < 512 || nfull >= 0xfffffff0u) return 0;
    const size_t groups = (p.n + 63) / 64;
    // Level k = waves per SIMD and launch.  Measured (256 KiB messages): the sliced launches run at a flat 1362 / 1458 / 1474 GB/s
    // at k = 2 / 3 / 4 -- as fast as the single launch at the NEXT whole number of waves per SIMD (1373 / 1445 / 1451), so a level
    // serves almost up to the next quantum; beyond four waves per SIMD (the occupancy of the kernel) level 4 serves every batch
    // that is not within a fifth of a quantum below a whole number of them (where the single launch is at its best: 1536-1560).
    uint32_t level = 0;
    if (groups > 2 * simds && groups * 100 <= 2 * simds * 148) level = 2;
    else if (groups >= 3 * simds && groups * 100 <= 3 * simds * 133) level = 3;  // (the whole numbers themselves included:
    else if (groups >= 4 * simds) {                                               //  1369 -> 1458, 1440 -> 1490 GB/s)
        const size_t q = (groups + simds - 1) / simds;
        if (groups * 5 <= (5 * q - 1) * simds) level = 4;
    }
    if (!level) return 0;
    const uint32_t turns = (uint32_t)std::min<uint64_t>(64, nfull / 64);
    const uint32_t bp = (uint32_t)((nfull + turns - 1) / turns);
    const uint32_t need0 = (uint32_t)((nfull + bp - 1) / bp);
    const size_t done_bytes = (groups * 4 + 255) & ~(size_t)255, state_bytes = groups * 50 * 64 * 4;
    WsScrubGuard scrub(s);  // keyed sponge states (KMAC heads): zeroed however this returns
    CAPY_WS(ws, uint8_t *, s, WS_STATE, done_bytes + state_bytes);
    if (p.head_len) scrub.add(WS_STATE, done_bytes + state_bytes);
    CAPY_HIP(hipMemsetAsync(ws, 0xff, done_bytes, s));
    SpongeParams q = p;
    q.sl_groups = (uint32_t)groups;
    q.sl_blocks = bp;
    q.sl_done = reinterpret_cast<uint32_t *>(ws);
    q.sl_state = reinterpret_cast<uint32_t *>(ws + done_bytes);
    const unsigned grid = (unsigned)(level * simds);
    std::vector<uint32_t> need(groups, need0);
    size_t open_groups = groups;
    for (uint32_t j = 0; open_groups; j++) {
        q.sl_launch = j;
        hipError_t e = launch_sponge_uniform(rw, q, level <= 3 ? (int)level : 0, s, grid);
        if (e == hipErrorInvalidValue) return fail(CAPY_ERR_ARG, "internal: no kernel instance for this rate");
        CAPY_HIP(e);
        for (size_t w = 0; w < grid; w++) {
            uint32_t &left = need[((size_t)j * grid + w) % groups];
            if (left && --left == 0) open_groups--;
        }
    }
    note_kernel(9, (int)q.sl_launch + 1);
    return 1;
}

static int sponge_plan(int rw, const SpongeParams &p, int *phases)
{
    const int forced = g_lanes_per_sponge.load();
    const size_t simds = device_simds();
    *phases = 1;
    MixedPlan m;
    const unsigned dbg = g_debug_flags.load();
    if (wide_digest_ok(rw, p, forced, dbg)) return 10;
    if (uniform_kernel_ok(rw, p, forced, dbg, simds)) return 7;
    if ((forced == 3 || (forced == 0 && g_mixed_enabled.load())) && mixed_plan(rw, p, forced == 3, m)) {
        *phases = (int)m.P;
        return 3;
    }
    RotPlan rp;
    if (forced == 0 && g_mixed_enabled.load() && rot_plan(rw, p, rp)) {
        *phases = (int)rp.P;
        return 8;
    }
    // the wave-quantisation split of launch_sponge(): a full-chip head of 64 S one-lane sponges + a remainder that
    // takes the two-lane kernel or the rotating schedule
    if (forced == 0 && (dbg & 256) && !p.offsets && !p.mask && !p.order && p.n > 64 * simds && p.n < 128 * simds) {
        SpongeParams tail = p;
        tail.n = p.n - 64 * simds;
        if (tail.n <= 32 * simds) {
            *phases = 2;
            return 5;
        }
        if (g_mixed_enabled.load() && mixed_plan(rw, tail, false, m)) {
            *phases = 1 + (int)m.P;
            return 5;
        }
    }
    if (forced == 2 || ((forced == 0 || forced == 3) && p.n <= 32 * simds)) return 2;
    return p.n > 128 * simds ? 4 : 1;
}

bool wants_device_order(const uint64_t *offsets, const uint32_t *order, uint64_t n)
{
    return offsets && !order && n >= 128 && n <= 0xffffffffULL && !(g_debug_flags.load() & 4);
}

static int launch_sponge(int rw, const SpongeParams &p, hipStream_t s)
{
    if (p.n == 0) return CAPY_OK;
    const int forced = g_lanes_per_sponge.load();
    SpongeParams q = p;
    q.debug_flags = g_debug_flags.load();
    if (wants_device_order(p.offsets, p.order, p.n)) {  // ragged device batch: longest first
        const int rc = device_order(p.offsets, p.lens, p.n, s, &q.order);
        if (rc) return rc;
    }
    hipError_t e;
    const size_t simds = device_simds();
    // per-lane message loads in the one-lane kernels (sponge_kernels.h phase B); debug bit 6: A/B switch to the
    // wave-cooperative loads through LDS of round 1
    const bool direct_ok = !(q.debug_flags & 64);
    if (direct_ok) q.debug_flags |= SPONGE_DIRECT_LOADS;
    const SpongeParams &p2 = q;
    if (forced == 3 || (forced == 0 && g_mixed_enabled.load())) {
        const int m = try_launch_mixed(rw, p, forced == 3, s);
        if (m < 0) return m;
        if (m > 0) return CAPY_OK;
    }
    // between one and two one-lane waves per SIMD: the rotating-occupancy schedule (sponge_rot.h; r04)
    if (forced == 0 && g_mixed_enabled.load()) {
        const int m = try_launch_rot(rw, p, s);
        if (m < 0) return m;
        if (m > 0) return CAPY_OK;
    }
    // Wave quantisation between one and two one-lane waves per SIMD: with the PLAIN round a uniform batch of 64 S + rem
    // sponges runs at the two-waves-per-SIMD time (1.96x) although most SIMDs hold one wave, so r01/r02 launched the first
    // 64 S on their own (1.0x) and the remainder with whatever suits its size (two-lane 0.68x, rotating schedule
    // 0.8-0.92x).  Since r03 the paired latency-tuned instance (two waves of a SIMD cost 1.32x, not 1.96x) takes the whole
    // batch in one launch: 73 728 / 81 920 / 98 304 / 114 688 x 1 MiB 92.4 / 92.8 / 99.6 / 103.2 ms against 99.5 / 99.1 /
    // 99.2 / 109.6 ms for the split (profiles/r03_chipfull.txt).  The split stays behind debug bit 8 (no paired instance).
    if (forced == 0 && (q.debug_flags & 256) && !p.offsets && !p.mask && !p.order && p.n > 64 * simds && p.n < 128 * simds) {
        auto subrange = [&](uint64_t first, uint64_t count) {
            SpongeParams r = p;
            r.msgs = p.msgs ? p.msgs + first * p.msg_stride : nullptr;
            r.keys = (p.keys && !p.key_offsets) ? p.keys + first * p.key_stride : p.keys;
            r.key_offsets = p.key_offsets ? p.key_offsets + first : nullptr;
            r.out = p.out ? p.out + first * p.out_stride : nullptr;
            r.n = count;
            return r;
        };
        const uint64_t head_n = 64 * simds;
        const SpongeParams tail = subrange(head_n, p.n - head_n);
        MixedPlan mp;
        if (tail.n <= 32 * simds || (g_mixed_enabled.load() && mixed_plan(rw, tail, false, mp))) {
            SpongeParams head = subrange(0, head_n);
            head.debug_flags = q.debug_flags;
            e = launch_sponge_k1_lat(rw, (int)p.out_mode, head, s);
            if (e == hipErrorInvalidValue) return fail(CAPY_ERR_ARG, "internal: no kernel instance for this rate / mode");
            CAPY_HIP(e);
            return launch_sponge(rw, tail, s);
        }
    }
    // Just above k waves per SIMD (k = 2, 3, 4): time slices of exactly k waves per SIMD instead of a further wave on a few
    {
        const int m = try_launch_uniform_sliced(rw, p2, forced, q.debug_flags, simds, s);
        if (m < 0) return m;
        if (m > 0) return CAPY_OK;
    }
    // Chip-full launches with wave-uniform framing (equal key, message and output lengths, 8-byte aligned): every framing
    // decision is scalar code in sponge_uniform.h.  Debug bit 7: never (A/B and tests).
    if (uniform_kernel_ok(rw, p2, forced, q.debug_flags, simds)) {
        e = launch_sponge_uniform(rw, p2, uniform_waves(), s);
        if (e == hipErrorInvalidValue) return fail(CAPY_ERR_ARG, "internal: no kernel instance for this rate");
        CAPY_HIP(e);
        note_kernel(7, 1);
        return CAPY_OK;
    }
    // Very small digest batches: one wave per item (sponge_wide_il.h), 2.5 us per permutation against 4.8 for the two-lane
    // kernel; up to two waves per SIMD (SHA3-256 of 5 MiB messages: 0.097 s for up to 256 items, 0.111 at 1024, 0.149 at 2048;
    // two-lane kernel 0.197).  Debug bit 4 / 5: never / for up to 4096 items.
    int kind = 1;
    if (wide_digest_ok(rw, p2, forced, q.debug_flags))
        kind = 10, e = launch_sponge_il_digest(rw, p2, s);
    else if (forced == 2 || ((forced == 0 || forced == 3) && p.n <= 32 * simds))
        kind = 2, e = launch_sponge_k2(rw, (int)p.out_mode, p2, s);
    // ragged batches stay on the latency-tuned instance at every size: its ragged path keeps the source pointers in
    // registers and prefetches a block ahead, which the 128-VGPR issue-tuned instance cannot afford (2^18 ragged
    // messages of 0..64 KiB: 16.0 vs 13.6 ms; equal lengths given through offsets: 9.7 vs 8.1 ms)
    else if (p.n > 128 * simds && !(q.debug_flags & 2) && !p2.offsets && !p2.order)  // debug bit 1: A/B switch
        kind = 4, e = launch_sponge_k1_full(rw, (int)p.out_mode, p2, s);
    // more than one wave on some SIMD: the paired form of the latency-tuned instance (debug bit 8: A/B switch)
    else if (p.n > 64 * simds && !(q.debug_flags & 256))
        e = launch_sponge_k1_lat_paired(rw, (int)p.out_mode, p2, s);
    else
        e = launch_sponge_k1_lat(rw, (int)p.out_mode, p2, s);
    if (e == hipErrorInvalidValue) return fail(CAPY_ERR_ARG, "internal: no kernel instance for this rate / mode");
    CAPY_HIP(e);
    note_kernel(kind, 1);
    return CAPY_OK;
}

static void body_args(SpongeParams &p, const MsgView &m)
{
    p.msgs = m.msgs;
    p.offsets = m.offsets;
    p.lens = m.lens;
    p.uniform_len = m.uniform_len;
    p.msg_stride = m.msg_stride;
    p.order = m.order;
}

// The (D224-only) raw prefix bytes reach the device as the ARGUMENT of a tiny kernel that writes them into the
// stream's prefix slot: stream-ordered behind every earlier reader of the slot, no host copy and no synchronisation,
// so the *_dev entry points stay asynchronous at D224 as well.  bytepad(encode_string("KMAC") || encode_string(S), 172)
// is two blocks (344 bytes) for every customisation string up to 162 bytes; a longer prefix takes the synchronous copy.
struct PreBytes {
    uint64_t w[44];
};
__global__ void pre_write_kernel(const PreBytes b, uint64_t *dst, uint32_t nwords)
{
    const uint32_t i = threadIdx.x;
    if (i < nwords) dst[i] = b.w[i];
}

static int launch_with_pre(int rw, SpongeParams &p, const std::vector<uint8_t> &pre_host, hipStream_t s)
{
    if (!pre_host.empty()) {
        PreBytes pb;
        const size_t cap = std::max(pre_host.size(), sizeof pb.w);
        if (pre_host.size() <= sizeof pb.w) {
            // the slot is sized for the largest inline prefix up front: steady-state calls never reallocate it
            CAPY_WS(pre_dev, uint8_t *, s, WS_PRE, cap);
            memset(pb.w, 0, sizeof pb.w);
            memcpy(pb.w, pre_host.data(), pre_host.size());
            hipLaunchKernelGGL(pre_write_kernel, dim3(1), dim3(64), 0, s, pb, reinterpret_cast<uint64_t *>(pre_dev),
                               (uint32_t)((pre_host.size() + 7) / 8));
            CAPY_HIP(hipGetLastError());
            p.pre = pre_dev;
        } else {
            CAPY_HIP(hipStreamSynchronize(s));  // an earlier launch on this stream may still read the slot
            CAPY_WS(pre_dev, uint8_t *, s, WS_PRE, cap);
            CAPY_HIP(hipMemcpy(pre_dev, pre_host.data(), pre_host.size(), hipMemcpyHostToDevice));
            p.pre = pre_dev;
        }
    }
    return launch_sponge(rw, p, s);
}

// A KMACXOF launch in all its forms (kmac_xof, shake_functions.rs:79-89): digest-style output
// (out_mode 0) or in-place keystream XOR over the message buffer (out_mode 1, X = ""), optional mask.
int kmac_launch(int d, size_t n, const KeyView &kv, const MsgView &m,
                       bool absorb_body, const uint8_t *custom, size_t custom_len, int out_mode, uint8_t *outs,
                       uint64_t out_stride, size_t out_len, const int32_t *mask, hipStream_t s)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (kv.key_len > CAPY_MAX_KEY_LEN) return fail(CAPY_ERR_ARG, "key too long");
    // the kernels count output bytes in 32 bits (the reference takes l: usize, shake_functions.rs:79): refuse, never truncate
    if (out_len > CAPY_MAX_OUT_LEN) return fail(CAPY_ERR_ARG, "output longer than 2^32 - 1 bytes per item (l_bits >= 2^35)");
    Framing f = cshake_framing(d);
    SpongeParams p;
    memset(&p, 0, sizeof p);
    std::vector<uint8_t> pre_host;
    cshake_prefix(d, (const uint8_t *)"KMAC", 4, custom, custom_len, f, p, pre_host);
    if (kv.key_offsets) {  // per-item key lengths: the kernels build each item's head (item_head, sponge_params.h)
        p.key_offsets = kv.key_offsets;
        p.bytepad_w = (uint32_t)((1600 - d) / 8);
    } else {
        kmac_head(d, kv.key_len, p);
    }
    p.keys = kv.keys;
    p.key_stride = kv.key_stride;
    body_args(p, m);
    p.absorb_body = absorb_body ? 1 : 0;
    p.suffix = 0x040100ULL;  // right_encode(0) = 00 01 (shake_functions.rs:86), then cSHAKE suffix 0x04 (:57)
    p.suffix_len = 3;
    p.stride_bytes = f.stride;
    p.out_mode = out_mode;
    p.sq_words = f.sq_words;
    p.out = outs;
    p.out_stride = out_stride;
    p.out_len = (uint32_t)out_len;
    p.mask = mask;
    p.n = n;
    return launch_with_pre(f.rw, p, pre_host, s);
}

// SHA3-d (shake, shake_functions.rs:24-32)
int sha3_launch(int d, size_t n, const MsgView &m, uint8_t *digests, uint64_t out_stride, hipStream_t s)
{
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    Framing f = sha3_framing(d);
    SpongeParams p;
    memset(&p, 0, sizeof p);
    body_args(p, m);
    p.absorb_body = 1;
    p.suffix = 0x06;
    p.suffix_len = 1;
    p.sha3_suffix_rule = 1;  // shake_functions.rs:25-29
    p.fips_pad = 0;          // sponge.rs:13: pad only when unaligned
    p.stride_bytes = f.stride;
    p.out_mode = 0;
    p.sq_words = f.sq_words;
    p.out = digests;
    p.out_stride = out_stride;
    p.out_len = (uint32_t)(d / 8);
    p.n = n;
    return launch_sponge(f.rw, p, s);
}

// cSHAKE (cshake, shake_functions.rs:49-64): N, S shared by the batch, no per-item head
// body_has_trailer: the messages already end in the reference's `04 || 06 || pad` trailer (the N = S = "" corner,
// see capy_cshake_batch); no suffix is appended, only the final pad-if-unaligned of sponge_absorb.
int cshake_launch(int d, size_t n, const MsgView &m, size_t l_bits, const uint8_t *fn, size_t fn_len,
                  const uint8_t *cs, size_t cs_len, uint8_t *outs, uint64_t out_stride, hipStream_t s, bool body_has_trailer)
{
    if (l_bits / 8 > CAPY_MAX_OUT_LEN) return fail(CAPY_ERR_ARG, "output longer than 2^32 - 1 bytes per item (l_bits >= 2^35)");
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (fn_len == 0 && cs_len == 0 && !body_has_trailer)
        return fail(CAPY_ERR_UNSUPPORTED,
                    "cshake with empty N and S (shake_functions.rs:59-61): the caller frames the trailer (capy_cshake_batch[_dev])");
    Framing f = cshake_framing(d);
    SpongeParams p;
    memset(&p, 0, sizeof p);
    std::vector<uint8_t> pre_host;
    cshake_prefix(d, fn, fn_len, cs, cs_len, f, p, pre_host);
    body_args(p, m);
    p.absorb_body = 1;
    p.suffix = 0x04;
    p.suffix_len = body_has_trailer ? 0 : 1;
    p.stride_bytes = f.stride;
    p.out_mode = 0;
    p.sq_words = f.sq_words;
    p.out = outs;
    p.out_stride = out_stride;
    p.out_len = (uint32_t)(l_bits / 8);
    p.n = n;
    return launch_with_pre(f.rw, p, pre_host, s);
}

// ---- longest-first processing order for ragged DEVICE batches (host batches are sorted on upload): a stable sort
// by a 512-step logarithmic length scale WITHIN chunks of ORDER_CHUNK consecutive items.  Sorting the whole batch
// scatters the messages of a wave all over the buffer, which costs short messages more than the idle lanes it saves
// (2^21 messages of 0..2 KiB: 11.0 ms globally sorted, 4.4 ms unsorted); inside a 4096-item neighbourhood the lanes of
// a wave still get near-equal lengths and their messages stay within a few MB.
constexpr int ORDER_BUCKETS = 512;
constexpr uint64_t ORDER_CHUNK = 1ull << ORDER_CHUNK_SHIFT;
__device__ __forceinline__ uint32_t len_bucket_desc(uint64_t len)
{
    uint32_t b;
    if (len < 8) {
        b = (uint32_t)len;
    } else {
        const int e = 63 - __clzll((long long)len);  // >= 3
        b = (uint32_t)(e - 2) * 8 + (uint32_t)((len >> (e - 3)) & 7);
    }
    return ORDER_BUCKETS - 1 - b;  // b <= 495
}
__device__ __forceinline__ uint64_t item_len(const uint64_t *offsets, const uint64_t *lens, uint64_t i)
{
    return lens ? lens[i] : offsets[i + 1] - offsets[i];
}
// One workgroup per neighbourhood: keys (length bucket << 12 | index in chunk) sorted ascending by a bitonic network
// in LDS.  The index in the low bits makes the order stable: equal lengths keep their input order, so a batch of
// equal-length messages given through offsets still reads memory sequentially (an atomic-cursor counting sort permuted
// them at random inside each bucket: 2^21 x 1 KiB through offsets 2.5 -> 3.2 ms).
__global__ __launch_bounds__(256) void order_chunk_sort_kernel(const uint64_t *offsets, const uint64_t *lens, uint64_t n,
                                                               uint32_t *order)
{
    __shared__ uint32_t key[ORDER_CHUNK];
    const uint64_t base = (uint64_t)blockIdx.x << ORDER_CHUNK_SHIFT;
    for (uint32_t r = threadIdx.x; r < ORDER_CHUNK; r += blockDim.x) {
        const uint64_t i = base + r;
        key[r] = i < n ? (len_bucket_desc(item_len(offsets, lens, i)) << ORDER_CHUNK_SHIFT) | r : 0xffffffffu;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= ORDER_CHUNK; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < ORDER_CHUNK / 2; t += blockDim.x) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const bool up = (lo & k) == 0;
                const uint32_t a = key[lo], b = key[hi];
                if ((a > b) == up) {
                    key[lo] = b;
                    key[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t r = threadIdx.x; r < ORDER_CHUNK; r += blockDim.x) {
        const uint64_t pos = base + r;
        if (pos < n) order[order_spread((uint32_t)pos, n)] = (uint32_t)base + (key[r] & (uint32_t)(ORDER_CHUNK - 1));
    }
}
int device_order(const uint64_t *offsets, const uint64_t *lens, size_t n, hipStream_t s, const uint32_t **out)
{
    const size_t chunks = (n + ORDER_CHUNK - 1) >> ORDER_CHUNK_SHIFT;
    CAPY_WS(order, uint32_t *, s, WS_ORDER, n * 4);
    hipLaunchKernelGGL(order_chunk_sort_kernel, dim3((unsigned)chunks), dim3(256), 0, s, offsets, lens, (uint64_t)n, order);
    CAPY_HIP(hipGetLastError());
    *out = order;
    return CAPY_OK;
}

// SplitMix64 counter-mode fill (harness PRNG)
__global__ void fill_random_kernel(uint64_t *dst, uint64_t nwords, uint64_t seed)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < nwords; i += stride) {
        uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        dst[i] = z ^ (z >> 31);
    }
}

// VALU ceiling probe: `iters` dependent keccak-f[1600] per lane, nothing else.
template <int VARIANT>
__global__ __launch_bounds__(64) void keccak_probe_kernel(uint64_t n_states, uint32_t iters, uint64_t *checksum)
{
    uint64_t id = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    KState a;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        a.lo[i] = (uint32_t)(id * 25 + i);
        a.hi[i] = (uint32_t)((id * 25 + i) * 0x9E3779B9u);
    }
    for (uint32_t it = 0; it < iters; it++) {
        if (VARIANT == 0)
            keccakf1600_unrolled(a);
        else if (VARIANT == 1)
            keccakf1600(a);
        else if (VARIANT == 2)
            keccakf1600_pipelined(a);
        else
            keccakf1600_paired<true>(a);
    }
    uint32_t x = 0, y = 0;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        x ^= a.lo[i];
        y ^= a.hi[i];
    }
    if (id < n_states && x == 0x12345678u && y == 0x9abcdef0u)  // practically never: keeps the work alive
        atomicXor((unsigned long long *)checksum, ((uint64_t)y << 32) | x);
}

}  // namespace capy

using namespace capy;

extern "C" {

int capy_set_sponge_lanes(int lanes)
{
    // undocumented A/B switches in the high bits; bit 18 of the argument = debug bit 8 (no paired latency-tuned instance)
    // bit 19 = debug bit 9 (blocked two-lane round in forced two-lane launches); bit 20 = debug bit 10 (SPONGE_BLOCK_OUT)
    g_debug_flags.store((((unsigned)lanes >> 8) & 0xff) | ((((unsigned)lanes >> 18) & 7) << 8));
    g_fused_enabled.store((((unsigned)lanes >> 16) & 1) == 0);  // bit 16: disable the fused encrypt kernel
    g_mixed_enabled.store((((unsigned)lanes >> 17) & 1) == 0);  // bit 17: disable the mixed one/two-lane schedule
    lanes &= 0xff;
    if (lanes < 0 || lanes > 3) return fail(CAPY_ERR_ARG, "lanes must be 0 (auto), 1, 2 or 3 (mixed where eligible)");
    g_lanes_per_sponge.store(lanes);
    return CAPY_OK;
}

int capy_debug_last_sponge_kernel(int *kind, int *launches)
{
    if (kind) *kind = t_last_kind;
    if (launches) *launches = t_last_launches;
    return CAPY_OK;
}

int capy_sha3_launch_plan(int d, size_t n, uint64_t uniform_len, uint64_t msg_stride, int *kind, int *phases)
{
    if (!kind || !phases) return fail(CAPY_ERR_ARG, "null output");
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    Framing f = sha3_framing(d);
    SpongeParams p;
    memset(&p, 0, sizeof p);
    p.uniform_len = uniform_len;
    p.msg_stride = msg_stride;
    p.absorb_body = 1;
    p.suffix_len = 1;  // the SHA3 domain-separation byte
    p.stride_bytes = f.stride;
    p.sq_words = f.sq_words;  // as sha3_launch() sets them: the kernel-choice predicates read these
    p.out_len = (uint32_t)(d / 8);
    p.out_stride = (uint64_t)(d / 8);
    p.n = n;
    *kind = sponge_plan(f.rw, p, phases);
    return CAPY_OK;
}

int capy_fill_random_dev(uint8_t *dst, uint64_t nbytes, uint64_t seed, void *stream)
{
    if (((uintptr_t)dst & 7) || (nbytes & 7)) return fail(CAPY_ERR_ARG, "dst and nbytes must be multiples of 8");
    if (!nbytes) return CAPY_OK;
    hipLaunchKernelGGL(fill_random_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, (uint64_t *)dst, nbytes / 8,
                       seed);
    CAPY_HIP(hipGetLastError());
    return CAPY_OK;
}

int capy_keccak_valu_probe_dev(uint64_t n_states, uint32_t iters, uint64_t *checksum_dev, void *stream)
{
    if (!n_states) return CAPY_OK;
    // top two bits of iters select the loop form (0 unrolled = default, 1 rolled, 2 rolled + constant prefetch,
    // 3 the blocked round with raised priority around its rotation blocks: the many-waves form of the kernels)
    const uint32_t variant = iters >> 30, it = iters & 0x3fffffffu;
    const dim3 grid((unsigned)((n_states + 63) / 64));
    if (variant == 0)
        hipLaunchKernelGGL(keccak_probe_kernel<0>, grid, dim3(64), 0, (hipStream_t)stream, n_states, it, checksum_dev);
    else if (variant == 1)
        hipLaunchKernelGGL(keccak_probe_kernel<1>, grid, dim3(64), 0, (hipStream_t)stream, n_states, it, checksum_dev);
    else if (variant == 2)
        hipLaunchKernelGGL(keccak_probe_kernel<2>, grid, dim3(64), 0, (hipStream_t)stream, n_states, it, checksum_dev);
    else
        hipLaunchKernelGGL(keccak_probe_kernel<3>, grid, dim3(64), 0, (hipStream_t)stream, n_states, it, checksum_dev);
    CAPY_HIP(hipGetLastError());
    return CAPY_OK;
}

}  // extern "C"
