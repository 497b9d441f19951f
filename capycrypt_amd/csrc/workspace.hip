// workspace.hip — device scratch pools, the staging-buffer cache and the pinned host arena behind DevBuf, secret
// scrubbing, and the host <-> device movement of packed message / key batches (PackedBatch, PackedKeys).
#include <string.h>
#include <algorithm>
#include <atomic>
#include <string>
#include <vector>
#include "common.h"
#include <mutex>
#include "sponge_host.h"

namespace capy {

// ------------------------------------------------------------------ workspace
namespace {
struct WsEntry {
    int device;
    hipStream_t stream;
    void *ptr[WS_NSLOTS];
    size_t cap[WS_NSLOTS];
};
// One list per host thread.  A slot that has to grow gets a new, larger block; the old block is RETIRED, not freed --
// kernels already enqueued may still use it, and hipFree would synchronise the device, which the *_dev entry points
// promise not to do.  Retired blocks (less than the final size in total, the growth is geometric) and the live ones
// are returned by capy_release_workspace() or when the thread ends (at process exit that runs before the HIP
// runtime's own teardown; a late hipFree only returns an error).
struct CachedBlock {
    void *p;
    size_t cap;
    int device;
};
struct WsList {
    std::vector<WsEntry> v;
    std::vector<void *> retired;
    std::vector<CachedBlock> cache;  // device blocks of finished host-buffer calls (DevBuf), see devbuf_take
    void release()
    {
        for (auto &w : v)
            for (void *q : w.ptr)
                if (q) (void)hipFree(q);
        for (void *q : retired) (void)hipFree(q);
        for (auto &c : cache) (void)hipFree(c.p);
        v.clear();
        retired.clear();
        cache.clear();
    }
    ~WsList() { release(); }
};
thread_local WsList g_ws_list;
}  // namespace

void *workspace(hipStream_t stream, WsSlot slot, size_t bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    WsEntry *e = nullptr;
    std::vector<WsEntry> &g_ws = g_ws_list.v;
    for (auto &w : g_ws)
        if (w.device == dev && w.stream == stream) e = &w;
    if (!e) {
        g_ws.push_back(WsEntry{dev, stream, {}, {}});
        e = &g_ws.back();
    }
    if (bytes == 0) bytes = 8;
    if (e->cap[slot] < bytes) {
        const size_t cap = bytes + bytes / 2 + 256;
        void *fresh = nullptr;
        if (hipMalloc(&fresh, cap) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (e->ptr[slot]) g_ws_list.retired.push_back(e->ptr[slot]);
        e->ptr[slot] = fresh;
        e->cap[slot] = cap;
    }
    return e->ptr[slot];
}

void arena_release();
void workspace_release()
{
    g_ws_list.release();
    arena_release();
}

// Device blocks of the host-buffer entry points (DevBuf: message / key / output staging).  r02 paid a hipMalloc and a
// hipFree -- a device synchronisation -- per buffer per call; now a finished call's blocks wait in a per-thread cache
// and the next call of that thread takes the smallest one that fits without wasting more than half of it.  All users
// enqueue on the thread's default stream (or have synchronised their side stream before the DevBuf dies), so reuse is
// stream-ordered.  At most 24 blocks are kept (the oldest go first); capy_release_workspace() / thread exit frees them.
void *devbuf_take(size_t bytes, size_t *cap)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    auto &c = g_ws_list.cache;
    size_t best = c.size();
    for (size_t i = 0; i < c.size(); i++)
        if (c[i].device == dev && c[i].cap >= bytes && c[i].cap <= 2 * bytes + 4096 && (best == c.size() || c[i].cap < c[best].cap))
            best = i;
    if (best != c.size()) {
        void *p = c[best].p;
        *cap = c[best].cap;
        c.erase(c.begin() + best);
        return p;
    }
    const size_t want = (bytes + 255) & ~(size_t)255;
    void *p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        // memory may be held by the cache itself: drop it and try once more
        (void)hipGetLastError();
        for (auto &b : c) (void)hipFree(b.p);
        c.clear();
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    *cap = want;
    return p;
}
void devbuf_give(void *p, size_t cap)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipFree(p);
        return;
    }
    auto &c = g_ws_list.cache;
    c.push_back(CachedBlock{p, cap, dev});
    if (c.size() > 24) {
        (void)hipFree(c.front().p);
        c.erase(c.begin());
    }
}

// ---- the per-thread arena of small DevBufs (common.h): 8 MiB of pinned, device-mapped, portable host memory, bump
// allocated in 256-byte steps, reset when the last buffer of a call has been given back
namespace {
struct HostArena {
    char *host = nullptr, *dev = nullptr;
    size_t used = 0, live = 0;
    bool tried = false;
    static constexpr size_t SIZE = 8 * 1024 * 1024;
    ~HostArena()
    {
        if (host) (void)hipHostFree(host);
    }
};
thread_local HostArena t_arena;
}  // namespace
void *arena_take(size_t bytes, void **host)
{
    HostArena &a = t_arena;
    if (!a.tried) {
        a.tried = true;
        if (debug_knob("host_arena", 1) != 0) {
            void *h = nullptr, *d = nullptr;
            if (hipHostMalloc(&h, HostArena::SIZE, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess &&
                hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                a.host = (char *)h;
                a.dev = (char *)d;
            } else {
                (void)hipGetLastError();
                if (h) (void)hipHostFree(h);
            }
        }
    }
    const size_t step = (bytes + 255) & ~(size_t)255;
    if (!a.host || a.used + step > HostArena::SIZE) return nullptr;
    *host = a.host + a.used;
    void *d = a.dev + a.used;
    a.used += step;
    a.live++;
    return d;
}
void arena_release()  // capy_release_workspace(): the pinned block goes back too (no call of this thread is in flight)
{
    HostArena &a = t_arena;
    if (a.host && a.live == 0) {
        (void)hipHostFree(a.host);
        a.host = a.dev = nullptr;
        a.used = 0;
        a.tried = false;
    }
}
void arena_give()
{
    HostArena &a = t_arena;
    if (a.live && --a.live == 0) a.used = 0;
}

// Secret intermediates (z||pw, ke||ka, the Schnorr secret s and nonce k, the ECDH point W) sit in pooled scratch that
// later, unrelated calls reuse: zero them on the same stream once the call's last reader has been enqueued.
void workspace_scrub(hipStream_t stream, WsSlot slot, size_t bytes)
{
    int dev = 0;
    if (!bytes || hipGetDevice(&dev) != hipSuccess) return;
    for (auto &w : g_ws_list.v)
        if (w.device == dev && w.stream == stream && w.ptr[slot])
            (void)hipMemsetAsync(w.ptr[slot], 0, std::min(bytes, w.cap[slot]), stream);
}

// up to four ranges zeroed by ONE launch (a protocol call on one item spent 4 x 5 us in four memsets, 7 % of a signature)
struct ScrubRanges {
    uint8_t *ptr[4];
    uint64_t bytes[4];
};
__global__ __launch_bounds__(256) void scrub_kernel(const ScrubRanges r)
{
    uint8_t *p = r.ptr[blockIdx.y];
    const uint64_t nb = r.bytes[blockIdx.y], quads = nb / 16;  // the slots are 256-byte aligned allocations
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += stride)
        reinterpret_cast<uint4 *>(p)[q] = uint4{0, 0, 0, 0};
    if (blockIdx.x == 0 && threadIdx.x < (nb & 15)) p[quads * 16 + threadIdx.x] = 0;
}
// what the calling thread's LAST guarded call scrubbed (capy_debug_secret_scratch_nonzero): a protocol call nests scrub guards
// (z || pw and ke || ka outside, the keyed sponge states of a phase schedule inside), so the ranges of one outermost scope
// accumulate -- at most 16, one entry per block; the next outermost guard starts afresh
static thread_local std::vector<std::pair<uint8_t *, uint64_t>> t_scrubbed;
static thread_local int t_scrub_depth = 0;
void workspace_scrub_scope(int delta)
{
    if (delta > 0 && t_scrub_depth++ == 0) t_scrubbed.clear();
    if (delta < 0 && t_scrub_depth > 0) t_scrub_depth--;
}
static void note_scrubbed(uint8_t *p, uint64_t nb)
{
    for (auto &e : t_scrubbed)
        if (e.first == p) {
            e.second = std::max(e.second, nb);
            return;
        }
    if (t_scrubbed.size() >= 16) t_scrubbed.erase(t_scrubbed.begin());
    t_scrubbed.emplace_back(p, nb);
}
void workspace_scrub_many(hipStream_t stream, const WsSlot *slots, const size_t *bytes, int count)
{
    int dev = 0;
    if (count <= 0 || hipGetDevice(&dev) != hipSuccess) return;
    ScrubRanges r{};
    int k = 0;
    uint64_t most = 0;
    for (auto &w : g_ws_list.v) {
        if (w.device != dev || w.stream != stream) continue;
        for (int i = 0; i < count && k < 4; i++) {
            const size_t nb = std::min(bytes[i], w.cap[slots[i]]);
            if (!nb || !w.ptr[slots[i]]) continue;
            r.ptr[k] = (uint8_t *)w.ptr[slots[i]];
            r.bytes[k] = nb;
            most = std::max<uint64_t>(most, nb);
            k++;
        }
    }
    if (!k) return;
    for (int i = 0; i < k; i++) note_scrubbed(r.ptr[i], r.bytes[i]);
    const unsigned gx = (unsigned)std::min<uint64_t>((most / 16 + 255) / 256 + 1, 4096);
    hipLaunchKernelGGL(scrub_kernel, dim3(gx, (unsigned)k), dim3(256), 0, stream, r);
    if (hipGetLastError() != hipSuccess)  // never leave secrets behind because a launch failed: fall back to memsets
        for (int i = 0; i < count; i++) workspace_scrub(stream, slots[i], bytes[i]);
}



// ------------------------------------------------------------------ bulk host <-> device copies
// A first hipMemcpy from freshly allocated pageable memory runs at ~5.7 GiB/s on this platform (the runtime pins
// it piecemeal); registering the range first costs 0.05 s per GiB and the copy then runs at 53 GiB/s
// (tools/h2d_probe.hip: 14.6 GiB/s cold overall, no loss when the pages are already pinned).  Used for the message
// buffers only; registration failures (read-only mappings, limits) fall back to the plain copy.
static const size_t BULK_COPY_MIN = (size_t)32 << 20;
static hipError_t bulk_copy(void *dst, const void *src, size_t n, hipMemcpyKind kind)
{
    void *host = kind == hipMemcpyHostToDevice ? const_cast<void *>(src) : dst;
    bool registered = false;
    if (n >= BULK_COPY_MIN) {
        registered = hipHostRegister(host, n, hipHostRegisterDefault) == hipSuccess;
        if (!registered) (void)hipGetLastError();  // clear the sticky error of the failed attempt
    }
    const hipError_t e = hipMemcpy(dst, src, n, kind);
    if (registered) (void)hipHostUnregister(host);
    return e;
}

// n rows of `row` bytes, `stride` apart in the buffer, to a dense host array
hipError_t copy_rows_out(uint8_t *dst, size_t row, const DevBuf &b, size_t stride, size_t n)
{
    if (!b.host) return hipMemcpy2D(dst, row, b.p, stride, row, n, hipMemcpyDeviceToHost);
    const hipError_t e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return e;
    for (size_t i = 0; i < n; i++) memcpy(dst + i * row, static_cast<const char *>(b.host) + i * stride, row);
    return hipSuccess;
}

// ------------------------------------------------------------------ PackedBatch
int PackedBatch::upload(size_t n, const uint8_t *host_msgs, const uint64_t *host_offsets)
{
    h_starts.assign(n + 1, 0);
    h_lens.assign(n ? n : 1, 0);
    bool aligned = true;
    for (size_t i = 0; i < n; i++) {
        if (host_offsets[i + 1] < host_offsets[i]) return fail(CAPY_ERR_ARG, "offsets must be non-decreasing");
        if ((host_offsets[i] - host_offsets[0]) & 7) aligned = false;
        h_lens[i] = host_offsets[i + 1] - host_offsets[i];
    }
    if (!host_msgs && n && host_offsets[n] != host_offsets[0]) return fail(CAPY_ERR_ARG, "null message buffer");
    repacked = !aligned;
    std::vector<uint8_t> staging;
    const uint8_t *src = host_msgs ? host_msgs + (n ? host_offsets[0] : 0) : nullptr;
    if (aligned) {
        for (size_t i = 0; i <= n; i++) h_starts[i] = host_offsets[i] - host_offsets[0];
        total = h_starts[n];
    } else {
        uint64_t pos = 0;
        for (size_t i = 0; i < n; i++) {
            h_starts[i] = pos;
            pos = (pos + h_lens[i] + 15) & ~15ULL;
        }
        h_starts[n] = pos;
        total = pos;
        staging.assign(total + 16, 0);
        for (size_t i = 0; i < n; i++)
            if (h_lens[i]) memcpy(staging.data() + h_starts[i], host_msgs + host_offsets[i], h_lens[i]);
        src = staging.data();
    }
    uniform = n > 0;
    uniform_len = n ? h_lens[0] : 0;
    uniform_stride = n > 1 ? h_starts[1] - h_starts[0] : (uniform_len + 7) & ~7ULL;
    for (size_t i = 0; i < n && uniform; i++)
        uniform = h_lens[i] == uniform_len && h_starts[i] == i * uniform_stride;
    if ((uniform_stride & 7) || uniform_stride < uniform_len) uniform = false;
    // Ragged batch: process the items longest-first.  All lanes of a wave run until the wave's longest message is
    // done, so grouping similar lengths removes the idle lanes (and the longest waves start first).
    has_order = !uniform && n >= 128 && n <= 0xffffffffULL && !(sponge_debug_flags() & 4);  // debug bit 2: A/B switch
    std::vector<uint32_t> h_order;
    if (has_order) {
        h_order.resize(n);
        for (size_t i = 0; i < n; i++) h_order[i] = (uint32_t)i;
        // within neighbourhoods of 4096 items (see device_order): keeps a wave's messages close together in memory
        for (size_t c0 = 0; c0 < n; c0 += 4096)
            std::stable_sort(h_order.begin() + c0, h_order.begin() + std::min(n, c0 + 4096),
                             [&](uint32_t a, uint32_t b) { return h_lens[a] > h_lens[b]; });
        std::vector<uint32_t> spread(n);
        for (size_t k = 0; k < n; k++) spread[order_spread((uint32_t)k, n)] = h_order[k];
        CAPY_HIP(order.alloc(n * 4));
        CAPY_HIP(order.put(spread.data(), n * 4));
    }
    CAPY_HIP(msgs.alloc(total + 16));
    if (total) CAPY_HIP(msgs.host ? msgs.put(src, total) : bulk_copy(msgs.p, src, total, hipMemcpyHostToDevice));
    if (!uniform) {  // a uniform batch is described by (length, stride) alone
        CAPY_HIP(starts.alloc((n + 1) * 8));
        CAPY_HIP(lens.alloc((n ? n : 1) * 8));
        CAPY_HIP(starts.put(h_starts.data(), (n + 1) * 8));
        CAPY_HIP(lens.put(h_lens.data(), (n ? n : 1) * 8));
    }
    return CAPY_OK;
}

int PackedBatch::download(size_t n, uint8_t *host_msgs, const uint64_t *host_offsets) const
{
    if (!n || !total) return CAPY_OK;
    if (!repacked) {
        CAPY_HIP(msgs.host ? msgs.get(host_msgs + host_offsets[0], total) : bulk_copy(host_msgs + host_offsets[0], msgs.p, total, hipMemcpyDeviceToHost));
        return CAPY_OK;
    }
    std::vector<uint8_t> staging(total);
    CAPY_HIP(msgs.host ? msgs.get(staging.data(), total) : bulk_copy(staging.data(), msgs.p, total, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++)
        if (h_lens[i]) memcpy(host_msgs + host_offsets[i], staging.data() + h_starts[i], h_lens[i]);
    return CAPY_OK;
}

int PackedKeys::upload(size_t n, const uint8_t *keys, size_t key_len, const uint64_t *offsets)
{
    data.secret = true;  // KMAC keys / passwords: zeroed before the buffer is freed
    if (!offsets) {
        if (key_len > CAPY_MAX_KEY_LEN) return fail(CAPY_ERR_ARG, "key too long");
        total = (uint64_t)n * key_len;
        if (total && !keys) return fail(CAPY_ERR_ARG, "null key buffer");
        CAPY_HIP(data.alloc(total));
        if (total) CAPY_HIP(data.put(keys, total));
        view = fixed_keys(data.as<uint8_t>(), key_len, key_len);
        return CAPY_OK;
    }
    std::vector<uint64_t> rel(n + 1);
    for (size_t i = 0; i <= n; i++) {
        if (i && offsets[i] < offsets[i - 1]) return fail(CAPY_ERR_ARG, "key offsets must be non-decreasing");
        if (i && offsets[i] - offsets[i - 1] > CAPY_MAX_KEY_LEN) return fail(CAPY_ERR_ARG, "key too long");
        rel[i] = offsets[i] - offsets[0];
    }
    total = rel[n];
    if (total && !keys) return fail(CAPY_ERR_ARG, "null key buffer");
    CAPY_HIP(data.alloc(total));
    if (total) CAPY_HIP(data.put(keys + offsets[0], total));
    CAPY_HIP(offs.alloc((n + 1) * 8));
    CAPY_HIP(offs.put(rel.data(), (n + 1) * 8));
    view = KeyView();
    view.keys = data.as<uint8_t>();
    view.key_offsets = offs.as<uint64_t>();
    return CAPY_OK;
}

MsgView view_of(const PackedBatch &b)
{
    MsgView m;
    m.msgs = b.msgs.as<uint8_t>();
    m.aligned8 = true;  // PackedBatch keeps or makes every start 8-byte aligned
    if (b.uniform) {
        m.uniform_len = b.uniform_len;
        m.msg_stride = b.uniform_stride;
    } else {
        m.offsets = b.starts.as<uint64_t>();
        m.lens = b.lens.as<uint64_t>();
        if (b.has_order) m.order = b.order.as<uint32_t>();
    }
    return m;
}

MsgView view_dev(const uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride)
{
    MsgView m;
    m.msgs = msgs;
    m.offsets = offsets;
    m.uniform_len = uniform_len;
    m.msg_stride = msg_stride;
    m.aligned8 = offsets == nullptr && (((uintptr_t)msgs | msg_stride) & 7) == 0;  // device offsets are not inspected
    return m;
}

}  // namespace capy

using namespace capy;

extern "C" {

// ---------------------------------------------------------------- measurement helpers
int capy_release_workspace(void)
{
    workspace_release();
    return CAPY_OK;
}

int capy_debug_secret_scratch_nonzero(void *stream, uint64_t *nonzero_bytes)
{
    CAPY_REQUIRE(nonzero_bytes != nullptr, "nonzero_bytes");
    CAPY_HIP(hipStreamSynchronize((hipStream_t)stream));
    uint64_t total = 0;
    for (const auto &e : t_scrubbed) {
        std::vector<uint8_t> h(e.second);
        CAPY_HIP(hipMemcpy(h.data(), e.first, h.size(), hipMemcpyDeviceToHost));
        for (uint8_t b : h) total += b != 0;
    }
    *nonzero_bytes = total;
    return CAPY_OK;
}

}  // extern "C"
