// common.h — host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <functional>
#include <string>
#include <vector>
#include "../../include/capyhip.h"

namespace capy {

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define CAPY_HIP(expr)                                                                                    \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return capy::fail(CAPY_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

// Per-call options (capy_call_options, include/capyhip.h): the *_ex entry points set them for the calling thread for
// the duration of the call; the sharded host forms hand them to their per-device workers.  -1 / 0 = the process-wide
// defaults of the setters.
struct CallOpts {
    int hardened = -1;
    int scalar_star = -1;
    int generator = 0;
    void *stream = nullptr;  // host-buffer forms: unused (they work on the calling thread's default stream)
};
CallOpts &thread_opts();
int parse_call_options(const capy_call_options *opt, CallOpts &out);  // CAPY_OK or CAPY_ERR_ARG
struct OptScope {
    CallOpts saved;
    explicit OptScope(const CallOpts &o) : saved(thread_opts()) { thread_opts() = o; }
    OptScope(const OptScope &) = delete;
    OptScope &operator=(const OptScope &) = delete;
    ~OptScope() { thread_opts() = saved; }
};

// Debug / A-B knobs (r04: ONE environment variable, parsed once when the library first needs a knob):
//   CAPY_DEBUG="key=value,key=value,..."   e.g. CAPY_DEBUG="uniform_waves=3,fused_max=65536"
// keys: fused_max, wide_max, mixed_ratio, uniform_waves, rot (0), rot_ratio (sponge launch thresholds); ed448_pair (0 / 1), ed448_wave_max,
// host_overlap (0), host_arena (0), worker_affinity (0).  Unknown keys are reported once on stderr.  A knob that is not
// set returns `dflt`.  These are measurement switches, not product settings (those are function arguments / call options).
double debug_knob(const char *key, double dflt);

// RAII device allocation for the host-pointer entry points, served from a per-thread cache of blocks (sponge.hip:
// devbuf_take / devbuf_give) so that repeated calls neither allocate nor free (= synchronise).
// Buffers of up to ARENA_MAX_BUF bytes come from the thread's ARENA instead: one block of pinned host memory that the device
// maps.  The kernels read such inputs from it and write such outputs to it directly, so a small call makes no copy
// calls at all -- filling it is a memcpy, reading a result is one stream synchronisation and a memcpy (r03: a KMAC tag
// of one 1 KiB message took 112 us through this ABI against 38 us on device buffers, the difference being five small
// synchronous hipMemcpy).  CAPY_DEBUG=host_arena=0 switches it off.  `host` is the CPU's address of an arena buffer.
void *devbuf_take(size_t bytes, size_t *cap);  // nullptr on allocation failure
void devbuf_give(void *p, size_t cap);
constexpr size_t ARENA_MAX_BUF = (1 << 20) + 16;  // measured 16 KiB / 64 KiB / 1 MiB: profiles/r03_small_calls.txt
void *arena_take(size_t bytes, void **host);  // device address, or nullptr: does not fit / no arena
void arena_give();                            // the last buffer given back resets the arena
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;  // requested size
    size_t cap = 0;    // size of the block behind it
    void *host = nullptr;  // non-null: an arena buffer (p is the device's address of the same bytes)
    bool secret = false;  // holds key material (passwords, secret scalars, derived keys): zeroed before it is reused
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf()
    {
        if (!p) return;
        if (host) {
            // nothing may still read or write it when the CPU reuses the bytes (also on the error paths)
            (void)hipStreamSynchronize(nullptr);
            if (secret) memset(host, 0, bytes);
            arena_give();
            return;
        }
        // the host-buffer entry points enqueue on the default stream, so the memset runs after their kernels and
        // before any later user of the block
        if (secret) (void)hipMemsetAsync(p, 0, bytes, nullptr);
        devbuf_give(p, cap);
    }
    hipError_t alloc(size_t n)
    {
        bytes = n ? n : 8;
        if (bytes <= ARENA_MAX_BUF && (p = arena_take(bytes, &host)) != nullptr) return hipSuccess;
        host = nullptr;
        p = devbuf_take(bytes, &cap);
        return p ? hipSuccess : hipErrorOutOfMemory;
    }
    template <class T>
    T *as() const
    {
        return reinterpret_cast<T *>(p);
    }
    // host -> buffer / buffer -> host at a byte offset.  Arena buffers: plain memory; a read first waits for the thread's
    // default stream (callers that launched on a side stream have synchronised it themselves, as before)
    hipError_t put(const void *src, size_t n, size_t off = 0) const
    {
        if (!n) return hipSuccess;
        if (host) {
            memcpy(static_cast<char *>(host) + off, src, n);
            return hipSuccess;
        }
        return hipMemcpy(static_cast<char *>(p) + off, src, n, hipMemcpyHostToDevice);
    }
    hipError_t get(void *dst, size_t n, size_t off = 0) const
    {
        if (!n) return hipSuccess;
        if (host) {
            const hipError_t e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) return e;
            memcpy(dst, static_cast<const char *>(host) + off, n);
            return hipSuccess;
        }
        return hipMemcpy(dst, static_cast<const char *>(p) + off, n, hipMemcpyDeviceToHost);
    }
};

// Grow-only device scratch, one set of named slots per (thread, device, stream).  All users of a slot
// enqueue on that stream, so reuse across calls is ordered by the stream itself; a slot that has to grow
// gets a new block and the old one is retired until capy_release_workspace() / thread exit, so no call
// ever frees (= synchronises) on the way.  (hipMallocAsync / hipFreeAsync proved unreliable on the default
// stream of this ROCm build: results raced.)
enum WsSlot { WS_PRE = 0, WS_ZPW, WS_ZOFF, WS_KEKA, WS_TAG2, WS_A, WS_B, WS_C, WS_D, WS_E, WS_F, WS_TABLE, WS_STATE, WS_ORDER, WS_NSLOTS };
void *workspace(hipStream_t stream, WsSlot slot, size_t bytes);  // nullptr on allocation failure
void workspace_release();                                           // free this thread's scratch (synchronises)
void workspace_scrub(hipStream_t stream, WsSlot slot, size_t bytes);  // zero a slot's first bytes, stream-ordered
void workspace_scrub_many(hipStream_t stream, const WsSlot *slots, const size_t *bytes, int count);  // the same, one launch
// nesting of scrub guards on the calling thread (+1 / -1): the outermost guard of a call starts a new record of scrubbed ranges
// for the test hook capy_debug_secret_scratch_nonzero, inner guards (keyed sponge states of a phase schedule) add to it
void workspace_scrub_scope(int delta);
// a slot of the calling thread's scratch as a typed pointer, or return CAPY_ERR_HIP from the enclosing function
#define CAPY_WS(var, type, stream, slot, bytes)                                      \
    type var = reinterpret_cast<type>(capy::workspace(stream, slot, bytes));          \
    if (!var) return capy::fail(CAPY_ERR_HIP, "workspace allocation failed")

// Scrubs the named slots when the enclosing function returns -- on EVERY path, also the early error returns (a failed
// launch must not leave z || pw, ke || ka, a secret scalar or an ECDH point behind in scratch that later calls reuse).
struct WsScrubGuard {
    hipStream_t stream;
    struct Item {
        WsSlot slot;
        size_t bytes;
    } items[4];
    int count = 0;
    explicit WsScrubGuard(hipStream_t s) : stream(s) { workspace_scrub_scope(+1); }
    WsScrubGuard(const WsScrubGuard &) = delete;
    WsScrubGuard &operator=(const WsScrubGuard &) = delete;
    void add(WsSlot slot, size_t bytes)
    {
        if (count < 4) items[count++] = {slot, bytes};
    }
    ~WsScrubGuard()
    {
        WsSlot slots[4];
        size_t bytes[4];
        for (int i = 0; i < count; i++) {
            slots[i] = items[i].slot;
            bytes[i] = items[i].bytes;
        }
        workspace_scrub_many(stream, slots, bytes, count);
        workspace_scrub_scope(-1);
    }
};

// Messages of a host batch on the device.  If every message already starts on an 8-byte boundary the
// packed buffer is copied as is; otherwise it is re-laid out so that every message starts on a
// 16-byte boundary (the kernels' coalesced fast path needs 8-byte aligned message starts).
struct PackedBatch {
    DevBuf msgs, starts, lens, order;    // device: bytes, n+1 starts, n lengths, processing order (ragged only)
    bool has_order = false;
    std::vector<uint64_t> h_starts, h_lens;
    uint64_t total = 0;
    bool repacked = false;
    // set when every message has the same length and the starts are equally spaced: the kernels then take the
    // wave-uniform addressing path (and the rotating schedule) exactly as for a strided device batch
    bool uniform = false;
    uint64_t uniform_len = 0, uniform_stride = 0;
    int upload(size_t n, const uint8_t *host_msgs, const uint64_t *host_offsets);
    // copy message bytes back into the caller's packed layout
    int download(size_t n, uint8_t *host_msgs, const uint64_t *host_offsets) const;
};

// ---- multi-device execution of the host-buffer entry points (capy_set_devices, include/capyhip.h).
// Items are independent, so a batch is cut into contiguous shards, one per configured device, balanced by message
// bytes where the call carries messages (byte_offsets: n+1 offsets) and by count otherwise; one worker thread per
// shard selects its device and runs the single-device body on its slice of the caller's arrays.  No collective, no
// peer traffic; output order and bytes are those of the single-device call for every device list.
// Returns false when no device list is configured (the calling thread's current device is used).
bool configured_devices(std::vector<int> &ids);
int run_sharded(const std::vector<int> &ids, size_t n, const uint64_t *byte_offsets,
                const std::function<int(size_t first, size_t count)> &body);
// CAPY_SHARD(n, offsets, call): run `call` (an expression in `first` / `count`) on every shard when a device list is
// configured, and return from the enclosing function with its status
#define CAPY_SHARD(n, byte_offsets, call)                                                                       \
    do {                                                                                                         \
        std::vector<int> _ids;                                                                                   \
        if (capy::configured_devices(_ids))                                                                      \
            return capy::run_sharded(_ids, (n), (byte_offsets), [&](size_t first, size_t count) { return (call); }); \
    } while (0)

// Argument checks of the device-buffer entry points: a null pointer that a kernel would dereference must come back as
// CAPY_ERR_ARG, never as a GPU fault.  Message / key buffers may be null only when they are empty by construction.
#define CAPY_REQUIRE(cond, what)                                              \
    do {                                                                       \
        if (!(cond)) return capy::fail(CAPY_ERR_ARG, "null or invalid argument: " what); \
    } while (0)
inline bool msgs_ok(const uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len)
{
    return msgs != nullptr || (offsets == nullptr && uniform_len == 0);
}
inline bool keys_ok(const uint8_t *keys, size_t key_len, const uint64_t *key_offsets)
{
    return keys != nullptr || (key_offsets == nullptr && key_len == 0);
}

// longest per-item key / password (the per-item head builder of the kernels encodes 8*|K| in at most three bytes)
constexpr size_t CAPY_MAX_KEY_LEN = (size_t)1 << 20;
// longest squeeze per item: SpongeParams::out_len is 32 bits wide
constexpr size_t CAPY_MAX_OUT_LEN = 0xffffffffULL;

inline bool valid_d(int d) { return d == 224 || d == 256 || d == 384 || d == 512; }

}  // namespace capy
