// sponge.hip — launchers and C-ABI entry points of the batched sponge path (SHA3 / cSHAKE / KMACXOF /
// sha3_encrypt / sha3_decrypt).  Host code here only frames the call (SP 800-185 prefixes, exactly
// as the reference builds them) and launches kernels; there is NO CPU fallback for the data path.
#include <string.h>
#include <algorithm>
#include <atomic>
#include <ctype.h>
#include <condition_variable>
#include <memory>
#include <deque>
#include <mutex>
#include <thread>
#include <sched.h>
#include "common.h"
#include "sponge_launch.h"
#include "sponge_fused.h"
#include "sponge_mixed.h"
#include "sponge_host.h"

namespace capy {

// ------------------------------------------------------------------ error plumbing
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

// ------------------------------------------------------------------ multi-device sharding (see common.h)
double debug_knob(const char *key, double dflt)
{
    static const std::vector<std::pair<std::string, double>> knobs = [] {
        std::vector<std::pair<std::string, double>> v;
        static const char *known[] = {"fused_max", "wide_max", "mixed_ratio", "uniform_waves", "ed448_pair", "ed448_wave_max",
                                      "host_overlap", "host_arena", "worker_affinity"};
        const char *e = getenv("CAPY_DEBUG");
        std::string txt = e ? e : "";
        size_t pos = 0;
        while (pos < txt.size()) {
            size_t end = txt.find(',', pos);
            if (end == std::string::npos) end = txt.size();
            const std::string item = txt.substr(pos, end - pos);
            pos = end + 1;
            const size_t eq = item.find('=');
            if (item.empty()) continue;
            const std::string k = item.substr(0, eq);
            bool ok = false;
            for (const char *n : known) ok = ok || k == n;
            if (!ok || eq == std::string::npos) {
                fprintf(stderr, "libcapyhip: CAPY_DEBUG: unknown or malformed item '%s' ignored\n", item.c_str());
                continue;
            }
            v.emplace_back(k, atof(item.c_str() + eq + 1));
        }
        return v;
    }();
    for (const auto &kv : knobs)
        if (kv.first == key) return kv.second;
    return dflt;
}

CallOpts &thread_opts()
{
    static thread_local CallOpts o;
    return o;
}
int parse_call_options(const capy_call_options *opt, CallOpts &out)
{
    out = thread_opts();  // nested calls inherit
    if (!opt) return CAPY_OK;
    if (opt->struct_size < sizeof(capy_call_options)) return fail(CAPY_ERR_ARG, "capy_call_options::struct_size is too small (use CAPY_CALL_OPTIONS_INIT)");
    if (opt->hardened != CAPY_OPT_DEFAULT && opt->hardened != CAPY_HARDEN_OFF && opt->hardened != CAPY_HARDEN_ALL &&
        opt->hardened != CAPY_HARDEN_PROTOCOL)
        return fail(CAPY_ERR_ARG, "capy_call_options::hardened must be CAPY_OPT_DEFAULT or a CAPY_HARDEN_* value");
    if (opt->scalar_star != CAPY_OPT_DEFAULT && (opt->scalar_star < 0 || opt->scalar_star > 2))
        return fail(CAPY_ERR_ARG, "capy_call_options::scalar_star must be CAPY_OPT_DEFAULT, 0, 1 or 2");
    if (opt->generator < 0) return fail(CAPY_ERR_ARG, "capy_call_options::generator must be a handle (0 = the process generator)");
    if (opt->hardened != CAPY_OPT_DEFAULT) out.hardened = opt->hardened;
    if (opt->scalar_star != CAPY_OPT_DEFAULT) out.scalar_star = opt->scalar_star;
    out.generator = opt->generator;
    out.stream = opt->stream;
    return CAPY_OK;
}

static std::mutex g_dev_mu;
static std::vector<int> g_dev_ids;  // empty: not configured
// a worker of run_sharded never shards again (its body is the single-device form of the same entry point)
static thread_local bool g_in_shard = false;

bool configured_devices(std::vector<int> &ids)
{
    if (g_in_shard) return false;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    ids = g_dev_ids;
    return !ids.empty();
}

// contiguous shard bounds: by bytes (lengths from n+1 offsets, an item goes to the shard its midpoint falls in --
// the rule of capycrypt_amd/sharding.py: shard_by_bytes) or by count
static std::vector<size_t> shard_bounds(size_t n, size_t world, const uint64_t *off)
{
    std::vector<size_t> b(world + 1, n);
    b[0] = 0;
    const uint64_t total = off ? off[n] - off[0] : 0;
    if (!off || total == 0) {
        const size_t base = n / world, extra = n % world;
        for (size_t r = 1; r < world; r++) b[r] = r * base + std::min(r, extra);
        return b;
    }
    size_t i = 0;
    for (size_t r = 1; r < world; r++) {
        const long double target = (long double)total * r / world;
        while (i < n && (long double)(off[i] - off[0]) + (long double)(off[i + 1] - off[i]) / 2 <= target) i++;
        b[r] = i;
    }
    return b;
}

// ---- persistent workers (r03).  One long-lived host thread per position of the device list: it selects its device
// once, pins itself to the CPUs the device is attached to (/sys/bus/pci/devices/<bdf>/local_cpulist -- SURVEY 8(e) names
// NUMA placement of the staging as the scaling risk), and keeps its thread-local scratch pools and device-buffer cache
// (workspace(), DevBuf) from call to call.  r02 spawned fresh std::threads per call: every sharded call re-allocated
// its pools and freed them (a device synchronisation each) at thread exit.  Sharded calls from several host threads
// take turns (one pool).
namespace {
// "0-15,128-143" -> CPU set; empty on any parse problem
static bool parse_cpulist(const char *text, cpu_set_t *set)
{
    CPU_ZERO(set);
    int any = 0;
    const char *q = text;
    while (*q) {
        char *end = nullptr;
        long a = strtol(q, &end, 10);
        if (end == q) break;
        long b = a;
        q = end;
        if (*q == '-') {
            b = strtol(q + 1, &end, 10);
            if (end == q + 1) return false;
            q = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) {
            CPU_SET((int)c, set);
            any++;
        }
        while (*q == ',' || *q == '\n' || *q == ' ') q++;
    }
    return any > 0;
}
static void pin_to_device_cpus(int device)
{
    if (debug_knob("worker_affinity", 1) == 0) return;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return;
    char line[4096] = {0};
    const bool ok = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    cpu_set_t want, have, both;
    if (!ok || !parse_cpulist(line, &want) || sched_getaffinity(0, sizeof have, &have) != 0) return;
    CPU_AND(&both, &want, &have);  // never leave the CPUs this process may use (containers)
    if (CPU_COUNT(&both) > 0) (void)sched_setaffinity(0, sizeof both, &both);
}

// completion flag of one submitted job; shared with the submitting thread, so it outlives a worker that is replaced
struct Latch {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    void set()
    {
        std::lock_guard<std::mutex> lk(mu);
        done = true;
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
    }
};
struct Worker {
    int device = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    // a FIFO per worker (r04): sharded calls from several host threads queue their shards and wait on their own
    // latches, so the pool lock is held only while a call submits -- r03 held it for the whole call and serialised the
    // callers, PCIe copies included
    std::deque<std::pair<std::function<void()>, std::shared_ptr<Latch>>> q;
    bool quit = false;
    void loop()
    {
        g_in_shard = true;  // a worker never shards again: its body is the single-device form of the entry point
        const bool dev_ok = hipSetDevice(device) == hipSuccess;
        if (!dev_ok) (void)hipGetLastError();
        pin_to_device_cpus(device);
        std::unique_lock<std::mutex> lk(mu);
        while (true) {
            cv.wait(lk, [&] { return !q.empty() || quit; });
            if (q.empty()) break;  // quit, and every queued shard has run
            auto item = std::move(q.front());
            q.pop_front();
            lk.unlock();
            item.first();
            item.second->set();
            lk.lock();
        }
        lk.unlock();
        workspace_release();  // on this thread: its scratch pools and buffer cache (the list changed; the runtime is alive)
    }
    std::shared_ptr<Latch> submit(std::function<void()> f)
    {
        auto l = std::make_shared<Latch>();
        std::lock_guard<std::mutex> lk(mu);
        q.emplace_back(std::move(f), l);
        cv.notify_all();
        return l;
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};
struct WorkerPool {
    std::mutex mu;  // guards ids / workers; held while a call submits its shards, not while they run
    std::vector<int> ids;
    std::vector<std::unique_ptr<Worker>> workers;
    void reset(const std::vector<int> &want)
    {
        for (auto &w : workers) w->stop();  // finishes what is queued first
        workers.clear();
        ids = want;
        for (int id : want) {
            workers.emplace_back(new Worker);
            Worker *w = workers.back().get();
            w->device = id;
            w->th = std::thread([w] { w->loop(); });
        }
    }
    ~WorkerPool()
    {
        // Process exit: the HIP runtime may already be shutting down, so the workers must not run their scratch release
        // (nor the thread-local destructors that free device memory).  They are left blocked on their condition
        // variables -- the process ends them -- and their Worker objects are deliberately not destroyed.
        for (auto &w : workers) {
            if (w->th.joinable()) w->th.detach();
            (void)w.release();
        }
    }
};
static WorkerPool g_pool;
}  // namespace

int run_sharded(const std::vector<int> &ids, size_t n, const uint64_t *byte_offsets,
                const std::function<int(size_t, size_t)> &body)
{
    const size_t world = ids.size();
    if (world == 1) {
        // one device: on the calling thread, as a plain single-device call on that device (no worker, no lock)
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur != ids[0] && hipSetDevice(ids[0]) != hipSuccess) {
            (void)hipGetLastError();
            return fail(CAPY_ERR_HIP, "hipSetDevice(" + std::to_string(ids[0]) + ") failed");
        }
        g_in_shard = true;
        const int rc = body(0, n);
        g_in_shard = false;
        if (cur >= 0 && cur != ids[0]) (void)hipSetDevice(cur);
        return rc;
    }
    const std::vector<size_t> b = shard_bounds(n, world, byte_offsets);
    std::vector<int> rcs(world, CAPY_OK);
    std::vector<std::string> errs(world);
    std::vector<std::shared_ptr<Latch>> latches;
    const CallOpts opts = thread_opts();  // the caller's per-call options travel with its shards
    {
        std::lock_guard<std::mutex> pool_lock(g_pool.mu);
        if (g_pool.ids != ids) g_pool.reset(ids);  // first call after capy_set_devices (or a changed list)
        for (size_t r = 0; r < world; r++) {
            if (b[r + 1] <= b[r]) continue;
            latches.push_back(g_pool.workers[r]->submit([&, r] {
                int cur = -1;
                if (hipGetDevice(&cur) != hipSuccess || cur != ids[r]) {
                    if (hipSetDevice(ids[r]) != hipSuccess) {
                        (void)hipGetLastError();
                        rcs[r] = CAPY_ERR_HIP;
                        errs[r] = "hipSetDevice(" + std::to_string(ids[r]) + ") failed";
                        return;
                    }
                }
                OptScope sc(opts);
                rcs[r] = body(b[r], b[r + 1] - b[r]);
                if (rcs[r]) errs[r] = g_err;
            }));
        }
    }
    for (auto &l : latches) l->wait();
    for (size_t r = 0; r < world; r++)
        if (rcs[r]) return fail(rcs[r], "device " + std::to_string(ids[r]) + ": " + errs[r]);
    return CAPY_OK;
}

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
static thread_local ScrubRanges t_last_scrub{};  // what the calling thread's last protocol call scrubbed (test hook)
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
    t_last_scrub = r;
    const unsigned gx = (unsigned)std::min<uint64_t>((most / 16 + 255) / 256 + 1, 4096);
    hipLaunchKernelGGL(scrub_kernel, dim3(gx, (unsigned)k), dim3(256), 0, stream, r);
    if (hipGetLastError() != hipSuccess)  // never leave secrets behind because a launch failed: fall back to memsets
        for (int i = 0; i < count; i++) workspace_scrub(stream, slots[i], bytes[i]);
}

#define CAPY_WS(var, type, stream, slot, bytes)                                      \
    type var = reinterpret_cast<type>(capy::workspace(stream, slot, bytes));          \
    if (!var) return capy::fail(CAPY_ERR_HIP, "workspace allocation failed")

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

struct Framing {
    int rw;             // absorb words per block
    uint32_t stride;    // reference `r`
    uint32_t sq_words;  // squeeze words per block
};

static Framing sha3_framing(int d)
{ // Capacity::from_bit_length(d) = 2d, src/sha3/constants.rs:38-45 ; Rate::from(&d), shake_functions.rs:31
    uint32_t r = (1600 - 2 * d) / 8;
    return {(int)(r / 8), r, (uint32_t)((1600 - d) / 64)};
}
static Framing cshake_framing(int d)
{ // capacity = d, shake_functions.rs:63 ; bytes_to_state takes (r*8)/64 words per block, sponge.rs:52
    uint32_t r = (1600 - d) / 8;
    return {(int)(r / 8), r, (uint32_t)((1600 - d) / 64)};
}

// Fold the batch-shared cSHAKE prefix bytepad(encode_string(N) || encode_string(S), w) into p:
// either as init_state (whole blocks) or as raw prefix bytes `pre_host` (D224: r = 172, 168 consumed).
static void cshake_prefix(int d, const uint8_t *fn, size_t fn_len, const uint8_t *cs, size_t cs_len,
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
static void kmac_head(int d, size_t key_len, SpongeParams &p)
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
static std::atomic<bool> g_fused_enabled{true};
// 16 items per wave x one wave per SIMD with the plain round; beyond that the blocked round at raised priority pairs the
// waves of a SIMD (r03, profiles/r03_chipfull.txt: 32 768 x 5 MiB 353 -> 451 GiB/s, 49 152 x 4 MiB 405 -> 472, 98 304 x
// 1 MiB 432 -> 515 against the two-pass form; at 131 072 x 1 MiB the two passes win again, 541 vs 527).
// CAPY_DEBUG=fused_max=N overrides for A/B.
static const size_t FUSED_ONE_WAVE_ITEMS = 16384;
static const size_t FUSED_MAX_ITEMS = [] {
    const double v = debug_knob("fused_max", 98304);
    return (size_t)(v > 0 ? v : 98304);
}();
// Kernel choice by batch size relative to the device's SIMD count S (1024 on MI355X; measured crossovers, profiles/):
//   n <= 32 S        two lanes per sponge, at most one wave per SIMD
//   32 S < n < 64 S  rotating one-lane / two-lane schedule when eligible (sponge_mixed.h), else one lane
//   n <= 128 S       one lane per sponge, latency-tuned instance; uniform batches above 64 S are launched as a
//                    head of 64 S + a remainder that follows the rules above (wave quantisation)
//   above            one lane per sponge, issue-tuned instance (> 2 waves per SIMD); ragged batches stay on the
//                    latency-tuned instance

static std::atomic<bool> g_mixed_enabled{true};
// largest batch that takes the one-wave-per-item encrypt kernel: one wave per SIMD (the digest kernel holds two items per
// wave, so twice as many).  Measured r03 with the DPP theta (profiles/r03_wide_round_probe.txt): at one wave per SIMD the
// wave-per-item kernels still win 1.2-1.3x (1024 x 5 MiB encrypt 0.163 s vs 0.211, 2048 x 5 MiB digest 0.164 vs 0.198),
// at 1.5 waves per SIMD they lose (0.88x).  CAPY_DEBUG=wide_max=N overrides
static size_t wide_max_items();

// SIMDs of the current device (4 per CU)
static unsigned device_simds()
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

static size_t wide_max_items()
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
    CAPY_WS(state, uint64_t *, s, WS_STATE, 25 * n_pad * sizeof(uint64_t));
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
    return 1;
}

// which kernel launch_sponge() picks: 1 one-lane latency-tuned, 2 two-lane, 3 rotating schedule, 4 one-lane issue-tuned,
// 5 wave-quantisation split, 6 wave-per-item digest, 7 uniform-framing kernel
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
// digest kernel (sponge_wide.h) -- shared with sponge_plan(), so that capy_sha3_launch_plan reports the kernel that
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
    // any message length: measured r03 (profiles/r03_small_calls.txt), KMACXOF256 of 64 B / 1 KiB / 16 KiB messages at
    // n <= 2048: 0.029 -> 0.015, 0.066 -> 0.038, 0.645 -> 0.394 ms against the two-lane kernel (r02 took this kernel for
    // messages of at least 64 KiB only)
    return shape_ok && (((dbg & 32) && p.n <= 4096) ||
                        (forced == 0 && !(dbg & 16) && p.n <= 2 * wide_max_items()));
}

static int sponge_plan(int rw, const SpongeParams &p, int *phases)
{
    const int forced = g_lanes_per_sponge.load();
    const size_t simds = device_simds();
    *phases = 1;
    MixedPlan m;
    const unsigned dbg = g_debug_flags.load();
    if (wide_digest_ok(rw, p, forced, dbg)) return 6;
    if (uniform_kernel_ok(rw, p, forced, dbg, simds)) return 7;
    if ((forced == 3 || (forced == 0 && g_mixed_enabled.load())) && mixed_plan(rw, p, forced == 3, m)) {
        *phases = (int)m.P;
        return 3;
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

static int device_order(const uint64_t *offsets, const uint64_t *lens, size_t n, hipStream_t s, const uint32_t **out);
static bool wants_device_order(const uint64_t *offsets, const uint32_t *order, uint64_t n)
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
    // Chip-full launches with wave-uniform framing (equal key, message and output lengths, 8-byte aligned): every framing
    // decision is scalar code in sponge_uniform.h.  Debug bit 7: never (A/B and tests).
    if (uniform_kernel_ok(rw, p2, forced, q.debug_flags, simds)) {
        e = launch_sponge_uniform(rw, p2, uniform_waves(), s);
        if (e == hipErrorInvalidValue) return fail(CAPY_ERR_ARG, "internal: no kernel instance for this rate");
        CAPY_HIP(e);
        return CAPY_OK;
    }
    // Very small digest batches of long messages: one sponge per 25 lanes (sponge_wide.h), 1.3x the two-lane kernel per
    // permutation while every wave has most of a SIMD pair's LDS bandwidth to itself (n / 2 waves <= SIMDs / 2).
    // Debug bit 4 / 5: never / always.
    if (wide_digest_ok(rw, p2, forced, q.debug_flags))
        e = launch_sponge_wide_digest(rw, p2, s);
    else if (forced == 2 || ((forced == 0 || forced == 3) && p.n <= 32 * simds))
        e = launch_sponge_k2(rw, (int)p.out_mode, p2, s);
    // ragged batches stay on the latency-tuned instance at every size: its ragged path keeps the source pointers in
    // registers and prefetches a block ahead, which the 128-VGPR issue-tuned instance cannot afford (2^18 ragged
    // messages of 0..64 KiB: 16.0 vs 13.6 ms; equal lengths given through offsets: 9.7 vs 8.1 ms)
    else if (p.n > 128 * simds && !(q.debug_flags & 2) && !p2.offsets && !p2.order)  // debug bit 1: A/B switch
        e = launch_sponge_k1_full(rw, (int)p.out_mode, p2, s);
    // more than one wave on some SIMD: the paired form of the latency-tuned instance (debug bit 8: A/B switch)
    else if (p.n > 64 * simds && !(q.debug_flags & 256))
        e = launch_sponge_k1_lat_paired(rw, (int)p.out_mode, p2, s);
    else
        e = launch_sponge_k1_lat(rw, (int)p.out_mode, p2, s);
    if (e == hipErrorInvalidValue) return fail(CAPY_ERR_ARG, "internal: no kernel instance for this rate / mode");
    CAPY_HIP(e);
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
static int sha3_launch(int d, size_t n, const MsgView &m, uint8_t *digests, uint64_t out_stride, hipStream_t s)
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
static int cshake_launch(int d, size_t n, const MsgView &m, size_t l_bits, const uint8_t *fn, size_t fn_len,
                         const uint8_t *cs, size_t cs_len, uint8_t *outs, uint64_t out_stride, hipStream_t s,
                         bool body_has_trailer = false)
{
    if (l_bits / 8 > CAPY_MAX_OUT_LEN) return fail(CAPY_ERR_ARG, "output longer than 2^32 - 1 bytes per item (l_bits >= 2^35)");
    if (!valid_d(d)) return fail(CAPY_ERR_UNSUPPORTED_SECPARAM, "unsupported security parameter");
    if (fn_len == 0 && cs_len == 0 && !body_has_trailer)
        return fail(CAPY_ERR_UNSUPPORTED,
                    "cshake with empty N and S (shake_functions.rs:59-61) is served by the host-buffer entry point only");
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

// ---- longest-first processing order for ragged DEVICE batches (host batches are sorted on upload): a stable sort
// by a 512-step logarithmic length scale WITHIN chunks of ORDER_CHUNK consecutive items.  Sorting the whole batch
// scatters the messages of a wave all over the buffer, which costs short messages more than the idle lanes it saves
// (2^21 messages of 0..2 KiB: 11.0 ms globally sorted, 4.4 ms unsorted); inside a 4096-item neighbourhood the lanes of
// a wave still get near-equal lengths and their messages stay within a few MB.
constexpr int ORDER_BUCKETS = 512;
constexpr int ORDER_CHUNK_SHIFT = 12;
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
// Dispatch order of the 64-item groups: rank-major over the full neighbourhoods -- first every neighbourhood's longest
// group, then every second-longest, ... -- so the grid as a whole starts its long groups first (longest-processing-time
// first, what keeps the tail short when the groups do not all fit on the chip at once) while each group still reads
// from one neighbourhood.  Groups of equal rank sit next to each other, so the workgroups that end up on one SIMD
// (indices a multiple of the slot count apart) differ in rank as well.
__host__ __device__ __forceinline__ uint32_t order_spread(uint32_t pos, uint64_t n)
{
    const uint32_t full = (uint32_t)(n >> ORDER_CHUNK_SHIFT);  // complete neighbourhoods
    const uint32_t c = pos >> ORDER_CHUNK_SHIFT;
    if (c >= full) return pos;  // partial last neighbourhood: plain sorted order, at the end
    const uint32_t r = (pos >> 6) & 63u;
    return ((r * full + c) << 6) + (pos & 63u);
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
static int device_order(const uint64_t *offsets, const uint64_t *lens, size_t n, hipStream_t s, const uint32_t **out)
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
        return kmac_launch(d, n, fixed_keys(keka, key_len, keka_stride), m, false, (const uint8_t *)ke_custom,
                           strlen(ke_custom), 1, nullptr, 0, 0, mask, s);
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
    const bool fused_ok = g_fused_enabled.load() && ff.stride == (uint32_t)ff.rw * 8 && n <= FUSED_MAX_ITEMS &&
                          m.aligned8 && m.msgs != nullptr && tag_len <= 64 && (tag_len & 3) == 0;
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
        fp.staged = (g_debug_flags.load() & 64) ? 1 : 0;  // A/B switch (debug bit 6)
        fp.paired = (n > FUSED_ONE_WAVE_ITEMS && !fp.staged) ? 1 : 0;
        fp.n = n;
        // One wave per item (sponge_wide.h) while every wave still has most of a SIMD pair's LDS bandwidth to itself:
        // 1.3x per permutation at n = 128, break-even near one wave per SIMD (profiles/r02_wide_lane_probe.txt).
        // Worth it only when the serial chains are long; debug bits 4 / 5: never / always (A/B and tests).
        {
            const unsigned dbg = g_debug_flags.load();
            // any message length (r03: 1.3-2.0x the four-lane kernel from 64 B to 5 MiB at n <= 1024, profiles/r03_small_calls.txt)
            fp.wide = ((dbg & 32) && n <= 4096) || (!(dbg & 16) && n <= wide_max_items()) ? 1 : 0;
        }
        fp.tags = encrypt ? tags : tag2;
        CAPY_HIP(launch_sponge_fused(ff.rw, fp, s));
        if (encrypt) return CAPY_OK;
        tag_compare_launch(tags, tag_len, tag2, tag_len, (uint32_t)tag_len, status, n, s);
        return keystream(status);
    }
    int rc;
    if (encrypt) {
        rc = tag(tags);
        if (rc == CAPY_OK) rc = keystream(nullptr);
        return rc;
    }
    rc = keystream(nullptr);
    if (rc == CAPY_OK) rc = tag(tag2);
    if (rc) return rc;
    tag_compare_launch(tags, tag_len, tag2, tag_len, (uint32_t)tag_len, status, n, s);
    return keystream(status);
}

// sha3_encrypt / sha3_decrypt on device buffers (src/sha3/encryptable.rs:29-83)
// (and the sponge half of KEMEncryptable, src/kem/encryptable.rs:47-59,84-104: same flow, tags "KEMKE"/"KEMKA")
// pw: n passwords, fixed length or per item (KeyView); pws_bytes = total password bytes (sizes the scratch of the
// per-item form without reading device memory)
static int sha3_crypt_dev(bool encrypt, int d, size_t n, const KeyView &pw, uint64_t pws_bytes, const uint8_t *zs,
                          const MsgView &m, uint8_t *tags, int32_t *status, hipStream_t s, const char *ke_custom = "SKE",
                          const char *ka_custom = "SKA")
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
static hipError_t copy_rows_out(uint8_t *dst, size_t row, const DevBuf &b, size_t stride, size_t n)
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
    has_order = !uniform && n >= 128 && n <= 0xffffffffULL && !(g_debug_flags.load() & 4);  // debug bit 2: A/B switch
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

const char *capy_last_error(void) { return capy::g_err.c_str(); }
const char *capy_version(void) { return "capyhip 0.3 (gfx950)"; }

int capy_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int capy_set_device(int device)
{
    CAPY_HIP(hipSetDevice(device));
    return CAPY_OK;
}

int capy_set_devices(const int *ids, int n)
{
    if (n < 0 || (n > 0 && !ids)) return fail(CAPY_ERR_ARG, "bad device list");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) have = 0;
    for (int i = 0; i < n; i++)
        if (ids[i] < 0 || ids[i] >= have) return fail(CAPY_ERR_ARG, "device id out of range");
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        g_dev_ids.assign(ids, ids + n);
    }
    // the workers of the previous list end now (their scratch is returned on their own threads); the new ones start
    // with the first sharded call
    std::lock_guard<std::mutex> pool_lock(g_pool.mu);
    if (g_pool.ids != std::vector<int>(ids, ids + n)) g_pool.reset({});
    return CAPY_OK;
}

int capy_shard_plan(size_t n, int n_devices, const uint64_t *byte_offsets, uint64_t *bounds)
{
    if (n_devices < 1 || !bounds) return fail(CAPY_ERR_ARG, "bad shard plan request");
    const std::vector<size_t> b = shard_bounds(n, (size_t)n_devices, byte_offsets);
    for (int r = 0; r <= n_devices; r++) bounds[r] = b[r];
    return CAPY_OK;
}

int capy_get_devices(int *ids, int capacity)
{
    std::lock_guard<std::mutex> lk(g_dev_mu);
    const int n = (int)g_dev_ids.size();
    for (int i = 0; i < n && i < capacity; i++) ids[i] = g_dev_ids[i];
    return n;
}

int capy_device_synchronize(void)
{
    std::vector<int> ids;
    if (configured_devices(ids)) {
        int cur = 0;
        CAPY_HIP(hipGetDevice(&cur));
        for (int id : ids) {
            CAPY_HIP(hipSetDevice(id));
            CAPY_HIP(hipDeviceSynchronize());
        }
        CAPY_HIP(hipSetDevice(cur));
        return CAPY_OK;
    }
    CAPY_HIP(hipDeviceSynchronize());
    return CAPY_OK;
}

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
    for (int k = 0; k < 4; k++) {
        if (!t_last_scrub.ptr[k] || !t_last_scrub.bytes[k]) continue;
        std::vector<uint8_t> h(t_last_scrub.bytes[k]);
        CAPY_HIP(hipMemcpy(h.data(), t_last_scrub.ptr[k], h.size(), hipMemcpyDeviceToHost));
        for (uint8_t b : h) total += b != 0;
    }
    *nonzero_bytes = total;
    return CAPY_OK;
}

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
