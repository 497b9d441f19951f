// shard.hip — error plumbing, debug knobs, per-call options, the device list of capy_set_devices and the persistent
// per-device workers that run the shards of a host-buffer call (contiguous shards, no collective).
#include <string.h>
#include <algorithm>
#include <atomic>
#include <string>
#include <vector>
#include "common.h"
#include <ctype.h>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <sched.h>

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
                                      "host_overlap", "host_arena", "worker_affinity", "rot", "rot_ratio", "ed448_quad_min", "ed448_quad_max",
                                      "ed448_duo_min", "ed448_duo_max", "ed448_quad_ct_max", "ed448_peel", "fused_slices", "uniform_slices",
                                      "fused1_min", "fused1_form", "fused1_waves", "fused1_direct", "fused1_slices", "fused1_rot", "fused1_ratio", "fused1_turns", "fused1_store", "fused1_lone_direct", "fused1_level", "ed448_duo_ct"};
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
// capy_set_min_items_per_device: a sharded call uses only as many of the listed devices as leave each at least this many
// items (1 = every device that gets an item).  Latency-bound batches gain nothing from a finer cut: INTEGRATION.md section 5.
static std::atomic<size_t> g_min_items_per_device{1};
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

// the cut of a sharded call: the minimum-shard rule first (only the first `used` devices of the list take part; the others
// get empty shards), then the byte- or count-balanced cut over those
static std::vector<size_t> shard_bounds_ruled(size_t n, size_t world, const uint64_t *off)
{
    const size_t min_items = g_min_items_per_device.load();
    if (min_items <= 1) return shard_bounds(n, world, off);  // the default: the plain cut over the whole list
    const size_t used = std::max<size_t>(1, std::min(world, n / min_items));
    std::vector<size_t> b = shard_bounds(n, used, off);
    b.resize(world + 1, n);
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
// The CPUs a worker pins itself to: the device's local_cpulist intersected with the CPUs this process may use (never leave
// the container's set); false = leave the affinity alone (unparsable list, or an empty intersection)
static bool affinity_intersection(const char *local_cpulist, const cpu_set_t &have, cpu_set_t *both)
{
    cpu_set_t want;
    if (!parse_cpulist(local_cpulist, &want)) return false;
    CPU_AND(both, &want, &have);
    return CPU_COUNT(both) > 0;
}
// one line of /sys/bus/pci/devices/<bdf>/<leaf> of a device; false when the device has no such file (a container may hide it)
static bool device_sysfs_line(int device, const char *leaf, char *bdf, size_t bdf_cap, char *line, size_t line_cap)
{
    if (hipDeviceGetPCIBusId(bdf, (int)bdf_cap, device) != hipSuccess) {
        (void)hipGetLastError();
        bdf[0] = 0;
        return false;
    }
    for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/" + leaf;
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    const bool ok = fgets(line, (int)line_cap, f) != nullptr;
    fclose(f);
    return ok;
}
// the CPUs a worker of `device` runs on: local_cpulist of the device intersected with the calling thread's affinity
static bool device_cpus(int device, char *bdf, size_t bdf_cap, cpu_set_t *both)
{
    char line[4096] = {0};
    cpu_set_t have;
    if (!device_sysfs_line(device, "local_cpulist", bdf, bdf_cap, line, sizeof line)) return false;
    if (sched_getaffinity(0, sizeof have, &have) != 0) return false;
    return affinity_intersection(line, have, both);
}
static void pin_to_device_cpus(int device)
{
    if (debug_knob("worker_affinity", 1) == 0) return;
    char bdf[64] = {0};
    cpu_set_t both;
    if (device_cpus(device, bdf, sizeof bdf, &both)) (void)sched_setaffinity(0, sizeof both, &both);
}
}  // namespace
// capy_device_topology: where a device sits (what the first real multi-GPU record should say about itself)
int device_topology(int device, char *pci_bus_id, size_t cap, int *numa_node, int *cpus, int capacity)
{
    char bdf[64] = {0}, line[256] = {0};
    cpu_set_t both;
    CPU_ZERO(&both);
    const bool pinned = device_cpus(device, bdf, sizeof bdf, &both);
    if (pci_bus_id && cap) snprintf(pci_bus_id, cap, "%s", bdf);
    if (numa_node) {
        char bdf2[64];
        *numa_node = device_sysfs_line(device, "numa_node", bdf2, sizeof bdf2, line, sizeof line) ? atoi(line) : -1;
    }
    if (!pinned) return 0;
    int k = 0;
    for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, &both)) {
            if (k < capacity && cpus) cpus[k] = c;
            k++;
        }
    return k;
}
// test hook (capy_debug_affinity_plan): the same arithmetic on a caller-supplied sysfs string and allowed-CPU list
int affinity_plan(const char *local_cpulist, const int *allowed, int n_allowed, int *out, int capacity)
{
    cpu_set_t have, both;
    CPU_ZERO(&have);
    for (int i = 0; i < n_allowed; i++)
        if (allowed[i] >= 0 && allowed[i] < CPU_SETSIZE) CPU_SET(allowed[i], &have);
    if (!affinity_intersection(local_cpulist, have, &both)) return 0;
    int k = 0;
    for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, &both)) {
            if (k < capacity) out[k] = c;
            k++;
        }
    return k;
}
namespace {

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
    const std::vector<size_t> b = shard_bounds_ruled(n, world, byte_offsets);
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

}  // namespace capy

using namespace capy;

extern "C" {

const char *capy_last_error(void) { return capy::g_err.c_str(); }
const char *capy_version(void) { return "capyhip 0.6 (gfx950)"; }
int capy_abi_version(void) { return CAPY_ABI_VERSION; }

int capy_set_min_items_per_device(size_t n)
{
    capy::g_min_items_per_device.store(n ? n : 1);
    return CAPY_OK;
}

int capy_debug_affinity_plan(const char *local_cpulist, const int *allowed_cpus, int n_allowed, int *out_cpus, int capacity)
{
    if (!local_cpulist || (!allowed_cpus && n_allowed) || (!out_cpus && capacity) || n_allowed < 0 || capacity < 0)
        return capy::fail(CAPY_ERR_ARG, "null or invalid argument");
    return capy::affinity_plan(local_cpulist, allowed_cpus, n_allowed, out_cpus, capacity);
}

int capy_device_topology(int device, char *pci_bus_id, size_t pci_capacity, int *numa_node, int *cpus, int capacity)
{
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || device < 0 || device >= have || capacity < 0 || (capacity && !cpus)) {
        (void)hipGetLastError();
        return capy::fail(CAPY_ERR_ARG, "bad device or buffer");
    }
    return capy::device_topology(device, pci_bus_id, pci_capacity, numa_node, cpus, capacity);
}

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
    const std::vector<size_t> b = shard_bounds_ruled(n, (size_t)n_devices, byte_offsets);
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

}  // extern "C"
