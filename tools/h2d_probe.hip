// h2d_probe.hip — how fast can a caller's pageable buffer reach HBM?  (decides the staging policy of PackedBatch)
//   hipcc --offload-arch=gfx950 -O2 -o tools/h2d_probe tools/h2d_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t N = (size_t)1280 << 20;
    uint8_t *h = (uint8_t *)malloc(N);
    for (size_t i = 0; i < N; i += 4096) h[i] = (uint8_t)i;  // touch
    memset(h, 1, N);
    uint8_t *d;
    CK(hipMalloc(&d, N));
    for (int rep = 0; rep < 2; rep++) {
        double t = now();
        CK(hipMemcpy(d, h, N, hipMemcpyHostToDevice));
        printf("pageable hipMemcpy H2D       : %6.2f GiB/s\n", N / (now() - t) / (1 << 30));
    }
    for (int rep = 0; rep < 2; rep++) {
        double t = now();
        CK(hipMemcpy(h, d, N, hipMemcpyDeviceToHost));
        printf("pageable hipMemcpy D2H       : %6.2f GiB/s\n", N / (now() - t) / (1 << 30));
    }
    for (int rep = 0; rep < 2; rep++) {
        double t = now();
        CK(hipHostRegister(h, N, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpy(d, h, N, hipMemcpyHostToDevice));
        double t2 = now();
        CK(hipHostUnregister(h));
        double t3 = now();
        printf("register %.3fs + copy %.3fs + unregister %.3fs : %6.2f GiB/s overall, copy alone %6.2f GiB/s\n", t1 - t, t2 - t1,
               t3 - t2, N / (t3 - t) / (1 << 30), N / (t2 - t1) / (1 << 30));
    }
    // pinned ring: T threads memcpy into pinned chunks, async H2D behind them
    for (int T : {1, 2, 4, 8}) {
        const size_t CH = (size_t)16 << 20;
        const int NB = 2 * T;
        std::vector<uint8_t *> pin(NB);
        std::vector<hipEvent_t> ev(NB);
        for (int i = 0; i < NB; i++) {
            CK(hipHostMalloc((void **)&pin[i], CH, hipHostMallocDefault));
            CK(hipEventCreate(&ev[i]));
        }
        hipStream_t st;
        CK(hipStreamCreate(&st));
        double t = now();
        const size_t nch = N / CH;
        // each thread owns chunks c = tid, tid+T, ... and two pinned buffers
        std::vector<std::thread> th;
        std::vector<hipStream_t> sts(T);
        for (int k = 0; k < T; k++) CK(hipStreamCreate(&sts[k]));
        for (int k = 0; k < T; k++)
            th.emplace_back([&, k] {
                int b = 0;
                for (size_t c = k; c < nch; c += T, b ^= 1) {
                    const int slot = 2 * k + b;
                    (void)hipEventSynchronize(ev[slot]);
                    memcpy(pin[slot], h + c * CH, CH);
                    (void)hipMemcpyAsync(d + c * CH, pin[slot], CH, hipMemcpyHostToDevice, sts[k]);
                    (void)hipEventRecord(ev[slot], sts[k]);
                }
                (void)hipStreamSynchronize(sts[k]);
            });
        for (auto &x : th) x.join();
        printf("pinned ring, %d copy threads   : %6.2f GiB/s\n", T, N / (now() - t) / (1 << 30));
        for (int i = 0; i < NB; i++) (void)hipHostFree(pin[i]);
    }
    return 0;
}
