// common.h — host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

// RAII device allocation for the host-pointer entry points.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n)
    {
        bytes = n ? n : 8;
        return hipMalloc(&p, bytes);
    }
    template <class T>
    T *as() const
    {
        return reinterpret_cast<T *>(p);
    }
};

// Grow-only device scratch, one set of named slots per (thread, device, stream).  All users of a slot
// enqueue on that stream, so reuse across calls is ordered by the stream itself; growing a slot
// frees the old block with hipFree, which synchronises the device first.  (hipMallocAsync /
// hipFreeAsync proved unreliable on the default stream of this ROCm build: results raced.)
enum WsSlot { WS_PRE = 0, WS_ZPW, WS_KEKA, WS_TAG2, WS_A, WS_B, WS_C, WS_D, WS_E, WS_F, WS_TABLE, WS_STATE, WS_ORDER, WS_NSLOTS };
void *workspace(hipStream_t stream, WsSlot slot, size_t bytes);  // nullptr on allocation failure

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

inline bool valid_d(int d) { return d == 224 || d == 256 || d == 384 || d == 512; }

}  // namespace capy
