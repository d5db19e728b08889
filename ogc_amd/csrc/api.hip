// api.hip — library-level entry points: version and thread-local error text.
#include <stdarg.h>
#include <stdio.h>

#include <map>
#include <mutex>
#include <utility>

#include "ogc_common.h"

namespace {
thread_local char g_err[512] = "";
}

void ogc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Stream-ordered scratch memory that persists between calls: one grow-only buffer per (device, stream).  Kernels of
// successive calls on a stream run in order, so they can share it; hipMallocAsync / hipFreeAsync per call put two
// extra operations on the queue for a buffer whose size hardly ever changes.
void *ogc_workspace(hipStream_t stream, size_t bytes) {
    struct Slot { void *ptr; size_t size; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Slot> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    Slot &sl = slots[{dev, stream}];
    if (sl.size >= bytes && sl.ptr) return sl.ptr;
    if (sl.ptr) (void)hipFreeAsync(sl.ptr, stream); // after the work already queued on this stream
    sl.ptr = nullptr;
    sl.size = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    void *p = nullptr;
    if (hipMallocAsync(&p, want, stream) != hipSuccess || !p) {
        (void)hipGetLastError();
        return nullptr;
    }
    sl.ptr = p;
    sl.size = want;
    return p;
}

// ---- one zero fill per step ----------------------------------------------------------------------------------------------
// A training step runs ~40 operators that accumulate with atomics into small caller-owned buffers (GroupNorm statistics, weight
// gradients, moment matrices, counters) and each zeroes its buffer first: ~40 launches of ~4 us, every one a dependent
// kernel boundary.  The caller may instead carve those buffers out of ONE region: ogc_zero_arena_begin zeroes it with a
// single launch on `stream` and remembers its extent; until ogc_zero_arena_end, an operator asked to zero memory INSIDE that
// region on the same stream skips its fill.  The caller's side of the contract: every byte of the region is handed to at
// most one operator between begin and end (ogc_amd/utils/zero_arena.py: a bump allocator), and only operators launched on
// `stream` get pieces of it.  One region per process at a time (the training step's).
namespace {
std::mutex g_arena_mu;
uintptr_t g_arena_lo = 0, g_arena_hi = 0;
hipStream_t g_arena_stream = nullptr;
} // namespace

bool ogc_zero_arena_covers(const void *ptr, size_t bytes, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_arena_mu);
    const uintptr_t a = (uintptr_t)ptr;
    return g_arena_hi != 0 && stream == g_arena_stream && a >= g_arena_lo && a + bytes <= g_arena_hi;
}

extern "C" int ogc_zero_arena_begin(void *base, int bytes, ogc_stream_t stream) {
    OGC_REQUIRE(bytes >= 0 && (bytes & 3) == 0 && (((uintptr_t)base) & 3) == 0, "ogc_zero_arena_begin: region must be 4-byte aligned");
    {
        std::lock_guard<std::mutex> lock(g_arena_mu);
        g_arena_lo = g_arena_hi = 0; // (the fill below must not skip itself)
    }
    if (bytes == 0) return OGC_OK;
    OGC_REQUIRE(base != nullptr, "ogc_zero_arena_begin: null pointer");
    if (ogc_zero_async(base, (size_t)bytes, (hipStream_t)stream) != hipSuccess) {
        ogc_set_error("ogc_zero_arena_begin: fill failed");
        return OGC_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_arena_mu);
    g_arena_lo = (uintptr_t)base;
    g_arena_hi = g_arena_lo + (size_t)bytes;
    g_arena_stream = (hipStream_t)stream;
    return OGC_OK;
}

extern "C" int ogc_zero_arena_end(void) {
    std::lock_guard<std::mutex> lock(g_arena_mu);
    g_arena_lo = g_arena_hi = 0;
    g_arena_stream = nullptr;
    return OGC_OK;
}

extern "C" int ogc_version(void) { return OGC_VERSION; }

extern "C" const char *ogc_last_error(void) { return g_err; }
