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

extern "C" int ogc_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char *ogc_last_error(void) { return g_err; }
