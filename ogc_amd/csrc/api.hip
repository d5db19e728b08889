// api.hip — library-level entry points: version and thread-local error text.
#include <stdarg.h>
#include <stdio.h>

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

extern "C" int ogc_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char *ogc_last_error(void) { return g_err; }
