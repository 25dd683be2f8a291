// Error plumbing and version of libp2n_hip.so.
#include <stdarg.h>
#include "p2n_common.h"

static thread_local char g_err[512] = "";

void p2n_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *p2n_last_error(void) { return g_err; }
extern "C" int p2n_version(void) { return 100; }
