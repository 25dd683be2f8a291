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

// ---- optional per-kernel timing (measurement aid for bench.py; no reference counterpart) ----
// While enabled, every kernel launch of the library is bracketed by two HIP events
// recorded on the launch stream; p2n_profile_collect() synchronises and returns the
// summed duration and launch count per kernel id.  Not thread-safe; off by default.
#define P2N_PROF_MAX 8192
static bool g_prof_on = false;
static int g_prof_sel = -1;   // >= 0: only launches of this kernel id are bracketed
static int g_prof_n = 0;
static hipEvent_t g_prof_ev[P2N_PROF_MAX][2];
static int g_prof_id[P2N_PROF_MAX];
static int g_prof_created = 0;

int p2n_prof_begin(int kernel_id, hipStream_t s) {
    if (!g_prof_on || g_prof_n >= P2N_PROF_MAX) return -1;
    if (g_prof_sel >= 0 && kernel_id != g_prof_sel) return -1;
    const int i = g_prof_n;
    if (i >= g_prof_created) {
        if (hipEventCreate(&g_prof_ev[i][0]) != hipSuccess || hipEventCreate(&g_prof_ev[i][1]) != hipSuccess) return -1;
        g_prof_created = i + 1;
    }
    g_prof_id[i] = kernel_id;
    (void)hipEventRecord(g_prof_ev[i][0], s);
    return i;
}

void p2n_prof_end(int slot, hipStream_t s) {
    if (slot < 0) return;
    (void)hipEventRecord(g_prof_ev[slot][1], s);
    g_prof_n = slot + 1;
}

extern "C" int p2n_profile_enable(int on) {
    g_prof_on = on != 0;
    g_prof_n = 0;
    return P2N_OK;
}

extern "C" int p2n_profile_select(int kernel_id) {
    g_prof_sel = kernel_id;
    return P2N_OK;
}

extern "C" int p2n_profile_collect(int max_ids, float *ms_sum_host, int *count_host) {
    for (int k = 0; k < max_ids; ++k) { ms_sum_host[k] = 0.f; count_host[k] = 0; }
    for (int i = 0; i < g_prof_n; ++i) {
        P2N_CHECK_HIP(hipEventSynchronize(g_prof_ev[i][1]));
        float ms = 0.f;
        P2N_CHECK_HIP(hipEventElapsedTime(&ms, g_prof_ev[i][0], g_prof_ev[i][1]));
        const int k = g_prof_id[i];
        if (k >= 0 && k < max_ids) { ms_sum_host[k] += ms; count_host[k] += 1; }
    }
    g_prof_n = 0;
    return P2N_OK;
}
