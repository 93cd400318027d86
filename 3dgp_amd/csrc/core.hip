// core.hip -- version / error plumbing of the C ABI.
#include "common.h"

static thread_local char g_err[512] = "";

void tdgp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

TDGP_API int tdgp_version(void) { return 100; }
TDGP_API const char* tdgp_last_error(void) { return g_err; }
