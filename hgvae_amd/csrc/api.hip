// Library-level entry points: version and the thread-local error string.
#include "common.h"

static thread_local char g_err[512] = "";

void hgv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* hgv_last_error(void) { return g_err; }

// ABI version: bumped whenever an exported signature changes.
extern "C" int hgv_version(void) { return 2; }
