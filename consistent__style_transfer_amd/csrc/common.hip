// Error reporting of libcst_hip.so: every entry point returns an int status and never throws;
// the message of the last failure on this thread is available through cst_last_error().
#include "cst_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void cst_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cst_last_error() { return g_err; }

extern "C" int cst_abi_version() { return 1; }
