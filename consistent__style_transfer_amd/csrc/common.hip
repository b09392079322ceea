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

// Data-parallel rank of this process for the dropout index contract (cst_common.h, CstDrop::base).  One process per GPU.
static uint32_t g_drop_shard_rank = 0;
uint32_t cst_drop_shard_rank() { return g_drop_shard_rank; }
extern "C" int cst_set_drop_shard(int rank) {
    CST_REQUIRE(rank >= 0, "cst_set_drop_shard: rank=%d must be >= 0", rank);
    g_drop_shard_rank = (uint32_t)rank;
    return CST_OK;
}
