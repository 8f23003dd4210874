// C-ABI plumbing shared by every entry point of libss2d_hip.so (include/ss2d.h).
#include <cstdarg>
#include <cstdio>
#include "ss2d_common.h"
#include "../../include/ss2d.h"

static thread_local char g_err[512] = "";

void ss2d_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ss2d_abi_version(void) { return SS2D_ABI_VERSION; }
extern "C" const char *ss2d_last_error(void) { return g_err; }
