// ABI bookkeeping: version, build tag, thread-local error string.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/lnerf_hip.h"

namespace lnerf {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace lnerf

#ifndef LNERF_BUILD_TAG
#define LNERF_BUILD_TAG "dev"
#endif

extern "C" {
int lnerf_abi_version(void) { return LNERF_ABI_VERSION; }
const char *lnerf_last_error(void) { return lnerf::g_err; }
#ifdef LNERF_EXPERIMENTS   // (an experiment build also carries the measured-and-rejected kernel variants)
const char *lnerf_build_info(void) { return "gfx950;" LNERF_BUILD_TAG ";experiments"; }
#else
const char *lnerf_build_info(void) { return "gfx950;" LNERF_BUILD_TAG; }
#endif
}
