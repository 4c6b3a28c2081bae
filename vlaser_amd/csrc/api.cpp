// Error plumbing + ABI version of libvlaser_hip.so (no C++ exceptions cross the ABI).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/vlaser_hip.h"

static thread_local char g_err[512] = "";

void vlaser_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vlaser_last_error(void) { return g_err; }
extern "C" int vlaser_abi_version(void) { return 6; }
