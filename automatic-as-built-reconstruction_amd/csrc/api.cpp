// api.cpp -- error reporting and version of libaabr_hip.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/aabr_hip.h"

namespace aabr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
} // namespace aabr

extern "C" const char *aabr_last_error(void) { return aabr::g_err; }
extern "C" int aabr_version(void) { return 100; }
