// Error plumbing + version for libtavsr_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace tavsr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tavsr

extern "C" int tavsr_version(void) { return 3; }
extern "C" const char* tavsr_last_error_string(void) { return tavsr::g_err; }
