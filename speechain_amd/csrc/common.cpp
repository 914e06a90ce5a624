// Error bookkeeping and version of the speechain_amd C-ABI library.
#include "sc_common.h"

static thread_local char g_err[512] = "";

int sc_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

SC_API const char* sc_last_error(void) { return g_err; }
SC_API int sc_version(void) { return 1; }
