// ABI bookkeeping: version + thread-local last-error text.
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void hamt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int hamt_version(void) { return HAMT_ABI_VERSION; }

extern "C" int hamt_last_error(char* buf, size_t n) {
  if (!buf || n == 0) return (int)strlen(g_err);
  strncpy(buf, g_err, n - 1);
  buf[n - 1] = 0;
  return (int)strlen(buf);
}
