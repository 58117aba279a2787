// ABI version + thread-local error string for libcsam_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" {
void csam_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int csam_abi_version(void) { return 1; }
const char* csam_last_error(void) { return g_err; }
}
