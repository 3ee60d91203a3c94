// C-ABI plumbing of libagl.so: error string, version.  Kernels live in the *.hip units.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void agl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {
const char* agl_last_error(void) { return g_err; }
int agl_version(void) { return 4; }   // = AGL_ABI_VERSION of include/agl.h (bumped with every signature change)
}
