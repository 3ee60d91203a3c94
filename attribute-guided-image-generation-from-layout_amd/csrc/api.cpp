// C-ABI plumbing of libagl.so: error string, version.  Kernels live in the *.hip units.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/agl.h"      // AGL_ABI_VERSION, and the declarations this unit defines

static thread_local char g_err[512] = "";

void agl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {
const char* agl_last_error(void) { return g_err; }
int agl_version(void) { return AGL_ABI_VERSION; }   // include/agl.h (bumped with every signature change)
}
