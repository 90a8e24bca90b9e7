// Library-level entry points: version, arch, thread-local error string.
#include <stdarg.h>

#include "common.h"

namespace trid {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace trid

extern "C" int trid_version(void) { return TRID_VERSION; }
extern "C" const char* trid_arch(void) { return "gfx950"; }
extern "C" const char* trid_last_error_string(void) { return trid::g_err; }
