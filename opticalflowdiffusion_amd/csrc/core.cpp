// Error channel and version of libofd_hip.
#include <cstdarg>
#include <cstdio>
#include "../../include/ofd.h"

namespace ofd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace ofd

extern "C" int ofd_version(void) { return (0 << 16) | 1; }
extern "C" const char* ofd_last_error(void) { return ofd::g_err; }
