// ABI version + thread-local error text for libgeogcn.so.
#include "common.h"

#include <stdarg.h>

namespace geogcn {
namespace {
thread_local char g_err[512] = "";
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace geogcn

extern "C" {
int geogcn_version(void) { return GEOGCN_ABI_VERSION; }
const char* geogcn_last_error(void) { return geogcn::g_err; }
}
