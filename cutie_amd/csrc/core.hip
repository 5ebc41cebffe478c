// Error plumbing and version of the C ABI (include/cutie_hip.h).
#include "common.h"

#include <string>

namespace cutie {
static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace cutie

extern "C" int cutie_version(void) { return 100; }

extern "C" const char* cutie_last_error(void) { return cutie::g_last_error.c_str(); }
