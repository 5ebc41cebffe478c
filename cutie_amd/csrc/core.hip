// Error plumbing and version of the C ABI (include/cutie_hip.h).
#include "common.h"

#include <stdlib.h>

#include <mutex>
#include <string>
#include <vector>

namespace cutie {
static thread_local std::string g_last_error;

// A/B switches of the measurement scripts, read ONCE when the library is loaded (never per call)
static bool env_on(const char* name, bool dflt = true) {
    const char* v = getenv(name);
    return v ? atoi(v) != 0 : dflt;
}
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static const Config g_config = {env_on("CUTIE_CONV_WIDE"), env_on("CUTIE_FUSE_NORM_BWD"), env_on("CUTIE_CONV_LPT"), env_on("CUTIE_RENDEZVOUS"), env_on("CUTIE_CONV_RING", false), env_int("CUTIE_CONV_PATCH", 0), env_int("CUTIE_CONV_DUO", 1)};
const Config& config() { return g_config; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): set when a kernel first needs more
// than it has been granted there, not on every launch.  The table is a cache of what the runtime already holds.
int ensure_dyn_lds(const void* fn, int bytes) {
    struct Entry { const void* fn; int dev, bytes; };
    static std::mutex mu;
    static std::vector<Entry> granted;
    int dev = 0;
    CUTIE_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (Entry& e : granted) {
        if (e.fn == fn && e.dev == dev) {
            if (e.bytes >= bytes) return CUTIE_OK;
            CUTIE_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            e.bytes = bytes;
            return CUTIE_OK;
        }
    }
    CUTIE_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    granted.push_back({fn, dev, bytes});
    return CUTIE_OK;
}

// compute units of the current device (cached per device; 0 if the runtime will not say): the tile meeting inside a conv
// launch needs all workgroups of a sample resident at once - on a partitioned GPU (CPX: 32 CUs) fewer fit than on a whole one
int device_cus() {
    static std::mutex mu;
    static std::vector<int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
    std::lock_guard<std::mutex> lock(mu);
    if ((int)cus.size() <= dev) cus.resize(dev + 1, -1);
    if (cus[dev] < 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus[dev] = n;
    }
    return cus[dev];
}

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
