// tfk_api.hip -- error plumbing, version and device query of libtfk.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "tfk_common.h"

namespace tfk {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TFK_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return TFK_OK;
}

int cu_count() {
    static std::atomic<int> cached{0};
    int v = cached.load(std::memory_order_relaxed);
    if (v > 0) return v;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0) {
        v = prop.multiProcessorCount;
        cached.store(v, std::memory_order_relaxed);
        return v;
    }
    (void)hipGetLastError();
    return 256;
}

}  // namespace tfk

extern "C" {

int tfk_abi_version(void) { return TFK_ABI_VERSION; }

const char *tfk_last_error(void) { return tfk::g_err; }

int tfk_device_info(char *name, int32_t name_len, int32_t *compute_units) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return tfk::fail(TFK_ENODEV, "tfk_device_info: no HIP device");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        return tfk::fail(TFK_ENODEV, "tfk_device_info: hipGetDeviceProperties failed");
    }
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return tfk::fail(TFK_ENODEV, "tfk_device_info: device is %s, this library is built for gfx950 only",
                         prop.gcnArchName);
    return TFK_OK;
}

}  // extern "C"
