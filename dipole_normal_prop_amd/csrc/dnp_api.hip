// dnp_api.hip - housekeeping entry points and the thread-local error string of libdnp.so.
#include <string.h>

#include "dnp_common.h"

namespace dnp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void clear_error() { g_err[0] = '\0'; }

}  // namespace dnp

extern "C" {

int dnp_version(void) { return DNP_VERSION; }

int dnp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();  // "no device" is an answer here, not an error
        return 0;
    }
    return n;
}

const char* dnp_last_error(void) { return dnp::g_err; }

}  // extern "C"
