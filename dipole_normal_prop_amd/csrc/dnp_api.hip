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

#ifdef DNP_BOUNDS
static unsigned int* g_bounds_err = nullptr;
unsigned int* bounds_err_buffer() {
    if (!g_bounds_err) {
        if (hipMalloc((void**)&g_bounds_err, 16 * sizeof(unsigned int)) != hipSuccess) return nullptr;
        (void)hipMemset(g_bounds_err, 0, 16 * sizeof(unsigned int));
    }
    return g_bounds_err;
}
#endif

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

#ifdef DNP_BOUNDS
// check builds only: copies the counters to the host - synchronises the device - and optionally clears them: [0..7] the
// out-of-bounds accesses by table (pair_kernel.h kBnd*: chunk offsets, chunk boxes, tile boxes, target groups, tile partials,
// partial slab, exchange records, source range), [8] the (slab, tile) items whose rows break the two-group precondition of
// w_part (a caller's error, not an access).  Returns the sum of [0..7], or -1 when the buffer could not be read.
long long dnp_debug_bounds_errors(unsigned int* host_out16, int reset) {
    unsigned int* d = dnp::bounds_err_buffer();
    unsigned int h[16] = {0};
    if (!d || hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    long long sum = 0;
    for (int i = 0; i < 16; ++i) { if (i < 8) sum += h[i]; if (host_out16) host_out16[i] = h[i]; }
    if (reset) (void)hipMemset(d, 0, sizeof(h));
    return sum;
}
#endif

}  // extern "C"
