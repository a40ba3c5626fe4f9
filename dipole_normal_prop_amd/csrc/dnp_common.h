// dnp_common.h - shared host-side helpers for libdnp.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <new>

#include "../../include/dnp.h"

namespace dnp {

// thread-local error text (dnp_last_error)
void set_error(const char* fmt, ...);
void clear_error();

#define DNP_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            dnp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                    \
            return DNP_EHIP;                                                             \
        }                                                                                \
    } while (0)

#define DNP_REQUIRE(cond, ...)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            dnp::set_error(__VA_ARGS__);                                                 \
            return DNP_EINVAL;                                                           \
        }                                                                                \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef DNP_BOUNDS
// check builds only (pair_kernel.h, PairBounds): the library-owned device error word, eight counters by table
unsigned int* bounds_err_buffer();
#endif

// Nothing C++ may leave an extern "C" entry point (include/dnp.h: "never aborts ... no exceptions cross this line"): the
// entry points whose host side allocates (std::vector growth in the launch planner, the hash map of the cell merge) run
// their body through guarded(): std::bad_alloc -> DNP_ENOMEM, anything else -> DNP_EINTERNAL, message in dnp_last_error().
template <typename Body>
static inline int guarded(const char* what, Body&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        set_error("%s: out of host memory", what);
        return DNP_ENOMEM;
    } catch (...) {
        set_error("%s: unexpected C++ exception", what);
        return DNP_EINTERNAL;
    }
}

}  // namespace dnp
