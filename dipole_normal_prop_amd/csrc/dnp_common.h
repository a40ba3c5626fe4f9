// dnp_common.h - shared host-side helpers for libdnp.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

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

}  // namespace dnp
