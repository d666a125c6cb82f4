// Shared host-side helpers for libdsdiff (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <stdexcept>
#include <string>

namespace dsd {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

[[noreturn]] inline void fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(buf);
}

#define DSD_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) ::dsd::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define DSD_CHECK(cond, ...)                 \
    do {                                     \
        if (!(cond)) ::dsd::fail(__VA_ARGS__); \
    } while (0)

inline void check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) fail("launch of %s failed: %s", what, hipGetErrorString(e));
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Activation tensor view: NHWC fp32, contiguous.
struct T4 {
    float* p = nullptr;
    int n = 0, h = 0, w = 0, c = 0;
    int64_t numel() const { return (int64_t)n * h * w * c; }
    int64_t pixels() const { return (int64_t)n * h * w; }
};

}  // namespace dsd
