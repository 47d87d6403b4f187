// Shared host-side helpers for liblocov_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/locov_hip.h"

namespace locov {

// thread-local error message returned by locov_last_error()
char *err_buf();
int set_error(int code, const char *fmt, ...);

inline hipStream_t as_stream(locov_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the launch that was just enqueued (no synchronisation).
int check_launch(const char *what);

constexpr int kWave = 64;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace locov

#define LOCOV_REQUIRE(cond, ...)                                          \
    do {                                                                  \
        if (!(cond)) return locov::set_error(LOCOV_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)
