// Shared host-side helpers for libtamago_hip.so (error reporting, HIP checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/tamago_hip.h"

namespace tg {

std::string &last_error();
int fail(int code, const char *fmt, ...);

#define TG_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return tg::fail(TG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                          \
    } while (0)

}  // namespace tg
