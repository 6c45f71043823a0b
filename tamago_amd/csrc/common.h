// Shared host-side helpers for libtamago_hip.so (error reporting, HIP checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstdint>
#include <string>

#include "../../include/tamago_hip.h"

namespace tg {

std::string &last_error();
int fail(int code, const char *fmt, ...);

// Debug / experiment knobs (kernel-variant selectors, forced groupings, test hooks - INTEGRATION.md 2.6) are read through knob():
// it returns the variable's value only when TG_DEBUG_KNOBS=1 is set in the process environment, nullptr otherwise, so a product
// process ignores them (round 6).  The knobs a deployment may set - TG_FWD_ALGO, TG_HOST_THREADS, TG_SHARED_DEVICE,
// TG_SP_TIMING - are read with getenv directly.
const char *knob(const char *name);

#define TG_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return tg::fail(TG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                          \
    } while (0)

// hipFuncSetAttribute costs ~20 us of host time per call (measured: the single-tree leg lost 0.19 ms per move with it in
// front of every launch), so it is called once per kernel AND DEVICE - thread-safe, unlike a plain static flag: returns
// true for the first caller on `device` (devices beyond 63 always return true).
inline bool first_on_device(std::atomic<uint64_t> &mask, int device) {
    if (device < 0 || device > 63) return true;
    const uint64_t bit = 1ull << device;
    return (mask.fetch_or(bit, std::memory_order_acq_rel) & bit) == 0;
}

}  // namespace tg
