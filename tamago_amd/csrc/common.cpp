// Library-level entry points of libtamago_hip.so: error reporting and device queries.
#include "common.h"

#include <cstdlib>

namespace tg {

std::string &last_error() {
    static thread_local std::string msg;
    return msg;
}

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

const char *knob(const char *name) {
    static const bool enabled = [] {
        const char *v = getenv("TG_DEBUG_KNOBS");
        return v && atoi(v) != 0;
    }();
    return enabled ? getenv(name) : nullptr;
}

}  // namespace tg

extern "C" {

int tg_abi_version(void) { return 1; }

const char *tg_last_error(void) { return tg::last_error().c_str(); }

int tg_device_count(int *count) {
    if (!count) return tg::fail(TG_ERR_ARG, "tg_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return tg::fail(TG_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return TG_OK;
}

}  // extern "C"

#include "legacy_stream.h"
#include "legacy_rng_device.h"

// The arithmetic of the device-resident legacy streams on the HOST (same functions: MT19937, random_sample, the restated glibc
// log), so that the CPU tests can hold it against numpy and the reference-recorded draws without a GPU.
extern "C" int tg_legacy_exponentials(uint32_t *mt_key, int *mt_pos, size_t n, double *out) {
    if (!mt_key || !mt_pos || (!out && n)) return tg::fail(TG_ERR_ARG, "tg_legacy_exponentials: null argument");
    if (*mt_pos < 0 || *mt_pos > 624) return tg::fail(TG_ERR_ARG, "tg_legacy_exponentials: MT19937 position outside [0, 624]");
    tg::Mt19937 g;
    std::memcpy(g.key, mt_key, sizeof(g.key));
    g.pos = *mt_pos;
    for (size_t i = 0; i < n; ++i) out[i] = -tg_rng::glibc_log(1.0 - g.next_double(), tg_rng::kLogTabHost);
    std::memcpy(mt_key, g.key, sizeof(g.key));
    *mt_pos = g.pos;
    return TG_OK;
}

// out[i] = the restated glibc log of x[i] (positive, normal arguments - what the streams feed it)
extern "C" int tg_glibc_log(const double *x, size_t n, double *out) {
    if ((!x || !out) && n) return tg::fail(TG_ERR_ARG, "tg_glibc_log: null argument");
    for (size_t i = 0; i < n; ++i) out[i] = tg_rng::glibc_log(x[i], tg_rng::kLogTabHost);
    return TG_OK;
}
