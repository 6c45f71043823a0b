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

extern "C" int tg_legacy_exponentials(uint32_t *mt_key, int *mt_pos, size_t n, double *out) {
    if (!mt_key || !mt_pos || (!out && n)) return tg::fail(TG_ERR_ARG, "tg_legacy_exponentials: null argument");
    if (*mt_pos < 0 || *mt_pos > 624) return tg::fail(TG_ERR_ARG, "tg_legacy_exponentials: MT19937 position outside [0, 624]");
    tg::Mt19937 g;
    std::memcpy(g.key, mt_key, sizeof(g.key));
    g.pos = *mt_pos;
    for (size_t i = 0; i < n; ++i) out[i] = -std::log(1.0 - g.next_double());
    std::memcpy(mt_key, g.key, sizeof(g.key));
    *mt_pos = g.pos;
    return TG_OK;
}

// Host-only walk of a LegacyStream (what a search does to it): per step a window is staged (the step's draws + `slack`) and the
// step's draws are consumed; afterwards the generator state at the logical position is handed back - from the nearest snapshot,
// which is what this entry point lets the CPU tests check against numpy (tests/test_host_rng.py).
extern "C" int tg_legacy_stream_walk(const uint32_t *mt_key, int mt_pos, const int64_t *steps, int n_steps, int64_t slack,
                                     uint32_t *mt_key_out, int *mt_pos_out, double *next_draws_out, int n_next) {
    if (!mt_key || !steps || !mt_key_out || !mt_pos_out || n_steps < 0 || slack < 0 || (n_next > 0 && !next_draws_out))
        return tg::fail(TG_ERR_ARG, "tg_legacy_stream_walk: bad argument");
    if (mt_pos < 0 || mt_pos > 624) return tg::fail(TG_ERR_ARG, "tg_legacy_stream_walk: MT19937 position outside [0, 624]");
    tg::LegacyStream ls;
    ls.seed(mt_key, mt_pos);
    for (int i = 0; i < n_steps; ++i) {
        if (steps[i] < 0) return tg::fail(TG_ERR_ARG, "tg_legacy_stream_walk: negative step");
        ls.ensure((size_t)(steps[i] + slack));
        ls.consume((size_t)steps[i]);
    }
    const tg::Mt19937 &g = ls.state_at_position();
    std::memcpy(mt_key_out, g.key, sizeof(g.key));
    *mt_pos_out = g.pos;
    if (n_next > 0) {                                   // the staged draws at the position (what the device would be sent next)
        ls.ensure((size_t)n_next);
        std::memcpy(next_draws_out, ls.data(), (size_t)n_next * sizeof(double));
    }
    return TG_OK;
}

