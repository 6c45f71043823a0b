// numpy's legacy random stream (RandomState = MT19937 + the legacy distributions) restated for
// the host side of the search: the reference draws its Dirichlet priors and Gumbel noise from
// the process-global legacy generator (mcts/tree.py:518 np.random.dirichlet, mcts/node.py:278
// np.random.gumbel), so a tree that has to reproduce the reference's visits needs the same
// doubles in the same order.
//
//   random_sample        u = (a * 2^26 + b) / 2^53,  a = next32 >> 5, b = next32 >> 6
//   standard_exponential e = -log(1 - u)                       (legacy, non-ziggurat)
//   dirichlet(ones(n))   e_0..e_{n-1} normalised               (legacy standard_gamma(1.0) is
//                                                               standard_exponential)
//   gumbel(0, 1)         -log(-log(1 - u)) = -log(e)
// `log` is glibc's, restated (csrc/legacy_rng_device.h glibc_log: the function the DEVICE streams use; here on the host for
// tg_legacy_exponentials / tg_glibc_log, which tests/test_host_rng.py holds against numpy, Python's math.log (= libm) and the
// reference-recorded draws without a GPU).  Since round 6 the search's streams live on the device; this header keeps the
// generator for those host entry points.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace tg {

struct Mt19937 {                 // state layout of numpy.random.RandomState.get_state()
    uint32_t key[624];
    int pos = 624;

    void regenerate() {
        constexpr uint32_t kMatrixA = 0x9908b0dfu, kUpper = 0x80000000u, kLower = 0x7fffffffu;
        int i = 0;
        for (; i < 624 - 397; ++i) {
            const uint32_t y = (key[i] & kUpper) | (key[i + 1] & kLower);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
        }
        for (; i < 623; ++i) {
            const uint32_t y = (key[i] & kUpper) | (key[i + 1] & kLower);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
        }
        const uint32_t y = (key[623] & kUpper) | (key[0] & kLower);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
        pos = 0;
    }
    uint32_t next32() {
        if (pos >= 624) regenerate();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    double next_double() {
        const uint32_t a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

}  // namespace tg
