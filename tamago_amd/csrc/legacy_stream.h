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
// `log` is the libm the numpy build links against (checked bit-for-bit against numpy in
// tests/test_host_rng.py).
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

// Position-addressable view of one legacy stream as exponentials e_i = -log(1 - u_i): values
// are generated ahead of their consumption (the device gets whole windows), `consume` moves
// the logical position, and the generator state AT that position can be handed back to numpy.
struct LegacyStream {
    Mt19937 ahead;               // generator at position base + available()
    Mt19937 behind;              // generator at position base - lag
    size_t lag = 0;              // draws consumed since `behind` was last brought up to date
    std::vector<double> buf;     // buf[head..] = e at positions base, base + 1, ...
    size_t head = 0;
    bool seeded = false;
    // Snapshots of `ahead` every kSnapEvery generated draws (round 5): handing the state at the logical position back to numpy
    // (search_best_move does, per move: mcts/tree.py leaves np.random where the search left it) used to replay every consumed
    // draw through `behind` - 0.2 ms per 9x9 move, 1.5 ms per 19x19 move (half a million draws).  From the nearest snapshot it
    // is at most kSnapEvery draws.  generated / consumed count draws since seed(); snapshots older than the position are dropped.
    static constexpr size_t kSnapEvery = 2048;
    size_t generated = 0, consumed = 0;
    std::vector<std::pair<size_t, Mt19937>> snaps;

    void seed(const uint32_t *key624, int pos) {
        std::memcpy(ahead.key, key624, sizeof(ahead.key));
        ahead.pos = pos;
        behind = ahead;
        lag = 0;
        buf.clear();
        head = 0;
        generated = consumed = 0;
        snaps.clear();
        seeded = true;
    }
    size_t available() const { return buf.size() - head; }
    const double *data() const { return buf.data() + head; }
    void ensure(size_t need) {
        if (available() >= need) return;
        if (head) {
            buf.erase(buf.begin(), buf.begin() + (std::ptrdiff_t)head);
            head = 0;
        }
        const size_t have = buf.size();
        buf.resize(need);
        for (size_t i = have; i < need; ++i) {
            if (generated % kSnapEvery == 0 && generated > 0) snaps.emplace_back(generated, ahead);
            buf[i] = -std::log(1.0 - ahead.next_double());
            ++generated;
        }
    }
    void consume(size_t n) {     // n <= available()
        head += n;
        lag += n;
        consumed += n;
        // keep the last snapshot at or before the position, drop the ones before it
        size_t keep = 0;
        while (keep + 1 < snaps.size() && snaps[keep + 1].first <= consumed) ++keep;
        if (keep > 0) snaps.erase(snaps.begin(), snaps.begin() + (std::ptrdiff_t)keep);
    }
    // generator state at the logical position (next unconsumed draw); at most kSnapEvery draws of replay
    const Mt19937 &state_at_position() {
        if (lag > kSnapEvery && !snaps.empty() && snaps.front().first <= consumed && consumed - snaps.front().first < lag) {
            behind = snaps.front().second;
            lag = consumed - snaps.front().first;
        }
        for (; lag; --lag) (void)behind.next_double();
        return behind;
    }
};

}  // namespace tg
