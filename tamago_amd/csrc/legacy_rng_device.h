// numpy's legacy random stream ON THE DEVICE (round 6): MT19937, random_sample and the two logarithms behind the reference's
// Dirichlet priors (mcts/tree.py:509-519: np.random.dirichlet(ones(n)) = normalised standard exponentials -log(1 - u)) and
// Gumbel noise (mcts/node.py:275-278: -log(-log(1 - u))), bit for bit.
//
//   * MT19937 + the 53-bit double are integer work (csrc/legacy_stream.h states them for the host).
//   * `log` is glibc's (numpy's legacy distributions call libm): sysdeps/ieee754/dbl-64/e_log.c, the table-driven algorithm
//     of ARM's optimized-routines, in the build x86-64 selects on a CPU with FMA (`__log_fma`: -mfma -mavx2, so GCC contracts
//     a * b + c).  glibc_log() below repeats THAT build's operations one by one - which products are fused and which are
//     rounded first was read off the emitted instructions (libm.so.6 of glibc 2.35, 0x76660) and is noted per line; the
//     constants come from the same binary (glibc_log_table.h).  A CPU restatement of the same sequence agrees with libm on
//     6e7 arguments (tests/test_host_rng.py keeps a smaller run of that check); tests/test_gpu_rng.py pins the device function
//     to tests/golden/rng.npz and to numpy on the box.  The file is compiled with -ffp-contract=off: every fma is explicit.
//
// Per tree the device keeps the generator state AT THE LOGICAL POSITION of its stream (`base`: 624 words + pos, numpy's
// get_state() layout incl. its lazy regeneration: pos = 624 means "twist before the next word") and a continuation state
// (`cont`) behind the last generated piece of a window.  The host only tracks how many draws were consumed since (`lag`).
#pragma once
#include "glibc_log_table.h"

namespace tg_rng {

constexpr int kMtN = 624, kMtM = 397, kStateWords = 625;     // key[624] + pos

__host__ __device__ __forceinline__ double glibc_log(double x, const double *tab) {     // tab: kLogTab (an LDS copy) / kLogTabHost
    uint64_t ix;
    __builtin_memcpy(&ix, &x, 8);
    if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {          // 1 - 0x1p-4 <= x < 1 + 0x1.09p-4
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        double t1 = __builtin_fma(r, kLogB[2], kLogB[1]);
        double t2 = __builtin_fma(r, kLogB[5], kLogB[4]);
        const double r2 = r * r;
        double t3 = __builtin_fma(r, kLogB[8], kLogB[7]);
        t1 = __builtin_fma(r2, kLogB[3], t1);
        t2 = __builtin_fma(r2, kLogB[6], t2);
        const double r3 = r * r2;                                     // (rounded product, then used as a factor)
        t3 = __builtin_fma(r2, kLogB[9], t3);
        t3 = __builtin_fma(r3, kLogB[10], t3);
        double p = __builtin_fma(t3, r3, t2);
        p = __builtin_fma(p, r3, t1);
        const double rw = __builtin_fma(r, 0x1p27, r);                // r + w, w = r * 2^27 (fused)
        const double rhi = __builtin_fma(-0x1p27, r, rw);             // (r + w) - w (fused)
        const double rhi2 = rhi * rhi;
        const double rlo = r - rhi;
        const double hi = __builtin_fma(rhi2, kLogB[0], r);           // r + rhi * rhi * B0
        const double d = r - hi;
        const double s = r + rhi;
        double lo = __builtin_fma(rhi2, kLogB[0], d);                 // r - hi + w
        const double m = kLogB[0] * rlo;
        lo = __builtin_fma(m, s, lo);
        const double y = __builtin_fma(p, r3, lo);
        return y + hi;
    }
    if (x == 0.0) return -__builtin_inf();                            // (never a subnormal here: the arguments are 1 - u and -log(1 - u))
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    double z;
    __builtin_memcpy(&z, &iz, 8);
    const double invc = tab[2 * i], logc = tab[2 * i + 1];
    const double kd = (double)k;
    const double r = __builtin_fma(z, invc, -1.0);
    const double w = __builtin_fma(kd, kLn2Hi, logc);
    const double p12 = __builtin_fma(r, kLogA[2], kLogA[1]);
    const double hi = w + r;
    const double r2 = r * r;
    double lo = (w - hi) + r;
    lo = __builtin_fma(kd, kLn2Lo, lo);
    const double rr2 = r * r2;
    const double p34 = __builtin_fma(r, kLogA[4], kLogA[3]);
    lo = __builtin_fma(r2, kLogA[0], lo);
    const double q = __builtin_fma(p34, r2, p12);
    const double y = __builtin_fma(rr2, q, lo);
    return y + hi;
}

__device__ __forceinline__ void rng_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// One regeneration of the 624-word state in LDS by ONE wavefront (legacy_stream.h Mt19937::regenerate): word i needs the OLD
// words i, i + 1 and - below 227 - the old word i + 397, from 227 on the NEW word i - 227.  Passes of 64 words in rising order:
// a pass reads before it writes (one instruction stream), and i - 227 always lies in an earlier pass.
__device__ __forceinline__ void mt_twist(uint32_t *key, int lane) {
    for (int base = 0; base < kMtN - 1; base += 64) {
        const int i = base + lane;
        const bool on = i < kMtN - 1;
        uint32_t v = 0;
        if (on) v = (i < kMtN - kMtM ? key[i + kMtM] : key[i - (kMtN - kMtM)]) ^ mt_mix(key[i], key[i + 1]);
        rng_wave_sync();
        if (on) key[i] = v;
        rng_wave_sync();
    }
    if (lane == 0) key[kMtN - 1] = key[kMtM - 1] ^ mt_mix(key[kMtN - 1], key[0]);
    rng_wave_sync();
}

// State after `words` more 32-bit outputs, numpy's lazy convention (the block is regenerated when the next word is asked for).
// Returns the new pos; key is twisted as often as needed.  One wavefront.
__device__ __forceinline__ int mt_skip(uint32_t *key, int pos, long long words, int lane) {
    while (words > 0) {
        if (pos == kMtN) { mt_twist(key, lane); pos = 0; }
        const long long step = words < (long long)(kMtN - pos) ? words : (long long)(kMtN - pos);
        pos += (int)step;
        words -= step;
    }
    return pos;
}

struct FillArgs {
    uint32_t *base, *cont;        // [T][625]
    const long long *lag;         // [T] draws consumed since `base` was brought up to date (host-mapped); null: none
    const unsigned char *skip;    // [T] != 0: leave the tree alone (null: none)
    double *out;                  // window [T][pitch]; this piece = columns [first, first + count)
    long long pitch, first, count;
    int from_cont;                // 0: base += lag, generate from base (base itself does not move); 1: go on behind the last piece
    double *noise;                // != null: Gumbel mode - out is unused, noise[T][count] = -log(e), and base moves behind the draws
};

// A piece of every tree's window.  One workgroup of NW wavefronts per tree; wavefront w takes the stream's blocks w, w + NW, ...
// of 624 words - each from its own copy of the state, twisting NW times from one block to its next (a twist is ~450 cycles, the
// 312 logarithms of a block ~2 500: no hand-offs, and sixteen wavefronts still divide a long window by ~10) - and wavefront 0
// also leaves the state behind the piece in `cont`.
template <int NW>
__global__ __launch_bounds__(64 * NW) void rng_fill_kernel(FillArgs a) {
    __shared__ uint32_t keys[NW][kMtN];
    __shared__ double tab[256];
    __shared__ int start_pos;
    const int t = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (a.skip && a.skip[t]) {                                       // (a parked tree: zero noise, the stream does not move)
        if (a.noise)
            for (long long i = threadIdx.x; i < a.count; i += 64 * NW) a.noise[(size_t)t * a.count + i] = 0.0;
        return;
    }
    for (int i = threadIdx.x; i < 256; i += 64 * NW) tab[i] = kLogTab[i];
    uint32_t *st = (a.from_cont ? a.cont : a.base) + (size_t)t * kStateWords;
    if (wave == 0) {
        for (int i = lane; i < kMtN; i += 64) keys[0][i] = st[i];
        int pos = (int)st[kMtN];
        rng_wave_sync();
        const long long lag = (!a.from_cont && a.lag) ? a.lag[t] : 0;
        if (lag > 0) {                                               // commit what the searches consumed: base moves
            pos = mt_skip(keys[0], pos, 2 * lag, lane);
            for (int i = lane; i < kMtN; i += 64) st[i] = keys[0][i];
            if (lane == 0) st[kMtN] = (uint32_t)pos;
        }
        if (lane == 0) start_pos = pos;
    }
    __syncthreads();
    const int pos0 = start_pos;
    if (wave > 0)
        for (int i = lane; i < kMtN; i += 64) keys[wave][i] = keys[0][i];
    __syncthreads();                                                 // (wave 0 twists keys[0] from here on)
    uint32_t *key = keys[wave];
    // absolute word index = pos0 + stream word; block b holds [624 b, 624 b + 624); block 0 is the state as loaded
    const long long last_word = (long long)pos0 + 2 * a.count - 1;   // second word of the last draw
    const long long n_blocks = a.count > 0 ? last_word / kMtN + 1 : 0;
    long long at = 0;                                                // block the wavefront's key holds
    uint32_t prev_last = 0;                                          // word 623 of block at - 1 (a draw may straddle two blocks)
    for (long long b = wave; b < n_blocks; b += NW) {
        while (at < b) {
            prev_last = key[kMtN - 1];
            mt_twist(key, lane);
            at += 1;
        }
        // draws whose second word lies in block b:  624 b <= pos0 + 2 j + 1 < 624 b + 624
        long long j_lo = (kMtN * b - pos0 - 1 + 1) / 2;              // ceil((624 b - pos0 - 1) / 2), operands >= 0 when b >= 1
        if (b == 0 || j_lo < 0) j_lo = 0;
        long long j_hi = (kMtN * b + kMtN - 1 - pos0 - 1) / 2 + 1;   // floor(...) + 1, exclusive
        if (kMtN * b + kMtN - 1 - pos0 - 1 < 0) j_hi = 0;
        if (j_hi > a.count) j_hi = a.count;
        for (long long j = j_lo + lane; j < j_hi; j += 64) {
            const long long a0 = (long long)pos0 + 2 * j - kMtN * b;   // index of the first word inside block b, or -1
            const uint32_t w0 = mt_temper(a0 < 0 ? prev_last : key[a0]);
            const uint32_t w1 = mt_temper(key[a0 + 1]);
            const unsigned long long bits = ((unsigned long long)(w0 >> 5) << 26) | (unsigned long long)(w1 >> 6);
            const double u = (double)bits * 0x1p-53;                 // (a * 67108864.0 + b) / 9007199254740992.0, exact
            const double e = -glibc_log(1.0 - u, tab);
            if (a.noise) a.noise[(size_t)t * a.count + j] = -glibc_log(e, tab);
            else a.out[(size_t)t * a.pitch + a.first + j] = e;
        }
    }
    if (wave == 0) {
        // the state behind the piece: absolute word pos0 + 2 count, lazily (a position on a block boundary stays "624" of the
        // block before)
        const long long end = (long long)pos0 + 2 * a.count;
        long long be = end > 0 ? (end - 1) / kMtN : 0;
        if (a.count == 0) be = 0;
        while (at < be) { mt_twist(key, lane); at += 1; }
        // (at > be cannot happen: wavefront 0's last block is <= the last block, and be >= last block when count > 0)
        const int pos_e = a.count > 0 ? (int)(end - kMtN * be) : pos0;
        uint32_t *dst = (a.noise ? a.base : a.cont) + (size_t)t * kStateWords;
        for (int i = lane; i < kMtN; i += 64) dst[i] = key[i];
        if (lane == 0) dst[kMtN] = (uint32_t)pos_e;
    }
}

}  // namespace tg_rng
