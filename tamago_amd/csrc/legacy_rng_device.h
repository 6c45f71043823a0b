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

#ifdef TG_RNG_DEVICE_KERNELS      // (search.hip: the translation unit that launches them; common.cpp takes glibc_log only)
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

// The 624-word state of ONE wavefront in registers: lane l holds words l, l + 64, ..., l + 576 (k[9]: lanes 0..47).
// One regeneration (legacy_stream.h Mt19937::regenerate): word i needs the OLD words i, i + 1 and - below 227 - the old word
// i + 397, from 227 on the NEW word i - 227.  Register by register in rising order: i + 1 is the neighbouring lane (lane 0 of
// the next register for lane 63), i + 397 = lane + 13 of register r + 6 / r + 7, i - 227 = lane - 35 of register r - 3 / r - 4
// (already new) - cross-lane reads (ds_bpermute), no memory.  (The LDS version of round 6's first cut took ~1 600 cycles per
// regeneration - ten dependent read / write passes -, this one ~400; a window's cost is mostly regenerations.)
struct MtRegs {
    uint32_t k[10];
};

__device__ __forceinline__ void mt_load(MtRegs &m, const uint32_t *src, int lane) {
#pragma unroll
    for (int r = 0; r < 10; ++r) m.k[r] = (r < 9 || lane < 48) ? src[lane + 64 * r] : 0u;
}
__device__ __forceinline__ void mt_store(const MtRegs &m, uint32_t *dst, int lane) {
#pragma unroll
    for (int r = 0; r < 10; ++r)
        if (r < 9 || lane < 48) dst[lane + 64 * r] = m.k[r];
}

__device__ __forceinline__ void mt_twist(MtRegs &m, int lane) {
    const int up13 = (lane + 13) & 63, dn35 = (lane + 29) & 63, up1 = (lane + 1) & 63;
    const bool lo13 = lane + 13 < 64, hi35 = lane >= 35;
    // everything that reads OLD words first, ALL in flight together (fourteen cross-lane reads, then one wait): word i + 1 of every
    // register, word i + 397 of registers 0..3.  (Written as "all reads, fence, all uses": with a read next to its use the
    // compiler waited for each one in turn - ten round trips of ~100 cycles in a row, most of a regeneration.)
    uint32_t nx[10], su[4];
#pragma unroll
    for (int r = 0; r < 10; ++r) nx[r] = (uint32_t)__shfl((int)m.k[r], up1);
#pragma unroll
    for (int q = 0; q < 4; ++q) su[q] = (uint32_t)__shfl((int)m.k[6 + q], up13);
    __builtin_amdgcn_sched_barrier(0);
    uint32_t mix[10];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t nxt = nx[r];
        if (r < 9) {
            const uint32_t wrap = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.k[r + 1]);
            if (lane == 63) nxt = wrap;
        }
        mix[r] = mt_mix(m.k[r], nxt);                                 // (r = 9, lane 47: patched below with the NEW word 0)
    }
    const uint32_t old9 = m.k[9];
    // four levels of NEW words: registers 0..2, then 3..5, 6..8, 9 - each level reads the level before it once (lane - 35)
    uint32_t sh[7];
    m.k[0] = (lo13 ? su[0] : su[1]) ^ mix[0];
    m.k[1] = (lo13 ? su[1] : su[2]) ^ mix[1];
    m.k[2] = (lo13 ? su[2] : su[3]) ^ mix[2];
#pragma unroll
    for (int q = 0; q < 3; ++q) sh[q] = (uint32_t)__shfl((int)m.k[q], dn35);
    m.k[3] = (lane < 35 ? su[3] : sh[0]) ^ mix[3];
    m.k[4] = (hi35 ? sh[1] : sh[0]) ^ mix[4];
    m.k[5] = (hi35 ? sh[2] : sh[1]) ^ mix[5];
#pragma unroll
    for (int q = 3; q < 6; ++q) sh[q] = (uint32_t)__shfl((int)m.k[q], dn35);
    m.k[6] = (hi35 ? sh[3] : sh[2]) ^ mix[6];
    m.k[7] = (hi35 ? sh[4] : sh[3]) ^ mix[7];
    m.k[8] = (hi35 ? sh[5] : sh[4]) ^ mix[8];
    sh[6] = (uint32_t)__shfl((int)m.k[6], dn35);
    // register 9: word 623 (lane 47) mixes its old value with the NEW word 0
    const uint32_t new0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.k[0]);
    const uint32_t mix9 = lane == 47 ? mt_mix(old9, new0) : mix[9];
    m.k[9] = (hi35 ? sh[6] : sh[5]) ^ mix9;
}

// State after `words` more 32-bit outputs, numpy's lazy convention (the block is regenerated when the next word is asked for).
__device__ __forceinline__ int mt_skip(MtRegs &m, int pos, long long words, int lane) {
    while (words > 0) {
        if (pos == kMtN) { mt_twist(m, lane); pos = 0; }
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
    uint32_t *words;              // scratch [T][words_pitch]: the tempered 32-bit outputs of this piece, in stream order from word 0
    long long words_pitch;        // of the state's current block (rng_words_kernel -> rng_draws_kernel)
    int *pos0;                    // [T] position in that block the piece starts at
    // Snapshots (few trees): the raw state every kSnapEvery blocks behind `base`, so that committing a whole search's consumption
    // (one 19x19 tree: ~1 500 regenerations = 0.6-0.8 ms in front of tg_search_stream_state) starts from the nearest one.
    uint32_t *snap;               // [T][snap_cap][624] or null
    int snap_cap;
    int *snap_n;                  // [T] valid snapshots of the CURRENT base (snapshot i = the state (i + 1) * kSnapEvery blocks on)
    long long *cont_blk;          // [T] how many blocks behind `base` the state in `cont` is
};
constexpr int kSnapEvery = 32;

// A piece in two launches (a regeneration is serial, the logarithms are not - and they are 6x the work):
//   rng_words_kernel  one wavefront per tree: commits the consumed draws, walks the state over the piece's blocks and leaves
//                     their tempered words in the scratch row + the state behind the piece (~450 cycles per block of 312 draws);
//   rng_draws_kernel  one thread per draw: two words -> the 53-bit uniform -> -log(1 - u) (or the Gumbel value), any number of
//                     workgroups.
// (Round 6's first cut did both in one workgroup per tree, every wavefront regenerating its way from block to block: sixteen
// wavefronts on one CU share four SIMDs, and a 70 k-draw piece of one 19x19 tree took 210 us; this takes ~55.)
__global__ __launch_bounds__(64) void rng_words_kernel(FillArgs a) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (a.skip && a.skip[t]) return;
    uint32_t *st = (a.from_cont ? a.cont : a.base) + (size_t)t * kStateWords;
    MtRegs m;
    mt_load(m, st, lane);
    int pos = (int)st[kMtN];
    long long abs0 = (a.from_cont && a.cont_blk) ? a.cont_blk[t] : 0;     // blocks behind `base` of the state in hand
    int n_snap = a.snap ? a.snap_n[t] : 0;
    const long long lag = (!a.from_cont && a.lag) ? a.lag[t] : 0;
    if (lag > 0) {                                                   // commit what the searches consumed: base moves
        const long long end = (long long)pos + 2 * lag;              // first unconsumed word, counted from word 0 of block 0
        const long long be = (end - 1) / kMtN;                       // its block, lazily (a boundary stays with the block before)
        long long at = 0;
        if (a.snap) {
            long long k = be / kSnapEvery;
            if (k > n_snap) k = n_snap;
            if (k > 0) {
                mt_load(m, a.snap + ((size_t)t * a.snap_cap + (size_t)(k - 1)) * kMtN, lane);
                at = k * kSnapEvery;
            }
        }
        for (; at < be; ++at) mt_twist(m, lane);
        pos = (int)(end - kMtN * be);
        mt_store(m, st, lane);
        if (lane == 0) st[kMtN] = (uint32_t)pos;
        n_snap = 0;                                                  // (they belonged to the old base)
    }
    if (lane == 0) a.pos0[t] = pos;
    // absolute word index = pos + stream word; block b holds [624 b, 624 b + 624); block 0 is the state as it stands
    const long long end = (long long)pos + 2 * a.count;              // first word behind the piece
    const long long n_blocks = a.count > 0 ? (end - 1) / kMtN + 1 : 0;
    uint32_t *row = a.words + (size_t)t * a.words_pitch;
    for (long long b = 0; b < n_blocks; ++b) {
        if (b > 0) {
            mt_twist(m, lane);
            const long long abs_b = abs0 + b;
            if (a.snap && !a.noise && abs_b % kSnapEvery == 0) {
                const long long idx = abs_b / kSnapEvery - 1;
                if (idx < a.snap_cap) {
                    mt_store(m, a.snap + ((size_t)t * a.snap_cap + (size_t)idx) * kMtN, lane);
                    if (idx + 1 > n_snap) n_snap = (int)(idx + 1);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 10; ++r)
            if (r < 9 || lane < 48) row[kMtN * b + lane + 64 * r] = mt_temper(m.k[r]);
    }
    // the state behind the piece, lazily (a position on a block boundary stays "624" of the block before)
    const int pos_e = a.count > 0 ? (int)(end - kMtN * (n_blocks - 1)) : pos;
    uint32_t *dst = (a.noise ? a.base : a.cont) + (size_t)t * kStateWords;
    mt_store(m, dst, lane);
    if (lane == 0) {
        dst[kMtN] = (uint32_t)pos_e;
        if (a.noise) n_snap = 0;                                     // (base moved behind the noise draws)
        if (a.snap) a.snap_n[t] = n_snap;
        if (a.cont_blk && !a.noise) a.cont_blk[t] = abs0 + (n_blocks > 0 ? n_blocks - 1 : 0);
    }
}

__global__ __launch_bounds__(256) void rng_draws_kernel(FillArgs a) {
    __shared__ double tab[256];
    const int t = blockIdx.y;
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (a.skip && a.skip[t]) {                                       // (a parked tree: zero noise, the stream does not move)
        if (a.noise && j < a.count) a.noise[(size_t)t * a.count + j] = 0.0;
        return;
    }
    tab[threadIdx.x] = kLogTab[threadIdx.x];
    __syncthreads();
    if (j >= a.count) return;
    const uint32_t *w = a.words + (size_t)t * a.words_pitch + a.pos0[t] + 2 * j;
    const unsigned long long bits = ((unsigned long long)(w[0] >> 5) << 26) | (unsigned long long)(w[1] >> 6);
    const double u = (double)bits * 0x1p-53;                         // (a * 67108864.0 + b) / 9007199254740992.0, exact
    const double e = -glibc_log(1.0 - u, tab);
    if (a.noise) a.noise[(size_t)t * a.count + j] = -glibc_log(e, tab);
    else a.out[(size_t)t * a.pitch + a.first + j] = e;
}

#endif  // TG_RNG_DEVICE_KERNELS

}  // namespace tg_rng
