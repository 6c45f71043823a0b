// DualNet forward for gfx950 with SPLIT OPERANDS: the fp32 3x3 convolutions of the residual tower run on
// the 16-bit matrix pipe (v_mfma_f32_16x16x32_{f16,bf16}: 16x the rate of the fp32 MFMA) without giving up
// fp32-class accuracy, by splitting every fp32 operand into 16-bit pieces whose sum is (nearly) the operand
// and multiplying the pieces pairwise with fp32 accumulation:
//
//   f16 x 2 pieces  a = ah + 2^-11 al',  ah = rn16(a), al' = rn16((a - ah) 2^11)    (22 significand bits)
//                   a w ~ ah wh + 2^-11 (ah wl' + al' wh)          3 MFMAs, dropped term 2^-22 a w
//
// (fp32 itself rounds every product-sum to 2^-24.)  The cross terms go to a SECOND accumulator set and are
// scaled once per layer: the matrix pipe aligns the 32 products of an MFMA to its accumulator input, so small
// cross products added straight onto the large main sum would lose their low bits one MFMA at a time (that
// is what a 3-piece bf16 variant with one accumulator did: 1e-3 logit error, measured) - kept apart they
// keep full precision whatever the operand's magnitude.  Weights are pre-scaled by a power of two per layer
// (folded back into the BN scale).  Measured against the reference's fp64 forward the kernel is as close as
// the reference's own fp32 path (tools/check_forward_accuracy.py: 5.9e-7 / 1.1e-5 vs 6.7e-7 / 1.1e-5).
// f16 has a range limit: a layer output beyond 6e4 raises the network's range flag and the batch is redone
// by the exact-fp32 kernel (tg_net_forward_dev queues that launch right behind; it exits at once when the
// flag is clear) - no host round trip, no change of results for such networks.
//
// Same fusion as net_forward.hip (one persistent workgroup carries G boards through stem, 12 convolutions
// and both heads; activations never leave LDS), direct 3x3 convolution as an implicit GEMM
//   M = G*P rows (positions), N = 64, K = 9 taps x 64 channels.
// What is different, and why (tools/microbench/split_bf16_loop.hip, profiles/r02_microbench_split_loop.txt):
// at 16x the MFMA rate the loop is bound by LDS -> VGPR fragment traffic, which does NOT overlap the MFMAs
// of the same SIMD (~20 cycles per ds_read_b128 on top of the MFMA time).  So:
//   * wave tile = ALL 64 output channels x RTW row-tiles (64 x 64 for G = 3): the fewest fragment bytes per
//     MFMA a 256-row workgroup allows; one wave per SIMD, accumulators + residual + two fragment sets in the
//     512-register file;
//   * activations are stored as NP 16-bit images [piece][k-chunk][row][4 x 16 B], slot = lg ^ ((row >> 1) & 3):
//     conflict-free ds_read_b128 B-fragments for every tap shift (brute-forced against the hardware's lane
//     groups); an all-zero row per image serves the padding taps (uniform offsets, no branches);
//   * weights stream tap by tap (16 / 24 KB) from L2 into a two-slot LDS ring with global_load_lds_dwordx4 in
//     MFMA A-fragment order: one copy per workgroup instead of one per wave, conflict-free reads, ONE barrier
//     per tap;
//   * fragment loads are inline-asm ds_read_b128, issued between the MFMAs of the previous k-chunk, with one
//     hand-placed s_waitcnt per chunk (hipcc puts lgkmcnt(0) right behind loads it knows about).
#include "net_device.h"

#include <cstring>
#include <type_traits>
#include <utility>

namespace {

typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int kTowerLayers = 12;
constexpr int kSplitTaps = 1 + 9 * kTowerLayers;      // stem as one K = 64 pseudo-tap + 12 x 9

struct FmtF16 {
    static constexpr int NP = 2, NPROD = 3, NACC = 2;
    // product q: weight piece PA[q] x activation piece PB[q] -> accumulator set PC[q]
    static constexpr int PA[3] = {0, 1, 0}, PB[3] = {0, 0, 1}, PC[3] = {0, 1, 1};
};

template <int S, int G, typename F>
struct SplitCfg {
    static constexpr int P = S * S, A = P + 1, M = G * P;
    static constexpr int MT = (M + 15) / 16;
    static constexpr int RTW = G == 3 ? 4 : 2;                    // row-tiles per wave
    static constexpr int NW = (MT + RTW - 1) / RTW;               // waves per workgroup
    static constexpr int NTHR = NW * 64;
    static constexpr int IMG = (M + 1) * 64;                      // one [row][64 B] image + its zero row
    static constexpr int ACT_BYTES = F::NP * 2 * IMG;             // image index = piece * 2 + kc
    static constexpr int W_OFF = (ACT_BYTES + 255) & ~255;
    static constexpr int W_TAP = 2 * F::NP * 4 * 1024;            // [kc][piece][ct][lane][16 B]
    static constexpr int STAGE = W_OFF + 2 * W_TAP;               // input planes [G][6][P] fp32
    static constexpr int PIPE_BYTES = STAGE + ((G * 6 * P * 4 + 255) & ~255);
    // head phase (after the last layer): fp32 activations [M][72 floats] from offset 0 + run_heads scratch
    static constexpr int ROW_BYTES = kRowBytes;
    static constexpr int AUX = ((M * kRowBytes + 255) & ~255);
    static constexpr int HEAD_BYTES = AUX + G * (3 * P + A + 4) * 4 + 256;
    static constexpr int LDS_BYTES = PIPE_BYTES > HEAD_BYTES ? PIPE_BYTES : HEAD_BYTES;
};

template <int OFFSET>
__device__ __forceinline__ void lds_load_frag(i32x4v &dst, int addr) {
    static_assert(OFFSET >= 0 && OFFSET < 65536 && OFFSET % 16 == 0, "ds_read_b128 offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFFSET));
}

// compile-time loop: fn(std::integral_constant<int, 0>{}), ..., fn(std::integral_constant<int, N - 1>{})
template <typename Fn, int... Is>
__device__ __forceinline__ void static_for_impl(Fn &&fn, std::integer_sequence<int, Is...>) {
    (fn(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename Fn>
__device__ __forceinline__ void static_for(Fn &&fn) {
    static_for_impl(fn, std::make_integer_sequence<int, N>{});
}

template <typename F>
__device__ __forceinline__ f32x4 mfma16(const i32x4v &w, const i32x4v &a, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
}

// Split four fp32 values (consecutive channels) into NP packed 16-bit quadruples.
template <typename F>
__device__ __forceinline__ void split4(const f32x4 v, uint2 (&out)[F::NP]) {
    {
        f16x2 h01 = __builtin_convertvector(f32x2v{v[0], v[1]}, f16x2);
        f16x2 h23 = __builtin_convertvector(f32x2v{v[2], v[3]}, f16x2);
        const f32x2v b01 = __builtin_convertvector(h01, f32x2v), b23 = __builtin_convertvector(h23, f32x2v);
        f16x2 l01 = __builtin_convertvector(f32x2v{(v[0] - b01[0]) * 2048.f, (v[1] - b01[1]) * 2048.f}, f16x2);
        f16x2 l23 = __builtin_convertvector(f32x2v{(v[2] - b23[0]) * 2048.f, (v[3] - b23[1]) * 2048.f}, f16x2);
        out[0] = uint2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
        out[1] = uint2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
    }
}

template <int S, int G, typename F>
__global__ __launch_bounds__((SplitCfg<S, G, F>::NTHR), 1) void dualnet_fwd_split_kernel(
    NetDev net, const float *__restrict__ planes, int batch, int want_logits,
    float *__restrict__ policy, float *__restrict__ value, int *__restrict__ overflow) {
    using C = SplitCfg<S, G, F>;
    constexpr int P = C::P, M = C::M, RTW = C::RTW, NTHR = C::NTHR, NP = F::NP, IMG = C::IMG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

    // ---- per-lane geometry of this wave's row-tiles ------------------------------------------------
    int base_row[RTW];
    unsigned mask[RTW];                                   // bit t: tap t of this row is inside its board
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
        const int row = (wave * RTW + r) * 16 + li;
        const int p = row % P, y = p / S, x = p - y * S;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (row < M && yy >= 0 && yy < S && xx >= 0 && xx < S) m |= 1u << t;
        }
        mask[r] = m;
        base_row[r] = row;
    }
    // zero rows of the activation images (written once; the epilogues never touch row M)
    for (int e = tid; e < NP * 2 * 16; e += NTHR)
        reinterpret_cast<unsigned *>(smem + (e >> 4) * IMG + M * 64)[e & 15] = 0u;

    const unsigned char *wimg = net.wsplit + lane * 16;
    const float *bn_scale = net.sscale;
    // weight ring: global tap g -> slot g & 1; every wave copies its share of the 2 * NP * 4 one-KB pieces
    auto dma = [&](int g) {
        if (g >= kSplitTaps) return;
        constexpr int PIECES = 2 * NP * 4;
#pragma unroll
        for (int q = 0; q < (PIECES + C::NW - 1) / C::NW; ++q) {
            const int piece = q * C::NW + wave;
            if (piece < PIECES)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(wimg + ((size_t)g * PIECES + piece) * 1024),
                    (__attribute__((address_space(3))) void *)(smem + C::W_OFF + (g & 1) * C::W_TAP + piece * 1024), 16, 0, 0);
        }
    };

    int ovf = 0;
    const int n_groups = (batch + G - 1) / G;
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int b0 = grp * G;
        dma(0);
        dma(1);
        // ---- input planes -> LDS -> im2col'ed, split "layer -1" activations (K = 9 taps x 6 planes, padded to 64) ----
        {
            float *st = reinterpret_cast<float *>(smem + C::STAGE);
            for (int e = tid; e < G * 6 * P; e += NTHR) {
                const int b = b0 + e / (6 * P);
                st[e] = b < batch ? __builtin_nontemporal_load(&planes[(size_t)b0 * 6 * P + e]) : 0.f;
            }
            __syncthreads();
            for (int e = tid; e < M * 8; e += NTHR) {
                const int row = e >> 3, sl = e & 7;                // slot sl holds k = 8 sl .. 8 sl + 7
                const int bl = row / P, p = row - bl * P, y = p / S, x = p - y * S;
                f32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = sl * 8 + j, t = k / 6, c = k - t * 6;
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    const bool ok = k < 54 && yy >= 0 && yy < S && xx >= 0 && xx < S;
                    const float v = ok ? st[(bl * 6 + c) * P + yy * S + xx] : 0.f;
                    if (j < 4) lo[j] = v; else hi[j - 4] = v;
                }
                uint2 plo[NP], phi[NP];
                split4<F>(lo, plo);
                split4<F>(hi, phi);
                const int kc = sl >> 2, slot = (sl & 3) ^ ((row >> 1) & 3);
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    *reinterpret_cast<uint4 *>(smem + (q * 2 + kc) * IMG + row * 64 + slot * 16) =
                        uint4{plo[q].x, plo[q].y, phi[q].x, phi[q].y};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of taps 0 and 1 have landed
        __syncthreads();                                  // activations written; everybody's pieces have landed

        f32x4 acc[F::NACC][4][RTW];
        f32x4 res[4][RTW];
        i32x4v fa[2][4][NP], fb[2][RTW][NP];              // two fragment sets: weights (A operand), activations (B)

        // B-fragment base address of row-tile r for tap `tap` of layer kind `stem`
        auto row_addr = [&](int r, int tap, bool stem) {
            const int toff = stem ? 0 : (tap / 3 - 1) * S + (tap % 3 - 1);
            const bool ok = stem ? base_row[r] < M : ((mask[r] >> tap) & 1u) != 0;
            const int row = ok ? base_row[r] + toff : M;
            return lds0 + row * 64 + ((lg ^ ((row >> 1) & 3)) << 4);
        };
        auto load_b = [&](i32x4v &dst, auto P_, auto KC_, int addr) {
            // image (p, kc) at (p * 2 + kc) * IMG; the 16-bit offset field reaches 64 KB
            constexpr int off = (decltype(P_)::value * 2 + decltype(KC_)::value) * IMG;
            lds_load_frag<off>(dst, addr);
        };

        int g = 0;                                        // global tap index
#pragma unroll 1
        for (int layer = 0; layer <= kTowerLayers; ++layer) {
            const bool stem = layer == 0;
            const int ntaps = stem ? 1 : 9;
#pragma unroll
            for (int s = 0; s < F::NACC; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < RTW; ++r) acc[s][c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
            // first chunk of the layer: activation fragments now (the epilogue just wrote them); the weight
            // fragments of (g, kc 0) were requested by the previous layer's last chunk, except for the stem
            {
                int ba[RTW];
#pragma unroll
                for (int r = 0; r < RTW; ++r) ba[r] = row_addr(r, 0, stem);
                if (stem) {
                    const int wa = lds0 + C::W_OFF + (g & 1) * C::W_TAP + lane * 16;
                    static_for<4 * NP>([&](auto J) {
                        constexpr int c = decltype(J)::value % 4, p = decltype(J)::value / 4;
                        lds_load_frag<(p * 4 + c) * 1024>(fa[0][c][p], wa);
                    });
                }
                static_for<RTW * NP>([&](auto J) {
                    constexpr int r = decltype(J)::value % RTW, p = decltype(J)::value / RTW;
                    load_b(fb[0][r][p], std::integral_constant<int, p>{}, std::integral_constant<int, 0>{}, ba[r]);
                });
            }
#pragma unroll 1
            for (int tap = 0; tap < ntaps; ++tap, ++g) {
                int ba[RTW], bn[RTW];
#pragma unroll
                for (int r = 0; r < RTW; ++r) {
                    ba[r] = row_addr(r, tap, stem);
                    bn[r] = row_addr(r, tap + 1 < ntaps ? tap + 1 : tap, stem);
                }
                const int wa_cur = lds0 + C::W_OFF + (g & 1) * C::W_TAP + lane * 16;
                const int wa_nxt = lds0 + C::W_OFF + ((g + 1) & 1) * C::W_TAP + lane * 16;
                auto chunk = [&](auto KC_) {
                    constexpr int kc = decltype(KC_)::value;
                    // ---- wait for this chunk's fragments (set kc); the loads were issued a chunk ago ----
                    static_assert(NP == 2, "the wait below names every weight fragment register");
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(fa[kc][0][0]), "+v"(fa[kc][1][0]), "+v"(fa[kc][2][0]), "+v"(fa[kc][3][0]),
                                   "+v"(fa[kc][0][1]), "+v"(fa[kc][1][1]), "+v"(fa[kc][2][1]), "+v"(fa[kc][3][1]));
#pragma unroll
                    for (int r = 0; r < RTW; ++r)
#pragma unroll
                        for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(fb[kc][r][p]));
                    if constexpr (kc == 1) {
                        // every wave now holds all it needs of slot g & 1 in registers, and its own pieces of
                        // tap g + 1 have landed (vmcnt(0) inside __syncthreads): refill the slot with tap g + 2
                        // (hipcc does not wait for LDS-DMA traffic at a barrier by itself)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
                        dma(g + 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- MFMAs of this chunk, with the next chunk's fragment loads in between ----
                    constexpr int NLOAD = (4 + RTW) * NP;
                    constexpr int NMFMA = 4 * RTW * F::NPROD;
                    constexpr int SPAN = NMFMA * 3 / 4;                    // loads go out during the first 3/4
                    const bool next_b = kc == 0 || tap + 1 < ntaps;         // no activations beyond the layer yet
                    const bool next_a = kc == 0 || g + 1 < kSplitTaps;
                    static_for<NMFMA>([&](auto M_) {
                        constexpr int m = decltype(M_)::value;
                        constexpr int q = m / (4 * RTW), c = (m / RTW) % 4, r = m % RTW;
                        acc[F::PC[q]][c][r] = mfma16<F>(fa[kc][c][F::PA[q]], fb[kc][r][F::PB[q]], acc[F::PC[q]][c][r]);
                        constexpr int j0 = m * NLOAD / SPAN, j1 = (m + 1) * NLOAD / SPAN;
                        if constexpr (j1 > j0 && j0 < NLOAD) {
                            constexpr int j = j0;                          // load j of the next chunk
                            if constexpr (j < 4 * NP) {                    // weight fragment (ct = j % 4, piece = j / 4)
                                constexpr int c2 = j % 4, p2 = j / 4;
                                if constexpr (kc == 0) lds_load_frag<((NP + p2) * 4 + c2) * 1024>(fa[1][c2][p2], wa_cur);
                                else if (next_a) lds_load_frag<(p2 * 4 + c2) * 1024>(fa[0][c2][p2], wa_nxt);
                            } else {                                       // activation fragment
                                constexpr int r2 = (j - 4 * NP) % RTW, p2 = (j - 4 * NP) / RTW;
                                if constexpr (kc == 0)
                                    load_b(fb[1][r2][p2], std::integral_constant<int, p2>{}, std::integral_constant<int, 1>{}, ba[r2]);
                                else if (next_b)
                                    load_b(fb[0][r2][p2], std::integral_constant<int, p2>{}, std::integral_constant<int, 0>{}, bn[r2]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    __builtin_amdgcn_sched_barrier(0);
                };
                chunk(std::integral_constant<int, 0>{});
                chunk(std::integral_constant<int, 1>{});
            }
            // ---- epilogue: BN scale/shift (+ residual) + ReLU, split, overwrite the activation images ----
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();                              // every wave is done reading the layer input
            const bool block_out = (layer & 1) == 0;       // stem (0) and every conv2 (2, 4, .., 12)
            const bool add_res = block_out && layer > 0;
            const bool last = layer == kTowerLayers;
            float amax = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(bn_scale + layer * 64 + c * 16 + lg * 4);
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(net.shift + layer * 64 + c * 16 + lg * 4);
#pragma unroll
                for (int r = 0; r < RTW; ++r) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = acc[0][c][r][j];
                        if constexpr (F::NACC == 2) t = fmaf(acc[1][c][r][j], 1.f / 2048.f, t);
                        t = fmaf(t, sc[j], sh[j]);
                        if (add_res) t += res[c][r][j];
                        v[j] = fmaxf(t, 0.f);
                        amax = fmaxf(amax, v[j]);
                    }
                    if (block_out) res[c][r] = v;
                    const int row = base_row[r];
                    if (row < M) {
                        if (last) {
                            *reinterpret_cast<f32x4 *>(smem + row * kRowBytes + (c * 16 + lg * 4) * 4) = v;
                        } else {
                            uint2 pc[NP];
                            split4<F>(v, pc);
                            const int slot = (((c & 1) << 1) | (lg >> 1)) ^ ((row >> 1) & 3);
                            const int off = row * 64 + slot * 16 + (lg & 1) * 8;
#pragma unroll
                            for (int q = 0; q < NP; ++q)
                                *reinterpret_cast<uint2 *>(smem + (q * 2 + (c >> 1)) * IMG + off) = pc[q];
                        }
                    }
                }
            }
            if (!(amax < 60000.f)) ovf = 1;                // f16 range guard (also catches NaN)
            __syncthreads();
        }
        run_heads<S, G, C, NTHR>(smem, net, b0, batch, want_logits, policy, value, tid);
        __syncthreads();
        // the head scratch overlapped the activation images' zero rows
        for (int e = tid; e < NP * 2 * 16; e += NTHR)
            reinterpret_cast<unsigned *>(smem + (e >> 4) * IMG + M * 64)[e & 15] = 0u;
    }
    if (ovf && overflow) atomicOr(overflow, 1);
}

// ---- host: operand splitting of the weights -----------------------------------------------------
inline uint16_t f32_to_f16_rn(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t man = x & 0x7FFFFFu;
    if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - e;                              // 14 .. 24
        uint32_t h = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;     // may carry into the exponent: still right
    return (uint16_t)(sign | h);
}
inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1F, man = h & 0x3FFu;
    float out;
    if (e == 0) {
        out = std::ldexp((float)man, -24);
        uint32_t b;
        std::memcpy(&b, &out, 4);
        b |= sign;
        std::memcpy(&out, &b, 4);
        return out;
    }
    const uint32_t b = sign | ((e == 31 ? 0xFFu : e - 15 + 127) << 23) | (man << 13);
    std::memcpy(&out, &b, 4);
    return out;
}

// pieces of w (already scaled)
inline void split_weight(float w, uint16_t *out) {
    const uint16_t h = f32_to_f16_rn(w);
    out[0] = h;
    out[1] = f32_to_f16_rn((w - f16_to_f32(h)) * 2048.f);
}

template <int S, int G, typename F>
int launch_split(tg_net *net, const float *planes, int batch, int want_logits, float *policy, float *value,
                 int *overflow, hipStream_t stream) {
    using C = SplitCfg<S, G, F>;
    auto kern = dualnet_fwd_split_kernel<S, G, F>;
    static bool attr_set[16] = {};
    if (!attr_set[net->device & 15]) {
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set[net->device & 15] = true;
    }
    const int groups = (batch + G - 1) / G;
    const int grid = groups < net->num_cus ? groups : net->num_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHR), C::LDS_BYTES, stream, net->dev, planes, batch, want_logits,
                       policy, value, overflow);
    TG_HIP(hipGetLastError());
    return TG_OK;
}

}  // namespace

namespace tg {

// Build the split weight image.  conv0: [64][6][3][3]; tower[l]: [64][64][3][3];
// scale: folded BN scales [13][64] (the per-layer weight scaling is divided out of the copy used here).
int split_prepare(tg_net *net, const float *conv0, const float *const *tower, const float *scale) {
    {
        constexpr int np = FmtF16::NP;
        const size_t tap_bytes = (size_t)2 * np * 4 * 1024;
        std::vector<uint16_t> img((size_t)kSplitTaps * tap_bytes / 2, 0);
        std::vector<float> sscale(13 * 64);
        for (int layer = 0; layer <= kTowerLayers; ++layer) {
            // power-of-two pre-scaling (f16 only): largest weight of the layer into [2^9, 2^10)
            const float *w = layer == 0 ? conv0 : tower[layer - 1];
            const size_t n = layer == 0 ? (size_t)64 * 6 * 9 : (size_t)64 * 64 * 9;
            float mx = 0.f;
            for (size_t i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(w[i]));
            int e = 0;
            if (mx > 0.f && std::isfinite(mx)) {
                int ex;
                std::frexp(mx, &ex);                           // mx = f * 2^ex, f in [0.5, 1)
                e = 10 - ex;
            }
            const float up = std::ldexp(1.f, e), down = std::ldexp(1.f, -e);
            for (int i = 0; i < 64; ++i) sscale[layer * 64 + i] = scale[layer * 64 + i] * down;
            const int ntaps = layer == 0 ? 1 : 9;
            for (int tap = 0; tap < ntaps; ++tap) {
                const int g = layer == 0 ? 0 : 1 + (layer - 1) * 9 + tap;
                for (int kc = 0; kc < 2; ++kc)
                    for (int ct = 0; ct < 4; ++ct)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int el = 0; el < 8; ++el) {
                                const int cout = ct * 16 + (lane & 15), k = kc * 32 + (lane >> 4) * 8 + el;
                                float v;
                                if (layer == 0) {                  // k = tap' * 6 + plane
                                    const int t2 = k / 6, c2 = k % 6;
                                    v = k < 54 ? conv0[(cout * 6 + c2) * 9 + t2] : 0.f;
                                } else {
                                    v = w[((size_t)cout * 64 + k) * 9 + tap];
                                }
                                uint16_t pc[2];
                                split_weight(v * up, pc);
                                for (int p = 0; p < np; ++p)
                                    img[((((size_t)g * 2 + kc) * np + p) * 4 + ct) * 512 + lane * 8 + el] = pc[p];
                            }
            }
        }
        void *d = nullptr;
        TG_HIP(hipMalloc(&d, img.size() * 2));
        net->allocs.push_back(d);
        TG_HIP(hipMemcpy(d, img.data(), img.size() * 2, hipMemcpyHostToDevice));
        net->dev.wsplit = static_cast<const unsigned char *>(d);
        void *ds = nullptr;
        TG_HIP(hipMalloc(&ds, sscale.size() * 4));
        net->allocs.push_back(ds);
        TG_HIP(hipMemcpy(ds, sscale.data(), sscale.size() * 4, hipMemcpyHostToDevice));
        net->dev.sscale = static_cast<const float *>(ds);
    }
    return TG_OK;
}

// group = boards per workgroup (1 or 3); 9x9 only.
int split_forward(tg_net *net, int group, const float *planes, int batch, int want_logits, float *policy,
                  float *value, int *overflow, hipStream_t stream) {
    if (net->board_size != 9) return tg::fail(TG_ERR_ARG, "split forward: 9x9 only");
    if (group == 3) return launch_split<9, 3, FmtF16>(net, planes, batch, want_logits, policy, value, overflow, stream);
    return launch_split<9, 1, FmtF16>(net, planes, batch, want_logits, policy, value, overflow, stream);
}

}  // namespace tg
