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
//   * weights do not pass through LDS at all: the host lays them out in MFMA A-fragment order
//     [tap][k-chunk][piece][channel tile][lane][16 B] and every wave fetches its fragments with plain
//     coalesced 16-byte global loads (1 KB per wave instruction) from the L2-resident image (1.8 MB) - the
//     vector-memory path is idle otherwise, and LDS bandwidth is the scarce resource of this loop (a first
//     version streamed them through an LDS ring with global_load_lds: 15 % slower, one barrier per tap);
//   * fragment loads are issued between the MFMAs of earlier k-chunks (activations one chunk ahead, weights
//     two), their program positions pinned with sched_barrier; the chunk sequence of a layer is straight-line
//     code, so hipcc's counted waits land exactly where the fragments are first used.
#include "net_device.h"

#include <cstdlib>
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
    // 19x19 (one board per workgroup, 23 row-tiles): four waves of six row-tiles; the residual image (93 KB)
    // does not fit into LDS next to the activation images and lives in an L2-resident scratch image per workgroup;
    // the policy FC weights (1 MB) are read from L2 by the generic head code
    static constexpr bool BIG = S > 9;
    static constexpr int RTW = BIG ? 6 : (G == 3 ? 4 : 2);        // row-tiles per wave
    static constexpr int NW = (MT + RTW - 1) / RTW;               // waves per workgroup
    static constexpr int NTHR = NW * 64;
    static constexpr int IMG = (M + 2) * 64;                      // one [row][64 B] image + zero row M + dump row M + 1
                                                                  // (rows >= M of the last row-tile store there: no branches)
    static constexpr int ACT_BYTES = F::NP * 2 * IMG;             // image index = piece * 2 + kc
    static constexpr int CHUNK = F::NP * 4 * 1024;                // weight image per k-chunk: [piece][ct][lane][16 B]
    static constexpr int STAGE = (ACT_BYTES + 255) & ~255;        // input planes [G][6][P] fp32 (group start only)
    // residual-block input X as fp32 [row][16 x 16 B], slot ^= row & 15 (conflict-free 16-byte accesses);
    // written by the stem / conv2 epilogues, added back two layers later.  Keeping it out of the register
    // file (64 registers) is what lets accumulators + five fragment sets stay put.
    // (behind the fp32 image the LAST epilogue writes for the heads, which must not run into residuals that
    // other waves have yet to read).  During the head phase the same region receives the policy FC weights.
    static constexpr int STAGE_END = STAGE + ((G * 6 * P * 4 + 255) & ~255);
    static constexpr int HEAD_IMG = ((M + 1) * kRowBytes + 255) & ~255;
    static constexpr int RES_OFF = STAGE_END > HEAD_IMG ? STAGE_END : HEAD_IMG;
    static constexpr int FC_BYTES = ((2 * P * A * 4 + 4095) / 4096) * 4096;          // policy FC, [2P][A] fp32, padded
    // (19x19: the residual of the first RES_LDS_TILES row-tiles of every wave stays in LDS - what the 160 KB still
    // hold -, the rest goes through the global scratch image)
    static constexpr int RES_LDS_TILES = BIG ? 2 : 0;
    static constexpr int RES_BYTES = BIG ? NW * RES_LDS_TILES * 16 * 256 : ((M + 1) * 256 > FC_BYTES ? (M + 1) * 256 : FC_BYTES);
    static constexpr int RES_ROWS = M + 2;                        // rows of the global residual image (BIG)
    static constexpr int SS_OFF = RES_OFF + RES_BYTES;           // folded BN scale [13][64] + shift [13][64]
    // head tables, staged once per workgroup: 1x1 weights [64][4] (policy 0, policy 1, value, 0), policy FC bias [A]
    // (padded), BN scale / shift of the three head channels [8].  (In the heads every use of a kernel-argument
    // pointer was a reload from scratch followed by a dependent global load.)
    static constexpr int HW_OFF = SS_OFF + 2 * 13 * 64 * 4;
    static constexpr int HB_OFF = HW_OFF + 64 * 4 * 4;
    static constexpr int HS_OFF = HB_OFF + ((A + 3) & ~3) * 4;
    static constexpr int VW_OFF = HS_OFF + 8 * 4;                 // value FC weights [3][P] + bias [3]
    static constexpr int PIPE_BYTES = VW_OFF + ((3 * P + 3 + 3) & ~3) * 4;
    // head phase (after the last layer): fp32 activations [M][72 floats] from offset 0, scratch behind the BN table
    static constexpr int ROW_BYTES = kRowBytes;
    static constexpr int AUX = PIPE_BYTES;
    static constexpr int LDS_BYTES = AUX + G * (3 * P + A + 4) * 4 + 256;
};

// Fragment loads are PLAIN loads: hipcc tracks lgkmcnt / vmcnt for them exactly in straight-line code (counted
// waits at the first use, one / two k-chunks later) and knows which registers are still in flight.  (A first
// version issued them as inline asm with hand-placed s_waitcnt: hipcc then treats the destination as written at
// once, and under register pressure it copied or re-used such registers before the data had landed.)
template <int OFFSET>
__device__ __forceinline__ void lds_load_frag(i32x4v &dst, const unsigned char *smem, int addr) {
    dst = *reinterpret_cast<const i32x4v *>(smem + addr + OFFSET);
}

// weight fragment straight from the L2-resident image (1 KB per wave, coalesced)
__device__ __forceinline__ void gmem_load_frag(i32x4v &dst, const unsigned char *base, int byte_off) {
    dst = *reinterpret_cast<const i32x4v *>(base + byte_off);
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


// Heads (1x1 convolutions + BN + ReLU on the fp32 image the last epilogue left at offset 0, the two fully
// connected layers, softmax) - run_heads of net_device.h with the policy FC weights (53 KB, the one large
// operand: 162 dependent L2 round trips per output when read from global memory) served from LDS: they are
// copied with global_load_lds into the residual region, which is idle by now, while the 1x1 convolutions run.
template <int S, int G, typename C, int NTHR>
__device__ __forceinline__ void run_heads_split(unsigned char *smem, const NetDev &net, int b0, int batch, int want_logits,
                                                float *__restrict__ policy, float *__restrict__ value, int tid, int wave,
                                                long long *tl) {
    constexpr int P = C::P, A = C::A, M = C::M;
    asm volatile("" : "+v"(tid));                          // opaque: nothing derived from it below is hoisted out of
    const int lane = tid & 63;                             // the caller's group loop (and spilled there)
    auto stamp = [&](int i) { if (tl && tid == 0) tl[i] = (long long)__builtin_amdgcn_s_memtime(); };
    if constexpr (!C::BIG) {
        constexpr int PIECES = C::FC_BYTES / 1024;
        const unsigned char *src = reinterpret_cast<const unsigned char *>(net.pfc_wT) + lane * 16;
#pragma unroll 1
        for (int piece = wave; piece < PIECES; piece += NTHR / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 1024),
                                             (__attribute__((address_space(3))) void *)(smem + C::RES_OFF + piece * 1024), 16, 0, 0);
    }
    float *hpol = reinterpret_cast<float *>(smem + C::AUX);   // [G][2P]
    float *hval = hpol + G * 2 * P;                           // [G][P]
    float *plog = hval + G * P;                               // [G][A]
    float *vlog = plog + G * A;                               // [G][4]
    float *plog_part = reinterpret_cast<float *>(smem);       // [waves][G][A] partial FC sums: over the fp32 feature
                                                              // image, which nobody reads after the 1x1 convolutions
    const int li = lane & 15, lg = lane >> 4;
    constexpr int NW = NTHR / 64;
    {
        // 1x1 convolutions (64 -> 2 policy + 1 value channels) on the fp32 matrix pipe: rows = positions, columns =
        // the three head channels (13 of 16 columns idle - still 5x faster than one row per thread on the VALU, whose
        // 16-byte reads of consecutive rows collide in LDS).  k-step ks of lane group lg covers channel 16 lg + ks:
        // a lane reads its 16 channels as four 16-byte loads (rows 16 apart in the [row][72] image: conflict-free).
        const float *hw = reinterpret_cast<const float *>(smem + C::HW_OFF);
        const float *hs = reinterpret_cast<const float *>(smem + C::HS_OFF);
        const int col = li < 3 ? li : 3;                   // column 3 of the table is zero
        float wB[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wB[ks] = hw[(lg * 16 + ks) * 4 + col];
        const float sc = hs[2 * (li < 3 ? li : 0)], sh = hs[2 * (li < 3 ? li : 0) + 1];
        // a wave's tiles (t = wave, wave + NW, ...) together: their accumulator chains are independent, so the
        // dependent-issue latency of one hides behind the others
        constexpr int TPW = (C::MT + NW - 1) / NW;
        f32x4 xa[TPW][4], acc[TPW];
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int t = wave + q * NW;
            const int row = (t < C::MT ? t : wave) * 16 + li;
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[q][j] = lds_f32x4(smem, row * C::ROW_BYTES + lg * 64 + j * 16);
            acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int q = 0; q < TPW; ++q)
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[q][ks >> 2][ks & 3], wB[ks], acc[q], 0, 0, 0);
        if (li < 3) {
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int t = wave + q * NW;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = t * 16 + lg * 4 + v;
                    if (t < C::MT && r < M) {
                        const int bl = r / P, pp = r - bl * P;
                        const float o = fmaxf(fmaf(acc[q][v], sc, sh), 0.f);
                        if (li == 2) hval[bl * P + pp] = o;
                        else hpol[bl * 2 * P + li * P + pp] = o;
                    }
                }
            }
        }
    }
    stamp(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of the FC weights have landed
    __syncthreads();
    stamp(1);
    const float *fcw = reinterpret_cast<const float *>(smem + C::RES_OFF);
    if constexpr (C::BIG) {
        // 19x19 policy FC: 2P x A = 1 MB of weights per board, streamed from L2 exactly once - every wave takes a quarter
        // of K for ALL outputs (six per lane, coalesced rows of the transposed weight matrix), eight k-rows = 48
        // independent loads in flight; the partial sums meet in LDS like the small boards'.  (One output per thread
        // with four partial sums kept four loads in flight: 130 us per board, 40 % of the kernel.)
        static_assert(G == 1, "one board per workgroup");
        constexpr int K = 2 * P, KQ = (K + NW - 1) / NW;
        constexpr int OPL = (A + 63) / 64;                 // outputs per lane
        const int k0 = wave * KQ, k1 = k0 + KQ < K ? k0 + KQ : K;
        float accf[OPL];
#pragma unroll
        for (int i = 0; i < OPL; ++i) accf[i] = 0.f;
        const float *wrow = net.pfc_wT + lane;
        for (int k = k0; k < k1; k += 8) {
            float w[8][OPL], h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k + u < k1 ? k + u : k1 - 1;
                h[u] = k + u < k1 ? hpol[kk] : 0.f;
#pragma unroll
                for (int i = 0; i < OPL; ++i) {
                    const int a = lane + 64 * i;
                    w[u][i] = wrow[(size_t)kk * A + (a < A ? 64 * i : 0)];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < OPL; ++i) accf[i] = fmaf(h[u], w[u][i], accf[i]);
        }
#pragma unroll
        for (int i = 0; i < OPL; ++i) {
            const int a = lane + 64 * i;
            if (a < A) plog_part[wave * A + a] = accf[i];
        }
    } else {
        // policy FC on the fp32 matrix pipe: rows = the G boards (13+ of 16 rows idle), columns = 16 of the A
        // outputs per tile, K = 2P inputs in 41 steps of 4 with k = 41 lg + ks (contiguous per lane group; k >= 2P
        // masked).  One thread per output on the VALU needed 2 x 162 LDS reads per output: 13 k cycles.
        constexpr int KS = (2 * P + 3) / 4;                 // 41 k-steps
        constexpr int CT = (A + 15) / 16;                   // 6 column tiles
        constexpr int KW = (KS + NW - 1) / NW;              // k-steps per wave: every wave takes a slice of K for ALL
        f32x4 acc[CT];                                      // column tiles (six independent accumulators, equal work);
#pragma unroll                                              // the partial sums meet in LDS (plog_part) below
        for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            const int ks = wave * KW + kk;
            const int k = lg * KS + ks;
            const bool kin = ks < KS && k < 2 * P;
            const int kc = kin ? k : 0;
            const float hv = (li < G && kin) ? hpol[li * 2 * P + kc] : 0.f;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int a = ct * 16 + li;
                const float wv = kin ? fcw[kc * A + (a < A ? a : A - 1)] : 0.f;
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv, wv, acc[ct], 0, 0, 0);
            }
        }
        if (lg == 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int a = ct * 16 + li;
                if (a < A) {
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (v < G) plog_part[(wave * G + v) * A + a] = acc[ct][v];
                }
            }
        }
    }
    // value FC: sixteen lanes per output, strided partial sums, butterfly over the 16 lanes
    for (int o = tid >> 4; o < G * 3; o += NTHR / 16) {
        const int part = tid & 15, bl = o / 3, c = o - bl * 3;
        const float *h = hval + bl * P;
        const float *wv = reinterpret_cast<const float *>(smem + C::VW_OFF) + c * P;
        float sv = 0.f;
#pragma unroll
        for (int i = 0; i < (P + 15) / 16; ++i) {
            const int j = part + i * 16;
            if (j < P) sv = fmaf(h[j], wv[j], sv);
        }
        sv += __shfl_xor(sv, 8);
        sv += __shfl_xor(sv, 4);
        sv += __shfl_xor(sv, 2);
        sv += __shfl_xor(sv, 1);
        if (part == 0) vlog[bl * 4 + c] = sv + reinterpret_cast<const float *>(smem + C::VW_OFF)[3 * P + c];
    }
    stamp(2);
    __syncthreads();
    for (int bl = wave; bl < G; bl += NTHR / 64) {
        const int b = b0 + bl;
        if (b >= batch) continue;
        float m = -INFINITY;
        for (int a = lane; a < A; a += 64) {               // partial FC sums of the waves + bias (a lane re-reads only
            float lgt = reinterpret_cast<const float *>(smem + C::HB_OFF)[a];     // the entries it writes here)
#pragma unroll
            for (int w = 0; w < NTHR / 64; ++w) lgt += plog_part[(w * G + bl) * A + a];
            plog[bl * A + a] = lgt;
            m = fmaxf(m, lgt);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
        for (int a = lane; a < A; a += 64) sum += expf(plog[bl * A + a] - m);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.f / sum;
        for (int a = lane; a < A; a += 64) {
            const float lg_ = plog[bl * A + a];
            __builtin_nontemporal_store(want_logits ? lg_ : expf(lg_ - m) * inv, &policy[(size_t)b * A + a]);
        }
        if (lane < 3) {
            const float v0 = vlog[bl * 4], v1 = vlog[bl * 4 + 1], v2 = vlog[bl * 4 + 2];
            const float vm = fmaxf(v0, fmaxf(v1, v2));
            const float e0 = expf(v0 - vm), e1 = expf(v1 - vm), e2 = expf(v2 - vm);
            const float es = e0 + e1 + e2;
            const float mine = lane == 0 ? e0 : (lane == 1 ? e1 : e2);
            value[(size_t)b * 3 + lane] = mine / es;
        }
    }
}

template <int S, int G, typename F, int SPANQ = 6>
__global__ __launch_bounds__((SplitCfg<S, G, F>::NTHR), 1) void dualnet_fwd_split_kernel(
    NetDev net, const float *__restrict__ planes, int batch, int want_logits,
    float *__restrict__ policy, float *__restrict__ value, int *__restrict__ overflow) {
    using C = SplitCfg<S, G, F>;
    // 19x19: this workgroup's residual image [row][64] fp32 in the per-stream scratch (every lane re-reads only what
    // it wrote itself two layers earlier: no fence needed)
    float *const resg = C::BIG ? net.scratch + (size_t)blockIdx.x * C::RES_ROWS * 64 : nullptr;
    constexpr int P = C::P, M = C::M, RTW = C::RTW, NTHR = C::NTHR, NP = F::NP, IMG = C::IMG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: scalar branches around the DMA
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;

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

    // folded BN scale / shift of all 13 layers -> LDS, once (the epilogues would otherwise wait for L2 every layer)
    for (int e = tid; e < 13 * 64; e += NTHR) {
        reinterpret_cast<float *>(smem + C::SS_OFF)[e] = net.sscale[e];
        reinterpret_cast<float *>(smem + C::SS_OFF)[13 * 64 + e] = net.shift[e];
    }
    for (int e = tid; e < 64 * 4; e += NTHR) {
        const int k = e >> 2, c = e & 3;
        reinterpret_cast<float *>(smem + C::HW_OFF)[e] = c == 0 ? net.hp_w[k] : (c == 1 ? net.hp_w[64 + k] : (c == 2 ? net.hv_w[k] : 0.f));
    }
    for (int e = tid; e < C::A; e += NTHR) reinterpret_cast<float *>(smem + C::HB_OFF)[e] = net.pfc_b[e];
    if (tid < 6) reinterpret_cast<float *>(smem + C::HS_OFF)[tid] = net.head_ss[tid];
    for (int e = tid; e < 3 * P + 3; e += NTHR)
        reinterpret_cast<float *>(smem + C::VW_OFF)[e] = e < 3 * P ? net.vfc_w[e] : net.vfc_b[e - 3 * P];
    // weight stream: k-chunk gc = 2 * tap + kc of the whole network lies at wsplit + gc * CHUNK; a chunk's eight
    // fragments are at lane * 16 + (piece * 4 + ct) * 1024 (two lane offsets cover the 4 KB offset field)
    const int wv0 = lane * 16;
    constexpr int kChunks = 2 * kSplitTaps;

    // profiling stamps (tg_net_profile_phases): workgroup 0, wave 0: 0 group start, 1 input staged + split,
    // then per layer (stem first) "MFMA loop done" / "epilogue done", last: heads done
    int stamp_i = 0;
    auto stamp = [&]() {
        if (net.timeline && blockIdx.x == 0 && tid == 0 && stamp_i < 40)      // (40..42: head sub-phases)
            net.timeline[stamp_i++] = (long long)__builtin_amdgcn_s_memtime();
    };
    int ovf = 0;
    const int n_groups = (batch + G - 1) / G;
    constexpr int NPL = (G * 6 * P + NTHR - 1) / NTHR;
    float pre[NPL];
    auto fetch_planes = [&](int grp2) __attribute__((always_inline)) {
        int ft = tid;                                      // opaque: keeps the per-lane offsets and predicates from
        asm volatile("" : "+v"(ft));                       // being hoisted out of the group loop and spilled
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int e = ft + i * NTHR;
            const int b = grp2 * G + e / (6 * P);
            pre[i] = (e < G * 6 * P && grp2 < n_groups && b < batch)
                         ? __builtin_nontemporal_load(&planes[(size_t)grp2 * G * 6 * P + e]) : 0.f;
        }
    };
    fetch_planes(blockIdx.x);
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int b0 = grp * G;
        stamp();
        // ---- input planes (fetched into registers during the previous group's head phase) -> LDS -> im2col'ed, split "layer -1"
        //      activations: K = 9 taps x 6 planes (k = 6 tap + plane), padded to 64 ----
        {
            float *st = reinterpret_cast<float *>(smem + C::STAGE);
            int stid = tid;                                 // opaque, as above
            asm volatile("" : "+v"(stid));
#pragma unroll
            for (int i = 0; i < NPL; ++i)
                if (stid + i * NTHR < G * 6 * P) st[stid + i * NTHR] = pre[i];
            __syncthreads();
            for (int row = stid; row < M; row += NTHR) {     // one thread per position (19x19: two passes)
                const int bl = row / P, p = row - bl * P, y = p / S, x = p - y * S;
                const float *src = st + bl * 6 * P + p;
                const int swz = (row >> 1) & 3;
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {            // slot sl holds k = 8 sl .. 8 sl + 7
                    f32x4 lo, hi;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int k = sl * 8 + j, t = k / 6, c = k - t * 6;
                        const int dy = t / 3 - 1, dx = t % 3 - 1;
                        const bool ok = k < 54 && (unsigned)(y + dy) < (unsigned)S && (unsigned)(x + dx) < (unsigned)S;
                        const float v = ok ? src[c * P + dy * S + dx] : 0.f;
                        if (j < 4) lo[j] = v; else hi[j - 4] = v;
                    }
                    uint2 plo[NP], phi[NP];
                    split4<F>(lo, plo);
                    split4<F>(hi, phi);
                    const int kc = sl >> 2, slot = (sl & 3) ^ swz;
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        *reinterpret_cast<uint4 *>(smem + (q * 2 + kc) * IMG + row * 64 + slot * 16) =
                            uint4{plo[q].x, plo[q].y, phi[q].x, phi[q].y};
                }
            }
        }
        __syncthreads();                                  // activations written
        stamp();

        f32x4 acc[F::NACC][4][RTW];
        // fragment sets: weights (A operand) three deep - requested TWO k-chunks ahead from L2 -, activations
        // (B operand) two deep - requested one chunk ahead from LDS
        // (19x19, six row-tiles per wave: two weight sets, one chunk ahead - a third set does not fit the register file
        // next to 192 accumulators and 96 activation-fragment registers, and a chunk is 1.5x as long there)
        constexpr int NASET = C::BIG ? 2 : 3, ADIST = NASET - 1;
        i32x4v fa[NASET][4][NP], fb[2][RTW][NP];

        // B-fragment base address of row-tile r for tap `tap` of layer kind `stem`
        auto row_addr = [&](int r, int tap, bool stem) __attribute__((always_inline)) {
            const int toff = stem ? 0 : (tap / 3 - 1) * S + (tap % 3 - 1);
            const bool ok = stem ? base_row[r] < M : ((mask[r] >> tap) & 1u) != 0;
            const int row = ok ? base_row[r] + toff : M;
            return row * 64 + ((lg ^ ((row >> 1) & 3)) << 4);
        };
        auto load_b = [&](i32x4v &dst, auto P_, auto KC_, int addr) __attribute__((always_inline)) {
            constexpr int off = (decltype(P_)::value * 2 + decltype(KC_)::value) * IMG;   // image (p, kc)
            lds_load_frag<off>(dst, smem, addr);
        };
        // all eight weight fragments of chunk gc into set SET
        // (opaque lane offset: with a constant chunk index these sixteen addresses are invariant across the groups,
        // were hoisted out of the group loop as 64-bit pointers, spilled, and each load then waited for its own
        // pointer to come back from scratch)
        int wvg = wv0;
        asm volatile("" : "+v"(wvg));
        auto load_a_all = [&](auto SET_, int gc) __attribute__((always_inline)) {
            constexpr int set = decltype(SET_)::value;
            const unsigned char *base = net.wsplit + (size_t)(gc < kChunks ? gc : kChunks - 1) * C::CHUNK;
            static_for<4 * NP>([&](auto J) {
                constexpr int j = decltype(J)::value, c = j % 4, p = j / 4;
                gmem_load_frag(fa[set][c][p], base, wvg + (p * 4 + c) * 1024);
            });
        };
        // three sets: chunk gc uses weight set (gc + 1) % 3 - the stem's two chunks take sets 1 and 2, every tower
        // layer (18 chunks) starts at set 0; two sets: chunk gc uses set gc % 2
        if constexpr (NASET == 3) {
            load_a_all(std::integral_constant<int, 1>{}, 0);
            load_a_all(std::integral_constant<int, 2>{}, 1);
        } else {
            load_a_all(std::integral_constant<int, 0>{}, 0);
        }

        // One k-chunk: wait for its fragments, 4 * RTW * NPROD MFMAs; in between, the activation fragments of the
        // next chunk (LDS, set 1 - BSET) and the weight fragments of the chunk after next (L2, the set this
        // chunk's predecessor used).  ba: this tap's row addresses (for a KC = 0 chunk's successor), bn: the
        // next tap's.
        auto chunk = [&](auto KC_, auto ASET_, int gc, const int (&ba)[RTW], const int (&bn)[RTW]) __attribute__((always_inline)) {
            constexpr int kc = decltype(KC_)::value, aset = decltype(ASET_)::value % NASET, anext = (aset + ADIST) % NASET;
            // (this chunk's fragments went out one / two chunks ago; hipcc places the counted waits)
            const unsigned char *wnext = net.wsplit + (size_t)(gc + ADIST < kChunks ? gc + ADIST : kChunks - 1) * C::CHUNK;
            constexpr int NMFMA = 4 * RTW * F::NPROD;
            constexpr int NB = RTW * NP, NA = 4 * NP;
            constexpr int BSPAN = NMFMA * SPANQ / 16;      // activation loads: during the first SPANQ/16 of the chunk
            static_for<NMFMA>([&](auto M_) {
                constexpr int m = decltype(M_)::value;
                constexpr int q = m / (4 * RTW), c = (m / RTW) % 4, r = m % RTW;
                acc[F::PC[q]][c][r] = mfma16<F>(fa[aset][c][F::PA[q]], fb[kc][r][F::PB[q]], acc[F::PC[q]][c][r]);
                // activation fragments of the next chunk (uniformly spread over the first BSPAN MFMAs)
                constexpr int jb0 = m * NB / BSPAN, jb1 = (m + 1) * NB / BSPAN < NB ? (m + 1) * NB / BSPAN : NB;
                if constexpr (jb1 > jb0) {
                    static_for<jb1 - jb0>([&](auto D_) {
                        constexpr int jb = jb0 + decltype(D_)::value, r2 = jb % RTW, p2 = jb / RTW;
                        if constexpr (kc == 0) load_b(fb[1][r2][p2], std::integral_constant<int, p2>{}, std::integral_constant<int, 1>{}, ba[r2]);
                        else load_b(fb[0][r2][p2], std::integral_constant<int, p2>{}, std::integral_constant<int, 0>{}, bn[r2]);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
                // weight fragment j of the chunk after next (spread over the whole chunk)
                constexpr int ja0 = m * NA / NMFMA, ja1 = (m + 1) * NA / NMFMA;
                if constexpr (ja1 > ja0) {
                    constexpr int c2 = ja0 % 4, p2 = ja0 / 4;
                    gmem_load_frag(fa[anext][c2][p2], wnext, wv0 + (p2 * 4 + c2) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;

        int gc = 0;                                       // k-chunk index over the whole network
#pragma unroll 1
        for (int layer = 0; layer <= kTowerLayers; ++layer) {
            const bool stem = layer == 0;
#pragma unroll
            for (int s = 0; s < F::NACC; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < RTW; ++r) acc[s][c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
            // activation fragments of the layer's first chunk: the previous epilogue just wrote them
            int ba[RTW], bn[RTW];
#pragma unroll
            for (int r = 0; r < RTW; ++r) ba[r] = row_addr(r, 0, stem);
            static_for<RTW * NP>([&](auto J) {
                constexpr int r = decltype(J)::value % RTW, p = decltype(J)::value / RTW;
                load_b(fb[0][r][p], std::integral_constant<int, p>{}, I0{}, ba[r]);
            });
            if (stem) {
                chunk(I0{}, std::integral_constant<int, NASET == 3 ? 1 : 0>{}, gc, ba, ba);
                chunk(I1{}, std::integral_constant<int, NASET == 3 ? 2 : 1>{}, gc + 1, ba, ba);
                gc += 2;
            } else {
#pragma unroll 1
                for (int t3 = 0; t3 < 9; t3 += 3) {        // three taps = six chunks = two turns of the weight sets
#pragma unroll
                    for (int r = 0; r < RTW; ++r) bn[r] = row_addr(r, t3 + 1, false);
                    chunk(I0{}, I0{}, gc, ba, bn);
                    chunk(I1{}, I1{}, gc + 1, ba, bn);
#pragma unroll
                    for (int r = 0; r < RTW; ++r) { ba[r] = bn[r]; bn[r] = row_addr(r, t3 + 2, false); }
                    chunk(I0{}, I2{}, gc + 2, ba, bn);                      // (sets taken modulo the number of sets)
                    chunk(I1{}, std::integral_constant<int, 3>{}, gc + 3, ba, bn);
#pragma unroll
                    for (int r = 0; r < RTW; ++r) { ba[r] = bn[r]; bn[r] = row_addr(r, t3 + 3 < 9 ? t3 + 3 : 8, false); }
                    chunk(I0{}, std::integral_constant<int, 4>{}, gc + 4, ba, bn);
                    chunk(I1{}, std::integral_constant<int, 5>{}, gc + 5, ba, bn);
#pragma unroll
                    for (int r = 0; r < RTW; ++r) ba[r] = bn[r];
                    gc += 6;
                }
            }
            // ---- epilogue: BN scale/shift (+ residual) + ReLU, split, overwrite the activation images ----
            stamp();
            __syncthreads();                              // every wave is done reading the layer input
            float amax = 0.f;
            // three shapes, chosen once per layer (no per-tile branches): conv1 (plain), stem / conv2 (keep the
            // result as the next block's residual; conv2 adds the current one), last conv2 (fp32 image for the heads)
            auto epilogue = [&](auto KEEP_, auto ADD_, auto LAST_) __attribute__((always_inline)) {
                constexpr bool keep = decltype(KEEP_)::value, add_res = decltype(ADD_)::value, last = decltype(LAST_)::value;
                // every LDS operand first (one wave per SIMD: nothing else hides their latency)
                f32x4 xres[4][RTW], sc[4], sh[4];
                int brow[RTW], rrow[RTW], wrow[RTW];
#pragma unroll
                for (int r = 0; r < RTW; ++r) {
                    // opaque copy: the ~50 LDS addresses below are loop-invariant across the layers, hipcc hoisted
                    // them out of the layer loop, found no registers there and reloaded every one of them from
                    // scratch in every epilogue (the conv2 epilogue waited on ~30 scratch loads); recomputing them
                    // from the row costs a few VALU instructions
                    brow[r] = base_row[r];
                    asm volatile("" : "+v"(brow[r]));
                    rrow[r] = brow[r] < M ? brow[r] : 0;                  // rows >= M: read anything valid,
                    wrow[r] = brow[r] < M ? brow[r] : M + 1;              // store into the dump row
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    sc[c] = *reinterpret_cast<const f32x4 *>(smem + C::SS_OFF + (layer * 64 + c * 16 + lg * 4) * 4);
                    sh[c] = *reinterpret_cast<const f32x4 *>(smem + C::SS_OFF + (13 * 64 + layer * 64 + c * 16 + lg * 4) * 4);
                    if constexpr (add_res) {
#pragma unroll
                        for (int r = 0; r < RTW; ++r)
                            if constexpr (C::BIG) {
                                if (r < C::RES_LDS_TILES) {            // (r is a compile-time constant after unrolling)
                                    const int lrow = (wave * C::RES_LDS_TILES + r) * 16 + li;
                                    xres[c][r] = *reinterpret_cast<const f32x4 *>(smem + C::RES_OFF + lrow * 256 + (((c * 4 + lg) ^ (lrow & 15)) << 4));
                                } else {
                                    xres[c][r] = *reinterpret_cast<const f32x4 *>(resg + rrow[r] * 64 + c * 16 + lg * 4);
                                }
                            } else
                                xres[c][r] = *reinterpret_cast<const f32x4 *>(smem + C::RES_OFF + rrow[r] * 256 + (((c * 4 + lg) ^ (rrow[r] & 15)) << 4));
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int r = 0; r < RTW; ++r) {
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float t = acc[0][c][r][j];
                            if constexpr (F::NACC == 2) t = fmaf(acc[1][c][r][j], 1.f / 2048.f, t);
                            t = fmaf(t, sc[c][j], sh[c][j]);
                            if constexpr (add_res) t += xres[c][r][j];
                            v[j] = fmaxf(t, 0.f);
                        }
                        amax = fmaxf(fmaxf(amax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                        const int row = wrow[r];
                        if constexpr (last) {
                            const int hrow = brow[r] < M ? brow[r] : M;
                            *reinterpret_cast<f32x4 *>(smem + hrow * kRowBytes + (c * 16 + lg * 4) * 4) = v;
                        } else {
                            if constexpr (keep) {
                                if constexpr (C::BIG) {
                                    if (r < C::RES_LDS_TILES) {
                                        const int lrow = (wave * C::RES_LDS_TILES + r) * 16 + li;
                                        *reinterpret_cast<f32x4 *>(smem + C::RES_OFF + lrow * 256 + (((c * 4 + lg) ^ (lrow & 15)) << 4)) = v;
                                    } else {
                                        *reinterpret_cast<f32x4 *>(resg + (brow[r] < M ? brow[r] : M) * 64 + c * 16 + lg * 4) = v;
                                    }
                                } else
                                    *reinterpret_cast<f32x4 *>(smem + C::RES_OFF + (brow[r] < M ? brow[r] : M) * 256 +
                                                               (((c * 4 + lg) ^ (brow[r] & 15)) << 4)) = v;
                            }
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
            };
            using T = std::true_type;
            using N = std::false_type;
            if (layer == kTowerLayers) epilogue(N{}, T{}, T{});
            else if (layer == 0) epilogue(T{}, N{}, N{});
            else if (layer & 1) epilogue(N{}, N{}, N{});
            else epilogue(T{}, T{}, N{});
            if (!(amax < 60000.f)) ovf = 1;                // f16 range guard (also catches NaN)
            __syncthreads();
            stamp();
        }
        // next group's input planes: HBM latency, and vmcnt retires in order - requested here, where the only
        // wait behind them is the heads' own (the FC weight copy), not one of the tower's weight fragments
        fetch_planes(grp + gridDim.x);
        run_heads_split<S, G, C, NTHR>(smem, net, b0, batch, want_logits, policy, value, tid, wave,
                                       (net.timeline && blockIdx.x == 0 && grp == blockIdx.x) ? net.timeline + 40 : nullptr);
        __syncthreads();
        stamp();
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

template <int S, int G, typename F, int SPANQ = 6>
int launch_split(tg_net *net, const float *planes, int batch, int want_logits, float *policy, float *value,
                 int *overflow, hipStream_t stream) {
    using C = SplitCfg<S, G, F>;
    auto kern = dualnet_fwd_split_kernel<S, G, F, SPANQ>;
    static bool attr_set[16] = {};
    if (!attr_set[net->device & 15]) {
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set[net->device & 15] = true;
    }
    const int groups = (batch + G - 1) / G;
    const int grid = groups < net->num_cus ? groups : net->num_cus;
    NetDev dev = net->dev;
    if (C::BIG) {
        // residual images: the per-stream scratch the 19x19 Winograd kernel uses (two [P][64] images per workgroup -
        // more than the [P + 2][64] needed here); launches on one stream run in order
        static_assert(!C::BIG || C::RES_ROWS * 64 <= 2 * C::P * 64, "scratch image");
        std::lock_guard<std::mutex> lock(net->scratch_mu);
        float *&slot = net->scratch_by_stream[stream];
        if (!slot) {
            void *d = nullptr;
            TG_HIP(hipMalloc(&d, net->scratch_floats * sizeof(float)));
            slot = static_cast<float *>(d);
        }
        dev.scratch = slot;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHR), C::LDS_BYTES, stream, dev, planes, batch, want_logits,
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
    if (net->board_size == 19) return launch_split<19, 1, FmtF16>(net, planes, batch, want_logits, policy, value, overflow, stream);
    if (net->board_size != 9) return tg::fail(TG_ERR_ARG, "split forward: 9x9 and 19x19 only");
    if (group == 3) {
        if (const char *env = getenv("TG_SPLIT_SPAN")) {              // tuning knob
            const int q = atoi(env);
            if (q == 2) return launch_split<9, 3, FmtF16, 2>(net, planes, batch, want_logits, policy, value, overflow, stream);
            if (q == 3) return launch_split<9, 3, FmtF16, 3>(net, planes, batch, want_logits, policy, value, overflow, stream);
            if (q == 4) return launch_split<9, 3, FmtF16, 4>(net, planes, batch, want_logits, policy, value, overflow, stream);
            if (q == 8) return launch_split<9, 3, FmtF16, 8>(net, planes, batch, want_logits, policy, value, overflow, stream);
        }
        return launch_split<9, 3, FmtF16>(net, planes, batch, want_logits, policy, value, overflow, stream);
    }
    return launch_split<9, 1, FmtF16>(net, planes, batch, want_logits, policy, value, overflow, stream);
}

}  // namespace tg
