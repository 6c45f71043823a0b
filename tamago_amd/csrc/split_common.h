// Pieces shared by the split-operand forward kernels (net_forward_split.hip: 16x16x32 tiles, one wave per SIMD;
// net_forward_s32.hip: 32x32x16 tiles, two waves per SIMD): vector types, compile-time loops, the f16 x 2 operand
// split on the device and on the host, and the head phase.
#pragma once
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

// Head phase on the 16-bit matrix pipe (9x9; round 3).  The last tower epilogue writes the block output as the usual
// f16-pair activation images; from there:
//   1. the three 1x1-convolution channels are one more split-operand product per 16-row tile (K = 64: 6 MFMAs, four
//      B-fragment reads - the tower's own fragments at the centre tap), batch norm folded into the fragment image
//      (heads_prepare); channels 0..2 of a position land in lane group 0.  ReLU; the policy features go to LDS as f16
//      pairs in B-fragment order [piece][board 16][k 192] (k = channel * P + position), the value features as fp32;
//   2. the policy FC is a split-operand product too: column tile (16 outputs) x board (16, G used) x K = 192 in 6 k-steps
//      = 18 MFMAs per column tile; the 72 KB of weight fragments never touch LDS - every wave requests the 12 KB of its
//      column tile straight from L2 BEFORE step 1, so their latency is covered; the value FC (81 -> 3) stays on the VALU;
//   3. softmax, stores.
// (Before: fp32 MFMAs with most rows idle on an fp32 image of the last layer, the FC weights copied to LDS per group:
// 14.5 - 18.7 k cycles per group, profiles/r03_check_w2_ring_zero_block.txt.)
// C must provide: P, A, M, MT, IMG, ZOFF, HQ_OFF (12 KB, free during the head phase), HD1_OFF (the 1x1 fragment image
// 4 KB + its table 80 B, staged once per workgroup: stage_head_tables), AUX, HB_OFF, VW_OFF.
// once per workgroup: the 1x1-convolution fragment image (4 KB) and its table (accumulator start values [16], 2^-e) -> LDS
template <typename C, int NTHR>
__device__ __forceinline__ void stage_head_tables(unsigned char *smem, const NetDev &net, int tid) {
    for (int e = tid; e < 4096 / 16; e += NTHR)
        reinterpret_cast<uint4 *>(smem + C::HD1_OFF)[e] = reinterpret_cast<const uint4 *>(net.hd1_img)[e];
    if (tid < 20) reinterpret_cast<float *>(smem + C::HD1_OFF + 4096)[tid] = net.hd1_tab[tid];
}

template <int S, int G, typename C, int NTHR>
__device__ __forceinline__ void run_heads_mfma(unsigned char *smem, const NetDev &net, int b0, int batch, int want_logits,
                                               float *__restrict__ policy, float *__restrict__ value, int tid, int wave,
                                               long long *tl) {
    constexpr int P = C::P, A = C::A, M = C::M, IMG = C::IMG;
    constexpr int NW = NTHR / 64, NT = 6, KS = 6, NTW = (NT + NW - 1) / NW;
    using F = FmtF16;
    asm volatile("" : "+v"(tid));                          // opaque: nothing derived from it is hoisted out of the group loop
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
    auto stamp = [&](int i) { if (tl && tid == 0) tl[i] = (long long)__builtin_amdgcn_s_memtime(); };
    float *hval = reinterpret_cast<float *>(smem + C::AUX);   // [G][P]
    float *plog = hval + G * P;                               // [G][NT * 16]
    float *vlog = plog + G * NT * 16;                         // [G][4]
    // ---- policy FC weight fragments of this wave's column tiles: requested now, used in step 2 ----
    i32x4v fw[NTW][KS][2];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int nt = wave + u * NW;
        const unsigned char *base = net.pfc_img + (size_t)(nt < NT ? nt : 0) * KS * 2048 + lane * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int p = 0; p < 2; ++p) gmem_load_frag(fw[u][s][p], base, (s * 2 + p) * 1024);
    }
    // the feature image [piece][board 16][k 192]: boards < G get their 2P features below and zeros in the K padding here
    // (the FC weights are zero there, but 0 x Inf would not be - the region held other data); boards >= G are never
    // written: whatever they hold only reaches output columns that are discarded
    for (int e = tid; e < 2 * G * (192 - 2 * P); e += NTHR) {
        const int pc = e / (G * (192 - 2 * P)), r2 = e - pc * G * (192 - 2 * P), bl = r2 / (192 - 2 * P), kk = r2 - bl * (192 - 2 * P);
        reinterpret_cast<_Float16 *>(smem + C::HQ_OFF)[(pc * 16 + bl) * 192 + 2 * P + kk] = (_Float16)0.f;
    }
    // ---- 1. 1x1 convolutions: fragments of all of this wave's row tiles first, then the products ----
    i32x4v ha[2][2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int p = 0; p < 2; ++p) lds_load_frag<0>(ha[kc][p], smem, C::HD1_OFF + (kc * 2 + p) * 1024 + lane * 16);
    const f32x4 ini = *reinterpret_cast<const f32x4 *>(smem + C::HD1_OFF + 4096 + lg * 16);
    const float down1 = *reinterpret_cast<const float *>(smem + C::HD1_OFF + 4096 + 64), down1x = down1 * (1.f / 2048.f);
    constexpr int TPW = (C::MT + NW - 1) / NW;
    i32x4v fb[TPW][2][2];                                  // [tile][piece][kc]
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * NW;
        const int row = (t < C::MT ? t : wave) * 16 + li;
        const int nat = row * 64 + ((lg ^ ((row >> 1) & 3)) << 4);
        const int addr = row < M ? nat : C::ZOFF + (nat & 255);
        lds_load_frag<0 * IMG>(fb[q][0][0], smem, addr);
        lds_load_frag<1 * IMG>(fb[q][0][1], smem, addr);
        lds_load_frag<2 * IMG>(fb[q][1][0], smem, addr);
        lds_load_frag<3 * IMG>(fb[q][1][1], smem, addr);
    }
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * NW;
        const int row = t * 16 + li;
        f32x4 a0 = ini, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            a0 = mfma16<F>(ha[kc][0], fb[q][0][kc], a0);
            a1 = mfma16<F>(ha[kc][1], fb[q][0][kc], a1);
            a1 = mfma16<F>(ha[kc][0], fb[q][1][kc], a1);
        }
        if (lg == 0 && t < C::MT && row < M) {
            const int bl = row / P, pp = row - bl * P;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float v = fmaxf(fmaf(a1[j], down1x, a0[j] * down1), 0.f);
                if (j == 2) {
                    hval[bl * P + pp] = v;
                } else {
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)((v - (float)h) * 2048.f);
                    _Float16 *hq = reinterpret_cast<_Float16 *>(smem + C::HQ_OFF);
                    hq[(0 * 16 + bl) * 192 + j * P + pp] = h;
                    hq[(1 * 16 + bl) * 192 + j * P + pp] = l;
                }
            }
        }
    }
    stamp(0);
    __syncthreads();
    stamp(1);
    // ---- 2. policy FC (waves take column tiles), value FC (all threads) ----
    const float down2 = net.pfc_tab[0], down2x = down2 * (1.f / 2048.f);
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int nt = wave + u * NW;
        if (nt < NT) {
            f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                i32x4v fh, fl;
                const int off = C::HQ_OFF + (li * 192 + s * 32 + lg * 8) * 2;
                lds_load_frag<0>(fh, smem, off);
                lds_load_frag<16 * 192 * 2>(fl, smem, off);
                a0 = mfma16<F>(fw[u][s][0], fh, a0);
                a1 = mfma16<F>(fw[u][s][1], fh, a1);
                a1 = mfma16<F>(fw[u][s][0], fl, a1);
            }
            if (li < G) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int a = nt * 16 + lg * 4 + j;
                    if (a < A)
                        plog[li * NT * 16 + a] = fmaf(a1[j], down2x, a0[j] * down2) + reinterpret_cast<const float *>(smem + C::HB_OFF)[a];
                }
            }
        }
    }
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
    // ---- 3. softmax, stores ----
    for (int bl = wave; bl < G; bl += NW) {
        const int b = b0 + bl;
        if (b >= batch) continue;
        const float l0 = plog[bl * NT * 16 + lane];
        const float l1 = lane + 64 < A ? plog[bl * NT * 16 + lane + 64] : -INFINITY;
        float m = fmaxf(l0, l1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float e0 = expf(l0 - m), e1 = lane + 64 < A ? expf(l1 - m) : 0.f;
        float sum = e0 + e1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.f / sum;
        __builtin_nontemporal_store(want_logits ? l0 : e0 * inv, &policy[(size_t)b * A + lane]);
        if (lane + 64 < A) __builtin_nontemporal_store(want_logits ? l1 : e1 * inv, &policy[(size_t)b * A + lane + 64]);
        if (lane < 3) {
            const float v0 = vlog[bl * 4], v1 = vlog[bl * 4 + 1], v2 = vlog[bl * 4 + 2];
            const float vm = fmaxf(v0, fmaxf(v1, v2));
            const float x0 = expf(v0 - vm), x1 = expf(v1 - vm), x2 = expf(v2 - vm);
            const float es = x0 + x1 + x2;
            const float mine = lane == 0 ? x0 : (lane == 1 ? x1 : x2);
            value[(size_t)b * 3 + lane] = mine / es;
        }
    }
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

}  // namespace
