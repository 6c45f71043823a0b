// DualNet forward for gfx950: the residual tower as Winograd F(2,3) along ONE axis on split operands (round 4; its own file
// since round 5 - the 2-D Winograd tower it grew out of is in the git history: commit 04640d1, tools/experiments/kernels/).
//
// net_forward_split.hip runs the fp32 3x3 convolutions on the 16-bit matrix pipe as three f16 products per fp32 product
// (a = ah + al, w = wh + wl:  a w ~ ah wh + ah wl + al wh) and is bound by the MFMA count (857 per wave and layer for three
// boards); 2-D Winograd F(2x2,3x3) issues 480 but 5.3 VALU instructions beside each and is bound by instruction issue.
// Transforming along x only sits between: 600 MFMAs per wave and layer, one transform pass per side - 1.3 VALU instructions
// per MFMA, inside what an MFMA's 16 cycles hide.  Shared with the kernels of this family:
//   * V = B^T d is computed in fp32 (one add per value, exact inputs) and split AFTERWARDS into two f16 pieces; the weights
//     U = G g are computed in fp64 on the host with the batch-norm scale folded in, scaled by a power of two per layer and
//     split there.  Logit error against the reference's fp64 forward: the class of the reference's own fp32 path.
//   * the LOW pieces are kept UNSCALED (al = rn16(a - ah), no 2^11): v_mfma_f32_16x16x32_f16 keeps f16 subnormals, so the
//     cross terms have the right magnitude by themselves and ONE accumulator set takes all three products.
//   * wave w of the four owns transform POINT w: its weight fragments of a layer stay in registers (AGPRs, requested by
//     inline asm with explicit waits - tests/test_isa_waits.py checks the emitted code), the sum over the points goes
//     through a 32 KB LDS exchange, wave w' finishes output channels [16 w', 16 w' + 16).
//   * activations stay in LDS as fp32 [position][64 channels] (block input X - also the residual - and the intermediate
//     H), 16-byte chunk index XOR-swizzled by a function of the position under which the reads of sixteen units are
//     conflict-free for every cell.
// Stem (6 -> 64 channels) and heads are the direct split kernel's (im2col'ed K = 64 product; 1x1 convolutions and policy
// FC on the 16-bit pipe), reading / writing the fp32 images.  f16 range guard as there: |V| <= 2 |d| must stay below 65504,
// so a layer output beyond 16000 raises the flag and the exact-fp32 kernel redoes the batch.
// Reference: nn/network/res_block.py:8-38, nn/network/dual_net.py:41-52.
#include "w1d_common.h"

// W1_ABL (experiments only, tools/experiments/wb_ablation.sh: results are wrong, the timing says what a class of riders costs in the
// three-board variant): 1 no weight requests, 2 no input transforms, 4 no cell reads, 8 no epilogue arithmetic, 16 no exchange
// traffic, 32 no row barrier, 128 no MFMAs, 256 no heads, 512 no stem
#ifndef W1_ABL
#define W1_ABL 0
#endif

namespace {


template <int G>
struct WsCfg {
    static constexpr int S = 9, P = 81, A = 82, M = G * P;
    static constexpr int MT = (M + 15) / 16;              // row tiles of 16 positions (stem, heads)
    static constexpr int NT = G * 25;                     // Winograd tiles (5 x 5 per board)
    static constexpr int NRT = (NT + 15) / 16;            // row tiles of 16 Winograd tiles
    static constexpr int NTHR = 256, NW = 4;
    // activation buffers: fp32 [row][16 x 16 B]; row M = dump row (stores of positions outside the board), row M + 1 =
    // zero row (patch cells outside the board, 256-byte aligned: a read keeps the bank of its natural address)
    static constexpr int ROWS = M + 2;
    static constexpr int BUF = ROWS * 256;
    static constexpr int DUMP_REL = M * 256, ZERO_REL = (M + 1) * 256;
    static constexpr int X_OFF = 0, H_OFF = BUF;
    static constexpr int EX_OFF = 2 * BUF;                // exchange: [wave 4][z 2][ct 4][lane 64][16 B]
    static constexpr int EX_BYTES = 32768;
    // head tables, staged once per workgroup
    static constexpr int HD1_OFF = EX_OFF + EX_BYTES;     // 1x1 fragment image 4 KB + table 128 B
    static constexpr int HB_OFF = HD1_OFF + 4096 + 128;   // policy FC bias [A] (padded to 84)
    static constexpr int VW_OFF = HB_OFF + 84 * 4;        // value FC weights [3][P] + bias [3] (padded)
    static constexpr int LDS_BYTES = VW_OFF + ((3 * P + 3 + 3) & ~3) * 4;
    // stem overlay (over H and the exchange): im2col'ed input as f16-pair images [piece 2][kc 2][row][64 B] + planes
    static constexpr int ZOFF = ((M + 1) * 64 + 255) & ~255;
    static constexpr int IMG = ZOFF + 256;
    static constexpr int SI_OFF = H_OFF;
    static constexpr int STAGE = SI_OFF + 4 * IMG;
    static constexpr int RTW = (MT + NW - 1) / NW;        // stem: row tiles per wave
    static constexpr int SS_OFF = (STAGE + G * 6 * P * 4 + 15) & ~15;   // stem batch-norm scale [64] + shift [64] (w1d kernel)
    // head overlay (over H): policy features as f16 pairs 12 KB, scratch
    static constexpr int HQ_OFF = H_OFF;
    static constexpr int AUX = H_OFF + 12288;
    static_assert(SS_OFF + 512 <= EX_OFF + EX_BYTES, "stem overlay");
    static_assert(AUX + G * (P + 96 + 4) * 4 <= H_OFF + M * 256, "head overlay");
    static_assert(LDS_BYTES <= 163840, "LDS");
};

// 16-byte chunk XOR of activation row R (position 81 b + 9 y + x), spread over chunk-index bits 0, 2, 3 (bit 1 is the one in
// which the two lane groups of a ds_read_b128 cycle differ).  The kernel's sixteen MFMA columns are the units
// u = 5 board + t (outputs x = 2t, 2t + 1 of ONE board row); a unit's cells x = 2t - 1 .. 2t + 2 have (x + 1) / 2 = t or t + 1,
// so g = (5 board + (x + 1) / 2) mod 8 is distinct over the eight units of either half of a ds_read_b128 cycle
__host__ __device__ inline int w1_swz(int R) {
    const int b = R >= 162 ? 2 : (R >= 81 ? 1 : 0);
    const int p = R - 81 * b, y = (p * 57) >> 9, x = p - 9 * y;
    const int g = (5 * b + ((x + 1) >> 1)) & 7;
    return (g & 1) | ((g & 6) << 1);
}

// ... one board per workgroup (units u = 5 (row mod 3) + t): g = ((x + 1) / 2 + 5 (y mod 3)) mod 8
__host__ __device__ inline int w1g1_swz(int R) {
    const int y = (R * 57) >> 9, x = R - 9 * y;
    const int g = (((x + 1) >> 1) + 5 * (y % 3)) & 7;
    return (g & 1) | ((g & 6) << 1);
}

// Request schedule of the three-board variant (round 5).  A wave's 48 fragments of a layer are 48 KB; four waves' requests
// pass through the CU's vector L1 at 64 B per clock, i.e. 16 clocks - one MFMA - per request and wave.  Round 4 issued 32 of
// the 48 inside row 8 (768 clocks of MFMAs, 2 048 of L1): row 8 took 2 200 - 2 700 ticks, row 0 and row 1 waited for what
// was still in flight.  A fragment can only be requested once its register is free, so the schedule follows the uses:
//   kind 0: tap 1 of the next layer -> the spare slot S1N (free all layer): rows 1 - 6, 3 / 3 / 3 / 3 / 2 / 2
//   kind 1: tap 2, k-chunk 1 of the next layer -> its OTHER buffer (ua[2][1] for even layers, the VGPR fragments ux for odd
//           ones: double-buffered since round 5, 32 VGPRs): rows 5 - 6, in the order the next layer's row 0 uses them
//   kind 2: tap 2, k-chunk 0 of the next layer -> ua[2][0], freed by row 7's last MFMAs: row 7 from slice 53, row 8's first
//   kind 3: tap 0 of the next layer -> ua[0], freed by row 8's first 24 MFMAs: row 8 from slice 13 (12 fragments)
//   kind 4: tap 0 / k-chunk 1 / high pieces of THIS layer (the four fragments row 8 of the previous layer freed last): row 0
// -> -1, or 16 kind + fragment (8 kc + 4 piece + ct).  Issue order = wait order (w1_wait_* below).
constexpr int w1_sched(int y, int m) {
    if (y >= 1 && y <= 4 && (m == 46 || m == 52 || m == 58)) return 3 * (y - 1) + (m - 46) / 6;
    if ((y == 5 || y == 6) && (m == 46 || m == 52)) return 12 + 2 * (y - 5) + (m - 46) / 6;
    if ((y == 5 || y == 6) && (m == 49 || m == 55 || m == 61 || m == 67)) {
        const int i = 4 * (y - 5) + (m - 49) / 6;              // 0 .. 3: low pieces (used first), 4 .. 7: high pieces
        return 16 + (i < 4 ? 12 + i : 8 + (i - 4));
    }
    if (y == 7 && m >= 53 && m <= 69 && (m - 53) % 4 == 0) { const int i = (m - 53) / 4; return 32 + (i < 4 ? 4 + i : 0); }
    if (y == 8 && (m == 1 || m == 5 || m == 9)) return 32 + 1 + (m - 1) / 4;
    if (y == 8 && m >= 13 && m <= 46 && (m - 13) % 3 == 0) {
        const int i = (m - 13) / 3;                            // 0 .. 11: (kc 0, low), (kc 0, high), (kc 1, low)
        return 48 + (i < 4 ? 4 + i : (i < 8 ? i - 4 : 12 + (i - 8)));
    }
    if (y == 0 && (m == 1 || m == 4 || m == 7 || m == 10)) return 64 + 8 + (m - 1) / 3;
    return -1;
}
// Requests behind the last one a wait is for (in issue order): the layer top needs tap 1 (last: row 6, slice 52), row 0's
// slice 24 needs tap 2 (last: row 8, slice 9), row 1 needs tap 0 (last: row 0, slice 10: nothing behind it)
constexpr int w1_count_from(int y0, int m0) {                  // scheduled requests strictly behind (y0, m0), up to the end of row 8
    int n = 0;
    for (int y = y0; y <= 8; ++y)
        for (int m = (y == y0 ? m0 + 1 : 0); m < 72; ++m)
            if (y != 0 && w1_sched(y, m) >= 0) ++n;
    return n;
}
constexpr int kW1WaitTop = w1_count_from(6, 52);               // 23: tap 1 has arrived
constexpr int kW1WaitTap2 = w1_count_from(8, 9) + 2 + 4;       // 18: + the layer's shift and scale, + row 0's four requests
static_assert(kW1WaitTop == 23 && kW1WaitTap2 == 18, "request schedule and wait counts");

// Heads on the 16-bit matrix pipe (as split_common.h: run_heads_mfma), reading the block output from the fp32 image X:
// a B fragment (position li of a 16-row tile, channels 32 kc + 8 lg ..) is two 16-byte reads + the operand split.
template <int G, typename C, int SWZ>
__device__ __forceinline__ void run_heads_x32(unsigned char *smem, const NetDev &net, int b0, int batch, int want_logits,
                                              float *__restrict__ policy, float *__restrict__ value, int tid, int wave,
                                              long long *tl) {
    constexpr int P = C::P, A = C::A, M = C::M, NTHR = C::NTHR;
    constexpr int NW = NTHR / 64, NT = 6, KS = 6, NTW = (NT + NW - 1) / NW;
    using F = FmtF16;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
    auto stamp = [&](int i) { if (tl && tid == 0) tl[i] = (long long)__builtin_amdgcn_s_memtime(); };
    // (__shfl_xor derives its addresses from a lane id that hipcc computes once per kernel and keeps in scratch)
    auto lane_xor = [&](float v, int o) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ o) << 2, __builtin_bit_cast(int, v)));
    };
    float *hval = reinterpret_cast<float *>(smem + C::AUX);   // [G][P]
    float *plog = hval + G * P;                               // [G][NT * 16]
    float *vlog = plog + G * NT * 16;                         // [G][4]
    const float down2 = net.pfc_tab[0], down2x = down2 * (1.f / 2048.f);   // (requested here: behind the barrier its L2 round trip is exposed)
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
    for (int e = tid; e < 2 * G * (192 - 2 * P); e += NTHR) {
        const int pc = e / (G * (192 - 2 * P)), r2 = e - pc * G * (192 - 2 * P), bl = r2 / (192 - 2 * P), kk = r2 - bl * (192 - 2 * P);
        reinterpret_cast<_Float16 *>(smem + C::HQ_OFF)[(pc * 16 + bl) * 192 + 2 * P + kk] = (_Float16)0.f;
    }
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
        const int rr = row < M ? row : M + 1;              // zero row
        const int sw = row < M ? (SWZ == 2 ? w1g1_swz(row) : w1_swz(row)) : 0;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int a0 = C::X_OFF + rr * 256 + (((kc * 8 + lg * 2) ^ sw) << 4);
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(smem + a0);
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(smem + (a0 ^ 16));
            uint2 p0[2], p1[2];
            split4<F>(v0, p0);
            split4<F>(v1, p1);
            fb[q][0][kc] = i32x4v{(int)p0[0].x, (int)p0[0].y, (int)p1[0].x, (int)p1[0].y};
            fb[q][1][kc] = i32x4v{(int)p0[1].x, (int)p0[1].y, (int)p1[1].x, (int)p1[1].y};
        }
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
        sv += lane_xor(sv, 8);
        sv += lane_xor(sv, 4);
        sv += lane_xor(sv, 2);
        sv += lane_xor(sv, 1);
        if (part == 0) vlog[bl * 4 + c] = sv + reinterpret_cast<const float *>(smem + C::VW_OFF)[3 * P + c];
    }
    stamp(2);
    __syncthreads();
    for (int bl = wave; bl < G; bl += NW) {
        const int b = b0 + bl;
        if (b >= batch) continue;
        const float l0 = plog[bl * NT * 16 + lane];
        const float l1 = lane + 64 < A ? plog[bl * NT * 16 + lane + 64] : -INFINITY;
        float m = fmaxf(l0, l1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, lane_xor(m, o));
        const float e0 = expf(l0 - m), e1 = lane + 64 < A ? expf(l1 - m) : 0.f;
        float sum = e0 + e1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += lane_xor(sum, o);
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

// The 9x9 tower as Winograd F(2,3) along x only (the 9x9 default; TG_FWD_ALGO=w1d): 600 MFMAs per wave and layer for three
// boards (2-D Winograd: 480, direct: 857), one transform pass per side.  G = 3 boards per workgroup for throughput launches,
// G = 1 for launches up to the CU count - same bits either way.
// PROF: s_memtime stamps of workgroup 0 / wave 0: [0] group start, [1] input staged, [2] stem done, [3..14] layer done, [15] heads done, [64..66] inside the heads: 1x1 convolutions done, barrier passed, FCs done
template <int G, bool PROF>
__global__ __launch_bounds__(256, 1) void dualnet_fwd_w1d_kernel(
    NetDev net, const float *__restrict__ planes, int batch, int want_logits,
    float *__restrict__ policy, float *__restrict__ value, int *__restrict__ overflow, int *__restrict__ group_bits) {
    using C = WsCfg<G>;
    using F = FmtF16;
    constexpr int P = C::P, M = C::M, NTHR = C::NTHR, RTW = C::RTW, IMG = C::IMG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    if (static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)smem)) != 0u)
        __builtin_trap();                                      // absolute LDS addressing below
    // ---- once per workgroup: zero + dump rows, head tables ----
    for (int e = tid; e < 2 * 2 * 64; e += NTHR) {             // rows M, M + 1 of X and H
        const int buf = e >> 7, r = (e >> 6) & 1, c = e & 63;
        reinterpret_cast<float *>(smem + buf * C::BUF + (M + r) * 256)[c] = 0.f;
    }
    for (int e = tid; e < C::A; e += NTHR) reinterpret_cast<float *>(smem + C::HB_OFF)[e] = net.pfc_b[e];
    for (int e = tid; e < 3 * P + 3; e += NTHR)
        reinterpret_cast<float *>(smem + C::VW_OFF)[e] = e < 3 * P ? net.vfc_w[e] : net.vfc_b[e - 3 * P];
    stage_head_tables<C, NTHR>(smem, net, tid);

    int stamp_i = 0;
    auto stamp = [&]() {
        if constexpr (PROF)
            if (blockIdx.x == 0 && wave == 0 && stamp_i < 40 && fresh_lane() == 0) net.timeline[stamp_i++] = (long long)__builtin_amdgcn_s_memtime();
    };
    int ovf = 0;
    const int n_groups = (batch + G - 1) / G;
    constexpr int NPL = (G * 6 * P + NTHR - 1) / NTHR;
    float pre[NPL];
    float ssv;                                                 // the stem's batch-norm scale (threads 0 .. 63) / shift (64 .. 127): travels with the planes
    auto fetch_planes = [&](int grp2) __attribute__((always_inline)) {
        const int ft = wave * 64 + fresh_lane();
        ssv = ft < 64 ? net.sscale[ft] : (ft < 128 ? net.shift[ft - 64] : 0.f);
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int e = ft + i * NTHR;
            const int b = grp2 * G + e / (6 * P);
            pre[i] = (e < G * 6 * P && grp2 < n_groups && b < batch)
                         ? __builtin_nontemporal_load(&planes[(size_t)grp2 * G * 6 * P + e]) : 0.f;
        }
    };
    fetch_planes(blockIdx.x);
    const float sgn = wave == 1 ? 1.f : -1.f;                  // row pass of point row w: d[ra] + sgn d[rb]
    static_assert(G == 3 || G == 1, "dualnet_fwd_w1d_kernel: one or three boards per workgroup");
    // This wave's weight fragments of a layer, [slot 4][kc 2][piece 2][ct 4]: slots 0 / 2 = taps ky 0 / 2, slots 1 and 3 take
    // tap 1 of even / odd layers in turn (the spare one is filled for the next layer while this one runs).  AGPRs, requested
    // by inline asm (see the kernel above): explicit waits, in-order returns.
    i32x4v ua[4][2][2][4];
    const int wlane = lane * 16;
    {
        const unsigned char *w0 = net.w1_w + (size_t)wave * 49152;
        w1_request_tap<1>(ua, w0 + 16384, wlane);
        if constexpr (G == 3) {
            // (the order the first layer waits for them in - that of the steady-state schedule w1_sched: tap 2 with k-chunk 1
            // first, tap 0 without the four fragments row 0 of every layer requests itself)
            static_for<8>([&](auto I_) { constexpr int i = decltype(I_)::value; w1_request<2>(ua, w0 + 2 * 16384, wlane, std::integral_constant<int, (i < 4 ? 12 + i : 8 + (i - 4))>{}); });
            static_for<8>([&](auto I_) { constexpr int i = decltype(I_)::value; w1_request<2>(ua, w0 + 2 * 16384, wlane, std::integral_constant<int, (i < 4 ? 4 + i : i - 4)>{}); });
            static_for<12>([&](auto I_) { constexpr int i = decltype(I_)::value; w1_request<0>(ua, w0, wlane, std::integral_constant<int, (i < 4 ? 4 + i : (i < 8 ? i - 4 : 12 + (i - 8)))>{}); });
        } else {
            w1_request_tap<0>(ua, w0, wlane);
            w1_request_tap<2>(ua, w0 + 2 * 16384, wlane);
        }
    }

    // Groups beyond a workgroup's first are handed out by a ticket counter (overflow[1], zeroed with the range flag): a
    // workgroup that starts late - its CU was running another stream's tree kernel - takes fewer groups instead of
    // holding the launch up with a full static share.  The ticket travels through a spare word of the bias table.
    int *const ticket_lds = reinterpret_cast<int *>(smem + C::HB_OFF + 83 * 4);
    for (int grp = blockIdx.x; grp < n_groups;) {
        const int b0 = grp * G;
        if (wave == 0 && fresh_lane() == 0) *ticket_lds = overflow ? (int)gridDim.x + atomicAdd(overflow + 1, 1) : grp + (int)gridDim.x;
        stamp();
        // ================= stem: planes -> im2col'ed f16-pair images (K = 9 taps x 6 planes, padded to 64) =================
        // (its 16 weight fragments are requested first: their L2 round trip runs under the staging pass)
        i32x4v fa[2][2][4];                                      // [kc][piece][ct]
        {
            const int wvg = fresh_lane() * 16;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        gmem_load_frag(fa[kc][p][c], net.wsplit + (size_t)kc * 8192, wvg + (p * 4 + c) * 1024);
        }
        {
            float *st = reinterpret_cast<float *>(smem + C::STAGE);
            const int stid = wave * 64 + fresh_lane();
#pragma unroll
            for (int i = 0; i < NPL; ++i)
                if (stid + i * NTHR < G * 6 * P) st[stid + i * NTHR] = pre[i];
            for (int e = stid; e < 4 * 64; e += NTHR)           // zero blocks of the four images
                reinterpret_cast<unsigned *>(smem + C::SI_OFF + (e >> 6) * IMG + C::ZOFF)[e & 63] = 0u;
            // (scale / shift through the overlay: sixteen exposed L2 round trips per group when the stem's epilogue fetched them itself)
            if (stid < 128) reinterpret_cast<float *>(smem + C::SS_OFF)[stid] = ssv;
            __syncthreads();
            for (int row = stid; row < M; row += NTHR) {
                const int bl = row / P, p = row - bl * P, y = p / 9, x = p - y * 9;
                const float *src = st + bl * 6 * P + p;
                const int swz = (row >> 1) & 3;
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {
                    f32x4 lo, hi;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int k = sl * 8 + j, t = k / 6, c = k - t * 6;
                        const int dy = t / 3 - 1, dx = t % 3 - 1;
                        const bool ok = k < 54 && (unsigned)(y + dy) < 9u && (unsigned)(x + dx) < 9u;
                        const float v = ok ? src[c * P + dy * 9 + dx] : 0.f;
                        if (j < 4) lo[j] = v; else hi[j - 4] = v;
                    }
                    uint2 plo[2], phi[2];
                    split4<F>(lo, plo);
                    split4<F>(hi, phi);
                    const int kc = sl >> 2, slot = (sl & 3) ^ swz;
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        *reinterpret_cast<uint4 *>(smem + C::SI_OFF + (q * 2 + kc) * IMG + row * 64 + slot * 16) =
                            uint4{plo[q].x, plo[q].y, phi[q].x, phi[q].y};
                }
            }
        }
        __syncthreads();
        stamp();
        float amax = 0.f;
        {
            // stem product: 2 k-chunks x 4 channel tiles x RTW row tiles x 3 f16 products (two accumulator sets, scaled
            // low pieces: the direct split kernel's image and weights), batch norm, ReLU -> X (fp32, swizzled)
            const int slane = fresh_lane();                      // (per group: what hangs off the lane id is recomputed, not spilled)
            const int sli = slane & 15, slg = slane >> 4;
#pragma unroll
            for (int r = 0; r < RTW; ++r) {
                int row = (wave * RTW + r) * 16 + sli;
                asm volatile("" : "+v"(row));
                const int nat = row * 64 + ((slg ^ ((row >> 1) & 3)) << 4);
                const int addr = C::SI_OFF + (row < M ? nat : C::ZOFF + (nat & 255));
                i32x4v fb[2][2];                                 // [piece][kc]
                lds_load_frag<0 * IMG>(fb[0][0], smem, addr);
                lds_load_frag<1 * IMG>(fb[0][1], smem, addr);
                lds_load_frag<2 * IMG>(fb[1][0], smem, addr);
                lds_load_frag<3 * IMG>(fb[1][1], smem, addr);
                const int orow = row < M ? row : M;
                const int osw = row < M ? (G == 1 ? w1g1_swz(row) : w1_swz(row)) : 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
                        a0 = mfma16<F>(fa[kc][0][c], fb[0][kc], a0);
                        a1 = mfma16<F>(fa[kc][1][c], fb[0][kc], a1);
                        a1 = mfma16<F>(fa[kc][0][c], fb[1][kc], a1);
                    }
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(smem + C::SS_OFF + (c * 16 + slg * 4) * 4);
                    const f32x4 sh = *reinterpret_cast<const f32x4 *>(smem + C::SS_OFF + 256 + (c * 16 + slg * 4) * 4);
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = fmaf(a1[j], 1.f / 2048.f, a0[j]);
                        t = fmaf(t, sc[j], sh[j]);
                        v[j] = fmaxf(t, 0.f);
                    }
                    amax = fmaxf(fmaxf(amax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                    *reinterpret_cast<f32x4 *>(smem + C::X_OFF + orow * 256 + (((c * 4 + slg) ^ osw) << 4)) = v;
                }
            }
        }
        __syncthreads();                                        // X complete; the overlay is free again
        if (wave == 0) reinterpret_cast<float *>(smem + C::H_OFF + (M + 1) * 256)[fresh_lane()] = 0.f;   // H's zero row was under it
        stamp();

        // ================= tower: 12 layers, Winograd F(2,3) along x, the three taps along y direct =================
        // Output row y of all three boards is one MFMA column set (unit u = 5 board + t: outputs (y, 2t), (y, 2t + 1)); wave w
        // owns transform point w: its V_w rows (f16 hi / lo pieces, 16 registers a row) live in registers and serve three
        // output rows each - tap ky of row y multiplies V_w[y + ky - 1] -, its 48 weight fragments of the layer in AGPRs.
        // M_w goes through the LDS exchange; wave w' finishes output channels [16 w', 16 w' + 16): out0 = m0 + m1 + m2,
        // out1 = m1 - m2 - m3, shift, residual, ReLU.  Everything but the MFMAs of row y rides along them: exchange + epilogue
        // of row y - 1, input transform of row y + 2, cell reads of row y + 3, weight requests of the next layer.
        if constexpr (G == 1) {
            // ================= one board per workgroup: MFMA column u = 5 yi + t, row tile j = output rows 3j + yi =================
            // Same arithmetic per output as the three-board variant below, in the same order (taps ky = 0, 1, 2 - rows outside the
            // board contribute exact zeros instead of being skipped; k-chunks; cross terms first): results do not depend on which of
            // the two a position went through.  A row tile needs three V rows per lane (input rows 3j + yi - 1 .. + 1), none shared
            // with the next row tile: 120 VALU of input transform per 72 MFMAs instead of 40.
            const int glane = fresh_lane(), gli = glane & 15, glg = glane >> 4, wlane = glane * 16;   // (per group: not to be hoisted out of the group loop)
            const int uyi = gli / 5, ut = gli - 5 * uyi;
            const bool uv = gli < 15;
            const int xa = wave == 0 ? 2 * ut - 1 : (wave == 2 ? 2 * ut + 1 : 2 * ut);
            const int xb = wave == 0 ? 2 * ut + 1 : (wave == 1 ? 2 * ut + 1 : (wave == 2 ? 2 * ut : 2 * ut + 2));
            // cell (row 3j + yi + q, column x) at row tile 0: address (may lie above the image: q = -1, yi = 0 - never read), stride per row tile
            auto cell1 = [&](int x, int q, int chunk, int invalid_rel, int &adr, int &str) {
                const bool ok = uv && x >= 0 && x < 9;
                const int r0 = uyi + q, rm = (r0 + 3) % 3;
                const int g = (((x + 1) >> 1) + 5 * rm) & 7, sw = (g & 1) | ((g & 6) << 1);
                adr = (ok ? (9 * r0 + x) * 256 : invalid_rel) + ((chunk ^ sw) << 4);
                str = ok ? 27 * 256 : 0;
            };
            int cA[3], cB[3], sA, sB, sdummy, curO0, curO1, curR0, curR1, strO0, strO1, strR0, strR1;
            cell1(xa, -1, glg * 2, C::ZERO_REL, cA[0], sA);
            cell1(xa, 0, glg * 2, C::ZERO_REL, cA[1], sdummy);
            cell1(xa, 1, glg * 2, C::ZERO_REL, cA[2], sdummy);
            cell1(xb, -1, glg * 2, C::ZERO_REL, cB[0], sB);
            cell1(xb, 0, glg * 2, C::ZERO_REL, cB[1], sdummy);
            cell1(xb, 1, glg * 2, C::ZERO_REL, cB[2], sdummy);
            cell1(2 * ut, 0, wave * 4 + glg, C::DUMP_REL, curO0, strO0);
            cell1(2 * ut + 1, 0, wave * 4 + glg, C::DUMP_REL, curO1, strO1);
            cell1(2 * ut, 0, wave * 4 + glg, C::ZERO_REL, curR0, strR0);
            cell1(2 * ut + 1, 0, wave * 4 + glg, C::ZERO_REL, curR1, strR1);
            f32x4 dq[2][2][2];
            i32x4v vh[2][3][2], vl[2][3][2];                           // [row tile parity][q + 1][kc]
            f32x4 acc[2][4];
            f32x4 ez[4], eres[2], ev[2];
            float tvv[4];
            unsigned thh[2];
            // one of the eight cell reads of (row tile J, relative row Q)
            auto rd = [&](auto IN_, auto J_, auto Q_, auto I_) __attribute__((always_inline)) {
                constexpr int IN = decltype(IN_)::value, j = decltype(J_)::value, qi = decltype(Q_)::value, i = decltype(I_)::value;
                constexpr int cb = i >> 2, kc = (i >> 1) & 1, h = i & 1;
                int a = (cb ? cB[qi] + j * sB : cA[qi] + j * sA);
                // rows -1 / 9: the zero row, same chunk (the low byte of the address: rows are 256 bytes)
                if constexpr (j == 0 && qi == 0) a = uyi == 0 ? C::ZERO_REL + (a & 255) : a;
                if constexpr (j == 2 && qi == 2) a = uyi == 2 ? C::ZERO_REL + (a & 255) : a;
                dq[cb][kc][h] = lds_f32x4_at<IN>(a ^ ((kc << 7) | (h << 4)));
            };
            // sub-step I (0 .. 15) of the input transform of relative row Q into V buffer P
            auto tr = [&](auto P_, auto Q_, auto I_) __attribute__((always_inline)) {
                constexpr int pb = decltype(P_)::value, qi = decltype(Q_)::value, i = decltype(I_)::value, kc = i >> 3, h = (i >> 2) & 1, q = i & 3;
                if constexpr (q == 0) {
                    tvv[0] = fmaf(dq[1][kc][h][0], sgn, dq[0][kc][h][0]);
                    tvv[1] = fmaf(dq[1][kc][h][1], sgn, dq[0][kc][h][1]);
                } else if constexpr (q == 1) {
                    tvv[2] = fmaf(dq[1][kc][h][2], sgn, dq[0][kc][h][2]);
                    tvv[3] = fmaf(dq[1][kc][h][3], sgn, dq[0][kc][h][3]);
                } else if constexpr (q == 2) {
                    thh[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{tvv[0], tvv[1]}, f16x2));
                    vh[pb][qi][kc][2 * h] = (int)thh[0];
                    vl[pb][qi][kc][2 * h] = (int)low_pieces(tvv[0], tvv[1], thh[0]);
                } else {
                    thh[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{tvv[2], tvv[3]}, f16x2));
                    vh[pb][qi][kc][2 * h + 1] = (int)thh[1];
                    vl[pb][qi][kc][2 * h + 1] = (int)low_pieces(tvv[2], tvv[3], thh[1]);
                }
            };
            f32x4 pshf;                                            // the previous layer's epilogue constants (its row tile 2 rides in this layer's row tile 0)
            float pdown;
            // Schedule of a row tile's 72 slices (one MFMA each + what rides along):  0-3 exchange writes of the PREVIOUS row tile
            // (row tile 0: of the previous LAYER's row tile 2 - output buffer = this layer's input, the other residual flag, constants
            // pshf / pdown; a group's first layer: a null epilogue - zero accumulators and constants, zero-row reads, dump-row stores),
            // 10 barrier, 11-12 exchange reads, 13 residual reads, 19-34 sums / shift / residual / ReLU, 35-36 stores | 11-18 cell
            // reads and 19-26 transform of this row tile's OWN third V row (input rows 3j + yi + 1: the last of them was stored
            // under the previous row tile and is visible behind this row tile's barrier) | 27-34 / 35-50 and 51-58 / 59-71 cell reads
            // and transforms of the NEXT row tile's first two V rows (row tile 2: of the next layer's row tile 0, from this layer's
            // output rows -1 .. 2, complete since row tile 1) | weight requests.  No barrier and no prologue at the layer boundary:
            // between a store and any other wave's read of it lies at least one row-tile barrier.
            auto layer_fn = [&](auto IN_, auto OUT_, auto RES_, int layer) __attribute__((always_inline)) {
                constexpr int IN = decltype(IN_)::value, OUT = decltype(OUT_)::value;
                constexpr bool RES = decltype(RES_)::value;
                constexpr int PAR = RES ? 1 : 0;
                constexpr int S1 = PAR ? 3 : 1, S1N = PAR ? 1 : 3;
                const int next_layer = layer + 1 < kTowerLayers ? layer + 1 : 0;
                const unsigned char *wnext = net.w1_w + ((size_t)next_layer * 4 + wave) * 49152;
                f32x4 shf = *reinterpret_cast<const f32x4 *>(net.w1_shift + layer * 64 + wave * 16 + glg * 4);
                float down = net.w1_down[layer];
                int exw = C::EX_OFF + wave * 4096 + glane * 16, exr = C::EX_OFF + wave * 1024 + glane * 16;
                asm volatile("" : "+v"(exw), "+v"(exr));
                if constexpr (!RES) {
                    if (layer == 0) {
                        // a group's first layer: nothing was prepared under a previous layer - the first two V rows of row tile 0
                        static_for<2>([&](auto Q_) {
                            static_for<8>([&](auto I_) { rd(IN_, std::integral_constant<int, 0>{}, Q_, I_); });
                            static_for<16>([&](auto I_) { tr(std::integral_constant<int, PAR>{}, Q_, I_); });
                        });
                    }
                }
                // taps 1 and 0 must have arrived (requested in that order; behind them tap 2's 16 requests, the shift and the scale)
                asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);                 // (an MFMA is no memory operation: nothing else keeps it behind the wait)
                // exchange + epilogue of a row tile: PREV = row tile 2 of the previous layer, else row tile J of this one
                auto epi = [&](auto PREV_, auto J_, auto I_) __attribute__((always_inline)) {
                    constexpr bool PREV = decltype(PREV_)::value;
                    constexpr int j = PREV ? 2 : decltype(J_)::value, i = decltype(I_)::value;
                    constexpr int par = PREV ? 1 - PAR : (j + PAR) & 1;
                    constexpr int OB = PREV ? IN : OUT;
                    constexpr bool RS = PREV ? !RES : RES;
                    const bool null_epi = PREV && layer == 0;
                    if constexpr (i < 4) {
                        lds_f32x4_put<par * 16384 + i * 1024>(exw, acc[par][i]);
                    } else if constexpr (i == 10) {
                        __syncthreads();
                    } else if constexpr (i == 11 || i == 12) {
                        ez[2 * (i - 11)] = lds_f32x4_at<par * 16384 + (2 * (i - 11)) * 4096>(exr);
                        ez[2 * (i - 11) + 1] = lds_f32x4_at<par * 16384 + (2 * (i - 11) + 1) * 4096>(exr);
                    } else if constexpr (i == 13) {
                        if constexpr (RS) {
                            const int zr = C::ZERO_REL + (glane * 16) % 256;
                            eres[0] = lds_f32x4_at<OB>(null_epi ? zr : curR0 + j * strR0);
                            eres[1] = lds_f32x4_at<OB>(null_epi ? zr : curR1 + j * strR1);
                        }
                    } else if constexpr (i >= 19 && i < 35) {
                        constexpr int k = i - 19, cc = k >> 3, e = (k >> 1) & 3, part = k & 1;
                        if constexpr (part == 0) {
                            ev[cc][e] = cc == 0 ? (ez[0][e] + ez[1][e]) + ez[2][e] : (ez[1][e] - ez[2][e]) - ez[3][e];
                        } else {
                            float tt = fmaf(ev[cc][e], PREV ? pdown : down, PREV ? pshf[e] : shf[e]);
                            if constexpr (RS) tt += eres[cc][e];
                            ev[cc][e] = fmaxf(tt, 0.f);
                        }
                    } else if constexpr (i == 35) {
                        amax = fmaxf(fmaxf(amax, ev[0][0]), ev[0][1]);
                        amax = fmaxf(fmaxf(amax, ev[0][2]), ev[0][3]);
                        lds_f32x4_put<OB>(null_epi ? C::DUMP_REL + (glane * 16) % 256 : curO0 + j * strO0, ev[0]);
                    } else if constexpr (i == 36) {
                        amax = fmaxf(fmaxf(amax, ev[1][0]), ev[1][1]);
                        amax = fmaxf(fmaxf(amax, ev[1][2]), ev[1][3]);
                        lds_f32x4_put<OB>(null_epi ? C::DUMP_REL + (glane * 16) % 256 : curO1 + j * strO1, ev[1]);
                    }
                };
                static_for<3>([&](auto J_) {
                    constexpr int j = decltype(J_)::value, par = (j + PAR) & 1, nb = 1 - par;
                    using PB = std::integral_constant<int, par>;
                    using NB = std::integral_constant<int, nb>;
                    using Q0 = std::integral_constant<int, 0>;
                    using Q1 = std::integral_constant<int, 1>;
                    using Q2 = std::integral_constant<int, 2>;
                    static_for<72>([&](auto M_) {
                        constexpr int m = decltype(M_)::value, ky = m / 24, q = m % 24, kc = q / 12, st = (q / 4) % 3, c = q % 4;
                        constexpr int slot = ky == 1 ? S1 : ky;
                        // tap 2, the shift and the scale (the next layer's requests start behind this wait: it would wait for them too)
                        if constexpr (j == 0 && m == 48) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            use_here(shf, down);
                            __builtin_amdgcn_sched_barrier(0);      // (hipcc had moved the first tap-2 MFMA in front of the wait: an MFMA is no memory operation)
                        }
                        if constexpr (st == 0)
                            acc[par][c] = mfma16<F>(ua[slot][kc][1][c], vh[par][ky][kc], m < 4 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[par][c]);
                        else if constexpr (st == 1) acc[par][c] = mfma16<F>(ua[slot][kc][0][c], vl[par][ky][kc], acc[par][c]);
                        else acc[par][c] = mfma16<F>(ua[slot][kc][0][c], vh[par][ky][kc], acc[par][c]);
                        // ---- what rides along ----
                        if constexpr (m < 37) {
                            if constexpr (j == 0) epi(std::true_type{}, J_, M_);
                            else epi(std::false_type{}, std::integral_constant<int, j - 1>{}, M_);
                        }
                        // this row tile's own third V row
                        if constexpr (m >= 11 && m < 19) rd(IN_, J_, Q2{}, std::integral_constant<int, m - 11>{});
                        if constexpr (m >= 19 && m < 27) { tr(PB{}, Q2{}, std::integral_constant<int, 2 * (m - 19)>{}); tr(PB{}, Q2{}, std::integral_constant<int, 2 * (m - 19) + 1>{}); }
                        // the next row tile's first two V rows (row tile 2: the next layer's row tile 0, from this layer's output)
                        if constexpr (m >= 27 && m < 35) {
                            if constexpr (j <= 1) rd(IN_, std::integral_constant<int, j + 1>{}, Q0{}, std::integral_constant<int, m - 27>{});
                            else rd(OUT_, Q0{}, Q0{}, std::integral_constant<int, m - 27>{});
                        }
                        if constexpr (m >= 35 && m < 51) tr(NB{}, Q0{}, std::integral_constant<int, m - 35>{});
                        if constexpr (m >= 51 && m < 59) {
                            if constexpr (j <= 1) rd(IN_, std::integral_constant<int, j + 1>{}, Q1{}, std::integral_constant<int, m - 51>{});
                            else rd(OUT_, Q0{}, Q1{}, std::integral_constant<int, m - 51>{});
                        }
                        if constexpr (m >= 59 && m < 62) { tr(NB{}, Q1{}, std::integral_constant<int, 2 * (m - 59)>{}); tr(NB{}, Q1{}, std::integral_constant<int, 2 * (m - 59) + 1>{}); }
                        if constexpr (m >= 62) tr(NB{}, Q1{}, std::integral_constant<int, m - 56>{});
                        // next layer's weights: tap 1 into the spare slot under row tiles 0 and 1, tap 0 right behind row tile 2's
                        // tap-0 MFMAs (the next layer needs it first), tap 2 behind the last MFMA
                        if constexpr (j <= 1 && m >= 50 && m < 72 && (m - 50) % 3 == 0) {
                            constexpr int f = j * 8 + (m - 50) / 3;
                            w1_request<S1N>(ua, wnext + 16384, wlane, std::integral_constant<int, f>{});
                        }
                        if constexpr (j == 2 && m >= 24 && m < 40) w1_request<0>(ua, wnext, wlane, std::integral_constant<int, m - 24>{});
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
                w1_request_tap<2>(ua, wnext + 2 * 16384, wlane);
                pshf = shf;
                pdown = down;
                if (!(amax < (float)kWsRangeLimit)) ovf = 1;            // f16 range guard (also catches NaN)
                stamp();
            };
            using IX = std::integral_constant<int, C::X_OFF>;
            using IH = std::integral_constant<int, C::H_OFF>;
            if (!(amax < (float)kWsRangeLimit)) ovf = 1;
            {
                // layer 0's null epilogue
                float z0, z1, z2, z3;
                asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z0), "=v"(z1), "=v"(z2), "=v"(z3));
                pshf = f32x4{z0, z1, z2, z3};
                pdown = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[1][c] = pshf;
            }
#pragma unroll 1
            for (int blk = 0; blk < kBlocks; ++blk) {
                layer_fn(IX{}, IH{}, std::false_type{}, 2 * blk);
                layer_fn(IH{}, IX{}, std::true_type{}, 2 * blk + 1);
            }
            {
                // the tower's last row tile (layer 11, output X, residual): on its own
                const int exw = C::EX_OFF + wave * 4096 + glane * 16, exr = C::EX_OFF + wave * 1024 + glane * 16;
                constexpr int par = (2 + 1) & 1;
                static_for<4>([&](auto I_) { lds_f32x4_put<par * 16384 + decltype(I_)::value * 1024>(exw, acc[par][decltype(I_)::value]); });
                eres[0] = lds_f32x4_at<C::X_OFF>(curR0 + 2 * strR0);
                eres[1] = lds_f32x4_at<C::X_OFF>(curR1 + 2 * strR1);
                __syncthreads();
                static_for<4>([&](auto I_) { ez[decltype(I_)::value] = lds_f32x4_at<par * 16384 + decltype(I_)::value * 4096>(exr); });
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o0 = (ez[0][e] + ez[1][e]) + ez[2][e], o1 = (ez[1][e] - ez[2][e]) - ez[3][e];
                    ev[0][e] = fmaxf(fmaf(o0, pdown, pshf[e]) + eres[0][e], 0.f);
                    ev[1][e] = fmaxf(fmaf(o1, pdown, pshf[e]) + eres[1][e], 0.f);
                    amax = fmaxf(fmaxf(amax, ev[0][e]), ev[1][e]);
                }
                lds_f32x4_put<C::X_OFF>(curO0 + 2 * strO0, ev[0]);
                lds_f32x4_put<C::X_OFF>(curO1 + 2 * strO1, ev[1]);
                if (!(amax < (float)kWsRangeLimit)) ovf = 1;
                __syncthreads();
            }
        } else {
            // ---- per-lane geometry: MFMA column u = 5 board + t (15 = padding), k-group lg; wave = transform point
            // (per group, behind the stem: kept alive through stem and heads these twelve registers spill) ----
            const int glane = fresh_lane(), gli = glane & 15, glg = glane >> 4, wlane = glane * 16;   // (per group: not to be hoisted out of the group loop)
            const int ub = gli / 5, ut = gli - 5 * ub;
            const bool uv = gli < 15;
            // the two cells of point `wave`: V = d[xa] + sgn d[xb]
            const int xa = wave == 0 ? 2 * ut - 1 : (wave == 2 ? 2 * ut + 1 : 2 * ut);
            const int xb = wave == 0 ? 2 * ut + 1 : (wave == 1 ? 2 * ut + 1 : (wave == 2 ? 2 * ut : 2 * ut + 2));
            auto cell = [&](int x, int chunk, int invalid_rel, int &adr, int &str) {
                const bool ok = uv && x >= 0 && x < 9;
                // (cells outside the board - and the padding column - keep the chunk their class would give them: the reads of a
                // ds_read_b128 cycle stay on sixteen distinct chunks, zero row included)
                const int g = (5 * ub + ((x + 1) >> 1)) & 7, sw = (g & 1) | ((g & 6) << 1);
                adr = (ok ? (81 * ub + x) * 256 : invalid_rel) + ((chunk ^ sw) << 4);
                str = ok ? 9 * 256 : 0;
            };
            int curA, strA, curB, strB, curO0, strO0, curO1, strO1, curR0, curR1, strR0, strR1;      // cursors (row 0) and row strides
            cell(xa, glg * 2, C::ZERO_REL, curA, strA);
            cell(xb, glg * 2, C::ZERO_REL, curB, strB);
            cell(2 * ut, wave * 4 + glg, C::DUMP_REL, curO0, strO0);        // stores of the channels 16 wave + 4 lg ..
            cell(2 * ut + 1, wave * 4 + glg, C::DUMP_REL, curO1, strO1);
            cell(2 * ut, wave * 4 + glg, C::ZERO_REL, curR0, strR0);        // residual reads (outside the board: zeros)
            cell(2 * ut + 1, wave * 4 + glg, C::ZERO_REL, curR1, strR1);
            f32x4 dq[2][2][2];                                     // cells read ahead: [cell a / b][kc][channel half]
            i32x4v vh[5][2], vl[5][2];                             // V rows: slot 4 = row 0, slot r & 3 = rows 1 .. 8; [kc]
            f32x4 acc[2][4];                                       // [row parity][channel tile]
            f32x4 ez[4], eres[2], ev[2];
            float tvv[4];
            unsigned thh[2];
            i32x4v ux[2][4];                                       // tap 2 / k-chunk 1 of ODD layers, [piece][ct] (even layers: ua[2][1]; see w1_sched)
            auto vslot = [](int r) constexpr { return r == 0 ? 4 : (r & 3); };
            auto rd = [&](auto IN_, auto I_) __attribute__((always_inline)) {          // one of the eight cell reads of the row at curA / curB
                constexpr int IN = decltype(IN_)::value, i = decltype(I_)::value, cb = i >> 2, kc = (i >> 1) & 1, h = i & 1;
                const int a0 = (cb ? curB : curA) ^ ((kc << 7) | (h << 4));
                if constexpr (!(W1_ABL & 4)) dq[cb][kc][h] = lds_f32x4_at<IN>(a0);
                if constexpr (i == 7) { curA += strA; curB += strB; }
            };
            // input transform of one row in 16 slices: per (kc, half) t = d_a + sgn d_b (2 x 2 values), high pieces, low pieces
            auto tr = [&](auto R_, auto I_) __attribute__((always_inline)) {
                constexpr int r = decltype(R_)::value, i = decltype(I_)::value, kc = i >> 3, h = (i >> 2) & 1, q = i & 3, s = vslot(r);
                if constexpr (W1_ABL & 2) {
                } else if constexpr (q == 0) {
                    tvv[0] = fmaf(dq[1][kc][h][0], sgn, dq[0][kc][h][0]);
                    tvv[1] = fmaf(dq[1][kc][h][1], sgn, dq[0][kc][h][1]);
                } else if constexpr (q == 1) {
                    tvv[2] = fmaf(dq[1][kc][h][2], sgn, dq[0][kc][h][2]);
                    tvv[3] = fmaf(dq[1][kc][h][3], sgn, dq[0][kc][h][3]);
                } else if constexpr (q == 2) {
                    thh[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{tvv[0], tvv[1]}, f16x2));
                    vh[s][kc][2 * h] = (int)thh[0];
                    vl[s][kc][2 * h] = (int)low_pieces(tvv[0], tvv[1], thh[0]);
                } else {
                    thh[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{tvv[2], tvv[3]}, f16x2));
                    vh[s][kc][2 * h + 1] = (int)thh[1];
                    vl[s][kc][2 * h + 1] = (int)low_pieces(tvv[2], tvv[3], thh[1]);
                }
            };
            f32x4 pshf = f32x4{0.f, 0.f, 0.f, 0.f};                // the previous layer's epilogue constants (its row 8 rides in this layer's row 0)
            float pdown = 0.f;
            int pO0 = C::DUMP_REL + (glane * 16) % 256, pO1 = pO0, pR0 = C::ZERO_REL + (glane * 16) % 256, pR1 = pR0;   // its row-8 cells (layer 0: dump / zero rows)
            // Schedule of a row's slices (one MFMA each + what rides along):  0-15 input transform of row y + 2 | 0-3 exchange
            // writes of row y - 1, 10 barrier, 11-12 exchange reads, 13 residual reads, 19-34 sums / shift / residual / ReLU,
            // 35-36 stores | 37-44 cell reads of row y + 3 | from 46: weight requests.  Row 8's exchange + epilogue ride in the
            // NEXT layer's row 0 (layer 0: a null epilogue - zero accumulators, zero constants, dump-row stores); the next layer's
            // V rows 0 and 1 are transformed under rows 7 and 8 (its input rows 0 - 2 are complete since row 3).  No barrier at
            // the layer boundary: between a store and any other wave's read of it lies at least one row barrier.
            auto layer_fn = [&](auto IN_, auto OUT_, auto RES_, int layer) __attribute__((always_inline)) {
                constexpr int IN = decltype(IN_)::value, OUT = decltype(OUT_)::value;
                constexpr bool RES = decltype(RES_)::value;
                constexpr int PAR = RES ? 1 : 0;                   // conv2 of a block = odd layer
                constexpr int S1 = PAR ? 3 : 1, S1N = PAR ? 1 : 3; // AGPR slot of tap ky = 1 in this / the next layer (taps 0, 2: slots 0, 2)
                const int next_layer = layer + 1 < kTowerLayers ? layer + 1 : 0;
                const unsigned char *wnext = net.w1_w + ((size_t)next_layer * 4 + wave) * 49152;
                const unsigned char *wcur = net.w1_w + ((size_t)layer * 4 + wave) * 49152;
                int exw = C::EX_OFF + wave * 4096 + glane * 16, exr = C::EX_OFF + wave * 1024 + glane * 16;
                asm volatile("" : "+v"(exw), "+v"(exr));
                if (layer == 0) {
                    // a group's first layer: nothing was prepared under a previous layer - V rows 0 and 1, the cells of row 2
                    // (cursors: set at the top of the group)
                    static_for<8>([&](auto I_) { rd(IN_, I_); });
                    static_for<16>([&](auto I_) { tr(std::integral_constant<int, 0>{}, I_); });
                    static_for<8>([&](auto I_) { rd(IN_, I_); });
                    static_for<16>([&](auto I_) { tr(std::integral_constant<int, 1>{}, I_); });
                    static_for<8>([&](auto I_) { rd(IN_, I_); });
                }
                // this layer's tap 1 must have arrived (behind its last request: the 23 of rows 6 - 8, w1_sched)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kW1WaitTop) : "memory");
                __builtin_amdgcn_sched_barrier(0);                 // (an MFMA is no memory operation: nothing else keeps it behind the wait)
                // shift and scale of this layer: requested HERE (two vector loads the wait counts below include), first used
                // by row 0's epilogue of row 1, i.e. behind the vmcnt(0) at the top of row 1
                const f32x4 shf = *reinterpret_cast<const f32x4 *>(net.w1_shift + layer * 64 + wave * 16 + glg * 4);
                float down;
                {
                    const float *dp = net.w1_down + layer;
                    int zoff = 0;
                    asm volatile("" : "+v"(zoff));                 // a VECTOR load whatever hipcc proves about the address: the count must not depend on it
                    asm volatile("global_load_dword %0, %1, %2" : "=v"(down) : "v"(zoff), "s"(dp) : "memory");
                }
                // exchange + epilogue of a row: PREV = row 8 of the previous layer (output buffer = this layer's input, the other
                // residual flag, constants pshf / pdown, cells at row 8), else row Y of this layer (cells at curO / curR)
                auto epi = [&](auto PREV_, auto Y_, auto I_) __attribute__((always_inline)) {
                    constexpr bool PREV = decltype(PREV_)::value;
                    constexpr int y = decltype(Y_)::value, i = decltype(I_)::value;
                    constexpr int par = PREV ? (8 + 1 - PAR) & 1 : (y + PAR) & 1;
                    constexpr int OB = PREV ? IN : OUT;
                    constexpr bool RS = PREV ? !RES : RES;
                    if constexpr (i < 4) {
                        if constexpr (!(W1_ABL & 16)) lds_f32x4_put<par * 16384 + i * 1024>(exw, acc[par][i]);
                    } else if constexpr (i == 13) {
                        if constexpr (RS) {
                            eres[0] = lds_f32x4_at<OB>(PREV ? pR0 : curR0);
                            eres[1] = lds_f32x4_at<OB>(PREV ? pR1 : curR1);
                        }
                        if constexpr (!PREV) { curR0 += strR0; curR1 += strR1; }   // (also without a residual: the next layer's deferred row needs them at row 8)
                    } else if constexpr (i == 10) {
                        if constexpr (!(W1_ABL & 32)) __syncthreads();
                    } else if constexpr (i == 11 || i == 12) {
                        if constexpr (!(W1_ABL & 16)) {
                        ez[2 * (i - 11)] = lds_f32x4_at<par * 16384 + (2 * (i - 11)) * 4096>(exr);
                        ez[2 * (i - 11) + 1] = lds_f32x4_at<par * 16384 + (2 * (i - 11) + 1) * 4096>(exr);
                        }
                    } else if constexpr (i >= 19 && i < 35) {
                        constexpr int k = i - 19, cc = k >> 3, e = (k >> 1) & 3, part = k & 1;
                        if constexpr (W1_ABL & 8) {
                        } else if constexpr (part == 0) {
                            ev[cc][e] = cc == 0 ? (ez[0][e] + ez[1][e]) + ez[2][e] : (ez[1][e] - ez[2][e]) - ez[3][e];
                        } else {
                            float tt = fmaf(ev[cc][e], PREV ? pdown : down, PREV ? pshf[e] : shf[e]);
                            if constexpr (RS) tt += eres[cc][e];
                            ev[cc][e] = fmaxf(tt, 0.f);
                        }
                    } else if constexpr (i == 35) {
                        amax = fmaxf(fmaxf(amax, ev[0][0]), ev[0][1]);
                        amax = fmaxf(fmaxf(amax, ev[0][2]), ev[0][3]);
                        lds_f32x4_put<OB>(PREV ? pO0 : curO0, ev[0]);
                        if constexpr (!PREV) curO0 += strO0;
                    } else if constexpr (i == 36) {
                        amax = fmaxf(fmaxf(amax, ev[1][0]), ev[1][1]);
                        amax = fmaxf(fmaxf(amax, ev[1][2]), ev[1][3]);
                        lds_f32x4_put<OB>(PREV ? pO1 : curO1, ev[1]);
                        if constexpr (!PREV) curO1 += strO1;
                    }
                };
                static_for<9>([&](auto Y_) {
                    constexpr int y = decltype(Y_)::value, par = (y + PAR) & 1;
                    constexpr int NT = (y == 0 || y == 8) ? 2 : 3, NM = 24 * NT, KY0 = y == 0 ? 1 : 0;
                    if constexpr (PROF)
                        if (blockIdx.x == 0 && wave == 0 && glane == 0 && (layer == 2 || layer == 3) && grp == (int)blockIdx.x)
                            net.timeline[40 + 12 * (layer - 2) + y] = (long long)__builtin_amdgcn_s_memtime();
                    // tap 0 - its last four fragments were requested in row 0 - and with it the layer's shift and scale
                    if constexpr (y == 1) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (y == 6) { curA -= 9 * strA; curB -= 9 * strB; }              // from here on: the next layer's rows 0 .. 2
                    static_for<NM>([&](auto M_) {
                        constexpr int m = decltype(M_)::value, ti = m / 24, q = m % 24, kc = q / 12, st = (q / 4) % 3, c = q % 4;
                        constexpr int ky = KY0 + ti, r = y + ky - 1, s = vslot(r), slot = ky == 1 ? S1 : ky;
                        constexpr bool UX = PAR == 1 && ky == 2 && kc == 1;      // odd layers: tap 2 / k-chunk 1 from the VGPR fragments
                        // tap 2 (row 0: the second tap): its last request was row 8's slice 9 of the previous layer
                        if constexpr (y == 0 && m == 24) {
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kW1WaitTap2) : "memory");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (W1_ABL & 128) {
                        } else if constexpr (st == 0)
                            acc[par][c] = mfma16<F>(UX ? ux[1][c] : ua[slot][kc][1][c], vh[s][kc], m < 4 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[par][c]);
                        else if constexpr (st == 1) acc[par][c] = mfma16<F>(UX ? ux[0][c] : ua[slot][kc][0][c], vl[s][kc], acc[par][c]);
                        else acc[par][c] = mfma16<F>(UX ? ux[0][c] : ua[slot][kc][0][c], vh[s][kc], acc[par][c]);
                        // ---- what rides along ----
                        if constexpr (m < 37) {
                            if constexpr (y == 0) epi(std::true_type{}, Y_, M_);
                            else epi(std::false_type{}, std::integral_constant<int, y - 1>{}, M_);
                        }
                        if constexpr (m < 16) {
                            if constexpr (y + 2 <= 8) tr(std::integral_constant<int, y + 2>{}, M_);
                            else tr(std::integral_constant<int, y - 7>{}, M_);                  // rows 7 / 8: the next layer's V rows 0 / 1
                        }
                        if constexpr (m >= 37 && m < 45) {
                            if constexpr (y + 3 <= 8) rd(IN_, std::integral_constant<int, m - 37>{});
                            else rd(OUT_, std::integral_constant<int, m - 37>{});                // rows 6 .. 8: the next layer's rows 0 .. 2
                        }
                        // weight requests (w1_sched): each behind the last use of the register it goes to
                        {
                            constexpr int code = w1_sched(y, m), kind = code >> 4;
                            using FR = std::integral_constant<int, (code & 15)>;
                            if constexpr (code >= 0 && !(W1_ABL & 1)) {
                                if constexpr (kind == 0) w1_request<S1N>(ua, wnext + 16384, wlane, FR{});
                                else if constexpr (kind == 1) {
                                    // (the next layer is odd when this one is even: its tap 2 / k-chunk 1 goes to the VGPR fragments)
                                    if constexpr (PAR == 0) w1_request_v(ux[(FR::value >> 2) & 1][FR::value & 3], wnext + 2 * 16384, wlane, FR{});
                                    else w1_request<2>(ua, wnext + 2 * 16384, wlane, FR{});
                                } else if constexpr (kind == 2) w1_request<2>(ua, wnext + 2 * 16384, wlane, FR{});
                                else if constexpr (kind == 3) w1_request<0>(ua, wnext, wlane, FR{});
                                else w1_request<0>(ua, wcur, wlane, FR{});
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
                if constexpr (PROF)
                    if (blockIdx.x == 0 && wave == 0 && glane == 0 && (layer == 2 || layer == 3) && grp == (int)blockIdx.x)
                        net.timeline[40 + 12 * (layer - 2) + 9] = (long long)__builtin_amdgcn_s_memtime();
                pshf = shf;
                pdown = down;
                // the cursors stand at row 8 (eight rows stored): that is where the deferred epilogue goes; back to row 0 for the next layer
                pO0 = curO0; pO1 = curO1; pR0 = curR0; pR1 = curR1;
                curO0 -= 8 * strO0; curO1 -= 8 * strO1; curR0 -= 8 * strR0; curR1 -= 8 * strR1;
                if (!(amax < (float)kWsRangeLimit)) ovf = 1;            // f16 range guard (also catches NaN)
                stamp();
            };
            using IX = std::integral_constant<int, C::X_OFF>;
            using IH = std::integral_constant<int, C::H_OFF>;
            if (!(amax < (float)kWsRangeLimit)) ovf = 1;
            // row 8's cells (the deferred epilogue) and layer 0's null epilogue
            {
                float z0, z1, z2, z3;
                asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z0), "=v"(z1), "=v"(z2), "=v"(z3));
                pshf = f32x4{z0, z1, z2, z3};
                pdown = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[1][c] = pshf;
            }
#pragma unroll 1
            for (int blk = 0; blk < kBlocks; ++blk) {
                layer_fn(IX{}, IH{}, std::false_type{}, 2 * blk);
                layer_fn(IH{}, IX{}, std::true_type{}, 2 * blk + 1);
            }
            {
                // the tower's last row (layer 11, output X, residual): on its own
                const int exw = C::EX_OFF + wave * 4096 + glane * 16, exr = C::EX_OFF + wave * 1024 + glane * 16;
                constexpr int par = (8 + 1) & 1;
                static_for<4>([&](auto I_) { lds_f32x4_put<par * 16384 + decltype(I_)::value * 1024>(exw, acc[par][decltype(I_)::value]); });
                eres[0] = lds_f32x4_at<C::X_OFF>(pR0);
                eres[1] = lds_f32x4_at<C::X_OFF>(pR1);
                __syncthreads();
                static_for<4>([&](auto I_) { ez[decltype(I_)::value] = lds_f32x4_at<par * 16384 + decltype(I_)::value * 4096>(exr); });
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o0 = (ez[0][e] + ez[1][e]) + ez[2][e], o1 = (ez[1][e] - ez[2][e]) - ez[3][e];
                    ev[0][e] = fmaxf(fmaf(o0, pdown, pshf[e]) + eres[0][e], 0.f);
                    ev[1][e] = fmaxf(fmaf(o1, pdown, pshf[e]) + eres[1][e], 0.f);
                    amax = fmaxf(fmaxf(amax, ev[0][e]), ev[1][e]);
                }
                lds_f32x4_put<C::X_OFF>(pO0, ev[0]);
                lds_f32x4_put<C::X_OFF>(pO1, ev[1]);
                if (!(amax < (float)kWsRangeLimit)) ovf = 1;
                __syncthreads();
            }
        }
        // a group that left the f16 range says so: the exact kernel behind this launch redoes the marked groups only
        if (group_bits && !(amax < (float)kWsRangeLimit)) atomicOr(group_bits + (grp >> 5), 1 << (grp & 31));
        // next group's input planes: requested here, consumed after the heads
        const int next = __builtin_amdgcn_readfirstlane(*ticket_lds);   // (written before the stem's barriers)
        fetch_planes(next);
        if constexpr (!(W1_ABL & 256))
        run_heads_x32<G, C, (G == 1 ? 2 : 1)>(smem, net, b0, batch, want_logits, policy, value, wave * 64 + fresh_lane(), wave,
                                              PROF && blockIdx.x == 0 && grp == (int)blockIdx.x ? net.timeline + 64 : nullptr);   // [64..66]: 1x1 convolutions done, barrier passed, FCs done
        __syncthreads();
        stamp();
        grp = next;
    }
    if (ovf && overflow) atomicOr(overflow, 1);
}

template <int G, bool PROF = false>
int launch_w1d(tg_net *net, const float *planes, int batch, int want_logits, float *policy, float *value,
               int *overflow, int *group_bits, hipStream_t stream) {
    using C = WsCfg<G>;
    if (!PROF && net->dev.timeline)
        return launch_w1d<G, true>(net, planes, batch, want_logits, policy, value, overflow, group_bits, stream);
    auto kern = dualnet_fwd_w1d_kernel<G, PROF>;
    static std::atomic<uint64_t> configured{0};
    if (tg::first_on_device(configured, net->device))
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    const int groups = (batch + G - 1) / G;
    int grid = groups < net->num_cus ? groups : net->num_cus;
    if (const int cap = tg::launch_caps().forward; cap > 0 && grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHR), C::LDS_BYTES, stream, net->dev, planes, batch, want_logits,
                       policy, value, overflow, group_bits);
    TG_HIP(hipGetLastError());
    return TG_OK;
}

}  // namespace

namespace tg {

// dualnet_fwd_w1d_kernel's weights: U_p[ky] = (G g[ky])_p along kx - p0 = g0, p1 = (g0 + g1 + g2) / 2, p2 = (g0 - g1 + g2) / 2,
// p3 = g2 - in fp64 with the batch-norm scale folded in, x 2^e per layer (largest entry into [2^9, 2^10)), pieces hi = rn16(u),
// lo = rn16(u - hi) UNSCALED; folded shift [12][64]; 2^-e [12].  tower[l]: [64][64][3][3]; scale / shift: folded batch norm
// [13][64] (index 0 = stem).  The image does not depend on the board size.
int w1d_prepare(tg_net *net, const float *const *tower, const float *scale, const float *shift) {
    std::vector<uint16_t> img((size_t)12 * 4 * 3 * 2 * 2 * 4 * 512);
    std::vector<float> down(12), shf(12 * 64);
    std::vector<double> u((size_t)4 * 3 * 64 * 64);
    for (int layer = 0; layer < 12; ++layer) {
        const float *w = tower[layer];
        double mx = 0.0;
        for (int cout = 0; cout < 64; ++cout)
            for (int cin = 0; cin < 64; ++cin) {
                const float *g = &w[((size_t)cout * 64 + cin) * 9];
                const double sc = scale[(layer + 1) * 64 + cout];
                for (int ky = 0; ky < 3; ++ky) {
                    const double g0 = g[ky * 3], g1 = g[ky * 3 + 1], g2 = g[ky * 3 + 2];
                    const double pt[4] = {g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2};
                    for (int p = 0; p < 4; ++p) {
                        const double v = pt[p] * sc;
                        u[(((size_t)p * 3 + ky) * 64 + cin) * 64 + cout] = v;
                        mx = std::fmax(mx, std::fabs(v));
                    }
                }
            }
        int e = 0;
        if (mx > 0.0 && std::isfinite(mx)) {
            int ex;
            std::frexp(mx, &ex);
            e = 10 - ex;
        }
        e = e > 40 ? 40 : (e < -40 ? -40 : e);
        down[layer] = std::ldexp(1.f, -e);
        for (int c = 0; c < 64; ++c) shf[layer * 64 + c] = shift[(layer + 1) * 64 + c];
        for (int p = 0; p < 4; ++p)
            for (int ky = 0; ky < 3; ++ky)
                for (int kc = 0; kc < 2; ++kc)
                    for (int ct = 0; ct < 4; ++ct)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int el = 0; el < 8; ++el) {
                                const int cout = ct * 16 + (lane & 15), cin = kc * 32 + (lane >> 4) * 8 + el;
                                const double v = std::ldexp(u[(((size_t)p * 3 + ky) * 64 + cin) * 64 + cout], e);
                                const uint16_t h = f32_to_f16_rn((float)v);
                                const uint16_t l = f32_to_f16_rn((float)(v - (double)f16_to_f32(h)));
                                const size_t frag = ((((size_t)layer * 4 + p) * 3 + ky) * 2 + kc) * 2;
                                img[((frag + 0) * 4 + ct) * 512 + lane * 8 + el] = h;
                                img[((frag + 1) * 4 + ct) * 512 + lane * 8 + el] = l;
                            }
    }
    auto up = [&](const void *src, size_t bytes, const void **dst) {
        void *d = nullptr;
        TG_HIP(hipMalloc(&d, bytes));
        net->allocs.push_back(d);
        TG_HIP(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
        *dst = d;
        return (int)TG_OK;
    };
    int rc;
    if ((rc = up(img.data(), img.size() * 2, reinterpret_cast<const void **>(&net->dev.w1_w))) ||
        (rc = up(shf.data(), shf.size() * 4, reinterpret_cast<const void **>(&net->dev.w1_shift))) ||
        (rc = up(down.data(), down.size() * 4, reinterpret_cast<const void **>(&net->dev.w1_down))))
        return rc;
    return TG_OK;
}

int w1d_forward(tg_net *net, int group, const float *planes, int batch, int want_logits, float *policy, float *value, int *overflow,
                int *group_bits, hipStream_t stream) {
    if (net->board_size != 9) return tg::fail(TG_ERR_ARG, "w1d forward: 9x9 only");
    if (group == 3) return launch_w1d<3>(net, planes, batch, want_logits, policy, value, overflow, group_bits, stream);
    return launch_w1d<1>(net, planes, batch, want_logits, policy, value, overflow, group_bits, stream);
}

}  // namespace tg
