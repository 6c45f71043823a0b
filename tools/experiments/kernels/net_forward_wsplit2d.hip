// DualNet forward for gfx950, 9x9: Winograd F(2x2,3x3) residual tower ON SPLIT OPERANDS (round 4).
//
// net_forward_split.hip runs the fp32 3x3 convolutions on the 16-bit matrix pipe as three f16 products per fp32
// product (a = ah + al, w = wh + wl:  a w ~ ah wh + ah wl + al wh) - and is bound by the MFMA count: the pipe is
// power- and issue-capped, and 228.6 MFLOP of MFMAs are issued per position for 72.3 MFLOP of
// algorithm.  This kernel cuts the count: per 2x2 output tile Y = A^T [ (G g G^T) . (B^T d B) ] A, i.e. 16
// transform points x (25 tiles per board) instead of 9 taps x 81 positions = 1.82x fewer MFMA rows; 127.9 MFLOP
// issued per position.  What makes it work (measured first: tools/microbench/wino_issue_model.hip,
// profiles/r04_microbench_wino_issue_model.txt; tools/experiments/winograd_split_accuracy.py):
//   * V = B^T d B is computed in fp32 (adds only, exact inputs) and split AFTERWARDS into two f16 pieces; the weights
//     U = G g G^T are computed in fp64 on the host with the batch-norm scale folded in, scaled by a power of two per
//     layer and split there.  Logit error against the reference's fp64 forward: the same class as the direct split
//     kernel and the reference's own fp32 path.
//   * the LOW pieces are kept UNSCALED (al = rn16(a - ah), no 2^11): v_mfma_f32_16x16x32_f16 keeps f16 subnormals
//     (probe in the micro-benchmark), so the cross terms have the right magnitude by themselves and ONE accumulator
//     set takes all three products (K = 64 per Winograd point: six MFMAs per accumulator).
//   * a Winograd point's GEMM is tiny (tiles x 64 x 64), so the operands decide the structure: wave w of the four
//     owns POINT ROW w (four points, all 64 output channels, all tiles).  Its 64 weight fragments (64 KB) are
//     loaded once per layer and stay in registers; it computes only its own row of the input transform (no wave
//     repeats another's VALU work - the loop is VALU-bound: beside an MFMA stream every VALU instruction beyond
//     two per MFMA costs ~3.4 cycles); the transform along the point row's own axis is done in registers, and the
//     sum over the four point rows goes through a 32 KB LDS exchange: Z = M A (two values per row instead of
//     four), then wave w' finishes output channels [16 w', 16 w' + 16): folded shift, residual, ReLU, store.
//   * activations stay in LDS as fp32 [position][64 channels] (two buffers: block input X - also the residual -
//     and the intermediate H), 16-byte chunk index XOR-swizzled by a function of the position under which the
//     4 x 4 patch reads of sixteen tiles are conflict-free for every patch cell; the tiles are assigned to lanes
//     so that this holds (host: ws_geometry).  All per-lane LDS addresses of a row tile come from two small
//     tables in global memory (the geometry is the same for every layer and workgroup): no address arithmetic in
//     the loop.
// Stem (6 -> 64 channels) and heads are the direct split kernel's (im2col'ed K = 64 product; 1x1 convolutions and
// policy FC on the 16-bit pipe), reading / writing the fp32 images.  f16 range guard as there: |V| <= 4 |d| must stay
// below 65504, so a layer output beyond 16000 raises the flag and the exact-fp32 kernel redoes the batch.
// Reference: nn/network/res_block.py:8-38, nn/network/dual_net.py:41-52.
#include "split_common.h"

namespace {

constexpr int kWsRangeLimit = 16000;

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

// 16-byte chunk XOR of activation row R (position 81 b + 9 y + x): g = ((y + 1) / 2 + 5 ((x + 1) / 2) + b) mod 8, spread
// over chunk-index bits 0, 2, 3 (bit 1 is the one in which the two lane groups of a ds_read_b128 cycle differ)
__host__ __device__ inline int ws_swz(int R) {
    const int b = R >= 162 ? 2 : (R >= 81 ? 1 : 0);
    const int p = R - 81 * b, y = (p * 57) >> 9, x = p - 9 * y;
    const int g = (((y + 1) >> 1) + 5 * ((x + 1) >> 1) + b) & 7;
    return (g & 1) | ((g & 6) << 1);
}

// ... and of the kernel that transforms along x only (dualnet_fwd_w1d_kernel): its sixteen MFMA columns are the units
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

// LDS accesses by ABSOLUTE LDS byte address (the kernel has no static LDS: the dynamic array starts at 0, checked at
// kernel start).  Through `smem + addr` every access costs a v_add_u32 with the array's (relocatable) base.
typedef __attribute__((address_space(3))) f32x4 lds_f32x4_t;
template <int OFF>
__device__ __forceinline__ f32x4 lds_f32x4_at(int addr) {
    return *reinterpret_cast<const lds_f32x4_t *>(static_cast<unsigned>(addr + OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_f32x4_put(int addr, f32x4 v) {
    *reinterpret_cast<lds_f32x4_t *>(static_cast<unsigned>(addr + OFF)) = v;
}

// The lane id, computed where it is asked for: hipcc treats the mbcnt pair as a pure value, computes it once at the top of a
// persistent kernel and - with the register file full of weight fragments - keeps it in scratch, one exposed reload per use.
// A volatile asm is neither hoisted nor merged: two instructions per phase instead.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// "These values are used here": hipcc waits for a load it tracks where the value is first used, and with vmcnt(0) - it does
// not see the weight requests of the inline asm, so the wait must sit where none of them is in flight.  (A free function:
// clang rejects asm operands that name captured variables inside a generic lambda.)
__device__ __forceinline__ void use_here(f32x4 &v, float &s) {
    asm volatile("" : "+v"(v), "+v"(s));
}

// Low pieces of two values whose high pieces are packed in h: f16(v0 - h.lo) | f16(v1 - h.hi) << 16, i.e. v_fma_mixlo_f16 /
// v_fma_mixhi_f16 with the f16 halves of h as source 0, -1.0 as source 1 and the fp32 value as source 2: the difference is
// exact in fp32 and rounded once.  (hipcc does not form these from C: it emits v_cvt_f32_f16 + v_sub_f32 + v_cvt_pk_f16_f32.)
__device__ __forceinline__ unsigned low_pieces(float v0, float v1, unsigned h) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(v1));
    return r;
}
// four fp32 values -> two packed registers of high pieces, two of UNSCALED low pieces (6 VALU instructions)
__device__ __forceinline__ void split4_unscaled(const f32x4 v, unsigned (&hi)[2], unsigned (&lo)[2]) {
    hi[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{v[0], v[1]}, f16x2));
    hi[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{v[2], v[3]}, f16x2));
    lo[0] = low_pieces(v[0], v[1], hi[0]);
    lo[1] = low_pieces(v[2], v[3], hi[1]);
}

// Request the 32 weight fragments of k-chunk KC of a wave's layer block wb ([j 4][kc 2][piece 2][ct 4][lane][16 B]) into
// AGPRs.  (A free function: clang rejects asm operands that name captured variables inside a generic lambda.)
template <int KC>
__device__ __forceinline__ void ws_load_w(i32x4v (&ua)[4][2][2][4], const unsigned char *wb, int wlane) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const unsigned char *base = wb + ((j * 2 + KC) * 2 + p) * 4096;
            asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                         "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                         "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                         "global_load_dwordx4 %3, %4, %5 offset:3072"
                         : "=a"(ua[j][KC][p][0]), "=a"(ua[j][KC][p][1]), "=a"(ua[j][KC][p][2]), "=a"(ua[j][KC][p][3])
                         : "v"(wlane), "s"(base)
                         : "memory");
        }
}

// The same for ONE point J of k-chunk KC (eight fragments: two asm statements), or only its channel tile ct of both pieces
// (two fragments): the requests are placed between the instructions of the layer's last row tile.
template <int KC, int J>
__device__ __forceinline__ void ws_load_w_point(i32x4v (&ua)[4][2][2][4], const unsigned char *wb, int wlane, int ct = -1) {
    if (ct < 0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const unsigned char *base = wb + ((J * 2 + KC) * 2 + p) * 4096;
            asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                         "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                         "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                         "global_load_dwordx4 %3, %4, %5 offset:3072"
                         : "=a"(ua[J][KC][p][0]), "=a"(ua[J][KC][p][1]), "=a"(ua[J][KC][p][2]), "=a"(ua[J][KC][p][3])
                         : "v"(wlane), "s"(base)
                         : "memory");
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c == ct) {
                const unsigned char *base = wb + ((J * 2 + KC) * 2) * 4096 + c * 1024;
                asm volatile("global_load_dwordx4 %0, %2, %3\n\t"
                             "global_load_dwordx4 %1, %2, %4"
                             : "=a"(ua[J][KC][0][c]), "=a"(ua[J][KC][1][c])
                             : "v"(wlane), "s"(base), "s"(base + 4096)
                             : "memory");
            }
    }
}

// dualnet_fwd_w1d_kernel: fragment F = 8 kc + 4 piece + ct of a tap block ([kc 2][piece 2][ct 4][lane][16 B]) into AGPR slot SLOT
template <int SLOT, int F>
__device__ __forceinline__ void w1_request(i32x4v (&ua)[4][2][2][4], const unsigned char *tapbase, int wlane, std::integral_constant<int, F>) {
    constexpr int kc = F >> 3, p = (F >> 2) & 1, ct = F & 3;
    const unsigned char *base = tapbase + kc * 8192 + p * 4096;
    if constexpr (ct == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(ua[SLOT][kc][p][0]) : "v"(wlane), "s"(base) : "memory");
    else if constexpr (ct == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=a"(ua[SLOT][kc][p][1]) : "v"(wlane), "s"(base) : "memory");
    else if constexpr (ct == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=a"(ua[SLOT][kc][p][2]) : "v"(wlane), "s"(base) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=a"(ua[SLOT][kc][p][3]) : "v"(wlane), "s"(base) : "memory");
}
template <int SLOT>
__device__ __forceinline__ void w1_request_tap(i32x4v (&ua)[4][2][2][4], const unsigned char *tapbase, int wlane) {
    static_for<16>([&](auto F_) { w1_request<SLOT>(ua, tapbase, wlane, F_); });
}

// Heads on the 16-bit matrix pipe (split_common.h: run_heads_mfma), reading the block output from the fp32 image X:
// a B fragment (position li of a 16-row tile, channels 32 kc + 8 lg ..) is two 16-byte reads + the operand split.
template <int G, typename C, int SWZ = 0>
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
        const int sw = row < M ? (SWZ == 2 ? w1g1_swz(row) : (SWZ ? w1_swz(row) : ws_swz(row))) : 0;
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

// PROF: s_memtime stamps of workgroup 0 / wave 0 (tg_net_profile_phases): [0] group start, [1] input staged, [2] stem done,
// [3..14] layer done, [15] heads done; [40 + 35 (layer - 2) + 7 rt + i] for layers 2 and 3: i = 0 row tile start,
// 1 phase A done, 2 phase B done, 3 tail done + barrier passed, 4 (last row tile: own epilogue done)
template <int G, bool PROF>
__global__ __launch_bounds__(256, 1) void dualnet_fwd_wsplit_kernel(
    NetDev net, const float *__restrict__ planes, int batch, int want_logits,
    float *__restrict__ policy, float *__restrict__ value, int *__restrict__ overflow) {
    using C = WsCfg<G>;
    using F = FmtF16;
    constexpr int P = C::P, M = C::M, NTHR = C::NTHR, NRT = C::NRT, RTW = C::RTW, IMG = C::IMG;
    constexpr int GI = G == 3 ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;

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
            if (blockIdx.x == 0 && tid == 0 && stamp_i < 40) net.timeline[stamp_i++] = (long long)__builtin_amdgcn_s_memtime();
    };
    int dstamp_base = -1;                                      // detailed stamps of the current layer, or -1
    auto dstamp = [&](int rt, int i) {
        if constexpr (PROF)
            if (blockIdx.x == 0 && tid == 0 && dstamp_base >= 0 && stamp_i < 40)
                net.timeline[dstamp_base + 7 * rt + i] = (long long)__builtin_amdgcn_s_memtime();
    };
    int ovf = 0;
    const int n_groups = (batch + G - 1) / G;
    constexpr int NPL = (G * 6 * P + NTHR - 1) / NTHR;
    float pre[NPL];
    auto fetch_planes = [&](int grp2) __attribute__((always_inline)) {
        int ft = tid;
        asm volatile("" : "+v"(ft));
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
    const int *const tin_w = net.ws_tin[GI] + (size_t)wave * NRT * 64 * 8;
    const int *const tout_w = net.ws_tout[GI] + (size_t)wave * NRT * 64 * 8;
    // LDS address tables of the row tile about to be processed (carried across layers: the geometry repeats; fetched per
    // group behind the stem - kept alive through stem and heads they cost 16 spilled registers, and the scratch lines, 125 us
    // apart, came back from HBM: 3.6 KB per position of fabric traffic for nothing)
    i32x4v ta, tb, to, tr, to2, tr2, ta1, tb1;
    // This wave's 64 weight fragments of a layer, [j 4][kc 2][piece 2][ct 4]: resident in the accumulation half of the
    // register file for the whole layer (MFMA A operands are read from there directly).  They are requested by inline
    // asm with AGPR destinations - left to the register allocator they end up in VGPRs, spilled to AGPRs and copied back
    // before every use - so hipcc does not track them: the waits are explicit (body) and nothing may copy these registers
    // between a request and its wait (checked in the ISA).  All requests of a k-chunk go out in the layer BEFORE, in its
    // last row tile, as soon as the chunk's own MFMAs are done; layer 0's at kernel start and in layer 11's last row tile.
    i32x4v ua[4][2][2][4];
    const int wlane = lane * 16;
    auto load_w = [&](auto KC_, int layer) __attribute__((always_inline)) {
        ws_load_w<decltype(KC_)::value>(ua, net.ws_w + ((size_t)layer * 4 + wave) * 65536, wlane);
    };
    load_w(std::integral_constant<int, 0>{}, 0);
    load_w(std::integral_constant<int, 1>{}, 0);

    // Groups beyond a workgroup's first are handed out by a ticket counter (overflow[1], zeroed with the range flag): a
    // workgroup that starts late - its CU was running another stream's tree kernel - takes fewer groups instead of
    // holding the launch up with a full static share.  The ticket travels through a spare word of the bias table.
    int *const ticket_lds = reinterpret_cast<int *>(smem + C::HB_OFF + 83 * 4);
    for (int grp = blockIdx.x; grp < n_groups;) {
        const int b0 = grp * G;
        if (tid == 0) *ticket_lds = overflow ? (int)gridDim.x + atomicAdd(overflow + 1, 1) : grp + (int)gridDim.x;
        stamp();
        // ================= stem: planes -> im2col'ed f16-pair images (K = 9 taps x 6 planes, padded to 64) =================
        // (its 16 weight fragments are requested first: their L2 round trip runs under the staging pass)
        i32x4v fa[2][2][4];                                      // [kc][piece][ct]
        {
            int wvg = lane * 16;
            asm volatile("" : "+v"(wvg));
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
            int stid = tid;
            asm volatile("" : "+v"(stid));
#pragma unroll
            for (int i = 0; i < NPL; ++i)
                if (stid + i * NTHR < G * 6 * P) st[stid + i * NTHR] = pre[i];
            for (int e = stid; e < 4 * 64; e += NTHR)           // zero blocks of the four images
                reinterpret_cast<unsigned *>(smem + C::SI_OFF + (e >> 6) * IMG + C::ZOFF)[e & 63] = 0u;
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
#pragma unroll
            for (int r = 0; r < RTW; ++r) {
                int row = (wave * RTW + r) * 16 + li;
                asm volatile("" : "+v"(row));
                const int nat = row * 64 + ((lg ^ ((row >> 1) & 3)) << 4);
                const int addr = C::SI_OFF + (row < M ? nat : C::ZOFF + (nat & 255));
                i32x4v fb[2][2];                                 // [piece][kc]
                lds_load_frag<0 * IMG>(fb[0][0], smem, addr);
                lds_load_frag<1 * IMG>(fb[0][1], smem, addr);
                lds_load_frag<2 * IMG>(fb[1][0], smem, addr);
                lds_load_frag<3 * IMG>(fb[1][1], smem, addr);
                const int orow = row < M ? row : M;
                const int osw = row < M ? ws_swz(row) : 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
                        a0 = mfma16<F>(fa[kc][0][c], fb[0][kc], a0);
                        a1 = mfma16<F>(fa[kc][1][c], fb[0][kc], a1);
                        a1 = mfma16<F>(fa[kc][0][c], fb[1][kc], a1);
                    }
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(net.sscale + c * 16 + lg * 4);
                    const f32x4 sh = *reinterpret_cast<const f32x4 *>(net.shift + c * 16 + lg * 4);
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = fmaf(a1[j], 1.f / 2048.f, a0[j]);
                        t = fmaf(t, sc[j], sh[j]);
                        v[j] = fmaxf(t, 0.f);
                    }
                    amax = fmaxf(fmaxf(amax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                    *reinterpret_cast<f32x4 *>(smem + C::X_OFF + orow * 256 + (((c * 4 + lg) ^ osw) << 4)) = v;
                }
            }
        }
        __syncthreads();                                        // X complete; the overlay is free again
        if (tid < 64) reinterpret_cast<float *>(smem + C::H_OFF + (M + 1) * 256)[tid] = 0.f;   // H's zero row was under it
        stamp();

        // ================= tower: 12 Winograd layers =================
        // A row tile is two MFMA phases and a short tail (IN / OUT = byte offsets of the input / output buffer, RES: add the
        // residual from OUT):
        //   A  the 48 MFMAs of k-chunk 0 | input transform of k-chunk 1 (its patch cells were read a phase ago) | the
        //      EPILOGUE OF THE PREVIOUS ROW TILE (exchange reads, sum over the point rows, shift, residual, ReLU, stores) |
        //      patch reads of the next row tile's k-chunk 0
        //   B  the 48 MFMAs of k-chunk 1 | input transform of the next row tile's k-chunk 0 | output transform along the point
        //      row as far as the finished accumulators allow (Z0 complete and written behind point 2) | patch reads of the
        //      next row tile's k-chunk 1
        //   tail  Z1 = (m1 - m2) - m3, written; barrier
        // A and B are written slice by slice - one MFMA and the instructions that ride along - with a scheduling barrier
        // behind each slice: an MFMA occupies the pipe for 16 cycles and the wave issues two to three other instructions
        // meanwhile; everything beyond that costs its issue time, but no longer its LATENCY (LDS round trips, the exchange's
        // write bandwidth, barriers waiting for stragglers), which is what the un-overlapped version paid
        // (profiles/r04_phase_wsplit_v1.txt: 4.8 k cycles per row tile; r04_phase_wsplit_v2_pipelined.txt: 4.3 k with the
        // transform under the MFMAs and 2 k of those in the exchange + epilogue).  The layer's last row tile has no next
        // row tile to prepare (the next layer's input is still being written): its slices carry the REQUESTS for the next
        // layer's weight fragments instead - a k-chunk's 32 registers are dead once its MFMAs are issued -, its epilogue runs
        // on its own, and the next layer starts with one un-overlapped transform.
        f32x4 dq[2][4][2];                                     // patch cells read ahead: [row a / b][column s][channel half]
        i32x4v bh0[4], bl0[4];                                 // operand pieces of (row tile, k-chunk 0), built a phase ahead
        auto read_cell = [&](auto IN_, auto KC_, auto C8_, const i32x4v &pa, const i32x4v &pb) __attribute__((always_inline)) {
            constexpr int IN = decltype(IN_)::value, kc = decltype(KC_)::value, c8 = decltype(C8_)::value;
            const int a0 = (c8 < 4 ? pa[c8] : pb[c8 - 4]) ^ (kc << 7);
            dq[c8 >> 2][c8 & 3][0] = lds_f32x4_at<IN>(a0);
            dq[c8 >> 2][c8 & 3][1] = lds_f32x4_at<IN>(a0 ^ 16);
        };
        // the input transform of one k-chunk in 48 slices: 0..15 row pass t = d[ra] + sgn d[rb] (two values each), 16..47 per
        // (point j, channel half h) four slices of two to three instructions: column pass, high pieces, low pieces
        f32x4 tq[4][2];
        float tv[8][4];
        unsigned thi[8][2];
        auto tslice = [&](auto I_, i32x4v (&oh)[4], i32x4v (&ol)[4]) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value;
            if constexpr (i < 16) {
                constexpr int s_ = i >> 2, h = (i >> 1) & 1, e0 = (i & 1) * 2;
                tq[s_][h][e0] = fmaf(dq[1][s_][h][e0], sgn, dq[0][s_][h][e0]);
                tq[s_][h][e0 + 1] = fmaf(dq[1][s_][h][e0 + 1], sgn, dq[0][s_][h][e0 + 1]);
            } else {
                constexpr int k = (i - 16) >> 2, q = (i - 16) & 3, j = k >> 1, h = k & 1;
                auto col = [&](int e) __attribute__((always_inline)) {
                    return j == 0 ? tq[0][h][e] - tq[2][h][e] : (j == 1 ? tq[1][h][e] + tq[2][h][e]
                         : (j == 2 ? tq[2][h][e] - tq[1][h][e] : tq[1][h][e] - tq[3][h][e]));
                };
                if constexpr (q == 0) {
                    tv[k][0] = col(0); tv[k][1] = col(1); tv[k][2] = col(2);
                } else if constexpr (q == 1) {
                    tv[k][3] = col(3);
                    thi[k][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{tv[k][0], tv[k][1]}, f16x2));
                    thi[k][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{tv[k][2], tv[k][3]}, f16x2));
                } else if constexpr (q == 2) {
                    oh[j][2 * h] = (int)thi[k][0];
                    ol[j][2 * h] = (int)low_pieces(tv[k][0], tv[k][1], thi[k][0]);
                } else {
                    oh[j][2 * h + 1] = (int)thi[k][1];
                    ol[j][2 * h + 1] = (int)low_pieces(tv[k][2], tv[k][3], thi[k][1]);
                }
            }
        };
        // MFMA m of a k-chunk: point j = m / 12, product (m / 4) % 3 (cross terms first), channel tile m % 4
        f32x4 acc[4][4];
        auto mfma_slice = [&](auto KC_, auto M_, const i32x4v (&ph)[4], const i32x4v (&pl)[4]) __attribute__((always_inline)) {
            constexpr int kc = decltype(KC_)::value, m = decltype(M_)::value, j = m / 12, st = (m / 4) % 3, c = m % 4;
            if constexpr (st == 0)
                acc[j][c] = mfma16<F>(ua[j][kc][1][c], ph[j], kc == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[j][c]);
            else if constexpr (st == 1) acc[j][c] = mfma16<F>(ua[j][kc][0][c], pl[j], acc[j][c]);
            else acc[j][c] = mfma16<F>(ua[j][kc][0][c], ph[j], acc[j][c]);
        };
        // The epilogue of one row tile, output q = 2 r + c at (2 ty + r, 2 tx + c), in six steps: 0 the three exchange reads
        // (+ the residual), 1..4 one channel each (sum over the point rows, shift, residual, ReLU), 5 range check + store.
        // Output channels 16 wave + 4 lg ..; po / pr: the row tile's store / residual address tables.
        f32x4 ez[4][3], eres[4], ev[4];
        auto epi_step = [&](auto OUT_, auto RES_, auto Q_, auto I_, const i32x4v &po, const i32x4v &pr, const f32x4 shf, const float down,
                            int exr) __attribute__((always_inline)) {
            constexpr int OUT = decltype(OUT_)::value, q = decltype(Q_)::value, i = decltype(I_)::value, r = q >> 1, cc = q & 1;
            constexpr bool RES = decltype(RES_)::value;
            if constexpr (i == 0) {
                static_for<3>([&](auto U_) {                       // point rows r .. r + 2
                    constexpr int u = decltype(U_)::value;
                    ez[q][u] = lds_f32x4_at<((r + u) * 2 + cc) * 4096>(exr);
                });
                if constexpr (RES) eres[q] = lds_f32x4_at<OUT>(pr[q]);
            } else if constexpr (i <= 4) {
                constexpr int e = i - 1;
                const float y = r == 0 ? (ez[q][0][e] + ez[q][1][e]) + ez[q][2][e] : (ez[q][0][e] - ez[q][1][e]) - ez[q][2][e];
                float tt = fmaf(y, down, shf[e]);
                if constexpr (RES) tt += eres[q][e];
                ev[q][e] = fmaxf(tt, 0.f);
            } else {
                amax = fmaxf(fmaxf(amax, ev[q][0]), ev[q][1]);        // (v_max3_f32)
                amax = fmaxf(fmaxf(amax, ev[q][2]), ev[q][3]);
                lds_f32x4_put<OUT>(po[q], ev[q]);
            }
        };
        // output transform along the point row, riding along phase B: zs01 = m0 + m1 (behind point 1), Z0 = zs01 + m2 and
        // zd12 = m1 - m2 (behind point 2; Z0 written), Z1 = zd12 - m3 in the tail
        f32x4 zs01[4], zd12[4];
        constexpr bool DEFER = G == 3;                         // the last row tile's tail + epilogue ride in the next layer (see body)
        f32x4 pshf = f32x4{0.f, 0.f, 0.f, 0.f};                // the previous layer's epilogue constants
        float pdown = 0.f;
        auto ztail = [&](auto C_, int exw) __attribute__((always_inline)) {
            constexpr int c = decltype(C_)::value;
            f32x4 z1;
#pragma unroll
            for (int e = 0; e < 4; ++e) z1[e] = zd12[c][e] - acc[3][c][e];
            lds_f32x4_put<4096 + c * 1024>(exw, z1);
        };
        auto body = [&](auto IN_, auto OUT_, auto RES_, auto FIRST_, auto LAST_, int rt, int next_layer, const f32x4 shf,
                        const float down) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(FIRST_)::value, LAST = decltype(LAST_)::value, RES = decltype(RES_)::value;
            dstamp(rt, 0);
            const unsigned char *wnext = net.ws_w + ((size_t)next_layer * 4 + wave) * 65536;
            i32x4v bh1[4], bl1[4];
            // exchange addresses (absolute): this wave's block / its channel tile.  (The exchange lies beyond the 64 KB an
            // LDS instruction's offset field reaches: with the base as an immediate every access cost a v_add_u32.)
            int exw = C::EX_OFF + wave * 8192 + lane * 16, exr = C::EX_OFF + wave * 1024 + lane * 16;
            asm volatile("" : "+v"(exw), "+v"(exr));
            // (last row tile) the tables the code behind the weight requests needs are fetched here and waited for in slice 12
            // of phase A, before the first request goes out: hipcc's own counted waits know nothing of the asm requests, so a
            // wait for any load of its own that is older than requests in flight would drain those as well.  ta1 / tb1: row
            // tile 1's patch table (next layer); to2 / tr2: this row tile's store / residual table (its epilogue runs last)
            if constexpr (LAST) {
                ta1 = *reinterpret_cast<const i32x4v *>(tin_w + ((size_t)64 + lane) * 8);
                tb1 = *reinterpret_cast<const i32x4v *>(tin_w + ((size_t)64 + lane) * 8 + 4);
                to2 = *reinterpret_cast<const i32x4v *>(tout_w + ((size_t)(NRT - 1) * 64 + lane) * 8);
                tr2 = *reinterpret_cast<const i32x4v *>(tout_w + ((size_t)(NRT - 1) * 64 + lane) * 8 + 4);
            }
            // ---- phase A ----
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");     // this layer's k-chunk 0 fragments (requested a layer ago; the
                                                                  // 32 requests behind them - k-chunk 1's - may still be in flight)
            __builtin_amdgcn_sched_barrier(0);                    // (an MFMA is no memory operation: nothing else keeps it behind the wait)
            static_for<48>([&](auto M_) {
                constexpr int m = decltype(M_)::value;
                mfma_slice(std::integral_constant<int, 0>{}, M_, bh0, bl0);
                tslice(M_, bh1, bl1);
                // previous row tile's epilogue: output q's exchange reads in slice 6 q, its five compute / store steps in slices
                // 6 q + 7 .. 6 q + 11 - a whole output later, so that the LDS round trip (150+ cycles with four waves on the
                // LDS) is over when the values are needed (next slice: +500 cycles per phase, profiles/r04_phase_wsplit_v4.txt)
                // ... behind the previous row tile's Z1 = (m1 - m2) - m3 (slices 0 .. 3: point 3 finished with phase B's last
                // MFMA) and the barrier that publishes the exchange (slice 5): write latency and stragglers cost MFMA slots that
                // are filled anyway instead of a tail of their own
                // (three boards per workgroup: a layer's FIRST row tile carries the PREVIOUS LAYER's last row tile the same way -
                // that row tile holds tiles of the third board only, row tiles 0 and 1 tiles of the first two (ws_geometry), so
                // nothing this layer reads before the next barriers is written by it; its output buffer is this layer's
                // input, its residual flag the opposite, its constants pshf / pdown, its tables to2 / tr2)
                constexpr bool EPI = !FIRST || DEFER;
                if constexpr (!FIRST && m < 4) ztail(std::integral_constant<int, m>{}, exw);
                if constexpr (!FIRST && m == 5) __syncthreads();   // (the deferred row tile's Z1 went out before the layer's closing barrier)
                if constexpr (EPI && m >= 6 && m < 30 && m % 6 == 0) {
                    if constexpr (FIRST) epi_step(IN_, std::integral_constant<bool, !RES>{}, std::integral_constant<int, m / 6 - 1>{}, std::integral_constant<int, 0>{}, to2, tr2, pshf, pdown, exr);
                    else epi_step(OUT_, RES_, std::integral_constant<int, m / 6 - 1>{}, std::integral_constant<int, 0>{}, to, tr, shf, down, exr);
                }
                if constexpr (EPI && m >= 13 && m < 36 && (m - 12) % 6 != 0) {
                    if constexpr (FIRST) epi_step(IN_, std::integral_constant<bool, !RES>{}, std::integral_constant<int, (m - 12) / 6>{}, std::integral_constant<int, (m - 12) % 6>{}, to2, tr2, pshf, pdown, exr);
                    else epi_step(OUT_, RES_, std::integral_constant<int, (m - 12) / 6>{}, std::integral_constant<int, (m - 12) % 6>{}, to, tr, shf, down, exr);
                }
                if constexpr (!LAST && m >= 16 && m % 4 == 0)      // dq's old contents are dead behind slice 15
                    read_cell(IN_, std::integral_constant<int, 0>{}, std::integral_constant<int, (m - 16) / 4>{}, ta, tb);
                // last row tile: point j's k-chunk 0 fragments are dead behind slice 12 j + 11 - request the next layer's, two
                // fragments (one channel tile) every third slice (a request costs the wave ~40 cycles of issue: the CU's
                // address path takes 16 cycles per wave instruction and the four waves request at the same time)
                if constexpr (LAST && m == 12) asm volatile("" : "+v"(ta1), "+v"(tb1), "+v"(to2), "+v"(tr2));
                if constexpr (LAST && m >= 12 && m % 3 == 0)
                    ws_load_w_point<0, (m - 12) / 12>(ua, wnext, wlane, ((m - 12) % 12) / 3);
                __builtin_amdgcn_sched_barrier(0);
            });
            dstamp(rt, 1);
            // ---- phase B ----
            // k-chunk 1's fragments (only a layer's first row tile can wait here; in the last one requests are in flight
            // already, and everything it still needs has been waited for)
            if constexpr (!LAST) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            // this row tile's store / residual table (the previous one's is dead), for its epilogue in the next phase A
            if constexpr (!LAST) {
                to = *reinterpret_cast<const i32x4v *>(tout_w + ((size_t)rt * 64 + lane) * 8);
                tr = *reinterpret_cast<const i32x4v *>(tout_w + ((size_t)rt * 64 + lane) * 8 + 4);
            }
            // every wave has read the exchange (previous epilogue): a bare s_barrier - this wave's exchange reads have
            // returned (their values were consumed slices ago), and __syncthreads() would also drain the patch reads just
            // issued for the next row tile (s_waitcnt lgkmcnt(0): ~200 cycles per row tile)
            __builtin_amdgcn_s_barrier();
            static_for<48>([&](auto M_) {
                constexpr int m = decltype(M_)::value;
                mfma_slice(std::integral_constant<int, 1>{}, M_, bh1, bl1);
                if constexpr (!LAST) {
                    tslice(M_, bh0, bl0);
                    if constexpr (m >= 16 && m % 4 == 0)
                        read_cell(IN_, std::integral_constant<int, 1>{}, std::integral_constant<int, (m - 16) / 4>{}, ta, tb);
                } else {
                    // point 3's k-chunk 0 fragments (dead since phase A's last slice), then k-chunk 1's of points 0 .. 2 as
                    // their MFMAs are done
                    if constexpr (m < 12 && m % 3 == 0) ws_load_w_point<0, 3>(ua, wnext, wlane, m / 3);
                    if constexpr (m >= 12 && m % 3 == 0) ws_load_w_point<1, (m - 12) / 12>(ua, wnext, wlane, ((m - 12) % 12) / 3);
                }
                if constexpr (m >= 26 && m < 30) {                 // points 0 and 1 are complete behind slice 23
                    constexpr int c = m - 26;
#pragma unroll
                    for (int e = 0; e < 4; ++e) zs01[c][e] = acc[0][c][e] + acc[1][c][e];
                }
                if constexpr (m >= 38 && m < 46) {                 // point 2 behind slice 35
                    constexpr int c = (m - 38) >> 1;
                    if constexpr (((m - 38) & 1) == 0) {
                        f32x4 z0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) z0[e] = zs01[c][e] + acc[2][c][e];
                        lds_f32x4_put<c * 1024>(exw, z0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) zd12[c][e] = acc[1][c][e] - acc[2][c][e];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            dstamp(rt, 2);
            // next row tile's patch tables are dead now; fetch the ones after (cyclic over the row tiles - the geometry
            // repeats in every layer)
            if constexpr (!LAST) {
                const int r2 = rt + 2 < NRT ? rt + 2 : rt + 2 - NRT;
                ta = *reinterpret_cast<const i32x4v *>(tin_w + ((size_t)r2 * 64 + lane) * 8);
                tb = *reinterpret_cast<const i32x4v *>(tin_w + ((size_t)r2 * 64 + lane) * 8 + 4);
            }
            // ---- tail: Z1 = (m1 - m2) - m3 -> exchange: rides along the NEXT row tile's phase A; the last row tile's here ----
            if constexpr (LAST) {
                static_for<4>([&](auto C_) {
                    ztail(C_, exw);
                    ws_load_w_point<1, 3>(ua, wnext, wlane, decltype(C_)::value);    // point 3's k-chunk 1 registers are dead now
                });
                if constexpr (!DEFER) __syncthreads();
            }
            dstamp(rt, 3);
            if constexpr (LAST && !DEFER) {
                // ---- the last row tile's own epilogue (nothing to hide it behind), the remaining requests in between ----
                static_for<4>([&](auto Q_) {
                    static_for<6>([&](auto I_) { epi_step(OUT_, RES_, Q_, I_, to2, tr2, shf, down, exr); });
                });
            }
            dstamp(rt, 4);
        };
        auto conv = [&](auto IN_, auto OUT_, auto RES_, int layer) __attribute__((always_inline)) {
            // epilogue constants of the output channels this wave finishes: 16 wave + 4 lg ..
            const f32x4 shf = *reinterpret_cast<const f32x4 *>(net.ws_shift + layer * 64 + wave * 16 + lg * 4);
            const float down = net.ws_down[layer];
            const int next_layer = layer + 1 < kTowerLayers ? layer + 1 : 0;
            dstamp_base = (layer == 2 || layer == 3) && grp == (int)blockIdx.x ? 40 + 35 * (layer - 2) : -1;
            // prologue: row tile 0's k-chunk 0 transformed on its own, k-chunk 1's cells requested; then ta / tb = row tile 1's
            static_for<8>([&](auto C8_) { read_cell(IN_, std::integral_constant<int, 0>{}, C8_, ta, tb); });
            static_for<48>([&](auto I_) { tslice(I_, bh0, bl0); });
            static_for<8>([&](auto C8_) { read_cell(IN_, std::integral_constant<int, 1>{}, C8_, ta, tb); });
            ta = ta1;
            tb = tb1;
            // (row tile 0 peeled: inside the loop hipcc must assume that ta / tb are loads of the previous iteration and waits
            // for them - in row tile 0 that wait would drain the weight requests still in flight)
            body(IN_, OUT_, RES_, std::true_type{}, std::false_type{}, 0, next_layer, shf, down);
#pragma unroll 1
            for (int rt = 1; rt < NRT - 1; ++rt) body(IN_, OUT_, RES_, std::false_type{}, std::false_type{}, rt, next_layer, shf, down);
            body(IN_, OUT_, RES_, std::false_type{}, std::true_type{}, NRT - 1, next_layer, shf, down);
            // (ta / tb hold row tile 0's table again: the last fetch of the loop wrapped around)
            pshf = shf;
            pdown = down;
            if (!(amax < (float)kWsRangeLimit)) ovf = 1;        // f16 range guard (also catches NaN)
            __syncthreads();                                    // OUT complete (but a deferred last row tile) before the next layer reads it
            stamp();
        };
        using IX = std::integral_constant<int, C::X_OFF>;
        using IH = std::integral_constant<int, C::H_OFF>;
        if (!(amax < (float)kWsRangeLimit)) ovf = 1;
        ta = *reinterpret_cast<const i32x4v *>(tin_w + (size_t)lane * 8);
        tb = *reinterpret_cast<const i32x4v *>(tin_w + (size_t)lane * 8 + 4);
        to = *reinterpret_cast<const i32x4v *>(tout_w + (size_t)lane * 8);            // store / residual tables: see body
        tr = *reinterpret_cast<const i32x4v *>(tout_w + (size_t)lane * 8 + 4);
        ta1 = *reinterpret_cast<const i32x4v *>(tin_w + ((size_t)64 + lane) * 8);     // row tile 1's (see body)
        tb1 = *reinterpret_cast<const i32x4v *>(tin_w + ((size_t)64 + lane) * 8 + 4);
        asm volatile("" :: "v"(ta), "v"(tb), "v"(to), "v"(tr), "v"(ta1), "v"(tb1));  // waited for here (layer 0's weight requests, in flight, are needed now anyway)
        to2 = to;
        tr2 = tr;
        if constexpr (DEFER) {
            // layer 0 has no previous layer: its first row tile carries a NULL epilogue - zero exchange, zero constants,
            // every store to the dump row, every residual read from the zero row (cheaper than a third copy of the layer code)
            {
                // (zeros made HERE: hipcc hoists a constant vector out of the group loop and, short of registers, spills it)
                float z0, z1, z2, z3;
                asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z0), "=v"(z1), "=v"(z2), "=v"(z3));
                pshf = f32x4{z0, z1, z2, z3};
            }
            pdown = 0.f;
            for (int e = tid; e < C::EX_BYTES / 16; e += NTHR) reinterpret_cast<f32x4 *>(smem + C::EX_OFF)[e] = pshf;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                to2[q] = C::DUMP_REL + (lane * 16) % 256;
                tr2[q] = C::ZERO_REL + (lane * 16) % 256;
            }
            __syncthreads();
        }
#pragma unroll 1
        for (int blk = 0; blk < kBlocks; ++blk) {
            conv(IX{}, IH{}, std::false_type{}, 2 * blk);
            conv(IH{}, IX{}, std::true_type{}, 2 * blk + 1);
        }
        if constexpr (DEFER) {
            // the tower's last row tile (layer 11, output buffer X, residual): nothing left to hide it behind
            int exr = C::EX_OFF + wave * 1024 + lane * 16;
            asm volatile("" : "+v"(exr));
            static_for<4>([&](auto Q_) {
                static_for<6>([&](auto I_) { epi_step(IX{}, std::true_type{}, Q_, I_, to2, tr2, pshf, pdown, exr); });
            });
            if (!(amax < (float)kWsRangeLimit)) ovf = 1;
            __syncthreads();
        }
        // next group's input planes: requested here, consumed after the heads
        const int next = __builtin_amdgcn_readfirstlane(*ticket_lds);   // (written before the stem's barriers)
        fetch_planes(next);
        run_heads_x32<G, C>(smem, net, b0, batch, want_logits, policy, value, tid, wave, nullptr);
        __syncthreads();
        stamp();
        grp = next;
    }
    if (ovf && overflow) atomicOr(overflow, 1);
}

// The 9x9 tower as Winograd F(2,3) along x only (TG_FWD_ALGO=w1d): 2-D Winograd (above) issues the fewest MFMAs but 5.3 VALU
// instructions beside each and is bound by instruction issue; the direct kernel needs no transform and is bound by the matrix
// pipe.  One transformed axis sits between: 600 MFMAs per wave and layer for three boards (2-D: 480, direct: 857), one
// transform pass per side - 1.4 VALU instructions per MFMA, inside what an MFMA's 16 cycles hide.  Three boards per workgroup
// only (smaller launches take dualnet_fwd_wsplit_kernel<1>).  Stem, heads, operand pieces, range guard: as above.
// PROF: s_memtime stamps of workgroup 0 / wave 0: [0] group start, [1] input staged, [2] stem done, [3..14] layer done, [15] heads done, [64..66] inside the heads: 1x1 convolutions done, barrier passed, FCs done
template <int G, bool PROF>
__global__ __launch_bounds__(256, 1) void dualnet_fwd_w1d_kernel(
    NetDev net, const float *__restrict__ planes, int batch, int want_logits,
    float *__restrict__ policy, float *__restrict__ value, int *__restrict__ overflow) {
    using C = WsCfg<G>;
    using F = FmtF16;
    constexpr int P = C::P, M = C::M, NTHR = C::NTHR, NRT = C::NRT, RTW = C::RTW, IMG = C::IMG;
    constexpr int GI = G == 3 ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;

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
        if constexpr (G == 3) {                                  // (the order the first layer waits for them in)
            w1_request_tap<2>(ua, w0 + 2 * 16384, wlane);
            w1_request_tap<0>(ua, w0, wlane);
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
                f32x4 shf = *reinterpret_cast<const f32x4 *>(net.ws_shift + layer * 64 + wave * 16 + glg * 4);
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
            auto vslot = [](int r) constexpr { return r == 0 ? 4 : (r & 3); };
            auto rd = [&](auto IN_, auto I_) __attribute__((always_inline)) {          // one of the eight cell reads of the row at curA / curB
                constexpr int IN = decltype(IN_)::value, i = decltype(I_)::value, cb = i >> 2, kc = (i >> 1) & 1, h = i & 1;
                const int a0 = (cb ? curB : curA) ^ ((kc << 7) | (h << 4));
                dq[cb][kc][h] = lds_f32x4_at<IN>(a0);
                if constexpr (i == 7) { curA += strA; curB += strB; }
            };
            // input transform of one row in 16 slices: per (kc, half) t = d_a + sgn d_b (2 x 2 values), high pieces, low pieces
            auto tr = [&](auto R_, auto I_) __attribute__((always_inline)) {
                constexpr int r = decltype(R_)::value, i = decltype(I_)::value, kc = i >> 3, h = (i >> 2) & 1, q = i & 3, s = vslot(r);
                if constexpr (q == 0) {
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
                const f32x4 shf = *reinterpret_cast<const f32x4 *>(net.ws_shift + layer * 64 + wave * 16 + glg * 4);
                const float down = net.w1_down[layer];
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
                // this layer's taps 1 and 2 must have arrived (requested in that order; behind them: tap 0's 16 requests, the shift)
                asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);                 // (an MFMA is no memory operation: nothing else keeps it behind the wait)
                // exchange + epilogue of a row: PREV = row 8 of the previous layer (output buffer = this layer's input, the other
                // residual flag, constants pshf / pdown, cells at row 8), else row Y of this layer (cells at curO / curR)
                auto epi = [&](auto PREV_, auto Y_, auto I_) __attribute__((always_inline)) {
                    constexpr bool PREV = decltype(PREV_)::value;
                    constexpr int y = decltype(Y_)::value, i = decltype(I_)::value;
                    constexpr int par = PREV ? (8 + 1 - PAR) & 1 : (y + PAR) & 1;
                    constexpr int OB = PREV ? IN : OUT;
                    constexpr bool RS = PREV ? !RES : RES;
                    if constexpr (i < 4) {
                        lds_f32x4_put<par * 16384 + i * 1024>(exw, acc[par][i]);
                    } else if constexpr (i == 13) {
                        if constexpr (RS) {
                            eres[0] = lds_f32x4_at<OB>(PREV ? pR0 : curR0);
                            eres[1] = lds_f32x4_at<OB>(PREV ? pR1 : curR1);
                        }
                        if constexpr (!PREV) { curR0 += strR0; curR1 += strR1; }   // (also without a residual: the next layer's deferred row needs them at row 8)
                    } else if constexpr (i == 10) {
                        __syncthreads();
                    } else if constexpr (i == 11 || i == 12) {
                        ez[2 * (i - 11)] = lds_f32x4_at<par * 16384 + (2 * (i - 11)) * 4096>(exr);
                        ez[2 * (i - 11) + 1] = lds_f32x4_at<par * 16384 + (2 * (i - 11) + 1) * 4096>(exr);
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
                    // tap 0; behind it the layer's shift and scale (requested at the layer top, 48 MFMAs ago: hipcc waits for them
                    // where the epilogue first uses them, 19 slices on)
                    if constexpr (y == 1) {
                        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (y == 6) { curA -= 9 * strA; curB -= 9 * strB; }              // from here on: the next layer's rows 0 .. 2
                    static_for<NM>([&](auto M_) {
                        constexpr int m = decltype(M_)::value, ti = m / 24, q = m % 24, kc = q / 12, st = (q / 4) % 3, c = q % 4;
                        constexpr int ky = KY0 + ti, r = y + ky - 1, s = vslot(r), slot = ky == 1 ? S1 : ky;
                        if constexpr (st == 0)
                            acc[par][c] = mfma16<F>(ua[slot][kc][1][c], vh[s][kc], m < 4 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[par][c]);
                        else if constexpr (st == 1) acc[par][c] = mfma16<F>(ua[slot][kc][0][c], vl[s][kc], acc[par][c]);
                        else acc[par][c] = mfma16<F>(ua[slot][kc][0][c], vh[s][kc], acc[par][c]);
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
                        // next layer's weights: tap 1 into the spare slot during rows 4 .. 6, tap 2 at the start of row 8 (its last use
                        // was row 7), tap 0 behind row 8's tap-0 MFMAs
                        if constexpr (y >= 4 && y <= 6 && m >= 46 && m < 64 && (m - 46) % 3 == 0) {
                            constexpr int f = (y - 4) * 6 + (m - 46) / 3;                      // 0 .. 17
                            if constexpr (f < 16) w1_request<S1N>(ua, wnext + 16384, wlane, std::integral_constant<int, f>{});
                        }
                        if constexpr (y == 8 && m < 32 && m % 2 == 1)
                            w1_request<2>(ua, wnext + 2 * 16384, wlane, std::integral_constant<int, m / 2>{});
                        if constexpr (y == 8 && m >= 32) w1_request<0>(ua, wnext, wlane, std::integral_constant<int, m - 32>{});
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
        // next group's input planes: requested here, consumed after the heads
        const int next = __builtin_amdgcn_readfirstlane(*ticket_lds);   // (written before the stem's barriers)
        fetch_planes(next);
        run_heads_x32<G, C, (G == 1 ? 2 : 1)>(smem, net, b0, batch, want_logits, policy, value, wave * 64 + fresh_lane(), wave,
                                              PROF && blockIdx.x == 0 && grp == (int)blockIdx.x ? net.timeline + 64 : nullptr);   // [64..66]: 1x1 convolutions done, barrier passed, FCs done
        __syncthreads();
        stamp();
        grp = next;
    }
    if (ovf && overflow) atomicOr(overflow, 1);
}

template <int G, bool PROF = false>
int launch_wsplit(tg_net *net, const float *planes, int batch, int want_logits, float *policy, float *value,
                  int *overflow, hipStream_t stream) {
    using C = WsCfg<G>;
    if (!PROF && net->dev.timeline)
        return launch_wsplit<G, true>(net, planes, batch, want_logits, policy, value, overflow, stream);
    auto kern = dualnet_fwd_wsplit_kernel<G, PROF>;
    static std::atomic<uint64_t> configured{0};
    if (tg::first_on_device(configured, net->device))
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    const int groups = (batch + G - 1) / G;
    int grid = groups < net->num_cus ? groups : net->num_cus;
    if (const int cap = net->forward_grid_cap.load(); cap > 0 && grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHR), C::LDS_BYTES, stream, net->dev, planes, batch, want_logits,
                       policy, value, overflow);
    TG_HIP(hipGetLastError());
    return TG_OK;
}

template <int G, bool PROF = false>
int launch_w1d(tg_net *net, const float *planes, int batch, int want_logits, float *policy, float *value,
               int *overflow, hipStream_t stream) {
    using C = WsCfg<G>;
    if (!PROF && net->dev.timeline)
        return launch_w1d<G, true>(net, planes, batch, want_logits, policy, value, overflow, stream);
    auto kern = dualnet_fwd_w1d_kernel<G, PROF>;
    static std::atomic<uint64_t> configured{0};
    if (tg::first_on_device(configured, net->device))
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    const int groups = (batch + G - 1) / G;
    int grid = groups < net->num_cus ? groups : net->num_cus;
    if (const int cap = net->forward_grid_cap.load(); cap > 0 && grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHR), C::LDS_BYTES, stream, net->dev, planes, batch, want_logits,
                       policy, value, overflow);
    TG_HIP(hipGetLastError());
    return TG_OK;
}

// ---- host: tile -> lane assignment and the LDS address tables ---------------------------------------------------
// Tile (b, ty, tx) has class h = (ty + 5 tx + b) mod 8 - the value ws_swz's g takes on its patch cell (r, s), up to a
// constant that depends on the cell only.  A ds_read_b128 serves lanes {li 0-3, 12-15 of lane group a} + {li 4-11 of
// lane group a ^ 1} in one LDS cycle: with the eight tiles of each of the two sets in eight different classes the
// sixteen 16-byte chunks are distinct for every cell.  No class has more than 2 NRT tiles, so: k-th tile of class h ->
// set k (row tile k / 2, half k % 2), lane position h inside the set.
template <int G>
void ws_geometry(std::vector<int> &tin, std::vector<int> &tout) {
    using C = WsCfg<G>;
    constexpr int NRT = C::NRT;
    std::vector<int> slot_tile(NRT * 16, -1);
    int cnt[8] = {};
    for (int b = 0; b < G; ++b)
        for (int ty = 0; ty < 5; ++ty)
            for (int tx = 0; tx < 5; ++tx) {
                const int h = (ty + 5 * tx + b) & 7, k = cnt[h]++;
                const int rt = k >> 1, li = (k & 1) ? 4 + h : (h < 4 ? h : h + 8);
                slot_tile[rt * 16 + li] = (b * 5 + ty) * 5 + tx;
            }
    static const int rows[4][2] = {{0, 2}, {1, 2}, {2, 1}, {1, 3}};
    tin.assign((size_t)4 * NRT * 64 * 8, 0);
    tout.assign((size_t)4 * NRT * 64 * 8, 0);
    for (int w = 0; w < 4; ++w)
        for (int rt = 0; rt < NRT; ++rt)
            for (int lane = 0; lane < 64; ++lane) {
                const int li = lane & 15, lg = lane >> 4, tile = slot_tile[rt * 16 + li];
                const int b = tile < 0 ? 0 : tile / 25, ty = tile < 0 ? 0 : (tile / 5) % 5, tx = tile < 0 ? 0 : tile % 5;
                int *ti = &tin[(((size_t)w * NRT + rt) * 64 + lane) * 8];
                for (int c8 = 0; c8 < 8; ++c8) {
                    const int r = rows[w][c8 >> 2], s = c8 & 3;
                    const int y = 2 * ty - 1 + r, x = 2 * tx - 1 + s;
                    // chunk index of channels 8 lg .. (k-chunk 0, first half): 2 lg; the kernel XORs in 8 kc + half
                    if (tile >= 0 && y >= 0 && y < 9 && x >= 0 && x < 9) {
                        const int R = b * 81 + y * 9 + x;
                        ti[c8] = R * 256 + (((lg * 2) ^ ws_swz(R)) << 4);
                    } else {
                        // zero row, at the chunk its natural address would have had (class of the cell: the same formula
                        // extended beyond the board; a tile-less lane takes the class of its lane position)
                        const int hcls = tile >= 0 ? (ty + 5 * tx + b) & 7 : (li < 4 ? li : (li < 12 ? li - 4 : li - 8));
                        const int g = (hcls + (r >> 1) + 5 * (s >> 1)) & 7;
                        const int sw = (g & 1) | ((g & 6) << 1);
                        ti[c8] = C::ZERO_REL + (((lg * 2) ^ sw) << 4);
                    }
                }
                // epilogue of output channels 16 w + 4 lg ..: chunk 4 w + lg.  [0..3] store addresses of outputs
                // (2 ty + r, 2 tx + c), q = 2 r + c (outside the board: dump row); [4..7] residual read addresses (outside: zero row)
                int *to = &tout[(((size_t)w * NRT + rt) * 64 + lane) * 8];
                for (int q = 0; q < 4; ++q) {
                    const int y = 2 * ty + (q >> 1), x = 2 * tx + (q & 1);
                    if (tile >= 0 && y < 9 && x < 9) {
                        const int R = b * 81 + y * 9 + x;
                        to[q] = to[4 + q] = R * 256 + (((w * 4 + lg) ^ ws_swz(R)) << 4);
                    } else {
                        to[q] = C::DUMP_REL + lane * 16 % 256;
                        to[4 + q] = C::ZERO_REL + lane * 16 % 256;
                    }
                }
            }
}

}  // namespace

namespace tg {

// Winograd weight image [layer 12][wave = point row i 4][j 4][kc 2][piece 2][ct 4][lane 64][8 x f16]: U = G g G^T in
// fp64, batch-norm scale folded in, x 2^e (largest entry of the layer into [2^9, 2^10)), pieces hi = rn16(u),
// lo = rn16(u - hi) UNSCALED; shift table [12][64]; 2^-e [12]; the address tables of both workgroup shapes.
// tower[l]: [64][64][3][3]; scale / shift: folded batch norm [13][64] (index 0 = stem).
int wsplit_prepare(tg_net *net, const float *const *tower, const float *scale, const float *shift) {
    std::vector<uint16_t> img((size_t)12 * 4 * 64 * 512, 0);
    std::vector<float> down(12), shf(12 * 64);
    std::vector<double> u((size_t)16 * 64 * 64);
    for (int layer = 0; layer < 12; ++layer) {
        const float *w = tower[layer];
        double mx = 0.0;
        for (int cout = 0; cout < 64; ++cout)
            for (int cin = 0; cin < 64; ++cin) {
                const float *g = &w[((size_t)cout * 64 + cin) * 9];
                double gg[4][3];
                for (int k = 0; k < 3; ++k) {
                    gg[0][k] = g[k];
                    gg[1][k] = 0.5 * ((double)g[k] + g[3 + k] + g[6 + k]);
                    gg[2][k] = 0.5 * ((double)g[k] - g[3 + k] + g[6 + k]);
                    gg[3][k] = g[6 + k];
                }
                const double sc = scale[(layer + 1) * 64 + cout];
                for (int a = 0; a < 4; ++a) {
                    const double row[4] = {gg[a][0], 0.5 * (gg[a][0] + gg[a][1] + gg[a][2]),
                                           0.5 * (gg[a][0] - gg[a][1] + gg[a][2]), gg[a][2]};
                    for (int bq = 0; bq < 4; ++bq) {
                        const double v = row[bq] * sc;
                        u[((size_t)(a * 4 + bq) * 64 + cin) * 64 + cout] = v;
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
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                for (int kc = 0; kc < 2; ++kc)
                    for (int ct = 0; ct < 4; ++ct)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int el = 0; el < 8; ++el) {
                                const int cout = ct * 16 + (lane & 15), cin = kc * 32 + (lane >> 4) * 8 + el;
                                const double v = std::ldexp(u[((size_t)(i * 4 + j) * 64 + cin) * 64 + cout], e);
                                const uint16_t h = f32_to_f16_rn((float)v);
                                const uint16_t l = f32_to_f16_rn((float)(v - (double)f16_to_f32(h)));
                                const size_t frag = ((((size_t)layer * 4 + i) * 4 + j) * 2 + kc) * 2;
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
    std::vector<int> tin1, tout1, tin3, tout3;
    ws_geometry<1>(tin1, tout1);
    ws_geometry<3>(tin3, tout3);
    int rc;
    if ((rc = up(img.data(), img.size() * 2, reinterpret_cast<const void **>(&net->dev.ws_w))) ||
        (rc = up(shf.data(), shf.size() * 4, reinterpret_cast<const void **>(&net->dev.ws_shift))) ||
        (rc = up(down.data(), down.size() * 4, reinterpret_cast<const void **>(&net->dev.ws_down))) ||
        (rc = up(tin1.data(), tin1.size() * 4, reinterpret_cast<const void **>(&net->dev.ws_tin[0]))) ||
        (rc = up(tout1.data(), tout1.size() * 4, reinterpret_cast<const void **>(&net->dev.ws_tout[0]))) ||
        (rc = up(tin3.data(), tin3.size() * 4, reinterpret_cast<const void **>(&net->dev.ws_tin[1]))) ||
        (rc = up(tout3.data(), tout3.size() * 4, reinterpret_cast<const void **>(&net->dev.ws_tout[1]))))
        return rc;
    return TG_OK;
}

// dualnet_fwd_w1d_kernel's weights: U_p[ky] = (G g[ky])_p along kx - p0 = g0, p1 = (g0 + g1 + g2) / 2, p2 = (g0 - g1 + g2) / 2,
// p3 = g2 - in fp64 with the batch-norm scale folded in, x 2^e per layer, pieces as in wsplit_prepare (low pieces unscaled).
int w1d_prepare(tg_net *net, const float *const *tower, const float *scale) {
    if (net->board_size != 9) return TG_OK;
    std::vector<uint16_t> img((size_t)12 * 4 * 3 * 2 * 2 * 4 * 512);
    std::vector<float> down(12);
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
        (rc = up(down.data(), down.size() * 4, reinterpret_cast<const void **>(&net->dev.w1_down))))
        return rc;
    return TG_OK;
}

int w1d_forward(tg_net *net, int group, const float *planes, int batch, int want_logits, float *policy, float *value, int *overflow,
                hipStream_t stream) {
    if (net->board_size != 9) return tg::fail(TG_ERR_ARG, "w1d forward: 9x9 only");
    if (group == 3) return launch_w1d<3>(net, planes, batch, want_logits, policy, value, overflow, stream);
    return launch_w1d<1>(net, planes, batch, want_logits, policy, value, overflow, stream);
}

// group = boards per workgroup (1 or 3); 9x9 only.
int wsplit_forward(tg_net *net, int group, const float *planes, int batch, int want_logits, float *policy,
                   float *value, int *overflow, hipStream_t stream) {
    if (net->board_size != 9) return tg::fail(TG_ERR_ARG, "winograd split forward: 9x9 only");
    if (group == 3) return launch_wsplit<3>(net, planes, batch, want_logits, policy, value, overflow, stream);
    return launch_wsplit<1>(net, planes, batch, want_logits, policy, value, overflow, stream);
}

}  // namespace tg
