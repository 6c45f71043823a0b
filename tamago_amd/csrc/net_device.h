// Device-side pieces shared by the DualNet forward kernels (net_forward.hip: exact-fp32 MFMA kernels,
// net_forward_split.hip: split-operand f16 / bf16 MFMA kernel) and the network handle they hang off.
#pragma once
#include "common.h"

#include <atomic>
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlocks = 6;
constexpr int kConvLayers = 1 + 2 * kBlocks;  // 13
constexpr int kRowFloats = 72;                // activation row stride in LDS (64 + 8 pad)
constexpr int kRowBytes = kRowFloats * 4;     // 288
// Winograd kernel: lanes of a fragment read patches of CONSECUTIVE TILES (2 positions apart);
// a row stride of 68 floats spreads 8 consecutive tiles over all 64 banks (72 would give 4)
constexpr int kWinoRowFloats = 68;
constexpr int kWinoRowBytes = kWinoRowFloats * 4;   // 272

struct NetDev {
    const float *w0frag;  // [4 wave][9 tap][64 lane][2]         stem 6->64 (cin padded to 8)
    const float *wfrag;   // [12 layer][4 wave][9 tap][4 s][64 lane][4]
    const float *wwino;   // [12 layer][4 wave][4 s][16 xi][64 lane][4]  Winograd G g G^T, fragment order
                          // (the 16 fragments of a work unit are 16 contiguous KB: two address bases reach them
                          //  all through the +-4 KB immediate of global_load)
    const float *scale;   // [13][64] folded BN scale
    const float *shift;   // [13][64] folded BN shift
    const float *hp_w;    // [2][64]  policy 1x1 conv
    const float *hv_w;    // [64]     value 1x1 conv
    const float *head_ss; // [6]      policy scale0,shift0,scale1,shift1, value scale,shift
    const float *pfc_wT;  // [2P][A]  policy FC, transposed
    const float *pfc_b;   // [A]
    const float *vfc_w;   // [3][P]
    const float *vfc_b;   // [3]
    // split-operand kernel (net_forward_split.hip): per global tap g (0 = stem as one K = 64 pseudo-tap,
    // 1 + 9*layer + tap for the tower) an LDS-ready image [k-chunk 2][piece][cout tile 4][lane 64][8 x 16 bit]
    const unsigned char *wsplit;      // f16 x 2 pieces
    const float *sscale;              // [13][64] folded BN scale incl. the per-layer weight scaling 2^-e
    // head phase on the 16-bit matrix pipe (split_common.h run_heads_mfma): the three 1x1-convolution channels
    // (policy 0, policy 1, value; batch norm folded) as A fragments [k-chunk 2][piece 2][lane 64][8 x f16], their
    // accumulator start values [16] + 2^-e at [16]; the policy FC as A fragments
    // [column tile 6][k-step 6][piece 2][lane 64][8 x f16] (K = 2P padded to 192), 2^-e at pfc_tab[0]
    const unsigned char *hd1_img;
    const float *hd1_tab;
    const unsigned char *pfc_img;
    const float *pfc_tab;
    // Winograd F(2,3) along x on split operands (net_forward_w1d.hip, net_forward_w1dband.hip): weight image
    // [layer 12][point 4][tap ky 3][kc 2][piece 2][ct 4][lane 64][16 B] (batch-norm scale folded in, x 2^e, low pieces
    // unscaled), folded shift [12][64], 2^-e [12] - independent of the board size
    const unsigned char *w1_w;
    const float *w1_shift;
    const float *w1_down;
    unsigned long long *fallbacks;    // [0] launches (partly) redone by the exact-fp32 kernel behind a raised range flag (tg_net_range_fallbacks), [1] positions redone (tg_net_range_fallback_positions)
    unsigned int *band_timeouts;      // banded 19x19 kernels: bounded waits that gave up (host-mapped: band_count reads it without a sync)
    int *overflow;                    // f16 range guard: set when a layer output leaves the f16 range
    float *scratch;       // 19x19 Winograd: per workgroup two [P][64] activation images (L2-resident)
    long long *timeline;  // optional [128] s_memtime stamps of workgroup 0 (tg_net_profile_phases)
};

template <int S, int G>
struct FwdCfg {
    static constexpr int P = S * S;
    static constexpr int A = P + 1;
    static constexpr int M = G * P;
    static constexpr int MT = (M + 15) / 16;
    static constexpr int ROW_BYTES = kRowBytes;
    static constexpr int ACT_BYTES = M * kRowBytes;
    static constexpr int ZROW = ACT_BYTES;               // 288 B zero row
    static constexpr int AUX = ACT_BYTES + kRowBytes;    // in8 [M][8] + zero8, later head scratch
    static constexpr int ZERO8 = AUX + M * 32;
    static constexpr int LDS_BYTES = ZERO8 + 32;
    static constexpr int WAVES_PER_SIMD = (2 * LDS_BYTES <= 160 * 1024) ? 2 : 1;
};

__device__ __forceinline__ float lds_f32(const unsigned char *smem, int byte_off) {
    return *reinterpret_cast<const float *>(smem + byte_off);
}
__device__ __forceinline__ f32x4 lds_f32x4(const unsigned char *smem, int byte_off) {
    return *reinterpret_cast<const f32x4 *>(smem + byte_off);
}

// a - b as two v_pk_add_f32 with negated second operand (hipcc emits four scalar v_sub_f32 for a
// float4 subtraction; a + (-b) is the same IEEE operation)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    const f32x2 alo = __builtin_shufflevector(a, a, 0, 1), ahi = __builtin_shufflevector(a, a, 2, 3);
    const f32x2 blo = __builtin_shufflevector(b, b, 0, 1), bhi = __builtin_shufflevector(b, b, 2, 3);
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(alo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(ahi), "v"(bhi));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}


// ---- pieces shared by the direct and the Winograd kernels -------------------------------------
template <int S, int G, typename C = FwdCfg<S, G>, int NTHR = 256>
__device__ __forceinline__ void stage_planes(unsigned char *smem, const float *__restrict__ planes, int b0,
                                             int batch, int tid) {
    constexpr int P = C::P, M = C::M;
    // global [b][6][P] -> LDS in8 [row][8]
        {
            float *in8 = reinterpret_cast<float *>(smem + C::AUX);
            for (int e = tid; e < G * 6 * P; e += NTHR) {
                const int bl = e / (6 * P);
                const int rem = e - bl * 6 * P;
                const int c = rem / P;
                const int p = rem - c * P;
                const int b = b0 + bl;
                // streamed once: non-temporal, so the planes do not evict the L2-resident weights
                const float v = (b < batch) ? __builtin_nontemporal_load(&planes[(size_t)b * 6 * P + rem]) : 0.f;
                in8[(bl * P + p) * 8 + c] = v;
            }
            for (int e = tid; e < M * 2; e += NTHR) in8[(e >> 1) * 8 + 6 + (e & 1)] = 0.f;
        }
}

template <int S, int G, typename C = FwdCfg<S, G>, int NTHR = 256>
__device__ __forceinline__ void run_heads(unsigned char *smem, const NetDev &net, int b0, int batch,
                                          int want_logits, float *__restrict__ policy,
                                          float *__restrict__ value, int tid) {
    constexpr int P = C::P, A = C::A, M = C::M;
    const int wave = tid >> 6, lane = tid & 63;
        float *hpol = reinterpret_cast<float *>(smem + C::AUX);   // [G][2P]
        float *hval = hpol + G * 2 * P;                           // [G][P]
        float *plog = hval + G * P;                               // [G][A]
        float *vlog = plog + G * A;                               // [G][4]
        {
            const float ps0 = net.head_ss[0], pt0 = net.head_ss[1];
            const float ps1 = net.head_ss[2], pt1 = net.head_ss[3];
            const float vs = net.head_ss[4], vt = net.head_ss[5];
            for (int r = tid; r < M; r += NTHR) {
                float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
                for (int k4 = 0; k4 < 16; ++k4) {
                    const f32x4 xv = lds_f32x4(smem, r * C::ROW_BYTES + k4 * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = k4 * 4 + j;
                        d0 = fmaf(xv[j], net.hp_w[k], d0);
                        d1 = fmaf(xv[j], net.hp_w[64 + k], d1);
                        d2 = fmaf(xv[j], net.hv_w[k], d2);
                    }
                }
                const int bl = r / P, p = r - bl * P;
                hpol[bl * 2 * P + p] = fmaxf(fmaf(d0, ps0, pt0), 0.f);
                hpol[bl * 2 * P + P + p] = fmaxf(fmaf(d1, ps1, pt1), 0.f);
                hval[bl * P + p] = fmaxf(fmaf(d2, vs, vt), 0.f);
            }
        }
        __syncthreads();
        for (int e = tid; e < G * A + G * 3; e += NTHR) {
            if (e < G * A) {
                const int bl = e / A, a = e - bl * A;
                const float *h = hpol + bl * 2 * P;
                const float *wT = net.pfc_wT + a;
                float s0 = net.pfc_b[a], s1 = 0.f, s2 = 0.f, s3 = 0.f;
                int j = 0;
                for (; j + 4 <= 2 * P; j += 4) {
                    s0 = fmaf(h[j], wT[(size_t)j * A], s0);
                    s1 = fmaf(h[j + 1], wT[(size_t)(j + 1) * A], s1);
                    s2 = fmaf(h[j + 2], wT[(size_t)(j + 2) * A], s2);
                    s3 = fmaf(h[j + 3], wT[(size_t)(j + 3) * A], s3);
                }
                for (; j < 2 * P; ++j) s0 = fmaf(h[j], wT[(size_t)j * A], s0);
                plog[e] = (s0 + s1) + (s2 + s3);
            } else {
                const int q = e - G * A;
                const int bl = q / 3, c = q - bl * 3;
                const float *h = hval + bl * P;
                const float *wv = net.vfc_w + c * P;
                float s0 = net.vfc_b[c];
                for (int j = 0; j < P; ++j) s0 = fmaf(h[j], wv[j], s0);
                vlog[bl * 4 + c] = s0;
            }
        }
        __syncthreads();
        for (int bl = wave; bl < G; bl += NTHR / 64) {
            const int b = b0 + bl;
            if (b >= batch) continue;
            float m = -INFINITY;
            for (int a = lane; a < A; a += 64) m = fmaxf(m, plog[bl * A + a]);
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

}  // namespace

// ======================================================================================
// the network handle
// ======================================================================================
struct tg_net {
    int board_size = 0;
    int device = 0;
    int num_cus = 256;
    NetDev dev{};
    std::vector<void *> allocs;
    // staging buffers for the host-pointer entry point (guarded by host_mu: the host API of one
    // handle may be called from several threads, e.g. self-play group threads sharing a network)
    float *st_planes = nullptr, *st_policy = nullptr, *st_value = nullptr;
    int st_cap = 0;
    std::mutex host_mu;
    // 19x19 Winograd kernel: one scratch image set PER STREAM.  Launches on one stream run in order,
    // launches on different streams may overlap on the device and must not share activation images.
    // > 0: the exact-fp32 kernel queued behind a split launch as its range guard takes at most this many workgroups.  Each
    // needs a CU to itself even to read a clear flag: with several streams sharing the device (self-play sub-groups) a
    // full-size guard launch waits for the other streams' forward passes to drain.  The rare real fallback is slower.
    // (Round 6: these caps are a property of the LAUNCH, not of the network - tg::launch_caps() below, set by the launching
    // thread around a self-play move's sub-group launches; a handle shared by group threads carries no such state any more.)
    std::mutex scratch_mu;
    std::map<hipStream_t, float *> scratch_by_stream;
    std::map<hipStream_t, int *> flag_by_stream;      // f16 split kernel: range flag per launch stream
    std::map<hipStream_t, unsigned> flag_seq_by_stream;   // 9x9: launches so far on the stream (which of its two flag sets is next)
    // ... and which GROUPS (workgroup passes: 3 / 1 boards at 9x9, a board at 19x19) raised it: one bit per group, all zero between
    // launches (the exact kernel clears the bits it consumes), so that it redoes those groups only
    struct GroupBits { int *mem = nullptr; int words = 0; };
    std::map<hipStream_t, GroupBits> bits_by_stream;
    // 19x19 one-axis Winograd kernel (net_forward_w1dband.hip): per stream the pairs' exchange rows + the feature image
    struct WbScratch { float *mem = nullptr; int cap = 0; };
    std::map<hipStream_t, WbScratch> wb_by_stream;
    // banded 19x19 kernel: its launches follow each other even across streams (two of them half-resident on the device would
    // hold each other's missing bands off the CUs until the bounded waits give up) - the last launch's completion event
    hipEvent_t band_done = nullptr;
    hipStream_t band_stream = nullptr;
    bool band_recorded = false;
    // the host's view of dev.band_timeouts (pinned, mapped): once a banded launch has run into its bounded wait - the device is
    // shared with somebody whose kernels keep bands off the CUs - this network stays on the one-workgroup kernels
    volatile unsigned int *band_timeouts_host = nullptr;
    bool shared_device = false;             // tg_net_set_shared_device: several processes drive this GPU
    size_t scratch_floats = 0;
};

namespace tg {
// Per-launch grid caps of the forward family, owned by the launching THREAD (search.hip's play_move_chain scopes them around the
// sub-group launches of one self-play move):
//   guard   > 0: the exact-fp32 kernel queued behind a split launch as its range guard takes at most this many workgroups
//   forward > 0: workgroups a forward launch may take (CUs left to other streams' tree kernels)
struct LaunchCaps { int guard = 0, forward = 0; };
LaunchCaps &launch_caps();
struct LaunchCapsScope {
    LaunchCaps saved;
    LaunchCapsScope(int guard, int forward) : saved(launch_caps()) { launch_caps() = LaunchCaps{guard, forward}; }
    ~LaunchCapsScope() { launch_caps() = saved; }
    LaunchCapsScope(const LaunchCapsScope &) = delete;
    LaunchCapsScope &operator=(const LaunchCapsScope &) = delete;
};
}  // namespace tg
