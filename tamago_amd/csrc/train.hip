// Training step of the DualNet on gfx950 (MI355X): forward with batch statistics, backward, SGD-Nesterov -
// hand-written HIP kernels, no library call.  Replaces one mini-batch of nn/learn.py:360-376 (RL: KLD policy
// loss + value cross entropy, nn/loss.py:33-55) and of the supervised trainer (learn.py:150-180) as executed
// by torch autograd over the modules of nn/network/dual_net.py:16-52, res_block.py:8-40, head/*.py, with
// torch.optim.SGD(momentum 0.9, weight decay 1e-4, nesterov) - in fp32 (the reference runs it under fp16
// autocast; fp32 is what its CPU trainer does and the stricter of the two).
//
// Board sizes: 9x9 (the size the reference trains at; every tuning decision below was made for it) and 19x19 (the same kernels
// instantiated for S = 19: a board in four staging passes and four passes of the MFMA loop, one LDS buffer in the weight
// gradient, FC layers by groups of outputs).
// Data layout (HBM): activations are NHWC fp32 [B][P][64] (P = S * S positions).  Kept per step: Z_l = convolution output before
// its batch norm (13 layers), Y_b = block outputs after ReLU (stem + 6 blocks), D_l = dL/d(batch-norm output
// of layer l, ReLU mask applied).  Nothing else is materialised: every kernel rebuilds the tensor it needs
// while staging a board into LDS (batch norm + ReLU of the producer's Z, batch-norm backward of D), so a
// step is ~45 launches instead of the ~230 of the operator-by-operator library path.
//
// Kernels (one workgroup = one board at a time, 4 waves; wave w owns output channels [16w, 16w+16)):
//   conv_kernel<FWD>   implicit GEMM 81(96) x 64 x 576 on v_mfma_f32_16x16x4_f32, epilogue: Z + per-channel
//                      sum / sum of squares (the batch statistics) by atomics
//   conv_kernel<DGRAD> the same loop on rotated / transposed weights over dZ; epilogue: + skip gradient,
//                      ReLU mask, D of the layer below + its two batch-norm-backward sums
//   wgrad_kernel       dW[tap][co][ci] = sum_rows dZ[row][co] * A[row + tap][ci]; a workgroup owns 16 output channels for a
//                      chunk of boards (accumulated in registers, staged through two LDS buffers), one partial image per chunk
//   head_* kernels     1x1 convolutions, their batch norms, both fully connected layers, the losses and all
//                      of their gradients
//   sgd_kernel         reduces the partial images, adds weight decay, momentum / Nesterov, updates; batch-norm
//                      running statistics in the same launch
#include "common.h"

#include <cmath>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 64;
constexpr int kLayers = 13;                       // conv layers: stem + 12
constexpr int kRow = 72;                          // LDS row stride in floats (conflict-free ds_read_b128)
// Board geometry of a kernel instantiated for board size S (9: the size the reference trains at and every tuning decision
// here was made for; 19: the same kernels walking the board in more passes): positions, actions, padded width / cells, 16-row tiles
#define TG_GEO(S) constexpr int P = (S) * (S), A = P + 1, W = (S) + 2, kCells = W * W, kMT = (P + 15) / 16; \
    (void)A; (void)W; (void)kCells; (void)kMT
constexpr int kConvW = 64 * 64 * 9;
constexpr int kRep = 16;                          // replicas of every atomically accumulated statistic

// ---- parameter blob (tg_net_param_count order, include/tamago_hip.h) -----------------------------------
struct Layout {
    size_t conv[kLayers], bn_w[kLayers], bn_b[kLayers], bn_m[kLayers], bn_v[kLayers];
    size_t p_conv, p_bn_w, p_bn_b, p_bn_m, p_bn_v, p_fc_w, p_fc_b;
    size_t v_conv, v_bn_w, v_bn_b, v_bn_m, v_bn_v, v_fc_w, v_fc_b;
    size_t total;
};
Layout make_layout(int S) {
    const int P = S * S, A = P + 1;
    Layout L{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += n; return r; };
    L.conv[0] = take(64 * 6 * 9);
    L.bn_w[0] = take(64); L.bn_b[0] = take(64); L.bn_m[0] = take(64); L.bn_v[0] = take(64);
    for (int b = 0; b < 6; ++b) {
        L.conv[1 + 2 * b] = take(kConvW);
        L.conv[2 + 2 * b] = take(kConvW);
        for (int k = 0; k < 2; ++k) {
            const int l = 1 + 2 * b + k;
            L.bn_w[l] = take(64); L.bn_b[l] = take(64); L.bn_m[l] = take(64); L.bn_v[l] = take(64);
        }
    }
    L.p_conv = take(2 * 64); L.p_bn_w = take(2); L.p_bn_b = take(2); L.p_bn_m = take(2); L.p_bn_v = take(2);
    L.p_fc_w = take((size_t)A * 2 * P); L.p_fc_b = take(A);
    L.v_conv = take(64); L.v_bn_w = take(1); L.v_bn_b = take(1); L.v_bn_m = take(1); L.v_bn_v = take(1);
    L.v_fc_w = take(3 * P); L.v_fc_b = take(3);
    L.total = o;
    return L;
}

struct TrainDev {
    float *param, *grad, *mom;      // flat blobs (grad / mom: running statistics slots unused)
    float *wf, *wb;                 // [13][4 wave][9 tap][4 s][64 lane][4]: forward / transposed-rotated fragments
    float *Z, *Y, *D;               // [13][B][81][64], [7][B][81][64], [13][B][81][64]
    double *stat;                   // per layer [kRep][4][64]: sum z, sum z^2 (forward), S1 = sum D, S2 = sum D*xhat (backward);
                                    // double: var = E[z^2] - mean^2 cancels, and the stem's gradient sees 13 layers of it.
                                    // kRep replicas (workgroup w adds to replica w % kRep): 256 same-address atomics
                                    // per channel serialise in L2 and cost more than the convolution itself
    float *partial;                 // [13][WCH][9][64][64] weight-gradient partial images, one per chunk of boards
    float *hz, *hD;                 // head conv outputs / their D: [B][81][4] (2 policy, 1 value, 1 pad)
    double *hstat;                  // [kRep][4][4]: sum, sumsq, S1, S2 for the 3 head channels
    float *hact;                    // [B][3*81] ReLU(BN(hz)) flattened: policy c*81+p (162), value (81)
    float *dlog;                    // [B][A + 3] dL/dlogits
    double *loss;                   // [kRep][4] accumulated: total, policy, value
    Layout L;
    int B, NWG, WCH, P;             // batch, workgroups of the per-board kernels, board chunks of wgrad_kernel, positions per board
};

struct TrainDev;
__device__ __forceinline__ double stat_sum(const double *base, int stride) {   // sum over the replicas
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < kRep; ++r) v += base[(size_t)r * stride];
    return v;
}
__device__ __forceinline__ f32x4 lds4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// compile-time loop: fn(std::integral_constant<int, 0>{}), ..., fn(std::integral_constant<int, N - 1>{})
template <typename Fn, int... Is>
__device__ __forceinline__ void static_for_impl(Fn &&fn, std::integer_sequence<int, Is...>) {
    (fn(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename Fn>
__device__ __forceinline__ void static_for(Fn &&fn) {
    static_for_impl(fn, std::make_integer_sequence<int, N>{});
}

// batch-norm constants of layer `l` from its statistics: scale = gamma * rstd, shift = beta - mean * scale
__device__ __forceinline__ void bn_consts(const TrainDev &T, int l, int c, float eps, float &mean, float &rstd) {
    const double n = (double)(T.B * T.P);
    const double m = stat_sum(T.stat + (size_t)l * kRep * 256 + c, 256) / n;
    const double var = fmax(stat_sum(T.stat + (size_t)l * kRep * 256 + 64 + c, 256) / n - m * m, 0.0);
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

enum ConvMode { FWD = 0, DGRAD = 1 };

// -DTG_TRAIN_PROF (tools/experiments/train_prof.sh): workgroup 0 stamps the 100 MHz clock at the phase boundaries of four launches
// of a step (slot 0: conv FWD of layer 5, 1: conv DGRAD of layer 5, 2: wgrad of layer 5, 3: head_loss); tg_trainer_debug_read(3)
#ifdef TG_TRAIN_PROF
__device__ unsigned long long g_prof[4 * 16];
#define TG_PROF(slot, k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_prof[(slot) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define TG_PROF(slot, k) do { } while (0)
#endif

// ---- 3x3 convolution, forward and data gradient ----------------------------------------------------------
// layer l (0 = stem).  FWD: input = activation feeding conv l, output Z_l + statistics.
// DGRAD (l >= 1): input dZ_l (batch-norm backward of D_l, rebuilt on load), output D_{l-1} + its sums.
// 4 waves: wave w owns output channels [16w, 16w+16) of all six row tiles.  A 256-position batch is one board per CU
// = one wave per SIMD, so the loop hides its own latencies: weight fragments (L2) are requested two (tap, channel
// group) steps ahead, activation fragments (LDS) one step ahead, positions pinned with sched_barrier.
// `per_board` workgroups share a board (1, or - 19x19 batches below the CU count - one per pass of the MFMA loop: each stages the
// board and computes six of its row tiles; a 64-position 19x19 batch on 64 CUs was slower than the library).
template <int MODE, int S>
__global__ __launch_bounds__(256) void conv_kernel(TrainDev T, const float *__restrict__ planes, int l, int per_board) {
    TG_GEO(S);
    const int pb = kMT > 6 ? per_board : 1;              // (9x9: one pass, a constant)
    const int wg = blockIdx.x / pb, part = blockIdx.x - wg * pb, nwg = gridDim.x / pb;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *act = smem;                       // [cells of the (S + 2)^2 padded board][72]: the border stays zero, a tap is a constant offset
    float *tab = smem + kCells * kRow;       // per-channel constants [9][64]
    constexpr int NT = 256, HMT = 6;                       // row tiles per pass of the MFMA loop (9x9: all six of the board)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const float eps_l = 2e-5f;
    for (int e = tid; e < kCells * kRow / 4; e += NT) reinterpret_cast<f32x4 *>(act)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- the board's global loads are requested before the per-channel tables are made: the tables cost a round trip to the
    //      statistics + fp64 division and square root, the board a round trip of its own (1.5 us per launch when one followed
    //      the other; a 256-position batch is ONE board per workgroup).  A thread stages six float4 per pass: the whole 9x9
    //      board in one pass, a 19x19 board in four ----
    constexpr int NV = (P * C / 4 + NT - 1) / NT, NVP = 6;  // float4 per thread: of the board, of a pass
    const size_t bstride = (size_t)P * C;
    const bool fwd = MODE == FWD;
    const bool conv1 = (l & 1) != 0;
    const int yb = (l - 1) / 2;
    const bool staged = !(MODE == FWD && l == 0);
    const bool has_second = fwd ? (conv1 && yb >= 1) : true;
    f32x4 zv[NVP], sv[NVP];
    auto request = [&](int b, int i0) {
        // FWD: z = Z_{l-1}; conv1 also adds the previous block output and materialises Y.  DGRAD: d = D_l, z = Z_l.
        const float *zsrc = T.Z + ((size_t)(fwd ? l - 1 : l) * T.B + b) * bstride;
        const float *second = !has_second ? zsrc : fwd ? T.Y + ((size_t)(yb - 1) * T.B + b) * bstride : T.D + ((size_t)l * T.B + b) * bstride;
#pragma unroll
        for (int i = 0; i < NVP; ++i) {
            const int e = tid + (i0 + i) * NT;
            if (e < P * C / 4) {
                zv[i] = *reinterpret_cast<const f32x4 *>(zsrc + e * 4);
                if (has_second) sv[i] = *reinterpret_cast<const f32x4 *>(second + e * 4);
            }
        }
    };
    const int pslot = l == 5 ? MODE : -1;
    if (pslot >= 0) TG_PROF(pslot, 0);
    if (staged && wg < T.B) request(wg, 0);
    // ---- per-channel tables -------------------------------------------------------------------------
    // FWD: tab[0] = scale, tab[1] = shift of the PRODUCER's batch norm (layer l - 1)
    // DGRAD: tab[0] = gamma*rstd of layer l, tab[1] = mean, tab[2] = rstd, tab[3] unused; m1/m2 in tab[4..5]
    if (MODE == FWD) {
        if (tid < 64 && l >= 1) {
            float mean, rstd;
            bn_consts(T, l - 1, tid, l - 1 == 0 ? 1e-5f : eps_l, mean, rstd);
            const float sc = T.param[T.L.bn_w[l - 1] + tid] * rstd;
            tab[tid] = sc;
            tab[64 + tid] = T.param[T.L.bn_b[l - 1] + tid] - mean * sc;
        }
    } else {
        // three waves, a third of the constants each (one wave making all of them: 3.1 us of statistics round trip + fp64 chain)
        const int c = tid & 63, part = tid >> 6;
        if (part == 0) {
            float mean, rstd;
            bn_consts(T, l, c, eps_l, mean, rstd);
            tab[c] = T.param[T.L.bn_w[l] + c] * rstd;
            tab[64 + c] = mean;
            tab[128 + c] = rstd;
        } else if (part == 1) {
            const double n = (double)(T.B * T.P);
            tab[192 + c] = (float)(stat_sum(T.stat + (size_t)l * kRep * 256 + 128 + c, 256) / n);      // m1 = mean(D)
            tab[256 + c] = (float)(stat_sum(T.stat + (size_t)l * kRep * 256 + 192 + c, 256) / n);      // m2 = mean(D * xhat)
        } else if (part == 2) {
            // constants of the layer BELOW (mask / xhat of D_{l-1})
            float mean2, rstd2;
            bn_consts(T, l - 1, c, l - 1 == 0 ? 1e-5f : eps_l, mean2, rstd2);
            const float sc2 = T.param[T.L.bn_w[l - 1] + c] * rstd2;
            tab[320 + c] = sc2;
            tab[384 + c] = T.param[T.L.bn_b[l - 1] + c] - mean2 * sc2;
            tab[448 + c] = mean2;
            tab[512 + c] = rstd2;
        }
    }
    __syncthreads();
    if (pslot >= 0) TG_PROF(pslot, 1);
    const float *wfrag = (MODE == FWD ? T.wf : T.wb) + (size_t)l * 4 * 9 * 4 * 64 * 4;
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(wfrag) + (size_t)wave * 9 * 4 * 64 + lane;
    float s_sum[4] = {0.f, 0.f, 0.f, 0.f}, s_sq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = wg; b < T.B; b += nwg) {
        // ---- stage the board: LDS act[cell][c] from the registers requested above / at the end of the board before ----
        if (MODE == FWD && l == 0) {
            for (int e = tid; e < P * C; e += NT) {
                const int row = e >> 6, c = e & 63, y = row / S;
                act[((y + 1) * W + row - y * S + 1) * kRow + c] = c < 6 ? planes[((size_t)b * 6 + c) * P + row] : 0.f;
            }
        } else {
            float *yout = fwd && conv1 && part == 0 ? T.Y + ((size_t)yb * T.B + b) * bstride : nullptr;
            for (int i0 = 0; i0 < NV; i0 += NVP) {
                if (i0 > 0) request(b, i0);
#pragma unroll
                for (int i = 0; i < NVP; ++i) {
                    const int e = tid + (i0 + i) * NT;
                    if (e < P * C / 4) {
                        const int row = e >> 4, c = (e & 15) * 4, y = row / S, cell = (y + 1) * W + row - y * S + 1;
                        f32x4 v;
                        if (fwd) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaf(zv[i][j], tab[c + j], tab[64 + c + j]);
                            if (has_second) v += sv[i];
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                            if (yout) *reinterpret_cast<f32x4 *>(yout + e * 4) = v;
                        } else {
                            // dZ_l = gamma*rstd * (D - m1 - xhat * m2), xhat = (Z - mean) * rstd
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float xh = (zv[i][j] - tab[64 + c + j]) * tab[128 + c + j];
                                v[j] = tab[c + j] * (sv[i][j] - tab[192 + c + j] - xh * tab[256 + c + j]);
                            }
                        }
                        *reinterpret_cast<f32x4 *>(act + cell * kRow + c) = v;
                    }
                }
            }
        }
        __syncthreads();
        if (pslot >= 0) TG_PROF(pslot, 2);
        const bool ep_conv1 = MODE == DGRAD && (l & 1) != 0;   // input of conv1 = block output Y_{(l-1)/2}; the skip carries D_{l+1}
        const int c0 = wave * 16 + lg * 4;
        // ---- implicit GEMM: acc[mt] (16 couts x 16 rows) over 9 taps x 16 k-groups of 4 channels, HMT row tiles per pass ----
        for (int mt0 = pb > 1 ? part * HMT : 0; mt0 < (pb > 1 ? (part + 1) * HMT : kMT); mt0 += HMT) {
            f32x4 acc[HMT];
#pragma unroll
            for (int mt = 0; mt < HMT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // DGRAD: the epilogue's operands (Z, Y of the layer below, the skip gradient) are requested now, behind the loop
            f32x4 ep_z[HMT], ep_y[HMT], ep_s[HMT];
            if (MODE == DGRAD) {
                const float *zb = T.Z + ((size_t)(l - 1) * T.B + b) * bstride;
                const float *yb2 = T.Y + ((size_t)((l - 1) / 2) * T.B + b) * bstride;
                const float *dskip = T.D + ((size_t)(l + 1 < kLayers ? l + 1 : l) * T.B + b) * bstride;
#pragma unroll
                for (int mt = 0; mt < HMT; ++mt) {
                    const int row = (mt0 + mt) * 16 + li, r2 = row < P ? row : 0;
                    ep_z[mt] = *reinterpret_cast<const f32x4 *>(zb + r2 * C + c0);
                    if (ep_conv1) {
                        ep_y[mt] = *reinterpret_cast<const f32x4 *>(yb2 + r2 * C + c0);
                        ep_s[mt] = *reinterpret_cast<const f32x4 *>(dskip + r2 * C + c0);
                    }
                }
            }
            // per row tile: the lane's row on the padded board, one cell up-left (tap (0, 0)); rows beyond the board (the last
            // tile) read row 0's cells - their products are never stored
            const float *rowp[HMT];
#pragma unroll
            for (int mt = 0; mt < HMT; ++mt) {
                const int r = (mt0 + mt) * 16 + li, r2 = r < P ? r : 0, y = r2 / S;
                rowp[mt] = act + (y * W + r2 - y * S) * kRow + lg * 4;
            }
            f32x4 wq[3], av[2][HMT];
            wq[0] = wl[0];
            wq[1] = wl[64];
#pragma unroll
            for (int mt = 0; mt < HMT; ++mt) av[0][mt] = lds4(rowp[mt]);
            static_for<36>([&](auto ST_) {
                constexpr int st = decltype(ST_)::value, nx = st + 1, tapn = nx / 4, sn = nx % 4;
                constexpr int toffn = ((tapn / 3) * W + tapn % 3) * kRow;
                wq[(st + 2) % 3] = wl[(st + 2 < 36 ? st + 2 : 35) * 64];
                if constexpr (nx < 36) {
#pragma unroll
                    for (int mt = 0; mt < HMT; ++mt) av[nx & 1][mt] = lds4(rowp[mt] + toffn + sn * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < HMT; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[st % 3][j], av[st & 1][mt][j], acc[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            // ---- epilogue ----
            if (pslot >= 0) TG_PROF(pslot, 3);
            if (MODE == FWD) {
                float *z = T.Z + ((size_t)l * T.B + b) * bstride;
#pragma unroll
                for (int mt = 0; mt < HMT; ++mt) {
                    const int row = (mt0 + mt) * 16 + li;
                    if (row < P) {
                        *reinterpret_cast<f32x4 *>(z + row * C + c0) = acc[mt];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { s_sum[j] += acc[mt][j]; s_sq[j] = fmaf(acc[mt][j], acc[mt][j], s_sq[j]); }
                    }
                }
            } else {
                // D_{l-1} = (dA + skip) * [A_{l-1} > 0]; sums S1 = sum D, S2 = sum D * xhat_{l-1}
                float *dout = T.D + ((size_t)(l - 1) * T.B + b) * bstride;
#pragma unroll
                for (int mt = 0; mt < HMT; ++mt) {
                    const int row = (mt0 + mt) * 16 + li;
                    if (row < P) {
                        f32x4 g = acc[mt];
                        if (ep_conv1) g += ep_s[mt];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float pre = ep_conv1 ? ep_y[mt][j] : fmaf(ep_z[mt][j], tab[320 + c0 + j], tab[384 + c0 + j]);
                            g[j] = pre > 0.f ? g[j] : 0.f;
                            const float xh = (ep_z[mt][j] - tab[448 + c0 + j]) * tab[512 + c0 + j];
                            s_sum[j] += g[j];
                            s_sq[j] = fmaf(g[j], xh, s_sq[j]);
                        }
                        *reinterpret_cast<f32x4 *>(dout + row * C + c0) = g;
                    }
                }
            }
        }
        if (staged && b + nwg < T.B) request(b + nwg, 0);
        __syncthreads();
        if (pslot >= 0) TG_PROF(pslot, 4);
    }
    // per-channel sums: reduce over the 16 lanes that share lg, one atomic per channel and workgroup-wave
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s_sum[j] += __shfl_xor(s_sum[j], o); s_sq[j] += __shfl_xor(s_sq[j], o); }
    }
    if (li == 0) {
        const int sl = MODE == FWD ? l : l - 1;
        const int base = (sl * kRep + (int)(blockIdx.x % kRep)) * 256 + (MODE == FWD ? 0 : 128) + wave * 16 + lg * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&T.stat[base + j], (double)s_sum[j]);
            atomicAdd(&T.stat[base + 64 + j], (double)s_sq[j]);
        }
    }
    if (pslot >= 0) TG_PROF(pslot, 5);
}

// ---- weight gradient ---------------------------------------------------------------------------------------
// dW[tap][co][ci] = sum over boards and rows of dZ[row][co] * A[row + tap][ci].  Workgroup (q, chunk) owns the output channels
// [16q, 16q+16) of every tap and input channel for the boards chunk, chunk + chunks, ...: 36 MFMA tiles (9 taps x 4 ci tiles)
// shared out over 8 waves (wave w: ci tile w & 3, taps (w >> 2) + 2i - five on waves 0-3, four on waves 4-7, nine per SIMD).
// One partial image per CHUNK (64 of them at batch >= 64: 9.4 MB per layer; one image per workgroup and board was 37.7 MB
// written and read again, 491 MB per step).  A board's dZ slice (16 channels) and activations (all 64) are staged into one of
// two LDS buffers while the MFMAs of the board before run on the other: activations on an 11 x 11 padded board whose border
// stays zero, so that a tap is a constant address offset (no validity test per MFMA), cells 80 floats apart (the four rows x 16
// channels of a ds_read_b32 hit 32 distinct banks twice).
// 19x19: the padded board alone is 127 KB at 72 floats per cell - one buffer (the staging of the next board's registers still
// runs under the MFMAs, its deposit does not), lane addresses computed per k-step instead of kept in 91 registers.
template <int S>
struct WGeo {
    static constexpr int P = S * S, W = S + 2, KS = (P + 3) / 4;                 // k-steps of four rows
    static constexpr int kWRow = S == 9 ? 80 : 72, NBUF = S == 9 ? 2 : 1;
    static constexpr int kWAct = W * W * kWRow, kWBuf = kWAct + KS * 4 * 16;
    static constexpr int kLdsFloats = NBUF * kWBuf + 5 * 16 + 2 * 64;
};

template <int S>
__global__ __launch_bounds__(512) void wgrad_kernel(TrainDev T, const float *__restrict__ planes, int l) {
    using G = WGeo<S>;
    constexpr int P = G::P, W = G::W, KS = G::KS, kWRow = G::kWRow, NBUF = G::NBUF, kWAct = G::kWAct, kWBuf = G::kWBuf;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tab = smem + NBUF * kWBuf;          // [5][16]: gamma*rstd, mean, rstd, m1, m2 of the slice; [2][64]: scale / shift of layer l - 1
    constexpr int NT = 512;
    const int tid = threadIdx.x, wave = tid >> 6, ct = wave & 3, tpar = wave >> 2, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int q = blockIdx.x & 3, chunk = blockIdx.x >> 2, chunks = gridDim.x >> 2;
    const float eps_l = 2e-5f;
    const bool rebuild = l >= 2 && (l & 1) == 0;          // conv2: its input h = relu(bn(Z_{l-1})) is rebuilt
    for (int e = tid; e < NBUF * kWBuf / 4; e += NT) reinterpret_cast<f32x4 *>(smem)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 16) {
        const int c = q * 16 + tid;
        float mean, rstd;
        bn_consts(T, l, c, l == 0 ? 1e-5f : eps_l, mean, rstd);
        const double n = (double)(T.B * T.P);
        tab[tid] = T.param[T.L.bn_w[l] + c] * rstd;
        tab[16 + tid] = mean;
        tab[32 + tid] = rstd;
        tab[48 + tid] = (float)(stat_sum(T.stat + (size_t)l * kRep * 256 + 128 + c, 256) / n);
        tab[64 + tid] = (float)(stat_sum(T.stat + (size_t)l * kRep * 256 + 192 + c, 256) / n);
    } else if (rebuild && tid >= 64 && tid < 128) {
        const int c = tid - 64;
        float mean2, rstd2;
        bn_consts(T, l - 1, c, eps_l, mean2, rstd2);
        const float sc2 = T.param[T.L.bn_w[l - 1] + c] * rstd2;
        tab[80 + c] = sc2;
        tab[144 + c] = T.param[T.L.bn_b[l - 1] + c] - mean2 * sc2;
    }
    const size_t bstride = (size_t)P * C;
    const float *asrc = l == 0 ? nullptr : (l & 1) ? T.Y + (size_t)((l - 1) / 2) * T.B * bstride : T.Z + (size_t)(l - 1) * T.B * bstride;
    constexpr int NV = (P * C / 4 + NT - 1) / NT;          // float4 of activations per thread
    constexpr int ND = (P * 4 + NT - 1) / NT;              // float4 of the dZ slice per thread
    f32x4 ra[NV], rd[ND], rz[ND];
    auto request = [&](int b) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + i * NT;
            if (e < P * C / 4) {
                if (l == 0) {
                    const int row = e >> 4, c = (e & 15) * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ra[i][j] = c + j < 6 ? planes[((size_t)b * 6 + c + j) * P + row] : 0.f;
                } else {
                    ra[i] = *reinterpret_cast<const f32x4 *>(asrc + (size_t)b * bstride + e * 4);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int e = tid + i * NT;
            if (e < P * 4) {
                const size_t off = ((size_t)l * T.B + b) * bstride + (e >> 2) * C + q * 16 + (e & 3) * 4;
                rd[i] = *reinterpret_cast<const f32x4 *>(T.D + off);
                rz[i] = *reinterpret_cast<const f32x4 *>(T.Z + off);
            }
        }
    };
    auto deposit = [&](float *buf) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + i * NT;
            if (e < P * C / 4) {
                const int row = e >> 4, c = (e & 15) * 4, y = row / S, cell = (y + 1) * W + (row - y * S) + 1;
                f32x4 a = ra[i];
                if (rebuild) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = fmaxf(fmaf(a[j], tab[80 + c + j], tab[144 + c + j]), 0.f);
                }
                *reinterpret_cast<f32x4 *>(buf + cell * kWRow + c) = a;
            }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int e = tid + i * NT;
            if (e < P * 4) {
                const int c = (e & 3) * 4;
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (rz[i][j] - tab[16 + c + j]) * tab[32 + c + j];
                    v[j] = tab[c + j] * (rd[i][j] - tab[48 + c + j] - xh * tab[64 + c + j]);
                }
                *reinterpret_cast<f32x4 *>(buf + kWAct + (e >> 2) * 16 + c) = v;
            }
        }
    };
    // lane's activation address of k-step ks (rows ks * 4 + lg), relative to the cell one row and one column up-left of it, so
    // that every tap is a non-negative constant; rows beyond the board (dZ is zero there) read a cell of finite values
    auto aoff_of = [&](int ks) {
        const int row = ks * 4 + lg, r2 = row < P ? row : 0, y = r2 / S;
        return (y * W + (r2 - y * S)) * kWRow + ct * 16 + li;
    };
    int aoff[S == 9 ? KS : 1];
    if constexpr (S == 9) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) aoff[ks] = aoff_of(ks);
    }
    f32x4 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int b = chunk;
    if (l == 5) TG_PROF(2, 0);
    if (b < T.B) request(b);
    __syncthreads();                                       // tables and the zeroed buffers
    if (l == 5) TG_PROF(2, 1);
    for (int it = 0; b < T.B; b += chunks, ++it) {
        float *buf = smem + (NBUF == 2 ? (it & 1) : 0) * kWBuf;
        deposit(buf);
        __syncthreads();
        if (l == 5 && it < 4) TG_PROF(2, 2 + it);
        if (b + chunks < T.B) request(b + chunks);
        const float *dz = buf + kWAct + lg * 16 + li;
        auto run = [&](auto TP_) {
            constexpr int TP = decltype(TP_)::value;
            auto step = [&](int ks, int ao) {
                const float dzv = dz[ks * 64];
                const float *ap = buf + ao;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int tap = TP + 2 * i;
                    if (tap < 9) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, ap[((tap / 3) * W + tap % 3) * kWRow], acc[i], 0, 0, 0);
                }
            };
            if constexpr (S == 9) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) step(ks, aoff[ks]);
            } else {
#pragma unroll 7
                for (int ks = 0; ks < KS; ++ks) step(ks, aoff_of(ks));
            }
        };
        if (tpar == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 1>{});
        if constexpr (NBUF == 1) __syncthreads();          // the one buffer is free for the next board's deposit
    }
    if (l == 5) TG_PROF(2, 6);
    // partial image of the chunk: [chunk][tap][co][ci]; lane holds co = 16 q + 4 lg + j, ci = 16 ct + li
    float *out = T.partial + ((size_t)l * chunks + chunk) * 9 * 64 * 64;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int tap = tpar + 2 * i;
        if (tap < 9)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(tap * 64 + q * 16 + lg * 4 + j) * 64 + ct * 16 + li] = acc[i][j];
    }
    if (l == 5) TG_PROF(2, 7);
}

// ---- weight fragments from the master weights (they change every step) ---------------------------------
__global__ void repack_kernel(TrainDev T) {
    const int l = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;          // fragment element index
    // first launch of a step: the accumulators of the step are cleared here (four memset launches of ~5 us each before)
    {
        const int g = l * (int)(gridDim.x * blockDim.x) + e;
        constexpr int n_stat = kLayers * kRep * 256, n_h = kRep * 16;
        if (g < n_stat) T.stat[g] = 0.0;
        else if (g < n_stat + n_h) T.hstat[g - n_stat] = 0.0;
        else if (g < n_stat + n_h + 128) T.grad[T.L.p_conv + (g - n_stat - n_h)] = 0.f;
        else if (g < n_stat + n_h + 192) T.grad[T.L.v_conv + (g - n_stat - n_h - 128)] = 0.f;
    }
    if (e >= 4 * 9 * 4 * 64 * 4) return;
    const int j = e & 3, lane = (e >> 2) & 63, s = (e >> 8) & 3, tap = (e >> 10) % 9, wv = e / (9 * 1024);
    const int n = lane & 15, g = lane >> 4;
    const int k = 16 * s + 4 * g + j, o = 16 * wv + n;             // k: reduction channel, o: output channel of the pass
    float f, bwd = 0.f;
    if (l == 0) {
        f = k < 6 ? T.param[T.L.conv[0] + (o * 6 + k) * 9 + tap] : 0.f;
    } else {
        const float *w = T.param + T.L.conv[l];
        f = w[(o * 64 + k) * 9 + tap];                             // forward: out = cout, k = cin
        bwd = w[(k * 64 + o) * 9 + (8 - tap)];                     // data gradient: out = cin, k = cout, taps mirrored
    }
    T.wf[(size_t)l * 36864 + e] = f;
    T.wb[(size_t)l * 36864 + e] = bwd;
}

// ---- heads -----------------------------------------------------------------------------------------------
// 1. per board: Y_6 = relu(Y_5 + bn(Z_12)) (materialised), head 1x1 convolutions -> hz[b][p][3], statistics.
//    Thread (row group r = tid / 16, channel quad c = 4 (tid % 16)): a row is one coalesced 256-byte read by 16 lanes, the three
//    dot products are finished by a butterfly over those lanes (a thread per ROW read 64 scattered lines per instruction).
template <int S>
__global__ __launch_bounds__(256) void head_conv_kernel(TrainDev T) {
    TG_GEO(S);
    __shared__ float red[4][6];
    __shared__ float tab[128];
    const int tid = threadIdx.x, c = (tid & 15) * 4, r0 = tid >> 4;
    const size_t bstride = (size_t)P * C;
    constexpr int NR = (P + 15) / 16, NRP = 6;             // rows per thread: of the board, of a pass (9x9: one pass)
    f32x4 zv[NRP], yv[NRP];
    auto request = [&](int b, int k0) {
        const float *z = T.Z + ((size_t)12 * T.B + b) * bstride;
        const float *yp = T.Y + ((size_t)5 * T.B + b) * bstride;
#pragma unroll
        for (int k = 0; k < NRP; ++k) {
            const int row = r0 + 16 * (k0 + k);
            if (row < P) {
                zv[k] = *reinterpret_cast<const f32x4 *>(z + row * C + c);
                yv[k] = *reinterpret_cast<const f32x4 *>(yp + row * C + c);
            }
        }
    };
    if ((int)blockIdx.x < T.B) request(blockIdx.x, 0);    // in flight while the batch-norm constants are made
    if (tid < 64) {
        float mean, rstd;
        bn_consts(T, 12, tid, 2e-5f, mean, rstd);
        const float sc1 = T.param[T.L.bn_w[12] + tid] * rstd;
        tab[tid] = sc1;
        tab[64 + tid] = T.param[T.L.bn_b[12] + tid] - mean * sc1;
    }
    float w0[4], w1[4], w2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w0[j] = T.param[T.L.p_conv + c + j];
        w1[j] = T.param[T.L.p_conv + 64 + c + j];
        w2[j] = T.param[T.L.v_conv + c + j];
    }
    __syncthreads();
    float sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = tab[c + j]; sh[j] = tab[64 + c + j]; }
    float s[3] = {0.f, 0.f, 0.f}, q[3] = {0.f, 0.f, 0.f};
    for (int b = blockIdx.x; b < T.B; b += gridDim.x) {
        float *yo = T.Y + ((size_t)6 * T.B + b) * bstride;
        for (int k0 = 0; k0 < NR; k0 += NRP) {
            if (k0 > 0) request(b, k0);
#pragma unroll
            for (int k = 0; k < NRP; ++k) {
                const int row = r0 + 16 * (k0 + k);       // uniform over the 16 lanes of a row: the butterfly below stays inside them
                if (row < P) {
                    f32x4 v = yv[k];
                    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = fmaxf(v[j] + fmaf(zv[k][j], sc[j], sh[j]), 0.f);
                        d0 = fmaf(v[j], w0[j], d0);
                        d1 = fmaf(v[j], w1[j], d1);
                        d2 = fmaf(v[j], w2[j], d2);
                    }
                    *reinterpret_cast<f32x4 *>(yo + row * C + c) = v;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); }
                    if ((tid & 15) == 0) {
                        *reinterpret_cast<f32x4 *>(T.hz + ((size_t)b * P + row) * 4) = f32x4{d0, d1, d2, 0.f};
                        s[0] += d0; s[1] += d1; s[2] += d2;
                        q[0] = fmaf(d0, d0, q[0]); q[1] = fmaf(d1, d1, q[1]); q[2] = fmaf(d2, d2, q[2]);
                    }
                }
            }
        }
        if (b + (int)gridDim.x < T.B) request(b + gridDim.x, 0);
    }
    // lanes 0, 16, 32, 48 of a wave hold sums: one more butterfly, then wave -> workgroup through LDS, one atomic per statistic
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s[k] += __shfl_xor(s[k], 16); s[k] += __shfl_xor(s[k], 32);
        q[k] += __shfl_xor(q[k], 16); q[k] += __shfl_xor(q[k], 32);
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { red[tid >> 6][k] = s[k]; red[tid >> 6][3 + k] = q[k]; }
    }
    __syncthreads();
    double *hs = T.hstat + (blockIdx.x % kRep) * 16;
    if (tid < 6) {
        const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        atomicAdd(&hs[tid < 3 ? tid : 4 + tid - 3], (double)v);
    }
}

// 2. per board: batch norm + ReLU of the head convolutions, both FC layers, losses, dL/dlogits, gradient back to the
//    head activations (hD = dL/d(bn output), ReLU mask applied) and its batch-norm-backward sums
template <int S>
__global__ __launch_bounds__(256) void head_loss_kernel(TrainDev T, const float *__restrict__ target_policy,
                                                         const long long *__restrict__ target_value, int sl_mode,
                                                         float value_weight) {
    TG_GEO(S);
    __shared__ float h[3 * P];            // [2 P policy | P value]
    __shared__ float logit[A + 3], dl[A + 3];
    __shared__ float hc[12];              // head bn: scale[3], shift[3], mean[3], rstd[3]
    __shared__ float wred[4][6];
    const int tid = threadIdx.x;
    TG_PROF(3, 0);
    if (tid < 3) {
        const double n = (double)(T.B * T.P);
        const double dmean = stat_sum(T.hstat + tid, 16) / n;
        const float mean = (float)dmean;
        const float rstd = (float)(1.0 / sqrt(fmax(stat_sum(T.hstat + 4 + tid, 16) / n - dmean * dmean, 0.0) + 2e-5));
        const float gamma = tid < 2 ? T.param[T.L.p_bn_w + tid] : T.param[T.L.v_bn_w];
        const float beta = tid < 2 ? T.param[T.L.p_bn_b + tid] : T.param[T.L.v_bn_b];
        hc[tid] = gamma * rstd;
        hc[3 + tid] = beta - mean * gamma * rstd;
        hc[6 + tid] = mean;
        hc[9 + tid] = rstd;
    }
    __syncthreads();
    TG_PROF(3, 1);
    float lp = 0.f, lv = 0.f, s1[3] = {0.f, 0.f, 0.f}, s2[3] = {0.f, 0.f, 0.f};
    for (int b = blockIdx.x; b < T.B; b += gridDim.x) {
        for (int e = tid; e < 3 * P; e += 256) {
            const int k = e / P, p = e - k * P;
            const float v = fmaxf(fmaf(T.hz[((size_t)b * P + p) * 4 + k], hc[k], hc[3 + k]), 0.f);
            h[e] = v;
            T.hact[(size_t)b * 3 * P + e] = v;
        }
        __syncthreads();
        TG_PROF(3, 2);
        {
            // both FC layers: wave w takes the outputs w, w + 4, ... (82 policy logits, 3 value logits), its lanes the inputs
            // (coalesced weight rows; a thread per OUTPUT read 64 scattered lines per instruction), butterfly per output
            const int wv = tid >> 6, ln = tid & 63;
            // in phases over a group of the wave's outputs - every load first, then the products, then the butterflies level by
            // level - so that the chains overlap (written output by output hipcc keeps them in program order: 15 us).  9x9: all
            // 22 outputs of a wave in one group, three weights per lane and output; 19x19: groups of 6, twelve weights
            constexpr int NJ = (2 * P + 63) / 64;                       // inputs per lane (policy rows are the longer ones)
            constexpr int NOW = (A + 3 + 3) / 4;                        // outputs per wave
            constexpr int NO = S == 9 ? NOW : 6;                        // ... per group
            float hp[NJ], hv[NJ];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = 64 * jj + ln;
                hp[jj] = j < 2 * P ? h[j] : 0.f;
                hv[jj] = j < P ? h[2 * P + j] : 0.f;
            }
            for (int g0 = 0; g0 < NOW; g0 += NO) {
                float wgt[NO][NJ], bias[NO], acc[NO];
#pragma unroll
                for (int i = 0; i < NO; ++i) {
                    const int a = wv + 4 * (g0 + i), a2 = a < A + 3 ? a : A + 2;
                    const bool pol = a2 < A;
                    const float *w = pol ? T.param + T.L.p_fc_w + (size_t)a2 * 2 * P : T.param + T.L.v_fc_w + (size_t)(a2 - A) * P;
                    const int n = pol ? 2 * P : P;
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj) wgt[i][jj] = w[64 * jj + ln < n ? 64 * jj + ln : 0];
                    bias[i] = pol ? T.param[T.L.p_fc_b + a2] : T.param[T.L.v_fc_b + a2 - A];
                }
#pragma unroll
                for (int i = 0; i < NO; ++i) {
                    const bool pol = wv + 4 * (g0 + i) < A;
                    const int n = pol ? 2 * P : P;
                    float sacc = 0.f;
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj) {
                        const float x = 64 * jj + ln < n ? (pol ? hp[jj] : hv[jj]) : 0.f;
                        sacc = jj == 0 ? x * wgt[i][0] : fmaf(x, wgt[i][jj], sacc);
                    }
                    acc[i] = sacc;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    float t[NO];
#pragma unroll
                    for (int i = 0; i < NO; ++i) t[i] = __shfl_xor(acc[i], o);
#pragma unroll
                    for (int i = 0; i < NO; ++i) acc[i] += t[i];
                }
                if (ln == 0) {
#pragma unroll
                    for (int i = 0; i < NO; ++i)
                        if (wv + 4 * (g0 + i) < A + 3) logit[wv + 4 * (g0 + i)] = acc[i] + bias[i];
                }
            }
        }
        __syncthreads();
        TG_PROF(3, 3);
        if (tid < 64) {
            // policy: log-softmax, loss, dL/dlogit - lane a takes actions a, a + 64, ..., sums by butterfly over the wave
            // (one lane walking the 82 actions three times was 22 of this kernel's 44 us)
            auto wsum = [](float v) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                return v;
            };
            constexpr int RA = (A + 63) / 64;                            // actions per lane: a = lane + 64 r
            float lg_[RA], tg_[RA];
            const float *tp = target_policy + (size_t)b * A;
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < RA; ++r) {
                const int a = tid + 64 * r;
                lg_[r] = a < A ? logit[a] : -INFINITY;
                tg_[r] = a < A ? tp[a] : 0.f;
                m = fmaxf(m, lg_[r]);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            float se_l = 0.f;
#pragma unroll
            for (int r = 0; r < RA; ++r) se_l += tid + 64 * r < A ? expf(lg_[r] - m) : 0.f;
            const float lse = m + logf(wsum(se_l));
            const float invb = 1.f / (float)T.B;
            float part = 0.f;
            if (!sl_mode) {
                // kl_div(logp, t, batchmean): sum t * (log t - logp) / B (0 where t == 0); gradient (p * sum t - t) / B
                float st_l = 0.f;
#pragma unroll
                for (int r = 0; r < RA; ++r) st_l += tg_[r];
                const float st = wsum(st_l);
#pragma unroll
                for (int r = 0; r < RA; ++r) {
                    const int a = tid + 64 * r;
                    if (a < A) {
                        const float lpr = lg_[r] - lse;
                        if (tg_[r] > 0.f) part += tg_[r] * (logf(tg_[r]) - lpr);
                        dl[a] = (expf(lpr) * st - tg_[r]) * invb;
                    }
                }
            } else {
                // -sum t * log(softmax + 1e-8) per sample, mean over the batch
                float pr[RA], dot_l = 0.f;
#pragma unroll
                for (int r = 0; r < RA; ++r) {
                    pr[r] = tid + 64 * r < A ? expf(lg_[r] - lse) : 0.f;
                    if (tid + 64 * r < A) {
                        part -= tg_[r] * logf(pr[r] + 1e-8f);
                        dot_l += tg_[r] * pr[r] / (pr[r] + 1e-8f);
                    }
                }
                const float dot = wsum(dot_l);
#pragma unroll
                for (int r = 0; r < RA; ++r)
                    if (tid + 64 * r < A) dl[tid + 64 * r] = (pr[r] * dot - tg_[r] * pr[r] / (pr[r] + 1e-8f)) * invb;
            }
            const float loss = wsum(part);
            if (tid == 0) {
                lp += loss;
                // value: cross entropy against the class
                float vm = fmaxf(logit[A], fmaxf(logit[A + 1], logit[A + 2]));
                float ve = expf(logit[A] - vm) + expf(logit[A + 1] - vm) + expf(logit[A + 2] - vm);
                const float vlse = vm + logf(ve);
                const int cls = (int)target_value[b];
                lv += vlse - logit[A + cls];
                for (int k = 0; k < 3; ++k)
                    dl[A + k] = value_weight * (expf(logit[A + k] - vlse) - (k == cls ? 1.f : 0.f)) * invb;
            }
        }
        __syncthreads();
        TG_PROF(3, 4);
        for (int a = tid; a < A + 3; a += 256) T.dlog[(size_t)b * (A + 3) + a] = dl[a];
        // back through the FC layers to the head activations, ReLU mask, D + sums
        for (int e = tid; e < 3 * P; e += 256) {
            const int k = e / P, p = e - k * P;
            const float hzv = T.hz[((size_t)b * P + p) * 4 + k];
            float g = 0.f;
            if (k < 2) {
                constexpr int kUnroll = S == 9 ? A : 16;           // 9x9: all 82 loads in flight
#pragma unroll kUnroll
                for (int a = 0; a < A; ++a) g = fmaf(dl[a], T.param[T.L.p_fc_w + (size_t)a * 2 * P + e], g);
            } else {
                for (int c = 0; c < 3; ++c) g = fmaf(dl[A + c], T.param[T.L.v_fc_w + (size_t)c * P + p], g);
            }
            g = h[e] > 0.f ? g : 0.f;
            T.hD[((size_t)b * P + p) * 4 + k] = g;
            const float xh = (hzv - hc[6 + k]) * hc[9 + k];
            s1[k] += g;
            s2[k] = fmaf(g, xh, s2[k]);
        }
        __syncthreads();
    }
    TG_PROF(3, 5);
    // every thread contributes to every k (its elements may span channels): butterfly over the wave, the four waves through LDS
    // (256 x 6 same-address LDS atomics serialised)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1[k] += __shfl_xor(s1[k], o); s2[k] += __shfl_xor(s2[k], o); }
    }
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { wred[tid >> 6][k] = s1[k]; wred[tid >> 6][3 + k] = s2[k]; }
    }
    __syncthreads();
    double *hs = T.hstat + (blockIdx.x % kRep) * 16;
    if (tid < 6) {
        const float v = (wred[0][tid] + wred[1][tid]) + (wred[2][tid] + wred[3][tid]);
        atomicAdd(&hs[tid < 3 ? 8 + tid : 12 + tid - 3], (double)v);
    }
    if (tid == 0) {
        const double pol = (double)lp / T.B, val = (double)lv / T.B;      // this workgroup's share of the batch means
        double *ls = T.loss + (blockIdx.x % kRep) * 4;
        atomicAdd(&ls[1], pol);
        atomicAdd(&ls[2], val);
        atomicAdd(&ls[0], pol + (double)value_weight * val);
    }
    TG_PROF(3, 6);
}

// 3. FC weight / bias gradients: dW[a][j] = sum_b dlog[b][a] * hact[b][j] (policy: j < 162; value rows a >= A: the last
//    81 of hact).  One workgroup per output a, thread (j, y): hact reads coalesced over j, dlog[b][a] a broadcast; the four y
//    take a quarter of the batch each, sixteen loads in flight (one thread walking the batch four loads at a time was 20 us of
//    memory latency), summed in y order through LDS.
template <int S>
__global__ __launch_bounds__(768) void head_fc_grad_kernel(TrainDev T) {
    TG_GEO(S);
    constexpr int NX = 192;                                  // blockDim.x; a policy row has 2 P inputs: 162 (one pass) / 722 (four)
    __shared__ float part[4][NX], partb[4];
    const int a = blockIdx.x, y = threadIdx.y;
    const bool pol = a < A;
    const int nj = pol ? 2 * P : P;
    const int per = (T.B + 3) / 4, b0 = y * per, b1 = b0 + per < T.B ? b0 + per : T.B;
    for (int j0 = 0; j0 < nj; j0 += NX) {
        const int j = j0 + threadIdx.x;
        float g = 0.f, gb = 0.f;
        if (j < nj) {
            const float *h = T.hact + (pol ? j : 2 * P + j);
            const float *d = T.dlog + a;
#pragma unroll 16
            for (int b = b0; b < b1; ++b) {
                const float dv = d[(size_t)b * (A + 3)];
                g = fmaf(dv, h[(size_t)b * 3 * P], g);
                gb += dv;
            }
        }
        __syncthreads();                                     // (the pass before has been read)
        part[y][threadIdx.x] = g;
        if (j == 0) partb[y] = gb;
        __syncthreads();
        if (y == 0 && j < nj) {
            T.grad[(pol ? T.L.p_fc_w + (size_t)a * 2 * P : T.L.v_fc_w + (size_t)(a - A) * P) + j] =
                (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
            if (j == 0) T.grad[pol ? T.L.p_fc_b + a : T.L.v_fc_b + a - A] = (partb[0] + partb[1]) + (partb[2] + partb[3]);
        }
    }
}

// 4. per board: batch-norm backward of the head convolutions, their weight gradients, gradient into Y_6,
//    ReLU mask -> D_12 and its sums.  Thread (row group g, channel c): coalesced over c, sums in registers.
template <int S>
__global__ __launch_bounds__(256) void head_back_kernel(TrainDev T) {
    TG_GEO(S);
    __shared__ float hc[15];              // gamma*rstd[3], mean[3], rstd[3], m1[3], m2[3]
    __shared__ float dzs[P * 4];          // dZ of the three head channels, per row
    __shared__ float red[4][5][64];       // per row group: S1, S2, three weight-gradient rows
    const int tid = threadIdx.x, c = tid & 63, g = tid >> 6;
    if (tid < 3) {
        const double n = (double)(T.B * T.P);
        const double dmean = stat_sum(T.hstat + tid, 16) / n;
        const float mean = (float)dmean;
        const float rstd = (float)(1.0 / sqrt(fmax(stat_sum(T.hstat + 4 + tid, 16) / n - dmean * dmean, 0.0) + 2e-5));
        const float gamma = tid < 2 ? T.param[T.L.p_bn_w + tid] : T.param[T.L.v_bn_w];
        hc[tid] = gamma * rstd;
        hc[3 + tid] = mean;
        hc[6 + tid] = rstd;
        hc[9 + tid] = (float)(stat_sum(T.hstat + 8 + tid, 16) / n);
        hc[12 + tid] = (float)(stat_sum(T.hstat + 12 + tid, 16) / n);
    }
    float mean12, rstd12;
    bn_consts(T, 12, c, 2e-5f, mean12, rstd12);
    const float w0 = T.param[T.L.p_conv + c], w1 = T.param[T.L.p_conv + 64 + c], w2 = T.param[T.L.v_conv + c];
    __syncthreads();
    const size_t bstride = (size_t)P * C;
    float s1 = 0.f, s2 = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int b = blockIdx.x; b < T.B; b += gridDim.x) {
        for (int e = tid; e < P * 3; e += 256) {
            const int row = e / 3, k = e - row * 3;
            const float z = T.hz[((size_t)b * P + row) * 4 + k], d = T.hD[((size_t)b * P + row) * 4 + k];
            const float xh = (z - hc[3 + k]) * hc[6 + k];
            dzs[row * 4 + k] = hc[k] * (d - hc[9 + k] - xh * hc[12 + k]);
        }
        __syncthreads();
        const float *y6 = T.Y + ((size_t)6 * T.B + b) * bstride;
        const float *z12 = T.Z + ((size_t)12 * T.B + b) * bstride;
        float *d12 = T.D + ((size_t)12 * T.B + b) * bstride;
        for (int row = g; row < P; row += 4) {
            const float yv = y6[row * C + c], zv = z12[row * C + c];
            const float d0 = dzs[row * 4], d1 = dzs[row * 4 + 1], d2 = dzs[row * 4 + 2];
            float v = d0 * w0 + d1 * w1 + d2 * w2;
            v = yv > 0.f ? v : 0.f;
            d12[row * C + c] = v;
            s1 += v;
            s2 = fmaf(v, (zv - mean12) * rstd12, s2);
            g0 = fmaf(d0, yv, g0);
            g1 = fmaf(d1, yv, g1);
            g2 = fmaf(d2, yv, g2);
        }
        __syncthreads();
    }
    red[g][0][c] = s1; red[g][1][c] = s2; red[g][2][c] = g0; red[g][3][c] = g1; red[g][4][c] = g2;
    __syncthreads();
    if (tid < 64) {
        float r[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) r[q] = (red[0][q][c] + red[1][q][c]) + (red[2][q][c] + red[3][q][c]);
        double *st12 = T.stat + ((size_t)12 * kRep + blockIdx.x % kRep) * 256;
        atomicAdd(&st12[128 + c], (double)r[0]);
        atomicAdd(&st12[192 + c], (double)r[1]);
        atomicAdd(&T.grad[T.L.p_conv + c], r[2]);
        atomicAdd(&T.grad[T.L.p_conv + 64 + c], r[3]);
        atomicAdd(&T.grad[T.L.v_conv + c], r[4]);
    }
}

// ---- optimiser: torch.optim.SGD(momentum, weight_decay, nesterov) over every parameter, running statistics ------
struct SgdArgs { float lr, momentum, weight_decay; int first_step; };

// number of parameters that are NOT convolution weights (those are sgd_conv_kernel's): the stem's batch norm, two batch norms per
// block, everything from the head convolutions on - 4 % of the blob; the kernel is launched over these only
__host__ __device__ inline size_t sgd_small_count(const Layout &L) { return 256 + 6 * 512 + (L.total - L.p_conv); }

__global__ void sgd_kernel(TrainDev T, SgdArgs a) {
    const Layout &L = T.L;
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= sgd_small_count(L)) return;
    // k-th non-convolution element -> its place in the blob
    const size_t i = k < 256 ? L.bn_w[0] + k : k < 256 + 6 * 512 ? L.bn_w[1 + 2 * ((k - 256) / 512)] + (k - 256) % 512 : L.p_conv + (k - 256 - 6 * 512);
    // which tensor is element i in?
    float g;
    bool is_stat = false;
    int l = -1, kind = -1;       // kind 0 conv weight, 1 bn weight, 2 bn bias, 3 running mean, 4 running var
    for (int k = 0; k < kLayers; ++k) {
        const size_t wn = k == 0 ? 64 * 6 * 9 : kConvW;
        if (i >= L.conv[k] && i < L.conv[k] + wn) { l = k; kind = 0; break; }
        if (i >= L.bn_w[k] && i < L.bn_w[k] + 64) { l = k; kind = 1; break; }
        if (i >= L.bn_b[k] && i < L.bn_b[k] + 64) { l = k; kind = 2; break; }
        if (i >= L.bn_m[k] && i < L.bn_m[k] + 64) { l = k; kind = 3; break; }
        if (i >= L.bn_v[k] && i < L.bn_v[k] + 64) { l = k; kind = 4; break; }
    }
    const double n = (double)(T.B * T.P);
    if (kind == 0) {
        return;                                                       // convolution weights: sgd_conv_kernel
    } else if (kind == 1) {
        g = (float)stat_sum(T.stat + (size_t)l * kRep * 256 + 192 + (i - L.bn_w[l]), 256);   // d gamma = S2
    } else if (kind == 2) {
        g = (float)stat_sum(T.stat + (size_t)l * kRep * 256 + 128 + (i - L.bn_b[l]), 256);   // d beta = S1
    } else if (kind == 3 || kind == 4) {
        const int c = (int)(i - (kind == 3 ? L.bn_m[l] : L.bn_v[l]));
        const float m = l == 0 ? 0.1f : 0.01f;
        const double mean = stat_sum(T.stat + (size_t)l * kRep * 256 + c, 256) / n;
        const double var = fmax(stat_sum(T.stat + (size_t)l * kRep * 256 + 64 + c, 256) / n - mean * mean, 0.0);
        T.param[i] = kind == 3 ? (1.f - m) * T.param[i] + m * (float)mean
                               : (1.f - m) * T.param[i] + m * (float)(var * (n / (n - 1.0)));
        is_stat = true;
        g = 0.f;
    } else {
        // heads
        auto in = [&](size_t off, size_t cnt) { return i >= off && i < off + cnt; };
        if (in(L.p_bn_w, 2)) g = (float)stat_sum(T.hstat + 12 + (i - L.p_bn_w), 16);
        else if (in(L.p_bn_b, 2)) g = (float)stat_sum(T.hstat + 8 + (i - L.p_bn_b), 16);
        else if (in(L.v_bn_w, 1)) g = (float)stat_sum(T.hstat + 12 + 2, 16);
        else if (in(L.v_bn_b, 1)) g = (float)stat_sum(T.hstat + 8 + 2, 16);
        else if (in(L.p_bn_m, 2) || in(L.p_bn_v, 2) || in(L.v_bn_m, 1) || in(L.v_bn_v, 1)) {
            const bool is_mean = in(L.p_bn_m, 2) || in(L.v_bn_m, 1);
            const int k = in(L.p_bn_m, 2) ? (int)(i - L.p_bn_m) : in(L.p_bn_v, 2) ? (int)(i - L.p_bn_v) : 2;
            const double mean = stat_sum(T.hstat + k, 16) / n;
            const double var = fmax(stat_sum(T.hstat + 4 + k, 16) / n - mean * mean, 0.0);
            T.param[i] = is_mean ? 0.99f * T.param[i] + 0.01f * (float)mean
                                 : 0.99f * T.param[i] + 0.01f * (float)(var * (n / (n - 1.0)));
            is_stat = true;
            g = 0.f;
        } else {
            g = T.grad[i];                                             // FC layers, head convolutions
        }
    }
    if (is_stat) return;
    const float p = T.param[i];
    g = fmaf(a.weight_decay, p, g);
    float buf = a.first_step ? g : fmaf(a.momentum, T.mom[i], g);
    T.mom[i] = buf;
    T.param[i] = p - a.lr * fmaf(a.momentum, buf, g);
}

// convolution weights: thread e walks the partial images [wg][tap][co][ci] in their own order (coalesced), sums
// them over the workgroups and updates the parameter it belongs to
__global__ void sgd_conv_kernel(TrainDev T, SgdArgs a) {
    const int l = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 36864) return;
    const int ci = e & 63, co = (e >> 6) & 63, tap = e >> 12;
    if (l == 0 && ci >= 6) return;
    const float *p = T.partial + (size_t)l * T.WCH * 36864 + e;
    float g = 0.f;
#pragma unroll 8
    for (int w = 0; w < T.WCH; ++w) g += p[(size_t)w * 36864];
    const size_t i = T.L.conv[l] + (l == 0 ? (size_t)(co * 6 + ci) * 9 + tap : (size_t)(co * 64 + ci) * 9 + tap);
    const float w0 = T.param[i];
    g = fmaf(a.weight_decay, w0, g);
    const float buf = a.first_step ? g : fmaf(a.momentum, T.mom[i], g);
    T.mom[i] = buf;
    T.param[i] = w0 - a.lr * fmaf(a.momentum, buf, g);
}

}  // namespace

struct tg_trainer {
    int device = 0, batch = 0, size = 9;
    TrainDev dev{};
    std::vector<void *> allocs;
    bool first_step = true;
};

namespace {
template <typename T>
int talloc(tg_trainer *t, T **out, size_t count) {
    void *p = nullptr;
    TG_HIP(hipMalloc(&p, count * sizeof(T)));
    TG_HIP(hipMemset(p, 0, count * sizeof(T)));
    TG_HIP(hipStreamSynchronize(nullptr));      // (the fill is queued on the null stream: not ordered before a non-blocking stream's step)
    t->allocs.push_back(p);
    *out = static_cast<T *>(p);
    return TG_OK;
}
template <int S> constexpr int kConvLds = ((S + 2) * (S + 2) * kRow + 9 * 64) * 4;
template <int S> constexpr int kWgradLds = WGeo<S>::kLdsFloats * 4;

template <int S>
int configure_kernels() {
    static_assert(kConvLds<S> <= 160 * 1024 && kWgradLds<S> <= 160 * 1024, "LDS");
    TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_kernel<FWD, S>), hipFuncAttributeMaxDynamicSharedMemorySize, kConvLds<S>));
    TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_kernel<DGRAD, S>), hipFuncAttributeMaxDynamicSharedMemorySize, kConvLds<S>));
    TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, kWgradLds<S>));
    return TG_OK;
}

// one mini-batch: ~45 launches on the caller's stream
template <int S>
int launch_step(tg_trainer *t, const float *planes_dev, const float *policy_dev, const long long *value_dev, int sl_mode,
                float value_weight, float lr, hipStream_t st) {
    TrainDev &D = t->dev;
    const int grid = D.NWG, A = S * S + 1;
    // 19x19 below the CU count: a board's four passes of the MFMA loop on four workgroups
    constexpr int kPasses = ((S * S + 15) / 16 + 5) / 6;
    const int per_board = kPasses > 1 && D.B * kPasses <= 256 ? kPasses : 1;
    hipLaunchKernelGGL(repack_kernel, dim3(36864 / 256, kLayers), dim3(256), 0, st, D);
    for (int l = 0; l < kLayers; ++l)
        hipLaunchKernelGGL((conv_kernel<FWD, S>), dim3(grid * per_board), dim3(256), kConvLds<S>, st, D, planes_dev, l, per_board);
    hipLaunchKernelGGL(head_conv_kernel<S>, dim3(grid), dim3(256), 0, st, D);
    hipLaunchKernelGGL(head_loss_kernel<S>, dim3(grid), dim3(256), 0, st, D, policy_dev, value_dev, sl_mode, value_weight);
    hipLaunchKernelGGL(head_fc_grad_kernel<S>, dim3(A + 3), dim3(192, 4), 0, st, D);
    hipLaunchKernelGGL(head_back_kernel<S>, dim3(grid), dim3(256), 0, st, D);
    // backward.  (Measured and dropped: wgrad of layer l on a second stream beside the data-gradient chain - the two kernels do
    // run side by side, 34 + 32 us overlapping into 41 instead of 24 + 20 one after the other, but the event between two
    // launches of the chain costs 8 us on its stream: 0.91 against 0.93 ms per step.)
    for (int l = kLayers - 1; l >= 0; --l) {
        hipLaunchKernelGGL(wgrad_kernel<S>, dim3(4 * D.WCH), dim3(512), kWgradLds<S>, st, D, planes_dev, l);
        if (l >= 1) hipLaunchKernelGGL((conv_kernel<DGRAD, S>), dim3(grid * per_board), dim3(256), kConvLds<S>, st, D, planes_dev, l, per_board);
    }
    SgdArgs a{lr, 0.9f, 1e-4f, t->first_step ? 1 : 0};
    hipLaunchKernelGGL(sgd_conv_kernel, dim3(36864 / 256, kLayers), dim3(256), 0, st, D, a);
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((sgd_small_count(D.L) + 255) / 256)), dim3(256), 0, st, D, a);
    TG_HIP(hipGetLastError());
    t->first_step = false;
    return TG_OK;
}
}  // namespace

extern "C" {

int tg_trainer_create(int board_size, int device, int batch, const float *params_host, size_t n_params, tg_trainer **out) {
    if (!params_host || !out) return tg::fail(TG_ERR_ARG, "tg_trainer_create: null argument");
    if (board_size != 9 && board_size != 19)
        return tg::fail(TG_ERR_ARG, "tg_trainer_create: the HIP training step is built for 9x9 and 19x19 boards, not %d", board_size);
    if (batch < 2) return tg::fail(TG_ERR_ARG, "tg_trainer_create: batch must be >= 2 (batch statistics)");
    const Layout L = make_layout(board_size);
    const int P = board_size * board_size, A = P + 1;
    if (n_params != L.total) return tg::fail(TG_ERR_ARG, "tg_trainer_create: expected %zu parameters, got %zu", L.total, n_params);
    TG_HIP(hipSetDevice(device));
    tg_trainer *t = new tg_trainer;
    t->device = device;
    t->batch = batch;
    t->size = board_size;
    TrainDev &D = t->dev;
    D.L = L;
    D.B = batch;
    D.P = P;
    D.NWG = batch < 256 ? batch : 256;
    D.WCH = batch < 64 ? batch : 64;
    const size_t act = (size_t)batch * P * C;
    int rc = TG_OK;
    if ((rc = talloc(t, &D.param, L.total)) || (rc = talloc(t, &D.grad, L.total)) || (rc = talloc(t, &D.mom, L.total)) ||
        (rc = talloc(t, &D.wf, (size_t)kLayers * 36864)) || (rc = talloc(t, &D.wb, (size_t)kLayers * 36864)) ||
        (rc = talloc(t, &D.Z, kLayers * act)) || (rc = talloc(t, &D.Y, 7 * act)) || (rc = talloc(t, &D.D, kLayers * act)) ||
        (rc = talloc(t, &D.stat, (size_t)kLayers * kRep * 256)) || (rc = talloc(t, &D.partial, (size_t)kLayers * D.WCH * 36864)) ||
        (rc = talloc(t, &D.hz, (size_t)batch * P * 4)) || (rc = talloc(t, &D.hD, (size_t)batch * P * 4)) ||
        (rc = talloc(t, &D.hstat, (size_t)kRep * 16)) || (rc = talloc(t, &D.hact, (size_t)batch * 3 * P)) ||
        (rc = talloc(t, &D.dlog, (size_t)batch * (A + 3))) || (rc = talloc(t, &D.loss, (size_t)kRep * 4))) {
        for (void *p : t->allocs) (void)hipFree(p);
        delete t;
        return rc;
    }
    TG_HIP(hipMemcpy(D.param, params_host, L.total * sizeof(float), hipMemcpyHostToDevice));
    if ((rc = board_size == 9 ? configure_kernels<9>() : configure_kernels<19>())) {
        for (void *p : t->allocs) (void)hipFree(p);
        delete t;
        return rc;
    }
    *out = t;
    return TG_OK;
}

int tg_trainer_destroy(tg_trainer *t) {
    if (!t) return TG_OK;
    (void)hipSetDevice(t->device);
    for (void *p : t->allocs) (void)hipFree(p);
    delete t;
    return TG_OK;
}

int tg_trainer_step(tg_trainer *t, const float *planes_dev, const float *policy_dev, const long long *value_dev,
                    int sl_mode, float value_weight, float lr, void *stream) {
    if (!t || !planes_dev || !policy_dev || !value_dev) return tg::fail(TG_ERR_ARG, "tg_trainer_step: null argument");
    TG_HIP(hipSetDevice(t->device));          // the launches below belong to the trainer's device whatever the caller's is
    hipStream_t st = static_cast<hipStream_t>(stream);
    return t->size == 9 ? launch_step<9>(t, planes_dev, policy_dev, value_dev, sl_mode, value_weight, lr, st)
                        : launch_step<19>(t, planes_dev, policy_dev, value_dev, sl_mode, value_weight, lr, st);
}

int tg_trainer_read_losses(tg_trainer *t, double *sums_host, int reset) {
    if (!t || !sums_host) return tg::fail(TG_ERR_ARG, "tg_trainer_read_losses: null argument");
    TG_HIP(hipSetDevice(t->device));
    TG_HIP(hipDeviceSynchronize());
    double rep[kRep * 4];
    TG_HIP(hipMemcpy(rep, t->dev.loss, sizeof(rep), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) {
        sums_host[k] = 0.0;
        for (int r = 0; r < kRep; ++r) sums_host[k] += rep[r * 4 + k];
    }
    if (reset) {
        TG_HIP(hipMemset(t->dev.loss, 0, sizeof(rep)));
        TG_HIP(hipStreamSynchronize(nullptr));  // (done before the caller queues the next step on whatever stream)
    }
    return TG_OK;
}

int tg_trainer_get_params(tg_trainer *t, float *params_host, float *momentum_host, size_t n) {
    if (!t || !params_host) return tg::fail(TG_ERR_ARG, "tg_trainer_get_params: null argument");
    if (n != t->dev.L.total) return tg::fail(TG_ERR_ARG, "tg_trainer_get_params: expected %zu floats", t->dev.L.total);
    TG_HIP(hipSetDevice(t->device));
    TG_HIP(hipDeviceSynchronize());
    TG_HIP(hipMemcpy(params_host, t->dev.param, n * sizeof(float), hipMemcpyDeviceToHost));
    if (momentum_host) TG_HIP(hipMemcpy(momentum_host, t->dev.mom, n * sizeof(float), hipMemcpyDeviceToHost));
    return TG_OK;
}

int tg_trainer_debug_read(tg_trainer *t, int which, int index, float *out_host) {
    if (!t || !out_host) return tg::fail(TG_ERR_ARG, "tg_trainer_debug_read: null argument");
    TG_HIP(hipSetDevice(t->device));
    TG_HIP(hipDeviceSynchronize());
    const TrainDev &D = t->dev;
    const size_t act = (size_t)D.B * D.P * C;
#ifdef TG_TRAIN_PROF
    if (which == 3) {
        TG_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 64));
        return TG_OK;
    }
#endif
    const float *src = which == 0 ? D.Z + index * act : which == 1 ? D.Y + index * act : which == 2 ? D.D + index * act : nullptr;
    if (!src) return tg::fail(TG_ERR_ARG, "tg_trainer_debug_read: which must be 0 (Z), 1 (Y) or 2 (D)");
    TG_HIP(hipMemcpy(out_host, src, act * sizeof(float), hipMemcpyDeviceToHost));
    return TG_OK;
}

int tg_trainer_set_momentum(tg_trainer *t, const float *momentum_host, size_t n) {
    if (!t || !momentum_host) return tg::fail(TG_ERR_ARG, "tg_trainer_set_momentum: null argument");
    if (n != t->dev.L.total) return tg::fail(TG_ERR_ARG, "tg_trainer_set_momentum: expected %zu floats", t->dev.L.total);
    TG_HIP(hipSetDevice(t->device));
    TG_HIP(hipMemcpy(t->dev.mom, momentum_host, n * sizeof(float), hipMemcpyHostToDevice));
    t->first_step = false;
    return TG_OK;
}

}  // extern "C"
