// DualNet forward for gfx950, 19x19: the residual tower as Winograd F(2,3) along x on split operands (the arithmetic of
// net_forward_w1d.hip), ONE BOARD OVER TWO WORKGROUPS (round 5).
//
// Why two workgroups: the one-axis Winograd tower wants its activations as fp32 [cell][64 channels] in LDS, block input X
// (also the residual) and intermediate H - 2 x 92 KB for a 19x19 board, more than a CU has.  The direct split kernel
// (net_forward_split.hip) gets by with ONE f16-pair image and sends two thirds of the residual through a scratch image in
// global memory: 348 KB of fabric traffic per position against 10 KB of input and output, and 1 029 MFLOP of MFMAs per
// position.  Here a board is cut into two horizontal BANDS (rows 0-9 | 10-18), each band's X and H (band + one halo row,
// 2 x 53 KB) stay in LDS for the whole pass, 793 MFLOP of MFMAs are issued per position, and what crosses between the two
// workgroups is one edge row (19 cells x 256 B) per layer and direction.
//
// Geometry.  A band numbers its rows l = 0 (the row next to the partner band) .. 9 away from the cut; the partner's
// edge row is its halo row l = -1.  (Band 0: y = 9 - l, band 1: y = 10 + l; the three kernel taps along y are used in
// mirrored order by band 0.)  A band's ten rows are three CLASSES c of three rows (l = 3c + k) plus row 9 (band 0 only);
// its 19 columns are two HALVES of five Winograd tiles (outputs x = 2t, 2t + 1; t = 5 half + t').  One STAGE = 72 MFMAs
// per wave = one step k of one half: the sixteen MFMA columns are the units li = 5 c + t' - exactly the three-board 9x9
// kernel's columns 5 board + t, with a row class where that kernel has a board.  As there, wave w owns transform point w,
// its 48 weight fragments of a layer sit in AGPRs, a lane's V rows (transformed input rows, f16 hi / lo pieces) serve three
// steps from registers, M_w goes through the LDS exchange and wave w' finishes channels [16 w', 16 w' + 16).  Row 9 of
// band 0 is a seventh stage of its own column mapping (li = tile).  A layer is S, 2A, 2B, 1A, 1B, 0A, 0B (step, half):
// descending steps, so that the edge row l = 0 - the only one that needs the halo - comes last and a layer's edge row has
// a whole layer to travel before the partner's last stages need it.
//
// Hand-off (measured first: tools/microbench/xwg_pingpong.hip, 1.3 us one way): the epilogue lanes that hold edge cells
// store them to the pair's exchange area as well (16 bytes, sc1), already in the partner's LDS layout; once a full vmcnt(0)
// has passed, one lane publishes a sequence number (relaxed, agent scope); the partner checks it a stage after asking for
// it, copies the row with LDS-DMA loads (sc1) and waits for them before the barrier in front of the first read.  Waits
// are bounded: a partner that does not show up raises the range flag - the exact kernel redoes the batch, nothing hangs.
//
// Heads: the 1x1 convolutions run here (fp32, per cell), the features go to global memory and dualnet_heads19_kernel -
// a batched fp32 product over all boards of the launch - does the two fully connected layers and the softmaxes: the 1 MB
// of policy-FC weights is streamed once per 16 boards instead of once per board.
// Reference: nn/network/dual_net.py:41-106, nn/network/res_block.py:8-38 at BOARD_SIZE = 19 (board/constant.py:4).
#include "w1d_common.h"

// WB_ABL (experiments only, tools/experiments/wb_ablation.sh: results are wrong, the timing says what a class of riders costs):
// 1 no weight requests, 2 no input transforms (and their cell reads), 4 no cell reads, 8 no epilogue arithmetic, 16 no exchange
// traffic, 32 no stage barrier, 64 no hand-off between the bands, 128 no MFMAs
#ifndef WB_ABL
#define WB_ABL 0
#endif

namespace {

constexpr int kWbSpinLimit = 1 << 17;                      // polls of the partner's sequence number (~ 0.1 s) before giving up

struct WbCfg {
    static constexpr int S = 19, P = 361, A = 362;
    static constexpr int LR = 11, CELLS = LR * S;          // local rows: halo + ten; cell (l, x) = (l + 1) * 19 + x
    static constexpr int NTHR = 256, NW = 4;
    static constexpr int DUMP_REL = CELLS * 256, ZERO_REL = (CELLS + 1) * 256;
    static constexpr int BUF = (CELLS + 2) * 256;          // 54 016
    static constexpr int X_OFF = 0, H_OFF = BUF;
    static constexpr int EX_OFF = 2 * BUF;                 // exchange [parity 2][wave 4][ct 4][lane 64][16 B]
    static constexpr int EX_BYTES = 32768;
    static constexpr int HW_OFF = EX_OFF + EX_BYTES;       // head 1x1 weights [64][4] (policy 0, policy 1, value, 0)
    static constexpr int HS_OFF = HW_OFF + 64 * 4 * 4;     // head batch norm: scale / shift x 3
    static constexpr int MISC = HS_OFF + 32;               // [0] the partner did not show up
    static constexpr int SH_OFF = MISC + 16;               // the tower's folded shifts [12][64] + 2^-e [12] (+ padding)
    static constexpr int PROF_OFF = SH_OFF + 12 * 64 * 4 + 64;   // 32 s_memtime stamps (profiling builds)
    static constexpr int LDS_BYTES = PROF_OFF + 512;
    // stem overlay (over H and the exchange): im2col'ed input of the band's cells as f16-pair images + the board's planes
    static constexpr int MT = (CELLS + 15) / 16;           // 14 row tiles
    static constexpr int RTW = (MT + NW - 1) / NW;         // 4 per wave
    static constexpr int ZOFF = ((RTW * NW * 16 + 1) * 64 + 255) & ~255;
    static constexpr int IMG = ZOFF + 256;
    static constexpr int SI_OFF = H_OFF;
    static constexpr int STAGE = SI_OFF + 4 * IMG;
    static constexpr int SS_OFF = (STAGE + 6 * P * 4 + 15) & ~15;
    static constexpr int ROWB = S * 256;                   // bytes per local row
    // per pair, in the per-stream scratch: edge rows [band 2][parity 2][19 x 64 floats] + sequence numbers [band 2] (+ padding)
    static constexpr int XROW_FLOATS = S * 64;
    static constexpr int PAIR_FLOATS = 4 * XROW_FLOATS + 64;
    static_assert(SS_OFF + 512 <= EX_OFF + EX_BYTES, "stem overlay");
    static_assert(LDS_BYTES <= 163840, "LDS");
};

// 16-byte chunk XOR of cell (l, x): g = (5 floor(l / 3) + (x + 1) / 2) mod 8 on chunk-index bits 0, 2, 3.  For the units of a
// stage (l = 3 c + s, x = 2 (t' + 5 half) + e: s, half, e common) g = li + const: distinct over the eight lanes of either
// half of a ds_read_b128 cycle.  The same for row 9's stage (li = tile).
__host__ __device__ inline int wb_swz(int l, int x) {
    const int fd = l < 0 ? -1 : l / 3;
    const int g = (5 * fd + ((x + 1) >> 1) + 40) & 7;
    return (g & 1) | ((g & 6) << 1);
}

// What rides along the MFMAs of the stages of a layer (n = 0 .. 6: S, 2A, 2B, 1A, 1B, 0A, 0B).
struct WbStage {
    int k, h;            // step (3: row 9's stage), half
    int ord;             // tap order: 0 = (-1, 0, +1), 1 = (+1, 0, -1), 2 = (0, +1, -1)
    // transform jobs: code = 100 + 10 * (s + 1) + half for V_half[s] of the regular mapping (s = -1 .. 3), 200 + r for row r = 8 / 9 of
    // row 9's mapping; + 1000 when the source is this layer's OUTPUT (a V row of the next layer); 0 = none
    int early, after, late;
};
constexpr WbStage wb_stage(int n) {
    switch (n) {
    case 0: return {3, 0, 0, 100 + 30 + 0, 100 + 20 + 0, 0};                         // S:  V_A[2], V_A[1]
    case 1: return {2, 0, 0, 100 + 20 + 1, 100 + 40 + 0, 100 + 30 + 1};              // 2A: V_B[1], V_A[3] (rows 3 / 6: stored by S, behind the barrier), V_B[2]
    case 2: return {2, 1, 0, 0, 100 + 40 + 1, 0};                                    // 2B: V_B[3]
    case 3: return {1, 0, 1, 0, 100 + 10 + 0, 100 + 10 + 1};                         // 1A: V_A[0] (used last: taps +1, 0, -1), V_B[0]
    case 4: return {1, 1, 0, 0, 0, 0};                                               // 1B: (the halo copy)
    case 5: return {0, 0, 1, 0, 100 + 0 + 0, 1200 + 8};                              // 0A: V_A[-1] (the halo row: behind the barrier), the next layer's row 8
    default: return {0, 1, 2, 0, 100 + 0 + 1, 1200 + 9};                             // 0B: V_B[-1], the next layer's row 9 (tap 0 first: its fragments are the next to be needed)
    }
}
// FIVE register slots (16 registers each: two k-chunks of f16 high / low pieces) hold the V rows of both halves and of row 9's
// stage; a row is written where the row before it in that slot has seen its last MFMA (slices of the stages in brackets):
//   P0: row 8 [S 0-23] -> V_A[3] (2A behind the barrier) [2A 48-71] -> V_B[3] (2B) [2B 48-71] -> V_A[0] (1A) [1A 48-71, 0A 24-47] -> row 8 (0A late)
//   P1: row 9 [S 24-47] -> V_B[2] (2A late) [2B, 1B] -> V_A[-1] (0A) [0A 48-71] -> row 9 (0B late)
//   P2: V_A[2] (S early) [2A, 1A 0-23] -> V_B[0] (1A late) [1B, 0B 0-23] -> V_A[2] ...
//   P3: V_A[1] (S behind its barrier) [2A, 1A, 0A 0-23] -> V_B[-1] (0B) [0B 48-71] -> V_A[1] ...
//   P4: V_B[1] (2A early) [2B, 1B, 0B 24-47] -> V_B[1] ...
// and every job reads rows a barrier has published: a layer's input rows 0 / 3 / 6 are its predecessor's last outputs (0A ->
// stored under 0B, 0B -> stored under S): the jobs that read them (V[0], V[3]) start behind 2A's barrier; row 9 is stored under 2A.
constexpr int wb_slot(int job) {                           // job code (without the + 1000 source flag) -> slot
    const int j = job % 1000;
    if (j >= 200) return j - 208;                          // rows 8, 9 -> P0, P1
    const int sr = (j - 100) / 10 - 1, h = (j - 100) % 10;
    constexpr int A[5] = {1, 0, 3, 2, 0}, B[5] = {3, 2, 4, 1, 0};
    return h == 0 ? A[sr + 1] : B[sr + 1];
}

// Weight requests, code = 16 kind + fragment (8 kc + 4 piece + ct).  A wave's request is 1 KB; the CU's vector L1 passes 64 B a
// clock, so four waves asking more often than every fourth slice (one MFMA = 16 clocks) fill the request queue and the
// MFMAs behind it wait (first version: twenty requests in 0B at every second slice - 0B took 2.5 x a stage).
//   kind 0: tap -1 of the NEXT layer -> the spare slot (double-buffered by layer parity), four a stage in 2B .. 0A
//   kind 1: tap 0 of the next layer, behind its last uses in 0B (taps in the order 0, +1, -1 there: free from slice 24; first
//           used at S's slice 24 - sixteen requests take the L1 1 024 clocks)
//   kind 2: tap +1 of THIS layer (free since the previous layer's 0B, first used at 2A's slice 48): twelve in S, four in 2A
constexpr int kWbTap0Order[16] = {4, 5, 6, 7, 0, 1, 2, 3, 12, 13, 14, 15, 8, 9, 10, 11};
constexpr int kWbTap0Slice0B[16] = {25, 28, 31, 34, 37, 40, 43, 46, 49, 52, 55, 58, 61, 64, 67, 70};
constexpr int wb_wreq(int n, int m) {
    if (n >= 2 && n <= 5 && (m == 50 || m == 56 || m == 62 || m == 68)) return 4 * (n - 2) + (m - 50) / 6;
    if (n == 6)
        for (int i = 0; i < 16; ++i)
            if (m == kWbTap0Slice0B[i]) return 16 + kWbTap0Order[i];
    if (n == 0 && m >= 1 && m <= 34 && (m - 1) % 3 == 0) return 32 + kWbTap0Order[(m - 1) / 3];
    if (n == 1 && m >= 1 && m <= 10 && (m - 1) % 3 == 0) return 32 + kWbTap0Order[12 + (m - 1) / 3];
    return -1;
}
constexpr int wb_count(int n0, int m0, int n1, int m1) {   // requests strictly behind (n0, m0) and before (n1, m1), going round the stages
    int cnt = 0;
    if (n0 == n1 && m1 > m0) {
        for (int m = m0 + 1; m < m1; ++m)
            if (wb_wreq(n0, m) >= 0) ++cnt;
        return cnt;
    }
    for (int m = m0 + 1; m < 72; ++m)
        if (wb_wreq(n0, m) >= 0) ++cnt;
    for (int n = (n0 + 1) % 7; n != n1; n = (n + 1) % 7)
        for (int m = 0; m < 72; ++m)
            if (wb_wreq(n, m) >= 0) ++cnt;
    for (int m = 0; m < m1; ++m)
        if (wb_wreq(n1, m) >= 0) ++cnt;
    return cnt;
}
constexpr int kWbWaitTop = wb_count(5, 68, 0, 0);          // S, slice 0: tap -1 (last request 0A / 68) - behind it 0B's sixteen
constexpr int kWbWaitTap0 = wb_count(6, 70, 0, 24);        // S, slice 24: tap 0 (last request 0B / 70) - behind it eight of tap +1
constexpr int kWbWaitTapP = wb_count(1, 10, 1, 48);        // 2A, slice 48: tap +1 (last request 2A / 10) - nothing behind it
static_assert(kWbWaitTop == 16 && kWbWaitTap0 == 8 && kWbWaitTapP == 0, "request schedule and wait counts");

template <bool PROF>
__global__ __launch_bounds__(256, 1) void dualnet_fwd_w1dband_kernel(
    NetDev net, const float *__restrict__ planes, int batch, float *__restrict__ feat, float *__restrict__ xmem,
    int *__restrict__ overflow, int *__restrict__ group_bits) {
    using C = WbCfg;
    using F = FmtF16;
    constexpr int S = C::S, P = C::P, NTHR = C::NTHR, RTW = C::RTW, IMG = C::IMG, ROWB = C::ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // band-major numbering: the two bands of a pair are n_pairs workgroups apart (the same XCD when n_pairs is a multiple of 8)
    const int n_pairs = gridDim.x >> 1;
    const int band = __builtin_amdgcn_readfirstlane((int)blockIdx.x >= n_pairs ? 1 : 0);
    const int pair = (int)blockIdx.x - band * n_pairs;
    const int lmax = band == 0 ? 9 : 8;                    // rows l = 0 .. lmax exist (band 1: nine rows)
    float *const pmem = xmem + (size_t)pair * C::PAIR_FLOATS;
    // sequence numbers: one per WAVE of a band ([band][4], 64 bytes apart): a wave publishes as soon as ITS stores have landed
    int *const seq_mine = reinterpret_cast<int *>(pmem + 4 * C::XROW_FLOATS) + band * 16;
    int *const seq_theirs = reinterpret_cast<int *>(pmem + 4 * C::XROW_FLOATS) + (1 - band) * 16;
    int *const dead = reinterpret_cast<int *>(smem + C::MISC);
    // test hook (TG_WB_TEST_MUTE, through the launch's second flag word): band 1 keeps its sequence numbers to itself - band 0 runs
    // into the bounded wait, the exact kernel redoes the batch
    const bool mute = overflow && __builtin_amdgcn_readfirstlane(overflow[1]) != 0 && band == 1;

    if (static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)smem)) != 0u)
        __builtin_trap();                                  // absolute LDS addressing below
    // ---- once per workgroup: dump + zero rows, head tables ----
    for (int e = tid; e < 2 * 2 * 64; e += NTHR) {
        const int buf = e >> 7, r = (e >> 6) & 1, c = e & 63;
        reinterpret_cast<float *>(smem + buf * C::BUF + (C::CELLS + r) * 256)[c] = 0.f;
    }
    for (int e = tid; e < 64 * 4; e += NTHR) {
        const int k = e >> 2, c = e & 3;
        reinterpret_cast<float *>(smem + C::HW_OFF)[e] = c == 0 ? net.hp_w[k] : (c == 1 ? net.hp_w[64 + k] : (c == 2 ? net.hv_w[k] : 0.f));
    }
    if (tid < 6) reinterpret_cast<float *>(smem + C::HS_OFF)[tid] = net.head_ss[tid];
    for (int e = tid; e < 12 * 64 + 12; e += NTHR)
        reinterpret_cast<float *>(smem + C::SH_OFF)[e] = e < 12 * 64 ? net.w1_shift[e] : net.w1_down[e - 12 * 64];
    if (tid == 0) *dead = 0;

    // PROF: s_memtime stamps of pair 0's first board (band 0: timeline[0 ..], band 1: timeline[64 ..]): [0] start, [1] stem done,
    // [2 + L] layer L done, [14] head convolutions done, [16 + n] stage n of layers 2 / 3 starts ([16 .. 22], [24 .. 30]), [23] / [31] their ends
    // (every workgroup stamps into its own LDS, wave 0 only - one s_memtime + one LDS store, no condition on registers the tower
    // would have to keep; pair 0 copies its stamps out after its first board)
    auto stamp = [&](int i) {
        if constexpr (PROF)
            if (wave == 0) {
                const long long t = (long long)__builtin_amdgcn_s_memtime();
                asm volatile("ds_write_b64 %0, %1" ::"v"(C::PROF_OFF + 8 * i), "v"(t) : "memory");
            }
    };
    int ovf = 0;
    constexpr int NPL = (6 * P + NTHR - 1) / NTHR;
    float pre[NPL];
    float ssv;
    auto fetch_planes = [&](int b) __attribute__((always_inline)) {
        const int ft = wave * 64 + fresh_lane();
        ssv = ft < 64 ? net.sscale[ft] : (ft < 128 ? net.shift[ft - 64] : 0.f);
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int e = ft + i * NTHR;
            pre[i] = (e < 6 * P && b < batch) ? __builtin_nontemporal_load(&planes[(size_t)b * 6 * P + e]) : 0.f;
        }
    };
    fetch_planes(pair);
    // Which way the edge rows travel.  HIP promises nothing about where a workgroup runs (observed: workgroup b on XCD b mod 8 - the
    // launch numbers partners eight apart for that), so the two bands TELL each other: each publishes its XCD (agent scope), reads
    // the partner's behind the first board's stem, and both take the L2 path - plain stores, reads past the vector L1 - only when
    // the numbers agree; a partner that is not there yet when asked, or anywhere else, means agent scope (correct everywhere).
    int my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc = (my_xcc & 15) + 1;
    int *const xcc_slot = reinterpret_cast<int *>(pmem + 4 * C::XROW_FLOATS) + 32;   // [band 2], behind the sequence numbers ([band][16])
    if (tid == 0) __hip_atomic_store(xcc_slot + band, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int l2x = 0;
    const float sgn = wave == 1 ? 1.f : -1.f;
    // taps along y in local order d = -1, 0, +1 (input row l + d): band 1 runs down the board (ky = 1 + d), band 0 up (ky = 1 - d)
    const int ky_m = band == 1 ? 0 : 2, ky_p = band == 1 ? 2 : 0;
    // this wave's weight fragments of a layer: slot 0 = tap 0, slot 2 = tap +1, slots 1 / 3 = tap -1 of even / odd layers
    i32x4v ua[4][2][2][4];
    {
        const int wlane0 = (tid & 63) * 16;
        const unsigned char *w0 = net.w1_w + (size_t)wave * 49152;
        w1_request_tap<1>(ua, w0 + ky_m * 16384, wlane0);
        static_for<16>([&](auto I_) { w1_request<0>(ua, w0 + 16384, wlane0, std::integral_constant<int, kWbTap0Order[decltype(I_)::value]>{}); });
        // (tap +1: the layer's own stages S and 2A ask for it)
    }

    int kiter = 0;
    for (int b = pair; b < batch; b += n_pairs, ++kiter) {
        stamp(0);
        // ================= stem: planes -> im2col'ed f16-pair images of the band's cells (halo row included) -> X =================
        i32x4v fa[2][2][4];
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
                if (stid + i * NTHR < 6 * P) st[stid + i * NTHR] = pre[i];
            for (int e = stid; e < 4 * 64; e += NTHR)
                reinterpret_cast<unsigned *>(smem + C::SI_OFF + (e >> 6) * IMG + C::ZOFF)[e & 63] = 0u;
            if (stid < 128) reinterpret_cast<float *>(smem + C::SS_OFF)[stid] = ssv;
            __syncthreads();
            if (stid < C::CELLS) {
                const int row = stid, lr = row / S, x = row - lr * S, l = lr - 1;
                const int y = band == 0 ? 9 - l : 10 + l;
                const bool cell_ok = y >= 0 && y < S;
                const float *src = st + y * S + x;
                const int swz = (row >> 1) & 3;
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {
                    f32x4 lo, hi;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int k = sl * 8 + j, t = k / 6, c = k - t * 6;
                        const int dy = t / 3 - 1, dx = t % 3 - 1;
                        const bool ok = cell_ok && k < 54 && (unsigned)(y + dy) < (unsigned)S && (unsigned)(x + dx) < (unsigned)S;
                        const float v = ok ? src[c * P + dy * S + dx] : 0.f;
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
        float amax = 0.f;
        {
            const int slane = fresh_lane();
            const int sli = slane & 15, slg = slane >> 4;
#pragma unroll
            for (int r = 0; r < RTW; ++r) {
                int row = (wave * RTW + r) * 16 + sli;
                asm volatile("" : "+v"(row));
                const int lr = row / S, x = row - lr * S, l = lr - 1;
                const bool ok = row < C::CELLS && l <= lmax;       // (band 1: its row l = 9 lies beyond the board)
                const int nat = row * 64 + ((slg ^ ((row >> 1) & 3)) << 4);
                const int addr = C::SI_OFF + (row < C::CELLS ? nat : C::ZOFF + (nat & 255));
                i32x4v fb[2][2];
                lds_load_frag<0 * IMG>(fb[0][0], smem, addr);
                lds_load_frag<1 * IMG>(fb[0][1], smem, addr);
                lds_load_frag<2 * IMG>(fb[1][0], smem, addr);
                lds_load_frag<3 * IMG>(fb[1][1], smem, addr);
                const int osw = wb_swz(l, x);
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
                    if (ok) amax = fmaxf(fmaxf(amax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                    *reinterpret_cast<f32x4 *>(smem + C::X_OFF + (ok ? row * 256 : C::DUMP_REL) + (((c * 4 + slg) ^ osw) << 4)) = v;
                }
            }
        }
        __syncthreads();                                    // X complete; the overlay is free again
        if (wave == 0) {                                    // H's dump / zero rows were under it
            reinterpret_cast<float *>(smem + C::H_OFF + C::DUMP_REL)[fresh_lane()] = 0.f;
            reinterpret_cast<float *>(smem + C::H_OFF + C::ZERO_REL)[fresh_lane()] = 0.f;
        }
        if (!(amax < (float)kWsRangeLimit)) ovf |= 1;
        if (kiter == 0) {
            // (both bands ask at the same point of their first board and both have published before their stem: they read the same two
            // numbers unless one of them started more than a stem late - then BOTH must fall back, so the late one's own answer does not
            // count either: a band takes the L2 path only if the partner's number was there within the bounded look AND the partner
            // says the same about ours - it publishes its verdict, we read it before the first row travels (layer 1))
            int theirs = 0;
            for (int spin = 0; spin < 4096 && theirs == 0; ++spin) {
                theirs = __hip_atomic_load(xcc_slot + (1 - band), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (theirs == 0) __builtin_amdgcn_s_sleep(2);
            }
            theirs = __builtin_amdgcn_readfirstlane(theirs);
            l2x = theirs == my_xcc ? 1 : 0;
            if (tid == 0) __hip_atomic_store(xcc_slot + 2 + band, l2x ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        stamp(1);

        // ================= tower =================
        {
            const int glane = fresh_lane(), gli = glane & 15, glg = glane >> 4, wlane = glane * 16;
            const int uc = gli / 5, ut = gli - 5 * uc;             // row class, tile inside the half (gli 15: no unit)
            const bool uv = gli < 15;
            const int xa0 = wave == 0 ? 2 * ut - 1 : (wave == 2 ? 2 * ut + 1 : 2 * ut);
            const int xb0 = wave == 0 ? 2 * ut + 1 : (wave == 1 ? 2 * ut + 1 : (wave == 2 ? 2 * ut : 2 * ut + 2));
            // Per-lane addressing with few registers (the kernel is written against the whole register file): the swizzle of a
            // cell a lane touches is always that of class g = (gli + K) mod 8 with a compile-time K (wb_swz: g = 5 floor(l / 3) +
            // (x + 1) / 2, l = 3 c + s, x = 2 (t' + 5 half) + e) - `pk` holds the eight 4-bit XOR values of g = gli .. gli + 7.
            unsigned pk = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int g = (gli + j) & 7;
                pk |= (unsigned)((g & 1) | ((g & 6) << 1)) << (4 * j);
            }
            asm volatile("" : "+v"(pk));
            auto swz_k = [&](int K) { return (int)((pk >> (4 * (K & 7))) & 15u); };   // K: compile-time at every use
            // V-row reads of the regular stages: UNswizzled address of cell (3 c, xa / xb + 10 half), chunk 2 glg - or of the zero row
            // with stride 0 (a column outside the board, lane 15); row s adds s * stride, the swizzle K = 5 fd(s) + 5 half + (e + 1) / 2
            int uA[2], uB[2], stA[2], stB[2];
            const int eA = wave == 0 ? -1 : (wave == 2 ? 1 : 0), eB = wave == 0 ? 1 : (wave == 1 ? 1 : (wave == 2 ? 0 : 2));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int xa = xa0 + 10 * h, xb = xb0 + 10 * h;
                const bool oka = uv && xa >= 0 && xa < S, okb = uv && xb >= 0 && xb < S;
                uA[h] = (oka ? ((3 * uc + 1) * S + xa) * 256 : C::ZERO_REL) + (glg << 5);
                uB[h] = (okb ? ((3 * uc + 1) * S + xb) * 256 : C::ZERO_REL) + (glg << 5);
                stA[h] = oka ? ROWB : 0;
                stB[h] = okb ? ROWB : 0;
            }
            // outputs of a regular stage: cells (3 c + k, 2 t + e), channels 16 wave + 4 glg .. (swizzled: the class does not depend on
            // k); outside the board: dump row, stride 0.  Residual reads: the same cell, or the zero row (= dump row + 256).
            int oS[2][2], oSt[2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int x = 2 * (ut + 5 * h) + e;
                    const bool ok = uv && x < S;
                    oS[h][e] = (ok ? ((3 * uc + 1) * S + x) * 256 : C::DUMP_REL) + (((wave * 4 + glg) ^ swz_k(5 * h + ((e + 1) >> 1))) << 4);
                }
            oSt[0] = uv ? ROWB : 0;                                // (half 0, and e = 0 of half 1)
            oSt[1] = (uv && 2 * (ut + 5) + 1 < S) ? ROWB : 0;      // (half 1, e = 1: x = 19 for the last tile)
            // row 9's stage (li = tile, 10 .. 15: no unit): addresses made where they are used (three times a layer)
            const bool sv = gli < 10;
            // edge row l = 0 as the PARTNER's halo row l = -1 (its LDS layout): bytes into an exchange row, or -1
            // (class 0 lanes: gli = t', so the halo row's class is g = gli + K, K = -5 + 5 h + (e + 1) / 2; through pk: not hoisted)
            auto x_off = [&](int h, int e) {
                const int x = 2 * (ut + 5 * h) + e;
                return (uv && uc == 0 && x < S) ? x * 256 + (((wave * 4 + glg) ^ swz_k(11 + 5 * h + ((e + 1) >> 1))) << 4) : -1;
            };

            f32x4 dq[2][2];                                        // cell reads in flight: [(kc, channel half) group parity][cell a / b]
            i32x4v vh[5][2], vl[5][2];                             // [slot (wb_slot)][kc]
            f32x4 acc[1][4];                                       // (one set: slice i stores the previous stage's acc[i] to the exchange before its MFMA restarts it)
            f32x4 ez[4], eres[2], ev[2];
            float tvv[4];
            unsigned thh[2];
            i32x4v flag_seen = i32x4v{0, 0, 0, 0};
            // ---- cell reads and transforms of a job ----
            // address of cell column a / b (CB) of a job's row, chunk 2 glg (the k-chunk / channel-half bits are XORed in by rd)
            auto job_addr = [&](auto IN_, auto OUT_, auto JOB_, auto CB_) __attribute__((always_inline)) {
                constexpr int job = decltype(JOB_)::value, j = job % 1000, cb = decltype(CB_)::value;
                constexpr int base = job >= 1000 ? decltype(OUT_)::value : decltype(IN_)::value;
                if constexpr (j >= 200) {
                    // row 9's mapping: cell (r, 2 li + e), class g = 5 floor(r / 3) + li + (e + 1) / 2
                    constexpr int r = j - 200;
                    const int e = cb ? eB : eA, x = 2 * gli + e;
                    const bool ok = sv && x >= 0 && x < S && r <= lmax;
                    const int K = 5 * (r / 3) + ((e + 1) >> 1);        // (e: wave-uniform, so is the shift below)
                    return base + (ok ? ((r + 1) * S + x) * 256 : C::ZERO_REL) + (((glg * 2) ^ (int)((pk >> (4 * (K & 7))) & 15u)) << 4);
                } else {
                    constexpr int sr = (j - 100) / 10 - 1, h = (j - 100) % 10;
                    constexpr int fd = sr < 0 ? -1 : sr / 3;
                    const int e = cb ? eB : eA;
                    const int K = 5 * fd + 5 * h + ((e + 1) >> 1) + 16;
                    int a = (cb ? uB[h] : uA[h]) + sr * (cb ? stB[h] : stA[h]);
                    if constexpr (sr == 3) {
                        // row 3 c + 3 of class 2 is row 9: beyond the board for band 1
                        if (band == 1 && uc == 2) a = C::ZERO_REL + (glg << 5);
                    }
                    return base + (a ^ ((int)((pk >> (4 * (K & 7))) & 15u) << 4));
                }
            };
            // read I (0 .. 7) of a job: group (kc, channel half) = I >> 1 in the order (0, 0), (0, 1), (1, 0), (1, 1), cell a / b = I & 1
            auto rd = [&](auto IN_, auto OUT_, auto JOB_, auto I_) __attribute__((always_inline)) {
                constexpr int i = decltype(I_)::value, cb = i & 1, g = i >> 1, kc = g >> 1, hh = g & 1;
                if constexpr (WB_ABL & 4) return;
                const int a = job_addr(IN_, OUT_, JOB_, std::integral_constant<int, cb>{});
                dq[g & 1][cb] = lds_f32x4_at<0>(a ^ ((kc << 7) | (hh << 4)));
            };
            auto tr = [&](auto JOB_, auto I_) __attribute__((always_inline)) {
                constexpr int job = decltype(JOB_)::value, j = job % 1000, i = decltype(I_)::value, g = i >> 2, kc = g >> 1, hh = g & 1, q = i & 3;
                constexpr int sl = wb_slot(j);
                if constexpr (WB_ABL & 2) return;
                if constexpr (q == 0) {
                    tvv[0] = fmaf(dq[g & 1][1][0], sgn, dq[g & 1][0][0]);
                    tvv[1] = fmaf(dq[g & 1][1][1], sgn, dq[g & 1][0][1]);
                } else if constexpr (q == 1) {
                    tvv[2] = fmaf(dq[g & 1][1][2], sgn, dq[g & 1][0][2]);
                    tvv[3] = fmaf(dq[g & 1][1][3], sgn, dq[g & 1][0][3]);
                } else if constexpr (q == 2) {
                    thh[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{tvv[0], tvv[1]}, f16x2));
                    vh[sl][kc][2 * hh] = (int)thh[0];
                    vl[sl][kc][2 * hh] = (int)low_pieces(tvv[0], tvv[1], thh[0]);
                } else {
                    thh[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{tvv[2], tvv[3]}, f16x2));
                    vh[sl][kc][2 * hh + 1] = (int)thh[1];
                    vl[sl][kc][2 * hh + 1] = (int)low_pieces(tvv[2], tvv[3], thh[1]);
                }
            };
            // A job in the slices of a stage, T0 = its first slice (early 0, behind-the-barrier 24, late 48): reads of group g at
            // T0 + {0, 2, 8, 12} (+ 1 for cell b), transform sub-steps of group g at T0 + {6, 10, 14, 18} + 0 .. 3 - a group's two
            // registers are read again only when the group two before it has been consumed
            auto job_slice = [&](auto IN_, auto OUT_, auto JOB_, auto T0_, auto M_) __attribute__((always_inline)) {
                constexpr int t = decltype(M_)::value - decltype(T0_)::value;
                if constexpr (decltype(JOB_)::value != 0 && t >= 0 && t < 22) {
                    if constexpr (t == 0 || t == 1) rd(IN_, OUT_, JOB_, std::integral_constant<int, t>{});
                    if constexpr (t == 2 || t == 3) rd(IN_, OUT_, JOB_, std::integral_constant<int, t>{});
                    if constexpr (t == 8 || t == 9) rd(IN_, OUT_, JOB_, std::integral_constant<int, t - 4>{});
                    if constexpr (t == 12 || t == 13) rd(IN_, OUT_, JOB_, std::integral_constant<int, t - 6>{});
                    if constexpr (t >= 6) tr(JOB_, std::integral_constant<int, t - 6>{});
                }
            };
            // a whole job at once (a group's first layer: nothing was prepared under a previous one)
            auto job_now = [&](auto IN_, auto OUT_, auto JOB_) __attribute__((always_inline)) {
                static_for<22>([&](auto M_) { job_slice(IN_, OUT_, JOB_, std::integral_constant<int, 0>{}, M_); });
            };

            // One layer.  PAR = layer parity (conv2 of a block = odd = the residual); layer instantiations alternate, so every
            // register choice that depends on it is a compile-time one.
            auto layer_fn = [&](auto IN_, auto OUT_, auto RES_, int layer) __attribute__((always_inline)) {
                constexpr int IN = decltype(IN_)::value, OUT = decltype(OUT_)::value;
                constexpr bool RES = decltype(RES_)::value;
                constexpr int PAR = RES ? 1 : 0;
                constexpr int S1 = PAR ? 3 : 1, S1N = PAR ? 1 : 3;
                const int next_layer = layer + 1 < kTowerLayers ? layer + 1 : 0;
                const unsigned char *wnext = net.w1_w + ((size_t)next_layer * 4 + wave) * 49152;
                const unsigned char *wcur = net.w1_w + ((size_t)layer * 4 + wave) * 49152;
                int exw = C::EX_OFF + wave * 4096 + glane * 16, exr = C::EX_OFF + wave * 1024 + glane * 16;
                asm volatile("" : "+v"(exw), "+v"(exr));
                // (hipcc hoists what is invariant across the block loop - some fifty cell addresses - out of it and spills it: the
                // values they derive from are made opaque per layer, the addresses cheap recomputations)
                asm volatile("" : "+v"(pk), "+v"(uA[0]), "+v"(uA[1]), "+v"(uB[0]), "+v"(uB[1]));
                asm volatile("" : "+v"(oS[0][0]), "+v"(oS[0][1]), "+v"(oS[1][0]), "+v"(oS[1][1]));
                // sequence number of the edge row the PREVIOUS layer published (this one consumes it), and its buffer parity
                const int pub = kiter * 11 + layer;                // number of edge rows published before this layer's
                // this layer's folded shift (channels 16 wave + 4 glg ..) and 2^-e, the previous layer's for the epilogue stage S carries
                const int prev_layer = layer > 0 ? layer - 1 : 0;
                const f32x4 shf = *reinterpret_cast<const f32x4 *>(smem + C::SH_OFF + (layer * 64 + wave * 16 + glg * 4) * 4);
                const float down = reinterpret_cast<const float *>(smem + C::SH_OFF)[12 * 64 + layer];
                const float pdown = layer > 0 ? reinterpret_cast<const float *>(smem + C::SH_OFF)[12 * 64 + prev_layer] : 0.f;
                f32x4 pshf;
                if (layer == 0) {
                    // a group's first layer: the row-9 stage's V rows, which the previous layer's 0A / 0B would have made
                    job_now(IN_, OUT_, std::integral_constant<int, 200 + 8>{});
                    job_now(IN_, OUT_, std::integral_constant<int, 200 + 9>{});
                }
                if (layer == 1 && kiter == 0) {
                    // the first edge row is about to travel: the partner's verdict on the L2 path (published behind its first stem, a
                    // layer ago).  Both verdicts must be "yes"; one that does not come at all is a partner that is not there - the
                    // bounded wait of the hand-off would find that out a few stages later anyway.
                    int theirs = 0, spins = 0;
                    while (theirs == 0) {
                        theirs = __hip_atomic_load(xcc_slot + 2 + (1 - band), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (theirs == 0) {
                            __builtin_amdgcn_s_sleep(2);
                            if (++spins > kWbSpinLimit) {
                                *reinterpret_cast<volatile int *>(dead) = 1;
                                break;
                            }
                        }
                    }
                    if (__builtin_amdgcn_readfirstlane(theirs) != 2) l2x = 0;
                }
                // tap -1 of this layer has arrived (and with it everything requested before 0B: the next shift / scale, the halo copy)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWbWaitTop) : "memory");
                __builtin_amdgcn_sched_barrier(0);

                // exchange + epilogue of the PREVIOUS stage, sub-step I (the slices of the stage they ride in)
                auto epi = [&](auto N_, auto I_) __attribute__((always_inline)) {
                    constexpr int n = decltype(N_)::value, i = decltype(I_)::value;
                    constexpr bool PREV = n == 0;                  // stage S carries the previous layer's 0B
                    constexpr int pn = PREV ? 6 : n - 1;
                    constexpr WbStage ps = wb_stage(pn);
                    constexpr int par = (PREV ? (1 - PAR) + 6 : PAR + pn) & 1;
                    constexpr int OB = PREV ? IN : OUT;
                    constexpr bool RS = PREV ? !RES : RES;
                    const bool null_epi = PREV && layer == 0;
                    constexpr int pk = ps.k, ph = ps.h;
                    // (output cells of the stage the epilogue belongs to: row 9's mapping, or cells (3 c + k, 2 t + e) of half ph)
                    // (output cell e of the stage the epilogue belongs to: row 9's mapping - cell (9, 2 li + e) - or cell (3 c + k, 2 t + e) of
                    // half ph; the residual comes from the same cell, or from the zero row = dump row + 256)
                    auto out_addr = [&](int e) {
                        if constexpr (pk == 3) {
                            const int x = 2 * gli + e;
                            const bool ok = sv && x < S && lmax == 9;
                            return (ok ? (10 * S + x) * 256 : C::DUMP_REL) + (((wave * 4 + glg) ^ swz_k(15 + ((e + 1) >> 1))) << 4);
                        } else {
                            return oS[ph & 1][e] + (pk % 3) * ((ph == 1 && e == 1) ? oSt[1] : oSt[0]);
                        }
                    };
#define WB_OADDR(e, store) ((store) ? out_addr(e) : (out_addr(e) >= C::DUMP_REL ? out_addr(e) + 256 : out_addr(e)))
                    if constexpr (i < 4) {
                        if constexpr (!(WB_ABL & 16)) lds_f32x4_put<par * 16384 + i * 1024>(exw, acc[0][i]);
                    } else if constexpr (i == 10) {
                        if constexpr (!(WB_ABL & 32)) __syncthreads();
                    } else if constexpr (i == 11 || i == 12) {
                        if constexpr (!(WB_ABL & 16)) {
                        ez[2 * (i - 11)] = lds_f32x4_at<par * 16384 + (2 * (i - 11)) * 4096>(exr);
                        ez[2 * (i - 11) + 1] = lds_f32x4_at<par * 16384 + (2 * (i - 11) + 1) * 4096>(exr);
                        }
                    } else if constexpr (i == 13) {
                        if constexpr (PREV) {
                            pshf = *reinterpret_cast<const f32x4 *>(smem + C::SH_OFF + (prev_layer * 64 + wave * 16 + glg * 4) * 4);
                            if (layer == 0) pshf = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        if constexpr (RS) {
                            const int zr = C::ZERO_REL + (glane * 16) % 256;
                            eres[0] = lds_f32x4_at<OB>(null_epi ? zr : WB_OADDR(0, false));
                            eres[1] = lds_f32x4_at<OB>(null_epi ? zr : WB_OADDR(1, false));
                        }
                    } else if constexpr (i >= 19 && i < 35) {
                        constexpr int kk = i - 19, cc = kk >> 3, e = (kk >> 1) & 3, part = kk & 1;
                        if constexpr (WB_ABL & 8) {
                        } else if constexpr (part == 0) {
                            ev[cc][e] = cc == 0 ? (ez[0][e] + ez[1][e]) + ez[2][e] : (ez[1][e] - ez[2][e]) - ez[3][e];
                        } else {
                            float tt = fmaf(ev[cc][e], PREV ? pdown : down, PREV ? pshf[e] : shf[e]);   // (pshf: read where stage S's epilogue starts)
                            if constexpr (RS) tt += eres[cc][e];
                            ev[cc][e] = fmaxf(tt, 0.f);
                        }
                    } else if constexpr (i == 35 || i == 36) {
                        constexpr int e = i - 35;
                        amax = fmaxf(fmaxf(amax, ev[e][0]), ev[e][1]);
                        amax = fmaxf(fmaxf(amax, ev[e][2]), ev[e][3]);
                        lds_f32x4_put<OB>(null_epi ? C::DUMP_REL + (glane * 16) % 256 : WB_OADDR(e, true), ev[e]);
                    }
#undef WB_OADDR
                };

                static_for<7>([&](auto N_) {
                    constexpr int n = decltype(N_)::value;
                    constexpr WbStage st = wb_stage(n);
                    using JE = std::integral_constant<int, st.early>;
                    using JA = std::integral_constant<int, st.after>;
                    using JL = std::integral_constant<int, st.late>;
                    // ---- before the stage ----
#ifdef WB_PROF_FIRST
                    if constexpr (PROF) { if (layer < 2) stamp(16 + 8 * PAR + n); }   // (experiments: the stages of a board's FIRST two layers)
#else
                    if constexpr (PROF) stamp(16 + 8 * PAR + n);       // (every layer overwrites: the last conv1 / conv2 layers' stay)
#endif
                    static_for<72>([&](auto M_) {
                        constexpr int m = decltype(M_)::value;
                        constexpr int NT = st.k == 3 ? 2 : 3;
                        if constexpr (PROF && PAR == 1 && n <= 1 && (m == 0 || m == 11 || m == 24 || m == 37 || m == 44 || m == 48 || m == 60 || m == 71)) {
                            constexpr int si = m == 0 ? 0 : (m == 11 ? 1 : (m == 24 ? 2 : (m == 37 ? 3 : (m == 44 ? 4 : (m == 48 ? 5 : (m == 60 ? 6 : 7))))));
                            stamp(32 + 8 * n + si);
                        }
                        if constexpr (m < 4) epi(N_, M_);           // (exchange write of acc[m]: in front of the MFMA that restarts it)
                        if constexpr (m < 24 * NT) {
                            constexpr int ti = m / 24, q = m % 24, kc = q / 12, pr = (q / 4) % 3, c = q % 4;
                            // row 9's stage: taps -1 (row 8: half A's slot 1), 0 (row 9: half B's slot 1); regular: ord 0 = (-1, 0, +1), 1 = (+1, 0, -1)
                            constexpr int d = st.k == 3 ? ti - 1 : (st.ord == 0 ? ti - 1 : (st.ord == 1 ? 1 - ti : (ti == 0 ? 0 : (ti == 1 ? 1 : -1))));
                            constexpr int slot = d == -1 ? S1 : (d == 0 ? 0 : 2);
                            constexpr int vsl = st.k == 3 ? ti : wb_slot(100 + 10 * (st.k + d + 1) + st.h);
                            // waits for the fragments (see wb_wreq)
                            if constexpr (n == 0 && m == 24) {
                                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWbWaitTap0) : "memory");
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if constexpr (n == 1 && m == 48) {
                                static_assert(kWbWaitTapP == 0, "tap +1 wait");
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tap +1; also: the edge stores of stage S (slices 39-42) have landed
                                __builtin_amdgcn_sched_barrier(0);
                                // ... so this wave's part of the previous layer's edge row is in memory: say so (the partner waits for all four waves)
                                if (layer > 0 && glane == 0 && !mute) __hip_atomic_store(seq_mine + wave, pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            // (Band 1 has no row 9: its stage S multiplies the zero row - the jobs of rows beyond lmax read it - and stores to
                            // the dump row.  A uniform `if (band == 0)` around these 48 MFMAs cost a compare and a branch per MFMA in BOTH
                            // bands, and band 1 gained nothing from skipping them: it waits for band 0's edge row a few stages later anyway.)
                            if constexpr (!(WB_ABL & 128)) {
                                if constexpr (pr == 0)
                                    acc[0][c] = mfma16<F>(ua[slot][kc][1][c], vh[vsl][kc], q < 4 && ti == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[0][c]);
                                else if constexpr (pr == 1) acc[0][c] = mfma16<F>(ua[slot][kc][0][c], vl[vsl][kc], acc[0][c]);
                                else acc[0][c] = mfma16<F>(ua[slot][kc][0][c], vh[vsl][kc], acc[0][c]);
                            }
                        }
                        // ---- what rides along ----
                        if constexpr (m >= 4 && m < 37) epi(N_, M_);
                        if constexpr (n == 0 && m >= 39 && m < 43) {
                            // The previous layer's edge row l = 0 (layers 0 .. 10) goes to the partner, in ITS halo row's layout: half A's cells were
                            // stored to LDS by this lane under 0B (read back: its own writes), half B's are the epilogue values of this stage.  HERE
                            // because a store's acknowledgement takes ~1 us and every counted wait behind it waits for it: the next one is 2A's slice 48.
                            constexpr int hx = (m - 39) >> 1, e = (m - 39) & 1;
                            const int xo = x_off(hx, e);
                            if (layer > 0 && xo >= 0 && !(WB_ABL & 64)) {
                                f32x4 v = ev[e];
                                if constexpr (hx == 0) v = lds_f32x4_at<IN>(oS[0][e]);
                                float *dst = pmem + (size_t)(band * 2 + ((pub - 1) & 1)) * C::XROW_FLOATS;
                                const float *pp = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(dst) + xo);
                                // (partner on this XCD: a plain store - the vector L1 writes through to the shared L2 and the row stays there;
                                // otherwise agent scope: written through to the fabric)
                                if (l2x) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(pp), "v"(v) : "memory");
                                else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(pp), "v"(v) : "memory");
                            }
                        }
                        job_slice(IN_, OUT_, JE{}, std::integral_constant<int, 0>{}, M_);
                        job_slice(IN_, OUT_, JA{}, std::integral_constant<int, 24>{}, M_);
                        job_slice(IN_, OUT_, JL{}, std::integral_constant<int, 48>{}, M_);
                        // weight requests
                        {
                            constexpr int code = wb_wreq(n, m), kind = code >> 4;
                            using FR = std::integral_constant<int, (code & 15)>;
                            if constexpr (code >= 0 && !(WB_ABL & 1)) {
                                if constexpr (kind == 0) w1_request<S1N>(ua, wnext + ky_m * 16384, wlane, FR{});
                                else if constexpr (kind == 1) w1_request<0>(ua, wnext + 16384, wlane, FR{});
                                else w1_request<2>(ua, wcur + ky_p * 16384, wlane, FR{});
                            }
                        }
                        // hand-off riders
                        if constexpr (n == 3 && m == 40) {
                            // 1A: ask for the partner's four sequence numbers (looked at under 1B)
                            if (layer > 0 && !(WB_ABL & 64)) {
                                int zoff = 0;
                                asm volatile("" : "+v"(zoff));
                                asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(flag_seen) : "v"(zoff), "s"(seq_theirs) : "memory");
                            }
                        }
                        if constexpr (n == 4 && m == 16) {
                            // 1B: the partner's edge row of the previous layer must have been published by all of its waves ...
                            if (layer > 0 && !(WB_ABL & 64)) {
                                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // (1A's four weight requests lie behind the question)
                                __builtin_amdgcn_sched_barrier(0);
                                int seen = min(min(flag_seen[0], flag_seen[1]), min(flag_seen[2], flag_seen[3]));
                                seen = __builtin_amdgcn_readfirstlane(seen);
                                int spins = 0;
                                while (seen < pub) {
                                    if (*reinterpret_cast<volatile int *>(dead)) break;
                                    __builtin_amdgcn_s_sleep(2);
                                    seen = __hip_atomic_load(seq_theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                                    for (int w = 1; w < 4; ++w) seen = min(seen, __hip_atomic_load(seq_theirs + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                                    if (++spins > kWbSpinLimit) {
                                        *reinterpret_cast<volatile int *>(dead) = 1;
                                        if (net.band_timeouts && glane == 0 && wave == 0) atomicAdd(net.band_timeouts, 1u);
                                        break;
                                    }
                                }
                                // ... then it is copied into the halo row of this layer's INPUT buffer: 19 x 256 B = 4 x 1 KB + 768 B, LDS-DMA, sc1
                                const float *src = pmem + (size_t)((1 - band) * 2 + ((pub - 1) & 1)) * C::XROW_FLOATS;
                                int dl = glane * 16;
                                asm volatile("" : "+v"(dl));                      // (per layer: a 64-bit per-lane pointer kept across the block loop is two spilled registers)
                                // (aux 16 = sc1: agent scope; aux 2 = nt: past this CU's vector L1, from the XCD's L2 - tools/microbench/xwg_pingpong.hip variant 1)
                                if (l2x) {
                                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(src) + wave * 1024 + dl),
                                                                     (__attribute__((address_space(3))) void *)(smem + IN + wave * 1024), 16, 0, 2);
                                    if (wave == 0 && dl < 48 * 16)
                                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(src) + 4096 + dl),
                                                                         (__attribute__((address_space(3))) void *)(smem + IN + 4096), 16, 0, 2);
                                } else {
                                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(src) + wave * 1024 + dl),
                                                                     (__attribute__((address_space(3))) void *)(smem + IN + wave * 1024), 16, 0, 16);
                                    if (wave == 0 && dl < 48 * 16)
                                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(src) + 4096 + dl),
                                                                         (__attribute__((address_space(3))) void *)(smem + IN + 4096), 16, 0, 16);
                                }
                            }
                        }
                        if constexpr (n == 5 && m == 9) {
                            // 0A, in front of its barrier: this wave's share of the halo copy has landed (behind it: 1B's four weight requests)
                            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
                if (!(amax < (float)kWsRangeLimit)) ovf |= 1;
                if constexpr (PROF) {
                    stamp(2 + layer);
#ifdef WB_PROF_FIRST
                    if (layer < 2) stamp(16 + 8 * PAR + 7);
#else
                    stamp(16 + 8 * PAR + 7);
#endif
                }
            };
            using IX = std::integral_constant<int, C::X_OFF>;
            using IH = std::integral_constant<int, C::H_OFF>;
            {
                float z0, z1, z2, z3;
                asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z0), "=v"(z1), "=v"(z2), "=v"(z3));
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[0][c] = f32x4{z0, z1, z2, z3};      // (layer 0's stage S carries a null epilogue)
            }
#pragma unroll 1
            for (int blk = 0; blk < kBlocks; ++blk) {
                layer_fn(IX{}, IH{}, std::false_type{}, 2 * blk);
                layer_fn(IH{}, IX{}, std::true_type{}, 2 * blk + 1);
            }
            {
                // the tower's last stage (layer 11's 0B: output X, residual) on its own
                const int exw = C::EX_OFF + wave * 4096 + glane * 16, exr = C::EX_OFF + wave * 1024 + glane * 16;
                constexpr int par = (1 + 6) & 1;
                static_for<4>([&](auto I_) { lds_f32x4_put<par * 16384 + decltype(I_)::value * 1024>(exw, acc[0][decltype(I_)::value]); });
                const f32x4 pshf = *reinterpret_cast<const f32x4 *>(smem + C::SH_OFF + (11 * 64 + wave * 16 + glg * 4) * 4);
                const float pdown = reinterpret_cast<const float *>(smem + C::SH_OFF)[12 * 64 + 11];
                const int o0 = oS[1][0], o1 = oS[1][1];
                eres[0] = lds_f32x4_at<C::X_OFF>(o0 >= C::DUMP_REL ? o0 + 256 : o0);
                eres[1] = lds_f32x4_at<C::X_OFF>(o1 >= C::DUMP_REL ? o1 + 256 : o1);
                __syncthreads();
                static_for<4>([&](auto I_) { ez[decltype(I_)::value] = lds_f32x4_at<par * 16384 + decltype(I_)::value * 4096>(exr); });
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o0 = (ez[0][e] + ez[1][e]) + ez[2][e], o1 = (ez[1][e] - ez[2][e]) - ez[3][e];
                    ev[0][e] = fmaxf(fmaf(o0, pdown, pshf[e]) + eres[0][e], 0.f);
                    ev[1][e] = fmaxf(fmaf(o1, pdown, pshf[e]) + eres[1][e], 0.f);
                    amax = fmaxf(fmaxf(amax, ev[0][e]), ev[1][e]);
                }
                lds_f32x4_put<C::X_OFF>(o0, ev[0]);
                lds_f32x4_put<C::X_OFF>(o1, ev[1]);
                if (!(amax < (float)kWsRangeLimit)) ovf |= 1;
                __syncthreads();
            }
        }
        if (*reinterpret_cast<volatile int *>(dead)) ovf |= 2;   // (a partner that did not show up: the exact kernel redoes the whole batch)
        // a board that left the f16 range says so: the exact kernel behind this launch redoes the marked boards only
        if (group_bits && !(amax < (float)kWsRangeLimit)) atomicOr(group_bits + (b >> 5), 1 << (b & 31));
        // next board's planes: requested here, consumed after the head convolutions
        fetch_planes(b + n_pairs);
        // ================= heads, first part: the three 1x1 convolutions of the band's cells (fp32), batch norm, ReLU -> feat =================
        {
            const int ht = wave * 64 + fresh_lane();
            const float *hw = reinterpret_cast<const float *>(smem + C::HW_OFF);
            const float *hs = reinterpret_cast<const float *>(smem + C::HS_OFF);
            const int ncell = (lmax + 1) * S;
            if (ht < ncell) {
                const int l = ht / S, x = ht - l * S, y = band == 0 ? 9 - l : 10 + l;
                const int sw = wb_swz(l, x), base = C::X_OFF + ((l + 1) * S + x) * 256;
                float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
                for (int k4 = 0; k4 < 16; ++k4) {
                    const f32x4 xv = *reinterpret_cast<const f32x4 *>(smem + base + ((k4 ^ sw) << 4));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 w = *reinterpret_cast<const f32x4 *>(hw + (k4 * 4 + j) * 4);
                        d0 = fmaf(xv[j], w[0], d0);
                        d1 = fmaf(xv[j], w[1], d1);
                        d2 = fmaf(xv[j], w[2], d2);
                    }
                }
                float *f = feat + (size_t)b * 3 * P + y * S + x;
                f[0] = fmaxf(fmaf(d0, hs[0], hs[1]), 0.f);
                f[P] = fmaxf(fmaf(d1, hs[2], hs[3]), 0.f);
                f[2 * P] = fmaxf(fmaf(d2, hs[4], hs[5]), 0.f);
            }
        }
        __syncthreads();
        stamp(14);
        if constexpr (PROF) {
            __syncthreads();
            // (the SECOND board of pair 0 when there is one: steady state, the partner is neither ahead nor behind by a launch skew)
            if (pair == 0 && kiter == (batch > n_pairs ? 1 : 0) && tid < 64)
                net.timeline[band * 64 + tid] = reinterpret_cast<volatile long long *>(smem + C::PROF_OFF)[tid];
        }
    }
    if (ovf && overflow) atomicOr(overflow, ovf);
}

// The two fully connected layers + softmaxes of a 19x19 launch, on the features dualnet_fwd_w1dband_kernel left in global memory
// ([board][policy 0 | policy 1 | value][361]): policy FC 722 -> 362, value FC 361 -> 3.  Plain fp32 FMAs (the reference's arithmetic) in
// ONE summation order whatever the launch size: the 722 inputs in four quarters (181, 181, 180, 180), each a sequential FMA chain
// from zero, logit = ((q0 + q1) + (q2 + q3)) + bias.  The FC matrix (transposed, [722][362] fp32, 1 MB) is the cost: a CU's L1
// passes it in 7.5 us, so
//   * small launches (one tree's mini-batch): dualnet_heads19_part_kernel - one workgroup per (board, quarter), partial sums to
//     global memory - then dualnet_heads19_fin_kernel per board (sum, bias, softmaxes, value FC);
//   * large launches: dualnet_heads19_kernel<16> - sixteen boards per workgroup share one pass over the matrix.
// The band kernel's sequence numbers ([pair][band 2][16] ints behind each pair's exchange rows) must read zero when a launch
// starts: the heads kernel queued behind a band launch - the band kernel is done with them by then - clears them for the stream's
// NEXT launch (until round 6 a memset node over the whole 2.5 MB exchange area in front of every launch: 5 us + its gap, a
// twentieth of a 64-board launch).
__device__ __forceinline__ void wb_clear_seq(int *seq, int n_pairs, int block, int blocks, int tid) {
    if (seq == nullptr || tid >= 32) return;
    for (int p = block; p < n_pairs; p += blocks) seq[(size_t)p * WbCfg::PAIR_FLOATS + tid] = 0;
}

__global__ __launch_bounds__(384) void dualnet_heads19_part_kernel(NetDev net, const float *__restrict__ feat, float *__restrict__ part) {
    constexpr int P = 361, A = 362;
    __shared__ float f[184];
    const int b = blockIdx.x >> 2, q = blockIdx.x & 3, tid = threadIdx.x;
    const int j0 = q < 2 ? q * 181 : 362 + (q - 2) * 180, n = q < 2 ? 181 : 180;
    if (tid < n) f[tid] = feat[(size_t)b * 3 * P + j0 + tid];
    __syncthreads();
    if (tid < A) {
        const float *wT = net.pfc_wT + (size_t)j0 * A + tid;
        float s = 0.f;
#pragma unroll 16
        for (int j = 0; j < n; ++j) s = fmaf(f[j], wT[(size_t)j * A], s);
        part[((size_t)b * 4 + q) * 384 + tid] = s;
    }
}

__global__ __launch_bounds__(128) void dualnet_heads19_fin_kernel(NetDev net, const float *__restrict__ feat, const float *__restrict__ part, int want_logits,
                                                                  float *__restrict__ policy, float *__restrict__ value, int *__restrict__ seq, int n_seq) {
    constexpr int P = 361, A = 362;
    __shared__ float lg[A + 6];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    wb_clear_seq(seq, n_seq, blockIdx.x, gridDim.x, tid);
    float m = -INFINITY;
    for (int a = tid; a < A; a += 128) {
        const float *p = part + (size_t)b * 4 * 384 + a;
        const float v = ((p[0] + p[384]) + (p[768] + p[1152])) + net.pfc_b[a];
        lg[a] = v;
        m = fmaxf(m, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(red[0], red[1]);
    float sum = 0.f;
    for (int a = tid; a < A; a += 128) sum += expf(lg[a] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[2 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / (red[2] + red[3]);
    for (int a = tid; a < A; a += 128) __builtin_nontemporal_store(want_logits ? lg[a] : expf(lg[a] - m) * inv, &policy[(size_t)b * A + a]);
    // value FC: wave 0, lanes 0 .. 47 = 3 outputs x 16 parts, 23 inputs each in sequence, the parts summed in a fixed (butterfly) order
    if (wave == 0) {
        const int c = lane >> 4, part_i = lane & 15;
        float sv = 0.f;
        if (c < 3) {
            const float *h = feat + (size_t)b * 3 * P + 2 * P, *wv = net.vfc_w + c * P;
            for (int j = part_i * 23; j < (part_i + 1) * 23 && j < P; ++j) sv = fmaf(h[j], wv[j], sv);
        }
        sv += __shfl_xor(sv, 8);
        sv += __shfl_xor(sv, 4);
        sv += __shfl_xor(sv, 2);
        sv += __shfl_xor(sv, 1);
        const float v0 = __shfl(sv, 0) + net.vfc_b[0], v1 = __shfl(sv, 16) + net.vfc_b[1], v2 = __shfl(sv, 32) + net.vfc_b[2];
        if (lane < 3) {
            const float vm = fmaxf(v0, fmaxf(v1, v2));
            const float e0 = expf(v0 - vm), e1 = expf(v1 - vm), e2 = expf(v2 - vm);
            value[(size_t)b * 3 + lane] = (lane == 0 ? e0 : (lane == 1 ? e1 : e2)) / (e0 + e1 + e2);
        }
    }
}

template <int TB>
__global__ __launch_bounds__(384) void dualnet_heads19_kernel(NetDev net, const float *__restrict__ feat, int batch, int want_logits,
                                                              float *__restrict__ policy, float *__restrict__ value, int *__restrict__ seq, int n_seq) {
    constexpr int P = 361, A = 362;
    __shared__ __attribute__((aligned(16))) float f[TB][3 * P + 1];     // (row stride 1 084 floats: a multiple of four)
    __shared__ float lg[TB][A + 2];
    __shared__ float vl[TB][4];
    const int tid = threadIdx.x, b0 = blockIdx.x * TB;
    wb_clear_seq(seq, n_seq, blockIdx.x, gridDim.x, tid);
    for (int e = tid; e < TB * 3 * P; e += 384) {
        const int bl = e / (3 * P), j = e - bl * 3 * P;
        f[bl][j] = b0 + bl < batch ? feat[(size_t)(b0 + bl) * 3 * P + j] : 0.f;
    }
    __syncthreads();
    if (tid < A) {
        // (the four quarters of dualnet_heads19_part_kernel, in its order)
        float s[TB], t[TB];
        const float *wT = net.pfc_wT + tid;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j0 = q < 2 ? q * 181 : 362 + (q - 2) * 180, n = q < 2 ? 181 : 180;
            float r[TB];
#pragma unroll
            for (int bl = 0; bl < TB; ++bl) r[bl] = 0.f;
            // (the same chain of FMAs per board, inputs in ascending order; the features of four inputs come with ONE 16-byte LDS
            // read per board - sixteen broadcast ds_read_b32 per input were what this loop issued, 63 % of its time waiting -
            // and four matrix rows are in flight at a time)
            auto one = [&](int jj) {
                const float w = wT[(size_t)jj * A];
#pragma unroll
                for (int bl = 0; bl < TB; ++bl) r[bl] = fmaf(f[bl][jj], w, r[bl]);
            };
            int j = j0;
            const int jend = j0 + n;
            for (; j < jend && (j & 3); ++j) one(j);
#pragma unroll 2
            for (; j + 4 <= jend; j += 4) {
                const float w0 = wT[(size_t)j * A], w1 = wT[(size_t)(j + 1) * A], w2 = wT[(size_t)(j + 2) * A], w3 = wT[(size_t)(j + 3) * A];
#pragma unroll
                for (int bl = 0; bl < TB; ++bl) {
                    const f32x4 fv = *reinterpret_cast<const f32x4 *>(&f[bl][j]);
                    r[bl] = fmaf(fv[0], w0, r[bl]);
                    r[bl] = fmaf(fv[1], w1, r[bl]);
                    r[bl] = fmaf(fv[2], w2, r[bl]);
                    r[bl] = fmaf(fv[3], w3, r[bl]);
                }
            }
            for (; j < jend; ++j) one(j);
#pragma unroll
            for (int bl = 0; bl < TB; ++bl) {
                if (q == 0) s[bl] = r[bl];
                else if (q == 1) s[bl] = s[bl] + r[bl];
                else if (q == 2) t[bl] = r[bl];
                else t[bl] = t[bl] + r[bl];
            }
        }
        const float bias = net.pfc_b[tid];
#pragma unroll
        for (int bl = 0; bl < TB; ++bl) lg[bl][tid] = (s[bl] + t[bl]) + bias;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int bl = wave; bl < TB; bl += 6) {
        const int b = b0 + bl;
        if (b >= batch) continue;
        float m = -INFINITY;
        for (int a = lane; a < A; a += 64) m = fmaxf(m, lg[bl][a]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        // (the same two-wave split of the sums as dualnet_heads19_fin_kernel: a = tid, tid + 128, .. per thread of a 128-thread block)
        float sum0 = 0.f, sum1 = 0.f;
        for (int a = lane; a < A; a += 128) sum0 += expf(lg[bl][a] - m);
        for (int a = lane + 64; a < A; a += 128) sum1 += expf(lg[bl][a] - m);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sum0 += __shfl_xor(sum0, o); sum1 += __shfl_xor(sum1, o); }
        const float inv = 1.f / (sum0 + sum1);
        for (int a = lane; a < A; a += 64) {
            const float v = lg[bl][a];
            __builtin_nontemporal_store(want_logits ? v : expf(v - m) * inv, &policy[(size_t)b * A + a]);
        }
        // value FC as in the small-launch kernel: 3 outputs x 16 parts of 23 inputs
        const int c = lane >> 4, part_i = lane & 15;
        float sv = 0.f;
        if (c < 3) {
            const float *h = &f[bl][2 * P], *wv = net.vfc_w + c * P;
            for (int j = part_i * 23; j < (part_i + 1) * 23 && j < P; ++j) sv = fmaf(h[j], wv[j], sv);
        }
        sv += __shfl_xor(sv, 8);
        sv += __shfl_xor(sv, 4);
        sv += __shfl_xor(sv, 2);
        sv += __shfl_xor(sv, 1);
        const float v0 = __shfl(sv, 0) + net.vfc_b[0], v1 = __shfl(sv, 16) + net.vfc_b[1], v2 = __shfl(sv, 32) + net.vfc_b[2];
        if (lane < 3) {
            const float vm = fmaxf(v0, fmaxf(v1, v2));
            const float e0 = expf(v0 - vm), e1 = expf(v1 - vm), e2 = expf(v2 - vm);
            value[(size_t)b * 3 + lane] = (lane == 0 ? e0 : (lane == 1 ? e1 : e2)) / (e0 + e1 + e2);
        }
    }
    (void)vl;
}

}  // namespace

namespace tg {

// pairs of workgroups a launch of `batch` boards would use (0: TG_FWD_ALGO / a shared device keep it off - see w1dband_wanted)
int w1dband_pairs(const tg_net *net, int batch) {
    int cus = net->num_cus;
    if (const int fc = tg::launch_caps().forward; fc > 0 && fc < cus) cus = fc;   // (a self-play move's sub-groups: CUs left to the other streams' tree kernels)
    const int cap = cus / 2;
    int pairs = batch < cap ? batch : cap;
    if (pairs >= 8) pairs &= ~7;                           // partners on the same XCD (consecutive workgroups go round the eight)
    return pairs;
}

// scratch: per pair the exchange rows and sequence numbers (zeroed once: the numbers only grow within a launch and every
// launch zeroes them again - memset node in front of the kernel), then the feature image [batch][3][361]
int w1dband_forward(tg_net *net, const float *planes, int batch, int want_logits, float *policy, float *value, int *overflow,
                    int *group_bits, hipStream_t stream) {
    using C = WbCfg;
    if (net->board_size != 19) return tg::fail(TG_ERR_ARG, "w1dband forward: 19x19 only");
    auto kern = net->dev.timeline ? dualnet_fwd_w1dband_kernel<true> : dualnet_fwd_w1dband_kernel<false>;
    static std::atomic<uint64_t> configured{0};
    if (tg::first_on_device(configured, net->device)) {
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(dualnet_fwd_w1dband_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(dualnet_fwd_w1dband_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    }
    const int pairs = w1dband_pairs(net, batch);
    const size_t xfloats = (size_t)(net->num_cus / 2) * C::PAIR_FLOATS;
    float *xmem = nullptr, *feat = nullptr;
    {
        std::lock_guard<std::mutex> lock(net->scratch_mu);
        auto &slot = net->wb_by_stream[stream];
        if (slot.cap < batch) {
            if (slot.mem) TG_HIP(hipFree(slot.mem));
            slot.mem = nullptr;
            slot.cap = 0;
            void *d = nullptr;
            const int cap = batch < 1024 ? 1024 : batch;
            TG_HIP(hipMalloc(&d, (xfloats + (size_t)cap * 3 * C::P + (size_t)512 * 4 * 384) * sizeof(float)));
            TG_HIP(hipMemsetAsync(d, 0, xfloats * sizeof(float), stream));      // (in the launching stream's order; from then on the heads kernels keep the sequence numbers at zero between launches)
            slot.mem = static_cast<float *>(d);
            slot.cap = cap;
        }
        xmem = slot.mem;
        feat = slot.mem + xfloats;
        // one cross-workgroup launch at a time on the device (as launch_band does, and sharing its state): two launches that each
        // got half of their workgroups onto the CUs would hold each other's partner bands off until the bounded waits give up.
        // When the launch stream changes, the new stream waits for what the previous one has queued.
        if (net->band_recorded && net->band_stream != stream) {
            if (!net->band_done) TG_HIP(hipEventCreateWithFlags(&net->band_done, hipEventDisableTiming));
            if (hipEventRecord(net->band_done, net->band_stream) == hipSuccess)
                TG_HIP(hipStreamWaitEvent(stream, net->band_done, 0));
            else
                (void)hipGetLastError();                   // (the previous stream is gone: nothing of it can be in flight)
        }
        net->band_stream = stream;
        net->band_recorded = true;
    }
    // (the sequence numbers start from zero in every launch: wb_clear_seq)
    int *const seq = reinterpret_cast<int *>(xmem + 4 * C::XROW_FLOATS);
    const int n_seq = net->num_cus / 2;
    if (tg::knob("TG_WB_TEST_MUTE")) TG_HIP(hipMemsetAsync(overflow + 1, 1, 1, stream));      // (tests: a non-zero second flag word mutes band 1)
    hipLaunchKernelGGL(kern, dim3(2 * pairs), dim3(C::NTHR), C::LDS_BYTES, stream, net->dev, planes, batch, feat, xmem, overflow, group_bits);
    TG_HIP(hipGetLastError());
    if (batch <= 512) {
        // partial sums [batch][4][384] behind the feature image
        float *part = feat + (size_t)batch * 3 * C::P;
        hipLaunchKernelGGL(dualnet_heads19_part_kernel, dim3(batch * 4), dim3(384), 0, stream, net->dev, feat, part);
        hipLaunchKernelGGL(dualnet_heads19_fin_kernel, dim3(batch), dim3(128), 0, stream, net->dev, feat, part, want_logits, policy, value, seq, n_seq);
    } else {
        hipLaunchKernelGGL(dualnet_heads19_kernel<16>, dim3((batch + 15) / 16), dim3(384), 0, stream, net->dev, feat, batch, want_logits, policy, value, seq, n_seq);
    }
    TG_HIP(hipGetLastError());
    return TG_OK;
}

}  // namespace tg
