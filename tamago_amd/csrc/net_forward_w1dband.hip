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
    static constexpr int LDS_BYTES = MISC + 16;
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
    int ord;             // tap order: 0 = (-1, 0, +1), 1 = (+1, 0, -1)
    // transform jobs: code = 100 + 10 * (s + 1) + half for V_half[s] of the regular mapping (s = -1 .. 3), 200 + r for row r = 8 / 9 of
    // row 9's mapping; + 1000 when the source is this layer's OUTPUT (a V row of the next layer); 0 = none
    int early, after, late;
};
constexpr WbStage wb_stage(int n) {
    switch (n) {
    case 0: return {3, 0, 0, 100 + 30 + 0, 100 + 20 + 0, 0};                         // S:  V_A[2], V_A[1]
    case 1: return {2, 0, 0, 100 + 20 + 1, 100 + 40 + 0, 100 + 30 + 1};              // 2A: V_B[1], V_A[3] (rows 3 / 6: stored by S, behind the barrier), V_B[2]
    case 2: return {2, 1, 0, 0, 100 + 40 + 1, 100 + 10 + 0};                         // 2B: V_B[3], V_A[0]
    case 3: return {1, 0, 0, 0, 0, 100 + 10 + 1};                                    // 1A: V_B[0]
    case 4: return {1, 1, 0, 0, 0, 0};                                               // 1B: (the halo copy)
    case 5: return {0, 0, 1, 0, 100 + 0 + 0, 1200 + 8};                              // 0A: V_A[-1] (the halo row: behind the barrier), the next layer's row 8
    default: return {0, 1, 1, 0, 100 + 0 + 1, 1200 + 9};                             // 0B: V_B[-1], the next layer's row 9
    }
}
// Every job writes a slot whose previous row has seen its last MFMA, and reads rows a barrier has published:
//   * slot (s + 1) mod 3 of a half: V[2] replaces V[-1] (last used 0x, slices 48-71), V[3] the row-9 stage's row (S), V[0]
//     replaces V[3] (2x, 48-71), V[-1] replaces V[2] (1x, 48-71), the row-9 rows replace V[0] (0x, 24-47), V[1] itself (0x, 0-23);
//   * a layer's input rows 0 / 3 / 6 are its predecessor's last outputs (0A -> stored under 0B, 0B -> stored under S): jobs that
//     read them (V[0] of classes 1 / 2, V[3] of classes 0 / 1) start behind 2A's barrier; row 9 is stored under 2A.

// Weight requests (next layer's unless noted), code = 16 kind + fragment:
//   kind 0: tap -1 -> the spare slot (double-buffered by layer parity), four a stage in 2B .. 0A
//   kind 1: tap 0, behind its last uses in 0B (taps in the order +1, 0, -1 there)
//   kind 2: tap +1, free since 0B's slice 24: four in 0B, twelve in the next layer's S (THIS layer's fragments there)
constexpr int wb_wreq(int n, int m) {
    if (n >= 2 && n <= 5 && (m == 50 || m == 56 || m == 62 || m == 68)) return 4 * (n - 2) + (m - 50) / 6;
    if (n == 6 && m >= 30 && m <= 60 && m % 2 == 0) {
        const int i = (m - 30) / 2;                        // (kc 0, low), (kc 0, high), (kc 1, low), (kc 1, high)
        return 16 + (i < 4 ? 4 + i : (i < 8 ? i - 4 : (i < 12 ? 12 + (i - 8) : 8 + (i - 12))));
    }
    if (n == 6 && (m == 62 || m == 65 || m == 68 || m == 71)) return 32 + 4 + (m - 62) / 3;
    if (n == 0 && m >= 1 && m <= 45 && (m - 1) % 4 == 0) {
        const int i = (m - 1) / 4;                         // 0 .. 11: (kc 0, high), (kc 1, low), (kc 1, high)
        return 32 + (i < 4 ? i : (i < 8 ? 12 + (i - 4) : 8 + (i - 8)));
    }
    return -1;
}
constexpr int wb_count(int n0, int m0, int n1, int m1) {   // requests strictly behind (n0, m0) up to and including (n1, m1 - 1) of the NEXT pass through the stages
    int cnt = 0;
    for (int n = n0, first = 1;; n = (n + 1) % 7, first = 0) {
        const int lo = first ? m0 + 1 : 0, hi = (n == n1 && !first) ? m1 : 72;
        for (int m = lo; m < hi; ++m)
            if (wb_wreq(n, m) >= 0) ++cnt;
        if (n == n1 && !first) break;
    }
    return cnt;
}
constexpr int kWbWaitTop = wb_count(5, 68, 0, 0);          // S, slice 0: tap -1 (last request 0A / 68) - behind it 0B's twenty
constexpr int kWbWaitTap0 = wb_count(6, 60, 0, 24);        // S, slice 24: tap 0 (last request 0B / 60)
static_assert(kWbWaitTop == 20 && kWbWaitTap0 == 10, "request schedule and wait counts");

template <bool PROF>
__global__ __launch_bounds__(256, 1) void dualnet_fwd_w1dband_kernel(
    NetDev net, const float *__restrict__ planes, int batch, float *__restrict__ feat, float *__restrict__ xmem,
    int *__restrict__ overflow) {
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
    int *const seq_mine = reinterpret_cast<int *>(pmem + 4 * C::XROW_FLOATS) + band * 16;
    int *const seq_theirs = reinterpret_cast<int *>(pmem + 4 * C::XROW_FLOATS) + (1 - band) * 16;
    int *const dead = reinterpret_cast<int *>(smem + C::MISC);

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
    if (tid == 0) *dead = 0;

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
    const float sgn = wave == 1 ? 1.f : -1.f;
    // taps along y in local order d = -1, 0, +1 (input row l + d): band 1 runs down the board (ky = 1 + d), band 0 up (ky = 1 - d)
    const int ky_m = band == 1 ? 0 : 2, ky_p = band == 1 ? 2 : 0;
    // this wave's weight fragments of a layer: slot 0 = tap 0, slot 2 = tap +1, slots 1 / 3 = tap -1 of even / odd layers
    i32x4v ua[4][2][2][4];
    {
        const int wlane0 = (tid & 63) * 16;
        const unsigned char *w0 = net.w1_w + (size_t)wave * 49152;
        w1_request_tap<1>(ua, w0 + ky_m * 16384, wlane0);
        static_for<16>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            w1_request<0>(ua, w0 + 16384, wlane0, std::integral_constant<int, (i < 4 ? 4 + i : (i < 8 ? i - 4 : (i < 12 ? 12 + (i - 8) : 8 + (i - 12))))>{});
        });
        static_for<4>([&](auto I_) { w1_request<2>(ua, w0 + ky_p * 16384, wlane0, std::integral_constant<int, 4 + decltype(I_)::value>{}); });
    }

    int kiter = 0;
    for (int b = pair; b < batch; b += n_pairs, ++kiter) {
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
        if (!(amax < (float)kWsRangeLimit)) ovf = 1;

        // ================= tower =================
        {
            const int glane = fresh_lane(), gli = glane & 15, glg = glane >> 4, wlane = glane * 16;
            const int uc = gli / 5, ut = gli - 5 * uc;             // row class, tile inside the half (gli 15: no unit)
            const bool uv = gli < 15;
            const int xa0 = wave == 0 ? 2 * ut - 1 : (wave == 2 ? 2 * ut + 1 : 2 * ut);
            const int xb0 = wave == 0 ? 2 * ut + 1 : (wave == 1 ? 2 * ut + 1 : (wave == 2 ? 2 * ut : 2 * ut + 2));
            // byte address of cell (l, x), 16-byte chunk `chunk` (before the k-chunk / half XOR), or of its class's chunk in the
            // zero / dump row
            auto cell = [&](int l, int x, int chunk, int invalid_rel, bool on) {
                const bool ok = on && x >= 0 && x < S && l >= -1 && l <= lmax;
                return (ok ? ((l + 1) * S + x) * 256 : invalid_rel) + ((chunk ^ wb_swz(l, x)) << 4);
            };
            // V-row reads of the regular stages: cell columns a / b of half h at row class base (s = 0); rows s = -1 and s = 3
            // have their own addresses (another swizzle class; s = -1 of class 0 is the halo row, s = 3 of class 2 is row 9)
            int rA[2][3], rB[2][3];                                // [half][0: s = -1, 1: s = 0 (+ s * ROWB for s = 1, 2), 2: s = 3]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                rA[h][0] = cell(3 * uc - 1, xa0 + 10 * h, glg * 2, C::ZERO_REL, uv);
                rA[h][1] = cell(3 * uc, xa0 + 10 * h, glg * 2, C::ZERO_REL, uv);
                rA[h][2] = cell(3 * uc + 3, xa0 + 10 * h, glg * 2, C::ZERO_REL, uv);
                rB[h][0] = cell(3 * uc - 1, xb0 + 10 * h, glg * 2, C::ZERO_REL, uv);
                rB[h][1] = cell(3 * uc, xb0 + 10 * h, glg * 2, C::ZERO_REL, uv);
                rB[h][2] = cell(3 * uc + 3, xb0 + 10 * h, glg * 2, C::ZERO_REL, uv);
            }
            // (rows s = 1, 2 of an invalid column must not step out of the zero row)
            int rAs[2], rBs[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool oka = uv && xa0 + 10 * h >= 0 && xa0 + 10 * h < S, okb = uv && xb0 + 10 * h >= 0 && xb0 + 10 * h < S;
                rAs[h] = oka ? ROWB : 0;
                rBs[h] = okb ? ROWB : 0;
            }
            // outputs of a regular stage: cells (3 c + k, 2 t) and (.., 2 t + 1), channels 16 wave + 4 glg ..: stores (outside: dump
            // row) and residual reads (outside: zero row)
            int oS[2][2], oR[2][2], oStr[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int x = 2 * (ut + 5 * h) + e;
                    oS[h][e] = cell(3 * uc, x, wave * 4 + glg, C::DUMP_REL, uv);
                    oR[h][e] = cell(3 * uc, x, wave * 4 + glg, C::ZERO_REL, uv);
                    oStr[h][e] = (uv && x < S) ? ROWB : 0;
                }
            // row 9's stage: li = tile (10 .. 15: no unit)
            const bool sv = gli < 10;
            const int sxa = wave == 0 ? 2 * gli - 1 : (wave == 2 ? 2 * gli + 1 : 2 * gli);
            const int sxb = wave == 0 ? 2 * gli + 1 : (wave == 1 ? 2 * gli + 1 : (wave == 2 ? 2 * gli : 2 * gli + 2));
            int sA[2], sB[2], sO[2];
            sA[0] = cell(8, sxa, glg * 2, C::ZERO_REL, sv);
            sA[1] = cell(9, sxa, glg * 2, C::ZERO_REL, sv);
            sB[0] = cell(8, sxb, glg * 2, C::ZERO_REL, sv);
            sB[1] = cell(9, sxb, glg * 2, C::ZERO_REL, sv);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                sO[e] = cell(9, 2 * gli + e, wave * 4 + glg, C::DUMP_REL, sv);   // (its residual: the same cell, or the zero row = dump row + 256)
            }
            // edge row l = 0 as the PARTNER's halo row l = -1 (its LDS layout): bytes into an exchange row, or -1
            auto x_off = [&](int h, int e) {
                const int x = 2 * (ut + 5 * h) + e;
                return (uv && uc == 0 && x < S) ? x * 256 + (((wave * 4 + glg) ^ wb_swz(-1, x)) << 4) : -1;
            };

            f32x4 dq[2][2][2];
            i32x4v vh[2][3][2], vl[2][3][2];                       // [half][slot (s + 1) mod 3][kc]
            f32x4 acc[1][4];                                       // (one set: slice i stores the previous stage's acc[i] to the exchange before its MFMA restarts it)
            f32x4 ez[4], eres[2], ev[2];
            float tvv[4];
            unsigned thh[2];
            f32x4 pshf, nshf;
            float pdown, ndown;
            int flag_seen = 0;
            // ---- cell reads and transforms of a job ----
            auto job_addr = [&](auto IN_, auto OUT_, auto JOB_, auto CB_) __attribute__((always_inline)) {
                constexpr int job = decltype(JOB_)::value, j = job % 1000, cb = decltype(CB_)::value;
                constexpr int base = job >= 1000 ? decltype(OUT_)::value : decltype(IN_)::value;
                int a;
                if constexpr (j >= 200) {
                    constexpr int r = j - 200 - 8;
                    a = cb ? sB[r] : sA[r];
                } else {
                    constexpr int s = (j - 100) / 10 - 1, h = (j - 100) % 10;
                    if constexpr (s == -1) a = cb ? rB[h][0] : rA[h][0];
                    else if constexpr (s == 3) a = cb ? rB[h][2] : rA[h][2];
                    else a = (cb ? rB[h][1] : rA[h][1]) + s * (cb ? rBs[h] : rAs[h]);
                }
                return a + base;
            };
            // read I (0 .. 7) of a job, in the order (a, b) x (kc 0 h 0), (kc 0 h 1), (kc 1 h 0), (kc 1 h 1)
            auto rd = [&](auto IN_, auto OUT_, auto JOB_, auto I_) __attribute__((always_inline)) {
                constexpr int i = decltype(I_)::value, cb = i & 1, kc = i >> 2, hh = (i >> 1) & 1;
                const int a = job_addr(IN_, OUT_, JOB_, std::integral_constant<int, cb>{});
                dq[cb][kc][hh] = lds_f32x4_at<0>(a ^ ((kc << 7) | (hh << 4)));
            };
            auto tr = [&](auto JOB_, auto I_) __attribute__((always_inline)) {
                constexpr int job = decltype(JOB_)::value, j = job % 1000, i = decltype(I_)::value, kc = i >> 3, hh = (i >> 2) & 1, q = i & 3;
                // destination: regular V_half[s] -> slot (s + 1) mod 3 of the half; row 9's stage: row 8 -> half A's slot 1, row 9 -> half B's
                constexpr int h = j >= 200 ? (j - 200 - 8) : (j - 100) % 10;
                constexpr int sl = j >= 200 ? 1 : (((j - 100) / 10 - 1) + 1) % 3;
                if constexpr (q == 0) {
                    tvv[0] = fmaf(dq[1][kc][hh][0], sgn, dq[0][kc][hh][0]);
                    tvv[1] = fmaf(dq[1][kc][hh][1], sgn, dq[0][kc][hh][1]);
                } else if constexpr (q == 1) {
                    tvv[2] = fmaf(dq[1][kc][hh][2], sgn, dq[0][kc][hh][2]);
                    tvv[3] = fmaf(dq[1][kc][hh][3], sgn, dq[0][kc][hh][3]);
                } else if constexpr (q == 2) {
                    thh[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{tvv[0], tvv[1]}, f16x2));
                    vh[h][sl][kc][2 * hh] = (int)thh[0];
                    vl[h][sl][kc][2 * hh] = (int)low_pieces(tvv[0], tvv[1], thh[0]);
                } else {
                    thh[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{tvv[2], tvv[3]}, f16x2));
                    vh[h][sl][kc][2 * hh + 1] = (int)thh[1];
                    vl[h][sl][kc][2 * hh + 1] = (int)low_pieces(tvv[2], tvv[3], thh[1]);
                }
            };
            // a whole job at once (a group's first layer: nothing was prepared under a previous one)
            auto job_now = [&](auto IN_, auto OUT_, auto JOB_) __attribute__((always_inline)) {
                static_for<8>([&](auto I_) { rd(IN_, OUT_, JOB_, I_); });
                static_for<16>([&](auto I_) { tr(JOB_, I_); });
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
                // sequence number of the edge row the PREVIOUS layer published (this one consumes it), and its buffer parity
                const int pub = kiter * 11 + layer;                // number of edge rows published before this layer's
                f32x4 shf;
                float down;
                if (layer == 0) {
                    // a group's first layer: its V rows of stages S and 2A / 2B that a previous layer would have made, its shift
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(nshf) : "v"(glg * 16 + wave * 64), "s"(net.w1_shift) : "memory");
                    {
                        int zoff = 0;
                        asm volatile("" : "+v"(zoff));
                        asm volatile("global_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(ndown) : "v"(zoff), "s"(net.w1_down) : "memory");
                    }
                    job_now(IN_, OUT_, std::integral_constant<int, 200 + 8>{});
                    job_now(IN_, OUT_, std::integral_constant<int, 200 + 9>{});
                }
                // tap -1 of this layer has arrived (and with it everything requested before 0B: the next shift / scale, the halo copy)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWbWaitTop) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                shf = nshf;
                down = ndown;

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
#define WB_OADDR(e, store) (pk == 3 ? ((store) ? sO[e] : (sO[e] >= C::DUMP_REL ? sO[e] + 256 : sO[e])) : ((store) ? oS[ph & 1][e] : oR[ph & 1][e]) + (pk % 3) * oStr[ph & 1][e])
                    if constexpr (i < 4) {
                        lds_f32x4_put<par * 16384 + i * 1024>(exw, acc[0][i]);
                    } else if constexpr (i == 10) {
                        __syncthreads();
                    } else if constexpr (i == 11 || i == 12) {
                        ez[2 * (i - 11)] = lds_f32x4_at<par * 16384 + (2 * (i - 11)) * 4096>(exr);
                        ez[2 * (i - 11) + 1] = lds_f32x4_at<par * 16384 + (2 * (i - 11) + 1) * 4096>(exr);
                    } else if constexpr (i == 13) {
                        if constexpr (RS) {
                            const int zr = C::ZERO_REL + (glane * 16) % 256;
                            eres[0] = lds_f32x4_at<OB>(null_epi ? zr : WB_OADDR(0, false));
                            eres[1] = lds_f32x4_at<OB>(null_epi ? zr : WB_OADDR(1, false));
                        }
                    } else if constexpr (i >= 19 && i < 35) {
                        constexpr int kk = i - 19, cc = kk >> 3, e = (kk >> 1) & 3, part = kk & 1;
                        if constexpr (part == 0) {
                            ev[cc][e] = cc == 0 ? (ez[0][e] + ez[1][e]) + ez[2][e] : (ez[1][e] - ez[2][e]) - ez[3][e];
                        } else {
                            float tt = fmaf(ev[cc][e], PREV ? pdown : down, PREV ? pshf[e] : shf[e]);
                            if constexpr (RS) tt += eres[cc][e];
                            ev[cc][e] = fmaxf(tt, 0.f);
                        }
                    } else if constexpr (i == 35 || i == 36) {
                        constexpr int e = i - 35;
                        amax = fmaxf(fmaxf(amax, ev[e][0]), ev[e][1]);
                        amax = fmaxf(fmaxf(amax, ev[e][2]), ev[e][3]);
                        lds_f32x4_put<OB>(null_epi ? C::DUMP_REL + (glane * 16) % 256 : WB_OADDR(e, true), ev[e]);
                    } else if constexpr (i == 37 || i == 38) {
                        // the edge row l = 0 (step 0 of row class 0) also goes to the partner: layers 0 .. 10 (PREV: the previous layer's)
                        if constexpr (pk == 0) {
                            constexpr int e = i - 37;
                            const int lay = PREV ? layer - 1 : layer;
                            const int xo = x_off(ph, e);
                            if (lay >= 0 && lay < kTowerLayers - 1 && xo >= 0) {
                                float *dst = pmem + (size_t)(band * 2 + ((kiter * 11 + lay) & 1)) * C::XROW_FLOATS;
                                const float *p = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(dst) + xo);
                                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(ev[e]) : "memory");
                            }
                        }
                    }
#undef WB_OADDR
                };

                static_for<7>([&](auto N_) {
                    constexpr int n = decltype(N_)::value;
                    constexpr WbStage st = wb_stage(n);
                    constexpr int par = (PAR + n) & 1;
                    using JE = std::integral_constant<int, st.early>;
                    using JA = std::integral_constant<int, st.after>;
                    using JL = std::integral_constant<int, st.late>;
                    // ---- before the stage ----
                    if constexpr (n == 4) {
                        // 1B: the partner's edge row of the previous layer must have been published (asked for in 1A)
                        if (layer > 0) {
                            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // (1A's four weight requests lie behind the question)
                            __builtin_amdgcn_sched_barrier(0);
                            int seen = __builtin_amdgcn_readfirstlane(flag_seen);
                            int spins = 0;
                            while (seen < pub) {
                                if (*reinterpret_cast<volatile int *>(dead)) break;
                                __builtin_amdgcn_s_sleep(2);
                                seen = __hip_atomic_load(seq_theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (++spins > kWbSpinLimit) {
                                    *reinterpret_cast<volatile int *>(dead) = 1;
                                    if (net.band_timeouts && glane == 0 && wave == 0) atomicAdd(net.band_timeouts, 1u);
                                    break;
                                }
                            }
                            // copy it into the halo row of this layer's INPUT buffer: 19 x 256 B = 4 x 1 KB + 768 B, LDS-DMA, sc1
                            const float *src = pmem + (size_t)((1 - band) * 2 + ((pub - 1) & 1)) * C::XROW_FLOATS;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(src) + wave * 1024 + glane * 16),
                                                             (__attribute__((address_space(3))) void *)(smem + IN + wave * 1024), 16, 0, 16);
                            if (wave == 0 && glane < 48)
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(src) + 4096 + glane * 16),
                                                                 (__attribute__((address_space(3))) void *)(smem + IN + 4096), 16, 0, 16);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    static_for<72>([&](auto M_) {
                        constexpr int m = decltype(M_)::value;
                        constexpr int NT = st.k == 3 ? 2 : 3;
                        if constexpr (m < 4) epi(N_, M_);           // (exchange write of acc[m]: in front of the MFMA that restarts it)
                        if constexpr (m < 24 * NT) {
                            constexpr int ti = m / 24, q = m % 24, kc = q / 12, pr = (q / 4) % 3, c = q % 4;
                            // row 9's stage: taps -1 (row 8: half A's slot 1), 0 (row 9: half B's slot 1); regular: ord 0 = (-1, 0, +1), 1 = (+1, 0, -1)
                            constexpr int d = st.k == 3 ? ti - 1 : (st.ord == 0 ? ti - 1 : 1 - ti);
                            constexpr int slot = d == -1 ? S1 : (d == 0 ? 0 : 2);
                            constexpr int vhf = st.k == 3 ? ti : st.h, vsl = st.k == 3 ? 1 : (st.k + d + 1) % 3;
                            // waits for the fragments (see wb_wreq)
                            if constexpr (n == 0 && m == 24) {
                                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWbWaitTap0) : "memory");
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if constexpr (n == 1 && m == 48) {
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tap +1; also: every edge store of the previous layer has landed
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            bool run = true;
                            if constexpr (st.k == 3) run = band == 0;             // (uniform: band 1 has no row 9)
                            if constexpr (st.k == 3 && q < 4 && ti == 0) {
                                if (!run) acc[0][c] = f32x4{0.f, 0.f, 0.f, 0.f};    // (its null epilogue must not see a stale accumulator)
                            }
                            if (run) {
                                if constexpr (pr == 0)
                                    acc[0][c] = mfma16<F>(ua[slot][kc][1][c], vh[vhf][vsl][kc], q < 4 && ti == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[0][c]);
                                else if constexpr (pr == 1) acc[0][c] = mfma16<F>(ua[slot][kc][0][c], vl[vhf][vsl][kc], acc[0][c]);
                                else acc[0][c] = mfma16<F>(ua[slot][kc][0][c], vh[vhf][vsl][kc], acc[0][c]);
                            }
                        }
                        // ---- what rides along ----
                        if constexpr (m >= 4 && m < 39) epi(N_, M_);
                        if constexpr (st.early != 0) {
                            if constexpr (m < 8) rd(IN_, OUT_, JE{}, M_);
                            if constexpr (m >= 8 && m < 24) tr(JE{}, std::integral_constant<int, m - 8>{});
                        }
                        if constexpr (st.after != 0) {
                            if constexpr (m >= 24 && m < 32) rd(IN_, OUT_, JA{}, std::integral_constant<int, m - 24>{});
                            if constexpr (m >= 32 && m < 48) tr(JA{}, std::integral_constant<int, m - 32>{});
                        }
                        if constexpr (st.late != 0) {
                            if constexpr (m >= 46 && m < 54) rd(IN_, OUT_, JL{}, std::integral_constant<int, m - 46>{});
                            if constexpr (m >= 56) tr(JL{}, std::integral_constant<int, m - 56>{});
                        }
                        // weight requests
                        {
                            constexpr int code = wb_wreq(n, m), kind = code >> 4;
                            using FR = std::integral_constant<int, (code & 15)>;
                            if constexpr (code >= 0) {
                                if constexpr (kind == 0) w1_request<S1N>(ua, wnext + ky_m * 16384, wlane, FR{});
                                else if constexpr (kind == 1) w1_request<0>(ua, wnext + 16384, wlane, FR{});
                                else w1_request<2>(ua, (n == 0 ? wcur : wnext) + ky_p * 16384, wlane, FR{});
                            }
                        }
                        // hand-off riders
                        if constexpr (n == 2 && m == 11) {
                            // 2B, behind its barrier: every wave has passed 2A's vmcnt(0) - the previous layer's edge row is in memory
                            if (layer > 0 && wave == 0 && glane == 0) __hip_atomic_store(seq_mine, pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if constexpr (n == 3 && m == 4) {
                            // 1A: ask for the partner's sequence number (looked at in front of 1B)
                            if (layer > 0) {
                                int zoff = 0;
                                asm volatile("" : "+v"(zoff));
                                asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(flag_seen) : "v"(zoff), "s"(seq_theirs) : "memory");
                            }
                        }
                        if constexpr (n == 4 && m == 20) {
                            // 1B: shift and scale of the next layer
                            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(nshf) : "v"(glg * 16 + wave * 64), "s"(net.w1_shift + next_layer * 64) : "memory");
                            int zoff = 0;
                            asm volatile("" : "+v"(zoff));
                            asm volatile("global_load_dword %0, %1, %2" : "=v"(ndown) : "v"(zoff), "s"(net.w1_down + next_layer) : "memory");
                        }
                        if constexpr (n == 5 && m == 9) {
                            // 0A, in front of its barrier: this wave's share of the halo copy has landed (behind it: 1B's two loads and four requests)
                            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
                pshf = shf;
                pdown = down;
                if (!(amax < (float)kWsRangeLimit)) ovf = 1;
            };
            using IX = std::integral_constant<int, C::X_OFF>;
            using IH = std::integral_constant<int, C::H_OFF>;
            {
                float z0, z1, z2, z3;
                asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z0), "=v"(z1), "=v"(z2), "=v"(z3));
                pshf = f32x4{z0, z1, z2, z3};
                pdown = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[0][c] = pshf;      // (layer 0's stage S carries a null epilogue)
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
                eres[0] = lds_f32x4_at<C::X_OFF>(oR[1][0]);
                eres[1] = lds_f32x4_at<C::X_OFF>(oR[1][1]);
                __syncthreads();
                static_for<4>([&](auto I_) { ez[decltype(I_)::value] = lds_f32x4_at<par * 16384 + decltype(I_)::value * 4096>(exr); });
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o0 = (ez[0][e] + ez[1][e]) + ez[2][e], o1 = (ez[1][e] - ez[2][e]) - ez[3][e];
                    ev[0][e] = fmaxf(fmaf(o0, pdown, pshf[e]) + eres[0][e], 0.f);
                    ev[1][e] = fmaxf(fmaf(o1, pdown, pshf[e]) + eres[1][e], 0.f);
                    amax = fmaxf(fmaxf(amax, ev[0][e]), ev[1][e]);
                }
                lds_f32x4_put<C::X_OFF>(oS[1][0], ev[0]);
                lds_f32x4_put<C::X_OFF>(oS[1][1], ev[1]);
                if (!(amax < (float)kWsRangeLimit)) ovf = 1;
                __syncthreads();
            }
        }
        if (*reinterpret_cast<volatile int *>(dead)) ovf = 1;
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
    }
    if (ovf && overflow) atomicOr(overflow, 1);
}

// The two fully connected layers + softmaxes of a 19x19 launch: policy FC 722 -> 362 and value FC 361 -> 3 on the features
// dualnet_fwd_w1dband_kernel left in global memory ([board][policy 0 | policy 1 | value][361]).  TB boards per workgroup:
// thread a (0 .. 361) owns policy output a of all TB boards - the FC matrix (transposed, [722][362] fp32, 1 MB) is read once
// per workgroup, coalesced over a, the features come from LDS as broadcasts.  Plain fp32 FMAs (the reference's arithmetic).
template <int TB>
__global__ __launch_bounds__(384) void dualnet_heads19_kernel(NetDev net, const float *__restrict__ feat, int batch, int want_logits,
                                                              float *__restrict__ policy, float *__restrict__ value) {
    constexpr int P = 361, A = 362;
    __shared__ float f[TB][3 * P + 1];
    __shared__ float lg[TB][A + 2];
    __shared__ float vl[TB][4];
    const int tid = threadIdx.x, b0 = blockIdx.x * TB;
    for (int e = tid; e < TB * 3 * P; e += 384) {
        const int bl = e / (3 * P), j = e - bl * 3 * P;
        f[bl][j] = b0 + bl < batch ? feat[(size_t)(b0 + bl) * 3 * P + j] : 0.f;
    }
    __syncthreads();
    if (tid < A) {
        float s[TB][2];
#pragma unroll
        for (int bl = 0; bl < TB; ++bl) { s[bl][0] = net.pfc_b[tid]; s[bl][1] = 0.f; }
        const float *wT = net.pfc_wT + tid;
#pragma unroll 2
        for (int j = 0; j < 2 * P; j += 2) {
            const float w0 = wT[(size_t)j * A], w1 = wT[(size_t)(j + 1) * A];
#pragma unroll
            for (int bl = 0; bl < TB; ++bl) {
                s[bl][0] = fmaf(f[bl][j], w0, s[bl][0]);
                s[bl][1] = fmaf(f[bl][j + 1], w1, s[bl][1]);
            }
        }
#pragma unroll
        for (int bl = 0; bl < TB; ++bl) lg[bl][tid] = s[bl][0] + s[bl][1];
    } else if (tid < A + 3 * TB && tid - A < 3 * TB) {
        const int q = tid - A, bl = q / 3, c = q - bl * 3;
        const float *wv = net.vfc_w + c * P;
        float s0 = net.vfc_b[c], s1 = 0.f;
        for (int j = 0; j + 1 < P; j += 2) {
            s0 = fmaf(f[bl][2 * P + j], wv[j], s0);
            s1 = fmaf(f[bl][2 * P + j + 1], wv[j + 1], s1);
        }
        s0 = fmaf(f[bl][2 * P + P - 1], wv[P - 1], s0);
        vl[bl][c] = s0 + s1;
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
        float sum = 0.f;
        for (int a = lane; a < A; a += 64) sum += expf(lg[bl][a] - m);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.f / sum;
        for (int a = lane; a < A; a += 64) {
            const float v = lg[bl][a];
            __builtin_nontemporal_store(want_logits ? v : expf(v - m) * inv, &policy[(size_t)b * A + a]);
        }
        if (lane < 3) {
            const float v0 = vl[bl][0], v1 = vl[bl][1], v2 = vl[bl][2];
            const float vm = fmaxf(v0, fmaxf(v1, v2));
            const float e0 = expf(v0 - vm), e1 = expf(v1 - vm), e2 = expf(v2 - vm);
            const float es = e0 + e1 + e2;
            value[(size_t)b * 3 + lane] = (lane == 0 ? e0 : (lane == 1 ? e1 : e2)) / es;
        }
    }
}

}  // namespace

namespace tg {

// pairs of workgroups a launch of `batch` boards would use (0: TG_FWD_ALGO / a shared device keep it off - see w1dband_wanted)
int w1dband_pairs(const tg_net *net, int batch) {
    const int cap = net->num_cus / 2;
    int pairs = batch < cap ? batch : cap;
    if (pairs >= 8) pairs &= ~7;                           // partners on the same XCD (consecutive workgroups go round the eight)
    return pairs;
}

// scratch: per pair the exchange rows and sequence numbers (zeroed once: the numbers only grow within a launch and every
// launch zeroes them again - memset node in front of the kernel), then the feature image [batch][3][361]
int w1dband_forward(tg_net *net, const float *planes, int batch, int want_logits, float *policy, float *value, int *overflow,
                    hipStream_t stream) {
    using C = WbCfg;
    if (net->board_size != 19) return tg::fail(TG_ERR_ARG, "w1dband forward: 19x19 only");
    auto kern = dualnet_fwd_w1dband_kernel<false>;
    static std::atomic<uint64_t> configured{0};
    if (tg::first_on_device(configured, net->device))
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
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
            TG_HIP(hipMalloc(&d, (xfloats + (size_t)cap * 3 * C::P) * sizeof(float)));
            slot.mem = static_cast<float *>(d);
            slot.cap = cap;
        }
        xmem = slot.mem;
        feat = slot.mem + xfloats;
    }
    // the sequence numbers start from zero in every launch
    TG_HIP(hipMemsetAsync(xmem, 0, xfloats * sizeof(float), stream));
    hipLaunchKernelGGL(kern, dim3(2 * pairs), dim3(C::NTHR), C::LDS_BYTES, stream, net->dev, planes, batch, feat, xmem, overflow);
    TG_HIP(hipGetLastError());
    if (batch <= 256) {
        hipLaunchKernelGGL(dualnet_heads19_kernel<2>, dim3((batch + 1) / 2), dim3(384), 0, stream, net->dev, feat, batch, want_logits, policy, value);
    } else {
        hipLaunchKernelGGL(dualnet_heads19_kernel<16>, dim3((batch + 15) / 16), dim3(384), 0, stream, net->dev, feat, batch, want_logits, policy, value);
    }
    TG_HIP(hipGetLastError());
    return TG_OK;
}

}  // namespace tg
