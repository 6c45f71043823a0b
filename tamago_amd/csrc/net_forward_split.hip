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
#include "split_common.h"

namespace {

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
    // one [row][64 B] image + dump row M (rows >= M of the last row-tile store there: no branches) + a 256-byte,
    // 256-byte-aligned block of zeros: a padding tap reads zero block + (address of the row it would have read) mod 256,
    // i.e. zeros from the banks its own row would have used - conflict-free like the real rows.  (One shared zero row
    // cost 8 instead of 4 LDS cycles on 94 of the 144 (row tile, tap) fragments of a 3-board group:
    // tools/microbench/lds_read_patterns.hip, profiles/r03_microbench_lds_read_patterns.txt.)
    static constexpr int ZOFF = ((M + 1) * 64 + 255) & ~255;
    static constexpr int IMG = ZOFF + 256;
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
    static constexpr int HQ_OFF = RES_OFF;                       // head phase (9x9): policy features as f16 pairs (12 KB)
    static constexpr int SS_OFF = RES_OFF + RES_BYTES;           // folded BN scale [13][64] + shift [13][64]
    // head tables, staged once per workgroup: 1x1 weights [64][4] (policy 0, policy 1, value, 0), policy FC bias [A]
    // (padded), BN scale / shift of the three head channels [8].  (In the heads every use of a kernel-argument
    // pointer was a reload from scratch followed by a dependent global load.)
    static constexpr int HW_OFF = SS_OFF + 2 * 13 * 64 * 4;
    static constexpr int HB_OFF = HW_OFF + 64 * 4 * 4;
    static constexpr int HS_OFF = HB_OFF + ((A + 3) & ~3) * 4;
    static constexpr int VW_OFF = HS_OFF + 8 * 4;                 // value FC weights [3][P] + bias [3]
    static constexpr int HD1_OFF = (VW_OFF + ((3 * P + 3 + 3) & ~3) * 4 + 15) & ~15;   // 9x9: 1x1 fragment image 4 KB + table
    static constexpr int PIPE_BYTES = HD1_OFF + (BIG ? 0 : 4096 + 128);
    // head phase (after the last layer): fp32 activations [M][72 floats] from offset 0, scratch behind the BN table
    static constexpr int ROW_BYTES = kRowBytes;
    static constexpr int AUX = PIPE_BYTES;
    static constexpr int LDS_BYTES = AUX + G * (3 * P + A + 4) * 4 + 256;
};


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
    for (int e = tid; e < NP * 2 * 64; e += NTHR)
        reinterpret_cast<unsigned *>(smem + (e >> 6) * IMG + C::ZOFF)[e & 63] = 0u;

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
    if constexpr (!C::BIG) stage_head_tables<C, NTHR>(smem, net, tid);
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
            const int row = base_row[r] + toff;             // (may lie outside the image when !ok: only its banks matter)
            const int nat = row * 64 + ((lg ^ ((row >> 1) & 3)) << 4);
            return ok ? nat : C::ZOFF + (nat & 255);
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
                    wrow[r] = brow[r] < M ? brow[r] : M;                  // store into the dump row
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
                        // (rows beyond the board do not count: at 19x19 their "residual" is row 0 of the global scratch image, which
                        // nobody writes - fresh from hipMalloc it could hold anything, and the first launch on a stream fell back)
                        if (brow[r] < M) amax = fmaxf(fmaxf(amax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
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
            if (layer == kTowerLayers) {
                if constexpr (C::BIG) epilogue(N{}, T{}, T{});          // 19x19: fp32 image for the generic head code
                else epilogue(N{}, T{}, N{});                           // 9x9: the heads read the activation images
            }
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
        if constexpr (C::BIG)
            run_heads_split<S, G, C, NTHR>(smem, net, b0, batch, want_logits, policy, value, tid, wave,
                                           (net.timeline && blockIdx.x == 0 && grp == blockIdx.x) ? net.timeline + 40 : nullptr);
        else
            run_heads_mfma<S, G, C, NTHR>(smem, net, b0, batch, want_logits, policy, value, tid, wave,
                                          (net.timeline && blockIdx.x == 0 && grp == blockIdx.x) ? net.timeline + 40 : nullptr);
        __syncthreads();
        stamp();
        // the head scratch overlapped the activation images' zero rows
        for (int e = tid; e < NP * 2 * 64; e += NTHR)
            reinterpret_cast<unsigned *>(smem + (e >> 6) * IMG + C::ZOFF)[e & 63] = 0u;
    }
    if (ovf && overflow) atomicOr(overflow, 1);
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
    // TG_FWD_CUS=n: at most n workgroups (CUs) for the forward pass - leaves CUs to the tree kernels of another lock-step
    // group running on a second stream (the persistent workgroups of a full-width launch own every CU's LDS and registers)
    static const int cu_cap = tg::knob("TG_FWD_CUS") ? atoi(tg::knob("TG_FWD_CUS")) : 0;
    const int cus = cu_cap > 0 && cu_cap < net->num_cus ? cu_cap : net->num_cus;
    const int grid = groups < cus ? groups : cus;
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

// Head phase on the 16-bit matrix pipe (9x9): fragment images of the 1x1 convolutions (batch norm folded: scale into
// the weights in fp64, shift -> accumulator start) and of the policy FC, both as f16 x 2 pieces with a power-of-two
// scaling per image.  hp_w [2][64], hv_w [64], head_ss = (scale0, shift0, scale1, shift1, value scale, value shift),
// pfc_w [A][2P].
int heads_prepare(tg_net *net, const float *hp_w, const float *hv_w, const float *head_ss, const float *pfc_w, int P) {
    const int A = P + 1, K = 2 * P;
    auto scale_exp = [](double mx) {
        int e = 0;
        if (mx > 0.0 && std::isfinite(mx)) {
            int ex;
            std::frexp(mx, &ex);
            e = 10 - ex;
        }
        return e > 40 ? 40 : (e < -40 ? -40 : e);
    };
    auto put = [](std::vector<uint16_t> &img, size_t base, double v) {       // two pieces of v at img[base], img[base + 512]
        const uint16_t h = f32_to_f16_rn((float)v);
        img[base] = h;
        img[base + 512] = f32_to_f16_rn((float)((v - (double)f16_to_f32(h)) * 2048.0));
    };
    // ---- 1x1 convolutions: channel 0 / 1 = policy, 2 = value, 3..15 = 0 ----
    auto w1 = [&](int ch, int k) -> double {
        if (ch == 0) return (double)hp_w[k] * head_ss[0];
        if (ch == 1) return (double)hp_w[64 + k] * head_ss[2];
        if (ch == 2) return (double)hv_w[k] * head_ss[4];
        return 0.0;
    };
    double mx = 0.0;
    for (int ch = 0; ch < 3; ++ch)
        for (int k = 0; k < 64; ++k) mx = std::fmax(mx, std::fabs(w1(ch, k)));
    const int e1 = scale_exp(mx);
    std::vector<uint16_t> img1((size_t)2 * 2 * 512, 0);      // [kc][piece][lane][8]
    for (int kc = 0; kc < 2; ++kc)
        for (int lane = 0; lane < 64; ++lane)
            for (int el = 0; el < 8; ++el)
                put(img1, ((size_t)kc * 2) * 512 + lane * 8 + el, w1(lane & 15, kc * 32 + (lane >> 4) * 8 + el) * std::ldexp(1.0, e1));
    std::vector<float> tab1(20, 0.f);
    tab1[0] = (float)((double)head_ss[1] * std::ldexp(1.0, e1));
    tab1[1] = (float)((double)head_ss[3] * std::ldexp(1.0, e1));
    tab1[2] = (float)((double)head_ss[5] * std::ldexp(1.0, e1));
    tab1[16] = std::ldexp(1.f, -e1);
    // ---- policy FC: logits[a] = sum_k W[a][k] h[k] + bias[a], k = channel * P + position ----
    constexpr int NT = 6, KS = 6;
    if (A > NT * 16 || K > KS * 32) return tg::fail(TG_ERR_ARG, "heads_prepare: board too large for the fragment image");
    mx = 0.0;
    for (size_t i = 0; i < (size_t)A * K; ++i) mx = std::fmax(mx, std::fabs((double)pfc_w[i]));
    const int e2 = scale_exp(mx);
    std::vector<uint16_t> img2((size_t)NT * KS * 2 * 512, 0);  // [nt][ks][piece][lane][8]
    for (int nt = 0; nt < NT; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int el = 0; el < 8; ++el) {
                    const int a = nt * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + el;
                    const double v = (a < A && k < K) ? (double)pfc_w[(size_t)a * K + k] * std::ldexp(1.0, e2) : 0.0;
                    put(img2, ((size_t)(nt * KS + ks) * 2) * 512 + lane * 8 + el, v);
                }
    std::vector<float> tab2(4, 0.f);
    tab2[0] = std::ldexp(1.f, -e2);
    auto up = [&](const void *src, size_t bytes, const void **dst) {
        void *d = nullptr;
        TG_HIP(hipMalloc(&d, bytes));
        net->allocs.push_back(d);
        TG_HIP(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
        *dst = d;
        return (int)TG_OK;
    };
    int rc;
    if ((rc = up(img1.data(), img1.size() * 2, reinterpret_cast<const void **>(&net->dev.hd1_img))) ||
        (rc = up(tab1.data(), tab1.size() * 4, reinterpret_cast<const void **>(&net->dev.hd1_tab))) ||
        (rc = up(img2.data(), img2.size() * 2, reinterpret_cast<const void **>(&net->dev.pfc_img))) ||
        (rc = up(tab2.data(), tab2.size() * 4, reinterpret_cast<const void **>(&net->dev.pfc_tab))))
        return rc;
    return TG_OK;
}

// group = boards per workgroup (1 or 3); 9x9 only.
int split_forward(tg_net *net, int group, const float *planes, int batch, int want_logits, float *policy,
                  float *value, int *overflow, hipStream_t stream) {
    if (net->board_size == 19) return launch_split<19, 1, FmtF16>(net, planes, batch, want_logits, policy, value, overflow, stream);
    if (net->board_size != 9) return tg::fail(TG_ERR_ARG, "split forward: 9x9 and 19x19 only");
    if (group == 3) {
        if (const char *env = tg::knob("TG_SPLIT_SPAN")) {              // tuning knob
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
