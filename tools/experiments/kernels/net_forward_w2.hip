// DualNet forward for gfx950, split operands (f16 x 2 pieces, three MFMAs per fp32 product - the scheme of
// net_forward_split.hip, same accuracy class) with TWO waves per SIMD.
//
// Why a second kernel (profiles/r02_pmc_forward_split_9x9_b65536.json, r02_phase_timeline_forward_split_9x9.txt):
// the one-wave-per-SIMD kernel keeps the matrix pipe 56 % busy.  Its loop loses 25 % to fragment reads a lone
// wave cannot overlap with its own MFMAs (in-order issue), its epilogues run on a VALU that one wave can feed
// one instruction every ~4 cycles, and a quarter of their instructions only move accumulators out of AGPRs.
// Here:
//   * eight waves per workgroup = two per SIMD, <= 256 registers each (accumulators in VGPRs: no v_accvgpr moves):
//     while one wave waits for a fragment or issues VALU / LDS work, its partner's MFMAs keep the pipe busy;
//   * wave tile = 32 output channels x 64 positions (2 x 4 tiles of v_mfma_f32_16x16x32_f16), 64 accumulator
//     registers; wave w: row quarter rq = w % 4, channel half chh = w / 4 (waves w and w + 4 share a SIMD);
//   * weights reach the CU ONCE per k-chunk: every wave copies 1 KB of the chunk's 8 KB (L2-resident image in
//     fragment order) into a two-slot LDS ring with global_load_lds, two chunks ahead; one s_barrier per chunk
//     publishes the slot that has landed and frees the one that was read.  (With every wave fetching its own
//     fragments from L2 - 32 KB per chunk and CU through the vector-memory path - the loop lost 4.8 k of 21 k cycles
//     per layer: profiles/r03_w2_direct_weights_ablation_phase_timeline.txt.)
//   * per k-chunk (K = 32) and wave: 8 + 4 ds_read_b128 (activations: four row tiles x two pieces; weights: two
//     channel tiles x two pieces), 24 MFMAs;
//   * batch norm is folded away on the host: the scale goes INTO the weights (in fp64, before the f16 split),
//     the shift becomes the accumulators' initial value, so an epilogue is: combine the two accumulator sets,
//     one power-of-two factor, (+ residual), ReLU, split, store.
//
// MFMA shape: the board is POWER-capped (rocm-smi: 1.31-1.37 kW of 1.4 kW while this kernel runs; the shader clock
// follows the work's energy, profiles/r03_power_and_pmc_s32_vs_split16.txt).  On random f16 data, MFMAs back to back
// from registers sustain 1.93 PFLOP/s as 16x16x32 but only 1.72 PFLOP/s as 32x32x16 (tools/microbench/mfma_power.hip,
// profiles/r03_microbench_mfma_power.txt).  A first version of this kernel on 32x32x16 tiles (git: "s32") had its
// loop at 90-97 % of the MFMA time and still ran 10 % slower in wall time than the 16x16x32 kernel: 1.91 vs 2.22 GHz.
//
// Layouts as in net_forward_split.hip: activation images [piece 2][k-chunk 2][row][4 x 16 B], slot = (k / 8) XOR
// ((row >> 1) & 3) - conflict-free ds_read_b128 B-fragments for every tap shift.  Padding taps read ZEROS FROM THE BANKS
// THEIR OWN ROW WOULD HAVE USED: every image ends in a 256-byte (one full bank row) block of zeros, and a lane whose tap
// falls outside its board reads zero block + (address of the row it would have read) mod 256.  (With ONE zero row
// shared by all padding lanes 94 of the 144 (row tile, tap) fragments of a 3-board group took 8 LDS cycles instead of
// 4: brute force over the hardware's lane groups; PMC: 37 % of the LDS-active cycles were bank conflicts.)  Row M takes
// the stores of rows >= M.  Residual image fp32 [row][16 x 16 B], slot XOR (row & 15).
#include "split_common.h"

namespace {

template <int S, int G>
struct W2Cfg {
    static constexpr int P = S * S, A = P + 1, M = G * P;
    static constexpr int MT = (M + 15) / 16;                      // 16-row tiles
    static constexpr int RTW = G >= 2 ? 4 : 2;                    // row tiles per wave
    static constexpr int NRQ = (MT + RTW - 1) / RTW;              // row groups
    static constexpr int CT = 2;                                  // 16-channel tiles per wave (one channel half)
    static constexpr int NW = 2 * NRQ, NTHR = NW * 64;
    static constexpr bool BIG = false;
    static constexpr int ZOFF = ((M + 1) * 64 + 255) & ~255;      // zero block of an image: 256 B, 256-byte aligned (row M: dump)
    static constexpr int IMG = ZOFF + 256;                        // (a multiple of 256: every image sees the same banks)
    static constexpr int ACT_BYTES = 4 * IMG;                     // image index = piece * 2 + kc
    static constexpr int CHUNK = 8192;                            // weights per k-chunk: [chh 2][piece 2][ct 2][lane][16 B]
    // the residual region serves three masters in turn: the input planes [G][6][P] fp32 while a group is staged, the
    // residual image during the tower, the policy features of the head phase (HQ)
    static constexpr int RES_OFF = (ACT_BYTES + 255) & ~255;
    static constexpr int STAGE = RES_OFF;
    static constexpr int RES_BYTES = (M + 1) * 256;
    static constexpr int HQ_OFF = RES_OFF;
    static constexpr int SS_OFF = RES_OFF + RES_BYTES;            // accumulator initial values [13][64] fp32
    static constexpr int HB_OFF = SS_OFF + 13 * 64 * 4;           // policy FC bias [A]
    static constexpr int VW_OFF = HB_OFF + ((A + 3) & ~3) * 4;    // value FC weights [3][P] + bias [3]
    static constexpr int HD1_OFF = (VW_OFF + ((3 * P + 3 + 3) & ~3) * 4 + 15) & ~15;   // 1x1 fragment image + table
    static constexpr int PIPE_BYTES = HD1_OFF + 4096 + 128;
    static constexpr int ROW_BYTES = kRowBytes;
    static constexpr int AUX = PIPE_BYTES;
    static constexpr int RING_OFF = (AUX + G * (3 * P + A + 4) * 4 + 255) & ~255;     // weight ring: two k-chunks
    static constexpr int LDS_BYTES = RING_OFF + 2 * CHUNK;
};

// ABL: timing-only ablations for tools/phase_profile.py (results are wrong): bit 0 = no activation-fragment reads in
// the loop, bit 1 = no weight-fragment loads in the loop, bit 2 = no MFMAs
template <int S, int G, int ABL = 0>
__global__ __launch_bounds__((W2Cfg<S, G>::NTHR), 2) void dualnet_fwd_w2_kernel(
    NetDev net, const float *__restrict__ planes, int batch, int want_logits,
    float *__restrict__ policy, float *__restrict__ value, int *__restrict__ overflow) {
    using C = W2Cfg<S, G>;
    using F = FmtF16;
    constexpr int P = C::P, M = C::M, RTW = C::RTW, CT = C::CT, NTHR = C::NTHR, IMG = C::IMG, NRQ = C::NRQ;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int rq = wave % NRQ, chh = wave / NRQ;

    // ---- per-lane geometry of this wave's row tiles ----
    int base_row[RTW];
    unsigned mask[RTW];                                   // bit t: tap t of this row is inside its board
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
        const int row = (rq * RTW + r) * 16 + li;
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
    // zero rows of the four activation images (the epilogues never touch row M)
    for (int e = tid; e < 4 * 64; e += NTHR)
        reinterpret_cast<unsigned *>(smem + (e >> 6) * IMG + C::ZOFF)[e & 63] = 0u;
    // tables -> LDS, once per workgroup
    for (int e = tid; e < 13 * 64; e += NTHR) reinterpret_cast<float *>(smem + C::SS_OFF)[e] = net.w2_init[e];
    for (int e = tid; e < C::A; e += NTHR) reinterpret_cast<float *>(smem + C::HB_OFF)[e] = net.pfc_b[e];
    stage_head_tables<C, NTHR>(smem, net, tid);
    for (int e = tid; e < 3 * P + 3; e += NTHR)
        reinterpret_cast<float *>(smem + C::VW_OFF)[e] = e < 3 * P ? net.vfc_w[e] : net.vfc_b[e - 3 * P];

    // weight stream: k-chunk gc of the whole network at ww2 + gc * CHUNK -> ring slot gc & 1; this wave's four
    // fragments (piece p, channel tile c) at chh * 4096 + (p * 2 + c) * 1024 + lane * 16 of the slot; its share of the
    // copy: bytes [wave * 1024, + 1024) of the chunk
    const int wv0 = C::RING_OFF + chh * 4096 + lane * 16;
    constexpr int kChunks = 2 * kSplitTaps;
    const unsigned dma_off = (unsigned)(wave * 1024 + lane * 16);
    auto dma_chunk = [&](int gc) __attribute__((always_inline)) {       // gc taken modulo the network: the ring runs on
        const int g2 = gc < kChunks ? gc : gc - kChunks;                 // into the next board group
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(net.ww2 + (size_t)g2 * C::CHUNK + dma_off),
                                         (__attribute__((address_space(3))) void *)(smem + C::RING_OFF + (gc & 1) * C::CHUNK + wave * 1024),
                                         16, 0, 0);
    };
    // publish: my share of every copy issued so far has landed, my fragment reads of the slot about to be refilled
    // are done; behind the barrier the other waves' shares have landed too
    auto ring_barrier = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0x0070);                 // vmcnt(0) lgkmcnt(0) - the builtin, so that hipcc's own counters
        asm volatile("" ::: "memory");                      // know (behind an asm wait it re-waited lgkmcnt(0) on the next
        __builtin_amdgcn_s_barrier();                       // chunk's first, freshly issued read)
        asm volatile("" ::: "memory");
    };
    dma_chunk(0);
    dma_chunk(1);

    // profiling stamps (tg_net_profile_phases), workgroup 0: wave 0 -> timeline[0..], the last wave -> timeline[64..]:
    // 0 group start, 1 input staged + split, per layer L (stem first) 2 + 3 L "MFMA loop done", 3 + 3 L "all waves
    // done" (barrier passed), 4 + 3 L "epilogue stored, barrier passed"; 41 heads done (44..46: head sub-phases)
    int stamp_i = 0;
    const bool stamper = net.timeline && blockIdx.x == 0 && (tid == 0 || tid == NTHR - 64);
    long long *const tl = net.timeline ? net.timeline + (tid == 0 ? 0 : 64) : nullptr;
    auto stamp = [&]() {
        if (stamper && stamp_i < 42) tl[stamp_i++] = (long long)__builtin_amdgcn_s_memtime();
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
    ring_barrier();                                       // chunks 0 and 1 of the weight stream are in the ring (and the tables above in LDS)
    fetch_planes(blockIdx.x);
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int b0 = grp * G;
        stamp();
        // ---- input planes -> LDS -> im2col'ed, split "layer -1" activations: k = 6 tap + plane, padded to 64 ----
        {
            float *st = reinterpret_cast<float *>(smem + C::STAGE);
            int stid = tid;
            asm volatile("" : "+v"(stid));
#pragma unroll
            for (int i = 0; i < NPL; ++i)
                if (stid + i * NTHR < G * 6 * P) st[stid + i * NTHR] = pre[i];
            __syncthreads();
            // two threads per position: thread parity = k-chunk
            for (int e = stid; e < 2 * M; e += NTHR) {
                const int row = e >> 1, kc = e & 1;
                const int bl = row / P, p = row - bl * P, y = p / S, x = p - y * S;
                const float *src = st + bl * 6 * P + p;
                const int swz = (row >> 1) & 3;
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {            // slot sl holds k = 32 kc + 8 sl .. + 7
                    f32x4 lo, hi;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int k = kc * 32 + sl * 8 + j, t = k / 6, c = k - t * 6;
                        const int dy = t / 3 - 1, dx = t % 3 - 1;
                        const bool ok = k < 54 && (unsigned)(y + dy) < (unsigned)S && (unsigned)(x + dx) < (unsigned)S;
                        const float v = ok ? src[c * P + dy * S + dx] : 0.f;
                        if (j < 4) lo[j] = v; else hi[j - 4] = v;
                    }
                    uint2 plo[2], phi[2];
                    split4<F>(lo, plo);
                    split4<F>(hi, phi);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        *reinterpret_cast<uint4 *>(smem + (q * 2 + kc) * IMG + row * 64 + ((sl ^ swz) << 4)) =
                            uint4{plo[q].x, plo[q].y, phi[q].x, phi[q].y};
                }
            }
        }
        __syncthreads();
        stamp();

        f32x4 acc[2][CT][RTW];
        i32x4v fa[2][CT][2];                               // [k-chunk parity][channel tile][piece]
        i32x4v fb[2][RTW][2];                              // [k-chunk parity][row tile][piece]
        i32x4v fbd[RTW][2];                                // (ablation 16 only)

        // B-fragment address of row tile r for tap `tap`
        auto row_addr = [&](int r, int tap, bool stem) __attribute__((always_inline)) {
            const int toff = stem ? 0 : (tap / 3 - 1) * S + (tap % 3 - 1);
            const bool ok = stem ? base_row[r] < M : ((mask[r] >> tap) & 1u) != 0;
            const int row = base_row[r] + toff;             // (may lie outside the image when !ok: only its banks matter)
            const int nat = row * 64 + ((lg ^ ((row >> 1) & 3)) << 4);
            return ok ? nat : C::ZOFF + (nat & 255);
        };
        auto load_b = [&](i32x4v &dst, auto P_, auto KC_, int addr) __attribute__((always_inline)) {
            constexpr int off = (decltype(P_)::value * 2 + decltype(KC_)::value) * IMG;
            lds_load_frag<off>(dst, smem, addr);
        };
        int wvg = wv0;
        asm volatile("" : "+v"(wvg));                      // opaque: one address register, slot and fragment in the immediate
        // weight fragments of a chunk with parity KC from its ring slot
        auto load_a = [&](i32x4v &dst, auto KC_, auto J_) __attribute__((always_inline)) {
            constexpr int off = decltype(KC_)::value * C::CHUNK + decltype(J_)::value * 1024;
            lds_load_frag<off>(dst, smem, wvg);
        };
        // chunk 0 of the network lies in slot 0 (first group: copied above; later groups: the previous group's last
        // chunks copied chunks 0 and 1 of the next pass), published by a ring barrier
        static_for<2 * CT>([&](auto J) {
            constexpr int j = decltype(J)::value;
            load_a(fa[0][j % CT][j / CT], std::integral_constant<int, 0>{}, J);
        });

        // One k-chunk gc (parity KC): copy of chunk gc + 2 into this chunk's slot (its fragments are in registers, every
        // wave's - ring barrier of the previous chunk), CT x RTW x 3 MFMAs with the weight fragments of chunk gc + 1
        // (other slot) and - unless this is a layer's last chunk - the activation fragments of the next chunk in
        // between, ring barrier.  ba: this tap's row addresses (for a KC = 0 chunk's successor), bn: the next tap's.
        auto chunk = [&](auto KC_, auto LASTC_, int gc, const int (&ba)[RTW], const int (&bn)[RTW]) __attribute__((always_inline)) {
            constexpr int kc = decltype(KC_)::value;
            constexpr bool lastc = decltype(LASTC_)::value;
            if constexpr (!(ABL & 2)) dma_chunk(gc + 2);
            constexpr int NMFMA = CT * RTW * F::NPROD;
            constexpr int NB = lastc ? 0 : RTW * 2, NA = CT * 2;
            // fragment reads: one behind each of the first NA + NB MFMAs (weights of the next chunk first - their slot was
            // published by the last barrier -, then the activations): nothing is still in flight at the ring barrier
            static_for<NMFMA>([&](auto M_) {
                constexpr int m = decltype(M_)::value;
                constexpr int q = m / (CT * RTW), c = (m / RTW) % CT, r = m % RTW;
                if constexpr (!(ABL & 4))
                    acc[F::PC[q]][c][r] = mfma16<F>(fa[kc][c][F::PA[q]], fb[kc][r][F::PB[q]], acc[F::PC[q]][c][r]);
                constexpr int jb0 = m < NA ? 0 : (m - NA < NB ? m - NA : NB), jb1 = m < NA ? 0 : (m + 1 - NA < NB ? m + 1 - NA : NB);
                if constexpr (jb1 > jb0 && !(ABL & 1)) {
                    static_for<jb1 - jb0>([&](auto D_) {
                        constexpr int jb = jb0 + decltype(D_)::value, r2 = jb % RTW, p2 = jb / RTW;
                        if constexpr (ABL & 16) {             // reads issued into a set no MFMA reads (no data dependence)
                            if constexpr (kc == 0) load_b(fbd[r2][p2], std::integral_constant<int, p2>{}, std::integral_constant<int, 1>{}, ba[r2]);
                            else load_b(fbd[r2][p2], std::integral_constant<int, p2>{}, std::integral_constant<int, 0>{}, bn[r2]);
                        } else if constexpr (kc == 0) load_b(fb[1][r2][p2], std::integral_constant<int, p2>{}, std::integral_constant<int, 1>{}, ba[r2]);
                        else load_b(fb[0][r2][p2], std::integral_constant<int, p2>{}, std::integral_constant<int, 0>{}, bn[r2]);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
                constexpr int ja0 = m < NA ? m : NA, ja1 = m < NA ? m + 1 : NA;
                if constexpr (ja1 > ja0 && !(ABL & 2)) {
                    load_a(fa[1 - kc][ja0 % CT][ja0 / CT], std::integral_constant<int, 1 - kc>{}, std::integral_constant<int, ja0>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ABL & 16) {
#pragma unroll
                for (int r = 0; r < RTW; ++r) { asm volatile("" :: "v"(fbd[r][0])); asm volatile("" :: "v"(fbd[r][1])); }
            }
            ring_barrier();
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using T = std::true_type;
        using N = std::false_type;

        int gc = 0;
#pragma unroll 1
        for (int layer = 0; layer <= kTowerLayers; ++layer) {
            const bool stem = layer == 0;
            // accumulator set 0 starts from the folded batch-norm shift (times the layer's weight scaling), set 1 from 0
            {
                int lz = lane;
                asm volatile("" : "+v"(lz));                // opaque: the table addresses are not hoisted out of the layer loop
                const int cb = C::SS_OFF + (layer * 64 + chh * 32 + (lz >> 4) * 4) * 4;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const f32x4 ini = *reinterpret_cast<const f32x4 *>(smem + cb + c * 64);
#pragma unroll
                    for (int r = 0; r < RTW; ++r) {
                        acc[0][c][r] = ini;
                        acc[1][c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            int ba[RTW], bn[RTW];
#pragma unroll
            for (int r = 0; r < RTW; ++r) ba[r] = row_addr(r, 0, stem);
            static_for<RTW * 2>([&](auto J) {
                constexpr int r = decltype(J)::value % RTW, p = decltype(J)::value / RTW;
                load_b(fb[0][r][p], std::integral_constant<int, p>{}, I0{}, ba[r]);
            });
            if (stem) {
                chunk(I0{}, N{}, gc, ba, ba);
                chunk(I1{}, T{}, gc + 1, ba, ba);
                gc += 2;
            } else {
#pragma unroll 1
                for (int t = 0; t < 8; ++t) {              // taps 0..7: two chunks each, activation fragments of the next tap
#pragma unroll
                    for (int r = 0; r < RTW; ++r) bn[r] = row_addr(r, t + 1, false);
                    chunk(I0{}, N{}, gc, ba, bn);
                    chunk(I1{}, N{}, gc + 1, ba, bn);
#pragma unroll
                    for (int r = 0; r < RTW; ++r) ba[r] = bn[r];
                    gc += 2;
                }
                chunk(I0{}, N{}, gc, ba, ba);              // tap 8
                chunk(I1{}, T{}, gc + 1, ba, ba);
                gc += 2;
            }
            // ---- epilogue: combine the accumulator sets, undo the weight scaling, (+ residual), ReLU, split, store ----
            stamp();                                      // (the last chunk's ring barrier: every wave is done reading the
            const float down = net.w2_down[layer];        // layer input)   2^-e of this layer's weights (uniform: scalar load)
            const float down_x = down * (1.f / 2048.f);
            stamp();
            float amax = 0.f;
            auto epilogue = [&](auto KEEP_, auto ADD_, auto LAST_) __attribute__((always_inline)) {
                constexpr bool keep = decltype(KEEP_)::value, add_res = decltype(ADD_)::value, last = decltype(LAST_)::value;
                int brow[RTW], rrow[RTW], wrow[RTW];
                f32x4 xres[CT][RTW];
#pragma unroll
                for (int r = 0; r < RTW; ++r) {
                    brow[r] = base_row[r];
                    asm volatile("" : "+v"(brow[r]));     // opaque: LDS addresses are recomputed, not hoisted + spilled
                    rrow[r] = brow[r] < M ? brow[r] : 0;
                    wrow[r] = brow[r] < M ? brow[r] : M;
                }
                const int c16 = chh * 8 + lg;               // 16-byte slot (4 channels) of channel tile c: c16 + 4 c
                if constexpr (add_res) {
#pragma unroll
                    for (int c = 0; c < CT; ++c)
#pragma unroll
                        for (int r = 0; r < RTW; ++r)
                            xres[c][r] = *reinterpret_cast<const f32x4 *>(smem + C::RES_OFF + rrow[r] * 256 +
                                                                          (((c16 + 4 * c) ^ (rrow[r] & 15)) << 4));
                }
#pragma unroll
                for (int c = 0; c < CT; ++c) {
#pragma unroll
                    for (int r = 0; r < RTW; ++r) {
                        const int row = wrow[r];
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float t = fmaf(acc[1][c][r][j], down_x, acc[0][c][r][j] * down);
                            if constexpr (add_res) t += xres[c][r][j];
                            v[j] = fmaxf(t, 0.f);
                        }
                        amax = fmaxf(fmaxf(amax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                        if constexpr (last) {
                            const int hrow = brow[r] < M ? brow[r] : M;
                            *reinterpret_cast<f32x4 *>(smem + hrow * kRowBytes + (chh * 32 + c * 16 + lg * 4) * 4) = v;
                        } else {
                            if constexpr (keep) {
                                const int krow = brow[r] < M ? brow[r] : M;
                                *reinterpret_cast<f32x4 *>(smem + C::RES_OFF + krow * 256 + (((c16 + 4 * c) ^ (krow & 15)) << 4)) = v;
                            }
                            uint2 pc[2];
                            split4<F>(v, pc);
                            const int slot = ((c << 1) | (lg >> 1)) ^ ((row >> 1) & 3);
                            const int off = row * 64 + slot * 16 + (lg & 1) * 8;
#pragma unroll
                            for (int q = 0; q < 2; ++q)
                                *reinterpret_cast<uint2 *>(smem + (q * 2 + chh) * IMG + off) = pc[q];
                        }
                    }
                }
            };
            if (layer == kTowerLayers) epilogue(N{}, T{}, N{});        // (the heads read the block output as activation images)
            else if (layer == 0) epilogue(T{}, N{}, N{});
            else if (layer & 1) epilogue(N{}, N{}, N{});
            else epilogue(T{}, T{}, N{});
            if (!(amax < 60000.f)) ovf = 1;                // f16 range guard (also catches NaN)
            __syncthreads();
            stamp();
        }
        fetch_planes(grp + gridDim.x);
        run_heads_mfma<S, G, C, NTHR>(smem, net, b0, batch, want_logits, policy, value, tid, wave,
                                       (net.timeline && blockIdx.x == 0 && grp == blockIdx.x) ? net.timeline + 44 : nullptr);
        __syncthreads();
        stamp();
        // the head scratch overlapped the activation images' zero rows
        for (int e = tid; e < 4 * 64; e += NTHR)
            reinterpret_cast<unsigned *>(smem + (e >> 6) * IMG + C::ZOFF)[e & 63] = 0u;
    }
    if (ovf && overflow) atomicOr(overflow, 1);
}

template <int S, int G, int ABL = 0>
int launch_w2(tg_net *net, const float *planes, int batch, int want_logits, float *policy, float *value,
              int *overflow, hipStream_t stream) {
    using C = W2Cfg<S, G>;
    static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(C::NRQ * C::RTW * 16 >= C::M, "row coverage");
    auto kern = dualnet_fwd_w2_kernel<S, G, ABL>;
    static bool attr_set[16] = {};
    if (!attr_set[net->device & 15]) {
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set[net->device & 15] = true;
    }
    const int groups = (batch + G - 1) / G;
    // TG_FWD_CUS=n: at most n workgroups (CUs) for the forward pass - leaves CUs to the tree kernels of another lock-step
    // group running on a second stream (the persistent workgroups of a full-width launch own every CU's LDS and registers)
    static const int cu_cap = getenv("TG_FWD_CUS") ? atoi(getenv("TG_FWD_CUS")) : 0;
    const int cus = cu_cap > 0 && cu_cap < net->num_cus ? cu_cap : net->num_cus;
    const int grid = groups < cus ? groups : cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHR), C::LDS_BYTES, stream, net->dev, planes, batch, want_logits,
                       policy, value, overflow);
    TG_HIP(hipGetLastError());
    return TG_OK;
}

}  // namespace

namespace tg {

// Weight image of the two-waves-per-SIMD kernel.  conv0: [64][6][3][3]; tower[l]: [64][64][3][3]; scale / shift: folded
// batch norm [13][64].  Per layer: w' = w * scale[cout] (fp64), scaled by a power of two 2^e so that the largest |w'|
// lies in [2^9, 2^10) (f16 pieces stay normal), split into two f16 pieces; accumulator start = shift * 2^e; 2^-e is
// applied in the epilogue.  Image: [k-chunk gc][chh 2][piece 2][ct 2][lane 64][8 x f16], gc = 2 * tap_g + kc; lane l of
// a fragment holds cout = 32 chh + 16 ct + (l & 15), k = 32 kc + 8 (l >> 4) + 0..7.
int w2_prepare(tg_net *net, const float *conv0, const float *const *tower, const float *scale, const float *shift) {
    std::vector<uint16_t> img((size_t)2 * kSplitTaps * 4096, 0);
    std::vector<float> init(13 * 64), down(16, 1.f);
    for (int layer = 0; layer <= kTowerLayers; ++layer) {
        const float *w = layer == 0 ? conv0 : tower[layer - 1];
        const int cin = layer == 0 ? 6 : 64;
        double mx = 0.0;
        for (int co = 0; co < 64; ++co)
            for (int i = 0; i < cin * 9; ++i)
                mx = std::fmax(mx, std::fabs((double)w[(size_t)co * cin * 9 + i] * (double)scale[layer * 64 + co]));
        int e = 0;
        if (mx > 0.0 && std::isfinite(mx)) {
            int ex;
            std::frexp(mx, &ex);
            e = 10 - ex;
        }
        if (e > 40) e = 40;                                   // (all-but-zero layers: keep 2^e and shift * 2^e finite)
        if (e < -40) e = -40;
        const double up = std::ldexp(1.0, e);
        down[layer] = std::ldexp(1.f, -e);
        for (int co = 0; co < 64; ++co) init[layer * 64 + co] = (float)((double)shift[layer * 64 + co] * up);
        const int ntaps = layer == 0 ? 1 : 9;
        for (int tap = 0; tap < ntaps; ++tap) {
            const int g = layer == 0 ? 0 : 1 + (layer - 1) * 9 + tap;
            for (int kc = 0; kc < 2; ++kc)
                for (int chh = 0; chh < 2; ++chh)
                    for (int ct = 0; ct < 2; ++ct)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int el = 0; el < 8; ++el) {
                                const int cout = chh * 32 + ct * 16 + (lane & 15), k = kc * 32 + (lane >> 4) * 8 + el;
                                double v;
                                if (layer == 0) {                  // k = tap' * 6 + plane
                                    const int t2 = k / 6, c2 = k % 6;
                                    v = k < 54 ? (double)conv0[(cout * 6 + c2) * 9 + t2] : 0.0;
                                } else {
                                    v = (double)w[((size_t)cout * 64 + k) * 9 + tap];
                                }
                                v *= (double)scale[layer * 64 + cout] * up;
                                // pieces of the fp64 product: the second piece takes what the first left of the exact value
                                uint16_t pc[2];
                                pc[0] = f32_to_f16_rn((float)v);
                                pc[1] = f32_to_f16_rn((float)((v - (double)f16_to_f32(pc[0])) * 2048.0));
                                for (int p = 0; p < 2; ++p)
                                    img[(((((size_t)g * 2 + kc) * 2 + chh) * 2 + p) * 2 + ct) * 512 + lane * 8 + el] = pc[p];
                            }
        }
    }
    void *d = nullptr;
    TG_HIP(hipMalloc(&d, img.size() * 2));
    net->allocs.push_back(d);
    TG_HIP(hipMemcpy(d, img.data(), img.size() * 2, hipMemcpyHostToDevice));
    net->dev.ww2 = static_cast<const unsigned char *>(d);
    void *di = nullptr;
    TG_HIP(hipMalloc(&di, init.size() * 4));
    net->allocs.push_back(di);
    TG_HIP(hipMemcpy(di, init.data(), init.size() * 4, hipMemcpyHostToDevice));
    net->dev.w2_init = static_cast<const float *>(di);
    void *dd = nullptr;
    TG_HIP(hipMalloc(&dd, down.size() * 4));
    net->allocs.push_back(dd);
    TG_HIP(hipMemcpy(dd, down.data(), down.size() * 4, hipMemcpyHostToDevice));
    net->dev.w2_down = static_cast<const float *>(dd);
    return TG_OK;
}

int w2_forward(tg_net *net, const float *planes, int batch, int want_logits, float *policy, float *value,
               int *overflow, hipStream_t stream) {
    if (net->board_size != 9) return tg::fail(TG_ERR_ARG, "w2 forward: 9x9 only");
    // timing-only ablations (WRONG RESULTS: see the kernel) - only with TG_ALLOW_WRONG_RESULTS=1 next to TG_W2_ABL, and never
    // silently: a stray environment variable must not turn the production path into a profiling experiment
    const char *abl = getenv("TG_W2_ABL");
    if (abl && !(getenv("TG_ALLOW_WRONG_RESULTS") && atoi(getenv("TG_ALLOW_WRONG_RESULTS")) == 1))
        return tg::fail(TG_ERR_ARG, "TG_W2_ABL selects timing-only kernel ablations with wrong results: set TG_ALLOW_WRONG_RESULTS=1 as well, or unset it");
    if (const char *env = abl) {
        static bool warned = false;
        if (!warned) {
            fprintf(stderr, "[tamago_hip] TG_W2_ABL=%s: the w2 forward kernel runs a timing-only ablation - its results are WRONG\n", env);
            warned = true;
        }
        switch (atoi(env)) {
        case 1: return launch_w2<9, 3, 1>(net, planes, batch, want_logits, policy, value, overflow, stream);
        case 2: return launch_w2<9, 3, 2>(net, planes, batch, want_logits, policy, value, overflow, stream);
        case 3: return launch_w2<9, 3, 3>(net, planes, batch, want_logits, policy, value, overflow, stream);
        case 4: return launch_w2<9, 3, 4>(net, planes, batch, want_logits, policy, value, overflow, stream);
        case 18: return launch_w2<9, 3, 18>(net, planes, batch, want_logits, policy, value, overflow, stream);
        default: break;
        }
    }
    return launch_w2<9, 3>(net, planes, batch, want_logits, policy, value, overflow, stream);
}

}  // namespace tg
