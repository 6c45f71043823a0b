// 19x19 DualNet forward with ONE BOARD SPREAD OVER NB WORKGROUPS (compute units) - the latency kernel of small 19x19
// launches (one tree's mini-batch of 64 leaves occupies 64 of 256 CUs with dualnet_fwd_split_kernel<19, 1>: 217 us a pass).
//
// Same arithmetic as net_forward_split.hip (f16 x 2 operand pieces, two accumulator sets, the same k-chunk order per
// output, the same epilogues and heads): a position's result is bit-identical to the one-workgroup kernel's.  What is
// different:
//   * workgroup (slot, band) owns the board rows [y0, y1) of board `slot`: its MFMA rows are the band's cells only
//     (NB = 2: 190 cells = 12 row tiles, three per wave; NB = 4: 95 cells = 6 row tiles), its activation images hold the
//     band plus one halo row above and below (local row = band cell + 19: tap offsets stay +-19 +-1);
//   * the residual of a band fits into LDS (48 KB at NB = 2): no residual round trips through the scratch image;
//   * after every epilogue but the last a band publishes its two edge rows (19 cells x 256 B each) in a double-buffered
//     exchange area of the per-stream scratch, releases a sequence number at agent scope and waits for its neighbours':
//     one L2 round trip of 5 KB per layer and side.  Parity by layer: a neighbour has read layer L's rows before it
//     publishes layer L + 1, which this band waits for before it writes layer L + 2's rows over layer L's;
//   * the last epilogue writes the fp32 block output to the gather image of the slot; band 0 waits for all bands, copies the
//     image into LDS in the heads' layout and runs the unchanged head code (run_heads_split) while the other bands start on
//     their next board.
// All workgroups of a launch must be resident at once (grid = NB x boards <= CUs, one workgroup per CU by LDS): the
// launcher only takes this kernel for such batches.  Waits are bounded: a band whose neighbour does not show up (CUs held
// by another stream's kernels for longer than the limit) raises the range flag - the exact-fp32 kernel queued behind
// every split launch then redoes the batch - and stops waiting; no launch can hang.  A network's banded launches follow each
// other even when they come from different streams (an event per network): two of them half-resident would wait for bands
// that the other one keeps off the CUs.
// Reference: nn/network/dual_net.py:41-52, nn/network/res_block.py:8-38 at BOARD_SIZE = 19 (board/constant.py:4).
#include "split_common.h"

namespace {

constexpr int kBandSpinLimit = 1 << 17;                    // polls of a neighbour's sequence number (~ 0.1 s) before giving up

// What crosses workgroups goes through memory with agent-scope accesses (sc1), 16 bytes at a time, and explicit waits - no
// release / acquire fences: on gfx950 those write back and invalidate a whole L2 (buffer_wbl2 / buffer_inv), once per layer
// and workgroup here, and the weight stream of every workgroup on the XCD would come from HBM again each time (measured
// with fences: 559 us per 64-board pass against 212 us for the one-workgroup kernel).
__device__ __forceinline__ void store16_agent(float *p, const i32x4v v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
// N (<= 3) loads in flight, one wait: the values are valid when the statement is done (hipcc assumes that of any asm output)
__device__ __forceinline__ void load16x3_agent(const float *p0, const float *p1, const float *p2, i32x4v &v0, i32x4v &v1, i32x4v &v2) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off sc1\n\t"
                 "global_load_dwordx4 %2, %5, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}
__device__ __forceinline__ void load16x8_agent(const float *p, int stride_floats, i32x4v (&v)[8]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\t"
                 "global_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\t"
                 "global_load_dwordx4 %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p), "v"(p + stride_floats), "v"(p + 2 * stride_floats), "v"(p + 3 * stride_floats), "v"(p + 4 * stride_floats),
                   "v"(p + 5 * stride_floats), "v"(p + 6 * stride_floats), "v"(p + 7 * stride_floats)
                 : "memory");
}

template <int NB>
struct BandCfg {
    static constexpr int S = 19, P = S * S, A = P + 1, M = P, G = 1;
    static constexpr int MT = (M + 15) / 16;               // row tiles of the whole board (heads)
    static constexpr bool BIG = true;
    static constexpr int RB = (S + NB - 1) / NB;           // board rows per band (the last band may have fewer)
    static constexpr int MB = RB * S;                      // cells per band (max)
    static constexpr int MTB = (MB + 15) / 16;
    static constexpr int NW = 4, NTHR = 256;               // four waves as in the one-workgroup kernel (the heads split K by wave)
    static constexpr int RTW = (MTB + NW - 1) / NW;
    static constexpr int ML = (RB + 2) * S;                // local image rows: halo row, band, halo row; row ML = dump row
    static constexpr int ZOFF = ((ML + 1) * 64 + 255) & ~255;
    static constexpr int IMG = ZOFF + 256;
    static constexpr int ACT_BYTES = 4 * IMG;              // image index = piece * 2 + kc
    static constexpr int CHUNK = 2 * 4 * 1024;
    static constexpr int RES_OFF = ACT_BYTES;              // residual fp32 [band cell][16 x 16 B], row MB = dump row
    static constexpr int RES_BYTES = (MB + 1) * 256;
    static constexpr int STAGE = RES_OFF;                  // input planes [6][P] fp32 (before the stem's epilogue writes the residual)
    static constexpr int HEAD_IMG = ((M + 16) * kRowBytes + 255) & ~255;   // fp32 [row][72] image of the whole board (band 0, heads)
    static constexpr int TAB = RES_OFF + RES_BYTES > HEAD_IMG ? RES_OFF + RES_BYTES : HEAD_IMG;
    static constexpr int SS_OFF = TAB;                     // folded BN scale [13][64] + shift [13][64]
    static constexpr int HW_OFF = SS_OFF + 2 * 13 * 64 * 4;
    static constexpr int HB_OFF = HW_OFF + 64 * 4 * 4;
    static constexpr int HS_OFF = HB_OFF + ((A + 3) & ~3) * 4;
    static constexpr int VW_OFF = HS_OFF + 8 * 4;
    static constexpr int MISC = (VW_OFF + ((3 * P + 3 + 3) & ~3) * 4 + 15) & ~15;   // [0] "a neighbour did not show up"
    static constexpr int AUX = MISC + 16;
    static constexpr int ROW_BYTES = kRowBytes;
    static constexpr int LDS_BYTES = AUX + G * (3 * P + A + 4) * 4 + 256;
    // per-slot areas of the per-stream scratch (floats): gather image [P][64] fp32, exchange [band][parity 2][side 2][S][64]
    static constexpr int HBUF = 0, XBUF = P * 64, SLOT_FLOATS = 2 * P * 64;
    static_assert(STAGE + 6 * P * 4 <= RES_OFF + RES_BYTES, "plane staging");
    static_assert(XBUF + NB * 4 * S * 64 <= SLOT_FLOATS, "exchange area");
    static_assert(LDS_BYTES <= 163840, "LDS");
};

template <int NB>
__global__ __launch_bounds__(256, 1) void dualnet_fwd_band_kernel(
    NetDev net, const float *__restrict__ planes, int batch, int want_logits,
    float *__restrict__ policy, float *__restrict__ value, int *__restrict__ overflow, int *__restrict__ flags, int mute_band) {
    using C = BandCfg<NB>;
    using F = FmtF16;
    constexpr int S = C::S, P = C::P, RTW = C::RTW, NTHR = C::NTHR, NP = F::NP, IMG = C::IMG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
    // band-major numbering: the bands of a slot are n_slots workgroups apart (the same XCD when n_slots is a multiple of 8)
    const int n_slots = gridDim.x / NB;
    const int band = blockIdx.x / n_slots, slot = blockIdx.x - band * n_slots;
    const int y0 = band * C::RB, y1 = y0 + C::RB < S ? y0 + C::RB : S;
    const int ncells = (y1 - y0) * S;
    float *const slot_mem = net.scratch + (size_t)slot * C::SLOT_FLOATS;
    float *const hbuf = slot_mem + C::HBUF;
    auto xarea = [&](int b, int par, int side) { return slot_mem + C::XBUF + (size_t)(((b * 2 + par) * 2 + side) * S) * 64; };
    int *const xflag = flags + slot * NB;                  // exchange sequence numbers [slot][band]
    int *const gflag = flags + n_slots * NB + slot * NB;   // gather sequence numbers [slot][band]
    int *const dead = reinterpret_cast<int *>(smem + C::MISC);

    // ---- per-lane geometry of this wave's row tiles: band cell, local image row, taps inside the board ----
    int base_row[RTW];                                     // local image row (band cell + S)
    unsigned mask[RTW];                                    // bit t: tap t inside the board; bit 9: a cell of this band
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
        const int cell = (wave * RTW + r) * 16 + li;
        const int y = y0 + cell / S, x = cell % S;
        unsigned m = 0;
        if (cell < ncells) {
            m = 1u << 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < S && xx >= 0 && xx < S) m |= 1u << t;
            }
        }
        mask[r] = m;
        base_row[r] = cell + S;
    }
    for (int e = tid; e < NP * 2 * 64; e += NTHR)
        reinterpret_cast<unsigned *>(smem + (e >> 6) * IMG + C::ZOFF)[e & 63] = 0u;
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
    if (tid == 0) *dead = 0;
    const int wv0 = lane * 16;
    constexpr int kChunks = 2 * kSplitTaps;

    int ovf = 0;
    constexpr int NPL = (6 * P + NTHR - 1) / NTHR;
    float pre[NPL];
    auto fetch_planes = [&](int grp2) __attribute__((always_inline)) {
        int ft = tid;
        asm volatile("" : "+v"(ft));
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int e = ft + i * NTHR;
            pre[i] = (e < 6 * P && grp2 < batch) ? __builtin_nontemporal_load(&planes[(size_t)grp2 * 6 * P + e]) : 0.f;
        }
    };
    // wait until workgroup-shared sequence number *f has reached `want` (one lane polls; bounded)
    auto wait_for = [&](int *f, int want) {
        if (*reinterpret_cast<volatile int *>(dead)) return;
        int n = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(2);
            if (++n > kBandSpinLimit) {
                *reinterpret_cast<volatile int *>(dead) = 1;
                if (net.band_timeouts) atomicAdd(net.band_timeouts, 1u);   // (host-mapped: tg_net_band_timeouts, band_count)
                break;
            }
        }
    };
    fetch_planes(slot);
    int kiter = 0;
    for (int grp = slot; grp < batch; grp += n_slots, ++kiter) {
        // ---- input planes of the whole board -> LDS -> im2col'ed, split "layer -1" activations of this band's cells ----
        {
            float *st = reinterpret_cast<float *>(smem + C::STAGE);
            int stid = tid;
            asm volatile("" : "+v"(stid));
#pragma unroll
            for (int i = 0; i < NPL; ++i)
                if (stid + i * NTHR < 6 * P) st[stid + i * NTHR] = pre[i];
            __syncthreads();
            for (int cell = stid; cell < ncells; cell += NTHR) {
                const int yb = cell / S, x = cell - yb * S, y = y0 + yb;
                const float *src = st + y * S + x;
                const int row = cell + S;
                const int swz = (row >> 1) & 3;
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {
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
                    const int kc = sl >> 2, sslot = (sl & 3) ^ swz;
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        *reinterpret_cast<uint4 *>(smem + (q * 2 + kc) * IMG + row * 64 + sslot * 16) =
                            uint4{plo[q].x, plo[q].y, phi[q].x, phi[q].y};
                }
            }
        }
        __syncthreads();

        f32x4 acc[F::NACC][4][RTW];
        constexpr int NASET = 3, ADIST = NASET - 1;
        i32x4v fa[NASET][4][NP], fb[2][RTW][NP];
        auto row_addr = [&](int r, int tap, bool stem) __attribute__((always_inline)) {
            const int toff = stem ? 0 : (tap / 3 - 1) * S + (tap % 3 - 1);
            const bool ok = stem ? ((mask[r] >> 9) & 1u) != 0 : ((mask[r] >> tap) & 1u) != 0;
            const int row = base_row[r] + toff;
            const int nat = row * 64 + ((lg ^ ((row >> 1) & 3)) << 4);
            return ok ? nat : C::ZOFF + (nat & 255);
        };
        auto load_b = [&](i32x4v &dst, auto P_, auto KC_, int addr) __attribute__((always_inline)) {
            constexpr int off = (decltype(P_)::value * 2 + decltype(KC_)::value) * IMG;
            lds_load_frag<off>(dst, smem, addr);
        };
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
        load_a_all(std::integral_constant<int, 1>{}, 0);
        load_a_all(std::integral_constant<int, 2>{}, 1);
        // one k-chunk (as net_forward_split.hip: activation fragments one chunk ahead from LDS, weights two ahead from L2)
        auto chunk = [&](auto KC_, auto ASET_, int gc, const int (&ba)[RTW], const int (&bn)[RTW]) __attribute__((always_inline)) {
            constexpr int kc = decltype(KC_)::value, aset = decltype(ASET_)::value % NASET, anext = (aset + ADIST) % NASET;
            const unsigned char *wnext = net.wsplit + (size_t)(gc + ADIST < kChunks ? gc + ADIST : kChunks - 1) * C::CHUNK;
            constexpr int NMFMA = 4 * RTW * F::NPROD;
            constexpr int NBF = RTW * NP, NA = 4 * NP;
            constexpr int BSPAN = NMFMA * 6 / 16;
            static_for<NMFMA>([&](auto M_) {
                constexpr int m = decltype(M_)::value;
                constexpr int q = m / (4 * RTW), c = (m / RTW) % 4, r = m % RTW;
                acc[F::PC[q]][c][r] = mfma16<F>(fa[aset][c][F::PA[q]], fb[kc][r][F::PB[q]], acc[F::PC[q]][c][r]);
                constexpr int jb0 = m * NBF / BSPAN, jb1 = (m + 1) * NBF / BSPAN < NBF ? (m + 1) * NBF / BSPAN : NBF;
                if constexpr (jb1 > jb0) {
                    static_for<jb1 - jb0>([&](auto D_) {
                        constexpr int jb = jb0 + decltype(D_)::value, r2 = jb % RTW, p2 = jb / RTW;
                        if constexpr (kc == 0) load_b(fb[1][r2][p2], std::integral_constant<int, p2>{}, std::integral_constant<int, 1>{}, ba[r2]);
                        else load_b(fb[0][r2][p2], std::integral_constant<int, p2>{}, std::integral_constant<int, 0>{}, bn[r2]);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
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

        int gc = 0;
#pragma unroll 1
        for (int layer = 0; layer <= kTowerLayers; ++layer) {
            const bool stem = layer == 0;
#pragma unroll
            for (int s = 0; s < F::NACC; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < RTW; ++r) acc[s][c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
            int ba[RTW], bn[RTW];
#pragma unroll
            for (int r = 0; r < RTW; ++r) ba[r] = row_addr(r, 0, stem);
            static_for<RTW * NP>([&](auto J) {
                constexpr int r = decltype(J)::value % RTW, p = decltype(J)::value / RTW;
                load_b(fb[0][r][p], std::integral_constant<int, p>{}, I0{}, ba[r]);
            });
            if (stem) {
                chunk(I0{}, I1{}, gc, ba, ba);
                chunk(I1{}, I2{}, gc + 1, ba, ba);
                gc += 2;
            } else {
#pragma unroll 1
                for (int t3 = 0; t3 < 9; t3 += 3) {
#pragma unroll
                    for (int r = 0; r < RTW; ++r) bn[r] = row_addr(r, t3 + 1, false);
                    chunk(I0{}, I0{}, gc, ba, bn);
                    chunk(I1{}, I1{}, gc + 1, ba, bn);
#pragma unroll
                    for (int r = 0; r < RTW; ++r) { ba[r] = bn[r]; bn[r] = row_addr(r, t3 + 2, false); }
                    chunk(I0{}, I2{}, gc + 2, ba, bn);
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
            // ---- epilogue: BN scale / shift (+ residual) + ReLU; split into the images, or (last layer) fp32 to the gather image ----
            __syncthreads();
            float amax = 0.f;
            auto epilogue = [&](auto KEEP_, auto ADD_, auto LAST_) __attribute__((always_inline)) {
                constexpr bool keep = decltype(KEEP_)::value, add_res = decltype(ADD_)::value, last = decltype(LAST_)::value;
                f32x4 xres[4][RTW], sc[4], sh[4];
                int brow[RTW], cellr[RTW];
                bool valid[RTW];
#pragma unroll
                for (int r = 0; r < RTW; ++r) {
                    brow[r] = base_row[r];
                    asm volatile("" : "+v"(brow[r]));
                    valid[r] = brow[r] - S < ncells;
                    cellr[r] = valid[r] ? brow[r] - S : C::MB;            // residual row (MB = dump row)
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    sc[c] = *reinterpret_cast<const f32x4 *>(smem + C::SS_OFF + (layer * 64 + c * 16 + lg * 4) * 4);
                    sh[c] = *reinterpret_cast<const f32x4 *>(smem + C::SS_OFF + (13 * 64 + layer * 64 + c * 16 + lg * 4) * 4);
                    if constexpr (add_res) {
#pragma unroll
                        for (int r = 0; r < RTW; ++r)
                            xres[c][r] = *reinterpret_cast<const f32x4 *>(smem + C::RES_OFF + cellr[r] * 256 + (((c * 4 + lg) ^ (cellr[r] & 15)) << 4));
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int r = 0; r < RTW; ++r) {
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float t = fmaf(acc[1][c][r][j], 1.f / 2048.f, acc[0][c][r][j]);
                            t = fmaf(t, sc[c][j], sh[c][j]);
                            if constexpr (add_res) t += xres[c][r][j];
                            v[j] = fmaxf(t, 0.f);
                        }
                        if (valid[r]) amax = fmaxf(fmaxf(amax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                        if constexpr (last) {
                            if (valid[r])
                                store16_agent(hbuf + (size_t)(y0 * S + brow[r] - S) * 64 + c * 16 + lg * 4, __builtin_bit_cast(i32x4v, v));
                        } else {
                            if constexpr (keep)
                                *reinterpret_cast<f32x4 *>(smem + C::RES_OFF + cellr[r] * 256 + (((c * 4 + lg) ^ (cellr[r] & 15)) << 4)) = v;
                            uint2 pc[NP];
                            split4<F>(v, pc);
                            const int row = valid[r] ? brow[r] : C::ML;
                            const int sslot = (((c & 1) << 1) | (lg >> 1)) ^ ((row >> 1) & 3);
                            const int off = row * 64 + sslot * 16 + (lg & 1) * 8;
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
            if (!(amax < 60000.f)) ovf = 1;
            if (layer < kTowerLayers) {
                // ---- halo exchange: publish this band's edge rows, wait for the neighbours', fetch theirs into the halo rows ----
                __syncthreads();                              // the images are complete
                const int par = layer & 1, seq = kiter * 16 + layer + 1;
                int xt = tid;
                asm volatile("" : "+v"(xt));
                // item e of the 2 sides x 19 cells x 4 images x 4 chunks (three per thread at most): side, cell, image, chunk
                auto item = [&](int e, int &side, int &cell, int &img, int &q) {
                    side = e / (S * 16);
                    const int rem = e - side * (S * 16);
                    cell = rem >> 4; img = (rem >> 2) & 3; q = rem & 3;
                    return e < 2 * S * 16 && (side == 0 ? band > 0 : band < NB - 1);
                };
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    int side, cell, img, q;
                    if (item(xt + i * NTHR, side, cell, img, q)) {
                        const int lrow = side == 0 ? S + cell : ncells + cell;
                        const i32x4v v = *reinterpret_cast<const i32x4v *>(smem + img * IMG + lrow * 64 + ((q ^ ((lrow >> 1) & 3)) << 4));
                        store16_agent(xarea(band, par, side) + cell * 64 + img * 16 + q * 4, v);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the stores have been acknowledged
                __syncthreads();
                // (mute_band: test hook - that band never announces its rows, its neighbours run into the bounded wait)
                if (xt == 0 && band != mute_band) __hip_atomic_store(&xflag[band], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (xt == 64 && band > 0) wait_for(&xflag[band - 1], seq);
                if (xt == 128 && band < NB - 1) wait_for(&xflag[band + 1], seq);
                __syncthreads();
                {
                    // my upper halo = the bottom edge (side 1) of the band above, my lower halo = the top edge of the band below
                    const float *src[3];
                    int dst[3];
                    bool on[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        int side, cell, img, q;
                        on[i] = item(xt + i * NTHR, side, cell, img, q);
                        const int nbr = side == 0 ? (band > 0 ? band - 1 : band) : (band < NB - 1 ? band + 1 : band);
                        src[i] = xarea(on[i] ? nbr : band, par, on[i] ? 1 - side : 0) + (on[i] ? cell * 64 + img * 16 + q * 4 : 0);
                        const int lrow = side == 0 ? cell : S + ncells + cell;
                        dst[i] = img * IMG + lrow * 64 + ((q ^ ((lrow >> 1) & 3)) << 4);
                    }
                    i32x4v v0, v1, v2;
                    load16x3_agent(src[0], src[1], src[2], v0, v1, v2);
                    if (on[0]) *reinterpret_cast<i32x4v *>(smem + dst[0]) = v0;
                    if (on[1]) *reinterpret_cast<i32x4v *>(smem + dst[1]) = v1;
                    if (on[2]) *reinterpret_cast<i32x4v *>(smem + dst[2]) = v2;
                }
            }
            __syncthreads();
        }
        if (*reinterpret_cast<volatile int *>(dead)) ovf = 1;
        fetch_planes(grp + n_slots);
        // ---- gather: every band's block output is in the slot's gather image; band 0 runs the heads ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the gather image's stores have been acknowledged
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&gflag[band], kiter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (band == 0) {
            if ((tid & 63) == 0 && (tid >> 6) < NB) wait_for(&gflag[tid >> 6], kiter + 1);
            __syncthreads();
            int ht = tid;
            asm volatile("" : "+v"(ht));
            // [P][64] floats = P x 16 chunks of 16 bytes; thread t takes the chunks t, t + 256, ...: eight in flight at a time
            for (int e0 = ht; e0 < P * 16; e0 += 8 * NTHR) {
                i32x4v v[8];
                load16x8_agent(hbuf + (size_t)e0 * 4, NTHR * 4, v);          // (chunks beyond the image: inside the slot's scratch, unused)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = e0 + i * NTHR;
                    if (e < P * 16) *reinterpret_cast<i32x4v *>(smem + (e >> 4) * kRowBytes + (e & 15) * 16) = v[i];
                }
            }
            __syncthreads();
            if (*reinterpret_cast<volatile int *>(dead)) ovf = 1;
            run_heads_split<S, 1, C, NTHR>(smem, net, grp, batch, want_logits, policy, value, tid, wave, nullptr);
            __syncthreads();
            for (int e = tid; e < NP * 2 * 64; e += NTHR)          // the head scratch overlapped the images' zero blocks
                reinterpret_cast<unsigned *>(smem + (e >> 6) * IMG + C::ZOFF)[e & 63] = 0u;
        }
    }
    if (ovf && overflow) atomicOr(overflow, 1);
}

template <int NB>
int launch_band(tg_net *net, const float *planes, int batch, int want_logits, float *policy, float *value,
                int *overflow, int *flags, hipStream_t stream) {
    using C = BandCfg<NB>;
    auto kern = dualnet_fwd_band_kernel<NB>;
    static std::atomic<uint64_t> configured{0};
    if (tg::first_on_device(configured, net->device))
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    NetDev dev = net->dev;
    // TG_BAND_TEST_MUTE=b: band b keeps its sequence number to itself (tests/test_gpu_net.py: the bounded waits end the launch,
    // the exact kernel redoes the batch)
    const char *mute_env = tg::knob("TG_BAND_TEST_MUTE");
    const int mute_band = mute_env ? atoi(mute_env) : -1;
    {
        static_assert((size_t)C::SLOT_FLOATS == (size_t)2 * C::P * 64, "one slot = one workgroup's share of the scratch");
        std::lock_guard<std::mutex> lock(net->scratch_mu);
        float *&slot = net->scratch_by_stream[stream];
        if (!slot) {
            void *d = nullptr;
            TG_HIP(hipMalloc(&d, net->scratch_floats * sizeof(float)));
            slot = static_cast<float *>(d);
        }
        dev.scratch = slot;
        // one banded launch at a time on the device: when the launch stream changes, the new stream waits for what the previous
        // one has queued (an event recorded there now: launches that stay on one stream - a search - pay nothing)
        if (net->band_recorded && net->band_stream != stream) {
            if (!net->band_done) TG_HIP(hipEventCreateWithFlags(&net->band_done, hipEventDisableTiming));
            if (hipEventRecord(net->band_done, net->band_stream) == hipSuccess)
                TG_HIP(hipStreamWaitEvent(stream, net->band_done, 0));
            else
                (void)hipGetLastError();                     // (the previous stream is gone: nothing of it can be in flight)
        }
        hipLaunchKernelGGL(kern, dim3(batch * NB), dim3(C::NTHR), C::LDS_BYTES, stream, dev, planes, batch, want_logits,
                           policy, value, overflow, flags, mute_band);
        TG_HIP(hipGetLastError());
        net->band_stream = stream;
        net->band_recorded = true;
    }
    return TG_OK;
}

}  // namespace

namespace tg {

// bands per board the banded kernel would use for this batch (0: the batch is too large for it - every workgroup of a
// launch must be resident at once - or TG_FWD_BANDS=0 switches it off; TG_FWD_BANDS=2 / 4 force a split)
int band_count(const tg_net *net, int batch) {
    const char *env = tg::knob("TG_FWD_BANDS");
    const int forced = env ? atoi(env) : -1;
    if (net->board_size != 19 || forced == 0) return 0;
    if (tg::launch_caps().forward > 0) return 0;               // CUs are held back for other streams' kernels
    // A device shared with other PROCESSES (more self-play shards than GPUs, TG_SINGLE_DEVICE): their kernels can keep bands
    // off the CUs for longer than the bounded waits - results stay right (the exact kernel redoes the batch) but every such
    // launch costs 0.1 s.  Announced (tg_net_set_shared_device) or found out (a first bounded wait gave up): stay on the
    // one-workgroup kernel.  A forced TG_FWD_BANDS still wins (tests).
    if (forced < 0 && (net->shared_device || (net->band_timeouts_host && *net->band_timeouts_host > 0))) return 0;
    // (Two banded launches whose workgroups do not all fit on the device could hold each other's missing bands off the CUs until
    // the bounded waits give up: launch_band lets a network's banded launches follow each other across streams.)  With the
    // sub-group streams of a self-play move in flight (tg::launch_caps().guard is set exactly then) a launch takes a quarter of the CUs
    // and leaves the rest to the other sub-groups' tree kernels.
    const int cus = tg::launch_caps().guard > 0 ? net->num_cus / 4 : net->num_cus;
    if ((forced == 4 || forced < 0) && batch * 4 <= cus) return 4;
    if ((forced == 2 || forced == 4 || forced < 0) && batch * 2 <= cus) return 2;
    return 0;
}

// flags: kBandFlagInts zeroed ints behind the range flag (sequence numbers of the exchange and of the gather)
int band_forward(tg_net *net, int bands, const float *planes, int batch, int want_logits, float *policy, float *value,
                 int *overflow, int *flags, hipStream_t stream) {
    if (bands == 4) return launch_band<4>(net, planes, batch, want_logits, policy, value, overflow, flags, stream);
    if (bands == 2) return launch_band<2>(net, planes, batch, want_logits, policy, value, overflow, flags, stream);
    return tg::fail(TG_ERR_ARG, "band forward: 2 or 4 bands");
}

}  // namespace tg
