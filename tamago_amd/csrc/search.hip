// Device-resident batched MCTS for gfx950 (MI355X).
//
// Replaces, for T independent search trees driven in lock-step:
//   mcts/tree.py:199-244   search_mcts        (PUCT descent, virtual loss, leaf queueing)
//   mcts/tree.py:247-270   expand_node        (+ :509-519 Dirichlet tentative prior)
//   mcts/tree.py:273-315   process_mini_batch (policy write-back, value backup)
//   mcts/tree.py:387-422   search_sequential_halving (Gumbel descent)
//   mcts/node.py:41-157, 275-361, mcts/pucb/pucb.py:8-29 (node math)
//   mcts/batch_data.py:7-34 (the leaf queue)
//   board/go_board.py:131-185, 237-304, 327-409 as called from the search (put_stone,
//   legality incl. positional superko, self-atari, complete-eye filter)
//   nn/feature.py:10-57 for the leaf boards.
//
// One workgroup owns one tree; inside it every WAVEFRONT has a role of its own (selector, board worker, value or
// policy backup) and works alone on its data - the one-wavefront kernels are the reference forms of the pipelined
// ones (select_puct_pipe / select_puct_mpipe / select_gumbel_pipe).  The node pool is a
// structure-of-arrays in HBM ([tree][node][child] per field); PUCB / Gumbel selection is
// a 64-lane arg-max reduction with lowest-index tie-break, in float64 exactly as numpy
// evaluates it; expansion tests all on-board points in parallel (one lane per point) on
// a board held in LDS; backup walks the recorded path (or parent links).  Boards use an O(1)-state
// representation (cell colour + string id = smallest stone coordinate); liberties,
// sizes and string hashes are recomputed with LDS atomics when a node is expanded, so
// nothing has to be copied per descent except 3 bytes per cell.
//
// Bit-exactness notes (verified against the oracle in tests/test_gpu_search.py):
//   * leaf values are float32; children_value_sum is accumulated in FLOAT32 and stored as
//     float64, node_value_sum is float32, "1 - v" is float32 (what numpy/torch do in
//     mcts/tree.py:297-313);
//   * PUCB is float64 with IEEE division and sqrt; this file is compiled with
//     -ffp-contract=off so no FMA is formed behind the source's back;
//   * the random draws are generated on the device (csrc/legacy_rng_device.h: MT19937 + glibc's log restated) or come from
//     the caller through tg_search_set_rng.
#include "common.h"
#define TG_RNG_DEVICE_KERNELS
#include "legacy_rng_device.h"

namespace tg {                                                        // net_forward.hip: per-thread grid caps of the forward launches
struct LaunchCaps { int guard = 0, forward = 0; };
LaunchCaps &launch_caps();
}

#include <sched.h>

#include <cmath>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <future>
#include <vector>

namespace {

// Hand-offs between the LANES of one wavefront (every role in this file is one wavefront).  LDS and vector-memory
// operations of a wavefront execute in order, so they need no s_barrier (and no drain of the outstanding global
// loads / stores that the compiler puts in front of one): wavefront-scope fences order the accesses for the
// compiler and cost no instruction.  Hand-offs BETWEEN wavefronts go through LDS flags with workgroup-scope
// release / acquire (pipe_store / pipe_load below).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


constexpr int kEmpty = 0, kBlack = 1, kWhite = 2, kOob = 3;
constexpr int kNotExpanded = -1;

struct RootMeta {
    uint64_t hash;
    int32_t moves, ko_pos, ko_move, prev, prevprev, to_move;
    int32_t num_nodes;
    int32_t hist_len;   // number of valid history slots (== moves, clamped)
};

struct NodeRec {
    int32_t visits, vl;         // node_visits, virtual_loss
    float vsum, raw;            // node_value_sum (float32 like the reference's np.float32 accumulator), raw_value
    int32_t children, parent, pedge, pad;
};
static_assert(sizeof(NodeRec) == 32, "two node records per 64-byte line");

struct SearchDev {
    // node pool, [T][N][A]
    int32_t *ch_index, *ch_visits, *ch_vl;
    double *ch_vsum, *ch_policy, *ch_value;
    int16_t *action;
    // node scalars, [T][N]: ONE 32-byte record per node (round 6; seven separate arrays until then - a path level of the backup
    // touched three 64-byte lines for visits / virtual loss / value sum, a selection step three for children / visits / virtual loss)
    NodeRec *node;
    double *noise;              // [T][A]   root Gumbel noise (0 for PUCT)
    // root position, per tree
    uint8_t *root_cells;        // [T][NC]
    uint64_t *root_hist;        // [T][HMAX]
    RootMeta *meta;             // [T]
    // leaf queue, [T][K]
    int32_t *q_node, *q_pnode, *q_pedge;
    // optional root->leaf path of a queued leaf, (node << 10 | edge) per level, kPathCap entries per slot;
    // q_depth = number of levels (0: not recorded - the backup then follows the parent pointers)
    int32_t *q_depth, *q_path;
    int32_t *n_leaves;          // [T]
    // random stream
    const double *rng;          // [T][rng_cap]  e_i = -log(1-u_i)
    int64_t *rng_cursor;        // [T]
    int64_t *cursor_pub;        // [T] host-mapped mirror of rng_cursor (written with it: the host reads the cursors behind the
                                // selection launch's event without a copy kernel, which would queue for a CU behind the forward pass)
    int64_t rng_cap;
    const uint64_t *zob;        // [4][NC]
    int32_t *err;               // [T] sticky error flags
    long long *prof;            // optional [16] s_memtime cycle accumulators of tree 0 (tg_search_profile)
    int32_t T, N, K, cgos, superko;
    int32_t gumbel_one_by_one;  // test hook (TG_GUMBEL_ONE_BY_ONE): select_gumbel_pipe_kernel takes every entry through its job ring, as a phase with a long path does
};

__device__ __forceinline__ void set_cursor(const SearchDev &D, int t, long long v) {
    D.rng_cursor[t] = v;
    if (D.cursor_pub) D.cursor_pub[t] = v;
}

constexpr int kPathCap = 48;      // (24 until round 5: the last mini-batches of a 1 600-visit 19x19 search walk 25 levels and fell to the one-wave backup, 560 us instead of 30)
enum : int32_t { kErrPoolFull = 1, kErrRngEmpty = 2, kErrPipeline = 4 };

template <int S>
struct Geo {
    static constexpr int W = S + 2;
    static constexpr int NC = W * W;
    static constexpr int P = S * S;
    static constexpr int A = P + 1;
    static constexpr int HMAX = 3 * P;
};

// ---- wave helpers ----------------------------------------------------------------------
// Cross-lane exchange for all-reduce butterflies.  Steps inside a row of 16 lanes use DPP
// (quad_perm for xor 1 / 2, row_half_mirror / row_mirror for the 8- and 16-lane steps: mirrors
// pair every lane with one from the other half, which is all an all-reduce needs); only the two
// cross-row steps go through ds_bpermute.  ~3x fewer LDS-crossbar round trips than six
// __shfl_xor rounds on a (double, int) pair.
template <int STEP>
__device__ __forceinline__ int lane_partner_i32(int v) {
    if constexpr (STEP == 0) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    else if constexpr (STEP == 1) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (STEP == 2) return __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    else if constexpr (STEP == 3) return __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true);  // row_mirror
    else if constexpr (STEP == 4) return __shfl_xor(v, 16);
    else return __shfl_xor(v, 32);
}
template <int STEP>
__device__ __forceinline__ double lane_partner_f64(double v) {
    const long long bits = __double_as_longlong(v);
    const int lo = lane_partner_i32<STEP>((int)(unsigned)bits);
    const int hi = lane_partner_i32<STEP>((int)(unsigned)((unsigned long long)bits >> 32));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// inclusive prefix sum over the wave's lanes: four row shifts, two row broadcasts (no LDS crossbar round trips)
__device__ __forceinline__ int wave_scan_add_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
    v += lane_partner_i32<0>(v);
    v += lane_partner_i32<1>(v);
    v += lane_partner_i32<2>(v);
    v += lane_partner_i32<3>(v);
    v += lane_partner_i32<4>(v);
    v += lane_partner_i32<5>(v);
    return v;
}
__device__ __forceinline__ uint64_t wave_xor64(uint64_t v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo ^= (unsigned)lane_partner_i32<0>((int)lo); hi ^= (unsigned)lane_partner_i32<0>((int)hi);
    lo ^= (unsigned)lane_partner_i32<1>((int)lo); hi ^= (unsigned)lane_partner_i32<1>((int)hi);
    lo ^= (unsigned)lane_partner_i32<2>((int)lo); hi ^= (unsigned)lane_partner_i32<2>((int)hi);
    lo ^= (unsigned)lane_partner_i32<3>((int)lo); hi ^= (unsigned)lane_partner_i32<3>((int)hi);
    lo ^= (unsigned)lane_partner_i32<4>((int)lo); hi ^= (unsigned)lane_partner_i32<4>((int)hi);
    lo ^= (unsigned)lane_partner_i32<5>((int)lo); hi ^= (unsigned)lane_partner_i32<5>((int)hi);
    return ((uint64_t)hi << 32) | lo;
}
// arg-max with lowest-index tie-break (np.argmax); idx < 0 marks "no candidate"
template <int STEP>
__device__ __forceinline__ void argmax_step(double &val, int &idx) {
    const double ov = lane_partner_f64<STEP>(val);
    const int oi = lane_partner_i32<STEP>(idx);
    const bool take = oi >= 0 && (idx < 0 || ov > val || (ov == val && oi < idx));
    if (take) { val = ov; idx = oi; }
}
__device__ __forceinline__ void wave_argmax(double &val, int &idx) {
    argmax_step<0>(val, idx);
    argmax_step<1>(val, idx);
    argmax_step<2>(val, idx);
    argmax_step<3>(val, idx);
    argmax_step<4>(val, idx);
    argmax_step<5>(val, idx);
}
// broadcast of a double from a wave-uniform lane (v_readlane: no LDS crossbar)
__device__ __forceinline__ double read_lane_f64(double v, int src_lane) {
    const long long bits = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, src_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bits >> 32), src_lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// np.argmax over per-lane candidates (idx < 0: none): the maximum by a max butterfly inside each row of 16 lanes
// (DPP) and a v_readlane of the four row results, then the lowest index among the lanes that hold it the same
// way.  ~40 instructions and no branches; the compare-and-swap butterfly above costs ~150 with its exec-mask
// juggling.  Returns the wave-uniform winner (-1: no candidate).  All 64 lanes must be active.
__device__ __forceinline__ int wave_argmax_first(double val, int idx) {
    // (compare + select, not fmax: fmax quiets signalling NaNs first - a v_max_f64 x, x in front of every step; the scores
    // compared here are never NaN unless the network has diverged, and then any deterministic answer will do)
    auto mx2 = [](double a, double b) { return b > a ? b : a; };
    double m = idx >= 0 ? val : -__builtin_inf();
    m = mx2(m, lane_partner_f64<0>(m));
    m = mx2(m, lane_partner_f64<1>(m));
    m = mx2(m, lane_partner_f64<2>(m));
    m = mx2(m, lane_partner_f64<3>(m));
    const double mx = mx2(mx2(read_lane_f64(m, 0), read_lane_f64(m, 16)), mx2(read_lane_f64(m, 32), read_lane_f64(m, 48)));
    int c = (idx >= 0 && val == mx) ? idx : 0x7fffffff;
    c = min(c, lane_partner_i32<0>(c));
    c = min(c, lane_partner_i32<1>(c));
    c = min(c, lane_partner_i32<2>(c));
    c = min(c, lane_partner_i32<3>(c));
    const int w = min(min(__builtin_amdgcn_readlane(c, 0), __builtin_amdgcn_readlane(c, 16)),
                      min(__builtin_amdgcn_readlane(c, 32), __builtin_amdgcn_readlane(c, 48)));
    return w == 0x7fffffff ? -1 : w;
}

// ---- board in LDS ------------------------------------------------------------------------
struct BoardScalars {
    uint64_t hash;
    int moves, ko_pos, ko_move, prev, prevprev;
};

// HALVING = false drops the two softmax scratch vectors that only the one-wave Gumbel kernel keeps in its board
template <int S, bool HALVING = true>
struct Lds {
    using G = Geo<S>;
    uint64_t hist[G::HMAX];
    double w1[HALVING ? G::A + 7 : 1];
    double w2[HALVING ? G::A + 7 : 1];
    uint64_t strhash[G::NC];
    uint32_t libcnt[G::NC];
    uint32_t strsize[G::NC];
    uint16_t root_sid[G::NC];
    uint16_t sid[G::NC];
    uint16_t cand[G::A + 3];
    uint8_t root_color[G::NC + 7];
    uint8_t color[G::NC + 7];
    int32_t scratch[8];
};

template <int S, typename LT>
__device__ __forceinline__ bool touches(const LT &L, int p, int id) {
    constexpr int W = Geo<S>::W;
    return L.sid[p - W] == id || L.sid[p - 1] == id || L.sid[p + 1] == id || L.sid[p + W] == id;
}

// go_board.py:131-185 on the LDS board.  All 64 lanes call it with wave-uniform arguments.
template <int S, typename LT>
__device__ void put_stone(LT &L, BoardScalars &b, int pos, int c, const uint64_t *zob, int lane) {
    using G = Geo<S>;
    constexpr int W = G::W, NC = G::NC;
    if (pos == 0) {                                   // PASS: record only (go_board.py:138-141)
        if (lane == 0 && b.moves < G::HMAX) L.hist[b.moves] = b.hash;
        b.prevprev = b.prev;
        b.prev = 0;
        b.moves += 1;
        wave_sync();
        return;
    }
    const int opp = 3 - c;
    if (lane == 0) L.color[pos] = (uint8_t)c;
    b.hash ^= zob[c * NC + pos];
    wave_sync();
    const int nb[4] = {pos - W, pos - 1, pos + 1, pos + W};
    int captured = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int n = nb[d];
        if (L.color[n] != opp) continue;               // wave-uniform
        const int id = L.sid[n];
        int cnt = 0;
        for (int p = lane; p < NC; p += 64)
            if (L.color[p] == kEmpty && touches<S>(L, p, id)) ++cnt;
        if (wave_sum(cnt) != 0) continue;
        // capture string `id` (string.py:292-325): clear stones, XOR their keys out
        uint64_t hx = 0;
        int removed = 0;
        for (int p = lane; p < NC; p += 64)
            if (L.sid[p] == id) { hx ^= zob[opp * NC + p]; ++removed; }
        wave_sync();
        for (int p = lane; p < NC; p += 64)
            if (L.sid[p] == id) { L.color[p] = kEmpty; L.sid[p] = 0; }
        b.hash ^= wave_xor64(hx);
        captured += wave_sum(removed);
        wave_sync();
    }
    // connect with friendly neighbours: string id = smallest stone coordinate
    int f[4];
    int newid = pos;
    bool any_friend = false;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        f[d] = (L.color[nb[d]] == c) ? L.sid[nb[d]] : 0;
        if (f[d]) { any_friend = true; newid = f[d] < newid ? f[d] : newid; }
    }
    if (any_friend) {
        for (int p = lane; p < NC; p += 64) {
            const int s = L.sid[p];
            if (s != 0 && (s == f[0] || s == f[1] || s == f[2] || s == f[3])) L.sid[p] = (uint16_t)newid;
        }
    } else if (captured == 1) {
        // ko, go_board.py:173-177: lone stone, exactly one capture, exactly one liberty
        int libs = 0, where = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (L.color[nb[d]] == kEmpty) { ++libs; if (!where) where = nb[d]; }
        if (libs == 1) { b.ko_move = b.moves; b.ko_pos = where; }
    }
    wave_sync();
    if (lane == 0) {
        L.sid[pos] = (uint16_t)newid;
        if (b.moves < G::HMAX) L.hist[b.moves] = b.hash;
    }
    b.prevprev = b.prev;
    b.prev = pos;
    b.moves += 1;
    wave_sync();
}

// String ids from colours by min-label propagation (used once per root position).
template <int S>
__device__ void label_strings(uint8_t *color, uint16_t *sid, int lane) {
    using G = Geo<S>;
    constexpr int W = G::W, NC = G::NC;
    for (int p = lane; p < NC; p += 64) {
        const int c = color[p];
        sid[p] = (c == kBlack || c == kWhite) ? (uint16_t)p : 0;
    }
    wave_sync();
    for (int iter = 0; iter < NC; ++iter) {
        int changed = 0;
        for (int p = lane; p < NC; p += 64) {
            const int c = color[p];
            if (c != kBlack && c != kWhite) continue;
            int s = sid[p];
            const int nn[4] = {p - W, p - 1, p + 1, p + W};
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (color[nn[d]] == c && sid[nn[d]] < s) s = sid[nn[d]];
            if (s != sid[p]) { sid[p] = (uint16_t)s; changed = 1; }
        }
        wave_sync();
        if (wave_sum(changed) == 0) break;
    }
}

template <int S, typename LT>
__device__ __forceinline__ int pat3_at(const LT &L, int p) {
    constexpr int W = Geo<S>::W;
    return L.color[p - W - 1] | (L.color[p - W] << 2) | (L.color[p - W + 1] << 4) |
           (L.color[p - 1] << 6) | (L.color[p + 1] << 8) | (L.color[p + W - 1] << 10) |
           (L.color[p + W] << 12) | (L.color[p + W + 1] << 14);
}

// Eye colour of an empty point from its 3x3 neighbourhood (pattern.py:52-98,153-162) by the rule
// the reference's table is exactly equal to on real boards (tests/test_oracle_board.py::
// test_eye_table pins the rule to the reference's table entry by entry):
// orthogonals own or border; 4 on-board diagonals: <= 1 opponent, or 2 opponent + 2 own;
// 2 on-board diagonals (side): >= 1 own or both opponent; 1 on-board diagonal (corner): always.
__device__ __forceinline__ bool is_eye_of(int code, int me) {
    const int opp = 3 - me;
    const int n = (code >> 2) & 3, w = (code >> 6) & 3, e = (code >> 8) & 3, s = (code >> 12) & 3;
    if (!((n == me || n == kOob) && (w == me || w == kOob) && (e == me || e == kOob) && (s == me || s == kOob)))
        return false;
    const int d0 = code & 3, d1 = (code >> 4) & 3, d2 = (code >> 10) & 3, d3 = (code >> 14) & 3;
    const int n_free = (d0 != kOob) + (d1 != kOob) + (d2 != kOob) + (d3 != kOob);
    const int n_own = (d0 == me) + (d1 == me) + (d2 == me) + (d3 == me);
    const int n_opp = (d0 == opp) + (d1 == opp) + (d2 == opp) + (d3 == opp);
    if (n_free == 4) return n_opp <= 1 || (n_opp == 2 && n_own == 2);
    if (n_free == 2) return n_own >= 1 || n_opp == 2;
    return true;
}

// Candidate list of MCTSTree.expand_node (tree.py:260-264): legal (go_board.py:260-304),
// check_self_atari_stone < 7 (:327-365), not a complete eye (:367-397); row-major, PASS last.
// Returns the number of candidates (>= 1); L.cand[] holds their coordinates.
template <int S, typename LT>
__device__ int gen_candidates(LT &L, const BoardScalars &b, int me, const SearchDev &D, int lane) {
    using G = Geo<S>;
    constexpr int W = G::W, NC = G::NC, P = G::P;
    const int opp = 3 - me;
    for (int p = lane; p < NC; p += 64) { L.libcnt[p] = 0; L.strsize[p] = 0; L.strhash[p] = 0; }
    wave_sync();
    for (int p = lane; p < NC; p += 64) {
        const int c = L.color[p];
        if (c == kBlack || c == kWhite) {
            atomicAdd(&L.strsize[L.sid[p]], 1u);
            // superko quirk (go_board.py:292-296): keys of the mover's OPPONENT colour are
            // used for every adjacent one-liberty string, whatever its colour
            if (D.superko) atomicXor((unsigned long long *)&L.strhash[L.sid[p]],
                                     (unsigned long long)D.zob[opp * NC + p]);
        } else if (c == kEmpty) {
            const int s0 = L.sid[p - W], s1 = L.sid[p - 1], s2 = L.sid[p + 1], s3 = L.sid[p + W];
            if (s0) atomicAdd(&L.libcnt[s0], 1u);
            if (s1 && s1 != s0) atomicAdd(&L.libcnt[s1], 1u);
            if (s2 && s2 != s0 && s2 != s1) atomicAdd(&L.libcnt[s2], 1u);
            if (s3 && s3 != s0 && s3 != s1 && s3 != s2) atomicAdd(&L.libcnt[s3], 1u);
        }
    }
    wave_sync();

    int n = 0;
    for (int q0 = 0; q0 < P; q0 += 64) {
        const int q = q0 + lane;
        const int p = (q < P) ? (q / S + 1) * W + (q % S) + 1 : 0;
        bool keep = false;
        bool slow = false;          // self-atari needs the exact liberty-union count
        int fr[4] = {0, 0, 0, 0};
        int col[4] = {0, 0, 0, 0}, sid[4] = {0, 0, 0, 0}, lib[4] = {0, 0, 0, 0};
        int ne = 0;
        bool legal = false;
        uint64_t h = 0;
        if (q < P && L.color[p] == kEmpty) {
            const int nn[4] = {p - W, p - 1, p + 1, p + W};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                col[d] = L.color[nn[d]];
                sid[d] = L.sid[nn[d]];
                lib[d] = sid[d] ? (int)L.libcnt[sid[d]] : 0;
                ne += col[d] == kEmpty;
            }
            // suicide (go_board.py:237-258), only consulted when there is no empty neighbour
            bool suicide = true;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (col[d] == opp && lib[d] == 1) suicide = false;
                if (col[d] == me && lib[d] > 1) suicide = false;
            }
            legal = !(ne == 0 && suicide);
            if (b.ko_pos == p && b.ko_move == b.moves - 1) legal = false;
            if (legal && D.superko) {
                h = b.hash ^ D.zob[me * NC + p];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    bool dup = false;
#pragma unroll
                    for (int e = 0; e < d; ++e) dup |= sid[e] == sid[d];
                    if (sid[d] && !dup && lib[d] == 1) h ^= L.strhash[sid[d]];
                }
            }
        }
        if (D.superko) {
            // record.has_same_hash scans the whole fixed array; unused slots are 0 and slot 0 is never
            // written, so comparing with slots [0, moves) is equivalent.  Wave-uniform loop, four
            // uniform-address reads per round and no early exit: the reads pipeline (a per-lane loop
            // with a break paid one LDS round trip per history entry - 10 k cycles at move 100).
            const int hl = b.moves < G::HMAX ? b.moves : G::HMAX;
            bool seen = false;
            for (int i = 0; i < hl; i += 4) {
                const uint64_t a0 = L.hist[i];
                const uint64_t a1 = L.hist[i + 1 < G::HMAX ? i + 1 : i];
                const uint64_t a2 = L.hist[i + 2 < G::HMAX ? i + 2 : i];
                const uint64_t a3 = L.hist[i + 3 < G::HMAX ? i + 3 : i];
                seen |= a0 == h;
                seen |= i + 1 < hl && a1 == h;
                seen |= i + 2 < hl && a2 == h;
                seen |= i + 3 < hl && a3 == h;
            }
            if (seen) legal = false;
        }
        if (legal) {
            {
                // self-atari size (go_board.py:327-365); only "< 7" matters
                if (ne <= 1) {
                    int size = 0;
                    bool opp_atari = false, big_lib = false;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        if (col[d] == me) {
                            bool dup = false;
#pragma unroll
                            for (int e = 0; e < d; ++e) dup |= (col[e] == me && sid[e] == sid[d]);
                            if (!dup) { size += (int)L.strsize[sid[d]]; fr[d] = sid[d]; big_lib |= lib[d] >= 3; }
                        } else if (col[d] == opp && lib[d] == 1) {
                            opp_atari = true;
                        }
                    }
                    if (size + 1 >= 7 && !opp_atari && !big_lib) { slow = true; }
                }
                // complete eye (go_board.py:367-397)
                bool eye = false;
                if (is_eye_of(pat3_at<S>(L, p), me)) {
                    const int cr[4] = {p - W - 1, p - W + 1, p + W - 1, p + W + 1};
                    int count = 0;
                    bool edge = false;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int v = L.color[cr[d]];
                        if (v == me || v == kOob) ++count;
                        else if (v == kEmpty && is_eye_of(pat3_at<S>(L, cr[d]), me)) ++count;
                        if (v == kOob) edge = true;
                    }
                    eye = (edge && count == 4) || (!edge && count >= 3);
                }
                keep = !eye;
                if (eye) slow = false;
            }
        }
        // rare slow path: |{empty nbrs of p} U liberties of the friendly neighbour strings| >= 3 ?
        unsigned long long pending = __ballot(slow);
        while (pending) {
            const int src = __ffsll((long long)pending) - 1;
            pending &= pending - 1;
            const int sp = __shfl(p, src);
            const int g0 = __shfl(fr[0], src), g1 = __shfl(fr[1], src), g2 = __shfl(fr[2], src),
                      g3 = __shfl(fr[3], src);
            int cnt = 0;
            for (int e = lane; e < NC; e += 64) {
                if (L.color[e] != kEmpty) continue;
                bool in = (e == sp) || e == sp - W || e == sp - 1 || e == sp + 1 || e == sp + W;
                if (!in) {
                    const int t0 = L.sid[e - W], t1 = L.sid[e - 1], t2 = L.sid[e + 1], t3 = L.sid[e + W];
                    const int ts[4] = {t0, t1, t2, t3};
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (ts[d] && (ts[d] == g0 || ts[d] == g1 || ts[d] == g2 || ts[d] == g3)) in = true;
                }
                cnt += in;
            }
            const int total = wave_sum(cnt);
            if (lane == src && total < 3) keep = false;   // self-atari of >= 7 stones: pruned
        }
        const unsigned long long kept = __ballot(keep);
        if (keep) L.cand[n + __popcll(kept & ((1ull << lane) - 1))] = (uint16_t)p;
        n += __popcll(kept);
    }
    if (lane == 0) L.cand[n] = 0;   // PASS
    wave_sync();
    return n + 1;
}

// e_0 + e_1 + ... + e_{n-1} in exactly that order (numpy's dirichlet), addend i held by lane i % 64 in v[i / 64].
// v_readlane moves the addends into scalar registers independently of the running sum, so only the n dependent
// float64 adds are on the critical path and nothing goes through LDS (staging the addends there and reading
// them back eight at a time cost 5.8 k cycles for 82 addends).  Blocks of eight; a block past n is skipped, the
// tail of the last block adds +0.0, which leaves a positive sum unchanged.
template <int R>
__device__ __forceinline__ double sequential_sum(const double (&v)[R], int n) {
    const int lane = threadIdx.x & 63;
    double z[R];
#pragma unroll
    for (int r = 0; r < R; ++r) z[r] = lane + 64 * r < n ? v[r] : 0.0;
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int l0 = 0; l0 < 64; l0 += 8) {
            if (64 * r + l0 < n) {                                  // wave-uniform
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = read_lane_f64(z[r], l0 + u);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += x[u];
            }
        }
    }
    return acc;
}

// tree.py:247-270 expand_node + node.py:41-73: returns the new node index (or -1 on error).
template <int S, typename LT>
__device__ int expand_node(LT &L, const BoardScalars &b, int to_move, const SearchDev &D, int t,
                           int &num_nodes, int parent, int pedge, int lane) {
    using G = Geo<S>;
    constexpr int A = G::A;
    if (num_nodes >= D.N) {
        if (lane == 0) atomicOr(&D.err[t], kErrPoolFull);
        return -1;
    }
    const int node = num_nodes;
    // Dirichlet(1,..,1) prior from the host stream: p_i = e_i / (e_0 + e_1 + ... sequentially).
    // The stream values are requested BEFORE the candidate generation (n <= A is not known yet,
    // every lane takes its ceil(A/64) slots) so that their latency hides behind the LDS work.
    constexpr int R = (A + 63) / 64;
    const int64_t cur = D.rng_cursor[t];
    const double *e = D.rng + (size_t)t * D.rng_cap;
    double mine[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t at = cur + lane + 64 * r;
        if (at >= D.rng_cap) at = D.rng_cap - 1;
        mine[r] = e[at];
    }
    const bool prof = D.prof && t == 0 && lane == 0;
    long long tp = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
    auto lap = [&](int slot) {
        if (prof) { const long long now = (long long)__builtin_amdgcn_s_memtime(); D.prof[slot] += now - tp; tp = now; }
    };
    const int n = gen_candidates<S>(L, b, to_move, D, lane);
    lap(8);
    if (cur + n > D.rng_cap) {
        if (lane == 0) atomicOr(&D.err[t], kErrRngEmpty);
        return -1;
    }
    // sequential sum e_0 + e_1 + ... exactly like numpy's dirichlet
    const double acc = sequential_sum<R>(mine, n);
    const double inv = 1.0 / acc;
    lap(9);
    const size_t base = ((size_t)t * D.N + node) * A;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        if (i < A) {
            D.ch_index[base + i] = kNotExpanded;
            D.ch_visits[base + i] = 0;
            D.ch_vl[base + i] = 0;
            D.ch_vsum[base + i] = 0.0;
            D.ch_value[base + i] = 0.0;
            D.ch_policy[base + i] = i < n ? mine[r] * inv : 0.0;
            D.action[base + i] = i < n ? (int16_t)L.cand[i] : (int16_t)0;
        }
    }
    if (lane == 0) {
        const size_t ns = (size_t)t * D.N + node;
        D.node[ns].children = n;
        D.node[ns].visits = 0;
        D.node[ns].vl = 0;
        D.node[ns].vsum = 0.f;
        D.node[ns].raw = 0.f;
        D.node[ns].parent = parent;
        D.node[ns].pedge = pedge;
        set_cursor(D, t, cur + n);
    }
    num_nodes += 1;
    wave_sync();
    lap(10);
    return node;
}

// nn/feature.py:10-57 from the LDS board
template <int S, typename LT>
__device__ void write_planes(const LT &L, const BoardScalars &b, int to_move, float *dst, int lane) {
    using G = Geo<S>;
    constexpr int W = G::W, P = G::P;
    const bool pass_plane = b.moves > 1 && b.prev == 0;
    const float side = to_move == kWhite ? -1.f : 1.f;
    for (int q = lane; q < P; q += 64) {
        const int p = (q / S + 1) * W + (q % S) + 1;
        int c = L.color[p];
        if (to_move == kWhite && c != 0) c = 3 - c;
        dst[q] = c == 0 ? 1.f : 0.f;
        dst[P + q] = c == 1 ? 1.f : 0.f;
        dst[2 * P + q] = c == 2 ? 1.f : 0.f;
        dst[3 * P + q] = (!pass_plane && p == b.prev) ? 1.f : 0.f;
        dst[4 * P + q] = pass_plane ? 1.f : 0.f;
        dst[5 * P + q] = side;
    }
}

constexpr int kRcpN = 2048;

// a / b for an integer-valued b whose reciprocal y = RN(1 / b) is at hand: q0 = RN(a y) is within an ulp of the quotient, the
// remainder r = a - b q0 is exact in one FMA, and RN(q0 + r y) is the correctly rounded quotient (Markstein; b's significand
// is never all ones here: b < 2^31) - the bits of the IEEE division in four instructions instead of the ~25 of v_div_*.
// Checked against exact rational arithmetic for 300 000 (a, b <= 4096) on the CPU and by every tree digest of the GPU suite.
__device__ __forceinline__ double div_by_count(double a, double b, double y) {
    const double q0 = a * y;
    const double r = __builtin_fma(-q0, b, a);
    return __builtin_fma(r, y, q0);
}

// node.py:141-157 + pucb.py:8-29.  Everything the descent step needs from the node is loaded
// in ONE round trip (the loads are independent): the three node scalars and, per lane, the
// child arrays of its (up to ceil(A/64)) slots incl. action and child index; the winner's
// fields travel with the arg-max.
struct EdgePick {
    int edge, move, child, count, edge_vl, node_vl;   // count = visits + virtual loss of the edge
};

// pucb.py:8-29 on a node's statistics held in registers (lane i + 64 r: child i + 64 r); `total` = visits + virtual
// losses of the node.  The arithmetic of select_puct, shared with the kernels that load the statistics themselves.
// `rcp` (optional, LDS): rcp[n] = RN(1 / n) for n < kRcpN - both quotients then cost four instructions each instead of the
// ~25 of the division sequence (div_by_count: the same bits).
template <int S>
__device__ __forceinline__ EdgePick score_puct(const SearchDev &D, const int (&vis)[(Geo<S>::A + 63) / 64],
                                               const int (&vl)[(Geo<S>::A + 63) / 64], const int (&idx)[(Geo<S>::A + 63) / 64],
                                               const int (&act)[(Geo<S>::A + 63) / 64], const double (&vsum)[(Geo<S>::A + 63) / 64],
                                               const double (&pol)[(Geo<S>::A + 63) / 64], int nc, int total, int node_vl, int lane,
                                               const double *rcp = nullptr) {
    constexpr int R = (Geo<S>::A + 63) / 64;
    const double sq = __dsqrt_rn((double)(total + 1));
    double best = 0.0;
    int best_i = -1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        const int cnt = vis[r] + vl[r];
        // A group of 64 children none of which has been tried needs no division: q = 0 and u = (p sqrt) / 1, and
        // 0.0 + x / 1.0 is x bit for bit.  Most groups of most nodes are like that.
        if (__any(i < nc && cnt != 0)) {
            if (rcp && !__any(i < nc && cnt + 1 >= kRcpN)) {
                if (i < nc) {
                    const int c1 = cnt != 0 ? cnt : 1;
                    const double q = cnt != 0 ? div_by_count(vsum[r], (double)c1, rcp[c1]) : 0.0;
                    const double u = div_by_count(pol[r] * sq, (double)(cnt + 1), rcp[cnt + 1]);
                    double sc = q + u;
                    if (D.cgos && i == nc - 1) sc -= 0.1;
                    if (best_i < 0 || sc > best) { best = sc; best_i = i; }
                }
            } else if (i < nc) {
                const double q = cnt != 0 ? vsum[r] / (double)cnt : 0.0;
                const double u = (pol[r] * sq) / (double)(cnt + 1);
                double sc = q + u;
                if (D.cgos && i == nc - 1) sc -= 0.1;
                if (best_i < 0 || sc > best) { best = sc; best_i = i; }
            }
        } else if (i < nc) {
            double sc = pol[r] * sq;
            if (D.cgos && i == nc - 1) sc -= 0.1;
            if (best_i < 0 || sc > best) { best = sc; best_i = i; }
        }
    }
    best_i = wave_argmax_first(best, best_i);
    const int owner = best_i & 63, slot = best_i >> 6;
    int my_move = 0, my_child = 0, my_cnt = 0, my_vl = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r == slot) { my_move = act[r]; my_child = idx[r]; my_cnt = vis[r] + vl[r]; my_vl = vl[r]; }
    EdgePick pick;
    pick.edge = best_i;
    const int src = __builtin_amdgcn_readfirstlane(owner);        // wave-uniform after the arg-max
    pick.move = __builtin_amdgcn_readlane(my_move, src);
    pick.child = __builtin_amdgcn_readlane(my_child, src);
    pick.count = __builtin_amdgcn_readlane(my_cnt, src);
    pick.edge_vl = __builtin_amdgcn_readlane(my_vl, src);
    pick.node_vl = node_vl;
    return pick;
}

template <int S>
__device__ EdgePick select_puct(const SearchDev &D, int t, int node, int lane, const double *rcp = nullptr) {
    constexpr int A = Geo<S>::A;
    constexpr int R = (A + 63) / 64;
    const size_t ns = (size_t)t * D.N + node;
    const size_t base = ns * A;
    int vis[R], vl[R], idx[R], act[R];
    double vsum[R], pol[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        const int ii = i < A ? i : A - 1;          // the arrays always hold A slots
        vis[r] = D.ch_visits[base + ii];
        vl[r] = D.ch_vl[base + ii];
        vsum[r] = D.ch_vsum[base + ii];
        pol[r] = D.ch_policy[base + ii];
        idx[r] = D.ch_index[base + ii];
        act[r] = D.action[base + ii];
    }
    const int nc = D.node[ns].children;
    const int node_vl = D.node[ns].vl;
    const int total = D.node[ns].visits + node_vl;
    return score_puct<S>(D, vis, vl, idx, act, vsum, pol, nc, total, node_vl, lane, rcp);
}

template <int S, typename LT>
__device__ void load_root(LT &L, BoardScalars &b, int &to_move, const SearchDev &D, int t, int lane) {
    using G = Geo<S>;
    for (int p = lane; p < G::NC; p += 64) L.root_color[p] = D.root_cells[(size_t)t * G::NC + p];
    const RootMeta m = D.meta[t];
    for (int i = lane; i < m.hist_len; i += 64) L.hist[i] = D.root_hist[(size_t)t * G::HMAX + i];
    b.hash = m.hash;
    b.moves = m.moves;
    b.ko_pos = m.ko_pos;
    b.ko_move = m.ko_move;
    b.prev = m.prev;
    b.prevprev = m.prevprev;
    to_move = m.to_move;
    wave_sync();
    label_strings<S>(L.root_color, L.root_sid, lane);
}

template <int S, typename LT>
__device__ void reset_work(LT &L, int lane) {
    for (int p = lane; p < Geo<S>::NC; p += 64) { L.color[p] = L.root_color[p]; L.sid[p] = L.root_sid[p]; }
    wave_sync();
}

// tree.py:49-54 / :330-336 (first half): reset the tree, expand the root, featurise it.
template <int S>
__global__ __launch_bounds__(64) void root_kernel(SearchDev D, float *planes) {   // planes [T][6][P]
    using G = Geo<S>;
    __shared__ Lds<S> L;
    const int t = blockIdx.x, lane = threadIdx.x;
    BoardScalars b;
    int to_move;
    load_root<S>(L, b, to_move, D, t, lane);
    reset_work<S>(L, lane);
    int num_nodes = 0;
    for (int i = lane; i < G::A; i += 64) D.noise[(size_t)t * G::A + i] = 0.0;
    const int root = expand_node<S>(L, b, to_move, D, t, num_nodes, -1, -1, lane);
    write_planes<S>(L, b, to_move, planes + (size_t)t * 6 * G::P, lane);
    if (lane == 0) {
        D.meta[t].num_nodes = num_nodes;
        D.q_node[(size_t)t * D.K] = root;
        D.q_pnode[(size_t)t * D.K] = -1;
        D.q_pedge[(size_t)t * D.K] = -1;
        D.q_depth[(size_t)t * D.K] = 0;
        D.n_leaves[t] = root >= 0 ? 1 : 0;
    }
}

// tree.py:199-244 search_mcts, `max_leaves` descents per tree.
template <int S>
__global__ __launch_bounds__(64) void select_puct_kernel(SearchDev D, int max_leaves, float *planes) {
    using G = Geo<S>;
    constexpr int A = G::A;
    __shared__ Lds<S> L;
    const int t = blockIdx.x, lane = threadIdx.x;
    BoardScalars rootb;
    int root_to_move;
    load_root<S>(L, rootb, root_to_move, D, t, lane);
    int num_nodes = D.meta[t].num_nodes;
    int queued = 0;
    const bool prof = D.prof && t == 0;
    long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tp = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
    auto lap = [&](int slot) {
        if (prof) { const long long now = (long long)__builtin_amdgcn_s_memtime(); pc[slot] += now - tp; tp = now; }
    };
    if (D.err[t] == 0 && num_nodes > 0) {
        for (int k = 0; k < max_leaves; ++k) {
            reset_work<S>(L, lane);
            BoardScalars b = rootb;
            int c = root_to_move;
            int node = 0;
            bool ok = true;
            lap(0);
            for (;;) {
                const size_t ns = (size_t)t * D.N + node;
                const size_t base = ns * A;
                const EdgePick pick = select_puct<S>(D, t, node, lane);
                const int e = pick.edge;
                const int mv = pick.move;
                lap(1);
                put_stone<S>(L, b, mv, c, D.zob, lane);
                lap(2);
                if (prof) pc[7] += 1;
                c = 3 - c;
                const int edge_cnt = pick.count + 1;                  // after the virtual loss
                int child = pick.child;
                if (lane == 0) {                                      // node.py:76-83
                    D.node[ns].vl = pick.node_vl + 1;
                    D.ch_vl[base + e] = pick.edge_vl + 1;
                }
                // two consecutive passes: never descend below (tree.py:224-229)
                const bool two_pass = b.moves > 2 && b.prev == 0 && b.prevprev == 0;
                const int threshold = two_pass ? 10000000 : 1;
                lap(3);
                if (edge_cnt < threshold + 1) {
                    if (child == kNotExpanded) {
                        child = expand_node<S>(L, b, c, D, t, num_nodes, node, e, lane);
                        if (child < 0) { ok = false; break; }
                        if (lane == 0) D.ch_index[base + e] = child;
                    }
                    lap(4);
                    write_planes<S>(L, b, c, planes + ((size_t)t * max_leaves + k) * 6 * G::P, lane);
                    if (lane == 0) {
                        D.q_node[(size_t)t * D.K + k] = child;
                        D.q_pnode[(size_t)t * D.K + k] = node;
                        D.q_pedge[(size_t)t * D.K + k] = e;
                        D.q_depth[(size_t)t * D.K + k] = 0;
                    }
                    wave_sync();
                    lap(5);
                    break;
                }
                node = child;
                wave_sync();
                lap(3);
            }
            if (!ok) break;
            ++queued;
        }
    }
    if (lane == 0) {
        D.meta[t].num_nodes = num_nodes;
        D.n_leaves[t] = queued;
        if (prof)
            for (int i = 0; i < 8; ++i) D.prof[i] += pc[i];
    }
}

// ---- PUCT selection, pipelined over three wavefronts per tree ------------------------------
// A descent is selection (PUCB walk: needs only the node pool) followed by board work (replay
// of the path on the LDS board, expansion, feature planes: needs no tree statistics).  With
// few trees per GPU the 256 descents of a mini-batch are a serial chain of ~12 us each in
// select_puct_kernel.  Here wave 0 (the selector) only walks the tree - it also assigns the
// node index of a new leaf, so numbering stays in descent order - and hands (path, parent,
// edge, node) to waves 1 and 2 (the workers, alternating), which replay the moves, generate
// candidates, draw the Dirichlet prior and write the planes.  The random draws are reserved in
// expansion order through a cursor chained between the workers; the selector waits for a
// worker only when it steps into a node whose expansion is still in flight.  Results are
// identical to the serial kernel (same tests).
constexpr int kPipeSlots = 4;           // job ring; even, so that a slot always belongs to one worker
constexpr int kPipeMaxDepth = 512;
constexpr int kPipeMaxK = 1024;
constexpr int kPipeSpinLimit = 1 << 24;

struct PipeJob {
    int k, parent, edge, child, expand, xseq, depth, src;
};

// Longest path a pipelined kernel follows (deeper: reported as an error, never silent).  A 9x9 game has at most
// 243 recorded moves (MAX_RECORDS = 3 P), so 256 covers every path there.
template <int S>
constexpr int kPathMax = S == 9 ? 256 : kPipeMaxDepth;

// The three-wave kernels are sized to sit NEXT TO a forward workgroup on a CU (the split-operand forward kernel
// leaves 19.6 KB of the 160 KB): boards without the softmax scratch, path buffers of kPathMax<S>, "job done"
// flags as a bit set.  Otherwise a selection launch of one board group waits for the forward pass of the other
// group to drain instead of overlapping it.
// (NW workers, 2 NW ring slots: with few trees per GPU the Gumbel kernel takes more workers per tree - the CUs are
// idle anyway and its jobs - expansions, leaves, ~100 plane copies per phase - are what a phase waits for.)
template <int S, int NW = 2>
struct PipeShared {
    static constexpr int kSlots = 2 * NW;
    Lds<S, false> board[NW];
    PipeJob job[kSlots];
    int16_t moves[kSlots][kPathMax<S>];
    int paths[kSlots][kPathMax<S>];       // Gumbel jobs: (node << 10 | edge) per level, for the worker's bookkeeping
    int job_seq[kSlots];              // k + 1 once job k sits in its slot
    int slot_done[kSlots];            // jobs finished in this slot so far
    unsigned done_bits[kPipeMaxK / 32];   // bit k: job k finished (node initialised, planes written)
    int16_t jobof[kPipeMaxK];         // node (n0 + i) is being created by job jobof[i]
    int cursor_seq;                   // expansions that have reserved their draws
    long long cursor_val;             // stream position after those reservations
    int final_count;                  // -1 while the selector is still queueing
    int err;
};

__device__ __forceinline__ int pipe_load(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void pipe_store(int *p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// job-done bits of PipeShared
template <typename Sh>
__device__ __forceinline__ void pipe_set_done(Sh &sh, int k) {
    __hip_atomic_fetch_or(&sh.done_bits[k >> 5], 1u << (k & 31), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <typename Sh>
__device__ __forceinline__ bool pipe_wait_done(const Sh &sh, int k);

// wave-uniform wait for *p >= want; false on a stall (reported, never a hang)
__device__ __forceinline__ bool pipe_wait_ge(const int *p, int want) {
    for (int spin = 0; spin < kPipeSpinLimit; ++spin) {
        if (pipe_load(p) >= want) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

template <typename Sh>
__device__ __forceinline__ bool pipe_wait_done(const Sh &sh, int k) {
    const unsigned *p = &sh.done_bits[k >> 5];
    const unsigned bit = 1u << (k & 31);
    for (int spin = 0; spin < kPipeSpinLimit; ++spin) {
        if (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

template <int S, typename LT>
__device__ void expand_fill(LT &L, const SearchDev &D, int t, int node, int parent, int pedge, int n, long long cur, int lane);

// expand_node for a worker: node index given by the selector, draws reserved through the chain
template <int S, typename LT, typename Sh>
__device__ bool expand_node_pipe(LT &L, const BoardScalars &b, int to_move, const SearchDev &D, int t,
                                 int node, int parent, int pedge, int xseq, Sh &sh, int lane) {
    const int n = gen_candidates<S>(L, b, to_move, D, lane);
    bool ok = pipe_wait_ge(&sh.cursor_seq, xseq);
    const long long cur = sh.cursor_val;
    if (!ok || cur + n > D.rng_cap) {
        if (lane == 0) {
            atomicOr(&D.err[t], ok ? kErrRngEmpty : kErrPipeline);
            pipe_store(&sh.err, 1);
            pipe_store(&sh.cursor_seq, xseq + 1);      // keep the chain moving
        }
        return false;
    }
    if (lane == 0) {
        sh.cursor_val = cur + n;
        pipe_store(&sh.cursor_seq, xseq + 1);
    }
    expand_fill<S>(L, D, t, node, parent, pedge, n, cur, lane);
    return true;
}

// the second half of an expansion: prior from the n draws at stream position `cur`, the node's arrays
template <int S, typename LT>
__device__ void expand_fill(LT &L, const SearchDev &D, int t, int node, int parent, int pedge, int n, long long cur, int lane) {
    using G = Geo<S>;
    constexpr int A = G::A;
    constexpr int R = (A + 63) / 64;
    const double *e = D.rng + (size_t)t * D.rng_cap;
    double mine[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        long long at = cur + lane + 64 * r;
        if (at >= D.rng_cap) at = D.rng_cap - 1;
        mine[r] = e[at];
    }
    // sequential sum e_0 + e_1 + ... exactly like numpy's dirichlet (see expand_node)
    const double acc = sequential_sum<R>(mine, n);
    const double inv = 1.0 / acc;
    const size_t base = ((size_t)t * D.N + node) * A;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        if (i < A) {
            D.ch_index[base + i] = kNotExpanded;
            D.ch_visits[base + i] = 0;
            D.ch_vl[base + i] = 0;
            D.ch_vsum[base + i] = 0.0;
            D.ch_value[base + i] = 0.0;
            D.ch_policy[base + i] = i < n ? mine[r] * inv : 0.0;
            D.action[base + i] = i < n ? (int16_t)L.cand[i] : (int16_t)0;
        }
    }
    if (lane == 0) {
        const size_t ns = (size_t)t * D.N + node;
        D.node[ns].children = n;
        D.node[ns].visits = 0;
        D.node[ns].vl = 0;
        D.node[ns].vsum = 0.f;
        D.node[ns].raw = 0.f;
        D.node[ns].parent = parent;
        D.node[ns].pedge = pedge;
    }
    wave_sync();
}

template <int S>
__global__ __launch_bounds__(192) void select_puct_pipe_kernel(SearchDev D, int max_leaves, float *planes) {
    using G = Geo<S>;
    constexpr int A = G::A;
    __shared__ PipeShared<S> sh;
    const int t = blockIdx.x;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const RootMeta meta = D.meta[t];
    const int n0 = meta.num_nodes;
    if (threadIdx.x < kPipeSlots) {
        sh.job_seq[threadIdx.x] = 0;
        sh.slot_done[threadIdx.x] = 0;
    }
    for (int i = threadIdx.x; i < kPipeMaxK / 32; i += 192) sh.done_bits[i] = 0u;
    if (threadIdx.x == 0) {
        sh.cursor_seq = 0;
        sh.cursor_val = D.rng_cursor[t];
        sh.final_count = -1;
        sh.err = 0;
    }
    __syncthreads();
    const bool active = D.err[t] == 0 && n0 > 0;
    int num_nodes = n0;
    int queued = 0;

    if (wid == 0) {
        // ---- selector -------------------------------------------------------------------
        int nexp = 0;
        for (int k = 0; active && k < max_leaves; ++k) {
            if (pipe_load(&sh.err)) break;
            const int slot = k % kPipeSlots;
            bool ok = pipe_wait_ge(&sh.slot_done[slot], k / kPipeSlots);       // ring slot free again
            int node = 0, depth = 0;
            int moves = meta.moves, prev = meta.prev, prevprev = meta.prevprev;
            while (ok) {
                if (node >= n0) ok = pipe_wait_done(sh, sh.jobof[node - n0]);   // expansion still in flight?
                if (!ok) break;
                const size_t ns = (size_t)t * D.N + node;
                const size_t base = ns * A;
                const EdgePick pick = select_puct<S>(D, t, node, lane);
                const int e = pick.edge;
                const int mv = pick.move;
                if (depth >= kPathMax<S>) { ok = false; break; }
                if (lane == 0) {
                    sh.moves[slot][depth] = (int16_t)mv;
                    D.node[ns].vl = pick.node_vl + 1;                             // node.py:76-83
                    D.ch_vl[base + e] = pick.edge_vl + 1;
                    if (depth < kPathCap) D.q_path[((size_t)t * D.K + k) * kPathCap + depth] = (node << 10) | e;
                }
                ++depth;
                prevprev = prev;
                prev = mv;
                ++moves;
                // two consecutive passes: never descend below (tree.py:224-229)
                const bool two_pass = moves > 2 && prev == 0 && prevprev == 0;
                const int threshold = two_pass ? 10000000 : 1;
                if (pick.count + 1 < threshold + 1) {
                    int child = pick.child;
                    const int expand = child == kNotExpanded;
                    int xseq = 0;
                    if (expand) {
                        if (num_nodes >= D.N || num_nodes - n0 >= kPipeMaxK) {
                            if (lane == 0) atomicOr(&D.err[t], kErrPoolFull);
                            ok = false;
                            break;
                        }
                        child = num_nodes++;
                        xseq = nexp++;
                    }
                    if (lane == 0) {
                        if (expand) {
                            D.ch_index[base + e] = child;
                            sh.jobof[child - n0] = (int16_t)k;
                        }
                        PipeJob &j = sh.job[slot];
                        j.k = k; j.parent = node; j.edge = e; j.child = child;
                        j.expand = expand; j.xseq = xseq; j.depth = depth;
                        D.q_node[(size_t)t * D.K + k] = child;
                        D.q_pnode[(size_t)t * D.K + k] = node;
                        D.q_pedge[(size_t)t * D.K + k] = e;
                        D.q_depth[(size_t)t * D.K + k] = (depth <= kPathCap && D.N <= (1 << 21)) ? depth : 0;
                        pipe_store(&sh.job_seq[slot], k + 1);
                    }
                    wave_sync();
                    break;
                }
                node = pick.child;
            }
            if (!ok) {
                if (lane == 0) {
                    if (!(D.err[t] & kErrPoolFull)) atomicOr(&D.err[t], kErrPipeline);
                    pipe_store(&sh.err, 1);
                }
                break;
            }
            ++queued;
        }
        if (lane == 0) pipe_store(&sh.final_count, queued);
    } else {
        // ---- workers ---------------------------------------------------------------------
        Lds<S, false> &L = sh.board[wid - 1];
        BoardScalars rootb;
        int root_to_move;
        load_root<S>(L, rootb, root_to_move, D, t, lane);
        for (int k = wid - 1; active; k += 2) {
            const int slot = k % kPipeSlots;
            bool have = false, stalled = true;
            for (int spin = 0; spin < kPipeSpinLimit; ++spin) {
                if (pipe_load(&sh.job_seq[slot]) == k + 1) { have = true; stalled = false; break; }
                const int fc = pipe_load(&sh.final_count);
                if (fc >= 0 && k >= fc) { stalled = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (stalled && lane == 0) {
                atomicOr(&D.err[t], kErrPipeline);
                pipe_store(&sh.err, 1);
            }
            if (!have) break;
            const PipeJob j = sh.job[slot];
            reset_work<S>(L, lane);
            BoardScalars b = rootb;
            int c = root_to_move;
            for (int i = 0; i < j.depth; ++i) {
                put_stone<S>(L, b, sh.moves[slot][i], c, D.zob, lane);
                c = 3 - c;
            }
            if (j.expand) expand_node_pipe<S>(L, b, c, D, t, j.child, j.parent, j.edge, j.xseq, sh, lane);
            write_planes<S>(L, b, c, planes + ((size_t)t * max_leaves + k) * 6 * G::P, lane);
            wave_sync();
            if (lane == 0) {
                pipe_set_done(sh, k);                                          // releases the node arrays
                pipe_store(&sh.slot_done[slot], k / kPipeSlots + 1);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        D.meta[t].num_nodes = num_nodes;
        D.n_leaves[t] = queued;
        set_cursor(D, t, sh.cursor_val);
    }
}

// ---- PUCT selection with the descents themselves pipelined ----------------------------------
// With few trees per GPU the selector wave above is the critical path: ~3.5 levels per descent, each an L2
// round trip plus ~2 k cycles of fp64 PUCB arithmetic, and descent k + 1 may not start before descent k has
// left its virtual loss behind.  But a descent only needs the virtual losses on the nodes it visits itself.
// So NSEL selector waves take the descents round-robin and a descent scores node n (at level L) once every
// earlier descent still in flight has either chosen a different level-L node or finished with n - a node is
// then always scored with exactly the statistics the serial order gives it:
//   * level_done[k] = levels descent k has left behind; lvl_node[k][L] = its level-L node, published with
//     level_done[k] = L.  Only the NSEL - 1 predecessors of k can still be in flight.
//   * the ROOT is visited by every descent, so its step is the chain that bounds the launch.  Its statistics
//     cannot change inside a launch (the backup runs between launches) except for the virtual losses, so every
//     selector keeps visits / value sums / priors of the root's children in registers and only the virtual
//     losses and child indices live in LDS.  sqrt(N + 1) and the prior products of descent k depend on k alone
//     and are computed BEFORE waiting for descent k - 1; after the wait only children whose virtual loss
//     changed since this wave's last visit get new quotients.  The root's virtual losses reach global memory
//     once, at the end of the launch.
//   * new node indices and the random draws stay in descent order: a descent allocates only after its
//     predecessor is completely done.  Board work goes to NWRK worker waves as before.
// Same trees bit for bit (tests/test_gpu_search.py).
constexpr int kMpDone = 1 << 30;        // level_done value of a finished descent
constexpr int kMpTrackDepth = 64;       // levels with a published node (deeper: wait for the predecessors outright)

template <int S, int NSEL, int NWRK>
struct MPipeShared {
    static constexpr int kMpSlots = 2 * NWRK;      // job ring; a multiple of the worker count
    static constexpr int kRing = 2 * NSEL;         // rows of lvl_node: a row is reused only after its reader is done
    Lds<S, false> board[NWRK];
    PipeJob job[kMpSlots];
    int16_t moves[kMpSlots][kPipeMaxDepth];
    int qpath[kMpSlots][kPathCap];    // (node << 10 | edge) of the first kPathCap levels: the worker writes the queue entry
    int job_seq[kMpSlots];            // k + 1 once job k sits in its slot
    int slot_done[kMpSlots];          // jobs finished in this slot so far
    int done[kPipeMaxK];              // job k finished (node initialised, planes written)
    int level_done[kPipeMaxK];        // levels descent k has left behind (kMpDone: leaf queued)
    int lvl_node[kRing][kMpTrackDepth];
    int16_t jobof[kPipeMaxK];         // node (n0 + i) is being created by job jobof[i]
    int root_vl[Geo<S>::A];           // virtual losses / child indices of the root's children
    int root_idx[Geo<S>::A];
    int num_nodes, nexp;              // owned by the descent whose predecessor is done
    int cursor_seq;
    long long cursor_val;
    int err;
};

// wave-uniform wait for *p >= want: the value read, or -1 on a stall or once another wave has reported an error
template <typename Sh>
__device__ __forceinline__ int mp_wait_val(const Sh &sh, const int *p, int want) {
    for (int spin = 0; spin < kPipeSpinLimit; ++spin) {
        const int v = pipe_load(p);
        if (v >= want) return v;
        if (spin >= 32) {
            if (pipe_load(&sh.err)) return -1;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return -1;
}
template <typename Sh>
__device__ __forceinline__ bool mp_wait_ge(const Sh &sh, const int *p, int want) {
    return mp_wait_val(sh, p, want) >= 0;
}
// ... and a word the writer stored BEFORE the flag, read in the same round trip: LDS operations of a wave are carried out in issue
// order on both sides (flag read first here, data written first there), so `data` is valid whenever the flag read says so
template <typename Sh>
__device__ __forceinline__ bool mp_wait_with(const Sh &sh, const int *flag, int want, const int *data, int &out) {
    for (int spin = 0; spin < kPipeSpinLimit; ++spin) {
        const int f = *reinterpret_cast<const volatile int *>(flag);
        const int d = *reinterpret_cast<const volatile int *>(data);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (f >= want) { out = d; return true; }
        if (spin >= 32) {
            if (pipe_load(&sh.err)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return false;
}

// flag store behind plain LDS stores of the same lane: LDS operations of a wave complete in order, the fence
// only keeps the compiler from reordering them
__device__ __forceinline__ void mp_publish(int *p, int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int S, int NSEL, int NWRK>
__global__ __launch_bounds__(64 * (NSEL + NWRK)) void select_puct_mpipe_kernel(SearchDev D, int max_leaves, float *planes) {
    using G = Geo<S>;
    using Shared = MPipeShared<S, NSEL, NWRK>;
    constexpr int A = G::A;
    constexpr int R = (A + 63) / 64;
    constexpr int NTHR = 64 * (NSEL + NWRK);
    constexpr int kMpSlots = Shared::kMpSlots, kRing = Shared::kRing;
    static_assert(kMpSlots % NWRK == 0 && kMpSlots >= NSEL, "job ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char mp_smem[];
    Shared &sh = *reinterpret_cast<Shared *>(mp_smem);
    const int t = blockIdx.x;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t_begin = (D.prof && t == 0) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    const RootMeta meta = D.meta[t];
    const int n0 = meta.num_nodes;
    const size_t root_ns = (size_t)t * D.N, root_base = root_ns * A;
    const bool active = D.err[t] == 0 && n0 > 0;
    if (threadIdx.x < kMpSlots) {
        sh.job_seq[threadIdx.x] = 0;
        sh.slot_done[threadIdx.x] = 0;
    }
    for (int i = threadIdx.x; i < kPipeMaxK; i += NTHR) { sh.done[i] = 0; sh.level_done[i] = 0; }
    if (active)
        for (int i = threadIdx.x; i < A; i += NTHR) {
            sh.root_vl[i] = D.ch_vl[root_base + i];
            sh.root_idx[i] = D.ch_index[root_base + i];
        }
    if (threadIdx.x == 0) {
        sh.cursor_seq = 0;
        sh.cursor_val = D.rng_cursor[t];
        sh.num_nodes = n0;
        sh.nexp = 0;
        sh.err = 0;
    }
    __syncthreads();
    // per-phase s_memtime accumulators of tree 0 (tg_search_profile with TG_MPIPE_PROF=1; tools/profile_select.py)
    const bool prof = D.prof && t == 0;
    long long pc[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tp = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
    auto lap = [&](int i) {
        if (prof) { const long long now = (long long)__builtin_amdgcn_s_memtime(); pc[i] += now - tp; tp = now; }
    };

    if (wid < NSEL) {
        // ---- selectors: descent k on wave k % NSEL ------------------------------------------
        __builtin_amdgcn_s_setprio(3);          // the critical path: ahead of the workers that share the SIMD
        // the root's children, constant for the launch; c_vl / c_q: virtual loss this wave last saw and its quotient
        int r_vis[R], r_act[R], c_vl[R];
        double r_vsum[R], r_pol[R], c_q[R];
        int root_nc = 0, root_total0 = 0;
        if (active) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = lane + 64 * r, ii = i < A ? i : A - 1;
                r_vis[r] = D.ch_visits[root_base + ii];
                r_act[r] = D.action[root_base + ii];
                r_vsum[r] = D.ch_vsum[root_base + ii];
                r_pol[r] = D.ch_policy[root_base + ii];
                c_vl[r] = D.ch_vl[root_base + ii];
                const int cnt = r_vis[r] + c_vl[r];
                c_q[r] = cnt != 0 ? r_vsum[r] / (double)cnt : 0.0;
            }
            root_nc = D.node[root_ns].children;
            root_total0 = D.node[root_ns].visits + D.node[root_ns].vl;
        }
        for (int k = wid; active && k < max_leaves; k += NSEL) {
            if (pipe_load(&sh.err)) break;
            const int slot = k % kMpSlots;
            int *my_nodes = sh.lvl_node[k % kRing];
            lap(8);
            bool ok = mp_wait_ge(sh, &sh.slot_done[slot], k / kMpSlots);          // ring slot free again
            lap(0);
            int node = 0, depth = 0;
            int moves = meta.moves, prev = meta.prev;
            bool pool_full = false;
            while (ok) {
                EdgePick pick;
                size_t ns = root_ns, base = root_base;
                if (depth == 0) {
                    // every descent before this one has added one virtual loss to the root (node.py:76-83)
                    const double sq = __dsqrt_rn((double)(root_total0 + k + 1));
                    // scores for "virtual loss as this wave last saw it" and for "one more" (what one intervening
                    // descent leaves behind) - all the divisions happen before the wait
                    double psq[R], sc[R], sc1[R], q1[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int cnt = r_vis[r] + c_vl[r];
                        psq[r] = r_pol[r] * sq;
                        sc[r] = c_q[r] + psq[r] / (double)(cnt + 1);
                        q1[r] = r_vsum[r] / (double)(cnt + 1);
                        sc1[r] = q1[r] + psq[r] / (double)(cnt + 2);
                        // keep the arithmetic on this side of the wait (hipcc would sink it to its first use)
                        asm volatile("" : "+v"(sc[r]), "+v"(sc1[r]), "+v"(q1[r]), "+v"(psq[r]));
                    }
                    lap(1);
                    if (k > 0) ok = mp_wait_ge(sh, &sh.level_done[k - 1], 1);      // predecessor has left the root
                    lap(2);
                    if (!ok) break;
                    int idx[R];
                    bool slow = false;
                    int vl[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int i = lane + 64 * r, ii = i < A ? i : A - 1;
                        vl[r] = sh.root_vl[ii];
                        idx[r] = sh.root_idx[ii];
                        const int d = vl[r] - c_vl[r];
                        if (d == 1) { sc[r] = sc1[r]; c_q[r] = q1[r]; c_vl[r] = vl[r]; }
                        slow |= d != 0 && d != 1;
                    }
                    if (__any(slow)) {                                         // several descents through one child
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (vl[r] != c_vl[r]) {
                                c_vl[r] = vl[r];
                                const int cnt = r_vis[r] + vl[r];
                                c_q[r] = cnt != 0 ? r_vsum[r] / (double)cnt : 0.0;
                                sc[r] = c_q[r] + psq[r] / (double)(cnt + 1);
                            }
                    }
                    double best = 0.0;
                    int best_i = -1;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int i = lane + 64 * r;
                        if (i < root_nc) {
                            double v = sc[r];
                            if (D.cgos && i == root_nc - 1) v -= 0.1;
                            if (best_i < 0 || v > best) { best = v; best_i = i; }
                        }
                    }
                    best_i = wave_argmax_first(best, best_i);
                    const int owner = best_i & 63, oslot = best_i >> 6;
                    int my_move = 0, my_child = 0, my_cnt = 0, my_vl = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (r == oslot) { my_move = r_act[r]; my_child = idx[r]; my_cnt = r_vis[r] + vl[r]; my_vl = vl[r]; }
                    const int src = __builtin_amdgcn_readfirstlane(owner);
                    pick.edge = best_i;
                    pick.move = __builtin_amdgcn_readlane(my_move, src);
                    pick.child = __builtin_amdgcn_readlane(my_child, src);
                    pick.count = __builtin_amdgcn_readlane(my_cnt, src);
                    pick.edge_vl = __builtin_amdgcn_readlane(my_vl, src);
                    pick.node_vl = 0;
                    lap(3);
                } else {
                    lap(8);
                    // predecessors still in flight: wait until each has chosen its node of this level, and until
                    // it has left that node if it is this one
                    for (int j = k - 1; ok && j > k - NSEL && j >= 0; --j) {
                        const bool tracked = depth < kMpTrackDepth;
                        const int v = mp_wait_val(sh, &sh.level_done[j], tracked ? depth : depth + 1);
                        ok = v >= 0;
                        if (ok && v == depth && sh.lvl_node[j % kRing][depth] == node)
                            ok = mp_wait_ge(sh, &sh.level_done[j], depth + 1);
                    }
                    lap(4);
                    if (ok && node >= n0) ok = mp_wait_ge(sh, &sh.done[sh.jobof[node - n0]], 1);   // expansion in flight?
                    lap(5);
                    if (!ok) break;
                    ns = (size_t)t * D.N + node;
                    base = ns * A;
                    pick = select_puct<S>(D, t, node, lane);
                    lap(6);
                }
                const int e = pick.edge;
                const int mv = pick.move;
                if (depth >= kPipeMaxDepth) { ok = false; break; }
                // two consecutive passes: never descend below (tree.py:224-229)
                const bool two_pass = moves + 1 > 2 && mv == 0 && prev == 0;
                const int threshold = two_pass ? 10000000 : 1;
                const bool leaf = pick.count + 1 < threshold + 1;
                if (lane == 0) {
                    if (depth == 0) {
                        sh.root_vl[e] = pick.edge_vl + 1;
                    } else {
                        D.node[ns].vl = pick.node_vl + 1;                             // node.py:76-83
                        D.ch_vl[base + e] = pick.edge_vl + 1;
                    }
                    // the successors may have this node as soon as the virtual loss is in place (a descent that ends
                    // here publishes when its leaf is queued: the child index has to be there first)
                    if (!leaf) {
                        if (depth + 1 < kMpTrackDepth) my_nodes[depth + 1] = pick.child;
                        if (depth == 0) mp_publish(&sh.level_done[k], 1);         // only LDS was written at the root
                        else pipe_store(&sh.level_done[k], depth + 1);
                    }
                    sh.moves[slot][depth] = (int16_t)mv;
                    if (depth < kPathCap) sh.qpath[slot][depth] = (node << 10) | e;
                }
                ++depth;
                prev = mv;
                ++moves;
                if (prof) pc[13] += 1;
                if (leaf) {
                    lap(8);
                    if (k > 0) ok = mp_wait_ge(sh, &sh.level_done[k - 1], kMpDone); // node numbers in descent order
                    lap(7);
                    if (!ok) break;
                    int num_nodes = sh.num_nodes, nexp = sh.nexp;
                    int child = pick.child;
                    const int expand = child == kNotExpanded;
                    int xseq = 0;
                    if (expand) {
                        if (num_nodes >= D.N || num_nodes - n0 >= kPipeMaxK) { pool_full = true; ok = false; break; }
                        child = num_nodes++;
                        xseq = nexp++;
                    }
                    if (lane == 0) {
                        if (expand) {
                            D.ch_index[base + e] = child;
                            if (node == 0) sh.root_idx[e] = child;
                            sh.jobof[child - n0] = (int16_t)k;
                            sh.num_nodes = num_nodes;
                            sh.nexp = nexp;
                        }
                        PipeJob &j = sh.job[slot];
                        j.k = k; j.parent = node; j.edge = e; j.child = child;
                        j.expand = expand; j.xseq = xseq; j.depth = depth;
                        // (the leaf's queue entry - node, parent, edge, recorded path - is written by the worker)
                        pipe_store(&sh.job_seq[slot], k + 1);
                        pipe_store(&sh.level_done[k], kMpDone);
                    }
                    break;
                }
                node = pick.child;
            }
            if (!ok) {
                if (lane == 0 && !pipe_load(&sh.err)) {
                    atomicOr(&D.err[t], pool_full ? kErrPoolFull : kErrPipeline);
                    pipe_store(&sh.err, 1);
                }
                break;
            }
        }
    } else {
        // ---- workers: job k on wave k % NWRK --------------------------------------------------
        const int w = wid - NSEL;
        Lds<S, false> &L = sh.board[w];
        BoardScalars rootb;
        int root_to_move;
        load_root<S>(L, rootb, root_to_move, D, t, lane);
        lap(14);
        for (int k = w; active && k < max_leaves; k += NWRK) {
            const int slot = k % kMpSlots;
            lap(12);
            const bool have = mp_wait_ge(sh, &sh.job_seq[slot], k + 1);
            lap(9);
            if (!have) {
                if (lane == 0 && !pipe_load(&sh.err)) {
                    atomicOr(&D.err[t], kErrPipeline);
                    pipe_store(&sh.err, 1);
                }
                break;
            }
            const PipeJob j = sh.job[slot];
            {
                // queue entry of leaf k (what the backup reads): off the selectors' critical path
                const size_t qs = (size_t)t * D.K + k;
                const int npath = j.depth < kPathCap ? j.depth : kPathCap;
                if (lane < npath) D.q_path[qs * kPathCap + lane] = sh.qpath[slot][lane];
                if (lane == 0) {
                    D.q_node[qs] = j.child;
                    D.q_pnode[qs] = j.parent;
                    D.q_pedge[qs] = j.edge;
                    D.q_depth[qs] = (j.depth <= kPathCap && D.N <= (1 << 21)) ? j.depth : 0;
                }
            }
            reset_work<S>(L, lane);
            BoardScalars b = rootb;
            int c = root_to_move;
            for (int i = 0; i < j.depth; ++i) {
                put_stone<S>(L, b, sh.moves[slot][i], c, D.zob, lane);
                c = 3 - c;
            }
            lap(10);
            if (j.expand) expand_node_pipe<S>(L, b, c, D, t, j.child, j.parent, j.edge, j.xseq, sh, lane);
            lap(11);
            write_planes<S>(L, b, c, planes + ((size_t)t * max_leaves + k) * 6 * G::P, lane);
            wave_sync();
            if (lane == 0) {
                pipe_store(&sh.done[k], 1);                                        // releases the node arrays
                pipe_store(&sh.slot_done[slot], k / kMpSlots + 1);
            }
        }
    }
    if (prof && lane == 0)
        for (int i = 0; i < 15; ++i) atomicAdd(reinterpret_cast<unsigned long long *>(D.prof + i), (unsigned long long)pc[i]);
    __syncthreads();
    const bool good = active && !sh.err;
    if (good) {
        // the root's virtual losses (max_leaves descents, one each) reach the pool here
        for (int i = threadIdx.x; i < A; i += NTHR) D.ch_vl[root_base + i] = sh.root_vl[i];
        if (threadIdx.x == 0) D.node[root_ns].vl += max_leaves;
    }
    if (prof && threadIdx.x == 0) D.prof[15] += (long long)__builtin_amdgcn_s_memtime() - t_begin;
    if (threadIdx.x == 0) {
        D.meta[t].num_nodes = sh.num_nodes;
        D.n_leaves[t] = good ? max_leaves : 0;
        set_cursor(D, t, sh.cursor_val);
    }
}

template <int S, int NSEL, int NWRK>
int launch_mpipe_cfg(const SearchDev &dev, int max_leaves, float *planes, hipStream_t st) {
    constexpr size_t lds = sizeof(MPipeShared<S, NSEL, NWRK>);
    static_assert(lds <= 160 * 1024, "LDS");
    static std::atomic<uint64_t> configured{0};          // per device, thread-safe (tg::first_on_device)
    int devid = 0;
    (void)hipGetDevice(&devid);
    if (tg::first_on_device(configured, devid))
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&select_puct_mpipe_kernel<S, NSEL, NWRK>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((select_puct_mpipe_kernel<S, NSEL, NWRK>), dim3(dev.T), dim3(64 * (NSEL + NWRK)), lds, st, dev,
                       max_leaves, planes);
    return TG_OK;
}

template <int S>
int launch_mpipe(const SearchDev &dev, int max_leaves, float *planes, hipStream_t st) {
    // selectors * 100 + workers (TG_MPIPE_CFG: tuning knob).  9x9: 16 wavefronts; 19x19: a worker's board is 25 KB of LDS
    static const int cfg = tg::knob("TG_MPIPE_CFG") ? atoi(tg::knob("TG_MPIPE_CFG")) : 0;
    if constexpr (S == 9) {
        if (cfg == 404) return launch_mpipe_cfg<S, 4, 4>(dev, max_leaves, planes, st);
        if (cfg == 408) return launch_mpipe_cfg<S, 4, 8>(dev, max_leaves, planes, st);
        if (cfg == 412) return launch_mpipe_cfg<S, 4, 12>(dev, max_leaves, planes, st);
        if (cfg == 808) return launch_mpipe_cfg<S, 8, 8>(dev, max_leaves, planes, st);
        return launch_mpipe_cfg<S, 6, 10>(dev, max_leaves, planes, st);
    } else {
        if (cfg == 404) return launch_mpipe_cfg<S, 4, 4>(dev, max_leaves, planes, st);
        if (cfg == 605) return launch_mpipe_cfg<S, 6, 5>(dev, max_leaves, planes, st);
        if (cfg == 806) return launch_mpipe_cfg<S, 8, 6>(dev, max_leaves, planes, st);
        return launch_mpipe_cfg<S, 6, 6>(dev, max_leaves, planes, st);
    }
}

constexpr int kOwnNotYet = -3;        // alloc_child of a descent whose leaf the allocator has not taken yet

__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, lane_partner_i32<0>(v));
    v = min(v, lane_partner_i32<1>(v));
    v = min(v, lane_partner_i32<2>(v));
    v = min(v, lane_partner_i32<3>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ---- PUCT selection of one tree on SEVERAL compute units -----------------------------------------------------------
// Both kernels above keep ~40 k wave-cycles of work per descent on the 16 waves a workgroup can have - on ONE CU -
// and 25 k of it is the workers' board work (path replay, candidates, priors, planes), which needs nothing from the
// selectors but the job.  Here a tree gets 1 + NWG workgroups: the first is the selecting half of the node-owner design (root
// owner, node owners, allocator, plus "shippers" and the draw cursor), the others are nothing but workers, on other
// CUs (consecutive workgroups go to different XCDs).  What crosses between them goes through memory with agent-coherent
// accesses (relaxed agent-scope atomics: sc1 loads and stores, served at the coherence point, no cache maintenance):
//   * shippers -> workers: the job (header, recorded path, moves) in a per-launch array of entries, one store
//     instruction per 64 words, `s_waitcnt vmcnt(0)`, then the entry's tag word.  Tags carry the launch number, so
//     nothing has to be cleared between launches.  All jobs of a launch have their own entry: the selecting half
//     never waits for the workers to free anything.  A store to the coherence point takes longer than the root takes for
//     a descent: NSHIP waves ship, job k on wave k % NSHIP;
//   * workers -> node owners: "node initialised" (only consulted when a descent steps into a node created in the same
//     launch): the node's arrays are ordinary stores, so the worker releases at agent scope (L2 write-back) before
//     it stores the tag, and the owner acquires (invalidate) before it loads the node;
//   * the random-draw cursor: inside ONE workgroup of workers a chain through LDS (NWG = 1); with more, counts in and
//     offsets out through a wave of the selecting half (see the kernel).
// Bounded spins everywhere (a stall is an error, never a hang).  Same trees bit for bit.
// -DTG_SPLIT_PROF (tools/experiments/split_prof.sh): s_memtime accumulators of tree 0 in D.prof - who waits for whom in a launch
// (0 root loop, 1 root's wait for a free slot, 2 allocator loop, 3 allocator's wait, 4 node owner 1 busy, 5 its steps, 6 shipper 0
// loop, 7-10 worker (1, 0): wait for a job / replay / expansion / planes, 12 start of the selecting half (absolute), 13 latest end
// of a worker (absolute), 15 selecting half start to end)
#ifdef TG_SPLIT_PROF
#define SP_NOW() ((long long)__builtin_amdgcn_s_memtime())
#define SP_ON(D, t) ((D).prof != nullptr && (t) == 0)
#else
#define SP_NOW() 0LL
#define SP_ON(D, t) false
#endif
constexpr int kXwHeader = 8;                                   // words: tag, parent, edge, child, expand, xseq, depth, k
template <int S>
constexpr int kXwEntryWords = kXwHeader + kPathCap + kPathMax<S> / 2;
constexpr int kXwMaxTrees = 16;

__device__ __forceinline__ int xw_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xw_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int S, int NNODE>
struct SplitSelShared {
    static constexpr int kSlots = 64;                 // descents inside the selecting half: one mail word per lane
    int16_t moves[kSlots][kPathMax<S>];
    int qpath[kSlots][kPathCap];
    int slot_free[kSlots];            // descents this slot has seen through (shipped to the workers)
    int mail[64];
    int st_node[kSlots], st_depth[kSlots], st_prev[kSlots], st_redge[kSlots];
    int lm_parent[kSlots], lm_edge[kSlots], lm_child[kSlots], lm_depth[kSlots];
    int leaf_ready[kSlots];
    int sp_child[kSlots], sp_expand[kSlots], sp_xseq[kSlots];   // the allocator's part of the job, for the shipper
    int ship_ready[kSlots];           // k + 1 once descent k's job can be shipped
    int alloc_child[kPipeMaxK];
    int exp_key[kPipeMaxK];
    int16_t jobof[kPipeMaxK];
    int num_nodes;
    int nexp_total;                   // expansions of the launch (once all_done)
    int all_done;
    int err;
    // the root owner's step is the chain a launch hangs on: sqrt(N + k + 1) of every descent and the correctly rounded
    // reciprocals of the counts a child can reach are made by the whole workgroup before the chain starts
    double sq[kPipeMaxK];
    double rcp[kRcpN];
    int choice[kPipeMaxK];            // root edge of descent k + 1 (0: not chosen yet): the chooser's only product, the clerk's input
};
template <int S, int NWRK>
struct SplitWrkShared {
    Lds<S, false> board[NWRK];
    int16_t moves[NWRK][kPathMax<S>];
    int cursor_seq;
    long long cursor_val;
    int err;
};

// NWG workgroups of workers.  With more than one, the draw cursor (expansion x takes the n_x draws behind those of
// expansions 0 .. x - 1) cannot be chained through one workgroup's LDS: a worker reports n_x once its candidates are
// counted, a wave of the selecting half adds the counts up in expansion order and hands out the offsets, and the worker
// - which has written its planes meanwhile - fills in the prior when its offset is there.
template <int S, int NNODE, int NWRK, int NSHIP, int NWG>
__global__ __launch_bounds__(1024) void select_puct_split_kernel(SearchDev D, int max_leaves, float *planes, int *xw_job,
                                                                   int *xw_done, int *xw_n, unsigned long long *xw_off,
                                                                   int tag_base, int cap, int test_mute) {
    using G = Geo<S>;
    constexpr int A = G::A;
    constexpr int R = (A + 63) / 64;
    constexpr int EW = kXwEntryWords<S>;
    extern __shared__ __attribute__((aligned(16))) unsigned char xw_smem[];
    const int t = blockIdx.x / (1 + NWG);
    const int role = blockIdx.x % (1 + NWG);          // 0: selecting half, 1 .. NWG: workers
    const bool selecting = role == 0;
    // test hook (TG_SPLIT_TEST_MUTE): the worker workgroups of a tree never show up - the selecting half must run into its
    // bounded waits and report a stalled pipeline (kErrPipeline), not hang (tests/test_gpu_search.py)
    if (test_mute && !selecting) return;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const RootMeta meta = D.meta[t];
    const int n0 = meta.num_nodes;
    const size_t root_ns = (size_t)t * D.N, root_base = root_ns * A;
    const bool active = D.err[t] == 0 && n0 > 0;
    int *const jobs = xw_job + (size_t)t * cap * EW;
    int *const done = xw_done + (size_t)t * cap;
    int *const xn = xw_n + (size_t)t * cap;                                   // candidates of expansion x (tagged)
    unsigned long long *const xoff = xw_off + (size_t)t * (cap + 1);          // draws before expansion x (tagged)
    const long long cursor0 = D.rng_cursor[t];

    const bool sp = SP_ON(D, t) && lane == 0;
    const long long sp_t0 = SP_NOW();
    if (selecting) {
        using Shared = SplitSelShared<S, NNODE>;
        constexpr int kSlots = Shared::kSlots;
        Shared &sh = *reinterpret_cast<Shared *>(xw_smem);
        if (sp && threadIdx.x == 0) D.prof[12] = sp_t0;
        if (threadIdx.x < kSlots) {
            sh.slot_free[threadIdx.x] = 0; sh.leaf_ready[threadIdx.x] = 0; sh.ship_ready[threadIdx.x] = 0; sh.mail[threadIdx.x] = 0;
        }
        for (int i = threadIdx.x; i < kPipeMaxK; i += 1024) { sh.alloc_child[i] = kOwnNotYet; sh.exp_key[i] = -1; sh.choice[i] = 0; }
        if (threadIdx.x == 0) { sh.num_nodes = n0; sh.all_done = 0; sh.err = 0; }
        {
            const int total0 = D.node[root_ns].visits + D.node[root_ns].vl;
            for (int i = threadIdx.x; i < max_leaves; i += 1024) sh.sq[i] = __dsqrt_rn((double)(total0 + i + 1));
            for (int i = threadIdx.x; i < kRcpN; i += 1024) sh.rcp[i] = 1.0 / (double)(i ? i : 1);
        }
        __syncthreads();
        auto fail = [&](int code, int site) {
            if (lane == 0 && !pipe_load(&sh.err)) {
                atomicOr(&D.err[t], code | (site << 8));
                pipe_store(&sh.err, 1);
            }
        };
        // ---- the root.  Every descent passes it, so its step is the chain a launch hangs on.  Two halves:
        //  * the CHOICE - scores of all children for sqrt(N + k + 1), arg-max, one more virtual loss on the winner - is all that is
        //    serial;
        //  * what a choice entails (move, child, leaf or step, the job for the owners / the allocator, slots) is clerical.
        // DUAL (9x9): wave 0 only chooses and leaves the edge in sh.choice[k]; the clerk (wave kClerk, on another SIMD) follows the
        // ring: one wave doing both took ~1 700 cycles per descent = 96 % of a launch, the chooser alone takes ~1 050.  At 19x19 the
        // sixteenth wave is worth more as a tenth node owner (deep trees: the launches hang on the owners): wave 0 does both there.
        constexpr bool DUAL = S == 9;
        constexpr int kClerk = DUAL ? NNODE + 2 + NSHIP + (NWG > 1 ? 1 : 0) : -1;
        int r_vis[R], r_act[R], r_idx[R], r_kref[R], c_vl[R], c_cnt[R], root_nc = 0;
        double r_vsum[R], r_pol[R], c_q[R], c_rcp[R], c_rcpn[R];
        auto chooser_init = [&]() {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = lane + 64 * r, ii = i < A ? i : A - 1;
                r_vsum[r] = D.ch_vsum[root_base + ii];
                r_pol[r] = D.ch_policy[root_base + ii];
                c_cnt[r] = D.ch_visits[root_base + ii] + D.ch_vl[root_base + ii];
                c_q[r] = c_cnt[r] != 0 ? r_vsum[r] / (double)c_cnt[r] : 0.0;
                // reciprocals of count + 1 (this step's denominator) and count + 2 (the next one, should this child be chosen:
                // requested a step ahead, so that no LDS round trip sits between two choices)
                if (__any(c_cnt[r] + 2 >= kRcpN)) { c_rcp[r] = 1.0 / (double)(c_cnt[r] + 1); c_rcpn[r] = 1.0 / (double)(c_cnt[r] + 2); }
                else { c_rcp[r] = sh.rcp[c_cnt[r] + 1]; c_rcpn[r] = sh.rcp[c_cnt[r] + 2]; }
            }
            root_nc = D.node[root_ns].children;
        };
        auto choose = [&](double sq) -> int {
            double best = 0.0;
            int best_i = -1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = lane + 64 * r;
                if (i < root_nc) {
                    double v = c_q[r] + div_by_count(r_pol[r] * sq, (double)(c_cnt[r] + 1), c_rcp[r]);
                    if (D.cgos && i == root_nc - 1) v -= 0.1;
                    if (best_i < 0 || v > best) { best = v; best_i = i; }
                }
            }
            best_i = wave_argmax_first(best, best_i);
            if (best_i < 0) best_i = 0;                                      // (a diverged network: every score NaN)
            return best_i;
        };
        auto chosen = [&](int best_i) {                                      // one more virtual loss on the winner
            const int owner = best_i & 63, oslot = best_i >> 6;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r == oslot) {
                    // the winner's new count is its old denominator: q = value sum / count by the reciprocal it holds
                    const double q = div_by_count(r_vsum[r], (double)(c_cnt[r] + 1), c_rcp[r]);
                    double nn;
                    if (__any(lane == owner && c_cnt[r] + 3 >= kRcpN)) nn = 1.0 / (double)(c_cnt[r] + 3);
                    else nn = sh.rcp[c_cnt[r] + 3 < kRcpN ? c_cnt[r] + 3 : 0];
                    if (lane == owner) { c_q[r] = q; c_cnt[r] += 1; c_rcp[r] = c_rcpn[r]; c_rcpn[r] = nn; }
                }
        };
        auto clerk_init = [&]() {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = lane + 64 * r, ii = i < A ? i : A - 1;
                r_vis[r] = D.ch_visits[root_base + ii];
                r_act[r] = D.action[root_base + ii];
                r_idx[r] = D.ch_index[root_base + ii];
                c_vl[r] = D.ch_vl[root_base + ii];
                r_kref[r] = -1;
            }
        };
        auto clerk_step = [&](int k, int slot, int best_i) -> bool {
            const int owner = best_i & 63, oslot = best_i >> 6;
            int my_move = 0, my_child = 0, my_cnt = 0, my_kref = -1;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r == oslot) { my_move = r_act[r]; my_child = r_idx[r]; my_cnt = r_vis[r] + c_vl[r]; my_kref = r_kref[r]; }
            const int src = __builtin_amdgcn_readfirstlane(owner);
            const int e = best_i;
            const int mv = __builtin_amdgcn_readlane(my_move, src);
            int child = __builtin_amdgcn_readlane(my_child, src);
            const int count = __builtin_amdgcn_readlane(my_cnt, src);
            const int kref = __builtin_amdgcn_readlane(my_kref, src);
            const bool two_pass = meta.moves + 1 > 2 && mv == 0 && meta.prev == 0;   // tree.py:224-229
            const int threshold = two_pass ? 10000000 : 1;
            const bool leaf = count + 1 < threshold + 1;
            if (child == kNotExpanded && kref >= 0) child = -2 - kref;
            const bool expands = leaf && child == kNotExpanded;
            if (!leaf && child == kNotExpanded) return false;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r == oslot && lane == owner) {
                    c_vl[r] += 1;
                    if (expands) r_kref[r] = k;
                }
            if (lane == 0) {
                sh.moves[slot][0] = (int16_t)mv;
                sh.qpath[slot][0] = e;
                if (leaf) {
                    sh.lm_parent[slot] = 0; sh.lm_edge[slot] = e; sh.lm_child[slot] = child; sh.lm_depth[slot] = 1;
                    mp_publish(&sh.leaf_ready[slot], k + 1);
                } else {
                    sh.st_node[slot] = child; sh.st_depth[slot] = 1; sh.st_prev[slot] = mv; sh.st_redge[slot] = e;
                    mp_publish(&sh.mail[slot], ((k + 1) << 8) | (1 + (1 + e) % NNODE));
                }
            }
            return true;
        };
        auto clerk_done = [&](bool ok) {
            if (!ok) fail(kErrPipeline, 1);
            if (ok) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = lane + 64 * r;
                    if (i < A) D.ch_vl[root_base + i] = c_vl[r];
                }
                if (lane == 0) D.node[root_ns].vl += max_leaves;
            }
        };
        if (wid == 0) {
            __builtin_amdgcn_s_setprio(3);
            if (active) {
                chooser_init();
                if constexpr (!DUAL) clerk_init();
                double sq = sh.sq[0];
                bool ok = true;
                long long sp_acc = 0;
                for (int k = 0; k < max_leaves; ++k) {
                    const double sqn = sh.sq[k + 1 < max_leaves ? k + 1 : k];      // (used by the next step)
                    if constexpr (!DUAL) {
                        const long long sp_w = SP_NOW();
                        ok = mp_wait_ge(sh, &sh.slot_free[k % kSlots], k / kSlots);
                        sp_acc += SP_NOW() - sp_w;
                        if (!ok) break;
                    }
                    const int best_i = choose(sq);
                    if constexpr (DUAL) {
                        if (lane == 0) pipe_store(&sh.choice[k], best_i + 1);
                    } else {
                        ok = clerk_step(k, k % kSlots, best_i);
                        if (!ok) break;
                    }
                    chosen(best_i);
                    sq = sqn;
                }
                if (sp) D.prof[14] += SP_NOW() - sp_t0;
                if constexpr (!DUAL) {
                    if (sp) { D.prof[0] += SP_NOW() - sp_t0; D.prof[1] += sp_acc; }
                    clerk_done(ok);
                }
            }
        } else if (DUAL && wid == kClerk) {
            __builtin_amdgcn_s_setprio(3);
            if (active) {
                clerk_init();
                bool ok = true;
                long long sp_acc = 0;
                for (int k = 0; k < max_leaves; ++k) {
                    const int slot = k % kSlots;
                    // the choice and the slot's release in one round trip
                    const long long sp_w = SP_NOW();
                    int ch = 0;
                    ok = false;
                    for (int spin = 0; spin < kPipeSpinLimit; ++spin) {
                        ch = *reinterpret_cast<const volatile int *>(&sh.choice[k]);
                        const int fr = *reinterpret_cast<const volatile int *>(&sh.slot_free[slot]);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (ch >= 1 && fr >= k / kSlots) { ok = true; break; }
                        if (spin >= 32) {
                            if (pipe_load(&sh.err)) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    sp_acc += SP_NOW() - sp_w;
                    if (!ok) break;
                    ok = clerk_step(k, slot, ch - 1);
                    if (!ok) break;
                }
                if (sp) { D.prof[0] += SP_NOW() - sp_t0; D.prof[1] += sp_acc; }
                clerk_done(ok);
            }
        } else if (wid <= NNODE) {
            // ---- owners of the nodes below the root ----------------------------------------------------
            __builtin_amdgcn_s_setprio(2);
            const int me = wid;
            int idle = 0;
            int unf0 = -1, unf1 = -1, unf2 = -1, unf3 = -1, n_unf = 0;
            long long sp_busy = 0, sp_steps = 0, sp_hits = 0;
            int sp_last = -1;
            while (active) {
                const int w = lane < kSlots ? pipe_load(&sh.mail[lane]) : 0;
                const bool mine = (w & 255) == me;
                int nd = mine ? sh.st_node[lane] : 0;
                bool ready = mine;
                bool fresh;
                {
                    const bool pending = mine && nd <= -2;
                    const int c = pipe_load(&sh.alloc_child[pending ? -2 - nd : 0]);
                    if (pending) {
                        if (c == kOwnNotYet) ready = false;
                        else nd = c;
                    }
                    fresh = ready && nd >= n0;                                    // created in this launch
                    if (__any(fresh)) {
                        // initialised by its worker?  (the other workgroup: an agent-coherent load)
                        const int jk = fresh ? (int)sh.jobof[nd - n0] : 0;
                        const int dn = xw_load(&done[jk]);
                        if (fresh && dn != (tag_base | (jk + 1))) ready = false;
                    }
                }
                const int kmin = wave_min_i32(ready ? (w >> 8) - 1 : 0x7fffffff);
                const int k = kmin == 0x7fffffff ? -1 : kmin;
                const int slot = k >= 0 ? k % kSlots : 0;
                const int node = __builtin_amdgcn_readlane(nd, slot);
                const bool was_fresh = __builtin_amdgcn_readlane((int)fresh, slot) != 0;
                if (k < 0) {
                    if (pipe_load(&sh.all_done) || pipe_load(&sh.err)) break;
                    if (++idle > kPipeSpinLimit / 16) { fail(kErrPipeline, 2); break; }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                idle = 0;
                const long long sp_b = SP_NOW();
                const int depth = sh.st_depth[slot], prev = sh.st_prev[slot], redge = sh.st_redge[slot];
                if (depth >= kPathMax<S>) { fail(kErrPipeline, 3); break; }
                if (was_fresh) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");              // the worker's stores, not a stale line
                    unf0 = unf1 = unf2 = unf3 = -1;
                    n_unf = 0;
                } else if (node == unf0 || node == unf1 || node == unf2 || node == unf3 || n_unf >= 4) {
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    unf0 = unf1 = unf2 = unf3 = -1;
                    n_unf = 0;
                } else {
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
                if (n_unf == 0) unf0 = node;
                else if (n_unf == 1) unf1 = node;
                else if (n_unf == 2) unf2 = node;
                else unf3 = node;
                ++n_unf;
                const EdgePick pick = select_puct<S>(D, t, node, lane, sh.rcp);
                const int e = pick.edge, mv = pick.move;
                const size_t ns = (size_t)t * D.N + node, base = ns * A;
                const bool two_pass = meta.moves + depth + 1 > 2 && mv == 0 && prev == 0;   // tree.py:224-229
                const int threshold = two_pass ? 10000000 : 1;
                const bool leaf = pick.count + 1 < threshold + 1;
                int child = pick.child;
                const int key = (node << 10) | e;
                if (child == kNotExpanded && pick.count >= 1) {
                    int kref = -1;
                    for (int b0 = 0; b0 < k && kref < 0; b0 += 64) {
                        const unsigned long long hit = __ballot(b0 + lane < k && sh.exp_key[b0 + lane] == key);
                        if (hit) kref = b0 + __ffsll((long long)hit) - 1;
                    }
                    if (kref < 0) { fail(kErrPipeline, 4); break; }
                    child = -2 - kref;
                }
                if (lane == 0) {
                    D.node[ns].vl = pick.node_vl + 1;                                   // node.py:76-83
                    D.ch_vl[base + e] = pick.edge_vl + 1;
                    sh.moves[slot][depth] = (int16_t)mv;
                    if (depth < kPathCap) sh.qpath[slot][depth] = key;
                    if (leaf) {
                        if (child == kNotExpanded) sh.exp_key[k] = key;
                        sh.lm_parent[slot] = node; sh.lm_edge[slot] = e; sh.lm_child[slot] = child; sh.lm_depth[slot] = depth + 1;
                        sh.mail[slot] = 0;
                        mp_publish(&sh.leaf_ready[slot], k + 1);
                    } else {
                        sh.st_node[slot] = child; sh.st_depth[slot] = depth + 1; sh.st_prev[slot] = mv;
                        mp_publish(&sh.mail[slot], ((k + 1) << 8) | (1 + (depth + 1 + redge) % NNODE));
                    }
                }
                wave_sync();
                sp_busy += SP_NOW() - sp_b;
                sp_steps += 1;
                sp_hits += (node == sp_last && !was_fresh);
                sp_last = node;
            }
#ifndef TG_BACKUP_PROF
            if (sp && me == 1) { D.prof[4] += sp_busy; D.prof[5] += sp_steps; }
#ifdef TG_SPLIT_PROF_HITS      // all owners: steps in slot 11, steps on the node the owner handled last in slot 14
            if (sp) { atomicAdd(reinterpret_cast<unsigned long long *>(D.prof + 11), (unsigned long long)sp_steps); atomicAdd(reinterpret_cast<unsigned long long *>(D.prof + 14), (unsigned long long)sp_hits); }
#endif
#endif
        } else if (wid == NNODE + 1) {
            // ---- the allocator: leaves in descent order (LDS only: what later descents may be waiting for) ----
            int num_nodes = n0, nexp = 0;
            bool ok = active;
            long long sp_acc = 0;
            for (int k = 0; ok && k < max_leaves; ++k) {
                const int slot = k % kSlots;
                const long long sp_w = SP_NOW();
                int child = 0;
                ok = mp_wait_with(sh, &sh.leaf_ready[slot], k + 1, &sh.lm_child[slot], child);
                sp_acc += SP_NOW() - sp_w;
                if (!ok) { fail(kErrPipeline, 5); break; }
                if (child <= -2) child = sh.alloc_child[-2 - child];
                const int expand = child == kNotExpanded;
                int xseq = 0;
                if (expand) {
                    if (num_nodes >= D.N || num_nodes - n0 >= kPipeMaxK) { fail(kErrPoolFull, 6); ok = false; break; }
                    child = num_nodes++;
                    xseq = nexp++;
                }
                // (the parent's child index in memory is the shipper's store: behind ship_ready, i.e. behind jobof)
                if (lane == 0) {
                    if (expand) sh.jobof[child - n0] = (int16_t)k;
                    sh.sp_child[slot] = child; sh.sp_expand[slot] = expand; sh.sp_xseq[slot] = xseq;
                    mp_publish(&sh.alloc_child[k], child);
                    mp_publish(&sh.ship_ready[slot], k + 1);
                }
            }
            if (lane == 0) {
                sh.num_nodes = num_nodes;
                sh.nexp_total = nexp;
                if (ok) mp_publish(&sh.all_done, 1);
            }
#ifndef TG_BACKUP_PROF
            if (sp) { D.prof[2] += SP_NOW() - sp_t0; D.prof[3] += sp_acc; }
#endif
        } else if (wid < NNODE + 2 + NSHIP) {
            // ---- shippers: job k to the workers' workgroup, on wave k % NSHIP (a store to the coherence point takes
            //      longer than the root takes for a descent: several jobs are under way at a time) ----
            const int j = wid - (NNODE + 2);
            for (int k = j; active && k < max_leaves; k += NSHIP) {
                const int slot = k % kSlots;
                if (!mp_wait_ge(sh, &sh.ship_ready[slot], k + 1)) { fail(kErrPipeline, 8); break; }
                const int depth = sh.lm_depth[slot];
                int *const entry = jobs + (size_t)k * EW;
                {
                    // header (words 1..7), recorded path, moves (two per word): one store per 64 words
                    int word = 0;
                    if (lane == 1) word = sh.lm_parent[slot];
                    else if (lane == 2) word = sh.lm_edge[slot];
                    else if (lane == 3) word = sh.sp_child[slot];
                    else if (lane == 4) word = sh.sp_expand[slot];
                    else if (lane == 5) word = sh.sp_xseq[slot];
                    else if (lane == 6) word = depth;
                    else if (lane == 7) word = k;
                    else if (lane >= kXwHeader && lane < kXwHeader + kPathCap) word = sh.qpath[slot][lane - kXwHeader];
                    if (lane >= 1 && lane < kXwHeader + kPathCap) xw_store(&entry[lane], word);
                    // a new node becomes findable from its parent (jobof has been in LDS since before ship_ready)
                    if (lane == 4 && word) D.ch_index[((size_t)t * D.N + sh.lm_parent[slot]) * A + sh.lm_edge[slot]] = sh.sp_child[slot];
                    for (int jj = lane; 2 * jj < depth; jj += 64) {
                        const int lo = (unsigned short)sh.moves[slot][2 * jj];
                        const int hi = 2 * jj + 1 < depth ? (unsigned short)sh.moves[slot][2 * jj + 1] : 0;
                        xw_store(&entry[kXwHeader + kPathCap + jj], lo | (hi << 16));
                    }
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): the entry is written ...
                if (lane == 0) {
                    xw_store(&entry[0], tag_base | (k + 1));                         // ... before its tag
                    mp_publish(&sh.slot_free[slot], k / kSlots + 1);
                }
            }
            if (sp && j == 0) D.prof[6] += SP_NOW() - sp_t0;
        }
        if (NWG > 1 && wid == NNODE + 2 + NSHIP && active) {
            // ---- the draw cursor: counts in, offsets out, in expansion order ----
            unsigned off = 0;
            bool ok = true;
            if (lane == 0) __hip_atomic_store(&xoff[0], ((unsigned long long)(unsigned)tag_base << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int x = 0; ok; ++x) {
                int v = 0;
                bool have = false;
                for (int spin = 0; spin < kPipeSpinLimit / 16; ++spin) {
                    if (x < cap) {
                        v = xw_load(&xn[x]);
                        if ((v & ~2047) == tag_base && (v & 2047) != 0) { have = true; break; }
                    }
                    if (pipe_load(&sh.all_done) && x >= sh.nexp_total) break;   // every expansion has been through
                    if (pipe_load(&sh.err)) { ok = false; break; }
                    if ((spin & 63) == 63 && xw_load(&D.err[t])) {
                        // a worker workgroup gave up (it has reported why): stop this workgroup's owners too, instead of
                        // letting them spin to their limits on nodes that will never be initialised
                        if (lane == 0) pipe_store(&sh.err, 1);
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!have) {
                    if (ok && !(pipe_load(&sh.all_done) && x >= sh.nexp_total)) { fail(kErrPipeline, 9); ok = false; }
                    break;
                }
                off += (unsigned)((v & 2047) - 1);
                if (lane == 0 && x + 1 <= cap)
                    __hip_atomic_store(&xoff[x + 1], ((unsigned long long)(unsigned)tag_base << 32) | off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (ok && lane == 0) set_cursor(D, t, cursor0 + (long long)off);
        }
        __syncthreads();
        const bool good = active && !sh.err;
        if (threadIdx.x == 0) {
            D.meta[t].num_nodes = sh.num_nodes;
            D.n_leaves[t] = good ? max_leaves : 0;
            if (sp) D.prof[15] += SP_NOW() - sp_t0;
        }
    } else {
        // ---- the workers' workgroup: job k on wave k % NWRK ----------------------------------------------------
        using Shared = SplitWrkShared<S, NWRK>;
        Shared &sh = *reinterpret_cast<Shared *>(xw_smem);
        if (threadIdx.x == 0) { sh.cursor_seq = 0; sh.cursor_val = D.rng_cursor[t]; sh.err = 0; }
        __syncthreads();
        if (wid < NWRK) {
            const int w = wid;
            Lds<S, false> &L = sh.board[w];
            BoardScalars rootb;
            int root_to_move;
            load_root<S>(L, rootb, root_to_move, D, t, lane);
            long long sp_a[4] = {0, 0, 0, 0};
            for (int k = (role - 1) * NWRK + w; active && k < max_leaves; k += NWG * NWRK) {
                const int *const entry = jobs + (size_t)k * EW;
                bool have = false;
                long long sp_x = SP_NOW();
                for (int spin = 0; spin < kPipeSpinLimit / 16; ++spin) {
                    if (xw_load(&entry[0]) == (tag_base | (k + 1))) { have = true; break; }
                    if ((spin & 63) == 63 && (xw_load(&D.err[t]) || pipe_load(&sh.err))) break;   // the other half gave up
                    __builtin_amdgcn_s_sleep(2);
                }
                if (!have) {
                    if (lane == 0 && !xw_load(&D.err[t])) atomicOr(&D.err[t], kErrPipeline | (7 << 8));
                    if (lane == 0) pipe_store(&sh.err, 1);
                    break;
                }
                { const long long n = SP_NOW(); sp_a[0] += n - sp_x; sp_x = n; }
                const int hw = lane < kXwHeader + kPathCap ? xw_load(&entry[lane]) : 0;
                const int parent = __builtin_amdgcn_readlane(hw, 1), edge = __builtin_amdgcn_readlane(hw, 2);
                const int child = __builtin_amdgcn_readlane(hw, 3), expand = __builtin_amdgcn_readlane(hw, 4);
                const int xseq = __builtin_amdgcn_readlane(hw, 5), depth = __builtin_amdgcn_readlane(hw, 6);
                for (int j = lane; 2 * j < depth; j += 64) {
                    const int mw = xw_load(&entry[kXwHeader + kPathCap + j]);
                    sh.moves[w][2 * j] = (int16_t)(mw & 0xffff);
                    if (2 * j + 1 < depth) sh.moves[w][2 * j + 1] = (int16_t)((unsigned)mw >> 16);
                }
                {
                    // queue entry of leaf k (what the backup reads)
                    const size_t qs = (size_t)t * D.K + k;
                    const int npath = depth < kPathCap ? depth : kPathCap;
                    if (lane >= kXwHeader && lane < kXwHeader + npath) D.q_path[qs * kPathCap + lane - kXwHeader] = hw;
                    if (lane == 0) {
                        D.q_node[qs] = child;
                        D.q_pnode[qs] = parent;
                        D.q_pedge[qs] = edge;
                        D.q_depth[qs] = (depth <= kPathCap && D.N <= (1 << 21)) ? depth : 0;
                    }
                }
                wave_sync();
                reset_work<S>(L, lane);
                BoardScalars b = rootb;
                int c = root_to_move;
                for (int i = 0; i < depth; ++i) {
                    put_stone<S>(L, b, sh.moves[w][i], c, D.zob, lane);
                    c = 3 - c;
                }
                { const long long n = SP_NOW(); sp_a[1] += n - sp_x; sp_x = n; }
                if constexpr (NWG == 1) {
                    if (expand) expand_node_pipe<S>(L, b, c, D, t, child, parent, edge, xseq, sh, lane);
                    write_planes<S>(L, b, c, planes + ((size_t)t * max_leaves + k) * 6 * G::P, lane);
                } else {
                    int n = 0;
                    if (expand) {
                        n = gen_candidates<S>(L, b, c, D, lane);
                        if (lane == 0) xw_store(&xn[xseq], tag_base | (n + 1));
                    }
                    { const long long n = SP_NOW(); sp_a[2] += n - sp_x; sp_x = n; }
                    write_planes<S>(L, b, c, planes + ((size_t)t * max_leaves + k) * 6 * G::P, lane);
                    { const long long n = SP_NOW(); sp_a[3] += n - sp_x; sp_x = n; }
                    if (expand) {
                        unsigned long long ov = 0;
                        bool got = false;
                        for (int spin = 0; spin < kPipeSpinLimit / 16; ++spin) {
                            ov = __hip_atomic_load(&xoff[xseq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((unsigned)(ov >> 32) == (unsigned)tag_base) { got = true; break; }
                            if ((spin & 63) == 63 && (xw_load(&D.err[t]) || pipe_load(&sh.err))) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        const long long cur = cursor0 + (long long)(unsigned)ov;
                        if (!got || cur + n > D.rng_cap) {
                            if (lane == 0) {
                                if (!got) { if (!xw_load(&D.err[t])) atomicOr(&D.err[t], kErrPipeline | (10 << 8)); }
                                else atomicOr(&D.err[t], kErrRngEmpty);
                                pipe_store(&sh.err, 1);
                            }
                            break;
                        }
                        expand_fill<S>(L, D, t, child, parent, edge, n, cur, lane);
                    }
                }
                wave_sync();
                if (expand) {
                    // the node's arrays reach memory before its "initialised" tag does (a descent of this launch may enter it)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (lane == 0) xw_store(&done[k], tag_base | (k + 1));
                }
                { const long long n = SP_NOW(); sp_a[2] += n - sp_x; sp_x = n; }
            }
            if (sp && role == 1 && w == 0) { D.prof[7] += sp_a[0]; D.prof[8] += sp_a[1]; D.prof[9] += sp_a[2]; D.prof[10] += sp_a[3]; }
            if (sp) atomicMax(reinterpret_cast<unsigned long long *>(D.prof + 13), (unsigned long long)SP_NOW());
        }
        __syncthreads();
        if (NWG == 1 && threadIdx.x == 0) set_cursor(D, t, sh.cursor_val);
    }
}

// tree.py:273-315 process_mini_batch for the leaves queued by the preceding kernel.
// Policies (node.py:86-93 update_policy via the {pos: policy} map of tree.py:287-295) are independent per leaf: the
// eight waves of a tree take the leaves round-robin.  Values walk leaf -> root in leaf order, and the float32
// accumulation order is part of the contract - per NODE.  Leaves below different root children touch disjoint nodes,
// so when every leaf carries its recorded path (q_depth > 0: the pipelined selectors) the leaves are shared out by
// root edge: wave w backs up, in leaf order, the leaves whose root edge is w mod 8 - every (node, edge) still sees
// its additions in leaf order.  What all leaves share is the root NODE's float32 value sum: each wave leaves its
// leaves' root-level values in LDS and one lane adds them up in leaf order at the end.  (One wave doing all leaves one
// after the other was 0.12-0.19 ms per 256-leaf mini-batch of a single tree and 0.13 ms per Gumbel phase.)
// Leaves without a recorded path (serial selectors, paths deeper than kPathCap) keep the one-wave walk.
constexpr int kBackupCap = 1024;        // leaves per tree the partitioned walk has LDS for

// NWAVE waves per tree: 8 when the trees crowd the CUs, 16 for a few trees (the leaves per wave are what a launch takes)
template <int S, int NWAVE>
__global__ __launch_bounds__(64 * NWAVE) void backup_kernel(SearchDev D, const float *policy, const float *value,
                                                     int stride, const int32_t *leaf_off, int use_logit) {
    using G = Geo<S>;
    constexpr int A = G::A, W = G::W, P = G::P;
    constexpr int NTHR = 64 * NWAVE;
    const int t = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n = D.n_leaves[t];
    const size_t leaf_base = leaf_off ? (size_t)leaf_off[t] : (size_t)t * stride;
    // the root's statistics are touched by every leaf: they live in LDS for the whole launch
    // (256 dependent read-modify-writes through L2 otherwise) and are written back at the end
    __shared__ double r_vsum[A];
    __shared__ int r_vis[A], r_vl[A];
    __shared__ float r_nvsum;
    __shared__ int r_nvis, r_nvl;
    __shared__ uint16_t leaf_edge[kBackupCap];     // root edge of leaf k; | 0x8000 once backed up WITHOUT a parent (nothing for the root)
    __shared__ float rootv[kBackupCap];            // its value at the root's level
    __shared__ uint8_t leaf_neg[kBackupCap];       // leaf k names the reference's node[-1] (q_node < 0)
    const size_t rbase = (size_t)t * D.N * A;                     // root = node 0
#ifdef TG_BACKUP_PROF       // tools/experiments/split_prof.sh: D.prof 2 set-up, 3 policies (slowest wave), 4 values (slowest wave), 5 whole kernel
    const bool bp = D.prof != nullptr && t == 0 && lane == 0;
    const long long bp_t0 = (long long)__builtin_amdgcn_s_memtime();
    __shared__ unsigned long long bp_max[2];
    if (threadIdx.x < 2) bp_max[threadIdx.x] = 0;
#define BP_MARK(i) do { if (bp) atomicMax(&bp_max[i], (unsigned long long)__builtin_amdgcn_s_memtime()); } while (0)
#else
#define BP_MARK(i) do { } while (0)
#endif
    bool part = n > 0 && n <= kBackupCap;
    if (n > 0) {
        for (int i = threadIdx.x; i < A; i += NTHR) {
            r_vsum[i] = D.ch_vsum[rbase + i];
            r_vis[i] = D.ch_visits[rbase + i];
            r_vl[i] = D.ch_vl[rbase + i];
        }
        if (threadIdx.x == 0) {
            r_nvsum = D.node[(size_t)t * D.N].vsum;
            r_nvis = D.node[(size_t)t * D.N].visits;
            r_nvl = D.node[(size_t)t * D.N].vl;
        }
        if (part)
            for (int k = threadIdx.x; k < n; k += NTHR) {
                const size_t slot = (size_t)t * D.K + k;
                leaf_neg[k] = D.q_node[slot] < 0;
                if (D.q_depth[slot] > 0 && D.q_pnode[slot] >= 0) leaf_edge[k] = (uint16_t)(D.q_path[slot * kPathCap] & 1023);
                else part = false;
            }
    }
    part = __syncthreads_and(part) != 0;      // (also: every wave has read the leaf count before anybody resets it)
#ifdef TG_BACKUP_PROF
    const long long bp_t1 = (long long)__builtin_amdgcn_s_memtime();
#endif
    {
        // policies.  Each leaf is a chain of four dependent loads (queue entry, child count, actions, policy values).
        // (Gumbel leaves all name the reference's node[-1], tree.py:412-416: the last pool slot, which has no children
        // unless the pool is full - an error - so nothing is written for them and their order is immaterial.)
        // Four leaves of a wave at a time, every stage of the chain for all four before the next stage: the round
        // trips of the four overlap (one leaf after the other: 32 leaves x 4 round trips = 80 us of a 256-leaf launch).
        // Round 5: all leaves of a wave at once where the registers allow (16 at 9x9, 6 at 19x19 instead of four), and the
        // child count and the actions - both indexed by the node - in ONE stage: three round trips per pass, and a
        // 256-leaf launch of one tree is a single pass (25 -> 8 us).
        constexpr int RP = (A + 63) / 64, U = RP <= 2 ? 16 : 6;
        for (int k0 = wid; k0 < n; k0 += U * NWAVE) {
            int node[U];
            bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * NWAVE;
                live[u] = k < n;
                node[u] = live[u] ? D.q_node[(size_t)t * D.K + k] : 0;
            }
            int nc[U], pos[U][RP];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (node[u] < 0) node[u] = D.N - 1;
                nc[u] = live[u] ? D.node[(size_t)t * D.N + node[u]].children : 0;
#pragma unroll
                for (int r = 0; r < RP; ++r) {
                    const int i = lane + 64 * r;
                    pos[u][r] = live[u] && i < A ? (int)D.action[((size_t)t * D.N + node[u]) * A + i] : -1;
                }
            }
            float pv[U][RP];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float *pol = policy + (leaf_base + k0 + u * NWAVE) * A;
#pragma unroll
                for (int r = 0; r < RP; ++r) {
                    if (lane + 64 * r >= nc[u]) pos[u][r] = -1;
                    const int ps = pos[u][r];
                    pv[u][r] = 0.f;
                    if (ps == 0) {
                        pv[u][r] = pol[P];
                        if (use_logit) pv[u][r] = pv[u][r] - 0.5f;
                    } else if (ps > 0) {
                        pv[u][r] = pol[(ps / W - 1) * S + (ps % W) - 1];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < RP; ++r)
                    if (pos[u][r] >= 0) D.ch_policy[((size_t)t * D.N + node[u]) * A + lane + 64 * r] = (double)pv[u][r];
        }
    }
    BP_MARK(0);
    if (part) {
        // values, leaves shared out by root edge.  Lane i handles level i of a leaf's recorded path (all loads of all
        // levels are independent: one memory round trip per leaf); the value at level j above the leaf edge is the
        // reference's iterated float32 `value = 1.0 - value`.
        // A wave's leaves are a three-stage pipeline (round 5: one leaf after the other was two dependent round trips each -
        // 60-80 us per 256-leaf launch of one tree, most of whose leaves hang below one root child, i.e. on one wave):
        //   M  queue entry, path and network value of the leaf four places ahead are requested,
        //   S  the statistics of the leaf two places ahead (its metadata has arrived) are requested,
        //   U  the leaf in front is applied and stored.
        // The statistics of stage S were requested BEFORE the two leaves in front of it were applied: a lane that meets the
        // same edge (node) again takes the values it has just computed instead of the loaded ones - a node always sits at
        // the same level, i.e. in the same lane, so the lane's own last two updates are all that can be stale (its older
        // stores precede the loads in program order).  Same additions in the same order on every (node, edge).
        {
            struct Meta { int k, node, depth, entry; float v0, v1, v2; };
            struct Stats { double vs; int cv, cl; float ns; int nv, nl; };
            struct Fwd { long long ekey; int nkey; double vs; int cv, cl; float ns; int nv, nl; };
            int blk = 0;
            unsigned long long todo = 0;
            auto next_k = [&]() -> int {                 // this wave's leaves in leaf order, -1 when there are no more
                while (!todo) {
                    if (blk >= n) return -1;
                    const int kk = blk + lane;
                    todo = __ballot(kk < n && (int)(leaf_edge[kk < n ? kk : 0] % NWAVE) == wid);
                    blk += 64;
                }
                const int k = blk - 64 + __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                return k;
            };
            auto fetch = [&](int k) {
                Meta m;
                m.k = k;
                if (k >= 0) {
                    const size_t slot = (size_t)t * D.K + k;
                    m.node = D.q_node[slot];
                    m.depth = D.q_depth[slot];
                    m.entry = lane < kPathCap ? D.q_path[slot * kPathCap + lane] : 0;
                    const float *val = value + (leaf_base + k) * 3;
                    m.v0 = val[0]; m.v1 = val[1]; m.v2 = val[2];
                } else {
                    m.node = 0; m.depth = 0; m.entry = 0; m.v0 = m.v1 = m.v2 = 0.f;
                }
                return m;
            };
            auto request = [&](const Meta &m) {
                Stats st{0.0, 0, 0, 0.f, 0, 0};
                if (m.k >= 0 && lane >= 1 && lane < m.depth) {
                    const size_t cs = (size_t)t * D.N + (m.entry >> 10), ce = cs * A + (m.entry & 1023);
                    st.vs = D.ch_vsum[ce]; st.cv = D.ch_visits[ce]; st.cl = D.ch_vl[ce];
                    st.ns = D.node[cs].vsum; st.nv = D.node[cs].visits; st.nl = D.node[cs].vl;
                }
                return st;
            };
            Fwd f1{-1, -1, 0.0, 0, 0, 0.f, 0, 0}, f2 = f1;              // this lane's last and last-but-one update
            Meta m0 = fetch(next_k()), m1 = fetch(next_k()), m2 = fetch(next_k()), m3 = fetch(next_k());
            Stats s0 = request(m0), s1 = request(m1);
            while (m0.k >= 0) {
                const Meta m4 = fetch(next_k());                                    // M
                const Stats s2 = request(m2);                                       // S
                {                                                                   // U
                    const int k = m0.k, depth = m0.depth;
                    // (a Gumbel leaf with q_node < 0 is the reference's node[-1] quirk: every such leaf writes the LAST pool slot's
                    // raw value, the last one in leaf order wins - that one write is done below by wave 0, in leaf order, instead
                    // of by whichever wave comes last here)
                    if (lane == 0 && m0.node >= 0) D.node[(size_t)t * D.N + m0.node].raw = m0.v1 * 0.5f + m0.v2;   // tree.py:299
                    const float vleaf = m0.v0 + m0.v1 * 0.5f;   // tree.py:302
                    Fwd nf{-1, -1, 0.0, 0, 0, 0.f, 0, 0};
                    if (lane < depth) {
                        const int pn = m0.entry >> 10, pe = m0.entry & 1023;
                        float v = vleaf;
                        for (int q = depth - 1 - lane; q > 0; --q) v = 1.0f - v;
                        const size_t cs = (size_t)t * D.N + pn;
                        const size_t ce = cs * A + pe;
                        if (lane == depth - 1) D.ch_value[ce] = (double)vleaf;   // set_leaf_value
                        if (lane == 0) {                                         // level 0 is the root: LDS copy
                            r_vsum[pe] = (double)((float)r_vsum[pe] + v);
                            r_vis[pe] += 1;
                            r_vl[pe] -= 1;
                            rootv[k] = v;
                        } else {
                            double vs = s0.vs;
                            int cv = s0.cv, cl = s0.cl;
                            float ns_ = s0.ns;
                            int nv = s0.nv, nl = s0.nl;
                            const long long ekey = (long long)ce;
                            if (f1.ekey == ekey) { vs = f1.vs; cv = f1.cv; cl = f1.cl; }
                            else if (f2.ekey == ekey) { vs = f2.vs; cv = f2.cv; cl = f2.cl; }
                            if (f1.nkey == pn) { ns_ = f1.ns; nv = f1.nv; nl = f1.nl; }
                            else if (f2.nkey == pn) { ns_ = f2.ns; nv = f2.nv; nl = f2.nl; }
                            nf.ekey = ekey; nf.nkey = pn;
                            nf.vs = (double)((float)vs + v);                     // float32 accumulation (file header)
                            nf.cv = cv + 1; nf.cl = cl - 1;
                            nf.ns = ns_ + v; nf.nv = nv + 1; nf.nl = nl - 1;
                            D.ch_vsum[ce] = nf.vs;
                            D.ch_visits[ce] = nf.cv;
                            D.ch_vl[ce] = nf.cl;
                            D.node[cs].vsum = nf.ns;
                            D.node[cs].visits = nf.nv;
                            D.node[cs].vl = nf.nl;
                        }
                    }
                    f2 = f1;
                    f1 = nf;
                }
                m0 = m1; m1 = m2; m2 = m3; m3 = m4;
                s0 = s1; s1 = s2;
                wave_sync();
            }
        }
        BP_MARK(1);
        __syncthreads();
        if (wid == 0) {
            // the root node's float32 value sum: every leaf's root-level value, in leaf order
            float acc = r_nvsum;
            for (int k0 = 0; k0 < n; k0 += 64) {
                const float mine = k0 + lane < n ? rootv[k0 + lane] : 0.f;
                const int cnt = min(64, n - k0);
                for (int j = 0; j < cnt; ++j)
                    acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), j));
            }
            if (lane == 0) {
                r_nvsum = acc;
                r_nvis += n;
                r_nvl -= n;
            }
            // node[-1]: the last leaf (in leaf order) without a node of its own decides the last slot's raw value
            int last = -1;
            for (int k0 = 0; k0 < n; k0 += 64) {
                const int kk = k0 + lane;
                const unsigned long long m = __ballot(kk < n && leaf_neg[kk < n ? kk : 0] != 0);      // (from the set-up pass: no round trip per 64 leaves here)
                if (m) last = k0 + 63 - __clzll((long long)m);
            }
            if (last >= 0 && lane == 0) {
                const float *val = value + (leaf_base + last) * 3;
                D.node[(size_t)t * D.N + D.N - 1].raw = val[1] * 0.5f + val[2];
            }
        }
    } else if (wid == 0 && n > 0) {
        // one wave, all leaves in order.  A leaf whose root->leaf path was recorded by the selector (q_depth > 0) is
        // backed up with ONE memory round trip: lane i handles level i of the path; a node always sits at the same
        // level, i.e. in the same lane, so updates of a node shared by consecutive leaves stay in program order.
        // Other leaves follow the parent pointers in lane 0.  The next leaf's scalars are requested ahead.
        const size_t slot0 = (size_t)t * D.K;
        int node = D.q_node[slot0], cur = D.q_pnode[slot0], e = D.q_pedge[slot0], depth = D.q_depth[slot0];
        int entry = lane < kPathCap ? D.q_path[slot0 * kPathCap + lane] : 0;
        const float *val = value + leaf_base * 3;
        float v0 = val[0], v1 = val[1], v2 = val[2];
        for (int k = 0; k < n; ++k) {
            // request the next leaf's scalars before backing this one up
            const int kn = k + 1 < n ? k + 1 : k;
            const size_t slot_n = (size_t)t * D.K + kn;
            const int node_n = D.q_node[slot_n], cur_n = D.q_pnode[slot_n], e_n = D.q_pedge[slot_n];
            const int depth_n = D.q_depth[slot_n];
            const int entry_n = lane < kPathCap ? D.q_path[slot_n * kPathCap + lane] : 0;
            const float *val_n = value + (leaf_base + kn) * 3;
            const float v0_n = val_n[0], v1_n = val_n[1], v2_n = val_n[2];
            if (node < 0) node = D.N - 1;
            if (lane == 0) D.node[(size_t)t * D.N + node].raw = v1 * 0.5f + v2;   // tree.py:299
            if (cur >= 0) {
                const float vleaf = v0 + v1 * 0.5f;   // tree.py:302
                if (depth > 0) {
                    if (lane < depth) {
                        const int pn = entry >> 10, pe = entry & 1023;
                        float v = vleaf;
                        for (int q = depth - 1 - lane; q > 0; --q) v = 1.0f - v;
                        const size_t cs = (size_t)t * D.N + pn;
                        const size_t ce = cs * A + pe;
                        if (lane == depth - 1) D.ch_value[ce] = (double)vleaf;   // set_leaf_value
                        if (pn == 0) {                                           // root: LDS copy
                            r_vsum[pe] = (double)((float)r_vsum[pe] + v);
                            r_vis[pe] += 1;
                            r_vl[pe] -= 1;
                            r_nvsum += v;
                            r_nvis += 1;
                            r_nvl -= 1;
                        } else {
                            const double vs = D.ch_vsum[ce];
                            const int cv = D.ch_visits[ce], cl = D.ch_vl[ce];
                            const float ns_ = D.node[cs].vsum;
                            const int nv = D.node[cs].visits, nl = D.node[cs].vl;
                            D.ch_vsum[ce] = (double)((float)vs + v);         // float32 accumulation (file header)
                            D.ch_visits[ce] = cv + 1;
                            D.ch_vl[ce] = cl - 1;
                            D.node[cs].vsum = ns_ + v;
                            D.node[cs].visits = nv + 1;
                            D.node[cs].vl = nl - 1;
                        }
                    }
                } else if (lane == 0) {
                    float v = vleaf;
                    D.ch_value[((size_t)t * D.N + cur) * A + e] = (double)v;   // set_leaf_value
                    while (cur > 0) {
                        const size_t cs = (size_t)t * D.N + cur;
                        const size_t ce = cs * A + e;
                        // all loads of a level first (independent, one round trip), then the stores
                        const double vs = D.ch_vsum[ce];
                        const int cv = D.ch_visits[ce], cl = D.ch_vl[ce];
                        const float ns_ = D.node[cs].vsum;
                        const int nv = D.node[cs].visits, nl = D.node[cs].vl;
                        const int pe = D.node[cs].pedge, pn = D.node[cs].parent;
                        D.ch_vsum[ce] = (double)((float)vs + v);
                        D.ch_visits[ce] = cv + 1;
                        D.ch_vl[ce] = cl - 1;
                        D.node[cs].vsum = ns_ + v;
                        D.node[cs].visits = nv + 1;
                        D.node[cs].vl = nl - 1;
                        v = 1.0f - v;
                        e = pe;
                        cur = pn;
                    }
                    if (cur == 0) {                                          // root: LDS copy
                        r_vsum[e] = (double)((float)r_vsum[e] + v);
                        r_vis[e] += 1;
                        r_vl[e] -= 1;
                        r_nvsum += v;
                        r_nvis += 1;
                        r_nvl -= 1;
                    }
                }
                wave_sync();
            }
            node = node_n; cur = cur_n; e = e_n; depth = depth_n; entry = entry_n;
            v0 = v0_n; v1 = v1_n; v2 = v2_n;
        }
    }
    __syncthreads();
    if (n > 0) {
        for (int i = threadIdx.x; i < A; i += NTHR) {
            D.ch_vsum[rbase + i] = r_vsum[i];
            D.ch_visits[rbase + i] = r_vis[i];
            D.ch_vl[rbase + i] = r_vl[i];
        }
        if (threadIdx.x == 0) {
            D.node[(size_t)t * D.N].vsum = r_nvsum;
            D.node[(size_t)t * D.N].visits = r_nvis;
            D.node[(size_t)t * D.N].vl = r_nvl;
        }
    }
    if (threadIdx.x == 0) D.n_leaves[t] = 0;
#ifdef TG_BACKUP_PROF
    if (bp && threadIdx.x == 0) {
        D.prof[2] += bp_t1 - bp_t0;
        D.prof[3] += (long long)bp_max[0] - bp_t1;
        D.prof[4] += (long long)bp_max[1] - (long long)bp_max[0];
        D.prof[5] += (long long)__builtin_amdgcn_s_memtime() - bp_t0;
    }
#endif
}


// numpy's float64 add.reduce order (pairwise_sum in numpy/core/src/umath/loops_utils.h.src):
// < 8 elements sequential; <= 128 elements eight interleaved accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) then the tail; larger arrays split in halves (multiple
// of 8).  The vector sits in LDS; the result is wave-uniform.  All 64 lanes call it with uniform arguments.
// Accumulator j only ever sees a[j], a[8 + j], a[16 + j], ... in that order, so lane j (< 8) owns it: its up to
// 15 LDS reads are independent and issued together, eight chains of dependent adds run side by side, the combine
// is three DPP steps (fp addition commutes exactly, so the mirrored partners give the same sums).
__device__ double np_sum_block(const double *a, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    const int lane = threadIdx.x & 63;
    const int n8 = n - (n % 8);
    const int j = lane & 7;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = 8 * k + j;
        v[k] = a[i < n8 ? i : j];                     // clamped: never past the vector
    }
    double r = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k)
        if (8 * k < n8) r += v[k];                    // wave-uniform
    r = r + lane_partner_f64<0>(r);                   // r0 + r1 | r2 + r3 | r4 + r5 | r6 + r7
    r = r + lane_partner_f64<1>(r);                   // (r0 + r1) + (r2 + r3) | (r4 + r5) + (r6 + r7)
    r = r + lane_partner_f64<2>(r);                   // lane i <-> 7 - i inside every group of eight
    double res = read_lane_f64(r, 0);
    for (int i = n8; i < n; ++i) res += a[i];
    return res;
}
__device__ double np_sum(const double *a, int n) {
    if (n <= 128) return np_sum_block(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    const int nr = n - n2;
    // one more level is enough for n <= 512 (A <= 362)
    double left, right;
    if (n2 <= 128) left = np_sum_block(a, n2);
    else { int m = n2 / 2; m -= m % 8; left = np_sum_block(a, m) + np_sum_block(a + m, n2 - m); }
    if (nr <= 128) right = np_sum_block(a + n2, nr);
    else { int m = nr / 2; m -= m % 8; right = np_sum_block(a + n2, m) + np_sum_block(a + n2 + m, nr - m); }
    return left + right;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double ov = __shfl_xor(v, o); v = ov > v ? ov : v; }
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int ov = __shfl_xor(v, o); v = ov > v ? ov : v; }
    return v;
}

// node.py:324-346 select_move_by_sequential_halving_for_root
template <int S>
__device__ int select_root_halving(const SearchDev &D, int t, int node, int count_threshold, int lane) {
    constexpr int A = Geo<S>::A;
    constexpr int R = (A + 63) / 64;
    const size_t ns = (size_t)t * D.N + node;
    const size_t base = ns * A;
    // one memory round trip: every lane requests its slots of all five arrays up front
    int vis[R], vl[R];
    double vsum[R], logit[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        const int ii = i < A ? i : A - 1;
        vis[r] = D.ch_visits[base + ii];
        vl[r] = D.ch_vl[base + ii];
        vsum[r] = D.ch_vsum[base + ii];
        logit[r] = D.ch_policy[base + ii] + D.noise[(size_t)t * A + ii];
    }
    const int nc = D.node[ns].children;
    int mx = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (lane + 64 * r < nc) mx = max(mx, vis[r]);
    mx = wave_max_i32(mx);
    const double sigma = (double)(50 + mx) * 1.0;                   // (C_VISIT + max) * C_SCALE
    double best = 0.0;
    int best_i = -1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        if (i < nc) {
            const int v = vis[r];
            const int cnt = v + vl[r];
            const double q = v > 0 ? vsum[r] / (double)v : 0.0;
            const double sc = cnt >= count_threshold ? -10000.0 : logit[r] + sigma * q;
            if (best_i < 0 || sc > best) { best = sc; best_i = i; }
        }
    }
    best_i = wave_argmax_first(best, best_i);
    return best_i;
}

template <int S, typename Scratch>
__device__ int select_node_halving(Scratch &L, const SearchDev &D, int t, int node, int lane) {   // uses L.w1, L.w2
    constexpr int A = Geo<S>::A;
    constexpr int R = (A + 63) / 64;
    const size_t ns = (size_t)t * D.N + node;
    const size_t base = ns * A;
    const int nc = D.node[ns].children;
    const int nv = D.node[ns].visits;
    const double raw = (double)D.node[ns].raw;
    double logit[R], q[R];
    int vis[R];
    double mx = -INFINITY;
    int maxv = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        logit[r] = 0.0; q[r] = 0.0; vis[r] = 0;
        if (i < nc) {
            logit[r] = D.ch_policy[base + i];
            vis[r] = D.ch_visits[base + i];
            q[r] = vis[r] > 0 ? D.ch_vsum[base + i] / (double)vis[r] : 0.0;
            mx = logit[r] > mx ? logit[r] : mx;
            maxv = max(maxv, vis[r]);
        }
    }
    maxv = wave_max_i32(maxv);
    double mixed;
    if (nv == 0 && maxv == 0) {
        // A node nobody has been below (every node a phase creates, when its first descent arrives): all q are 0, so
        // v_pi is a sum of zeros and mixed = (raw + (0 * 0) / sum_prob) / (0 + 1) = raw bit for bit - the prior's softmax
        // (an exp pass, three pairwise sums, a division pass) is not needed to know that.
        mixed = raw;
    } else {
        mx = wave_max_f64(mx);
        wave_sync();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane + 64 * r;
            if (i < nc) L.w1[i] = exp(logit[r] - mx);
        }
        wave_sync();
        const double s1 = np_sum(L.w1, nc);
        wave_sync();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane + 64 * r;
            if (i < nc) {
                const double pi = L.w1[i] / s1;
                L.w1[i] = pi;
                L.w2[i] = pi * q[r];
            }
        }
        wave_sync();
        const double sum_prob = np_sum(L.w1, nc);
        const double v_pi = np_sum(L.w2, nc);
        mixed = (raw + ((double)nv * v_pi) / sum_prob) / ((double)nv + 1.0);
    }
    const double sigma = (double)(50 + maxv) * 1.0;
    double il[R];
    double mx2 = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        il[r] = 0.0;
        if (i < nc) {
            il[r] = logit[r] + sigma * (vis[r] > 0 ? q[r] : mixed);
            mx2 = il[r] > mx2 ? il[r] : mx2;
        }
    }
    mx2 = wave_max_f64(mx2);
    wave_sync();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        if (i < nc) L.w1[i] = exp(il[r] - mx2);
    }
    wave_sync();
    const double s2 = np_sum(L.w1, nc);
    double best = 0.0;
    int best_i = -1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        if (i < nc) {
            const double sc = L.w1[i] / s2 - ((double)vis[r] / (1.0 + (double)nv));
            if (best_i < 0 || sc > best) { best = sc; best_i = i; }
        }
    }
    wave_sync();
    best_i = wave_argmax_first(best, best_i);
    return best_i;
}

// ---- Gumbel / sequential-halving selection: one workgroup per tree, a selector wave + NW workers -----------
// (root: node.py:324-346, below: :349-361).  Two modes inside one kernel:
//  * the usual one (round 6; `la_*` below): the selector ranks the root and works out which root children the phase enters
//    ("entries") and which descents repeat them; every wave walks its share of the entries; the selector numbers the new nodes;
//    a worker takes whole entries - expansion, the step into the new node, the leaf - off one board replay; the repeats of all
//    entries are shared out over the workers;
//  * one by one, through the job ring of select_puct_pipe_kernel (any path longer than the per-entry buffers, or the test hook
//    SearchDev::gumbel_one_by_one): the selector walks and queues LEAF (replay the path, write the planes of leaf `plane_slot`)
//    and EXPAND jobs (replay, expand the child it is about to enter - it then waits for exactly that job, because it continues
//    INTO the new node), the workers share the repeats out afterwards as plane copies.
// 9x9 (2 / 6 / 10 workers), 13x13 (2 / 6) and 19x19 (2 / 4: a worker's board is 19 KB there).
template <int S>
struct HalvingScratch {
    double w1[Geo<S>::A + 7];
    double w2[Geo<S>::A + 7];
};

template <int S, int NW>
__global__ __launch_bounds__(64 * (1 + NW)) void select_gumbel_pipe_kernel(SearchDev D, const int32_t *num_considered,
                                                                 const int32_t *max_count, int stride,
                                                                 const int32_t *leaf_off, float *planes) {
    using G = Geo<S>;
    constexpr int A = G::A;
    constexpr int kPipeSlots = PipeShared<S, NW>::kSlots;      // (shadows the three-wave kernels' ring size)
    constexpr int NTHR = 64 * (1 + NW);
    __shared__ PipeShared<S, NW> sh;
    __shared__ HalvingScratch<S> hs;
    __shared__ int16_t sel_moves[kPathMax<S>];
    __shared__ int sel_path[kPathMax<S>];
    // Leaves by root child.  ~100 descents of a phase go to 2..16 root children and all descents through one root
    // child end on the same leaf (nothing moves within a phase).  The root choices of the whole phase are simulated up
    // front (ballots on a copy of the counters): a root child's FIRST descent gets an entry f (rm_pos = its root child,
    // rm_slot = its leaf slot), every later descent only `sched[leaf slot] = f`.  Only the first descents are walked
    // (rm_* completed: leaf, path); the scheduled leaves are shared out among the workers - queue entry, path, planes written
    // from the entry's position code in LDS (one-by-one mode: when the ring has drained, virtual losses per leaf, planes copied
    // from the first leaf's slot) - no per-leaf work of the selector, which was the slowest wave of a phase (2.2 k cycles per
    // repeated descent, 87 of 100).
    constexpr int kRootMemo = 24, kRootPath = 32;        // (a phase enters at most 16 + 1 root children)
    __shared__ int rm_pos[kRootMemo], rm_parent[kRootMemo], rm_edge[kRootMemo], rm_child[kRootMemo], rm_job[kRootMemo],
        rm_slot[kRootMemo], rm_depth[kRootMemo];
    __shared__ int rm_path[kRootMemo][kRootPath];
    __shared__ int8_t sched[kPipeMaxK / 2];
    __shared__ int bulk_n;
    // The look-ahead walks (below: "Expansions first") of a phase's root children are independent of each other - disjoint
    // subtrees, statistics that do not move - and the workers have nothing to do until the first job is queued: every wave of
    // the workgroup walks its share (entry f on wave f mod (1 + NW), its own softmax scratch), the selector then hands out node
    // numbers, draws and jobs in entry order as before.  (The walks were 112 k of the selector's 194 k cycles per phase.)
    __shared__ HalvingScratch<S> hsw[NW];
    __shared__ int la_ready, la_done, la_n;
    __shared__ int la_mv[kRootMemo], la_vis[kRootMemo], la_child[kRootMemo], la_e[kRootMemo];       // entry f at the root
    __shared__ int la_res[kRootMemo], la_node[kRootMemo], la_edge[kRootMemo], la_chd[kRootMemo], la_depth[kRootMemo];   // where its walk ended
    __shared__ int16_t la_moves[kRootMemo][kRootPath];
    // ... and the first descents themselves are the workers' too: entry f - its expansion (node number and place in the draw
    // order given by the selector), the step into the node just made, the leaf's queue entry, virtual losses and planes - is one
    // piece of work on one board replay, entry f on worker f mod NW; the selector queues nothing.  (One by one through the job
    // ring - EXPAND job, wait, step, LEAF job - the first descents were 60 k of the selector's 123 k cycles per phase.)
    __shared__ int la_go, la_fast, la_ndesc, la_xseq[kRootMemo];
    // an entry's leaf position for its repeats, one byte per point (colour as the side to move sees it | 4: the previous move) +
    // (pass plane, side): the repeats are shared out over ALL workers once their entry's leaf stands (the best root child's ~50
    // repeats of a late phase were one worker's: 25 k of its 46 k cycles)
    __shared__ uint8_t ent_code[kRootMemo][(G::P + 3) & ~3];
    __shared__ int ent_meta[kRootMemo], ent_done[kRootMemo];
    constexpr int kWalkLeaf = 1, kWalkExpand = 2, kWalkDeep = 3, kWalkPoolFull = 4;
    const int t = blockIdx.x;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const RootMeta meta = D.meta[t];
    const int n0 = meta.num_nodes;
    if (threadIdx.x < kPipeSlots) {
        sh.job_seq[threadIdx.x] = 0;
        sh.slot_done[threadIdx.x] = 0;
    }
    for (int i = threadIdx.x; i < kPipeMaxK / 32; i += NTHR) sh.done_bits[i] = 0u;
    if (threadIdx.x < kRootMemo) ent_done[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        la_ready = 0; la_done = 0; la_n = 0; la_go = 0; la_fast = 0;
        sh.cursor_seq = 0;
        sh.cursor_val = D.rng_cursor[t];
        sh.final_count = -1;
        sh.err = 0;
    }
    __syncthreads();
    const int width = num_considered[t], levels = max_count[t];
    const size_t leaf_base = leaf_off ? (size_t)leaf_off[t] : (size_t)t * stride;
    const bool active = D.err[t] == 0 && n0 > 0 && width * levels <= stride;
    int num_nodes = n0;
    int queued = 0;

    // s_memtime accumulators of tree 0's selector (tg_search_profile; tools/profile_gumbel.py): 0 set-up (root into
    // registers, ranking), 1 the phase's root choices, 8 nodes and EXPAND jobs handed out, 7 this wave's share of the entries'
    // walks, 3 waiting for the other waves' shares, 2 first descents, 4 waiting for a free job slot (inside 1..2), 5 write-back,
    // 6 first descents (count)
    const bool prof = D.prof && t == 0;
    const long long t_begin = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
    long long tp = t_begin, pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto lap = [&](int i) {
        if (prof) { const long long now = (long long)__builtin_amdgcn_s_memtime(); pc[i] += now - tp; tp = now; }
    };
    // entry f's walk from its root child down to where the first descent will end - a leaf, or the point of expansion - without
    // side effects; paths longer than the per-entry buffers are left to the selector's one-by-one code (kWalkDeep)
    auto walk = [&](int f, HalvingScratch<S> &scratch) {
        int node = 0, depth = 0, res = kWalkDeep;
        int mv = la_mv[f], visits = la_vis[f], child = la_child[f], e = la_e[f];
        while (depth < (D.gumbel_one_by_one ? 0 : kRootPath - 2)) {
            if (lane == 0) { la_moves[f][depth] = (int16_t)mv; rm_path[f][depth] = (node << 10) | e; }
            ++depth;
            if (visits < 1) { res = kWalkLeaf; break; }
            if (child == kNotExpanded) { res = kWalkExpand; break; }
            if (child >= n0) break;                                       // (created in this launch: cannot happen, subtrees are disjoint)
            node = child;
            e = select_node_halving<S>(scratch, D, t, node, lane);
            const size_t nb = ((size_t)t * D.N + node) * A;
            mv = D.action[nb + e];
            visits = D.ch_visits[nb + e];
            child = D.ch_index[nb + e];
        }
        if (lane == 0) { la_res[f] = res; la_node[f] = node; la_edge[f] = e; la_chd[f] = child; la_depth[f] = depth; }
    };
    if (wid == 0) {
        // ---- selector -------------------------------------------------------------------
        // Within a phase nothing backs values up, so (a) the root's score "logit + noise + sigma q" of every child
        // is a constant - only "visits + virtual loss >= threshold" moves, and the virtual loss is added by this
        // wave: the root lives in registers for the whole launch (loaded once, written back once); (b) below the
        // root the choice (node.py:349-361: completed Q, two softmaxes - the costly part) is a pure function of
        // statistics that do not change: it is computed at a node's first visit and remembered (LDS, by node).
        // Virtual losses below the root are fire-and-forget atomics (nobody reads them before the backup).
        constexpr int R = (A + 63) / 64;
        int jid = 0, nexp = 0;
        bool ok = active;
        double r_score[R];
        int r_cnt[R], r_vis[R], r_idx[R], r_act[R], r_vl0[R];
        int r_nc = 0, r_added = 0;
        if (active) {
            const size_t base = (size_t)t * D.N * A;
            double vsum[R];
            r_nc = D.node[(size_t)t * D.N].children;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = lane + 64 * r, ii = i < A ? i : A - 1;
                r_vis[r] = D.ch_visits[base + ii];
                r_vl0[r] = D.ch_vl[base + ii];
                vsum[r] = D.ch_vsum[base + ii];
                r_score[r] = D.ch_policy[base + ii] + D.noise[(size_t)t * A + ii];
                r_idx[r] = D.ch_index[base + ii];
                r_act[r] = D.action[base + ii];
            }
            int mx = 0;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (lane + 64 * r < r_nc) mx = max(mx, r_vis[r]);
            mx = wave_max_i32(mx);
            const double sigma = (double)(50 + mx) * 1.0;                   // (C_VISIT + max) * C_SCALE
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double q = r_vis[r] > 0 ? vsum[r] / (double)r_vis[r] : 0.0;
                r_score[r] = r_score[r] + sigma * q;
                r_cnt[r] = r_vis[r] + r_vl0[r];
            }
        }
        // The scores are constants of the launch, so "arg-max over the children still under the threshold" is "the
        // first such child in score order": the children are ranked once (score descending, index ascending - the
        // order np.argmax resolves ties in) and every per-child register array is permuted into rank order.  A root
        // choice is then a compare, a ballot and a find-first-bit instead of a float64 arg-max over the wave
        // (1.4 k cycles of the 6.6 k a descent cost the selector).  When no child is under the threshold every
        // score is -10000 and np.argmax returns child 0: pos0 is where that child sits.
        int r_edge[R];
        int pos0 = 0;
        if (active) {
            wave_sync();
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (lane + 64 * r < A) hs.w1[lane + 64 * r] = r_score[r];
            wave_sync();
            int rank[R];
#pragma unroll
            for (int r = 0; r < R; ++r) rank[r] = 0;
            for (int j = 0; j < r_nc; ++j) {
                const double sj = hs.w1[j];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = lane + 64 * r;
                    rank[r] += (sj > r_score[r] || (sj == r_score[r] && j < i)) ? 1 : 0;
                }
            }
            __shared__ int ptmp[A + 1];
            auto permute = [&](int (&v)[R]) {
                wave_sync();
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (lane + 64 * r < r_nc) ptmp[rank[r]] = v[r];
                wave_sync();
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = lane + 64 * r < r_nc ? ptmp[lane + 64 * r] : 0;
            };
#pragma unroll
            for (int r = 0; r < R; ++r) r_edge[r] = lane + 64 * r;
            permute(r_edge);
            permute(r_cnt);
            permute(r_vis);
            permute(r_idx);
            permute(r_act);
            // rank of child 0
            wave_sync();
            if (lane == 0) ptmp[A] = rank[0];
            wave_sync();
            pos0 = ptmp[A];
        }
        // remembered choices below the root: tag = node, value = (edge, move, visits of the edge, child)
        constexpr int kMemo = 64;
        __shared__ int memo_tag[kMemo], memo_edge[kMemo], memo_move[kMemo], memo_vis[kMemo], memo_child[kMemo];
        for (int i = lane; i < kMemo; i += 64) memo_tag[i] = -1;
        // (leaves by root child: rm_* / sched above.  The network still evaluates every queued leaf.)
        int n_first = 0, n_desc = 0, n_iter = 0;             // entries so far; descents so far; threshold levels walked one by one
        wave_sync();
        auto publish = [&](int plane_slot, int parent, int edge, int child, int expand, int xseq, int depth, int src,
                           const int *path_src = nullptr, const int16_t *moves_src = nullptr) -> bool {
            if (jid >= kPipeMaxK) return false;
            const int slot = jid % kPipeSlots;
            const long long w0 = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
            if (!pipe_wait_ge(&sh.slot_done[slot], jid / kPipeSlots)) return false;
            if (prof) pc[4] += (long long)__builtin_amdgcn_s_memtime() - w0;
            if (path_src && moves_src) {                         // an entry's recorded walk
                for (int i = lane; i < depth; i += 64) { sh.moves[slot][i] = moves_src[i]; sh.paths[slot][i] = path_src[i]; }
            } else if (path_src) {                               // COPY job: the worker needs the path only
                for (int i = lane; i < depth; i += 64) sh.paths[slot][i] = path_src[i];
            } else {
                for (int i = lane; i < depth; i += 64) { sh.moves[slot][i] = sel_moves[i]; sh.paths[slot][i] = sel_path[i]; }
            }
            wave_sync();
            if (lane == 0) {
                PipeJob &j = sh.job[slot];
                j.k = plane_slot; j.parent = parent; j.edge = edge; j.child = child;
                j.expand = expand; j.xseq = xseq; j.depth = depth; j.src = src;
                pipe_store(&sh.job_seq[slot], jid + 1);
            }
            wave_sync();
            ++jid;
            return true;
        };
        // Expansions first.  Every root child is entered several times in a phase, and only its FIRST descent can meet
        // a node that has to be expanded (the later ones find it there; the new node's children are unvisited, so
        // the descent ends right below it).  Waiting for each of those expansions in turn (replay + candidates +
        // prior: ~5 us) was a third of a launch.  So the root choices of the phase are simulated on a copy of the
        // counters (ballots: cheap), and each root child, at its first appearance, is walked down WITHOUT side
        // effects to the point of expansion: the node is allocated and its EXPAND job queued - node numbers and
        // draws in descent order, exactly as before - and nobody waits.  The descents proper then find the children
        // allocated and wait, if at all, for a job that is about to finish.  Subtrees of different root children are
        // disjoint and statistics are constant within a phase, so every choice is what the one-by-one order makes.
        lap(0);
        // ---- the root choices of the phase, simulated on a copy of the counters: entries (first appearances) and the schedule ----
        // A choice is "the first child in rank order whose count is under the threshold, child 0 if there is none" and adds one to
        // that child's count: a threshold level hands its `width` descents out greedily in rank order - child i takes
        // min(th - count_i, what is left), whatever remains goes to child 0 - so a level is a prefix sum over the wave, not
        // `width` ballots in a row (one by one: 58 k of the selector's 156 k cycles per phase).
        if (ok) {
            int s_cnt[R], f_of[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { s_cnt[r] = r_cnt[r]; f_of[r] = -1; }
            const int owner0 = pos0 & 63, rr0 = pos0 >> 6;
            for (int th = 1; ok && th <= levels; ++th) {
                const int qbase = n_desc;
                ++n_iter;
                int need[R], pre[R], take[R];
                int running = 0;
                bool multi = false;                                       // a child more than one descent under the threshold
                const unsigned long long below = (1ull << lane) - 1ull;
                int under_before = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    need[r] = (lane + 64 * r < r_nc && s_cnt[r] < th) ? th - s_cnt[r] : 0;
                    // (every child under the threshold takes at least one descent: only the first `width` of them can take any -
                    // the never-visited children further down the ranking, `th` under it, stay out of the sums)
                    const unsigned long long under = __ballot(need[r] > 0);
                    if (under_before + __popcll(under & below) >= width) need[r] = 0;
                    under_before += __popcll(under);
                    multi = multi || __ballot(need[r] > 1) != 0ull;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (!multi) {                                         // (the usual level: the prefix sum is a bit count)
                        const unsigned long long m = __ballot(need[r] == 1);
                        pre[r] = running + __popcll(m & below);
                        running += __popcll(m);
                    } else {
                        const int incl = wave_scan_add_i32(need[r]);
                        pre[r] = running + incl - need[r];
                        running += __builtin_amdgcn_readlane(incl, 63);
                    }
                }
                const int given = running < width ? running : width;
                const int leftover = width - given;                       // descents that find no child under the threshold: child 0
                int max_take = 0;
                unsigned long long fresh[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int room = width - pre[r];
                    take[r] = room <= 0 ? 0 : (need[r] < room ? need[r] : room);
                    s_cnt[r] += take[r];
                    max_take = max(max_take, take[r]);
                    fresh[r] = __ballot(take[r] > 0 && f_of[r] < 0);
                }
                max_take = multi ? wave_max_i32(max_take) : (given > 0 ? 1 : 0);
                // entries: the children that appear for the first time, in the order of their first descents
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool mine = take[r] > 0 && f_of[r] < 0;
                    if (mine) f_of[r] = n_first + __popcll(fresh[r] & below);
                    n_first += __popcll(fresh[r]);
                    if (mine && f_of[r] < kRootMemo) {
                        const int f = f_of[r];
                        rm_pos[f] = lane + 64 * r; rm_slot[f] = qbase + pre[r];
                        la_mv[f] = r_act[r]; la_vis[f] = r_vis[r]; la_child[f] = r_idx[r]; la_e[f] = r_edge[r];
                    }
                }
                if (leftover > 0) {
                    int f0 = -1;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (r == rr0) f0 = __builtin_amdgcn_readlane(f_of[r], owner0);
                    const bool fresh0 = f0 < 0;
                    if (fresh0) f0 = n_first++;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (r == rr0 && lane == owner0) {
                            s_cnt[r] += leftover;
                            f_of[r] = f0;
                            if (fresh0 && f0 < kRootMemo) {
                                rm_pos[f0] = pos0; rm_slot[f0] = qbase + given;
                                la_mv[f0] = r_act[r]; la_vis[f0] = r_vis[r]; la_child[f0] = r_idx[r]; la_e[f0] = r_edge[r];
                            }
                        }
                    for (int k = lane; k < leftover; k += 64) sched[qbase + given + k] = (int8_t)((fresh0 && k == 0) ? -1 : f0);
                }
                if (n_first > kRootMemo) { ok = false; break; }          // (never: <= 17 root children per phase)
                // the schedule: descent q repeats the leaf of entry sched[q] (-1: it IS the entry's first descent)
                for (int k = 0; k < max_take; ++k)
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (k < take[r]) sched[qbase + pre[r] + k] = (int8_t)((k == 0 && ((fresh[r] >> lane) & 1ull)) ? -1 : f_of[r]);
                n_desc += width;
                // Steady state: `width` children stand exactly at the threshold and nothing ranked before the last of them can
                // come under a later one - every remaining level gives each of them one descent, in the same order.
                if (leftover == 0 && th < levels) {
                    unsigned long long tk[R];
                    int last = -1, ntk = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        tk[r] = __ballot(take[r] > 0);
                        if (tk[r]) last = 64 * r + 63 - __clzll((long long)tk[r]);
                        ntk += __popcll(tk[r]);
                    }
                    bool steady = ntk == width;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int i = lane + 64 * r;
                        const bool fine = i > last || i >= r_nc || (take[r] > 0 ? s_cnt[r] == th : s_cnt[r] >= levels);
                        steady = steady && __ballot(!fine) == 0ull;
                    }
                    if (steady) {
                        const int rem = levels - th;
                        int before = 0;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            if (take[r] > 0) {
                                const int ord = before + __popcll(tk[r] & below);
                                for (int l = 0; l < rem; ++l) sched[n_desc + l * width + ord] = (int8_t)f_of[r];
                                s_cnt[r] += rem;
                            }
                            before += __popcll(tk[r]);
                        }
                        n_desc += rem * width;
                        break;
                    }
                }
            }
            // every descent adds one virtual loss to its root child (node.py:76-83): the simulated counters are the final ones
#pragma unroll
            for (int r = 0; r < R; ++r) r_cnt[r] = s_cnt[r];
            r_added = n_desc;
        }
        wave_sync();
        lap(1);
        // ---- the entries' walks, shared out over the workgroup's waves (this one takes its share) ----
        if (active) {
            const int n_walk = ok ? n_first : 0;
            if (lane == 0) { la_n = n_walk; pipe_store(&la_ready, 1); }
            for (int f = 0; f < n_walk; f += 1 + NW) walk(f, hs);
            wave_sync();
            if (lane == 0) __hip_atomic_fetch_add(&la_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            lap(7);
            if (!pipe_wait_ge(&la_done, 1 + NW)) ok = false;
        }
        lap(3);
        // (a path longer than the per-entry buffers anywhere in the phase: every entry goes the one-by-one way through the job
        // ring - expansions of both kinds in one phase would wait for each other's place in the draw order across the two queues)
        bool fast = ok;
        if (ok) {
            if (__ballot(lane < n_first && la_res[lane] == kWalkDeep)) {
                fast = false;
                if (lane < n_first) la_res[lane] = kWalkDeep;
                wave_sync();
            }
        }
        // One entry, one by one (paths longer than the per-entry buffers): walked down WITHOUT side effects to the point of
        // expansion, the node allocated and its EXPAND job queued.
        auto lookahead_serial = [&](int f) {
            const int pos = rm_pos[f];
            const int owner = pos & 63, rr = pos >> 6;
            int node = 0, depth = 0;
            int mv = la_mv[f], visits = la_vis[f], child = la_child[f], e = la_e[f];
            while (ok) {
                if (depth >= kPathMax<S>) break;                  // the descent proper reports it
                if (lane == 0) { sel_moves[depth] = (int16_t)mv; sel_path[depth] = (node << 10) | e; }
                ++depth;
                wave_sync();
                if (visits < 1) break;                            // ends on a leaf: nothing to expand
                if (child == kNotExpanded) {
                    if (num_nodes >= D.N || num_nodes - n0 >= kPipeMaxK) break;   // reported by the descent proper
                    child = num_nodes++;
                    if (node == 0) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (r == rr && lane == owner) r_idx[r] = child;
                    } else if (lane == 0) {
                        memo_child[node & (kMemo - 1)] = child;
                    }
                    if (lane == 0) {
                        D.ch_index[((size_t)t * D.N + node) * A + e] = child;
                        sh.jobof[child - n0] = (int16_t)jid;
                    }
                    wave_sync();
                    ok = publish(-1, node, e, child, 1, nexp++, depth, -1);
                    break;                                        // its children are unvisited: a leaf follows
                }
                node = child;
                if (node >= n0) break;                            // (created in this launch: cannot happen, subtrees are disjoint)
                const int slot = node & (kMemo - 1);
                if (memo_tag[slot] != node) {
                    e = select_node_halving<S>(hs, D, t, node, lane);
                    const size_t nb = ((size_t)t * D.N + node) * A;
                    mv = D.action[nb + e];
                    visits = D.ch_visits[nb + e];
                    child = D.ch_index[nb + e];
                    wave_sync();
                    if (lane == 0) {
                        memo_tag[slot] = node; memo_edge[slot] = e; memo_move[slot] = mv;
                        memo_vis[slot] = visits; memo_child[slot] = child;
                    }
                    wave_sync();
                } else {
                    e = memo_edge[slot]; mv = memo_move[slot]; visits = memo_vis[slot]; child = memo_child[slot];
                }
            }
        };
        // Expansions first.  Every root child is entered several times in a phase, and only its FIRST descent can meet
        // a node that has to be expanded (the later ones find it there; the new node's children are unvisited, so
        // the descent ends right below it).  Waiting for each of those expansions in turn (replay + candidates +
        // prior: ~5 us) was a third of a launch.  So every entry has been walked down to its point of expansion (above):
        // here the nodes are allocated and the EXPAND jobs queued - node numbers and draws in descent order, exactly as the
        // one-by-one order makes them - and nobody waits.  The descents proper then find the children allocated and wait, if
        // at all, for a job that is about to finish.  Subtrees of different root children are disjoint and statistics are
        // constant within a phase, so every choice is what the one-by-one order makes.
        if (active && !fast && lane == 0) { la_fast = 0; pipe_store(&la_go, 1); }      // (the workers: straight to the job ring)
        if (!fast) {
            for (int f = 0; ok && f < n_first; ++f) lookahead_serial(f);
        } else {
            // lane f: entry f.  Node numbers and places in the draw order go to the entries that expand, in entry order; when
            // the pool runs out the rest are marked (reported where the one-by-one order reports them).
            const int res = lane < n_first ? la_res[lane] : 0;
            const unsigned long long xm = __ballot(res == kWalkExpand);
            const int ord = __popcll(xm & ((1ull << lane) - 1ull));
            const int room_pool = D.N - num_nodes, room_launch = kPipeMaxK - (num_nodes - n0);
            const int room = max(0, min(room_pool, room_launch));
            if (res == kWalkExpand) {
                if (ord >= room) {
                    la_res[lane] = kWalkPoolFull;
                } else {
                    const int child = num_nodes + ord;
                    la_chd[lane] = child;
                    la_xseq[lane] = nexp + ord;
                    D.ch_index[((size_t)t * D.N + la_node[lane]) * A + la_edge[lane]] = child;
                }
            }
            const int granted = min(__popcll(xm), room);
            num_nodes += granted;
            nexp += granted;
        }
        wave_sync();
        if (active && fast && lane == 0) { la_fast = ok ? 1 : 0; la_ndesc = n_desc; pipe_store(&la_go, 1); }
        lap(8);
        // One entry's first descent, one by one (see lookahead_serial).
        auto descend_serial = [&](int f) {
            {
                const int my_q = rm_slot[f];
                int node = 0, depth = 0, root_pos = 0;
                while (ok) {
                    const size_t ns = (size_t)t * D.N + node;
                    const size_t base = ns * A;
                    int e, mv, visits, child;
                    if (node == 0) {
                        // node.py:324-346: the root child this entry was made for (rank-ordered register copy)
                        const int pos = rm_pos[f];
                        const int owner = pos & 63, rr = pos >> 6;
                        int m_mv = 0, m_vis = 0, m_idx = 0, m_e = 0;
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (r == rr) { m_mv = r_act[r]; m_vis = r_vis[r]; m_idx = r_idx[r]; m_e = r_edge[r]; }
                        mv = __builtin_amdgcn_readlane(m_mv, owner);
                        visits = __builtin_amdgcn_readlane(m_vis, owner);
                        child = __builtin_amdgcn_readlane(m_idx, owner);
                        e = __builtin_amdgcn_readlane(m_e, owner);
                        root_pos = pos;
                    } else {
                        const int slot = node & (kMemo - 1);
                        if (memo_tag[slot] == node) {
                            e = memo_edge[slot]; mv = memo_move[slot]; visits = memo_vis[slot]; child = memo_child[slot];
                        } else {
                            if (node >= n0) ok = pipe_wait_done(sh, sh.jobof[node - n0]);   // expansion in flight?
                            if (!ok) break;
                            e = select_node_halving<S>(hs, D, t, node, lane);
                            mv = D.action[base + e];
                            visits = D.ch_visits[base + e];
                            child = D.ch_index[base + e];
                            wave_sync();
                            if (lane == 0) {
                                memo_tag[slot] = node; memo_edge[slot] = e; memo_move[slot] = mv;
                                memo_vis[slot] = visits; memo_child[slot] = child;
                            }
                            wave_sync();
                        }
                        // (node.py:76-83: the virtual losses below the root are added by the worker that takes the leaf
                        // job - nobody reads them before the backup)
                    }
                    if (depth >= kPathMax<S>) { ok = false; break; }
                    if (lane == 0) {
                        sel_moves[depth] = (int16_t)mv;
                        sel_path[depth] = (node << 10) | e;
                    }
                    ++depth;
                    wave_sync();
                    if (visits < 1) {                                     // tree.py:412-416
                        // the queue entry of this leaf (node to evaluate - still NOT_EXPANDED: node[-1] -, parent, edge,
                        // path) is written by the worker that takes the job
                        const int leaf_jid = jid;
                        if (depth <= kRootPath) {
                            // the later descents through this root child: left to the workers (sched)
                            if (lane == 0) { rm_parent[f] = node; rm_edge[f] = e; rm_child[f] = child; rm_job[f] = leaf_jid; rm_depth[f] = depth; }
                            if (lane < depth) rm_path[f][lane] = sel_path[lane];
                            wave_sync();
                            ok = publish(my_q, node, e, child, 0, 0, depth, -1);
                        } else {
                            // a path too long for rm_path: the later descents become COPY jobs here (planes of this leaf's slot)
                            ok = publish(my_q, node, e, child, 0, 0, depth, -1);
                            for (int qq = 0; ok && qq < n_desc; ++qq)
                                if (sched[qq] == f) {
                                    ok = publish(qq, node, e, child, 2, leaf_jid, depth, my_q);
                                    if (lane == 0) sched[qq] = (int8_t)-1;
                                }
                            wave_sync();
                        }
                        lap(2);
                        if (prof) pc[6] += 1;
                        break;
                    }
                    if (child == kNotExpanded) {                          // tree.py:418-420
                        if (num_nodes >= D.N || num_nodes - n0 >= kPipeMaxK) {
                            if (lane == 0) atomicOr(&D.err[t], kErrPoolFull);
                            ok = false;
                            break;
                        }
                        child = num_nodes++;
                        if (node == 0) {
                            const int owner = root_pos & 63, rr = root_pos >> 6;
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                if (r == rr && lane == owner) r_idx[r] = child;
                        } else if (lane == 0) {
                            memo_child[node & (kMemo - 1)] = child;
                        }
                        if (lane == 0) {
                            D.ch_index[base + e] = child;
                            sh.jobof[child - n0] = (int16_t)jid;
                        }
                        wave_sync();
                        ok = publish(-1, node, e, child, 1, nexp++, depth, -1);
                        if (!ok) break;
                    }
                    node = child;
                }
            }
        };
        // the first descents: the workers' (see la_go) - here only what the one-by-one order reports at this point, and the
        // one-by-one descents of a phase with a long path
        for (int f = 0; ok && f < n_first; ++f) {
            if (pipe_load(&sh.err)) { ok = false; break; }
            const int res = la_res[f];
            if (res == kWalkDeep) { descend_serial(f); continue; }
            if (res == kWalkPoolFull) {                                   // tree.py:418-420
                if (lane == 0) atomicOr(&D.err[t], kErrPoolFull);
                ok = false;
                break;
            }
            if (prof) pc[6] += 1;
        }
        lap(2);
        if (active) {
            // the root's virtual losses back to the pool
            const size_t base = (size_t)t * D.N * A;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = lane + 64 * r;
                if (i < r_nc) D.ch_vl[base + r_edge[r]] = r_cnt[r] - r_vis[r];       // rank order -> child index
            }
            if (lane == 0) D.node[(size_t)t * D.N].vl += r_added;
        }
        if (active && !ok && lane == 0) {
            if (!(D.err[t] & (kErrPoolFull | kErrRngEmpty))) atomicOr(&D.err[t], kErrPipeline);
            pipe_store(&sh.err, 1);
        }
        if (ok) queued = n_desc;                           // every descent has queued one leaf
        if (lane == 0) {
            bulk_n = ok ? queued : 0;
            pipe_store(&sh.final_count, jid);
        }
        lap(5);
        if (prof && lane == 0) {
            for (int i = 0; i < 9; ++i) D.prof[i] += pc[i];
            D.prof[12] += 1; D.prof[14] += n_iter;
            D.prof[15] += (long long)__builtin_amdgcn_s_memtime() - t_begin;
        }
    } else {
        // ---- workers ---------------------------------------------------------------------
        Lds<S, false> &L = sh.board[wid - 1];
        BoardScalars rootb;
        int root_to_move;
        load_root<S>(L, rootb, root_to_move, D, t, lane);
        // (s_memtime accumulators of tree 0's first worker: 9 waiting for the selector's go, 10 its entries, 11 the scheduled copies,
        // 13 start of the kernel to its end)
        const bool wprof = prof && wid == 1 && lane == 0;
        long long wt1 = 0, wt2 = 0, wt3 = 0, wt4 = 0;
        if (active) {
            // this wave's share of the entries' walks (see `walk`): nothing else to do until the first job is queued
            if (pipe_wait_ge(&la_ready, 1)) {
                const int n_walk = la_n;
                for (int f = wid; f < n_walk; f += 1 + NW) walk(f, hsw[wid - 1]);
            } else if (lane == 0) {
                atomicOr(&D.err[t], kErrPipeline);
                pipe_store(&sh.err, 1);
            }
            wave_sync();
            if (lane == 0) __hip_atomic_fetch_add(&la_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wprof) wt1 = (long long)__builtin_amdgcn_s_memtime();
            // this wave's share of the first descents (see la_go): expansions reserve their draws in entry order (xseq), an
            // entry waits only for entries before it - each on a worker that takes its entries in ascending order
            if (!pipe_wait_ge(&la_go, 1)) {
                if (lane == 0) { atomicOr(&D.err[t], kErrPipeline); pipe_store(&sh.err, 1); }
            } else if (la_fast) {
                if (wprof) wt2 = (long long)__builtin_amdgcn_s_memtime();
                const int n_ent = la_n;
                for (int f = wid - 1; f < n_ent; f += NW) {
                    const int res = la_res[f];
                    if (res != kWalkLeaf && res != kWalkExpand) continue;      // (pool full: reported by the selector)
                    if (pipe_load(&sh.err)) break;
                    int depth = la_depth[f], parent = la_node[f], edge = la_edge[f], child = la_chd[f];
                    reset_work<S>(L, lane);
                    BoardScalars b = rootb;
                    int c = root_to_move;
                    for (int i = 0; i < depth; ++i) {
                        put_stone<S>(L, b, la_moves[f][i], c, D.zob, lane);
                        c = 3 - c;
                    }
                    if (res == kWalkExpand) {
                        if (!expand_node_pipe<S>(L, b, c, D, t, child, parent, edge, la_xseq[f], sh, lane)) break;
                        // node.py:349-361 at the node just made (its arrays were written by this wave's lanes)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const int node = child;
                        const size_t nb = ((size_t)t * D.N + node) * A;
                        const int e2 = select_node_halving<S>(hsw[wid - 1], D, t, node, lane);
                        const int mv2 = D.action[nb + e2];
                        const int visits2 = D.ch_visits[nb + e2];
                        child = D.ch_index[nb + e2];
                        if (visits2 >= 1) {                                    // (a node made in this launch has no visited child)
                            if (lane == 0) { atomicOr(&D.err[t], kErrPipeline); pipe_store(&sh.err, 1); }
                            break;
                        }
                        if (lane == 0) { la_moves[f][depth] = (int16_t)mv2; rm_path[f][depth] = (node << 10) | e2; }
                        put_stone<S>(L, b, mv2, c, D.zob, lane);
                        c = 3 - c;
                        parent = node; edge = e2; depth += 1;
                        wave_sync();
                    }
                    // tree.py:412-416: the leaf's queue entry (node to evaluate - still NOT_EXPANDED: node[-1] -, parent, edge,
                    // path), the virtual losses of its path below the root (node.py:76-83), its planes
                    const int q = rm_slot[f];
                    const size_t qs = (size_t)t * D.K + q;
                    if (lane == 0) {
                        D.q_node[qs] = child;
                        D.q_pnode[qs] = parent;
                        D.q_pedge[qs] = edge;
                        D.q_depth[qs] = (depth <= kPathCap && D.N <= (1 << 21)) ? depth : 0;
                    }
                    const int my_entry = lane < depth ? rm_path[f][lane] : 0;
                    if (lane < depth && lane < kPathCap) D.q_path[qs * kPathCap + lane] = my_entry;
                    write_planes<S>(L, b, c, planes + (leaf_base + q) * 6 * G::P, lane);
                    // the later descents through this root child end on the same leaf (see `sched`): how many, for the virtual
                    // losses (node.py:76-83 below the root: one per descent on every node and edge of the path - added once); the
                    // position for whoever writes their queue entries and planes (below)
                    const int n_all = la_ndesc;
                    int n_through = 1;
                    for (int q0 = 0; q0 < n_all; q0 += 64)
                        n_through += __popcll(__ballot(q0 + lane < n_all && sched[q0 + lane] == f));
                    if (lane >= 1 && lane < depth) {
                        const size_t ns = (size_t)t * D.N + (my_entry >> 10);
                        atomicAdd(&D.node[ns].vl, n_through);
                        atomicAdd(&D.ch_vl[ns * A + (my_entry & 1023)], n_through);
                    }
                    {
                        const bool pass_plane = b.moves > 1 && b.prev == 0;
                        for (int pt = lane; pt < G::P; pt += 64) {
                            const int p = (pt / S + 1) * G::W + (pt % S) + 1;
                            int col = L.color[p];
                            if (c == kWhite && col != 0) col = 3 - col;
                            ent_code[f][pt] = (uint8_t)(col | ((!pass_plane && p == b.prev) ? 4 : 0));
                        }
                        if (lane == 0) {
                            ent_meta[f] = (pass_plane ? 1 : 0) | (c == kWhite ? 2 : 0);
                            rm_parent[f] = parent; rm_edge[f] = edge; rm_child[f] = child; rm_job[f] = -1; rm_depth[f] = depth;
                        }
                    }
                    wave_sync();
                    if (lane == 0) pipe_store(&ent_done[f], 1);
                }
                // the repeats, shared out over the workers: queue entry, path and planes of leaf slot q = those of its entry
                const int n_rep = la_ndesc;
                for (int q = wid - 1; q < n_rep; q += NW) {
                    const int f = sched[q];
                    if (f < 0) continue;
                    if (la_res[f] != kWalkLeaf && la_res[f] != kWalkExpand) continue;
                    bool there = false;
                    for (int spin = 0; spin < kPipeSpinLimit; ++spin) {
                        if (pipe_load(&ent_done[f])) { there = true; break; }
                        if (pipe_load(&sh.err)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!there) {
                        if (lane == 0) { atomicOr(&D.err[t], kErrPipeline); pipe_store(&sh.err, 1); }
                        break;
                    }
                    const int depth = rm_depth[f];
                    const size_t qs = (size_t)t * D.K + q;
                    if (lane == 0) {
                        D.q_node[qs] = rm_child[f];
                        D.q_pnode[qs] = rm_parent[f];
                        D.q_pedge[qs] = rm_edge[f];
                        D.q_depth[qs] = (depth <= kPathCap && D.N <= (1 << 21)) ? depth : 0;
                    }
                    if (lane < depth && lane < kPathCap) D.q_path[qs * kPathCap + lane] = rm_path[f][lane];
                    const int meta_f = ent_meta[f];
                    const float pass_f = (meta_f & 1) ? 1.f : 0.f, side_f = (meta_f & 2) ? -1.f : 1.f;
                    float *dst = planes + (leaf_base + q) * 6 * G::P;
                    for (int pt = lane; pt < G::P; pt += 64) {
                        const int code = ent_code[f][pt], col = code & 3;
                        dst[pt] = col == 0 ? 1.f : 0.f;
                        dst[G::P + pt] = col == 1 ? 1.f : 0.f;
                        dst[2 * G::P + pt] = col == 2 ? 1.f : 0.f;
                        dst[3 * G::P + pt] = (code & 4) ? 1.f : 0.f;
                        dst[4 * G::P + pt] = pass_f;
                        dst[5 * G::P + pt] = side_f;
                    }
                }
                if (wprof) wt3 = (long long)__builtin_amdgcn_s_memtime();
            }
        }
        for (int k = wid - 1; active; k += NW) {
            const int slot = k % kPipeSlots;
            bool have = false, stalled = true;
            for (int spin = 0; spin < kPipeSpinLimit; ++spin) {
                if (pipe_load(&sh.job_seq[slot]) == k + 1) { have = true; stalled = false; break; }
                const int fc = pipe_load(&sh.final_count);
                if (fc >= 0 && k >= fc) { stalled = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (stalled && lane == 0) {
                atomicOr(&D.err[t], kErrPipeline);
                pipe_store(&sh.err, 1);
            }
            if (!have) break;
            const PipeJob j = sh.job[slot];
            if (j.k >= 0) {
                // the leaf's queue entry and the virtual losses of its path below the root (node.py:76-83) - what the
                // selector would otherwise spend ~2 k cycles per descent on (64-bit addressing of seven arrays)
                const size_t qs = (size_t)t * D.K + j.k;
                if (lane == 0) {
                    D.q_node[qs] = j.child;
                    D.q_pnode[qs] = j.parent;
                    D.q_pedge[qs] = j.edge;
                    D.q_depth[qs] = (j.depth <= kPathCap && D.N <= (1 << 21)) ? j.depth : 0;
                }
                for (int i = lane; i < j.depth; i += 64) {
                    const int entry = sh.paths[slot][i];
                    if (i < kPathCap) D.q_path[qs * kPathCap + i] = entry;
                    if (i >= 1) {
                        const size_t ns = (size_t)t * D.N + (entry >> 10);
                        atomicAdd(&D.node[ns].vl, 1);
                        atomicAdd(&D.ch_vl[ns * A + (entry & 1023)], 1);
                    }
                }
            }
            if (j.expand == 2) {
                // COPY: the planes of leaf slot j.src (written by job j.xseq) are this leaf's planes
                if (!pipe_wait_done(sh, j.xseq)) {
                    if (lane == 0) { atomicOr(&D.err[t], kErrPipeline); pipe_store(&sh.err, 1); }
                    break;
                }
                static_assert((6 * G::P) % 2 == 0, "8-byte copies");
                const float2 *src = reinterpret_cast<const float2 *>(planes + (leaf_base + j.src) * 6 * G::P);
                float2 *dst = reinterpret_cast<float2 *>(planes + (leaf_base + j.k) * 6 * G::P);
                for (int i = lane; i < 3 * G::P; i += 64) dst[i] = src[i];
            } else {
                reset_work<S>(L, lane);
                BoardScalars b = rootb;
                int c = root_to_move;
                for (int i = 0; i < j.depth; ++i) {
                    put_stone<S>(L, b, sh.moves[slot][i], c, D.zob, lane);
                    c = 3 - c;
                }
                if (j.expand) expand_node_pipe<S>(L, b, c, D, t, j.child, j.parent, j.edge, j.xseq, sh, lane);
                if (j.k >= 0) write_planes<S>(L, b, c, planes + (leaf_base + j.k) * 6 * G::P, lane);
            }
            wave_sync();
            if (lane == 0) {
                pipe_set_done(sh, k);
                pipe_store(&sh.slot_done[slot], k / kPipeSlots + 1);
            }
        }
        // the scheduled leaves (see `sched`): leaf slot q of this tree repeats the leaf of root-child entry f
        if (wprof) wt4 = (long long)__builtin_amdgcn_s_memtime();
        const int nq = (active && !la_fast && pipe_load(&sh.final_count) >= 0 && !pipe_load(&sh.err)) ? bulk_n : 0;
        for (int q = wid - 1; q < nq; q += NW) {
            const int f = sched[q];
            if (f < 0) continue;
            const int depth = rm_depth[f];
            const size_t qs = (size_t)t * D.K + q;
            if (lane == 0) {
                D.q_node[qs] = rm_child[f];
                D.q_pnode[qs] = rm_parent[f];
                D.q_pedge[qs] = rm_edge[f];
                D.q_depth[qs] = depth <= kPathCap ? depth : 0;                     // (entries exist only if N <= 2^21)
            }
            if (lane < depth) {
                const int entry = rm_path[f][lane];
                if (lane < kPathCap) D.q_path[qs * kPathCap + lane] = entry;
                if (lane >= 1) {                                                   // node.py:76-83 below the root
                    const size_t ns = (size_t)t * D.N + (entry >> 10);
                    atomicAdd(&D.node[ns].vl, 1);
                    atomicAdd(&D.ch_vl[ns * A + (entry & 1023)], 1);
                }
            }
            if (!pipe_wait_done(sh, rm_job[f])) {                                  // the first leaf's planes are there
                if (lane == 0) { atomicOr(&D.err[t], kErrPipeline); pipe_store(&sh.err, 1); }
                break;
            }
            const float2 *src = reinterpret_cast<const float2 *>(planes + (leaf_base + rm_slot[f]) * 6 * G::P);
            float2 *dst = reinterpret_cast<float2 *>(planes + (leaf_base + q) * 6 * G::P);
            for (int i = lane; i < 3 * G::P; i += 64) dst[i] = src[i];
        }
        if (wprof) {
            const long long wt5 = (long long)__builtin_amdgcn_s_memtime();
            D.prof[9] += wt2 - wt1; D.prof[10] += wt3 - wt2; D.prof[11] += wt5 - wt4; D.prof[13] += wt5 - t_begin;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        D.meta[t].num_nodes = num_nodes;
        D.n_leaves[t] = queued;
        set_cursor(D, t, sh.cursor_val);
    }
}

// tree.py:359-422: one sequential-halving phase per launch: for threshold 1..max_count,
// num_considered descents each; every descent ends in a queued leaf.
template <int S>
__global__ __launch_bounds__(64) void select_gumbel_kernel(SearchDev D, const int32_t *num_considered,
                                                           const int32_t *max_count, int stride,
                                                           const int32_t *leaf_off, float *planes) {
    using G = Geo<S>;
    constexpr int A = G::A;
    __shared__ Lds<S> L;
    const int t = blockIdx.x, lane = threadIdx.x;
    BoardScalars rootb;
    int root_to_move;
    load_root<S>(L, rootb, root_to_move, D, t, lane);
    int num_nodes = D.meta[t].num_nodes;
    int queued = 0;
    const int width = num_considered[t], levels = max_count[t];
    // packed layout (leaf_off != null): this tree's leaves follow those of the trees before it
    const size_t leaf_base = leaf_off ? (size_t)leaf_off[t] : (size_t)t * stride;
    bool ok = D.err[t] == 0 && num_nodes > 0 && width * levels <= stride;
    for (int th = 1; ok && th <= levels; ++th) {
        for (int j = 0; ok && j < width; ++j) {
            reset_work<S>(L, lane);
            BoardScalars b = rootb;
            int c = root_to_move;
            int node = 0;
            for (;;) {
                const size_t ns = (size_t)t * D.N + node;
                const size_t base = ns * A;
                const int e = node == 0 ? select_root_halving<S>(D, t, node, th, lane)
                                        : select_node_halving<S>(L, D, t, node, lane);
                const int mv = D.action[base + e];
                put_stone<S>(L, b, mv, c, D.zob, lane);
                c = 3 - c;
                const int visits = D.ch_visits[base + e];
                int child = D.ch_index[base + e];
                wave_sync();
                if (lane == 0) {
                    D.node[ns].vl += 1;
                    D.ch_vl[base + e] += 1;
                }
                if (visits < 1) {                                     // tree.py:412-416
                    write_planes<S>(L, b, c, planes + (leaf_base + queued) * 6 * G::P, lane);
                    if (lane == 0) {
                        D.q_node[(size_t)t * D.K + queued] = child;   // still NOT_EXPANDED: node[-1]
                        D.q_pnode[(size_t)t * D.K + queued] = node;
                        D.q_pedge[(size_t)t * D.K + queued] = e;
                        D.q_depth[(size_t)t * D.K + queued] = 0;
                    }
                    wave_sync();
                    break;
                }
                if (child == kNotExpanded) {                          // tree.py:418-420
                    child = expand_node<S>(L, b, c, D, t, num_nodes, node, e, lane);
                    if (child < 0) { ok = false; break; }
                    if (lane == 0) D.ch_index[base + e] = child;
                }
                node = child;
                wave_sync();
            }
            if (ok) ++queued;
        }
    }
    if (lane == 0) {
        D.meta[t].num_nodes = num_nodes;
        D.n_leaves[t] = queued;
    }
}


// Apply one move per tree to its ROOT position on the device (GoBoard.put_stone on the game
// board, go_board.py:131-185) so that self-play boards never leave the GPU.
template <int S>
__global__ __launch_bounds__(64) void play_kernel(SearchDev D, const int32_t *moves) {
    using G = Geo<S>;
    __shared__ Lds<S> L;
    const int t = blockIdx.x, lane = threadIdx.x;
    int mv = moves[t];
    if (mv == -2) {
        // the move of the most visited root child (node.py:167-175 get_best_move_index: np.argmax, first index on ties),
        // chosen here so that a driver that only advances its positions needs no read-back between two searches
        constexpr int R = (G::A + 63) / 64;
        const size_t rbase = (size_t)t * D.N * G::A;
        const int nc = D.meta[t].num_nodes > 0 ? D.node[(size_t)t * D.N].children : 0;
        double best = 0.0;
        int best_i = -1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane + 64 * r;
            if (i < nc) {
                const double v = (double)D.ch_visits[rbase + i];
                if (best_i < 0 || v > best) { best = v; best_i = i; }
            }
        }
        best_i = wave_argmax_first(best, best_i);
        mv = best_i >= 0 ? (int)D.action[rbase + best_i] : -1;
    }
    if (mv < 0) return;                                 // RESIGN / idle tree
    BoardScalars b;
    int to_move;
    load_root<S>(L, b, to_move, D, t, lane);
    reset_work<S>(L, lane);
    const int at = b.moves;
    put_stone<S>(L, b, mv, to_move, D.zob, lane);
    for (int p = lane; p < G::NC; p += 64) D.root_cells[(size_t)t * G::NC + p] = L.color[p];
    if (lane == 0) {
        if (at < G::HMAX) D.root_hist[(size_t)t * G::HMAX + at] = b.hash;
        RootMeta m = D.meta[t];
        m.hash = b.hash;
        m.moves = b.moves;
        m.ko_pos = b.ko_pos;
        m.ko_move = b.ko_move;
        m.prev = b.prev;
        m.prevprev = b.prevprev;
        m.to_move = 3 - to_move;
        m.hist_len = D.superko ? (b.moves < G::HMAX ? b.moves : G::HMAX) : 1;
        m.num_nodes = 0;
        D.meta[t] = m;
    }
}

}  // namespace

// ======================================================================================
// host side
// ======================================================================================
// One node of one tree packed into one record (tg_search_read_node: one launch + one copy instead of thirteen
// copies): [num_children, node_visits, node_virtual_loss, node_value_sum bits, raw_value bits, error flags, 0, 0],
// children_index[A], children_visits[A], children_virtual_loss[A], action[A] (int32), then value_sum[A], policy[A],
// value[A] (float64, 8-byte aligned).
__global__ __launch_bounds__(64) void gather_node_kernel(SearchDev D, int A, int tree, int node, unsigned char *out) {
    const int lane = threadIdx.x;
    const size_t ns = (size_t)tree * D.N + node, base = ns * A;
    int32_t *head = reinterpret_cast<int32_t *>(out);
    if (lane == 0) {
        head[0] = D.node[ns].children;
        head[1] = D.node[ns].visits;
        head[2] = D.node[ns].vl;
        head[3] = __float_as_int(D.node[ns].vsum);
        head[4] = __float_as_int(D.node[ns].raw);
        head[5] = D.err[tree];
        head[6] = D.meta[tree].num_nodes;          // (tg_search_node_record_num_nodes: no second read after a search)
        head[7] = 0;
    }
    int32_t *idx = head + 8, *vis = idx + A, *vl = vis + A, *act = vl + A;
    double *vsum = reinterpret_cast<double *>(out + 32 + (((size_t)4 * A * 4 + 7) & ~(size_t)7));
    double *pol = vsum + A, *val = pol + A;
    for (int i = lane; i < A; i += 64) {
        idx[i] = D.ch_index[base + i];
        vis[i] = D.ch_visits[base + i];
        vl[i] = D.ch_vl[base + i];
        act[i] = D.action[base + i];
        vsum[i] = D.ch_vsum[base + i];
        pol[i] = D.ch_policy[base + i];
        val[i] = D.ch_value[base + i];
    }
}

// Root statistics of every tree packed into one record per tree (one launch + ONE device-to-host copy instead of
// nine strided copies, each a host round trip): [num_children, node_visits, raw_value bits, error flags] then
// visits[A], virtual_loss[A], action[A] (int32), value_sum[A], policy[A] (float64).
__global__ __launch_bounds__(64) void gather_roots_kernel(SearchDev D, int A, unsigned char *out, size_t rec_bytes) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const size_t ns = (size_t)t * D.N, base = ns * A;
    unsigned char *rec = out + (size_t)t * rec_bytes;
    int32_t *head = reinterpret_cast<int32_t *>(rec);
    if (lane == 0) {
        head[0] = D.node[ns].children;
        head[1] = D.node[ns].visits;
        head[2] = __float_as_int(D.node[ns].raw);
        head[3] = D.err[t];
    }
    int32_t *vis = head + 4, *vl = vis + A, *act = vl + A;
    double *vsum = reinterpret_cast<double *>(rec + 16 + (size_t)3 * A * 4 + (((size_t)3 * A * 4) % 8 ? 4 : 0));
    double *pol = vsum + A;
    for (int i = lane; i < A; i += 64) {
        vis[i] = D.ch_visits[base + i];
        vl[i] = D.ch_vl[base + i];
        act[i] = D.action[base + i];
        vsum[i] = D.ch_vsum[base + i];
        pol[i] = D.ch_policy[base + i];
    }
}

// Self-play, the end of a move WITHOUT the host (tg_selfplay_play_move): the root records as above, then the move itself -
// final root choice (tree.py:344, node.py:324-346: argmax of logit + noise + sigma(q) with the visit threshold), resign rule
// (worker.py:59-62), two-pass / move-limit end (worker.py:44, :80) - in the arithmetic the host used to do it in
// (float64, IEEE operations, first maximum wins), so that play_kernel and the next root expansion can follow at once.
// state [T][4]: {bit 0: slot takes no part (parked / game just started), bit 1: never resign; passes in a row; moves
// played; 0}.  moves_out[t]: the point played (0 = pass) or -1 (nothing to play: resigned, game over, idle slot).
// tail [T]: {best child, kind (0 move / 1 resign / 2 second pass / 3 move limit / 4 idle), point, 0, draw cursor (int64), 0}.
struct RootTail {
    int32_t best, kind, pos, pad;
    int64_t cursor, pad2;
};
__global__ __launch_bounds__(64) void finish_roots_kernel(SearchDev D, int A, unsigned char *out, size_t rec_bytes,
                                                          const int32_t *state, int max_moves, int32_t *moves_out, RootTail *tail) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const size_t ns = (size_t)t * D.N, base = ns * A;
    unsigned char *rec = out + (size_t)t * rec_bytes;
    int32_t *head = reinterpret_cast<int32_t *>(rec);
    const int n = D.node[ns].children;
    if (lane == 0) {
        head[0] = n;
        head[1] = D.node[ns].visits;
        head[2] = __float_as_int(D.node[ns].raw);
        head[3] = D.err[t];
    }
    int32_t *vis = head + 4, *vl = vis + A, *act = vl + A;
    double *vsum = reinterpret_cast<double *>(rec + 16 + (size_t)3 * A * 4 + (((size_t)3 * A * 4) % 8 ? 4 : 0));
    double *pol = vsum + A;
    int mc = 0;
    for (int i = lane; i < A; i += 64) {
        const int v = D.ch_visits[base + i];
        vis[i] = v;
        vl[i] = D.ch_vl[base + i];
        act[i] = D.action[base + i];
        vsum[i] = D.ch_vsum[base + i];
        pol[i] = D.ch_policy[base + i];
        if (i < n) mc = v > mc ? v : mc;
    }
    const int flags = state[4 * t], passes = state[4 * t + 1], played = state[4 * t + 2];
    int best = -1, kind = 4, pos = -1, mv = -1;
    if (!(flags & 1) && n > 0) {
        mc = wave_max_i32(mc);
        const double sigma_sel = (double)(50 + mc) * 1.0;
        double bv = 0.0;
        int bi = -1;
        for (int i = lane; i < n; i += 64) {
            const int v = D.ch_visits[base + i];
            const double qi = v > 0 ? D.ch_vsum[base + i] / (double)v : 0.0;
            const double ev = (v + D.ch_vl[base + i] >= 100) ? -10000.0
                                                              : (D.ch_policy[base + i] + D.noise[(size_t)t * A + i]) + sigma_sel * qi;
            if (bi < 0 || ev > bv) { bv = ev; bi = i; }
        }
        best = wave_argmax_first(bv, bi);
        if (best < 0) best = 0;                            // every evaluation NaN (a diverged network): the host's rule "i == 0 || ev > best" keeps child 0
        const int bvis = D.ch_visits[base + best];
        const double value = bvis == 0 ? 0.5 : D.ch_vsum[base + best] / (double)bvis;
        pos = D.action[base + best];
        if (!(flags & 2) && value < 0.05) {
            kind = 1;
        } else {
            const int p2 = pos == 0 ? passes + 1 : 0;
            if (p2 == 2) kind = 2;
            else if (played + 1 >= max_moves) kind = 3;
            else { kind = 0; mv = pos; }
        }
    }
    if (lane == 0) {
        moves_out[t] = mv;
        RootTail r{};
        r.best = best; r.kind = kind; r.pos = pos; r.cursor = D.rng_cursor[t];
        tail[t] = r;
    }
}

// the draw cursors into pinned host memory (self-play: read by the host a root evaluation later - no copy, no copy stream)
__global__ void publish_cursors_kernel(const int64_t *cursor, int64_t *out, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) out[t] = cursor[t];
}

struct tg_search {
    tg_search_config cfg{};
    SearchDev dev{};
    int num_cus = 256;
    bool prefill_enabled = true;           // (unused since the streams live on the device; kept for the callers that set it)
    int split_per_cu = -1;                 // resident select_puct_split_kernel workgroups per CU (queried at the first launch)
    std::vector<void *> allocs;
    int S = 0, W = 0, NC = 0, P = 0, A = 0, HMAX = 0;
    hipStream_t last_stream = nullptr;
    bool stream_known = false;             // a launch has named its stream (which may be the null stream: last_stream == nullptr)
    // root positions are staged on the host and uploaded in bulk by tg_search_root_planes
    std::vector<uint8_t> st_cells;
    std::vector<uint64_t> st_hist;
    std::vector<RootMeta> st_meta;
    std::vector<uint8_t> st_dirty_tree;
    bool st_dirty = false;
    int32_t *phase_dev = nullptr;          // [num_considered | max_count | packed leaf offsets], T each
    unsigned char *roots_dev = nullptr, *roots_host = nullptr;   // gather_roots_kernel records (device / pinned host)
    unsigned char *node_dev = nullptr, *node_host = nullptr;     // gather_node_kernel record
    // pinned staging ring for the phase description: the host never waits for the copy of the current call, only
    // (practically never) for the one kPhaseRing calls ago
    static constexpr int kPhaseRing = 8;
    int32_t *phase_pin = nullptr;
    hipEvent_t phase_ev[kPhaseRing] = {};
    bool phase_ev_used[kPhaseRing] = {};
    unsigned phase_seq = 0;
    bool packed_leaves = false;
    int32_t *moves_dev = nullptr;
    // select_puct_split_kernel: job entries and "node initialised" tags that cross between a tree's two workgroups
    int *xw_job = nullptr, *xw_done = nullptr, *xw_n = nullptr;
    unsigned long long *xw_off = nullptr;
    int xw_cap = 0;
    unsigned xw_seq = 0;
    // double-buffered random windows, uploaded on a private copy stream so that the host can
    // prepare mini-batch j+1 while the forward pass of mini-batch j runs
    hipStream_t copy_stream = nullptr;
    hipStream_t own_stream = nullptr;              // tg_search_own_stream: a launch stream of the handle's own (self-play lanes)
    hipEvent_t ev_rng[2] = {nullptr, nullptr};
    hipEvent_t ev_sel = nullptr;
    double *rng_buf[2] = {nullptr, nullptr};
    int64_t rng_buf_cap[2] = {0, 0};
    int rng_active = 0, rng_pending = -1;
    int64_t rng_pending_cap = 0;
    bool sel_recorded = false;
    // Library-owned legacy streams (tg_search_seed_stream), device-resident since round 6 (csrc/legacy_rng_device.h): per tree
    // the MT19937 state at the stream's logical position (`mt_base`) and behind the last generated piece of a window
    // (`mt_cont`), [T][625] words each; windows and Gumbel noise are GENERATED by rng_fill_kernel on `copy_stream`.  The host
    // keeps, per tree, only what it was seeded with until that is uploaded and the draws consumed since the device state was
    // last brought up to date (`lag`, from the cursor read-backs) - handed to the next generation launch through a ring of
    // host-mapped arrays.
    struct DevStream {
        bool seeded = false, dirty = false;
        uint32_t key[624];
        int pos = 624;
        int64_t lag = 0;
    };
    std::vector<DevStream> streams;
    uint32_t *mt_base = nullptr, *mt_cont = nullptr;
    uint32_t *rng_words = nullptr;                // scratch: the tempered words of the piece being generated (rng_words_kernel)
    size_t rng_words_cap = 0;
    int *rng_pos0 = nullptr;
    uint32_t *rng_snap = nullptr;                 // [T][rng_snap_cap][624]: state snapshots behind mt_base (T <= 64)
    int rng_snap_cap = 0;
    int *rng_snap_n = nullptr;
    long long *rng_cont_blk = nullptr;
    uint32_t *seed_pin = nullptr;                 // pinned [T][625]: seeds on their way up, one state on its way down
    hipEvent_t seed_ev = nullptr;
    bool seed_ev_used = false;
    static constexpr int kLagRing = 8;
    long long *lag_pin = nullptr, *lag_pin_dev = nullptr;          // host-mapped [kLagRing][T]
    unsigned char *skip_pin = nullptr, *skip_pin_dev = nullptr;    // host-mapped [kLagRing][T]
    hipEvent_t lag_ev[kLagRing] = {};
    bool lag_ev_used[kLagRing] = {};
    unsigned lag_seq = 0;
    size_t eager_need = 0;                        // few trees: the last whole-window request (advance_streams regenerates ahead)
    bool auto_rest = false;                       // feed_streams_impl over-generated: the rest of the window goes out at install_rng
    int64_t *consumed_pin = nullptr;              // host-mapped [T]: the kernels' mirror of the cursors (SearchDev::cursor_pub)
    double *noise_back = nullptr;                 // pinned [T][A]: the device-drawn root noise on its way to noise_host
    hipEvent_t noise_back_ev = nullptr, noise_order_ev = nullptr;
    bool noise_back_pending = false;
    int64_t win_cap = 0, win_left = 0;            // active / pending window: size, unread tail
    // split upload (feed_streams_impl / feed_streams_rest): buffer, columns already up, columns still to come; the next
    // selection launch must wait for the second part's event
    int rng_rest_idx = 0;
    size_t rng_rest_first = 0, rng_rest_cols = 0;
    bool rng_rest_wait = false;
    std::vector<int64_t> win_used;                // device cursor per tree at the last advance
    std::vector<double> noise_host;               // last root Gumbel noise [T][A] (tg_search_set_noise)
    // pinned staging rings for the small per-move uploads (root noise, chosen moves): queued on the launch stream behind
    // the kernels that still read the old contents - the host does not wait for the stream (it used to: a pipeline drain
    // per upload, twice per self-play move)
    static constexpr int kPinRing = 4;
    double *noise_pin = nullptr;
    int32_t *moves_pin = nullptr;
    hipEvent_t noise_ev[kPinRing] = {}, moves_ev[kPinRing] = {};
    bool noise_ev_used[kPinRing] = {}, moves_ev_used[kPinRing] = {};
    unsigned noise_seq = 0, moves_seq = 0;
    // finish_roots_kernel (self-play: the move decided on the device): per-tree state uploads, the event behind its records
    int64_t *cur_pin = nullptr, *cur_pin_dev = nullptr;     // publish_cursors_kernel's target (pinned, mapped) + its event
    hipEvent_t cur_ev = nullptr;
    uint8_t *roots_pin = nullptr;          // pinned mirror of st_cells / st_hist / st_meta (flush_roots)
    hipEvent_t roots_pin_ev = nullptr;
    bool roots_pin_used = false;
    int32_t *state_dev = nullptr, *state_pin = nullptr;
    hipEvent_t state_ev[kPinRing] = {}, fin_ev = nullptr;
    bool state_ev_used[kPinRing] = {};
    unsigned state_seq = 0;
};

namespace {

template <typename T>
int dev_alloc(tg_search *s, T **out, size_t count, bool zero = true) {
    void *p = nullptr;
    TG_HIP(hipMalloc(&p, count * sizeof(T)));
    s->allocs.push_back(p);
    if (zero) {
        // hipMemset on device memory is asynchronous (legacy null stream); callers go on to use the
        // buffer from NON-BLOCKING streams (torch side streams), which the null stream does not order
        TG_HIP(hipMemset(p, 0, count * sizeof(T)));
        TG_HIP(hipStreamSynchronize(nullptr));
    }
    *out = static_cast<T *>(p);
    return TG_OK;
}

// Host threads for the per-tree host work of a call (random windows, Gumbel noise, move bookkeeping): a small
// persistent pool (threads created once per process; a job is a [0, n) index range handed out through an atomic
// counter, the caller works along).  Spawning std::threads per call cost 20-30 us each - more than the work of a
// 16-board shard (0.28 ms of serial bookkeeping per lock-step move, TG_SP_TIMING).  One job at a time: a second caller
// (another lock-step group's host thread) finding the pool busy runs its trees itself.
class TreePool {
public:
    static TreePool &get() {
        static TreePool pool;
        return pool;
    }
    int size() const { return (int)workers_.size(); }
    template <typename F>
    bool run(int n, int max_threads, F &fn) {
        std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
        if (!job.owns_lock()) return false;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = [&fn](int t) { fn(t); };
            n_ = n;
            next_.store(0);
            active_ = std::min(max_threads - 1, (int)workers_.size());
            pending_ = active_;
            ++epoch_;
        }
        cv_.notify_all();
        for (int t; (t = next_.fetch_add(1)) < n;) fn(t);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
        return true;
    }

private:
    TreePool() {
        unsigned hw = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            const int n = CPU_COUNT(&set);
            if (n > 0) hw = (unsigned)n;
        }
        int n = (int)std::min<unsigned>(hw ? hw : 1u, 16u);
        if (const char *env = getenv("TG_HOST_THREADS")) {
            const int cap = atoi(env);
            if (cap >= 1 && cap < n) n = cap;
        }
        for (int w = 0; w + 1 < n; ++w) workers_.emplace_back([this, w] { loop(w); });
    }
    ~TreePool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &th : workers_) th.join();
    }
    void loop(int w) {
        long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
            if (stop_) return;
            seen = epoch_;
            if (w >= active_) continue;
            lk.unlock();
            for (int t; (t = next_.fetch_add(1)) < n_;) fn_(t);
            lk.lock();
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_cv_;
    std::function<void(int)> fn_;
    std::atomic<int> next_{0};
    int n_ = 0, active_ = 0, pending_ = 0;
    long epoch_ = 0;
    bool stop_ = false;
};

template <typename F>
void parallel_trees(int n, F &&fn) {
    TreePool &pool = TreePool::get();
    // threads worth waking: one per eight trees, from 32 trees on (waking sleeping threads costs ~0.1 ms: at 16 boards
    // the pool made the per-move bookkeeping slower, 0.44 vs 0.28 ms; at 64 boards the shard gained 8 %)
    const int want = n >= 32 ? std::min(pool.size() + 1, n / 8) : 1;
    if (want >= 2 && pool.run(n, want, fn)) return;
    for (int t = 0; t < n; ++t) fn(t);
}

}  // namespace

namespace {
// One pool array [T][n_old * per_node] to be re-allocated as [T][n_new * per_node], keeping every tree's rows.  The
// growth is transactional: every new array is allocated and filled FIRST; only when all of that has
// succeeded are the pointers swapped and the old arrays freed - a failing hipMalloc (old and new pools are resident
// together: up to 2 x 13 GB at the 4 M-node cap) leaves the handle exactly as it was.
struct GrowItem {
    void **field;          // address of the SearchDev pointer
    size_t elem;           // bytes per element
    size_t per_node;       // elements per node
    void *fresh = nullptr;
};

int grow_fill(const GrowItem &g, size_t trees, size_t n_old, size_t n_new) {
    const size_t row_old = n_old * g.per_node * g.elem, row_new = n_new * g.per_node * g.elem;
    TG_HIP(hipMemset(g.fresh, 0, trees * row_new));
    if (row_old < ((size_t)1 << 30)) {
        TG_HIP(hipMemcpy2D(g.fresh, row_new, *g.field, row_old, row_old, trees, hipMemcpyDeviceToDevice));
    } else {
        // rows beyond 1 GB (millions of nodes per tree): one linear copy per tree, no 2D pitch limits involved
        for (size_t t = 0; t < trees; ++t)
            TG_HIP(hipMemcpy(static_cast<char *>(g.fresh) + t * row_new, static_cast<const char *>(*g.field) + t * row_old,
                             row_old, hipMemcpyDeviceToDevice));
    }
    return TG_OK;
}
}  // namespace

namespace {
constexpr int kSplitNoRoom = 1;     // launch_split*: not launched - the device cannot hold all the tree's workgroups at once
template <int S, int NNODE, int NWRK, int NSHIP = 3, int NWG = 1>
int launch_split_cfg(tg_search *s, int max_leaves, float *planes, hipStream_t st) {
    constexpr size_t lds_a = sizeof(SplitSelShared<S, NNODE>), lds_b = sizeof(SplitWrkShared<S, NWRK>);
    constexpr size_t lds = lds_a > lds_b ? lds_a : lds_b;
    static_assert(lds <= 160 * 1024, "LDS");
    static_assert(NNODE + 2 + NSHIP + (NWG > 1 ? 1 : 0) + (S == 9 ? 1 : 0) <= 16 && NWRK <= 16, "wavefronts per workgroup");
    const int T = s->dev.T;
    if (!s->xw_job) {
        s->xw_cap = s->dev.K < kPipeMaxK ? s->dev.K : kPipeMaxK;
        int rc = dev_alloc(s, &s->xw_job, (size_t)T * s->xw_cap * kXwEntryWords<S>);
        if (rc) return rc;
        if ((rc = dev_alloc(s, &s->xw_done, (size_t)T * s->xw_cap))) return rc;
        if ((rc = dev_alloc(s, &s->xw_n, (size_t)T * s->xw_cap))) return rc;
        if ((rc = dev_alloc(s, &s->xw_off, (size_t)T * (s->xw_cap + 1)))) return rc;
    }
    if ((++s->xw_seq & 0xFFFFFu) == 0) {                 // the launch number in the tags wraps: start from clean buffers
        TG_HIP(hipMemsetAsync(s->xw_job, 0, (size_t)T * s->xw_cap * kXwEntryWords<S> * sizeof(int), st));
        TG_HIP(hipMemsetAsync(s->xw_done, 0, (size_t)T * s->xw_cap * sizeof(int), st));
        TG_HIP(hipMemsetAsync(s->xw_n, 0, (size_t)T * s->xw_cap * sizeof(int), st));
        TG_HIP(hipMemsetAsync(s->xw_off, 0, (size_t)T * (s->xw_cap + 1) * sizeof(unsigned long long), st));
        s->xw_seq = 1;
    }
    static std::atomic<uint64_t> configured{0};          // per device, thread-safe (a process-wide bool was neither)
    if (tg::first_on_device(configured, s->cfg.device))
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&select_puct_split_kernel<S, NNODE, NWRK, NSHIP, NWG>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // The tree's workgroups wait for each other through memory: all (1 + NWG) T of them must be resident at once, which an
    // ordinary launch does not promise.  They are one per CU (1024 threads, > 80 KB of LDS); when the device has fewer CUs
    // than that (partitioned devices) the caller falls back to the one-workgroup kernel (kSplitNoRoom).
    if (s->split_per_cu < 0) {                            // (once per handle: the query is not free)
        int per_cu = 0;
        TG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, select_puct_split_kernel<S, NNODE, NWRK, NSHIP, NWG>, 1024, lds));
        s->split_per_cu = per_cu;
    }
    if ((long long)(1 + NWG) * T > (long long)s->split_per_cu * s->num_cus) return kSplitNoRoom;
    hipLaunchKernelGGL((select_puct_split_kernel<S, NNODE, NWRK, NSHIP, NWG>), dim3((1 + NWG) * T), dim3(1024), lds, st, s->dev,
                       max_leaves, planes, s->xw_job, s->xw_done, s->xw_n, s->xw_off, (int)((s->xw_seq & 0xFFFFFu) << 11), s->xw_cap,
                       tg::knob("TG_SPLIT_TEST_MUTE") ? 1 : 0);
    return TG_OK;
}

template <int S>
int launch_split(tg_search *s, int max_leaves, float *planes, hipStream_t st) {
    // node owners * 100 + workers (TG_SPLIT_CFG: tuning knob)
    static const int cfg = tg::knob("TG_SPLIT_CFG") ? atoi(tg::knob("TG_SPLIT_CFG")) : 0;
    if constexpr (S == 9) {
        // (16 waves: chooser + clerk + owners + allocator + shippers + the draw cursor when there are several worker workgroups)
        if (cfg == 616) return launch_split_cfg<S, 6, 16>(s, max_leaves, planes, st);
        if (cfg == 816) return launch_split_cfg<S, 8, 16>(s, max_leaves, planes, st);
        if (cfg == 1016) return launch_split_cfg<S, 10, 16, 2>(s, max_leaves, planes, st);          // two shippers
        if (cfg == 912) return launch_split_cfg<S, 9, 12>(s, max_leaves, planes, st);
        if (cfg == 11016) return launch_split_cfg<S, 10, 16, 3, 1>(s, max_leaves, planes, st);    // one workgroup of workers
        if (cfg == 30916) return launch_split_cfg<S, 9, 16, 3, 3>(s, max_leaves, planes, st);
        return launch_split_cfg<S, 9, 16, 3, 2>(s, max_leaves, planes, st);
    } else {
        if (cfg == 607) return launch_split_cfg<S, 6, 7>(s, max_leaves, planes, st);
        if (cfg == 1207) return launch_split_cfg<S, 12, 7, 2>(s, max_leaves, planes, st);
        if (cfg == 11007) return launch_split_cfg<S, 10, 7, 3, 1>(s, max_leaves, planes, st);
        if (cfg == 31007) return launch_split_cfg<S, 10, 7, 3, 3>(s, max_leaves, planes, st);
        return launch_split_cfg<S, 10, 7, 3, 2>(s, max_leaves, planes, st);
    }
}
}  // namespace

extern "C" {

int tg_search_create(const tg_search_config *cfg, tg_search **out) {
    if (!cfg || !out) return tg::fail(TG_ERR_ARG, "tg_search_create: null argument");
    // 9 and 19 have the tuned kernels; 13 (the third size TamaGo is played at: board/constant.py:4 BOARD_SIZE is a free
    // constant, main.py --size) runs on the board-size-generic ones (one wavefront per tree for selection, the 19x19 backup) -
    // same trees, no pipelining.  Other sizes: instantiate them below (kGenericSize) and rebuild.
    if (cfg->board_size != 9 && cfg->board_size != 19 && cfg->board_size != 13)
        return tg::fail(TG_ERR_ARG, "tg_search_create: board size %d not built (9, 13 and 19 are)", cfg->board_size);
    if (cfg->num_trees < 1 || cfg->tree_size < 2 || cfg->batch_size < 1)
        return tg::fail(TG_ERR_ARG, "tg_search_create: num_trees/tree_size/batch_size out of range");
    TG_HIP(hipSetDevice(cfg->device));
    tg_search *s = new tg_search;
    s->cfg = *cfg;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) s->num_cus = cus;
    }
    s->S = cfg->board_size;
    s->W = s->S + 2;
    s->NC = s->W * s->W;
    s->P = s->S * s->S;
    s->A = s->P + 1;
    s->HMAX = 3 * s->P;
    SearchDev &D = s->dev;
    D.T = cfg->num_trees;
    D.N = cfg->tree_size;
    D.K = cfg->batch_size;
    D.cgos = cfg->cgos_mode;
    D.superko = cfg->check_superko;
    const size_t T = D.T, N = D.N, A = s->A, K = D.K;
    int rc = TG_OK;
#define ALLOC(field, count) if ((rc = dev_alloc(s, &D.field, (count)))) { tg_search_destroy(s); return rc; }
    ALLOC(ch_index, T * N * A) ALLOC(ch_visits, T * N * A) ALLOC(ch_vl, T * N * A)
    ALLOC(ch_vsum, T * N * A) ALLOC(ch_policy, T * N * A) ALLOC(ch_value, T * N * A)
    ALLOC(action, T * N * A)
    ALLOC(node, T * N)
    ALLOC(noise, T * A)
    ALLOC(root_cells, T * s->NC) ALLOC(root_hist, T * s->HMAX) ALLOC(meta, T)
    ALLOC(q_node, T * K) ALLOC(q_pnode, T * K) ALLOC(q_pedge, T * K) ALLOC(n_leaves, T)
    ALLOC(q_depth, T * K) ALLOC(q_path, T * K * kPathCap)
    ALLOC(rng_cursor, T) ALLOC(err, T)
#undef ALLOC
    // random windows: one mini-batch worth of expansions each (grown on demand)
    for (int b = 0; b < 2; ++b) {
        s->rng_buf_cap[b] = (int64_t)K * A;
        hipError_t e2 = hipMalloc(reinterpret_cast<void **>(&s->rng_buf[b]), T * (size_t)s->rng_buf_cap[b] * sizeof(double));
        if (e2 != hipSuccess) { tg_search_destroy(s); return tg::fail(TG_ERR_HIP, "rng window: %s", hipGetErrorString(e2)); }
    }
    D.rng = s->rng_buf[0];
    D.rng_cap = 0;              // nothing installed yet
    {
        // the cursors' host-mapped mirror (tg_search_rng_consumed)
        void *dp = nullptr;
        if (hipHostMalloc(reinterpret_cast<void **>(&s->consumed_pin), T * sizeof(int64_t), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&dp, s->consumed_pin, 0) != hipSuccess) {
            tg_search_destroy(s);
            return tg::fail(TG_ERR_HIP, "tg_search_create: cursor mirror");
        }
        std::memset(s->consumed_pin, 0, T * sizeof(int64_t));
        D.cursor_pub = static_cast<int64_t *>(dp);
    }
    if (hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_rng[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_rng[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_sel, hipEventDisableTiming) != hipSuccess) {
        tg_search_destroy(s);
        return tg::fail(TG_ERR_HIP, "tg_search_create: stream/event creation failed");
    }
    uint64_t *zob = nullptr;
    if ((rc = dev_alloc(s, &zob, (size_t)4 * s->NC))) { tg_search_destroy(s); return rc; }
    D.zob = zob;
    s->st_cells.assign(T * s->NC, 0);
    s->st_hist.assign(T * s->HMAX, 0);
    s->st_meta.assign(T, RootMeta{});
    s->st_dirty_tree.assign(T, 0);
    *out = s;
    return TG_OK;
}

int tg_search_destroy(tg_search *s) {
    if (!s) return TG_OK;
    (void)hipSetDevice(s->cfg.device);
    if (s->stream_known) (void)hipStreamSynchronize(s->last_stream);      // (nullptr: the null stream)
    if (s->noise_pin) {
        (void)hipHostFree(s->noise_pin);
        for (int i = 0; i < tg_search::kPinRing; ++i) (void)hipEventDestroy(s->noise_ev[i]);
    }
    if (s->roots_pin) { (void)hipHostFree(s->roots_pin); (void)hipEventDestroy(s->roots_pin_ev); }
    if (s->cur_pin) { (void)hipHostFree(s->cur_pin); (void)hipEventDestroy(s->cur_ev); }
    if (s->state_pin) {
        (void)hipHostFree(s->state_pin);
        for (int i = 0; i < tg_search::kPinRing; ++i) (void)hipEventDestroy(s->state_ev[i]);
        (void)hipEventDestroy(s->fin_ev);
    }
    if (s->moves_pin) {
        (void)hipHostFree(s->moves_pin);
        for (int i = 0; i < tg_search::kPinRing; ++i) (void)hipEventDestroy(s->moves_ev[i]);
    }
    for (void *p : s->allocs) (void)hipFree(p);
    if (s->roots_dev) { (void)hipFree(s->roots_dev); (void)hipHostFree(s->roots_host); }
    if (s->node_dev) { (void)hipFree(s->node_dev); (void)hipHostFree(s->node_host); }
    if (s->phase_pin) {
        (void)hipHostFree(s->phase_pin);
        for (int i = 0; i < tg_search::kPhaseRing; ++i) (void)hipEventDestroy(s->phase_ev[i]);
    }
    for (int b = 0; b < 2; ++b) {
        if (s->rng_buf[b]) (void)hipFree(s->rng_buf[b]);
        if (s->ev_rng[b]) (void)hipEventDestroy(s->ev_rng[b]);
    }
    if (s->ev_sel) (void)hipEventDestroy(s->ev_sel);
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
    if (s->own_stream) { (void)hipStreamSynchronize(s->own_stream); (void)hipStreamDestroy(s->own_stream); }
    if (s->mt_base) (void)hipFree(s->mt_base);
    if (s->mt_cont) (void)hipFree(s->mt_cont);
    if (s->rng_words) (void)hipFree(s->rng_words);
    if (s->rng_pos0) (void)hipFree(s->rng_pos0);
    if (s->rng_snap) (void)hipFree(s->rng_snap);
    if (s->rng_snap_n) (void)hipFree(s->rng_snap_n);
    if (s->rng_cont_blk) (void)hipFree(s->rng_cont_blk);
    if (s->seed_pin) (void)hipHostFree(s->seed_pin);
    if (s->seed_ev) (void)hipEventDestroy(s->seed_ev);
    if (s->lag_pin) (void)hipHostFree(s->lag_pin);
    if (s->skip_pin) (void)hipHostFree(s->skip_pin);
    for (hipEvent_t e : s->lag_ev) if (e) (void)hipEventDestroy(e);
    if (s->consumed_pin) (void)hipHostFree(s->consumed_pin);
    if (s->noise_back) (void)hipHostFree(s->noise_back);
    if (s->noise_back_ev) (void)hipEventDestroy(s->noise_back_ev);
    if (s->noise_order_ev) (void)hipEventDestroy(s->noise_order_ev);
    delete s;
    return TG_OK;
}

int tg_search_grow(tg_search *s, int new_tree_size) {
    if (!s) return tg::fail(TG_ERR_ARG, "tg_search_grow: null argument");
    SearchDev &D = s->dev;
    if (new_tree_size <= D.N) return TG_OK;
    TG_HIP(hipSetDevice(s->cfg.device));
    TG_HIP(hipDeviceSynchronize());           // nothing may still be running on the old arrays
    const size_t T = D.T, n0 = D.N, n1 = (size_t)new_tree_size, A = s->A;
    std::vector<GrowItem> items;
#define GROW(field, per) items.push_back(GrowItem{reinterpret_cast<void **>(&D.field), sizeof(*D.field), (size_t)(per)});
    GROW(ch_index, A) GROW(ch_visits, A) GROW(ch_vl, A) GROW(ch_vsum, A) GROW(ch_policy, A) GROW(ch_value, A)
    GROW(action, A)
    GROW(node, 1)
#undef GROW
    // phase 1: every new array allocated and filled; on any failure the new arrays are released and nothing changed
    int rc = TG_OK;
    for (GrowItem &g : items) {
        const hipError_t e = hipMalloc(&g.fresh, T * n1 * g.per_node * g.elem);
        if (e != hipSuccess) {
            g.fresh = nullptr;
            rc = tg::fail(TG_ERR_HIP, "tg_search_grow: %s allocating %zu bytes - the pool keeps its %zu nodes per tree",
                          hipGetErrorString(e), T * n1 * g.per_node * g.elem, n0);
            break;
        }
        if ((rc = grow_fill(g, T, n0, n1))) break;
    }
    if (rc == TG_OK && hipDeviceSynchronize() != hipSuccess) rc = tg::fail(TG_ERR_HIP, "tg_search_grow: copy failed");
    if (rc != TG_OK) {
        for (GrowItem &g : items)
            if (g.fresh) (void)hipFree(g.fresh);
        (void)hipGetLastError();
        return rc;
    }
    // phase 2: swap (cannot fail)
    for (GrowItem &g : items) {
        for (void *&a : s->allocs)
            if (a == *g.field) a = g.fresh;
        (void)hipFree(*g.field);
        *g.field = g.fresh;
    }
    D.N = new_tree_size;
    s->cfg.tree_size = new_tree_size;
    return TG_OK;
}

int tg_search_set_zobrist(tg_search *s, const uint64_t *keys, size_t n) {
    if (!s || !keys) return tg::fail(TG_ERR_ARG, "tg_search_set_zobrist: null argument");
    if (n != (size_t)4 * s->NC) return tg::fail(TG_ERR_ARG, "tg_search_set_zobrist: expected %d keys", 4 * s->NC);
    TG_HIP(hipMemcpy(const_cast<uint64_t *>(s->dev.zob), keys, n * sizeof(uint64_t), hipMemcpyHostToDevice));
    return TG_OK;
}

int tg_search_set_root(tg_search *s, int tree, const tg_root_position *pos) {
    if (!s || !pos || !pos->cells) return tg::fail(TG_ERR_ARG, "tg_search_set_root: null argument");
    if (tree < 0 || tree >= s->dev.T) return tg::fail(TG_ERR_ARG, "tg_search_set_root: tree %d out of range", tree);
    if (pos->moves < 1) return tg::fail(TG_ERR_ARG, "tg_search_set_root: moves must be >= 1");
    if (pos->to_move != kBlack && pos->to_move != kWhite)
        return tg::fail(TG_ERR_ARG, "tg_search_set_root: to_move must be 1 or 2");
    if (s->dev.superko && !pos->hash_history && pos->moves > 1)
        return tg::fail(TG_ERR_ARG, "tg_search_set_root: hash_history required with check_superko");
    RootMeta m{};
    m.hash = pos->hash;
    m.moves = pos->moves;
    m.ko_pos = pos->ko_pos;
    m.ko_move = pos->ko_move;
    m.prev = pos->prev_move;
    m.prevprev = pos->prev_prev_move;
    m.to_move = pos->to_move;
    m.num_nodes = 0;
    m.hist_len = pos->hash_history ? (pos->moves < s->HMAX ? pos->moves : s->HMAX) : 1;
    std::memcpy(&s->st_cells[(size_t)tree * s->NC], pos->cells, s->NC);
    uint64_t *hist = &s->st_hist[(size_t)tree * s->HMAX];
    hist[0] = 0;
    if (pos->hash_history) std::memcpy(hist, pos->hash_history, (size_t)m.hist_len * sizeof(uint64_t));
    s->st_meta[tree] = m;
    s->st_dirty_tree[tree] = 1;
    s->st_dirty = true;
    return TG_OK;
}

static int flush_roots(tg_search *s, hipStream_t st) {
    if (!s->st_dirty) return TG_OK;
    const SearchDev &D = s->dev;
    size_t n_dirty = 0;
    for (uint8_t d : s->st_dirty_tree) n_dirty += d;
    // out of a pinned mirror: a copy out of pageable memory makes the host wait for the stream (self-play queues this
    // behind a whole move's kernels)
    const size_t cells_b = s->st_cells.size(), hist_b = s->st_hist.size() * sizeof(uint64_t), meta_b = s->st_meta.size() * sizeof(RootMeta);
    if (!s->roots_pin) {
        TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->roots_pin), cells_b + hist_b + meta_b + 16, hipHostMallocDefault));
        TG_HIP(hipEventCreateWithFlags(&s->roots_pin_ev, hipEventDisableTiming));
    }
    if (s->roots_pin_used) TG_HIP(hipEventSynchronize(s->roots_pin_ev));     // (the previous flush: long done)
    uint8_t *pin_cells = s->roots_pin;
    uint64_t *pin_hist = reinterpret_cast<uint64_t *>(s->roots_pin + ((cells_b + 7) & ~(size_t)7));
    RootMeta *pin_meta = reinterpret_cast<RootMeta *>(reinterpret_cast<unsigned char *>(pin_hist) + hist_b);
    if (n_dirty == (size_t)D.T) {
        std::memcpy(pin_cells, s->st_cells.data(), cells_b);
        std::memcpy(pin_meta, s->st_meta.data(), meta_b);
        TG_HIP(hipMemcpyAsync(D.root_cells, pin_cells, cells_b, hipMemcpyHostToDevice, st));
        if (D.superko) {
            std::memcpy(pin_hist, s->st_hist.data(), hist_b);
            TG_HIP(hipMemcpyAsync(D.root_hist, pin_hist, hist_b, hipMemcpyHostToDevice, st));
        }
        TG_HIP(hipMemcpyAsync(D.meta, pin_meta, meta_b, hipMemcpyHostToDevice, st));
    } else {
        for (int t = 0; t < D.T; ++t) {
            if (!s->st_dirty_tree[t]) continue;
            std::memcpy(pin_cells + (size_t)t * s->NC, &s->st_cells[(size_t)t * s->NC], s->NC);
            TG_HIP(hipMemcpyAsync(D.root_cells + (size_t)t * s->NC, pin_cells + (size_t)t * s->NC, s->NC, hipMemcpyHostToDevice, st));
            if (D.superko) {
                const size_t hb = (size_t)s->st_meta[t].hist_len * sizeof(uint64_t);
                std::memcpy(pin_hist + (size_t)t * s->HMAX, &s->st_hist[(size_t)t * s->HMAX], hb);
                TG_HIP(hipMemcpyAsync(D.root_hist + (size_t)t * s->HMAX, pin_hist + (size_t)t * s->HMAX, hb, hipMemcpyHostToDevice, st));
            }
            pin_meta[t] = s->st_meta[t];
            TG_HIP(hipMemcpyAsync(D.meta + t, pin_meta + t, sizeof(RootMeta), hipMemcpyHostToDevice, st));
        }
    }
    TG_HIP(hipEventRecord(s->roots_pin_ev, st));
    s->roots_pin_used = true;
    TG_HIP(hipMemsetAsync(D.err, 0, (size_t)D.T * sizeof(int32_t), st));
    std::fill(s->st_dirty_tree.begin(), s->st_dirty_tree.end(), 0);
    s->st_dirty = false;
    return TG_OK;
}

// make the most recently uploaded random window the active one (stream-ordered)
static int feed_streams_rest(tg_search *s);
static int install_rng(tg_search *s, hipStream_t st) {
    if (s->rng_rest_wait && s->rng_pending < 0) {   // second part of a split upload into the ACTIVE window
        TG_HIP(hipStreamWaitEvent(st, s->ev_rng[s->rng_active], 0));
        s->rng_rest_wait = false;
    }
    if (s->rng_pending < 0) return TG_OK;
    s->rng_active = s->rng_pending;
    s->rng_pending = -1;
    TG_HIP(hipStreamWaitEvent(st, s->ev_rng[s->rng_active], 0));
    s->dev.rng = s->rng_buf[s->rng_active];
    s->dev.rng_cap = s->rng_pending_cap;
    TG_HIP(hipMemsetAsync(s->dev.rng_cursor, 0, (size_t)s->dev.T * sizeof(int64_t), st));
    if (s->dev.cursor_pub) TG_HIP(hipMemsetAsync(s->dev.cursor_pub, 0, (size_t)s->dev.T * sizeof(int64_t), st));
    if (s->auto_rest) {                             // (feed_streams_impl's few-tree over-generation: the rest, behind this launch's wait)
        s->auto_rest = false;
        return feed_streams_rest(s);
    }
    return TG_OK;
}

static int after_select(tg_search *s, hipStream_t st) {
    TG_HIP(hipEventRecord(s->ev_sel, st));
    s->sel_recorded = true;
    return TG_OK;
}

int tg_search_set_rng(tg_search *s, const double *exp_stream_host, size_t stride, size_t count) {
    if (!s || !exp_stream_host) return tg::fail(TG_ERR_ARG, "tg_search_set_rng: null argument");
    if (count > stride) return tg::fail(TG_ERR_ARG, "tg_search_set_rng: count > stride");
    const int idx = 1 - s->rng_active;          // never the window a running kernel may read
    if ((int64_t)count > s->rng_buf_cap[idx]) {
        (void)hipFree(s->rng_buf[idx]);          // implicit device synchronisation (rare)
        s->rng_buf[idx] = nullptr;
        s->rng_buf_cap[idx] = 0;
        TG_HIP(hipMalloc(reinterpret_cast<void **>(&s->rng_buf[idx]), (size_t)s->dev.T * count * sizeof(double)));
        s->rng_buf_cap[idx] = (int64_t)count;
    }
    // rows are packed with pitch `count`; the kernels index with rng_cap = count
    TG_HIP(hipMemcpy2DAsync(s->rng_buf[idx], count * sizeof(double), exp_stream_host, stride * sizeof(double),
                            count * sizeof(double), s->dev.T, hipMemcpyHostToDevice, s->copy_stream));
    TG_HIP(hipEventRecord(s->ev_rng[idx], s->copy_stream));
    TG_HIP(hipStreamSynchronize(s->copy_stream));   // the host buffer may be released on return
    s->rng_pending = idx;
    s->rng_pending_cap = (int64_t)count;
    return TG_OK;
}

int tg_search_rng_consumed(tg_search *s, int64_t *consumed_host) {
    if (!s || !consumed_host) return tg::fail(TG_ERR_ARG, "tg_search_rng_consumed: null argument");
    // The kernels that move a cursor mirror it into host-mapped memory (SearchDev::cursor_pub); the host waits for the last
    // selection launch's event only - the forward pass / backup behind it keep running - and reads the mirror.  (Until round 6
    // a copy on the side stream: for a few trees that copy is a one-workgroup kernel, and it queued for a CU behind the forward
    // pass's 256 workgroups - the cursors, and with them the generation of the next random window, arrived ~57 us late.)
    if (!s->sel_recorded) {
        TG_HIP(hipDeviceSynchronize());
    } else if (s->dev.T >= 256) {
        // Many trees: the selection launch takes milliseconds - sleep instead of spinning (a rank's driver thread burnt a whole
        // core doing nothing; eight ranks share the host; hipEventSynchronize spins whatever the event's flags say here).
        for (;;) {
            const hipError_t q = hipEventQuery(s->ev_sel);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return tg::fail(TG_ERR_HIP, "tg_search_rng_consumed: hipEventQuery: %s", hipGetErrorString(q));
            std::this_thread::sleep_for(std::chrono::microseconds(s->dev.T >= 1024 ? 200 : 50));
        }
    } else {
        TG_HIP(hipEventSynchronize(s->ev_sel));
    }
    std::memcpy(consumed_host, s->consumed_pin, (size_t)s->dev.T * sizeof(int64_t));
    return TG_OK;
}

static int check_errors(tg_search *s) {
    std::vector<int32_t> err(s->dev.T);
    TG_HIP(hipMemcpy(err.data(), s->dev.err, err.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int t = 0; t < s->dev.T; ++t)
        if (err[t])
            return tg::fail(TG_ERR_OVERFLOW, "tree %d: %s%s", t, (err[t] & kErrPoolFull) ? "node pool full " : "",
                            (err[t] & kErrRngEmpty) ? "random window exhausted " : (err[t] & kErrPipeline) ? "selection pipeline stalled or path too deep " : "");
    return TG_OK;
}

int tg_search_root_planes(tg_search *s, float *planes_dev, void *stream) {
    if (!s || !planes_dev) return tg::fail(TG_ERR_ARG, "tg_search_root_planes: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    s->last_stream = st;
    s->stream_known = true;
    {
        int rc = flush_roots(s, st);
        if (rc) return rc;
        if ((rc = install_rng(s, st))) return rc;
    }
    if (s->S == 9) hipLaunchKernelGGL(root_kernel<9>, dim3(s->dev.T), dim3(64), 0, st, s->dev, planes_dev);
    else if (s->S == 13) hipLaunchKernelGGL(root_kernel<13>, dim3(s->dev.T), dim3(64), 0, st, s->dev, planes_dev);
    else hipLaunchKernelGGL(root_kernel<19>, dim3(s->dev.T), dim3(64), 0, st, s->dev, planes_dev);
    TG_HIP(hipGetLastError());
    return after_select(s, st);
}

int tg_search_select_puct(tg_search *s, int max_leaves, float *planes_dev, int32_t *n_leaves_dev, void *stream) {
    if (!s || !planes_dev) return tg::fail(TG_ERR_ARG, "tg_search_select_puct: null argument");
    if (max_leaves < 0 || max_leaves > s->dev.K)
        return tg::fail(TG_ERR_ARG, "tg_search_select_puct: max_leaves %d outside [0, batch_size]", max_leaves);
    hipStream_t st = static_cast<hipStream_t>(stream);
    s->last_stream = st;
    s->stream_known = true;
    {
        int rc = install_rng(s, st);
        if (rc) return rc;
    }
    // three wavefronts per tree (selector + two workers) cut the serial chain of a mini-batch:
    // 1.8x for one tree, still +0.4 % with 2048 trees per GPU (measured); TG_SELECT_SERIAL=1 keeps
    // the one-wavefront kernel (also used while the per-phase profile counters are on)
    static const bool force_serial = tg::knob("TG_SELECT_SERIAL") != nullptr;
    static const bool mpipe_prof = tg::knob("TG_MPIPE_PROF") != nullptr;     // phase counters of the multi-selector kernel
    const bool pipelined = !force_serial && (!s->dev.prof || mpipe_prof) && max_leaves <= kPipeMaxK;
    // few trees: the descents themselves are pipelined over four selector waves (+ four workers); with many
    // trees per CU the three-wave kernel keeps more trees resident
    static const int mpipe_max_trees = tg::knob("TG_SELECT_MPIPE_TREES") ? atoi(tg::knob("TG_SELECT_MPIPE_TREES")) : 256;
    // up to kXwMaxTrees trees: a second workgroup (on another CU) for the board work of every tree (TG_SELECT_SPLIT=0: off)
    // (TG_SHARED_DEVICE=1 - several processes on this GPU: a tree's three workgroups may not get resident together - turns it off, too)
    static const bool shared_device = getenv("TG_SHARED_DEVICE") && atoi(getenv("TG_SHARED_DEVICE")) != 0;
    const bool split = !shared_device && (!tg::knob("TG_SELECT_SPLIT") || atoi(tg::knob("TG_SELECT_SPLIT")) != 0);
    int split_rc = kSplitNoRoom;
#ifdef TG_SPLIT_PROF
    const bool split_prof_ok = true;
#else
    const bool split_prof_ok = !s->dev.prof;
#endif
    if (pipelined && split && split_prof_ok && s->S != 13 && s->dev.T <= kXwMaxTrees && s->dev.N <= (1 << 21)) {   // (13x13: no split instantiation)
        split_rc = s->S == 9 ? launch_split<9>(s, max_leaves, planes_dev, st) : launch_split<19>(s, max_leaves, planes_dev, st);
        if (split_rc < 0) return split_rc;
    }
    if (split_rc == TG_OK) {
        // launched
    } else if (pipelined && s->dev.T <= mpipe_max_trees) {
        int rc = s->S == 9 ? launch_mpipe<9>(s->dev, max_leaves, planes_dev, st)
                           : (s->S == 13 ? launch_mpipe<13>(s->dev, max_leaves, planes_dev, st) : launch_mpipe<19>(s->dev, max_leaves, planes_dev, st));
        if (rc) return rc;
    } else if (pipelined) {
        if (s->S == 9)
            hipLaunchKernelGGL(select_puct_pipe_kernel<9>, dim3(s->dev.T), dim3(192), 0, st, s->dev, max_leaves, planes_dev);
        else if (s->S == 13)
            hipLaunchKernelGGL(select_puct_pipe_kernel<13>, dim3(s->dev.T), dim3(192), 0, st, s->dev, max_leaves, planes_dev);
        else
            hipLaunchKernelGGL(select_puct_pipe_kernel<19>, dim3(s->dev.T), dim3(192), 0, st, s->dev, max_leaves, planes_dev);
    } else if (s->S == 9)
        hipLaunchKernelGGL(select_puct_kernel<9>, dim3(s->dev.T), dim3(64), 0, st, s->dev, max_leaves, planes_dev);
    else if (s->S == 13)
        hipLaunchKernelGGL(select_puct_kernel<13>, dim3(s->dev.T), dim3(64), 0, st, s->dev, max_leaves, planes_dev);
    else
        hipLaunchKernelGGL(select_puct_kernel<19>, dim3(s->dev.T), dim3(64), 0, st, s->dev, max_leaves, planes_dev);
    TG_HIP(hipGetLastError());
    if (n_leaves_dev)
        TG_HIP(hipMemcpyAsync(n_leaves_dev, s->dev.n_leaves, (size_t)s->dev.T * sizeof(int32_t),
                              hipMemcpyDeviceToDevice, st));
    return after_select(s, st);
}



int tg_search_profile(tg_search *s, int enable, long long *cycles_host) {
    if (!s) return tg::fail(TG_ERR_ARG, "tg_search_profile: null argument");
    if (s->last_stream) TG_HIP(hipStreamSynchronize(s->last_stream));
    else TG_HIP(hipDeviceSynchronize());
    if (enable && !s->dev.prof) {
        int rc = dev_alloc(s, &s->dev.prof, (size_t)16);
        if (rc) return rc;
    }
    if (cycles_host && s->dev.prof)
        TG_HIP(hipMemcpy(cycles_host, s->dev.prof, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    if (s->dev.prof) TG_HIP(hipMemset(s->dev.prof, 0, 16 * sizeof(long long)));
    if (!enable) s->dev.prof = nullptr;       // buffer stays allocated until destroy
    return TG_OK;
}

int tg_search_play(tg_search *s, const int32_t *moves_host, void *stream) {
    if (!s || !moves_host) return tg::fail(TG_ERR_ARG, "tg_search_play: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    s->last_stream = st;
    s->stream_known = true;
    int rc = flush_roots(s, st);
    if (rc) return rc;
    if (!s->moves_dev && (rc = dev_alloc(s, &s->moves_dev, (size_t)s->dev.T))) return rc;
    {
        const size_t n = (size_t)s->dev.T;
        if (!s->moves_pin) {
            TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->moves_pin), tg_search::kPinRing * n * sizeof(int32_t), hipHostMallocDefault));
            for (int i = 0; i < tg_search::kPinRing; ++i) TG_HIP(hipEventCreateWithFlags(&s->moves_ev[i], hipEventDisableTiming));
        }
        const int slot = (int)(s->moves_seq++ % tg_search::kPinRing);
        if (s->moves_ev_used[slot]) TG_HIP(hipEventSynchronize(s->moves_ev[slot]));
        int32_t *pin = s->moves_pin + (size_t)slot * n;
        std::memcpy(pin, moves_host, n * sizeof(int32_t));
        TG_HIP(hipMemcpyAsync(s->moves_dev, pin, n * sizeof(int32_t), hipMemcpyHostToDevice, st));   // (the caller's buffer is free)
        TG_HIP(hipEventRecord(s->moves_ev[slot], st));
        s->moves_ev_used[slot] = true;
    }
    if (s->S == 9) hipLaunchKernelGGL(play_kernel<9>, dim3(s->dev.T), dim3(64), 0, st, s->dev, s->moves_dev);
    else if (s->S == 13) hipLaunchKernelGGL(play_kernel<13>, dim3(s->dev.T), dim3(64), 0, st, s->dev, s->moves_dev);
    else hipLaunchKernelGGL(play_kernel<19>, dim3(s->dev.T), dim3(64), 0, st, s->dev, s->moves_dev);
    TG_HIP(hipGetLastError());
    return TG_OK;
}

int tg_search_read_positions(tg_search *s, uint8_t *cells_host, int32_t *moves_host, int32_t *to_move_host) {
    if (!s) return tg::fail(TG_ERR_ARG, "tg_search_read_positions: null argument");
    if (s->last_stream) TG_HIP(hipStreamSynchronize(s->last_stream));
    else TG_HIP(hipDeviceSynchronize());
    {
        int rc = flush_roots(s, s->last_stream);
        if (rc) return rc;
        if (s->last_stream) TG_HIP(hipStreamSynchronize(s->last_stream));
        else TG_HIP(hipDeviceSynchronize());
    }
    if (cells_host)
        TG_HIP(hipMemcpy(cells_host, s->dev.root_cells, (size_t)s->dev.T * s->NC, hipMemcpyDeviceToHost));
    if (moves_host || to_move_host) {
        std::vector<RootMeta> meta(s->dev.T);
        TG_HIP(hipMemcpy(meta.data(), s->dev.meta, meta.size() * sizeof(RootMeta), hipMemcpyDeviceToHost));
        for (int t = 0; t < s->dev.T; ++t) {
            if (moves_host) moves_host[t] = meta[t].moves;
            if (to_move_host) to_move_host[t] = meta[t].to_move;
        }
    }
    return TG_OK;
}

int tg_search_set_noise(tg_search *s, const double *noise_host) {
    if (!s || !noise_host) return tg::fail(TG_ERR_ARG, "tg_search_set_noise: null argument");
    const size_t n = (size_t)s->dev.T * s->A;
    s->noise_host.assign(noise_host, noise_host + n);
    s->noise_back_pending = false;                 // (a device-drawn noise still on its way back is superseded)
    if (!s->stream_known) {                        // no launch stream yet: the null stream, synchronously
        TG_HIP(hipDeviceSynchronize());
        TG_HIP(hipMemcpy(s->dev.noise, noise_host, n * sizeof(double), hipMemcpyHostToDevice));
        return TG_OK;
    }
    if (!s->noise_pin) {
        TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->noise_pin), tg_search::kPinRing * n * sizeof(double), hipHostMallocDefault));
        for (int i = 0; i < tg_search::kPinRing; ++i) TG_HIP(hipEventCreateWithFlags(&s->noise_ev[i], hipEventDisableTiming));
    }
    const int slot = (int)(s->noise_seq++ % tg_search::kPinRing);
    if (s->noise_ev_used[slot]) TG_HIP(hipEventSynchronize(s->noise_ev[slot]));     // (four uploads ago: long done)
    double *pin = s->noise_pin + (size_t)slot * n;
    std::memcpy(pin, noise_host, n * sizeof(double));
    // in stream order: behind the kernels that read the previous noise (the last search), ahead of the next selection
    TG_HIP(hipMemcpyAsync(s->dev.noise, pin, n * sizeof(double), hipMemcpyHostToDevice, s->last_stream));
    TG_HIP(hipEventRecord(s->noise_ev[slot], s->last_stream));
    s->noise_ev_used[slot] = true;
    return TG_OK;
}

// ---- library-owned legacy streams, device-resident (csrc/legacy_rng_device.h) ---------------

static void wait_prefill(tg_search *) {}          // (round 5's host-side background generator: nothing to wait for any more)

static int rng_alloc(tg_search *s) {
    if (s->mt_base) return TG_OK;
    const size_t T = (size_t)s->dev.T, words = T * tg_rng::kStateWords;
    TG_HIP(hipMalloc(reinterpret_cast<void **>(&s->mt_base), words * sizeof(uint32_t)));
    TG_HIP(hipMalloc(reinterpret_cast<void **>(&s->mt_cont), words * sizeof(uint32_t)));
    TG_HIP(hipMalloc(reinterpret_cast<void **>(&s->rng_pos0), T * sizeof(int)));
    TG_HIP(hipMalloc(reinterpret_cast<void **>(&s->rng_snap_n), T * sizeof(int)));
    TG_HIP(hipMalloc(reinterpret_cast<void **>(&s->rng_cont_blk), T * sizeof(long long)));
    TG_HIP(hipMemset(s->rng_snap_n, 0, T * sizeof(int)));
    TG_HIP(hipMemset(s->rng_cont_blk, 0, T * sizeof(long long)));
    TG_HIP(hipMemset(s->mt_base, 0, words * sizeof(uint32_t)));
    TG_HIP(hipMemset(s->mt_cont, 0, words * sizeof(uint32_t)));
    TG_HIP(hipStreamSynchronize(nullptr));               // (the fills above: not ordered before a non-blocking stream otherwise)
    TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->seed_pin), words * sizeof(uint32_t), hipHostMallocDefault));
    TG_HIP(hipEventCreateWithFlags(&s->seed_ev, hipEventDisableTiming));
    TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->lag_pin), tg_search::kLagRing * T * sizeof(long long), hipHostMallocMapped));
    TG_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->lag_pin_dev), s->lag_pin, 0));
    TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->skip_pin), tg_search::kLagRing * T, hipHostMallocMapped));
    TG_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->skip_pin_dev), s->skip_pin, 0));
    for (hipEvent_t &e : s->lag_ev) TG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->noise_back), T * (size_t)s->A * sizeof(double), hipHostMallocDefault));
    TG_HIP(hipEventCreateWithFlags(&s->noise_back_ev, hipEventDisableTiming));
    TG_HIP(hipEventCreateWithFlags(&s->noise_order_ev, hipEventDisableTiming));
    return TG_OK;
}

// seeds that have not gone up yet: staged in pinned memory, copied on the generation stream ahead of the launch that needs them
static int rng_sync_seeds(tg_search *s) {
    const int T = s->dev.T;
    bool any = false;
    for (int t = 0; t < T && !any; ++t) any = s->streams[t].dirty;
    if (!any) return TG_OK;
    if (s->seed_ev_used) TG_HIP(hipEventSynchronize(s->seed_ev));           // (the previous use of the staging rows)
    int t = 0;
    while (t < T) {
        if (!s->streams[t].dirty) { ++t; continue; }
        int t1 = t;
        for (; t1 < T && s->streams[t1].dirty; ++t1) {
            uint32_t *row = s->seed_pin + (size_t)t1 * tg_rng::kStateWords;
            std::memcpy(row, s->streams[t1].key, sizeof(s->streams[t1].key));
            row[tg_rng::kMtN] = (uint32_t)s->streams[t1].pos;
            s->streams[t1].dirty = false;
        }
        const size_t o = (size_t)t * tg_rng::kStateWords, n = (size_t)(t1 - t) * tg_rng::kStateWords;
        TG_HIP(hipMemcpyAsync(s->mt_base + o, s->seed_pin + o, n * sizeof(uint32_t), hipMemcpyHostToDevice, s->copy_stream));
        TG_HIP(hipMemsetAsync(s->rng_snap_n + t, 0, (size_t)(t1 - t) * sizeof(int), s->copy_stream));   // (snapshots of the old stream)
        t = t1;
    }
    TG_HIP(hipEventRecord(s->seed_ev, s->copy_stream));
    s->seed_ev_used = true;
    return TG_OK;
}

// The draws consumed since the device states were brought up to date, handed to the next generation launch: a slot of the
// host-mapped ring (the launch reads it over the bus: T x 8 bytes).  skip[t] != 0: that tree takes no part (its lag stays here).
static int rng_take_lag(tg_search *s, const uint8_t *skip, int *slot_out) {
    const int T = s->dev.T;
    const int slot = (int)(s->lag_seq++ % tg_search::kLagRing);
    if (s->lag_ev_used[slot]) TG_HIP(hipEventSynchronize(s->lag_ev[slot]));     // (eight launches ago: long done)
    long long *lag = s->lag_pin + (size_t)slot * T;
    unsigned char *sk = s->skip_pin + (size_t)slot * T;
    for (int t = 0; t < T; ++t) {
        const bool out = skip && skip[t];
        sk[t] = out ? 1 : 0;
        lag[t] = out ? 0 : (long long)s->streams[t].lag;
        if (!out) s->streams[t].lag = 0;
    }
    *slot_out = slot;
    return TG_OK;
}

static int rng_launch(tg_search *s, const tg_rng::FillArgs &a_in, int slot) {
    const int T = s->dev.T;
    tg_rng::FillArgs a = a_in;
    // scratch row of a tree: the piece's words from word 0 of the block it starts in (<= 623 words in front, <= 623 behind)
    const long long pitch = 2 * a.count + 2 * tg_rng::kMtN;
    if ((size_t)T * (size_t)pitch > s->rng_words_cap) {
        if (s->rng_words) (void)hipFree(s->rng_words);             // implicit device synchronisation (rare: the largest piece so far)
        s->rng_words = nullptr;
        s->rng_words_cap = 0;
        TG_HIP(hipMalloc(reinterpret_cast<void **>(&s->rng_words), (size_t)T * (size_t)pitch * sizeof(uint32_t)));
        s->rng_words_cap = (size_t)T * (size_t)pitch;
    }
    a.words = s->rng_words; a.words_pitch = pitch; a.pos0 = s->rng_pos0;
    if (T <= 64 && !tg::knob("TG_RNG_NO_SNAP")) {
        // state snapshots every kSnapEvery blocks of the whole window (legacy_rng_device.h): a later commit of the consumed draws
        // starts from the nearest one
        const long long window = a.noise ? 0 : (a.pitch > a.count ? a.pitch : a.count);
        const int want = (int)((2 * window / tg_rng::kMtN + 2) / tg_rng::kSnapEvery + 1);
        if (want > s->rng_snap_cap) {
            if (s->rng_snap) (void)hipFree(s->rng_snap);           // implicit device synchronisation (rare: the largest window so far)
            s->rng_snap = nullptr;
            s->rng_snap_cap = 0;
            TG_HIP(hipMalloc(reinterpret_cast<void **>(&s->rng_snap), (size_t)T * want * tg_rng::kMtN * sizeof(uint32_t)));
            s->rng_snap_cap = want;
            TG_HIP(hipMemsetAsync(s->rng_snap_n, 0, (size_t)T * sizeof(int), s->copy_stream));
        }
        a.snap = s->rng_snap; a.snap_cap = s->rng_snap_cap; a.snap_n = s->rng_snap_n;
    }
    a.cont_blk = s->rng_cont_blk;
    hipLaunchKernelGGL(tg_rng::rng_words_kernel, dim3(T), dim3(64), 0, s->copy_stream, a);
    TG_HIP(hipGetLastError());
    if (a.count > 0) {
        hipLaunchKernelGGL(tg_rng::rng_draws_kernel, dim3((unsigned)((a.count + 255) / 256), T), dim3(256), 0, s->copy_stream, a);
        TG_HIP(hipGetLastError());
    }
    if (slot >= 0) {
        TG_HIP(hipEventRecord(s->lag_ev[slot], s->copy_stream));
        s->lag_ev_used[slot] = true;
    }
    return TG_OK;
}

int tg_search_seed_stream(tg_search *s, int tree, const uint32_t *mt_key, int mt_pos) {
    if (!s || !mt_key) return tg::fail(TG_ERR_ARG, "tg_search_seed_stream: null argument");
    if (tree < 0 || tree >= s->dev.T) return tg::fail(TG_ERR_ARG, "tg_search_seed_stream: tree %d out of range", tree);
    if (mt_pos < 0 || mt_pos > 624) return tg::fail(TG_ERR_ARG, "tg_search_seed_stream: MT19937 position %d outside [0, 624]", mt_pos);
    TG_HIP(hipSetDevice(s->cfg.device));
    if (int rc = rng_alloc(s)) return rc;
    if (s->streams.empty()) s->streams.resize(s->dev.T);
    tg_search::DevStream &ds = s->streams[tree];
    std::memcpy(ds.key, mt_key, sizeof(ds.key));
    ds.pos = mt_pos;
    ds.lag = 0;
    ds.seeded = ds.dirty = true;
    s->win_left = 0;                               // the generated window belongs to the old stream
    return TG_OK;
}

int tg_search_stream_state(tg_search *s, int tree, uint32_t *mt_key_out, int *mt_pos_out) {
    if (!s || !mt_key_out || !mt_pos_out) return tg::fail(TG_ERR_ARG, "tg_search_stream_state: null argument");
    if (tree < 0 || tree >= s->dev.T || s->streams.empty() || !s->streams[tree].seeded)
        return tg::fail(TG_ERR_ARG, "tg_search_stream_state: tree %d has no stream", tree);
    tg_search::DevStream &ds = s->streams[tree];
    if (ds.dirty && ds.lag == 0) {                 // never used since it was seeded
        std::memcpy(mt_key_out, ds.key, sizeof(ds.key));
        *mt_pos_out = ds.pos;
        return TG_OK;
    }
    // bring the device states up to date (a generation launch of zero draws commits the consumed ones), then read this tree's
    int rc, slot = -1;
    if ((rc = rng_sync_seeds(s)) || (rc = rng_take_lag(s, nullptr, &slot))) return rc;
    tg_rng::FillArgs a{};
    a.base = s->mt_base; a.cont = s->mt_cont; a.lag = s->lag_pin_dev + (size_t)slot * s->dev.T;
    if ((rc = rng_launch(s, a, slot))) return rc;
    if (s->seed_ev_used) TG_HIP(hipEventSynchronize(s->seed_ev));
    uint32_t *row = s->seed_pin + (size_t)tree * tg_rng::kStateWords;
    TG_HIP(hipMemcpyAsync(row, s->mt_base + (size_t)tree * tg_rng::kStateWords, tg_rng::kStateWords * sizeof(uint32_t),
                          hipMemcpyDeviceToHost, s->copy_stream));
    TG_HIP(hipStreamSynchronize(s->copy_stream));
    std::memcpy(mt_key_out, row, 624 * sizeof(uint32_t));
    *mt_pos_out = (int)row[tg_rng::kMtN];
    return TG_OK;
}

// A window in parts (self-play: the window of a move's four phases is 1.7 MB at 16 boards; a chained PUCT search: one window
// for all its mini-batches): feed_streams_impl(first > 0) generates the first `first` draws of every tree's row only - enough
// for the first launch -, feed_streams_rest() / feed_streams_part() the remaining columns while that launch runs, each piece
// going on from the generator state the piece before it left (`mt_cont`); the launches that need them wait for the event.
static int feed_streams_impl(tg_search *s, size_t need, int force, size_t first);
static int feed_streams_rest(tg_search *s);

int tg_search_feed_streams(tg_search *s, size_t need, int force) { return feed_streams_impl(s, need, force, 0); }

static int feed_streams_impl(tg_search *s, size_t need, int force, size_t first) {
    if (!s) return tg::fail(TG_ERR_ARG, "tg_search_feed_streams: null argument");
    const int T = s->dev.T;
    if (s->rng_rest_cols && !s->auto_rest) {       // (an earlier window in parts was never completed: complete it first; an
        int rc = feed_streams_rest(s);             //  over-generated window's rest goes out behind the launch that installs it)
        if (rc) return rc;
    }
    if (s->streams.size() != (size_t)T) return tg::fail(TG_ERR_ARG, "tg_search_feed_streams: streams are not seeded");
    for (int t = 0; t < T; ++t)
        if (!s->streams[t].seeded) return tg::fail(TG_ERR_ARG, "tg_search_feed_streams: tree %d has no stream", t);
    if (need == 0 || (!force && s->win_left >= (int64_t)need)) return TG_OK;
    // Few trees (a single search tree): a window of `need` draws is a bound - a mini-batch consumes a fraction of it - and
    // every regeneration costs the launch that waits for it ~25 us.  Generate two windows' worth instead: the first `need`
    // draws now, the rest behind the first launch that uses them (install_rng), and the following mini-batches find their
    // draws there (win_left).  The draws are the stream's, whatever the window they were generated in.
    s->auto_rest = false;
    s->eager_need = 0;
    if (first == 0 && T <= 16 && need >= 2048 && need <= ((size_t)1 << 20)) {
        s->eager_need = need;
        first = need;
        // (one tree, ms per move at 9x9 / 19x19 with state snapshots: x2 1.475 / 10.94, x3 1.483 / 10.92, x4 1.464 / 10.91, x8 1.595 / 10.97)
        static const int overgen = tg::knob("TG_RNG_OVERGEN") ? std::max(1, atoi(tg::knob("TG_RNG_OVERGEN"))) : 2;
        need *= (size_t)overgen;
        s->auto_rest = true;
    }
    const int idx = 1 - s->rng_active;             // never the window a running kernel may read
    if ((int64_t)need > s->rng_buf_cap[idx]) {
        (void)hipFree(s->rng_buf[idx]);             // implicit device synchronisation (rare)
        s->rng_buf[idx] = nullptr;
        s->rng_buf_cap[idx] = 0;
        TG_HIP(hipMalloc(reinterpret_cast<void **>(&s->rng_buf[idx]), (size_t)T * need * sizeof(double)));
        s->rng_buf_cap[idx] = (int64_t)need;
    }
    const bool split = first > 0 && first < need;
    const size_t cols = split ? first : need;
    int rc, slot = -1;
    if ((rc = rng_sync_seeds(s)) || (rc = rng_take_lag(s, nullptr, &slot))) return rc;
    tg_rng::FillArgs a{};
    a.base = s->mt_base; a.cont = s->mt_cont; a.lag = s->lag_pin_dev + (size_t)slot * T;
    a.out = s->rng_buf[idx]; a.pitch = (long long)need; a.first = 0; a.count = (long long)cols;
    if ((rc = rng_launch(s, a, slot))) return rc;
    TG_HIP(hipEventRecord(s->ev_rng[idx], s->copy_stream));
    s->rng_pending = idx;
    s->rng_pending_cap = (int64_t)need;
    s->win_cap = s->win_left = (int64_t)need;
    s->win_used.assign(T, 0);
    s->rng_rest_idx = idx;
    s->rng_rest_first = cols;
    s->rng_rest_cols = split ? need - cols : 0;
    return TG_OK;
}

// columns [first, upto) of the window in parts, behind the piece before them
static int feed_streams_piece(tg_search *s, size_t upto) {
    const int idx = s->rng_rest_idx;
    const size_t need = s->rng_rest_first + s->rng_rest_cols, first = s->rng_rest_first;
    tg_rng::FillArgs a{};
    a.base = s->mt_base; a.cont = s->mt_cont; a.from_cont = 1;
    a.out = s->rng_buf[idx]; a.pitch = (long long)need; a.first = (long long)first; a.count = (long long)(upto - first);
    if (int rc = rng_launch(s, a, -1)) return rc;
    TG_HIP(hipEventRecord(s->ev_rng[idx], s->copy_stream));
    s->rng_rest_first = upto;
    s->rng_rest_cols = need - upto;
    return TG_OK;
}

static int feed_streams_rest(tg_search *s) {
    if (!s->rng_rest_cols) return TG_OK;
    if (int rc = feed_streams_piece(s, s->rng_rest_first + s->rng_rest_cols)) return rc;
    s->rng_rest_wait = true;                       // the next selection launch waits for this event
    return TG_OK;
}

// A window in parts continued piece by piece (tg_search_puct_chain): the columns up to `upto`, and `st` waits for them.  The
// device cursor of a tree never passes the columns its launched selections may consume (leaves x A each), so a launch only
// needs the pieces up to its own.
static int feed_streams_part(tg_search *s, size_t upto, hipStream_t st) {
    if (!s->rng_rest_cols) return TG_OK;
    const size_t need = s->rng_rest_first + s->rng_rest_cols;
    if (upto > need) upto = need;
    if (upto <= s->rng_rest_first) return TG_OK;
    const int idx = s->rng_rest_idx;
    if (int rc = feed_streams_piece(s, upto)) return rc;
    TG_HIP(hipStreamWaitEvent(st, s->ev_rng[idx], 0));
    return TG_OK;
}

static int advance_streams_impl(tg_search *s, int64_t *consumed_host, const uint8_t *skip, const int64_t *used_in = nullptr);
int tg_search_advance_streams(tg_search *s, int64_t *consumed_host) { return advance_streams_impl(s, consumed_host, nullptr); }

// skip[t] != 0: tree t's stream was replaced since the window was generated - what the device consumed there is not its (a
// self-play slot whose game ended while the device had already gone on to the next root)
// used_in: the device cursors, if the caller has them already (publish_cursors_kernel)
static int advance_streams_impl(tg_search *s, int64_t *consumed_host, const uint8_t *skip, const int64_t *used_in) {
    if (!s) return tg::fail(TG_ERR_ARG, "tg_search_advance_streams: null argument");
    const int T = s->dev.T;
    if (s->streams.size() != (size_t)T) return tg::fail(TG_ERR_ARG, "tg_search_advance_streams: streams are not seeded");
    std::vector<int64_t> used(T);
    if (used_in) {
        std::memcpy(used.data(), used_in, (size_t)T * sizeof(int64_t));
    } else {
        int rc = tg_search_rng_consumed(s, used.data());
        if (rc) return rc;
    }
    if (s->win_used.size() != (size_t)T) s->win_used.assign(T, 0);
    int64_t most = 0;
    for (int t = 0; t < T; ++t) {
        const int64_t delta = used[t] - s->win_used[t];     // the device cursor is cumulative within a window
        if (skip && skip[t]) {
            if (consumed_host) consumed_host[t] = 0;
            s->win_used[t] = used[t];
            most = std::max(most, used[t]);
            continue;
        }
        if (delta < 0 || used[t] > s->win_cap)
            return tg::fail(TG_ERR_ARG, "tg_search_advance_streams: tree %d consumed %lld draws, cursor %lld of a window of %lld", t,
                            (long long)delta, (long long)used[t], (long long)s->win_cap);
        s->streams[t].lag += delta;
        if (consumed_host) consumed_host[t] = delta;
        s->win_used[t] = used[t];
        most = std::max(most, used[t]);
    }
    s->win_left = s->win_cap - most;
    // Few trees: when the next mini-batch will not find its draws in this window, the next window is generated NOW - the
    // consumed draws committed, four windows' worth generated - under the forward pass and backup that are running, instead
    // of in front of the next selection launch (one 19x19 tree: five regenerations per move, ~90 us each).  Nothing is
    // consumed between here and that launch, so its tg_search_feed_streams finds the window in place.
    if (s->eager_need > 0 && !skip && s->win_left < (int64_t)s->eager_need) {
        const size_t need = s->eager_need;
        return feed_streams_impl(s, need, 0, 0);
    }
    return TG_OK;
}

// the device-drawn noise of the last tg_search_draw_noise, on the host (finish_move's own arithmetic, callers that ask for it)
static int noise_host_sync(tg_search *s) {
    if (!s->noise_back_pending) return TG_OK;
    TG_HIP(hipEventSynchronize(s->noise_back_ev));
    const size_t n = (size_t)s->dev.T * s->A;
    s->noise_host.assign(s->noise_back, s->noise_back + n);
    s->noise_back_pending = false;
    return TG_OK;
}

static int draw_noise_impl(tg_search *s, double *noise_host, const uint8_t *skip, hipEvent_t order_after = nullptr);
int tg_search_draw_noise(tg_search *s, double *noise_host) { return draw_noise_impl(s, noise_host, nullptr); }

// node.py:275-278 for every root: A draws of each tree's stream as Gumbel(0, 1) = -log(-log(1 - u)), generated on the device
// straight into the root noise rows, in stream order behind the kernels that read the previous noise and ahead of the next
// selection.  skip[t] != 0: tree t draws nothing (zero noise) - its stream must not move yet.
// order_after: an event the caller knows to lie behind every reader of the previous noise (self-play's chained moves: the
// cursor event, recorded right behind the root expansion - the noise is then generated UNDER the root's forward pass and
// backup instead of behind them); null: behind everything queued on the launch stream so far.
static int draw_noise_impl(tg_search *s, double *noise_host, const uint8_t *skip, hipEvent_t order_after) {
    if (!s) return tg::fail(TG_ERR_ARG, "tg_search_draw_noise: null argument");
    const int T = s->dev.T, A = s->A;
    if (s->streams.size() != (size_t)T) return tg::fail(TG_ERR_ARG, "tg_search_draw_noise: streams are not seeded");
    for (int t = 0; t < T; ++t)
        if (!(skip && skip[t]) && !s->streams[t].seeded) return tg::fail(TG_ERR_ARG, "tg_search_draw_noise: tree %d has no stream", t);
    int rc, slot = -1;
    if ((rc = noise_host_sync(s))) return rc;                       // (the staging rows are about to be rewritten)
    if ((rc = rng_sync_seeds(s)) || (rc = rng_take_lag(s, skip, &slot))) return rc;
    if (order_after) {
        TG_HIP(hipStreamWaitEvent(s->copy_stream, order_after, 0));
    } else if (s->stream_known) {
        TG_HIP(hipEventRecord(s->noise_order_ev, s->last_stream));
        TG_HIP(hipStreamWaitEvent(s->copy_stream, s->noise_order_ev, 0));
    } else {
        TG_HIP(hipDeviceSynchronize());
    }
    tg_rng::FillArgs a{};
    a.base = s->mt_base; a.cont = s->mt_cont; a.lag = s->lag_pin_dev + (size_t)slot * T;
    a.skip = s->skip_pin_dev + (size_t)slot * T;
    a.count = A; a.noise = s->dev.noise;
    if ((rc = rng_launch(s, a, slot))) return rc;
    TG_HIP(hipMemcpyAsync(s->noise_back, s->dev.noise, (size_t)T * A * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
    TG_HIP(hipEventRecord(s->noise_back_ev, s->copy_stream));
    s->noise_back_pending = true;
    if (s->stream_known) TG_HIP(hipStreamWaitEvent(s->last_stream, s->noise_back_ev, 0));
    else TG_HIP(hipStreamSynchronize(s->copy_stream));
    s->win_left = 0;                               // the noise sits between two windows
    if (noise_host) {
        if ((rc = noise_host_sync(s))) return rc;
        std::memcpy(noise_host, s->noise_host.data(), (size_t)T * A * sizeof(double));
    }
    return TG_OK;
}

// A launch stream that belongs to the handle (non-blocking, created on first request): self-play lanes run each engine on one of
// these instead of streams out of the host framework's pool - the hardware queue a stream lands on goes by creation order, and
// a pool of 32 streams created in one go leaves the library's own streams sharing queues with them (measured: two lanes on pool
// streams 2.99-3.10 M leaf-evals/s, on streams created here 3.65 M).
int tg_search_own_stream(tg_search *s, void **stream_out) {
    if (!s || !stream_out) return tg::fail(TG_ERR_ARG, "tg_search_own_stream: null argument");
    TG_HIP(hipSetDevice(s->cfg.device));
    if (!s->own_stream) TG_HIP(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
    *stream_out = s->own_stream;
    return TG_OK;
}

// ---- test hooks of the device streams (tests/test_gpu_rng.py) ----
// columns [first, first + count) of tree `tree`'s row of the most recently generated window
int tg_search_debug_read_window(tg_search *s, int tree, size_t first, size_t count, double *out_host) {
    if (!s || (!out_host && count)) return tg::fail(TG_ERR_ARG, "tg_search_debug_read_window: null argument");
    const int idx = s->rng_pending >= 0 ? s->rng_pending : s->rng_active;
    const int64_t pitch = s->rng_pending >= 0 ? s->rng_pending_cap : s->dev.rng_cap;
    if (tree < 0 || tree >= s->dev.T || !s->rng_buf[idx] || (int64_t)(first + count) > pitch)
        return tg::fail(TG_ERR_ARG, "tg_search_debug_read_window: outside the window (%lld draws per tree)", (long long)pitch);
    TG_HIP(hipStreamSynchronize(s->copy_stream));
    TG_HIP(hipMemcpy(out_host, s->rng_buf[idx] + (size_t)tree * pitch + first, count * sizeof(double), hipMemcpyDeviceToHost));
    return TG_OK;
}

// What a search does to the streams, without a search: per step a window of steps[i] + slack draws is generated - whole
// (part == 0) or in parts of `part` draws (first part, continued pieces, the rest: feed_streams_impl / _part / _rest) - and
// steps[i] draws of every tree count as consumed.
int tg_search_debug_stream_walk(tg_search *s, const int64_t *steps, int n_steps, int64_t slack, int64_t part) {
    if (!s || !steps || n_steps < 0 || slack < 0 || part < 0) return tg::fail(TG_ERR_ARG, "tg_search_debug_stream_walk: bad argument");
    int rc;
    for (int i = 0; i < n_steps; ++i) {
        if (steps[i] < 0) return tg::fail(TG_ERR_ARG, "tg_search_debug_stream_walk: negative step");
        const size_t need = (size_t)(steps[i] + slack);
        if (need == 0) continue;
        if ((rc = feed_streams_impl(s, need, 1, part > 0 ? (size_t)part : 0))) return rc;
        if (part > 0) {
            for (size_t upto = (size_t)(2 * part); upto + (size_t)part < need; upto += (size_t)part)
                if ((rc = feed_streams_part(s, upto, s->copy_stream))) return rc;
            if ((rc = feed_streams_rest(s))) return rc;
            s->rng_rest_wait = false;
        }
        for (int t = 0; t < s->dev.T; ++t) s->streams[t].lag += steps[i];
        s->win_left = 0;
    }
    return TG_OK;
}

// D: the engine's device view or a slice of it (sub_dev: D.T trees from some tree on); the kernel variant goes by the
// ENGINE's tree count (how crowded the CUs are)
// `limit`: leaf slots per tree (the stride of the strided layout); `max_n`: the most descents any tree of this launch makes - what the
// pipelined kernel's per-phase tables are sized against (until the end of round 6 `limit` stood in for it: a shard at 800
// simulations per move - 800 slots, phases of ~200 descents - fell to the one-wavefront kernel, 0.96 instead of 3.4 M at 16 boards)
static int launch_gumbel_select(tg_search *s, const SearchDev &D, const int32_t *nc_dev, const int32_t *mc_dev, const int32_t *off,
                                int limit, int max_n, float *planes_dev, hipStream_t st) {
    const int T = D.T;
    static const bool force_serial = tg::knob("TG_SELECT_SERIAL") != nullptr;
    // workers per tree: two when the trees crowd the CUs, six when there are CUs to spare, ten for a handful of trees - a phase's
    // dozen entries (expansion + leaf, see the kernel) then take two rounds instead of three: one tree 0.53 -> 0.55 M, 4 boards
    // 1.55 -> 1.60 M, 16 boards 3.60 -> 3.65 M leaf evaluations/s; fifteen: no better (TG_GUMBEL_WORKERS overrides)
    static const int workers_env = tg::knob("TG_GUMBEL_WORKERS") ? atoi(tg::knob("TG_GUMBEL_WORKERS")) : 0;
    const int workers = workers_env ? workers_env : (s->dev.T <= 28 ? 10 : (s->dev.T <= 128 ? 6 : 2));
    const bool gpipe = s->S == 9 && !force_serial && max_n <= kPipeMaxK / 2 && D.N <= (1 << 21);   // (paths as node << 10 | edge)
    const bool gpipe19 = s->S == 19 && !force_serial && max_n <= kPipeMaxK / 2 && D.N <= (1 << 21);
    SearchDev Dk = D;
    Dk.gumbel_one_by_one = tg::knob("TG_GUMBEL_ONE_BY_ONE") ? 1 : 0;           // (read per call: a test toggles it)
    if (gpipe && workers == 15)
        hipLaunchKernelGGL((select_gumbel_pipe_kernel<9, 15>), dim3(T), dim3(64 * 16), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    else if (gpipe && workers == 10)
        hipLaunchKernelGGL((select_gumbel_pipe_kernel<9, 10>), dim3(T), dim3(64 * 11), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    else if (gpipe && workers == 6)
        hipLaunchKernelGGL((select_gumbel_pipe_kernel<9, 6>), dim3(T), dim3(64 * 7), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    else if (gpipe && workers == 4)
        hipLaunchKernelGGL((select_gumbel_pipe_kernel<9, 4>), dim3(T), dim3(64 * 5), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    else if (gpipe)
        hipLaunchKernelGGL((select_gumbel_pipe_kernel<9, 2>), dim3(T), dim3(192), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    else if (s->S == 13 && !force_serial && max_n <= kPipeMaxK / 2 && D.N <= (1 << 21) && workers >= 6)
        hipLaunchKernelGGL((select_gumbel_pipe_kernel<13, 6>), dim3(T), dim3(64 * 7), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    else if (s->S == 13 && !force_serial && max_n <= kPipeMaxK / 2 && D.N <= (1 << 21))
        hipLaunchKernelGGL((select_gumbel_pipe_kernel<13, 2>), dim3(T), dim3(192), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    else if (gpipe19 && (workers >= 4 || !workers_env))           // (a 19x19 workgroup has its CU to itself with two workers as well: 256 boards 1.12 -> 1.17 M with four)
        hipLaunchKernelGGL((select_gumbel_pipe_kernel<19, 4>), dim3(T), dim3(64 * 5), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    else if (gpipe19)
        hipLaunchKernelGGL((select_gumbel_pipe_kernel<19, 2>), dim3(T), dim3(192), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    else if (s->S == 9)
        hipLaunchKernelGGL(select_gumbel_kernel<9>, dim3(T), dim3(64), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    else if (s->S == 13)
        hipLaunchKernelGGL(select_gumbel_kernel<13>, dim3(T), dim3(64), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    else
        hipLaunchKernelGGL(select_gumbel_kernel<19>, dim3(T), dim3(64), 0, st, Dk, nc_dev, mc_dev, limit, off, planes_dev);
    TG_HIP(hipGetLastError());
    return TG_OK;
}

// trees [t0, t0 + n) of the engine as a device view of their own (every per-tree array moved on; same kernels)
static SearchDev sub_dev(const tg_search *s, int t0, int n) {
    SearchDev D = s->dev;
    const size_t o = (size_t)t0, N = (size_t)D.N, A = (size_t)s->A, K = (size_t)D.K;
    D.ch_index += o * N * A; D.ch_visits += o * N * A; D.ch_vl += o * N * A;
    D.ch_vsum += o * N * A; D.ch_policy += o * N * A; D.ch_value += o * N * A; D.action += o * N * A;
    D.node += o * N;
    D.noise += o * A;
    D.root_cells += o * s->NC; D.root_hist += o * s->HMAX; D.meta += o;
    D.q_node += o * K; D.q_pnode += o * K; D.q_pedge += o * K; D.q_depth += o * K; D.q_path += o * K * kPathCap;
    D.n_leaves += o;
    D.rng += o * (size_t)D.rng_cap; D.rng_cursor += o; D.err += o;
    if (D.cursor_pub) D.cursor_pub += o;
    D.T = n;
    return D;
}

static int launch_backup(tg_search *s, const SearchDev &D, const float *policy_dev, const float *value_dev, int slots_per_tree,
                         const int32_t *off, int use_logit, hipStream_t st) {
    const bool few = s->dev.T <= 64;          // few trees: 16 waves per tree
    const dim3 grid(D.T), block(64 * (few ? 16 : 8));
    if (s->S == 13) {
        hipLaunchKernelGGL((backup_kernel<13, 8>), grid, dim3(64 * 8), 0, st, D, policy_dev, value_dev, slots_per_tree, off, use_logit);
    } else if (s->S == 9 && few) {
        hipLaunchKernelGGL((backup_kernel<9, 16>), grid, block, 0, st, D, policy_dev, value_dev, slots_per_tree, off, use_logit);
    } else if (s->S == 9) {
        hipLaunchKernelGGL((backup_kernel<9, 8>), grid, block, 0, st, D, policy_dev, value_dev, slots_per_tree, off, use_logit);
    } else if (few) {
        hipLaunchKernelGGL((backup_kernel<19, 16>), grid, block, 0, st, D, policy_dev, value_dev, slots_per_tree, off, use_logit);
    } else {
        hipLaunchKernelGGL((backup_kernel<19, 8>), grid, block, 0, st, D, policy_dev, value_dev, slots_per_tree, off, use_logit);
    }
    TG_HIP(hipGetLastError());
    return TG_OK;
}

int tg_search_select_gumbel(tg_search *s, const int32_t *num_considered_host, const int32_t *max_count_host,
                            int slots_per_tree, float *planes_dev, void *stream) {
    if (!s || !num_considered_host || !max_count_host || !planes_dev)
        return tg::fail(TG_ERR_ARG, "tg_search_select_gumbel: null argument");
    const bool packed = slots_per_tree == 0;
    if (!packed && (slots_per_tree < 1 || slots_per_tree > s->dev.K))
        return tg::fail(TG_ERR_ARG, "tg_search_select_gumbel: slots_per_tree %d outside [1, batch_size]", slots_per_tree);
    const int T = s->dev.T;
    const int limit = packed ? s->dev.K : slots_per_tree;
    // staging: [num_considered | max_count | leaf offsets] in a ring of pinned buffers (a pageable staging vector
    // needed a stream synchronisation per phase: the host then waited for the previous phase's forward and backup)
    if (!s->phase_pin) {
        TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->phase_pin), (size_t)tg_search::kPhaseRing * 3 * T * sizeof(int32_t)));
        for (int i = 0; i < tg_search::kPhaseRing; ++i) TG_HIP(hipEventCreateWithFlags(&s->phase_ev[i], hipEventDisableTiming));
    }
    const int ring = (int)(s->phase_seq % tg_search::kPhaseRing);
    if (s->phase_ev_used[ring]) TG_HIP(hipEventSynchronize(s->phase_ev[ring]));
    int32_t *phase_host = s->phase_pin + (size_t)ring * 3 * T;
    int64_t total = 0, max_n = 0;
    for (int t = 0; t < T; ++t) {
        const int64_t n = (int64_t)num_considered_host[t] * max_count_host[t];
        if (num_considered_host[t] < 0 || max_count_host[t] < 0 || n > limit)
            return tg::fail(TG_ERR_ARG, "tg_search_select_gumbel: tree %d phase does not fit %d slots", t, limit);
        max_n = n > max_n ? n : max_n;
        phase_host[t] = num_considered_host[t];
        phase_host[T + t] = max_count_host[t];
        phase_host[2 * (size_t)T + t] = (int32_t)total;
        total += n;
    }
    if (total > (int64_t)T * s->dev.K)
        return tg::fail(TG_ERR_ARG, "tg_search_select_gumbel: %lld leaves exceed T * batch_size", (long long)total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    s->last_stream = st;
    s->stream_known = true;
    if (!s->phase_dev) {
        int rc = dev_alloc(s, &s->phase_dev, (size_t)3 * T);
        if (rc) return rc;
    }
    TG_HIP(hipMemcpyAsync(s->phase_dev, phase_host, (size_t)3 * T * sizeof(int32_t), hipMemcpyHostToDevice, st));
    TG_HIP(hipEventRecord(s->phase_ev[ring], st));
    s->phase_ev_used[ring] = true;
    s->phase_seq += 1;
    {
        int rc = install_rng(s, st);
        if (rc) return rc;
    }
    s->packed_leaves = packed;
    const int32_t *off = packed ? s->phase_dev + 2 * (size_t)T : nullptr;
    int rc = launch_gumbel_select(s, s->dev, s->phase_dev, s->phase_dev + T, off, limit, (int)max_n, planes_dev, st);
    if (rc) return rc;
    return after_select(s, st);
}

int tg_search_backup(tg_search *s, const float *policy_dev, const float *value_dev, int slots_per_tree,
                     int use_logit, void *stream) {
    if (!s || !policy_dev || !value_dev) return tg::fail(TG_ERR_ARG, "tg_search_backup: null argument");
    const bool packed = slots_per_tree == 0;
    if (packed && !s->packed_leaves)
        return tg::fail(TG_ERR_ARG, "tg_search_backup: packed layout (slots_per_tree 0) needs a preceding packed tg_search_select_gumbel");
    if (!packed && (slots_per_tree < 1 || slots_per_tree > s->dev.K))
        return tg::fail(TG_ERR_ARG, "tg_search_backup: slots_per_tree %d outside [1, batch_size]", slots_per_tree);
    hipStream_t st = static_cast<hipStream_t>(stream);
    s->last_stream = st;
    s->stream_known = true;
    const int32_t *off = packed ? s->phase_dev + 2 * (size_t)s->dev.T : nullptr;
    return launch_backup(s, s->dev, policy_dev, value_dev, slots_per_tree, off, use_logit, st);
}

}  // extern "C"

// root records of all trees -> s->roots_host (pinned); also surfaces the sticky error flags
static size_t root_rec_bytes(int A) {
    size_t ints = 16 + (size_t)3 * A * 4;
    if (ints % 8) ints += 4;
    return ints + (size_t)2 * A * 8;
}
static int alloc_root_records(tg_search *s) {                   // [T] records, then [T] RootTail
    if (s->roots_dev) return TG_OK;
    const size_t bytes = (root_rec_bytes(s->A) + sizeof(RootTail)) * (size_t)s->dev.T;
    TG_HIP(hipMalloc(reinterpret_cast<void **>(&s->roots_dev), bytes));
    TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->roots_host), bytes));
    return TG_OK;
}
static void parse_root_records(const tg_search *s, int32_t *num_children_host, int32_t *node_visits_host, float *raw_value_host,
                               int32_t *action_host, int32_t *visits_host, int32_t *virtual_loss_host, double *value_sum_host,
                               double *policy_host) {
    const size_t A = s->A, T = s->dev.T, rec = root_rec_bytes((int)A);
    for (size_t t = 0; t < T; ++t) {
        const unsigned char *r = s->roots_host + rec * t;
        const int32_t *head = reinterpret_cast<const int32_t *>(r);
        const int32_t *vis = head + 4, *vl = vis + A, *act = vl + A;
        const double *vsum = reinterpret_cast<const double *>(r + rec - 2 * A * 8), *pol = vsum + A;
        if (num_children_host) num_children_host[t] = head[0];
        if (node_visits_host) node_visits_host[t] = head[1];
        if (raw_value_host) std::memcpy(&raw_value_host[t], &head[2], 4);
        if (visits_host) std::memcpy(visits_host + t * A, vis, A * 4);
        if (virtual_loss_host) std::memcpy(virtual_loss_host + t * A, vl, A * 4);
        if (action_host) std::memcpy(action_host + t * A, act, A * 4);
        if (value_sum_host) std::memcpy(value_sum_host + t * A, vsum, A * 8);
        if (policy_host) std::memcpy(policy_host + t * A, pol, A * 8);
    }
}
static int root_record_errors(const tg_search *s) {
    const size_t rec = root_rec_bytes(s->A);
    for (int t = 0; t < s->dev.T; ++t) {
        const int32_t err = reinterpret_cast<const int32_t *>(s->roots_host + rec * t)[3];
        if (err)
            return tg::fail(TG_ERR_OVERFLOW, "tree %d: %s%s", t, (err & kErrPoolFull) ? "node pool full " : "",
                            (err & kErrRngEmpty) ? "random window exhausted " : (err & kErrPipeline) ? "selection pipeline stalled or path too deep " : "");
    }
    return TG_OK;
}
// the records + the device's own decision (finish_roots_kernel), queued on `st`; the host waits for fin_ev, not for the stream
static int launch_finish_roots(tg_search *s, const int32_t *state_host, int max_moves, hipStream_t st) {
    const int T = s->dev.T, A = s->A;
    const size_t rec = root_rec_bytes(A), n = (size_t)4 * T;
    if (int rc = alloc_root_records(s)) return rc;
    if (!s->moves_dev) { if (int rc = dev_alloc(s, &s->moves_dev, (size_t)T)) return rc; }
    if (!s->state_dev) {
        if (int rc = dev_alloc(s, &s->state_dev, n)) return rc;
        TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->state_pin), tg_search::kPinRing * n * sizeof(int32_t), hipHostMallocDefault));
        for (int i = 0; i < tg_search::kPinRing; ++i) TG_HIP(hipEventCreateWithFlags(&s->state_ev[i], hipEventDisableTiming));
        TG_HIP(hipEventCreateWithFlags(&s->fin_ev, hipEventDisableTiming));
    }
    const int slot = (int)(s->state_seq++ % tg_search::kPinRing);
    if (s->state_ev_used[slot]) TG_HIP(hipEventSynchronize(s->state_ev[slot]));
    int32_t *pin = s->state_pin + (size_t)slot * n;
    std::memcpy(pin, state_host, n * sizeof(int32_t));
    TG_HIP(hipMemcpyAsync(s->state_dev, pin, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    TG_HIP(hipEventRecord(s->state_ev[slot], st));
    s->state_ev_used[slot] = true;
    RootTail *tail_dev = reinterpret_cast<RootTail *>(s->roots_dev + rec * T);
    hipLaunchKernelGGL(finish_roots_kernel, dim3(T), dim3(64), 0, st, s->dev, A, s->roots_dev, rec, s->state_dev, max_moves,
                       s->moves_dev, tail_dev);
    TG_HIP(hipGetLastError());
    TG_HIP(hipMemcpyAsync(s->roots_host, s->roots_dev, (rec + sizeof(RootTail)) * T, hipMemcpyDeviceToHost, st));
    TG_HIP(hipEventRecord(s->fin_ev, st));
    return TG_OK;
}
static int gather_roots(tg_search *s) {
    const int T = s->dev.T, A = s->A;
    const size_t rec = root_rec_bytes(A);
    if (int rc = alloc_root_records(s)) return rc;
    hipStream_t st = s->last_stream;
    hipLaunchKernelGGL(gather_roots_kernel, dim3(T), dim3(64), 0, st, s->dev, A, s->roots_dev, rec);
    TG_HIP(hipGetLastError());
    TG_HIP(hipMemcpyAsync(s->roots_host, s->roots_dev, rec * T, hipMemcpyDeviceToHost, st));
    TG_HIP(hipStreamSynchronize(st));
    return root_record_errors(s);
}

extern "C" {

int tg_search_read_roots(tg_search *s, int32_t *num_children_host, int32_t *action_host,
                         int32_t *visits_host) {
    if (!s || !num_children_host || !action_host || !visits_host)
        return tg::fail(TG_ERR_ARG, "tg_search_read_roots: null argument");
    return tg_search_read_root_stats(s, num_children_host, nullptr, nullptr, action_host, visits_host, nullptr, nullptr, nullptr);
}

int tg_search_read_root_stats(tg_search *s, int32_t *num_children_host, int32_t *node_visits_host,
                              float *raw_value_host, int32_t *action_host, int32_t *visits_host,
                              int32_t *virtual_loss_host, double *value_sum_host, double *policy_host) {
    if (!s) return tg::fail(TG_ERR_ARG, "tg_search_read_root_stats: null argument");
    int rc = gather_roots(s);
    if (rc) return rc;
    parse_root_records(s, num_children_host, node_visits_host, raw_value_host, action_host, visits_host, virtual_loss_host,
                       value_sum_host, policy_host);
    return TG_OK;
}

int tg_search_read_path(tg_search *s, int tree, int slot, int32_t *nodes_host, int32_t *edges_host, int capacity,
                        int32_t *length_host) {
    if (!s || !nodes_host || !edges_host || !length_host)
        return tg::fail(TG_ERR_ARG, "tg_search_read_path: null argument");
    if (tree < 0 || tree >= s->dev.T || slot < 0 || slot >= s->dev.K)
        return tg::fail(TG_ERR_ARG, "tg_search_read_path: tree %d / slot %d out of range", tree, slot);
    if (s->last_stream) TG_HIP(hipStreamSynchronize(s->last_stream));
    else TG_HIP(hipDeviceSynchronize());
    const SearchDev &D = s->dev;
    // (the queue entries stay valid until the next selection launch, also after tg_search_backup)
    // leaf -> root over the parent pointers, then reversed into the reference's root -> leaf order
    std::vector<int32_t> nodes, edges;
    int32_t cur = -1, e = -1;
    const size_t q = (size_t)tree * D.K + slot;
    TG_HIP(hipMemcpy(&cur, D.q_pnode + q, sizeof(int32_t), hipMemcpyDeviceToHost));
    TG_HIP(hipMemcpy(&e, D.q_pedge + q, sizeof(int32_t), hipMemcpyDeviceToHost));
    while (cur >= 0) {
        nodes.push_back(cur);
        edges.push_back(e);
        const size_t cs = (size_t)tree * D.N + cur;
        NodeRec rec;
        TG_HIP(hipMemcpy(&rec, D.node + cs, sizeof(NodeRec), hipMemcpyDeviceToHost));
        e = rec.pedge;
        cur = rec.parent;
        if ((int)nodes.size() > D.N) return tg::fail(TG_ERR_HIP, "tg_search_read_path: parent chain does not end");
    }
    *length_host = (int32_t)nodes.size();
    if ((int)nodes.size() > capacity)
        return tg::fail(TG_ERR_ARG, "tg_search_read_path: path of %zu steps exceeds capacity %d", nodes.size(), capacity);
    for (size_t i = 0; i < nodes.size(); ++i) {
        nodes_host[i] = nodes[nodes.size() - 1 - i];
        edges_host[i] = edges[nodes.size() - 1 - i];
    }
    return TG_OK;
}

int tg_search_read_queue(tg_search *s, int tree, int32_t *node_index_host, int capacity, int32_t *count_host) {
    if (!s || !node_index_host || !count_host) return tg::fail(TG_ERR_ARG, "tg_search_read_queue: null argument");
    if (tree < 0 || tree >= s->dev.T) return tg::fail(TG_ERR_ARG, "tg_search_read_queue: tree %d out of range", tree);
    if (s->last_stream) TG_HIP(hipStreamSynchronize(s->last_stream));
    else TG_HIP(hipDeviceSynchronize());
    const SearchDev &D = s->dev;
    int32_t n = 0;
    TG_HIP(hipMemcpy(&n, D.n_leaves + tree, sizeof(int32_t), hipMemcpyDeviceToHost));
    *count_host = n;
    if (n > capacity) return tg::fail(TG_ERR_ARG, "tg_search_read_queue: %d leaves exceed capacity %d", n, capacity);
    if (n > 0)
        TG_HIP(hipMemcpy(node_index_host, D.q_node + (size_t)tree * D.K, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return TG_OK;
}

int tg_search_num_nodes(tg_search *s, int32_t *num_nodes_host) {
    if (!s || !num_nodes_host) return tg::fail(TG_ERR_ARG, "tg_search_num_nodes: null argument");
    if (s->last_stream) TG_HIP(hipStreamSynchronize(s->last_stream));
    else TG_HIP(hipDeviceSynchronize());
    int rc = check_errors(s);
    if (rc) return rc;
    std::vector<RootMeta> meta(s->dev.T);
    TG_HIP(hipMemcpy(meta.data(), s->dev.meta, meta.size() * sizeof(RootMeta), hipMemcpyDeviceToHost));
    for (int t = 0; t < s->dev.T; ++t) num_nodes_host[t] = meta[t].num_nodes;
    return TG_OK;
}

int tg_search_read_node(tg_search *s, int tree, int node, int32_t *num_children, int32_t *node_visits,
                        int32_t *node_virtual_loss, int32_t *action, int32_t *children_index, int32_t *children_visits,
                        int32_t *children_virtual_loss, double *children_value_sum, double *children_policy,
                        double *children_value, float *node_value_sum, float *raw_value) {
    if (!s) return tg::fail(TG_ERR_ARG, "tg_search_read_node: null argument");
    if (tree < 0 || tree >= s->dev.T || node < 0 || node >= s->dev.N)
        return tg::fail(TG_ERR_ARG, "tg_search_read_node: tree/node out of range");
    const size_t A = s->A;
    const size_t ints = 32 + (((size_t)4 * A * 4 + 7) & ~(size_t)7), rec = ints + (size_t)3 * A * 8;
    if (!s->node_dev) {
        TG_HIP(hipMalloc(reinterpret_cast<void **>(&s->node_dev), rec));
        TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->node_host), rec));
    }
    hipStream_t st = s->last_stream;
    hipLaunchKernelGGL(gather_node_kernel, dim3(1), dim3(64), 0, st, s->dev, (int)A, tree, node, s->node_dev);
    TG_HIP(hipGetLastError());
    TG_HIP(hipMemcpyAsync(s->node_host, s->node_dev, rec, hipMemcpyDeviceToHost, st));
    TG_HIP(hipStreamSynchronize(st));
    const int32_t *head = reinterpret_cast<const int32_t *>(s->node_host);
    if (head[5]) {
        // another tree's sticky error is reported by the calls that read all trees; this one checks its own
        const int err = head[5];
        return tg::fail(TG_ERR_OVERFLOW, "tree %d: %s%s(err 0x%x)", tree, (err & kErrPoolFull) ? "node pool full " : "",
                        (err & kErrRngEmpty) ? "random window exhausted " : (err & kErrPipeline) ? "selection pipeline stalled or path too deep " : "", err);
    }
    const int32_t *idx = head + 8, *vis = idx + A, *vl = vis + A, *act = vl + A;
    const double *vsum = reinterpret_cast<const double *>(s->node_host + ints), *pol = vsum + A, *val = pol + A;
    if (num_children) *num_children = head[0];
    if (node_visits) *node_visits = head[1];
    if (node_virtual_loss) *node_virtual_loss = head[2];
    if (node_value_sum) std::memcpy(node_value_sum, &head[3], 4);
    if (raw_value) std::memcpy(raw_value, &head[4], 4);
    if (children_index) std::memcpy(children_index, idx, A * 4);
    if (children_visits) std::memcpy(children_visits, vis, A * 4);
    if (children_virtual_loss) std::memcpy(children_virtual_loss, vl, A * 4);
    if (action) std::memcpy(action, act, A * 4);
    if (children_value_sum) std::memcpy(children_value_sum, vsum, A * 8);
    if (children_policy) std::memcpy(children_policy, pol, A * 8);
    if (children_value) std::memcpy(children_value, val, A * 8);
    return TG_OK;
}

// num_nodes of the tree the last tg_search_read_node read, as of that read (it travels in the same record: a search's caller
// wants both - mcts/tree.py:57-105 reads the root and tree.num_nodes - and a second call would be a second host round trip)
int tg_search_node_record_num_nodes(tg_search *s, int32_t *num_nodes_host) {
    if (!s || !num_nodes_host) return tg::fail(TG_ERR_ARG, "tg_search_node_record_num_nodes: null argument");
    if (!s->node_host) return tg::fail(TG_ERR_STATE, "tg_search_node_record_num_nodes: no node has been read");
    *num_nodes_host = reinterpret_cast<const int32_t *>(s->node_host)[6];
    return TG_OK;
}

}  // extern "C"

// ======================================================================================
// Self-play shard bookkeeping (host, C++): everything selfplay/worker.py:50-90 does per move and per board
// around the search - sequential-halving schedule (mcts/sequential_halving.py:7-60), final root choice
// (node.py:324-346), resign rule (tree.py:351-354), improved policy (node.py:281-321) and its ".3e" comment
// (sgf/selfplay_record.py:45-65), two-pass end + count_score (go_board.py:561-608), the SGF file
// (selfplay_record.py:67-110) - for all T boards of a search handle in ONE call per move, on host threads.
// Double arithmetic in the reference's evaluation order (np.sum's pairwise order, libm exp).
// ======================================================================================
namespace {

double host_np_sum_block(const double *a, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}
double host_np_sum(const double *a, int n) {        // numpy's pairwise summation (blocks of <= 128)
    if (n <= 128) return host_np_sum_block(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return host_np_sum(a, n2) + host_np_sum(a + n2, n - n2);
}

struct SpGame {
    int index = -1;                 // -1: slot parked
    bool never_resign = false, done = true;
    bool fresh = false;             // started, root not expanded yet (chained moves: the slot sits out one lock-step move)
    int to_move = kBlack, pass_count = 0, moves_played = 0;
    std::string body;               // ";B[ee]C[...]" ...
};

// {considered actions: levels} in phase order (mcts/sequential_halving.py)
std::vector<std::pair<int, int>> halving_pairs(int max_considered, int sims) {
    std::vector<int> seq;
    if (max_considered <= 1) {
        for (int i = 0; i < sims; ++i) seq.push_back(i);
    } else {
        const int log2max = (int)std::ceil(std::log2((double)max_considered));
        std::vector<int> visits(max_considered, 0);
        int width = max_considered;
        while ((int)seq.size() < sims) {
            const int rounds = std::max(1, (int)((double)sims / (double)(log2max * width)));
            for (int r = 0; r < rounds; ++r)
                for (int i = 0; i < width; ++i) seq.push_back(visits[i]++);
            width = std::max(2, width / 2);
        }
        seq.resize(sims);
    }
    int mx = 0;
    for (int v : seq) mx = std::max(mx, v);
    std::vector<int> width_at_level(mx + 1, 0);
    for (int v : seq) ++width_at_level[v];
    std::vector<std::pair<int, int>> pairs;
    for (int w : width_at_level) {
        bool found = false;
        for (auto &pr : pairs)
            if (pr.first == w) { ++pr.second; found = true; break; }
        if (!found) pairs.emplace_back(w, 1);
    }
    return pairs;
}

}  // namespace

struct tg_selfplay {
    tg_search *s = nullptr;
    std::string save_dir, komi_text;
    double komi = 7.0;
    int visits = 16;
    std::vector<SpGame> games;                                   // [T]
    std::map<int, std::vector<std::pair<int, int>>> schedule_cache;   // root candidates -> phases
    // per-move scratch (root statistics of all trees)
    std::vector<int32_t> nc, nv, visits_a, vl_a;
    std::vector<float> raw;
    std::vector<double> vsum_a, pol_a;
    std::vector<int16_t> act_a;
    std::vector<uint8_t> cells;
    std::vector<int32_t> ph_nc, ph_mc, mv, fin;
    std::vector<int64_t> consumed;
    bool nc_known = false;                 // nc holds the roots' child counts of the move in progress
    std::vector<int32_t> ph_seen;          // per tree: root children entered so far in this move (upper bound)       // tg_selfplay_play_move scratch
    bool force_feed = true;                          // a stream was (re)seeded: the next random window is regenerated
    tg_selfplay_observer observer = nullptr;         // audit hook (tg_selfplay_set_observer)
    void *observer_user = nullptr;
    std::vector<int32_t> nc_cursor;                  // root child counts as read off the draw cursor (cross-checked after the move)
    std::vector<int32_t> act32;                      // root actions of the move in progress (finish_move scratch)
    double t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // TG_SP_TIMING: host wall clock per section, per HANDLE (groups run on their own threads)
    long moves_timed = 0;
    // The improved-policy comment of a move (two softmaxes + up to 82 "%.3e" fields per board: most of finish_move's
    // host time) is not needed to go on - only the chosen move is.  finish_move saves what the comment is made of and
    // the text is appended to the game record later, while the NEXT move's kernels run (tg_selfplay_play_move, before
    // it waits for them), at the latest at the start of the next finish_move.
    struct PendingComment {
        bool valid = false;
        int n = 0, pos = 0, color = 0, nv = 0, max_count = 0;
        float raw = 0.f;
        std::vector<int32_t> act, vis;
        std::vector<double> pol, vsum;
    };
    std::vector<PendingComment> pending;             // [T]
    int64_t last_window = 0;                         // phase random window of the last move (draws per tree): pre-generation hint
    // chained moves (tg_selfplay_play_move, default): the device decides the move and goes on to the next root by itself
    bool chained = false;                            // this handle runs chained moves (set by the first tg_selfplay_play_move)
    bool sync_started = false;                       // ... or the round-trip scheme (TG_SP_CHAIN=0)
    bool chain_started = false;                      // a chain has run: the roots are expanded, c_phase / the cursors are valid
    std::vector<int64_t> c_phase;                    // per tree: draw cursor behind the phases of the last move (RootTail)
    std::vector<uint8_t> skip, skip_fresh;           // per tree: takes no part in this move / game just started
    std::vector<int32_t> state;                      // finish_roots_kernel's per-tree state [T][4]
    // sub-groups of a lock-step move (launch_phases_subgroups): streams 1.., events, the phase tables of a whole move
    static constexpr int kMaxSub = 16, kMaxPhases = 16;
    int n_sub_streams = 0;
    hipStream_t sub_stream[kMaxSub - 1] = {};
    hipEvent_t ev_start = nullptr, ev_first_sel[kMaxSub] = {}, ev_sub_done[kMaxSub - 1] = {};
    int32_t *phase_all_dev = nullptr, *phase_all_pin = nullptr;      // [kMaxPhases][3 T] (pinned: a ring of two)
    hipEvent_t phase_all_ev[2] = {};
    bool phase_all_used[2] = {};
    unsigned phase_all_seq = 0;
    // A chained move between tg_selfplay_move_begin (everything queued) and tg_selfplay_move_end (records awaited, bookkeeping):
    // what the second half needs from the first.  Several handles (lanes of one shard, each with its own engine and stream) are
    // kept in this state at once by ONE host thread, so that the device always has other lanes' moves queued while the host
    // waits for one lane's records - boards of different lanes are on different moves (round 6).
    struct PendingMove {
        bool active = false, any_phase = false;
        int64_t leaves = 0;
        int32_t n_phases = 0;
        float *planes = nullptr, *policy = nullptr, *value = nullptr;
        void *stream = nullptr;
        double t_last = 0.0;
    } pend;
};

namespace {

std::string gtp_name(int pos, int S) {                           // board/coordinate.py:38-43
    if (pos == 0) return "pass";
    static const char *letters = "IABCDEFGHJKLMNOPQRSTUVWXYZ";
    const int W = S + 2, col = pos % W - 1, row = pos / W - 1;
    return std::string(1, letters[col + 1]) + std::to_string(S - row);
}
std::string sgf_name(int pos, int S) {                           // board/coordinate.py:45-49
    if (pos == 0) return "tt";
    const int W = S + 2, col = pos % W - 1, row = pos / W - 1;
    return std::string(1, (char)('a' + col)) + std::string(1, (char)('a' + row));
}

// go_board.py:561-608: stones in atari count as dead; an empty point takes the colour of its direct
// neighbours only (both colours -> neutral), in row-major order, earlier colourings feeding later points
int count_score_cells(const uint8_t *cells, int S) {
    const int W = S + 2, NC = W * W;
    std::vector<uint8_t> work(cells, cells + NC);
    std::vector<int> stack;
    std::vector<uint8_t> seen(NC), lib_seen(NC);
    for (int y = 1; y <= S; ++y)
        for (int x = 1; x <= S; ++x) {
            const int pos = y * W + x, c = cells[pos];
            if (c != kBlack && c != kWhite) continue;
            std::fill(seen.begin(), seen.end(), 0);
            std::fill(lib_seen.begin(), lib_seen.end(), 0);
            int libs = 0;
            stack.assign(1, pos);
            seen[pos] = 1;
            while (!stack.empty() && libs < 2) {
                const int p = stack.back();
                stack.pop_back();
                const int nb[4] = {p - W, p - 1, p + 1, p + W};
                for (int n : nb) {
                    if (cells[n] == kEmpty) { if (!lib_seen[n]) { lib_seen[n] = 1; ++libs; } }
                    else if (cells[n] == c && !seen[n]) { seen[n] = 1; stack.push_back(n); }
                }
            }
            if (libs == 1) work[pos] = kEmpty;
        }
    for (int y = 1; y <= S; ++y)
        for (int x = 1; x <= S; ++x) {
            const int pos = y * W + x;
            if (work[pos] != kEmpty) continue;
            int color = kEmpty;
            const int nb[4] = {pos - W, pos - 1, pos + 1, pos + W};
            for (int n : nb) {
                const int v = work[n];
                if (v == kBlack || v == kWhite) {
                    if (color == kEmpty) color = v;
                    else if (color != v) color = kOob;
                }
            }
            work[pos] = (uint8_t)color;
        }
    int score = 0;
    for (int i = 0; i < NC; ++i) score += (work[i] == kBlack) - (work[i] == kWhite);
    return score;
}

int write_sgf(const tg_selfplay *sp, const SpGame &g, int winner, bool is_resign, double score) {
    char buf[96];
    std::string text = "(;FF[4]GM[1]SZ[" + std::to_string(sp->s->S) + "]\n";
    text += "AP[TamaGo]PB[TamaGo-Black]PW[TamaGo-White]";
    if (winner == kBlack) {
        if (is_resign) text += "RE[B+R]";
        else { snprintf(buf, sizeof(buf), "RE[B+%.1f]", score); text += buf; }
    } else if (winner == kWhite) {
        if (is_resign) text += "RE[W+R]";
        else { snprintf(buf, sizeof(buf), "RE[W+%.1f]", -score); text += buf; }
    } else {
        text += "RE[0]";
    }
    text += "KM[" + sp->komi_text + "]";
    text += g.body;
    text += "\n)";
    const std::string path = sp->save_dir + "/" + std::to_string(g.index) + ".sgf";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return tg::fail(TG_ERR_ARG, "tg_selfplay: cannot write %s", path.c_str());
    const bool ok = fwrite(text.data(), 1, text.size(), f) == text.size();
    fclose(f);
    return ok ? TG_OK : tg::fail(TG_ERR_ARG, "tg_selfplay: short write to %s", path.c_str());
}

// improved policy (node.py:281-321) -> ";B[ee]C[<n> <gtp>:<p:.3e> ...]" appended to the game record
void append_comment(tg_selfplay *sp, SpGame &g, const tg_selfplay::PendingComment &pc) {
    const int n = pc.n, S = sp->s->S;
    const double sigma_sel = (double)(50 + pc.max_count) * 1.0;
    std::vector<double> q(n), w1(n), w2(n);
    const double *pol = pc.pol.data(), *vsum = pc.vsum.data();
    const int32_t *vis = pc.vis.data(), *act = pc.act.data();
    for (int i = 0; i < n; ++i) q[i] = vis[i] > 0 ? vsum[i] / (double)vis[i] : 0.0;
    double mx = -INFINITY;
    for (int i = 0; i < n; ++i) mx = pol[i] > mx ? pol[i] : mx;
    for (int i = 0; i < n; ++i) w1[i] = std::exp(pol[i] - mx);
    const double s1 = host_np_sum(w1.data(), n);
    for (int i = 0; i < n; ++i) { w1[i] = w1[i] / s1; w2[i] = w1[i] * q[i]; }
    const double sum_prob = host_np_sum(w1.data(), n), v_pi = host_np_sum(w2.data(), n);
    const double nvd = (double)pc.nv;
    const double mixed = ((double)pc.raw * 1.0 + nvd * v_pi / sum_prob) / (nvd + 1.0);
    // np.max(self.children_visits) runs over the whole array (zeros beyond n): same value
    double mx2 = -INFINITY;
    for (int i = 0; i < n; ++i) {
        w2[i] = pol[i] + sigma_sel * (vis[i] > 0 ? q[i] : mixed);
        mx2 = w2[i] > mx2 ? w2[i] : mx2;
    }
    for (int i = 0; i < n; ++i) w1[i] = std::exp(w2[i] - mx2);
    const double s2 = host_np_sum(w1.data(), n);
    char buf[64];
    g.body += pc.color == kBlack ? ";B[" : ";W[";
    g.body += sgf_name(pc.pos, S);
    g.body += "]C[";
    g.body += std::to_string(n);
    for (int i = 0; i < n; ++i) {
        snprintf(buf, sizeof(buf), ":%.3e", w1[i] / s2);
        g.body += ' ';
        g.body += gtp_name(act[i], S);
        g.body += buf;
    }
    g.body += ']';
}

void flush_comments(tg_selfplay *sp) {
    const int T = sp->s->dev.T;
    if ((int)sp->pending.size() != T) return;
    bool any = false;
    for (int t = 0; t < T; ++t) any |= sp->pending[t].valid;
    if (!any) return;
    parallel_trees(T, [&](int t) {
        tg_selfplay::PendingComment &pc = sp->pending[t];
        if (!pc.valid) return;
        append_comment(sp, sp->games[t], pc);
        pc.valid = false;
    });
}

}  // namespace

extern "C" {

int tg_selfplay_create(tg_search *s, const char *save_dir, int visits, double komi, const char *komi_text,
                       tg_selfplay **out) {
    if (!s || !save_dir || !komi_text || !out) return tg::fail(TG_ERR_ARG, "tg_selfplay_create: null argument");
    if (visits < 1) return tg::fail(TG_ERR_ARG, "tg_selfplay_create: visits must be >= 1");
    tg_selfplay *sp = new tg_selfplay;
    sp->s = s;
    sp->save_dir = save_dir;
    sp->komi = komi;
    sp->komi_text = komi_text;
    sp->visits = visits;
    sp->games.assign(s->dev.T, SpGame{});
    *out = sp;
    return TG_OK;
}

int tg_selfplay_destroy(tg_selfplay *sp) {
    if (!sp) return TG_OK;
    if (sp->ev_start) {
        (void)hipSetDevice(sp->s->cfg.device);
        if (sp->s->stream_known) (void)hipStreamSynchronize(sp->s->last_stream);
        for (int i = 0; i < sp->n_sub_streams; ++i) { (void)hipStreamSynchronize(sp->sub_stream[i]); (void)hipStreamDestroy(sp->sub_stream[i]); }
        (void)hipEventDestroy(sp->ev_start);
        for (hipEvent_t e : sp->ev_first_sel) (void)hipEventDestroy(e);
        for (hipEvent_t e : sp->ev_sub_done) (void)hipEventDestroy(e);
        for (hipEvent_t e : sp->phase_all_ev) (void)hipEventDestroy(e);
        (void)hipFree(sp->phase_all_dev);
        (void)hipHostFree(sp->phase_all_pin);
    }
    delete sp;
    return TG_OK;
}

int tg_selfplay_start_game(tg_selfplay *sp, int slot, int index, int never_resign) {
    if (!sp) return tg::fail(TG_ERR_ARG, "tg_selfplay_start_game: null argument");
    if (slot < 0 || slot >= sp->s->dev.T) return tg::fail(TG_ERR_ARG, "tg_selfplay_start_game: slot %d out of range", slot);
    SpGame g;
    g.index = index;
    g.never_resign = never_resign != 0;
    g.done = index < 0;
    g.fresh = index >= 0;
    sp->games[slot] = g;
    if ((size_t)slot < sp->pending.size()) sp->pending[slot].valid = false;
    sp->force_feed = true;
    return TG_OK;
}

int tg_selfplay_schedule(tg_selfplay *sp, int32_t *num_considered_host, int32_t *max_count_host, int max_phases,
                         int32_t *n_phases_host) {
    if (!sp || !num_considered_host || !max_count_host || !n_phases_host)
        return tg::fail(TG_ERR_ARG, "tg_selfplay_schedule: null argument");
    tg_search *s = sp->s;
    const int T = s->dev.T;
    if (!sp->nc_known) {
        sp->nc.resize(T);
        int rc = tg_search_read_root_stats(s, sp->nc.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc) return rc;
    }
    sp->nc_known = false;
    std::fill(num_considered_host, num_considered_host + (size_t)max_phases * T, 0);
    std::fill(max_count_host, max_count_host + (size_t)max_phases * T, 0);
    int n_phases = 0;
    for (int t = 0; t < T; ++t) {
        if (sp->games[t].done || (sp->chained && sp->games[t].fresh)) continue;
        const int base = sp->nc[t] < 16 ? sp->nc[t] : 16;            // MAX_CONSIDERED_NODES (mcts/constant.py)
        auto it = sp->schedule_cache.find(base);
        if (it == sp->schedule_cache.end())
            it = sp->schedule_cache.emplace(base, halving_pairs(base, sp->visits)).first;
        const auto &pairs = it->second;
        if ((int)pairs.size() > max_phases) return tg::fail(TG_ERR_ARG, "tg_selfplay_schedule: %zu phases exceed max_phases", pairs.size());
        for (size_t ph = 0; ph < pairs.size(); ++ph) {
            num_considered_host[ph * T + t] = pairs[ph].first;
            max_count_host[ph * T + t] = pairs[ph].second;
        }
        n_phases = std::max(n_phases, (int)pairs.size());
    }
    *n_phases_host = n_phases;
    return TG_OK;
}

static int finish_move_impl(tg_selfplay *sp, int32_t *moves_host, int32_t *finished_host, int64_t *stats_host, const RootTail *tail);
int tg_selfplay_finish_move(tg_selfplay *sp, int32_t *moves_host, int32_t *finished_host, int64_t *stats_host) {
    return finish_move_impl(sp, moves_host, finished_host, stats_host, nullptr);
}

// tail != nullptr (chained moves): the root records are in roots_host already (launch_finish_roots, fin_ev waited for) and
// the device has decided the moves itself - the host's own decision below, from the same statistics, must be the same
static int finish_move_impl(tg_selfplay *sp, int32_t *moves_host, int32_t *finished_host, int64_t *stats_host, const RootTail *tail) {
    if (!sp || !moves_host || !finished_host) return tg::fail(TG_ERR_ARG, "tg_selfplay_finish_move: null argument");
    tg_search *s = sp->s;
    const SearchDev &D = s->dev;
    const int T = D.T, A = s->A, S = s->S;
    flush_comments(sp);                                               // the previous move's comments, if still waiting
    sp->pending.resize(T);
    sp->nc.resize(T); sp->nv.resize(T); sp->raw.resize(T);
    sp->visits_a.resize((size_t)T * A); sp->vl_a.resize((size_t)T * A);
    sp->vsum_a.resize((size_t)T * A); sp->pol_a.resize((size_t)T * A);
    std::vector<int32_t> &act32 = sp->act32;
    act32.resize((size_t)T * A);
    int rc = TG_OK;
    if (tail) {
        if ((rc = root_record_errors(s))) return rc;
        parse_root_records(s, sp->nc.data(), sp->nv.data(), sp->raw.data(), act32.data(), sp->visits_a.data(), sp->vl_a.data(),
                           sp->vsum_a.data(), sp->pol_a.data());
    } else {
        rc = tg_search_read_root_stats(s, sp->nc.data(), sp->nv.data(), sp->raw.data(), act32.data(), sp->visits_a.data(),
                                       sp->vl_a.data(), sp->vsum_a.data(), sp->pol_a.data());
    }
    if (rc) return rc;
    if (int nrc = noise_host_sync(s)) return nrc;                    // (the device-drawn noise has long arrived)
    if (s->noise_host.size() != (size_t)T * A) return tg::fail(TG_ERR_STATE, "tg_selfplay_finish_move: no root noise was set");
    // the boards are only needed to score a game that ends with this move, i.e. with a second pass in a row
    bool may_end = false;
    for (int t = 0; t < T; ++t) may_end |= !sp->games[t].done && sp->games[t].pass_count == 1 && (!tail || tail[t].kind == 2);
    sp->cells.resize((size_t)T * s->NC);
    if (may_end) TG_HIP(hipMemcpy(sp->cells.data(), D.root_cells, sp->cells.size(), hipMemcpyDeviceToHost));
    const int max_moves = S * S * 2;                                  // worker.py:44
    std::vector<int> status(T, TG_OK);
    std::vector<int64_t> n_moves(T, 0), n_games(T, 0);
    parallel_trees(T, [&](int t) {
        SpGame &g = sp->games[t];
        moves_host[t] = -1;
        finished_host[t] = 0;
        if (g.done || (tail && g.fresh)) return;
        const size_t o = (size_t)t * A;
        const int n = sp->nc[t];
        const int32_t *vis = &sp->visits_a[o], *vl = &sp->vl_a[o], *act = &act32[o];
        const double *vsum = &sp->vsum_a[o], *pol = &sp->pol_a[o], *noise = &s->noise_host[o];
        // ---- final root choice (tree.py:344, node.py:324-346 with count_threshold = PLAYOUTS = 100) ----
        int max_count = 0;
        for (int i = 0; i < n; ++i) max_count = std::max(max_count, vis[i]);
        const double sigma_sel = (double)(50 + max_count) * 1.0;
        int best = 0;
        double best_v = 0.0;
        for (int i = 0; i < n; ++i) {
            const double qi = vis[i] > 0 ? vsum[i] / (double)vis[i] : 0.0;
            const double ev = (vis[i] + vl[i] >= 100) ? -10000.0 : (pol[i] + noise[i]) + sigma_sel * qi;
            if (i == 0 || ev > best_v) { best_v = ev; best = i; }
        }
        const double value = vis[best] == 0 ? 0.5 : vsum[best] / (double)vis[best];
        n_moves[t] = 1;
        if (tail) {
            // what the device did with the same numbers (finish_roots_kernel): anything else is a bug, not a rounding matter
            const int p2 = act[best] == 0 ? g.pass_count + 1 : 0;
            const int kind = (!g.never_resign && value < 0.05) ? 1 : p2 == 2 ? 2 : g.moves_played + 1 >= max_moves ? 3 : 0;
            if (tail[t].best != best || tail[t].kind != kind || tail[t].pos != act[best]) {
                status[t] = tg::fail(TG_ERR_STATE, "tg_selfplay_play_move: board %d: the device chose child %d (point %d, kind %d), "
                                     "the host child %d (point %d, kind %d) from the same root statistics", t, tail[t].best,
                                     tail[t].pos, tail[t].kind, best, act[best], kind);
                return;
            }
        }
        if (!g.never_resign && value < 0.05) {                          // worker.py:59-62
            status[t] = write_sgf(sp, g, 3 - g.to_move, true, 0.0);
            g.done = true;
            finished_host[t] = 1;
            n_games[t] = 1;
            return;
        }
        const int pos = act[best];
        // ---- improved policy (node.py:281-321) -> comment "<n> <gtp>:<p:.3e> ...": saved, formatted later ----
        tg_selfplay::PendingComment &pc = sp->pending[t];
        pc.valid = true;
        pc.n = n; pc.pos = pos; pc.color = g.to_move; pc.nv = sp->nv[t]; pc.raw = sp->raw[t]; pc.max_count = max_count;
        pc.act.assign(act, act + n);
        pc.vis.assign(vis, vis + n);
        pc.pol.assign(pol, pol + n);
        pc.vsum.assign(vsum, vsum + n);
        moves_host[t] = pos;
        g.pass_count = pos == 0 ? g.pass_count + 1 : 0;
        g.to_move = 3 - g.to_move;
        g.moves_played += 1;
        if (g.pass_count == 2 || g.moves_played >= max_moves) {         // the game ends here: its record is written now
            append_comment(sp, g, pc);
            pc.valid = false;
        }
        if (g.pass_count == 2) {                                        // worker.py:80-87 (a pass leaves the cells as they are)
            const double score = (double)count_score_cells(&sp->cells[(size_t)t * s->NC], S) - sp->komi;
            const int winner = score > 0.1 ? kBlack : (score < -0.1 ? kWhite : kOob);
            status[t] = write_sgf(sp, g, winner, false, score);
            g.done = true;
        } else if (g.moves_played >= max_moves) {
            status[t] = write_sgf(sp, g, kEmpty, false, 0.0);
            g.done = true;
        }
        if (g.done) {
            finished_host[t] = 1;
            n_games[t] = 1;
            moves_host[t] = -1;                                         // the slot gets a new root anyway
        }
    });
    for (int t = 0; t < T; ++t)
        if (status[t] != TG_OK) return status[t];
    if (stats_host) {
        stats_host[0] = 0;
        stats_host[1] = 0;
        for (int t = 0; t < T; ++t) { stats_host[0] += n_games[t]; stats_host[1] += n_moves[t]; }
    }
    return TG_OK;
}

int tg_selfplay_set_observer(tg_selfplay *sp, tg_selfplay_observer fn, void *user) {
    if (!sp) return tg::fail(TG_ERR_ARG, "tg_selfplay_set_observer: null argument");
    sp->observer = fn;
    sp->observer_user = user;
    return TG_OK;
}

// One whole self-play move of every board, driven from here (no host-language code between the launches):
// root expansion + evaluation, Gumbel noise, halving schedule, every phase (selection, forward pass of the
// library's own network handle, backup), final choice / records / finished games, the moves played on the
// device-resident boards.  Buffers are the caller's (device): planes [T * batch_size][6][S][S],
// policy [T * batch_size][A], value [T * batch_size][3].
static int play_move_sync(tg_selfplay *sp, tg_net *net, float *planes_dev, float *policy_dev, float *value_dev,
                          void *stream, int32_t *finished_host, int64_t *stats_host);
static int play_move_chain(tg_selfplay *sp, tg_net *net, float *planes_dev, float *policy_dev, float *value_dev,
                           void *stream, int32_t *finished_host, int64_t *stats_host);
static int chain_begin(tg_selfplay *sp, tg_net *net, float *planes_dev, float *policy_dev, float *value_dev, void *stream);
static int chain_end(tg_selfplay *sp, int32_t *finished_host, int64_t *stats_host);

// PUCT mini-batches queued back to back (mcts/tree.py:146-152 with process_mini_batch, :273-315): ONE random window for all of them
// - uploaded in two parts, the first mini-batch's share in front of its selection launch, the rest behind it -, then per mini-batch
// selection, forward pass, backup on `stream`, with no host round trip in between.  The same launches on the same data as the
// per-mini-batch calls; for searches whose course does not depend on a mini-batch's outcome (STRICT_PLAYOUT, time_manager.py:160-161).
// The caller reads the cursors back afterwards (tg_search_advance_streams), as after a single mini-batch.
int tg_search_puct_chain(tg_search *s, tg_net *net, const int32_t *leaves_host, int n_batches, int force_window,
                         float *planes_dev, float *policy_dev, float *value_dev, void *stream) {
    if (!s || !net || !leaves_host || !planes_dev || !policy_dev || !value_dev)
        return tg::fail(TG_ERR_ARG, "tg_search_puct_chain: null argument");
    if (n_batches < 1) return tg::fail(TG_ERR_ARG, "tg_search_puct_chain: no mini-batch");
    if (tg_net_board_size(net) != s->S) return tg::fail(TG_ERR_ARG, "tg_search_puct_chain: network and search differ in board size");
    size_t total = 0;
    for (int b = 0; b < n_batches; ++b) {
        if (leaves_host[b] < 1 || leaves_host[b] > s->dev.K)
            return tg::fail(TG_ERR_ARG, "tg_search_puct_chain: mini-batch %d of %d leaves outside [1, batch_size]", b, leaves_host[b]);
        total += (size_t)leaves_host[b];
    }
    const size_t A = (size_t)s->A;
    // The window goes up one mini-batch's share at a time, each generated (MT19937 + log on this thread: ~90 us per 21 k draws)
    // and uploaded while the launches of the mini-batch before it run (select + forward + backup: 0.3 - 0.45 ms): a search that
    // starts from a freshly seeded stream - search_best_move hands numpy's state over per move - has nothing staged.
    int rc = feed_streams_impl(s, total * A, force_window, n_batches > 1 ? (size_t)leaves_host[0] * A : 0);
    if (rc) return rc;
    const bool pieces = s->rng_rest_cols > 0;                            // (a new window was started and split)
    hipStream_t st = static_cast<hipStream_t>(stream);
    size_t upto = 0;
    for (int b = 0; b < n_batches; ++b) {
        const int k = leaves_host[b];
        upto += (size_t)k * A;
        if (b > 0 && pieces && (rc = feed_streams_part(s, upto, st))) return rc;
        if ((rc = tg_search_select_puct(s, k, planes_dev, nullptr, stream))) return rc;
        if ((rc = tg_net_forward_dev(net, planes_dev, s->dev.T * k, 0, policy_dev, value_dev, stream))) return rc;
        if ((rc = tg_search_backup(s, policy_dev, value_dev, k, 0, stream))) return rc;
    }
    return feed_streams_rest(s);                                          // (nothing left unless the window was an older, larger one)
}

int tg_selfplay_play_move(tg_selfplay *sp, tg_net *net, float *planes_dev, float *policy_dev, float *value_dev,
                          void *stream, int32_t *finished_host, int64_t *stats_host) {
    if (!sp || !net || !planes_dev || !policy_dev || !value_dev || !finished_host)
        return tg::fail(TG_ERR_ARG, "tg_selfplay_play_move: null argument");
    // TG_SP_CHAIN=0: the move decided on the host, three host round trips per move (kept for comparison; a handle stays
    // with the scheme of its first move)
    if (!sp->chain_started && !sp->sync_started) sp->chained = !tg::knob("TG_SP_CHAIN") || atoi(tg::knob("TG_SP_CHAIN")) != 0;
    if (sp->chained) return play_move_chain(sp, net, planes_dev, policy_dev, value_dev, stream, finished_host, stats_host);
    sp->sync_started = true;
    return play_move_sync(sp, net, planes_dev, policy_dev, value_dev, stream, finished_host, stats_host);
}

// The two halves of a chained tg_selfplay_play_move (see PendingMove): `begin` queues the whole move on `stream` and returns with
// the device working; `end` waits for that move's records, does the bookkeeping and reports finished games.  Between the two
// the caller may run other handles' halves; slots are refilled (tg_selfplay_start_game) between an `end` and the next `begin`.
int tg_selfplay_move_begin(tg_selfplay *sp, tg_net *net, float *planes_dev, float *policy_dev, float *value_dev, void *stream) {
    if (!sp || !net || !planes_dev || !policy_dev || !value_dev) return tg::fail(TG_ERR_ARG, "tg_selfplay_move_begin: null argument");
    if (sp->sync_started || sp->observer)
        return tg::fail(TG_ERR_STATE, "tg_selfplay_move_begin: this handle runs the round-trip scheme / has an observer (whole moves only)");
    if (sp->pend.active) return tg::fail(TG_ERR_STATE, "tg_selfplay_move_begin: the previous move has not been ended");
    sp->chained = true;
    return chain_begin(sp, net, planes_dev, policy_dev, value_dev, stream);
}

int tg_selfplay_move_end(tg_selfplay *sp, int32_t *finished_host, int64_t *stats_host) {
    if (!sp || !finished_host) return tg::fail(TG_ERR_ARG, "tg_selfplay_move_end: null argument");
    if (!sp->pend.active) return tg::fail(TG_ERR_STATE, "tg_selfplay_move_end: no move has been begun");
    return chain_end(sp, finished_host, stats_host);
}

// Chained moves.  A call = one lock-step move of every board whose root is expanded:
//   host   cursors of the last chain -> root child counts; Gumbel noise; halving schedule; random window (phases + next root)
//   device the phases (selection, forward, backup); finish_roots_kernel: root records + THE MOVE (choice, resign, game end);
//          play_kernel on the device's own moves; next root expansion, forward, backup          <- no host in between
//   host   (while that runs) last move's record comments, draws generated ahead; then waits for the RECORDS only (fin_ev,
//          recorded ahead of play_kernel), does the bookkeeping - checking the device's decision against its own - and returns
//          with the device still ~0.15 ms from the end of the root evaluation the next call starts from.
// A slot whose game has just been started (tg_selfplay_start_game) sits out one call: its root is expanded by that call's
// chain.  The first call of a handle therefore evaluates roots only.  Games, records and draws are those of the
// round-trip scheme (the draws of a game are consumed in the same order: root prior, noise, phases, next root prior ...).
// The phases of a move with the boards in G sub-groups, each on a stream of its own and one selection behind the
// previous one: a sub-group's forward pass (most of the CUs) runs while the others' tree kernels (one workgroup per
// board) do - with ONE group the forward pass waits for the slowest board's selection and the selection for the whole
// forward pass, every phase.  Same kernels, on slices of the engine (sub_dev); leaf slots: sub-group g owns positions
// [first board x batch_size ...) of the caller's buffers, so sub-groups in different phases never share rows.
static int launch_phases_subgroups(tg_selfplay *sp, tg_net *net, int n_phases, int G, float *planes_dev, float *policy_dev,
                                   float *value_dev, hipStream_t st, int64_t &leaves, bool &any_phase) {
    tg_search *s = sp->s;
    const int T = s->dev.T, A = s->A, K = s->dev.K;
    const size_t P = (size_t)s->P;
    constexpr int kMaxPhases = tg_selfplay::kMaxPhases;
    int rc;
    if (!sp->ev_start) {
        TG_HIP(hipEventCreateWithFlags(&sp->ev_start, hipEventDisableTiming));
        for (hipEvent_t &e : sp->ev_first_sel) TG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t &e : sp->ev_sub_done) TG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t &e : sp->phase_all_ev) TG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        TG_HIP(hipMalloc(reinterpret_cast<void **>(&sp->phase_all_dev), (size_t)kMaxPhases * 3 * T * sizeof(int32_t)));
        TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&sp->phase_all_pin), (size_t)2 * kMaxPhases * 3 * T * sizeof(int32_t), hipHostMallocDefault));
    }
    while (sp->n_sub_streams < G - 1) {
        TG_HIP(hipStreamCreateWithFlags(&sp->sub_stream[sp->n_sub_streams], hipStreamNonBlocking));
        sp->n_sub_streams += 1;
    }
    int tb[tg_selfplay::kMaxSub + 1];
    for (int g = 0; g <= G; ++g) tb[g] = (int)((int64_t)g * T / G);
    const int ring = (int)(sp->phase_all_seq++ % 2);
    if (sp->phase_all_used[ring]) TG_HIP(hipEventSynchronize(sp->phase_all_ev[ring]));
    int32_t *tab = sp->phase_all_pin + (size_t)ring * kMaxPhases * 3 * T;
    int64_t counts[kMaxPhases][tg_selfplay::kMaxSub] = {};
    int32_t most[kMaxPhases][tg_selfplay::kMaxSub] = {};               // the most descents a tree of the sub-group makes in the phase
    for (int ph = 0; ph < n_phases; ++ph) {
        const int32_t *nc = &sp->ph_nc[(size_t)ph * T], *mc = &sp->ph_mc[(size_t)ph * T];
        int32_t *row = tab + (size_t)ph * 3 * T;
        for (int g = 0; g < G; ++g) {
            int64_t at = (int64_t)tb[g] * K;
            for (int t = tb[g]; t < tb[g + 1]; ++t) {
                const int64_t n = (int64_t)nc[t] * mc[t];
                if (nc[t] < 0 || mc[t] < 0 || n > K)
                    return tg::fail(TG_ERR_ARG, "tg_selfplay_play_move: tree %d phase does not fit %d slots", t, K);
                row[t] = nc[t];
                row[T + t] = mc[t];
                row[2 * (size_t)T + t] = (int32_t)at;
                at += n;
                most[ph][g] = n > most[ph][g] ? (int32_t)n : most[ph][g];
            }
            counts[ph][g] = at - (int64_t)tb[g] * K;
        }
    }
    TG_HIP(hipMemcpyAsync(sp->phase_all_dev, tab, (size_t)n_phases * 3 * T * sizeof(int32_t), hipMemcpyHostToDevice, st));
    TG_HIP(hipEventRecord(sp->phase_all_ev[ring], st));
    sp->phase_all_used[ring] = true;
    s->last_stream = st;
    s->stream_known = true;
    if ((rc = install_rng(s, st))) return rc;                          // (the first part of the window; the cursors back to 0)
    s->packed_leaves = true;
    TG_HIP(hipEventRecord(sp->ev_start, st));
    for (int g = 1; g < G; ++g) TG_HIP(hipStreamWaitEvent(sp->sub_stream[g - 1], sp->ev_start, 0));
    int launched[tg_selfplay::kMaxSub] = {};
    int last_started = -1;                                              // the last sub-group whose first selection is queued
    for (int ph = 0; ph < n_phases; ++ph) {
        const int32_t *row = sp->phase_all_dev + (size_t)ph * 3 * T;
        for (int g = 0; g < G; ++g) {
            const int64_t count = counts[ph][g];
            if (count == 0) continue;
            hipStream_t sg = g == 0 ? st : sp->sub_stream[g - 1];
            static const bool stagger = tg::knob("TG_SP_STAGGER") && atoi(tg::knob("TG_SP_STAGGER")) != 0;   // (measured: no gain at 16 boards, -3 % at 24 - the streams fall out of step by themselves)
            if (stagger && launched[g] == 0 && last_started >= 0) TG_HIP(hipStreamWaitEvent(sg, sp->ev_first_sel[last_started], 0));
            if (launched[g] == 1 && any_phase) TG_HIP(hipStreamWaitEvent(sg, s->ev_rng[s->rng_active], 0));   // second part of the window
            const SearchDev D = sub_dev(s, tb[g], tb[g + 1] - tb[g]);
            const int32_t *off = row + 2 * (size_t)T + tb[g];
            if ((rc = launch_gumbel_select(s, D, row + tb[g], row + T + tb[g], off, K, most[ph][g], planes_dev, sg))) return rc;
            if (launched[g] == 0) {
                TG_HIP(hipEventRecord(sp->ev_first_sel[g], sg));
                last_started = g;
            }
            const size_t o = (size_t)tb[g] * K;
            if ((rc = tg_net_forward_dev(net, planes_dev + o * 6 * P, (int)count, 1, policy_dev + o * A, value_dev + o * 3, sg))) return rc;
            if ((rc = launch_backup(s, D, policy_dev, value_dev, 0, off, 1, sg))) return rc;
            leaves += count;
            launched[g] += 1;
            if (!any_phase) {
                any_phase = true;
                if ((rc = feed_streams_rest(s))) return rc;           // behind the first launched selection
            }
        }
    }
    for (int g = 1; g < G; ++g) {
        TG_HIP(hipEventRecord(sp->ev_sub_done[g - 1], sp->sub_stream[g - 1]));
        TG_HIP(hipStreamWaitEvent(st, sp->ev_sub_done[g - 1], 0));
    }
    return TG_OK;
}

static int play_move_chain(tg_selfplay *sp, tg_net *net, float *planes_dev, float *policy_dev, float *value_dev,
                           void *stream, int32_t *finished_host, int64_t *stats_host) {
    if (sp->pend.active) return tg::fail(TG_ERR_STATE, "tg_selfplay_play_move: a move begun with tg_selfplay_move_begin has not been ended");
    if (int rc = chain_begin(sp, net, planes_dev, policy_dev, value_dev, stream)) return rc;
    return chain_end(sp, finished_host, stats_host);
}

static inline double sp_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int chain_begin(tg_selfplay *sp, tg_net *net, float *planes_dev, float *policy_dev, float *value_dev, void *stream) {
    tg_search *s = sp->s;
    const int T = s->dev.T, A = s->A;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc;
    s->prefill_enabled = false;
    static const bool timing = getenv("TG_SP_TIMING") != nullptr;
    double *acc = sp->t_acc;
    double t_last = timing ? sp_now() : 0.0;
    auto lap = [&](int i) { if (timing) { const double t = sp_now(); acc[i] += t - t_last; t_last = t; } };
    if (s->streams.size() != (size_t)T) return tg::fail(TG_ERR_ARG, "tg_selfplay_play_move: streams are not seeded");
    sp->skip.assign(T, 0);
    sp->skip_fresh.assign(T, 0);
    int64_t leaves = 0;
    for (int t = 0; t < T; ++t) {
        const SpGame &g = sp->games[t];
        sp->skip_fresh[t] = (!sp->chain_started || g.fresh) ? 1 : 0;          // (before the first chain no root is expanded)
        sp->skip[t] = (g.done || sp->skip_fresh[t]) ? 1 : 0;
        leaves += sp->skip[t] ? 0 : 1;                                        // the root evaluation this move starts from
    }
    // ---- the roots' child counts, off the draw cursors (a Dirichlet prior takes one draw per child) ----
    sp->consumed.assign(T, 0);
    sp->nc.assign(T, 0);
    sp->c_phase.resize(T, 0);
    if (sp->chain_started) {
        TG_HIP(hipEventSynchronize(s->cur_ev));
        if ((rc = advance_streams_impl(s, sp->consumed.data(), sp->skip_fresh.data(), s->cur_pin))) return rc;
        for (int t = 0; t < T; ++t) {
            if (sp->skip[t]) continue;
            const int64_t n = sp->consumed[t] - sp->c_phase[t];
            if (n < 1 || n > A)
                return tg::fail(TG_ERR_STATE, "tg_selfplay_play_move: board %d consumed %lld draws at its root (expected 1..%d)",
                                t, (long long)n, A);
            sp->nc[t] = (int32_t)n;
        }
    }
    sp->nc_known = true;
    sp->nc_cursor = sp->nc;
    lap(0);
    // (generated on the device, legacy_rng_device.h; the readers of the last noise - phases, finish_roots_kernel - lie before cur_ev)
    if ((rc = draw_noise_impl(s, nullptr, sp->skip_fresh.data(), sp->chain_started ? s->cur_ev : nullptr))) return rc;
    lap(1);
    // ---- sequential halving (tree.py:375-384) ----
    constexpr int kMaxPhases = tg_selfplay::kMaxPhases;
    sp->ph_nc.resize((size_t)kMaxPhases * T);
    sp->ph_mc.resize((size_t)kMaxPhases * T);
    int32_t n_phases = 0;
    if ((rc = tg_selfplay_schedule(sp, sp->ph_nc.data(), sp->ph_mc.data(), kMaxPhases, &n_phases))) return rc;
    // one window for the phases (bound as in the round-trip scheme: DESIGN 4.2) and the next root's prior (<= A draws)
    sp->ph_seen.assign(T, 0);
    int64_t window = 0, first_window = 0;
    for (int ph = 0; ph < n_phases; ++ph) {
        const int32_t *nc = &sp->ph_nc[(size_t)ph * T], *mc = &sp->ph_mc[(size_t)ph * T];
        int64_t expansions = 0;
        for (int t = 0; t < T; ++t) {
            const int64_t n = (int64_t)nc[t] * mc[t];
            const int64_t entered = std::min<int64_t>(n, std::min<int64_t>(A, (int64_t)nc[t] + sp->ph_seen[t]));
            expansions = entered > expansions ? entered : expansions;
            sp->ph_seen[t] = (int32_t)std::min<int64_t>(A, sp->ph_seen[t] + entered);
        }
        window += expansions * A;
        if (ph == 0) first_window = window;
    }
    lap(2);
    if ((rc = feed_streams_impl(s, (size_t)(window + A), 1, window > 0 ? (size_t)first_window : 0))) return rc;
    lap(3);
    bool any_phase = false;
    // sub-groups (TG_SP_SUBGROUPS overrides; 1 = the whole lock-step group at once; an observer sees whole phases)
    const int sub_env = tg::knob("TG_SP_SUBGROUPS") ? atoi(tg::knob("TG_SP_SUBGROUPS")) : 0;       // (read per call: tests toggle it)
    // measured (tools/bench_selfplay.py, 400 simulations, leaf evaluations/s, one group -> sub-groups): 4 boards 1.06 -> 1.20 M
    // (2), 8: 1.71 -> 1.86 M (2), 16: 2.49 -> 2.84 M (3), 24: 3.00 -> 3.30 M (4); from 32 boards on a sub-group's forward pass
    // needs every CU or comes in launches too small to be efficient (32 boards: 3.26 M whole, 2.95 M in six)
    // More boards: two halves (29 .. 96 boards) or four quarters (.. 224), the forward launches kept off as many CUs as the
    // OTHER sub-groups' tree kernels need (a forward workgroup holds its CU for the whole launch - without the cap a
    // selection waits for it to end, with 8 CUs too few the forward pass waits for the selection): 32 boards 3.33 -> 3.57 M,
    // 48: 3.82 -> 4.16 M, 64: 4.09 -> 4.50 M, 96: 4.43 -> 4.72 M, 128: 4.62 -> 4.87 M, 192: 4.83 -> 4.97 M; 256: no gain.
    int G = 1, fwd_cap = 0;
    // (re-measured when the selection launch got shorter - the workers take whole entries, round 6 -: three sub-groups beat four
    // from 13 boards on - 20 boards 3.50 -> 4.01 M, 24: 3.95 -> 4.11 M, 28: 4.22 -> 4.41 M - and a forward cap pays from 24 boards:
    // 24: 4.17 -> 4.25 M, 28: 4.46 -> 4.62 M; 8 / 12 boards: two sub-groups as before)
    if (T >= 4 && T <= 12) G = 2;
    else if (T > 12 && T <= 28) { G = 3; if (T >= 24) fwd_cap = s->num_cus - 32; }
    else if (T > 28 && T <= 96) { G = 2; fwd_cap = s->num_cus - std::max(32, T / 2); }
    else if (T > 96 && T <= 224) { G = 4; fwd_cap = s->num_cus - 32; }
    else if (T > 224 && T <= 384) { G = 2; fwd_cap = s->num_cus - 32; }     // (256 boards, one-axis forward kernel: 6.07 -> 6.30 M; 512: level)
    // (19x19: the pair kernel's launches follow each other across streams, net_device.h band_done - sub-groups only add launches:
    // 16 boards x 100 simulations 0.53 -> 0.55 M, 64 boards 0.86 -> 0.91 M leaf evaluations/s as one group)
    if (s->S == 19) { G = 1; fwd_cap = 0; }
    if (sub_env > 0) { G = sub_env; fwd_cap = 0; }
    if (tg::knob("TG_SP_FWD_CAP")) fwd_cap = atoi(tg::knob("TG_SP_FWD_CAP"));
    G = std::max(1, std::min(std::min(G, (int)tg_selfplay::kMaxSub), T));
    if (sp->observer || n_phases == 0) G = 1;
    if (G > 1) {
        const tg::LaunchCaps saved = tg::launch_caps();                // (caps belong to this thread's launches, net_device.h)
        tg::launch_caps() = tg::LaunchCaps{16, fwd_cap > 0 ? fwd_cap : 0};
        rc = launch_phases_subgroups(sp, net, n_phases, G, planes_dev, policy_dev, value_dev, st, leaves, any_phase);
        tg::launch_caps() = saved;
        if (rc) return rc;
    }
    for (int ph = 0; ph < n_phases && G == 1; ++ph) {
        const int32_t *nc = &sp->ph_nc[(size_t)ph * T], *mc = &sp->ph_mc[(size_t)ph * T];
        int64_t total = 0, slots = 0;
        for (int t = 0; t < T; ++t) {
            const int64_t n = (int64_t)nc[t] * mc[t];
            total += n;
            slots = n > slots ? n : slots;
        }
        if (slots == 0) continue;
        if ((rc = tg_search_select_gumbel(s, nc, mc, 0, planes_dev, stream))) return rc;
        if ((rc = tg_net_forward_dev(net, planes_dev, (int)total, 1, policy_dev, value_dev, stream))) return rc;
        if ((rc = tg_search_backup(s, policy_dev, value_dev, 0, 1, stream))) return rc;
        if (sp->observer) {
            tg_selfplay_event ev{};
            ev.kind = 0; ev.phase = ph; ev.trees = T; ev.positions = (int32_t)total;
            ev.num_considered = nc; ev.max_count = mc;
            ev.planes_dev = planes_dev; ev.policy_dev = policy_dev; ev.value_dev = value_dev; ev.stream = stream;
            sp->observer(sp->observer_user, &ev);
        }
        leaves += total;
        if (!any_phase && (rc = feed_streams_rest(s))) return rc;      // behind the first launched phase
        any_phase = true;
    }
    if ((rc = feed_streams_rest(s))) return rc;                        // (no phase was launched)
    // No phase at all - every board's game has just been started (the first call of a handle; later: all games of the group
    // ended with the same move, which a one-board group does every game): the new window must be in place BEFORE
    // finish_roots_kernel notes each board's cursor "behind the phases", or that note is the OLD window's last cursor and the
    // next move reads a root of nc - that many children off it (found in round 6 with one-board lanes: "board 0 has 82 root
    // children but its root expansion consumed 81 draws").
    if (!any_phase && (rc = install_rng(s, st))) return rc;
    // ---- the chain: records + decision, the moves played, the next roots expanded and evaluated ----
    const int max_moves = s->S * s->S * 2;                             // worker.py:44
    sp->state.assign((size_t)4 * T, 0);
    for (int t = 0; t < T; ++t) {
        const SpGame &g = sp->games[t];
        sp->state[4 * (size_t)t] = (sp->skip[t] ? 1 : 0) | (g.never_resign ? 2 : 0);
        sp->state[4 * (size_t)t + 1] = g.pass_count;
        sp->state[4 * (size_t)t + 2] = g.moves_played;
    }
    s->last_stream = st;
    s->stream_known = true;
    if ((rc = launch_finish_roots(s, sp->state.data(), max_moves, st))) return rc;
    if (s->S == 9) hipLaunchKernelGGL(play_kernel<9>, dim3(T), dim3(64), 0, st, s->dev, s->moves_dev);
    else if (s->S == 13) hipLaunchKernelGGL(play_kernel<13>, dim3(T), dim3(64), 0, st, s->dev, s->moves_dev);
    else hipLaunchKernelGGL(play_kernel<19>, dim3(T), dim3(64), 0, st, s->dev, s->moves_dev);
    TG_HIP(hipGetLastError());
    if ((rc = tg_search_root_planes(s, planes_dev, stream))) return rc;          // (also uploads the roots of games just started)
    if (!s->cur_pin) {
        TG_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->cur_pin), (size_t)T * sizeof(int64_t), hipHostMallocMapped));
        TG_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->cur_pin_dev), s->cur_pin, 0));
        TG_HIP(hipEventCreateWithFlags(&s->cur_ev, hipEventDisableTiming));
    }
    hipLaunchKernelGGL(publish_cursors_kernel, dim3((T + 255) / 256), dim3(256), 0, st, s->dev.rng_cursor, s->cur_pin_dev, T);
    TG_HIP(hipGetLastError());
    TG_HIP(hipEventRecord(s->cur_ev, st));                                      // (the host needs the cursors, not the evaluation)
    if ((rc = tg_net_forward_dev(net, planes_dev, T, 1, policy_dev, value_dev, stream))) return rc;
    if ((rc = tg_search_backup(s, policy_dev, value_dev, 1, 1, stream))) return rc;
    lap(4);
    // ---- host work that nobody is waiting for: the previous move's record comments, draws generated ahead ----
    flush_comments(sp);
    sp->last_window = window;            // (draws are no longer generated ahead on the host: the device produces them)
    lap(5);
    sp->pend = tg_selfplay::PendingMove{true, any_phase, leaves, n_phases, planes_dev, policy_dev, value_dev, stream, 0.0};
    return TG_OK;
}

static int chain_end(tg_selfplay *sp, int32_t *finished_host, int64_t *stats_host) {
    tg_search *s = sp->s;
    const int T = s->dev.T, A = s->A;
    int rc;
    static const bool timing = getenv("TG_SP_TIMING") != nullptr;
    double *acc = sp->t_acc;
    long &moves_timed = sp->moves_timed;
    double t_last = timing ? sp_now() : 0.0;
    auto lap = [&](int i) { if (timing) { const double t = sp_now(); acc[i] += t - t_last; t_last = t; } };
    const bool any_phase = sp->pend.any_phase;
    const int64_t leaves = sp->pend.leaves;
    const int32_t n_phases = sp->pend.n_phases;
    float *planes_dev = sp->pend.planes, *policy_dev = sp->pend.policy, *value_dev = sp->pend.value;
    void *stream = sp->pend.stream;
    sp->pend.active = false;
    // ---- the records (not the stream): bookkeeping ----
    TG_HIP(hipEventSynchronize(s->fin_ev));
    lap(6);
    const RootTail *tail = reinterpret_cast<const RootTail *>(s->roots_host + root_rec_bytes(A) * (size_t)T);
    for (int t = 0; t < T; ++t) sp->c_phase[t] = tail[t].cursor;
    sp->mv.resize(T);
    int64_t counts[2] = {0, 0};
    if (any_phase || sp->chain_started) {
        if ((rc = finish_move_impl(sp, sp->mv.data(), finished_host, counts, tail))) return rc;
        // the roots' child counts in the records must be what the draw cursors said
        for (int t = 0; t < T; ++t)
            if (!sp->skip[t] && sp->nc[t] != sp->nc_cursor[t])
                return tg::fail(TG_ERR_STATE, "tg_selfplay_play_move: board %d has %d root children but its root expansion "
                                "consumed %d draws - the halving schedule was built from a wrong width (cursor %lld behind the root, %lld "
                                "behind the phases before it; window %lld; game move %d)", t, sp->nc[t], sp->nc_cursor[t],
                                (long long)sp->consumed[t], (long long)(sp->consumed[t] - sp->nc_cursor[t]), (long long)s->win_cap,
                                sp->games[t].moves_played);
        if (sp->observer) {
            tg_selfplay_event ev{};
            ev.kind = 1; ev.phase = n_phases; ev.trees = T;
            ev.num_children = sp->nc.data(); ev.action = sp->act32.data(); ev.children_visits = sp->visits_a.data();
            ev.children_value_sum = sp->vsum_a.data(); ev.moves = sp->mv.data(); ev.finished = finished_host;
            sp->observer(sp->observer_user, &ev);
        }
    } else {
        for (int t = 0; t < T; ++t) finished_host[t] = 0;
        if ((rc = root_record_errors(s))) return rc;
    }
    if (sp->observer) {                                                // (the root evaluation's buffers are untouched since)
        tg_selfplay_event ev{};
        ev.kind = 0; ev.phase = -1; ev.trees = T; ev.positions = T;
        ev.planes_dev = planes_dev; ev.policy_dev = policy_dev; ev.value_dev = value_dev; ev.stream = stream;
        sp->observer(sp->observer_user, &ev);
    }
    for (int t = 0; t < T; ++t) sp->games[t].fresh = false;           // every root is expanded now (or about to be)
    sp->chain_started = true;
    lap(7);
    if (timing && ++moves_timed % 200 == 0)
        fprintf(stderr, "[selfplay timing (chained), ms per move over %ld moves] cursors %.3f | noise %.3f | schedule %.3f | feed %.3f | "
                "launches %.3f | comments + draws ahead %.3f | wait for records %.3f | bookkeeping %.3f\n", moves_timed,
                1e3 * acc[0] / moves_timed, 1e3 * acc[1] / moves_timed, 1e3 * acc[2] / moves_timed, 1e3 * acc[3] / moves_timed,
                1e3 * acc[4] / moves_timed, 1e3 * acc[5] / moves_timed, 1e3 * acc[6] / moves_timed, 1e3 * acc[7] / moves_timed);
    if (stats_host) { stats_host[0] = counts[0]; stats_host[1] = counts[1]; stats_host[2] = leaves; }
    return TG_OK;
}

static int play_move_sync(tg_selfplay *sp, tg_net *net, float *planes_dev, float *policy_dev, float *value_dev,
                          void *stream, int32_t *finished_host, int64_t *stats_host) {
    tg_search *s = sp->s;
    const int T = s->dev.T, A = s->A;
    int rc;
    s->prefill_enabled = false;
    // host-side wall clock per section (TG_SP_TIMING=1: printed every 200 moves) - where a move's time goes when
    // the GPU is not the bound
    static const bool timing = getenv("TG_SP_TIMING") != nullptr;
    double *acc = sp->t_acc;
    long &moves_timed = sp->moves_timed;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = timing ? now() : 0.0;
    auto lap = [&](int i) { if (timing) { const double t = now(); acc[i] += t - t_last; t_last = t; } };
    int live = 0;
    for (int t = 0; t < T; ++t) live += sp->games[t].done ? 0 : 1;
    // ---- root: expand, evaluate (tree.py:330-336) ----
    if ((rc = tg_search_feed_streams(s, (size_t)A, sp->force_feed ? 1 : 0))) return rc;
    sp->force_feed = false;
    lap(0);
    if ((rc = tg_search_root_planes(s, planes_dev, stream))) return rc;
    if ((rc = tg_net_forward_dev(net, planes_dev, T, 1, policy_dev, value_dev, stream))) return rc;
    if ((rc = tg_search_backup(s, policy_dev, value_dev, 1, 1, stream))) return rc;
    // the root expansion is the only consumer of draws in this launch, and a Dirichlet prior takes one draw per
    // child: the cursor read-back IS the roots' child counts (saves the schedule's own device read)
    sp->consumed.resize(T);
    if ((rc = tg_search_advance_streams(s, sp->consumed.data()))) return rc;
    // ... provided nothing went wrong in that launch (a sticky device error - pool full, window exhausted - would leave a
    // cursor that is not a child count): the count must be plausible here, and it is compared with the root statistics
    // read back behind the last phase, where the device error words are checked as well (no extra synchronisation here)
    sp->nc.resize(T);
    for (int t = 0; t < T; ++t) {
        if (sp->consumed[t] < 1 || sp->consumed[t] > A)
            return tg::fail(TG_ERR_STATE, "tg_selfplay_play_move: board %d consumed %lld draws at its root (expected 1..%d)",
                            t, (long long)sp->consumed[t], A);
        sp->nc[t] = (int32_t)sp->consumed[t];
    }
    sp->nc_known = true;
    sp->nc_cursor = sp->nc;
    if (sp->observer) {
        tg_selfplay_event ev{};
        ev.kind = 0; ev.phase = -1; ev.trees = T; ev.positions = T;
        ev.planes_dev = planes_dev; ev.policy_dev = policy_dev; ev.value_dev = value_dev; ev.stream = stream;
        sp->observer(sp->observer_user, &ev);
    }
    lap(1);
    if ((rc = tg_search_draw_noise(s, nullptr))) return rc;
    lap(2);
    // ---- sequential halving (tree.py:375-384) ----
    constexpr int kMaxPhases = 16;
    sp->ph_nc.resize((size_t)kMaxPhases * T);
    sp->ph_mc.resize((size_t)kMaxPhases * T);
    int32_t n_phases = 0;
    if ((rc = tg_selfplay_schedule(sp, sp->ph_nc.data(), sp->ph_mc.data(), kMaxPhases, &n_phases))) return rc;
    int64_t leaves = live;
    // Random draws the phases can consume: one Dirichlet prior (<= A draws) per EXPANSION, and only the first descent
    // through a root child can expand a node.  Root children entered in a phase: at most its width
    // (new ones: the picks among the unvisited children are nested prefixes of their score order, the first round's
    // being the largest) plus the children visited before the phase.  ONE window for all phases of the move is
    // generated and uploaded up front (the device cursor runs on from launch to launch) and the consumption is
    // read back once, behind the last phase - not a window, an upload and a host synchronisation per phase, each
    // sized as if every descent expanded (2.4x more draws).
    sp->ph_seen.assign(T, 0);
    int64_t window = 0, first_window = 0;
    for (int ph = 0; ph < n_phases; ++ph) {
        const int32_t *nc = &sp->ph_nc[(size_t)ph * T], *mc = &sp->ph_mc[(size_t)ph * T];
        int64_t expansions = 0;
        for (int t = 0; t < T; ++t) {
            const int64_t n = (int64_t)nc[t] * mc[t];
            const int64_t entered = std::min<int64_t>(n, std::min<int64_t>(A, (int64_t)nc[t] + sp->ph_seen[t]));
            expansions = entered > expansions ? entered : expansions;
            sp->ph_seen[t] = (int32_t)std::min<int64_t>(A, sp->ph_seen[t] + entered);
        }
        window += expansions * A;
        if (ph == 0) first_window = window;
    }
    lap(3);
    // (first part: what the first phase can consume; the rest goes up while that phase runs)
    if (window > 0 && (rc = feed_streams_impl(s, (size_t)window, 0, (size_t)first_window))) return rc;
    lap(4);
    bool any_phase = false;
    for (int ph = 0; ph < n_phases; ++ph) {
        const int32_t *nc = &sp->ph_nc[(size_t)ph * T], *mc = &sp->ph_mc[(size_t)ph * T];
        int64_t total = 0, slots = 0;
        for (int t = 0; t < T; ++t) {
            const int64_t n = (int64_t)nc[t] * mc[t];
            total += n;
            slots = n > slots ? n : slots;
        }
        if (slots == 0) continue;
        if ((rc = tg_search_select_gumbel(s, nc, mc, 0, planes_dev, stream))) return rc;
        if ((rc = tg_net_forward_dev(net, planes_dev, (int)total, 1, policy_dev, value_dev, stream))) return rc;
        if ((rc = tg_search_backup(s, policy_dev, value_dev, 0, 1, stream))) return rc;
        if (sp->observer) {
            tg_selfplay_event ev{};
            ev.kind = 0; ev.phase = ph; ev.trees = T; ev.positions = (int32_t)total;
            ev.num_considered = nc; ev.max_count = mc;
            ev.planes_dev = planes_dev; ev.policy_dev = policy_dev; ev.value_dev = value_dev; ev.stream = stream;
            sp->observer(sp->observer_user, &ev);
        }
        leaves += total;
        if (!any_phase && (rc = feed_streams_rest(s))) return rc;      // behind the first launched phase
        any_phase = true;
    }
    if ((rc = feed_streams_rest(s))) return rc;                        // (no phase was launched)
    lap(3);
    // ---- host work that nobody is waiting for, while the phase kernels run: the previous move's record comments, and
    //      the draws the next move will ask for (root prior, noise, a window like this move's) generated ahead ----
    flush_comments(sp);
    sp->last_window = window;
    lap(4);
    if (any_phase && (rc = tg_search_advance_streams(s, nullptr))) return rc;
    lap(5);
    // ---- move choice, records, finished games; play ----
    sp->mv.resize(T);
    int64_t counts[2] = {0, 0};
    lap(3);
    if ((rc = tg_selfplay_finish_move(sp, sp->mv.data(), finished_host, counts))) return rc;
    if ((rc = check_errors(s))) return rc;          // (the stream is drained here: finish_move has read the roots back)
    // the roots' child counts as the statistics read-back reports them must be what the draw cursor said
    for (int t = 0; t < T; ++t)
        if (sp->nc[t] != sp->nc_cursor[t])
            return tg::fail(TG_ERR_STATE, "tg_selfplay_play_move: board %d has %d root children but its root expansion "
                            "consumed %d draws - the halving schedule was built from a wrong width", t, sp->nc[t], sp->nc_cursor[t]);
    if (sp->observer) {
        tg_selfplay_event ev{};
        ev.kind = 1; ev.phase = n_phases; ev.trees = T;
        ev.num_children = sp->nc.data(); ev.action = sp->act32.data(); ev.children_visits = sp->visits_a.data();
        ev.children_value_sum = sp->vsum_a.data(); ev.moves = sp->mv.data(); ev.finished = finished_host;
        sp->observer(sp->observer_user, &ev);
    }
    lap(6);
    if ((rc = tg_search_play(s, sp->mv.data(), stream))) return rc;
    lap(7);
    if (timing && ++moves_timed % 200 == 0)
        fprintf(stderr, "[selfplay timing, ms per move over %ld moves] feed(root) %.3f | root planes + sync %.3f | root forward/backup/noise %.3f | "
                "launches %.3f | feed(phases) %.3f | select + sync %.3f | finish_move %.3f | play %.3f\n", moves_timed,
                1e3 * acc[0] / moves_timed, 1e3 * acc[1] / moves_timed, 1e3 * acc[2] / moves_timed, 1e3 * acc[3] / moves_timed,
                1e3 * acc[4] / moves_timed, 1e3 * acc[5] / moves_timed, 1e3 * acc[6] / moves_timed, 1e3 * acc[7] / moves_timed);
    if (stats_host) { stats_host[0] = counts[0]; stats_host[1] = counts[1]; stats_host[2] = leaves; }
    return TG_OK;
}

}  // extern "C"
