/*
 * tamago_hip.h - C ABI of libtamago_hip.so, the MI355X (gfx950) implementation of
 * TamaGo's batched MCTS leaf-evaluation path.
 *
 * The reference (kobanium/TamaGo @ 2024-12-20) is 100 % Python and has no FFI of its
 * own; each entry point below replaces the Python function(s) cited next to it, and is
 * what a ctypes binding inside the reference would call (see INTEGRATION.md for the
 * exact stubs).  Conventions:
 *   - every function returns 0 on success, a negative tg_status on failure; the
 *     message is available from tg_last_error() (thread-local);
 *   - handles are opaque; a handle is bound to one HIP device and is NOT thread-safe;
 *   - "host" pointers are caller-owned host memory, "dev" pointers are caller-owned
 *     device memory on the handle's device (e.g. torch tensors' data_ptr());
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls that
 *     take a stream only ENQUEUE work; the *_host convenience calls synchronise.
 *   - board coordinates follow the reference: pos = x + y*(S+2) on the padded board,
 *     PASS = 0, RESIGN = -1 (board/constant.py:4-31); colours EMPTY 0 / BLACK 1 /
 *     WHITE 2 / OUT_OF_BOARD 3 (board/stone.py:5-11).
 */
#ifndef TAMAGO_HIP_H
#define TAMAGO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum tg_status {
    TG_OK = 0,
    TG_ERR_ARG = -1,      /* bad argument                          */
    TG_ERR_HIP = -2,      /* HIP runtime error (no GPU, OOM, ...)  */
    TG_ERR_STATE = -3,    /* call sequence violated                */
    TG_ERR_OVERFLOW = -4  /* node pool / RNG window exhausted      */
} tg_status;

/* ---- library ------------------------------------------------------------------------ */
int tg_abi_version(void);                 /* increases on every incompatible change      */
const char *tg_last_error(void);          /* message of the last failing call            */
int tg_device_count(int *count);          /* replaces nn/utility.py:12-24 get_torch_device */

/* ---- DualNet forward (nn/network/dual_net.py:41-106, res_block.py:8-38,
 *      head/policy_head.py:7-39, head/value_head.py:7-39) ------------------------------- */
typedef struct tg_net tg_net;

/* Number of floats tg_net_create expects for a board size: the reference state_dict
 * (nn/utility.py:139-159) flattened in this key order, num_batches_tracked skipped:
 *   conv_layer.weight [64,6,3,3]; bn_layer.{weight,bias,running_mean,running_var} [64];
 *   for b in 0..5: blocks.b.conv1.weight, blocks.b.conv2.weight [64,64,3,3],
 *                  blocks.b.bn1.{w,b,mean,var}, blocks.b.bn2.{w,b,mean,var} [64];
 *   policy_head.conv_layer.weight [2,64,1,1]; policy_head.bn_layer.{w,b,mean,var} [2];
 *   policy_head.fc_layer.weight [A,2P]; policy_head.fc_layer.bias [A];
 *   value_head.conv_layer.weight [1,64,1,1]; value_head.bn_layer.{w,b,mean,var} [1];
 *   value_head.fc_layer.weight [3,P]; value_head.fc_layer.bias [3].       (P=S*S, A=P+1) */
size_t tg_net_param_count(int board_size);

/* Build a network on `device` from raw (un-folded) fp32 parameters; BatchNorm (eval
 * mode, eps 1e-5 stem / 2e-5 elsewhere) is folded and the 3x3 weights are re-ordered
 * into MFMA fragment order on the host.  Replaces load_network, nn/utility.py:139-159. */
int tg_net_create(int board_size, int device, const float *params, size_t n_params,
                  tg_net **out);
int tg_net_destroy(tg_net *net);
int tg_net_board_size(const tg_net *net);

/* Forward pass on device-resident planes [B,6,S,S] fp32 -> policy [B,A], value [B,3].
 * want_logits = 0: softmax(policy), softmax(value)     (DualNet.inference, :81-91)
 * want_logits = 1: raw policy logits, softmax(value)   (inference_with_policy_logits, :94-106) */
int tg_net_forward_dev(tg_net *net, const float *planes_dev, int batch, int want_logits,
                       float *policy_dev, float *value_dev, void *stream);
/* Same with host buffers (H2D, kernel, D2H, synchronise) - the reference's own boundary:
 * host tensor in, host tensors out. */
int tg_net_forward_host(tg_net *net, const float *planes_host, int batch, int want_logits,
                        float *policy_host, float *value_host);
/* Profiling aid: one forward pass with workgroup 0 writing s_memtime stamps at its phase
 * boundaries (group start, stem, per layer: work done / barrier passed, heads done).
 * Direct kernel: wave 0 -> stamps [0,64).  Winograd kernel: wave 0 -> [0,64) and wave 4 (the
 * other wave of the same SIMD) -> [64,128).  n_stamps <= 128.  Synchronises. */
int tg_net_profile_phases(tg_net *net, const float *planes_dev, int batch, float *policy_dev,
                          float *value_dev, long long *stamps_host, int n_stamps);
/* Name (for rocprof) and algorithmic FLOPs per position of the dominant kernel. */
const char *tg_net_kernel_name(const tg_net *net, int batch);
double tg_net_flops_per_position(int board_size);
/* What the kernel tg_net_forward_dev picks for `batch` actually EXECUTES on the matrix pipe, per position:
 * FLOPs of the issued MFMA instructions including tile padding and operand splitting (split-operand f16
 * kernel: 3 MFMAs per product-sum over the direct 3x3 convolution, 243 of 256 rows used; Winograd fp32
 * kernel: 25 tiles x 16 points, 75 of 80 tiles used; direct fp32 kernel: 9 taps).  *peak_tflops receives
 * the dense matrix peak of the precision those instructions run in (MI355X: 2500 f16, 157.3 fp32),
 * *dtype its name.  The roofline fraction of the forward kernel is executed FLOP/s over that peak. */
double tg_net_executed_flops_per_position(const tg_net *net, int batch, double *peak_tflops,
                                          const char **dtype);
/* The split-operand kernels compute in f16 pieces: a layer output beyond the f16 range raises a flag on the device and the
 * exact-fp32 kernel queued behind every launch redoes that batch (same results, ~2.5x the time; no host round trip).
 * *count receives the number of forward launches of this network that were redone so far - a network whose activations
 * run hot shows up here instead of only as a slower search.  Synchronises the device.  (No reference counterpart: the
 * reference computes in fp32 throughout, nn/network/dual_net.py:41-52.) */
int tg_net_range_fallbacks(tg_net *net, unsigned long long *count);
/* ... and how many POSITIONS the exact-fp32 kernel redid in those launches.  The one-axis Winograd kernels (the defaults at both
 * board sizes) mark the workgroup passes - 3 boards (1 in small launches) at 9x9, one board at 19x19 - whose activations left
 * the f16 range, and the exact kernel redoes those only: one hot position in a launch of half a million costs three positions'
 * redo, not the launch's.  (The direct split kernels and a band's time-out redo the whole launch.)  Synchronises the device. */
int tg_net_range_fallback_positions(tg_net *net, unsigned long long *count);
/* 19x19: launches of up to 128 boards spread a board over 2 / 4 workgroups that exchange halo rows through L2 with BOUNDED
 * waits.  *count receives how many of those waits gave up so far (each ends in the exact-fp32 redo above, so it is also
 * part of tg_net_range_fallbacks' count - this one tells the two causes apart).  After the first one the network keeps to
 * the one-workgroup kernel: somebody else's kernels hold this GPU's compute units.  Synchronises the device. */
int tg_net_band_timeouts(tg_net *net, unsigned long long *count);
/* shared != 0: other PROCESSES drive this GPU too (the reference's `--process N` on one device, nn/utility.py:22,
 * selfplay_main.py:44-65 with more workers than GPUs).  Kernels that need several workgroups of one launch resident at the
 * same time (the banded 19x19 forward) are not chosen then.  Results do not depend on this switch. */
int tg_net_set_shared_device(tg_net *net, int shared);

/* ---- featurise (nn/feature.py:10-57 + go_board.py:468-478) -------------------------- */
/* cells_dev: uint8 [B, P] on-board cell colours, row-major from the top-left point;
 * to_move: int8 [B] (1 black / 2 white); prev_move: int32 [B] padded-board coordinate of
 * the previous move (0 = PASS); moves: int32 [B] the board's move counter (starts at 1).
 * Writes fp32 planes [B,6,S,S]. */
int tg_featurize_dev(int board_size, const uint8_t *cells_dev, const int8_t *to_move_dev,
                     const int32_t *prev_move_dev, const int32_t *moves_dev, int batch,
                     float *planes_dev, void *stream);

/* Same with a board symmetry per position (sym_dev int8 [B], 0..7 as in go_board.py:80-104;
 * NULL = identity): the training-side call generate_input_planes(board, color, sym) of
 * nn/data_generator.py:108,125. */
int tg_featurize_sym_dev(int board_size, const uint8_t *cells_dev, const int8_t *to_move_dev,
                         const int32_t *prev_move_dev, const int32_t *moves_dev,
                         const int8_t *sym_dev, int batch, float *planes_dev, void *stream);

/* ---- batched tree search (mcts/tree.py, mcts/node.py, mcts/pucb/pucb.py,
 *      mcts/batch_data.py, board/ as called from the search) --------------------------- */
typedef struct tg_search tg_search;

typedef struct tg_search_config {
    int32_t board_size;      /* S: 9 or 19 (any 5..19)                                    */
    int32_t num_trees;       /* T independent search trees ("boards") driven in lock-step */
    int32_t tree_size;       /* nodes per tree (MCTSTree tree_size, tree.py:29)           */
    int32_t batch_size;      /* leaves per tree per mini-batch (NN_BATCH_SIZE)            */
    int32_t cgos_mode;       /* node.py:153-155                                           */
    int32_t check_superko;   /* GoBoard(check_superko=...), go_board.py:285-301           */
    int32_t device;
    int32_t reserved;
} tg_search_config;

/* Root position of one tree, as the reference's GoBoard holds it. */
typedef struct tg_root_position {
    const uint8_t *cells;        /* [(S+2)^2] padded board, colours incl. OUT_OF_BOARD    */
    const uint64_t *hash_history;/* [moves] positional hashes by move index (slot 0 = 0);
                                    may be NULL when check_superko == 0                    */
    uint64_t hash;               /* current positional hash                               */
    int32_t moves;               /* GoBoard.moves (1 = empty game)                         */
    int32_t ko_pos, ko_move;     /* go_board.py:173-177                                   */
    int32_t prev_move;           /* record.pos[moves-1] (0 = PASS / none)                 */
    int32_t prev_prev_move;      /* record.pos[moves-2]                                   */
    int32_t to_move;             /* 1 black / 2 white                                     */
} tg_root_position;

int tg_search_create(const tg_search_config *cfg, tg_search **out);
int tg_search_destroy(tg_search *s);

/* Zobrist keys uint64 [4][(S+2)^2] used for positional superko (board/zobrist_hash.py);
 * the caller owns the key table so that its own GoBoard hashes stay consistent. */
int tg_search_set_zobrist(tg_search *s, const uint64_t *keys, size_t n);

/* Stage the root position of tree `t` (copied; uploaded in bulk by the next
 * tg_search_root_planes, which resets the tree: tree.py:49-54 / :330-336). */
int tg_search_set_root(tg_search *s, int tree, const tg_root_position *pos);

/* Feed the per-tree random streams.  The reference draws the Dirichlet "tentative"
 * prior (tree.py:509-519) and Gumbel noise (node.py:275-278) from numpy's global
 * legacy MT19937 stream; bit-exact parity needs the host's libm log(), so the host
 * turns uniform doubles u_i into e_i = -log(1-u_i) and hands them over in stream
 * order: tree t consumes exp_stream[t*stride + cursor ...].  `count` values per tree.
 * The window is uploaded on a private copy stream into the inactive one of two device
 * windows and becomes active at the next root_planes / select call, so the host can
 * prepare mini-batch j+1 while the forward pass of mini-batch j is still running. */
int tg_search_set_rng(tg_search *s, const double *exp_stream_host, size_t stride, size_t count);
/* Doubles each tree consumed from the active window (host array [T]); waits only for the
 * last root_planes / select kernel, not for the forward or backup queued behind it. */
int tg_search_rng_consumed(tg_search *s, int64_t *consumed_host);

/* PUCT: run `max_leaves` (<= batch_size) descents per tree (tree.py:199-244 search_mcts:
 * select by PUCB, play, virtual loss, expand, featurise, queue).  Leaf planes go to
 * planes_dev [T, max_leaves, 6, S, S]; n_leaves_dev[T] (may be NULL) receives the number
 * queued per tree (== max_leaves unless the tree hit an error). */
int tg_search_select_puct(tg_search *s, int max_leaves, float *planes_dev, int32_t *n_leaves_dev,
                          void *stream);
/* n_batches PUCT mini-batches (leaves_host[b] descents per tree each) queued back to back on `stream`: one random window from the
 * library's streams for all of them (tg_search_feed_streams semantics, force_window as its `force`), then per mini-batch
 * tg_search_select_puct, tg_net_forward_dev (want_logits 0) and tg_search_backup - the launches of the per-mini-batch calls
 * without the host round trip between them.  For searches whose course does not depend on what a mini-batch found (the
 * reference's STRICT_PLAYOUT: no early stop, mcts/time_manager.py:160-161; mcts/tree.py:146-152).  policy_dev [T * batch_size, A],
 * value_dev [T * batch_size, 3] and planes_dev are reused by every mini-batch.  Afterwards: tg_search_advance_streams. */
int tg_search_puct_chain(tg_search *s, tg_net *net, const int32_t *leaves_host, int n_batches, int force_window,
                         float *planes_dev, float *policy_dev, float *value_dev, void *stream);
/* Reset every tree to its root position, expand the root (consumes the root's Dirichlet
 * draw) and write the root planes [T,6,S,S] (tree.py:49-53); one leaf per tree is queued. */
int tg_search_root_planes(tg_search *s, float *planes_dev, void *stream);
/* Everything the Gumbel move choice and the improved policy need (node.py:281-346), for the
 * roots of all trees in one call; arrays are [T] / [T][A]; any pointer may be NULL. */
int tg_search_read_root_stats(tg_search *s, int32_t *num_children_host, int32_t *node_visits_host,
                              float *raw_value_host, int32_t *action_host, int32_t *visits_host,
                              int32_t *virtual_loss_host, double *value_sum_host,
                              double *policy_host);
/* Profiling aid: with enable != 0 the PUCT selection kernel accumulates s_memtime cycles of
 * tree 0 per phase into 16 counters (0 board reset, 1 PUCB select, 2 put_stone, 3 edge
 * bookkeeping, 4 expansion, 5 planes + queue, 7 = number of tree levels walked, 8-10
 * expansion detail: candidates / Dirichlet sum / node init); the call returns the counters
 * accumulated so far (cycles_host [16], may be NULL) and clears them. */
int tg_search_profile(tg_search *s, int enable, long long *cycles_host);
/* Play moves_host[t] (padded coordinate, 0 = PASS, -1 = leave the tree alone, -2 = the move of the most
 * visited child of the tree's root, node.py:167-175 get_best_move_index, chosen on the device: no
 * read-back between two searches) on the ROOT position of every tree on the device (GoBoard.put_stone,
 * go_board.py:131-185) and flip the side to move: self-play boards stay resident between searches. */
int tg_search_play(tg_search *s, const int32_t *moves_host, void *stream);
/* Current root positions: cells uint8 [T][(S+2)^2], GoBoard.moves [T], side to move [T]
 * (any pointer may be NULL). Synchronises. */
int tg_search_read_positions(tg_search *s, uint8_t *cells_host, int32_t *moves_host,
                             int32_t *to_move_host);
/* ---- library-owned random streams ---------------------------------------------------------
 * Alternative to tg_search_set_rng / tg_search_rng_consumed / tg_search_set_noise for hosts that
 * do not want to generate the draws themselves: the library continues numpy's legacy stream
 * (RandomState: MT19937, random_sample, standard_exponential = -log(1-u), gumbel = -log(-log(1-u)))
 * from a given generator state - what np.random.dirichlet(ones(n)) (mcts/tree.py:518) and
 * np.random.gumbel(size=A) (mcts/node.py:278) consume - one stream per tree.
 * seed_stream: mt_key = the 624 words, mt_pos = the index of np.random.get_state()[1:3].
 * stream_state: generator state after everything the tree has consumed so far (hand it back with
 *   np.random.set_state(('MT19937', key, pos, 0, 0.0))).
 * feed_streams: make `need` draws per tree available to the next root / select launch (no-op
 *   while the generated window still covers `need` for every tree, unless force != 0); the window is
 *   generated by a kernel on a stream of the library's own and overlaps running kernels.
 * advance_streams: after a root / select launch - wait for it, move every stream by what its tree
 *   consumed (optionally reported in consumed_host [T]).
 * draw_noise: set_gumbel_noise for every root from the next A draws of each stream (generated on the
 *   device, ordered like tg_search_set_noise's upload; copy returned in noise_host [T][A] unless NULL). */
int tg_search_seed_stream(tg_search *s, int tree, const uint32_t *mt_key, int mt_pos);
int tg_search_stream_state(tg_search *s, int tree, uint32_t *mt_key_out, int *mt_pos_out);
int tg_search_feed_streams(tg_search *s, size_t need, int force);
int tg_search_advance_streams(tg_search *s, int64_t *consumed_host);
int tg_search_draw_noise(tg_search *s, double *noise_host);
/* Since round 6 the streams live ON THE DEVICE (csrc/legacy_rng_device.h: MT19937, random_sample and glibc's table-driven
 * double-precision log - the build numpy's legacy distributions reach through libm on an FMA-capable x86-64 - restated
 * operation by operation): windows and noise are generated there, the host only counts what was consumed.
 * Host-only helpers (no device needed) that run the SAME restated arithmetic on the host, for the CPU tests:
 *   tg_legacy_exponentials: the next n legacy standard_exponential draws of the generator (mt_key, *mt_pos), updated in place;
 *   tg_glibc_log: out[i] = log(x[i]) (positive normal arguments) - held against Python's math.log (= libm) bit for bit. */
int tg_legacy_exponentials(uint32_t *mt_key, int *mt_pos, size_t n, double *out);
int tg_glibc_log(const double *x, size_t n, double *out);
/* A HIP stream owned by the handle (created on first request, destroyed with it; hipStreamNonBlocking) for callers that run
 * several handles side by side - the lanes of a self-play shard (selfplay/worker.py) - and want each on a stream of its own
 * without going through a host framework's stream pool. */
int tg_search_own_stream(tg_search *s, void **stream_out);
/* Test hooks of the device streams (tests/test_gpu_rng.py): read columns [first, first + count) of tree `tree`'s row of the
 * most recently generated window; walk the streams as a search would - per step a window of steps[i] + slack draws, whole
 * (part 0) or in pieces of `part` draws, steps[i] of which count as consumed - without running a search. */
int tg_search_debug_read_window(tg_search *s, int tree, size_t first, size_t count, double *out_host);
int tg_search_debug_stream_walk(tg_search *s, const int64_t *steps, int n_steps, int64_t slack, int64_t part);
/* Gumbel root noise, float64 [T][A] (node.py:275-278 set_gumbel_noise), to be set after the
 * root evaluation of a Gumbel move. */
int tg_search_set_noise(tg_search *s, const double *noise_host);
/* One sequential-halving phase (tree.py:375-383): per tree, for count_threshold in
 * 1..max_count[t], num_considered[t] descents (root: node.py:324-346, below: :349-361);
 * every descent queues one leaf.  Planes [T, slots_per_tree, 6, S, S]; trees whose phase
 * is (0, 0) idle.  slots_per_tree = 0 selects the PACKED layout: the leaves of tree t start at
 * plane sum_{u<t} num_considered[u] * max_count[u], so the forward pass covers exactly the
 * queued leaves (one tree with a single root candidate runs 1 x visits levels and would
 * otherwise stretch every tree's slot range).  Follow with the forward pass and
 * tg_search_backup(same slots_per_tree, use_logit = 1). */
int tg_search_select_gumbel(tg_search *s, const int32_t *num_considered_host,
                            const int32_t *max_count_host, int slots_per_tree,
                            float *planes_dev, void *stream);
/* Write NN outputs back and back up values (tree.py:273-315 process_mini_batch).
 * policy_dev [T, slots_per_tree, A], value_dev [T, slots_per_tree, 3] in the slot order
 * of the preceding call (slots_per_tree = its max_leaves, 1 after root_planes, 0 after a packed
 * tg_search_select_gumbel: policy_dev [total, A], value_dev [total, 3]);
 * use_logit as in tree.py:293-294. */
int tg_search_backup(tg_search *s, const float *policy_dev, const float *value_dev,
                     int slots_per_tree, int use_logit, void *stream);

/* Path of queued leaf `slot` of tree `tree` after a selection launch, root first:
 * (node index, child index) per level - the `path` list of search_mcts (tree.py:199-244) that
 * search_with_callback (tree.py:177-196) hands to its callback.  Valid for the slots the last
 * selection launch queued, until the next one.  Synchronises. */
int tg_search_read_path(tg_search *s, int tree, int slot, int32_t *nodes_host, int32_t *edges_host,
                        int capacity, int32_t *length_host);
/* Leaf queue of tree `tree` as the last selection launch left it (mcts/batch_data.py:7-34, the
 * `node_index` list of BatchQueue): count_host = leaves queued, node_index_host[i] = node that
 * receives leaf i's policy (-1 = the reference's node[-1] slot, tree.py:222-233 when the leaf was
 * already expanded).  The matching `input_plane` entries are the planes the selection wrote, the
 * `path` entries come from tg_search_read_path.  Valid until the next selection launch.  Synchronises. */
int tg_search_read_queue(tg_search *s, int tree, int32_t *node_index_host, int capacity,
                         int32_t *count_host);
/* Grow the node pool of every tree to new_tree_size nodes, keeping the trees (the reference doubles
 * its node list in place when it fills up, mcts/tree.py:254-258).  No-op if the pool is already that
 * large.  Synchronises the device. */
int tg_search_grow(tg_search *s, int new_tree_size);
/* Read-side of MCTSNode for node `node` of tree `t` (node.py:21-39); any pointer may be
 * NULL.  Arrays have A entries. Synchronises the stream used by the last call. */
int tg_search_read_node(tg_search *s, int tree, int node, int32_t *num_children,
                        int32_t *node_visits, int32_t *node_virtual_loss, int32_t *action,
                        int32_t *children_index,
                        int32_t *children_visits, int32_t *children_virtual_loss,
                        double *children_value_sum, double *children_policy,
                        double *children_value, float *node_value_sum, float *raw_value);
/* num_nodes of the tree that tg_search_read_node read last, as of that read (same record, no device access). */
int tg_search_node_record_num_nodes(tg_search *s, int32_t *num_nodes_host);
int tg_search_num_nodes(tg_search *s, int32_t *num_nodes_host /* [T] */);
/* Root statistics of all trees in one call: num_children [T], action [T][A],
 * children_visits [T][A] (what get_best_move reads, node.py:169-184). Synchronises. */
int tg_search_read_roots(tg_search *s, int32_t *num_children_host, int32_t *action_host,
                         int32_t *visits_host);

/* ---- self-play shard bookkeeping (selfplay/worker.py:50-90, sgf/selfplay_record.py:45-110) -----------
 * Everything the reference's worker does per move and per board AROUND the search, for all T boards of a
 * search handle in one host call per move (C++, host threads): sequential-halving schedule
 * (mcts/sequential_halving.py), final root choice (tree.py:344, node.py:324-346), resign rule
 * (tree.py:351-354), improved-policy comment with ".3e" values (node.py:281-321, selfplay_record.py:45-65),
 * two-pass end with count_score (go_board.py:561-608), maximum length (worker.py:44), the SGF file
 * <save_dir>/<index>.sgf byte for byte as selfplay_record.py:67-110 writes it.
 * Per move the caller does: tg_search_root_planes / forward / tg_search_backup, tg_search_draw_noise,
 * tg_selfplay_schedule, one tg_search_select_gumbel / forward / tg_search_backup per phase,
 * tg_selfplay_finish_move, tg_search_play(moves); slots whose game finished get tg_search_set_root +
 * tg_search_seed_stream + tg_selfplay_start_game (or index -1 to park them). */
typedef struct tg_selfplay tg_selfplay;
/* komi_text: the komi as the SGF shall spell it (the reference prints Python's repr, e.g. "7.0"). */
int tg_selfplay_create(tg_search *s, const char *save_dir, int visits, double komi, const char *komi_text,
                       tg_selfplay **out);
int tg_selfplay_destroy(tg_selfplay *sp);
int tg_selfplay_start_game(tg_selfplay *sp, int slot, int index, int never_resign);
/* Phases of the coming move for every board: num_considered / max_count host arrays [max_phases][T] (zero
 * where a board has no such phase or is parked), *n_phases_host = phases of the longest schedule. */
int tg_selfplay_schedule(tg_selfplay *sp, int32_t *num_considered_host, int32_t *max_count_host,
                         int max_phases, int32_t *n_phases_host);
/* After the last phase: moves_host[T] = move to play on each board (-1 = none: parked, resigned or game
 * over), finished_host[T] = 1 where the game ended (its file is written), stats_host[2] = {games finished,
 * moves decided} by this call (may be NULL). */
int tg_selfplay_finish_move(tg_selfplay *sp, int32_t *moves_host, int32_t *finished_host,
                            int64_t *stats_host);
/* The whole move above in ONE call when the evaluator is a tg_net (root evaluation, noise, schedule, every
 * phase with its forward pass, tg_selfplay_finish_move, tg_search_play): nothing but library code runs
 * between the launches.  Caller's device buffers: planes [T*batch_size,6,S,S], policy [T*batch_size,A],
 * value [T*batch_size,3].  finished_host[T] as above; stats_host[3] = {games finished, moves decided, leaf
 * evaluations} of this move (may be NULL).
 * Default scheme ("chained"): the device decides the move itself (final choice, resign rule, game end - the host's
 * arithmetic, which the host repeats on the same statistics and compares), plays it and expands + evaluates the
 * next root without the host; the call returns once the root RECORDS have arrived, with that root evaluation still
 * running, and the next call starts from it.  Consequences a caller sees: the first call of a handle only evaluates
 * roots (no move, nothing finished); a slot whose game was just started (tg_selfplay_start_game) sits out one call;
 * below 385 boards the phases run in 2 - 4 sub-groups of boards on streams of the library's own (joined into
 * `stream` before the call returns control of the buffers).  Games, records and draw order per game are the same in
 * every scheme.  TG_SP_CHAIN=0: the move decided on the host (three round trips per move); TG_SP_SUBGROUPS=n /
 * TG_SP_FWD_CAP=n override the grouping.  With an observer the boards stay in one group. */
int tg_selfplay_play_move(tg_selfplay *sp, tg_net *net, float *planes_dev, float *policy_dev,
                          float *value_dev, void *stream, int32_t *finished_host, int64_t *stats_host);
/* The two halves of a chained tg_selfplay_play_move, for a caller that keeps several handles (lanes of one shard: each its own
 * tg_search, buffers and stream) in flight from one host thread - selfplay/worker.py:46-90 plays its games one after the other;
 * games are independent, so lanes need not be on the same move.  `begin` queues the whole move (phases, decision, the moves
 * played, next root evaluation) on `stream` and returns; `end` waits for that move's root records, does the bookkeeping
 * (records, finished games, SGF) and fills finished_host[T] / stats_host[3] as tg_selfplay_play_move does.  Between a handle's
 * `begin` and `end` only other handles may be driven; slots are refilled (tg_selfplay_start_game) after `end`.  The buffers
 * belong to the library until `end` returns.  Not with an observer or TG_SP_CHAIN=0 (TG_ERR_STATE). */
int tg_selfplay_move_begin(tg_selfplay *sp, tg_net *net, float *planes_dev, float *policy_dev, float *value_dev, void *stream);
int tg_selfplay_move_end(tg_selfplay *sp, int32_t *finished_host, int64_t *stats_host);
/* Audit hook of tg_selfplay_play_move (parity tests replay what the one-call path evaluated into the CPU oracle,
 * mini-batch by mini-batch: mcts/tree.py:273-315 process_mini_batch is where the reference would be tapped).
 * The observer is called on the calling thread
 *   kind 0: after a mini-batch's forward pass and backup were ENQUEUED on `stream` and before the next selection
 *           overwrites the buffers - synchronise `stream`, then planes_dev [positions,6,S,S], policy_dev
 *           [positions,A], value_dev [positions,3] hold that mini-batch.  phase = -1: root evaluation, one leaf
 *           per board in board order; phase >= 0: sequential-halving phase, PACKED layout - the leaves of board t
 *           start at sum_{u<t} num_considered[u] * max_count[u] (host arrays [trees]);
 *   kind 1: after the moves of all boards were decided (tg_selfplay_finish_move) and before they are played: root
 *           statistics of every board as host arrays num_children [trees], action / children_visits [trees][A],
 *           children_value_sum [trees][A], moves [trees] (-1 = none), finished [trees].
 * Pointers are valid until the observer returns.  fn = NULL removes the observer.  Results do not depend on it. */
typedef struct tg_selfplay_event {
    int32_t kind, phase, trees, positions;
    const int32_t *num_considered, *max_count;                 /* kind 0, phase >= 0 */
    const float *planes_dev, *policy_dev, *value_dev;          /* kind 0 */
    void *stream;
    const int32_t *num_children, *action, *children_visits;    /* kind 1 */
    const double *children_value_sum;
    const int32_t *moves, *finished;
} tg_selfplay_event;
typedef void (*tg_selfplay_observer)(void *user, const tg_selfplay_event *event);
int tg_selfplay_set_observer(tg_selfplay *sp, tg_selfplay_observer fn, void *user);

/* ---- training step (nn/learn.py:318-403, nn/loss.py:9-55; modules of nn/network/) ---------------------------
 * One mini-batch of the reference's GPU trainers as hand-written HIP kernels (forward with batch statistics,
 * backward, torch.optim.SGD(momentum 0.9, weight_decay 1e-4, nesterov=True) update, batch-norm running
 * statistics), fp32.  The trainer owns a device copy of the parameters in the tg_net_create blob order
 * (running_mean / running_var included; num_batches_tracked is the caller's counter) and a momentum blob of
 * the same layout.  board_size 9 or 19 (other sizes: TG_ERR_ARG); batch = positions per step (>= 2).
 *   tg_trainer_step: planes [B,6,9,9] fp32, policy targets [B,82] fp32, value classes [B] int64, all device
 *     memory; sl_mode 0 = RL objective (KL(target || softmax) batch mean + value_weight * cross entropy,
 *     learn.py:360-376), 1 = supervised (-sum t log(softmax + 1e-8), learn.py:150-180); ENQUEUES the step.
 *   tg_trainer_read_losses: device-accumulated sums of (total, policy, value) loss over the steps since the
 *     last reset, one read for many steps. */
typedef struct tg_trainer tg_trainer;
int tg_trainer_create(int board_size, int device, int batch, const float *params_host, size_t n_params,
                      tg_trainer **out);
int tg_trainer_destroy(tg_trainer *t);
int tg_trainer_step(tg_trainer *t, const float *planes_dev, const float *policy_dev,
                    const long long *value_dev, int sl_mode, float value_weight, float lr, void *stream);
int tg_trainer_read_losses(tg_trainer *t, double *sums_host /* [3] */, int reset);
/* parameters (and, unless NULL, the momentum buffers) back to the host; n = tg_net_param_count(board_size) */
int tg_trainer_get_params(tg_trainer *t, float *params_host, float *momentum_host, size_t n);
/* test aid: one saved tensor of the last step, NHWC fp32 [B][board_size^2][64]: which 0 = Z_index (convolution output
 * before its batch norm, index 0..12), 1 = Y_index (block output, 0..6), 2 = D_index (dL/d batch-norm output) */
int tg_trainer_debug_read(tg_trainer *t, int which, int index, float *out_host);
/* resume: momentum buffers of a loaded optimiser state (the next step is then not a "first" step) */
int tg_trainer_set_momentum(tg_trainer *t, const float *momentum_host, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* TAMAGO_HIP_H */
