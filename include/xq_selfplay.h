/*
 * xq_selfplay.h — C ABI of the MI355X-native batched self-play engine (libxq_hip.so).
 *
 * The reference (hpy666666/ChineseChessAI) is pure Python and has no FFI layer; its boundary
 * for this hot path is the Python call surface of chess_env.py / self_play.py
 * (SURVEY.md §8b).  The entry points below are what a ctypes binding of THAT surface needs;
 * each cites the reference interface it replaces.  The reference-side stub a maintainer would
 * add is shown in INTEGRATION.md; the in-tree host mirror is the chinesechessai_amd package.
 *
 * Conventions: extern "C", plain pointers and sizes, no exceptions across the boundary, return
 * 0 on success or a negative XQ_E_* code; xq_last_error() gives the message.  Handles are
 * thread-compatible (one thread at a time per handle).  Pointers named *_host are host memory,
 * *_dev are device (HBM) memory on the engine's device.  All launches go to the stream given to
 * xq_engine_set_stream (a hipStream_t; NULL = default stream), so they order with the caller's
 * own work on that stream (e.g. the PyTorch-ROCm network forward).
 *
 * Encodings shared by every call:
 *   board   int8[90], row-major 10x9, codes +1..+7 red K,A,B,N,R,C,P / -1..-7 black
 *           (config.py:66-74; row 0 = black back rank — chess_env.py:33-60)
 *   move    uint16 = (from_row*9+from_col)*90 + to_row*9+to_col — the same index the reference
 *           uses into its 8100 policy logits (neural_network.py:160)
 *   king    cached square row*9+col or -1 for None (chess_env.py:27-28)
 *   winner  1, -1, 0, or XQ_WINNER_NONE (Python None)
 */
#ifndef XQ_SELFPLAY_H
#define XQ_SELFPLAY_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XQ_MAX_MOVES    128
#define XQ_MAX_PLIES    70      /* literal 70 of chess_env.py:400 */
#define XQ_WINNER_NONE  2
#define XQ_NO_KING      (-1)
#define XQ_POLICY_SIZE  8100

#define XQ_E_INVALID    (-1)    /* bad argument                                   */
#define XQ_E_HIP        (-2)    /* HIP runtime error                              */
#define XQ_E_NOGPU      (-3)    /* no usable gfx950 device                        */
#define XQ_E_STATE      (-4)    /* call out of sequence                           */

/* end_reason codes (chess_env.py:297,359,366,373,381,389,397,404); the host formats the strings */
enum {
    XQ_R_NONE = 0, XQ_R_KING_CAPTURED = 1, XQ_R_CHECKMATE = 2, XQ_R_REPETITION = 3, XQ_R_FIFTY = 4,
    XQ_R_STALEMATE = 5, XQ_R_PERP_CHECK = 6, XQ_R_PERP_CHASE = 7, XQ_R_MOVE_CAP = 8
};

/* evaluator-output kinds accepted by the tree kernels */
enum {
    XQ_EVAL_PRIORS = 0,       /* a = float32 priors[G][128] in legal-move order, v = float64 values[G]   */
    XQ_EVAL_LOGITS_F32 = 1,   /* a = float32 logits[G][8100], v = float32 values[G]                      */
    XQ_EVAL_LOGITS_BF16 = 2   /* a = bf16 logits[G][8100],    v = bf16 values[G]                         */
};

/* layouts of the network-input planes the search kernel writes (neural_network.py:128-146) */
enum {
    XQ_PLANES_NONE = 0,
    XQ_PLANES_NCHW_F32 = 1,   /* float32 [G][15][10][9]  — exactly encode_board()                       */
    XQ_PLANES_NCHW_BF16 = 2,  /* bf16    [G][15][10][9]                                                  */
    XQ_PLANES_NHWC16_BF16 = 3 /* bf16    [G][10][9][16]  — channels-last, channel 15 = 0 padding        */
};

const char *xq_last_error(void);
int xq_device_count(void);
/* 1 when the library was built for the visible device's architecture (gfx950) */
int xq_device_ok(int device);

/* ------------------------------------------------------------------------------------------
 * Rules entry points on host buffers (batch of n independent envs).  They back the host mirror
 * of class ChineseChess (chess_env.py:9) one call at a time and the parity tests in bulk.
 * ------------------------------------------------------------------------------------------ */

/* ChineseChess.get_legal_moves (chess_env.py:76-121).  moves_host[n][128], counts_host[n]. */
int xq_rules_legal_moves(int n, const int8_t *boards_host, const int32_t *player_host,
                         const int32_t *red_king_host, const int32_t *black_king_host,
                         uint16_t *moves_host, int32_t *counts_host);

/* _is_in_check(1), _is_in_check(-1), _are_kings_facing (chess_env.py:506-548, 466-495) as seen
 * with self.current_player == player_host[i].  Each out array int32[n]. */
int xq_rules_query(int n, const int8_t *boards_host, const int32_t *player_host,
                   const int32_t *red_king_host, const int32_t *black_king_host,
                   int32_t *in_check_red_host, int32_t *in_check_black_host, int32_t *facing_host);

/* _get_threatened_pieces(side) (chess_env.py:550-596; called twice per make_move, :262 and :344, its second result
 * appended to chase_history :345, which nothing reads :674): the (attacker square, victim square) pairs `side`
 * threatens on the given boards with the given king caches, as from*90+to in the reference's scan order.
 * pairs_host [n][XQ_MAX_MOVES], counts_host [n].  Not on the self-play path (dead output there); the host mirror
 * fills chase_history from it on request. */
int xq_rules_threatened_pieces(int n, const int8_t *boards_host /*[n][90]*/, const int32_t *side_host,
                               const int32_t *red_king_host, const int32_t *black_king_host,
                               uint16_t *pairs_host, int32_t *counts_host);

/* _get_position_hash (chess_env.py:497-504) of (board, current_player) as the 64-bit key xq_rules_make_move
 * appends to / compares with position histories: equal keys <=> equal (board, player byte).  The reference's
 * values are salted Python hashes; only equality is part of the contract. */
int xq_rules_position_key(int n, const int8_t *boards_host /*[n][90]*/, const int32_t *player_host, uint64_t *keys_host);

/* ChineseChess.make_move (chess_env.py:253-406).
 * state_host: int32[n][XQ_STATE_WORDS], updated in place; boards_host updated in place.
 * Histories: pos_hist_host uint64[n][hist_stride] holds n_hist[i] keys previously returned in
 * key_out_host (position_history), check_hist_host uint8[n][hist_stride] holds n_check[i] flags
 * (check_history); the call does NOT append — it returns the new entry for the caller to append.
 * next_moves_host/next_count_host receive the legal moves of the new side to move (valid when the
 * game did not end by king capture). */
#define XQ_STATE_WORDS 10
enum { XQ_S_PLAYER = 0, XQ_S_MOVE_COUNT = 1, XQ_S_WINNER = 2, XQ_S_RED_KING = 3, XQ_S_BLACK_KING = 4,
       XQ_S_NO_CAPTURE = 5, XQ_S_CONSEC_CHECKS = 6, XQ_S_REASON = 7, XQ_S_REASON_SIDE = 8,
       XQ_S_REASON_COUNT = 9 };
int xq_rules_make_move(int n, int8_t *boards_host, int32_t *state_host, const int32_t *move_host,
                       const uint64_t *pos_hist_host, const int32_t *n_hist_host,
                       const uint8_t *check_hist_host, const int32_t *n_check_host, int hist_stride,
                       double *reward_host, int32_t *done_host, int32_t *is_check_host,
                       uint64_t *key_out_host, uint16_t *next_moves_host, int32_t *next_count_host);

/* ------------------------------------------------------------------------------------------
 * Engine: G concurrent games, one wavefront per game, state resident in HBM.
 * Replaces MCTS.search (self_play.py:89-154), self_play_game (self_play.py:178-312) and the
 * process pool of parallel_self_play (self_play.py:368-469).
 * ------------------------------------------------------------------------------------------ */
typedef struct xq_engine xq_engine;

typedef struct {
    int32_t n_games;          /* G                                                             */
    int32_t sims;             /* MCTS simulations per move (config.py:10)                      */
    int32_t leaf_batch;       /* 8 (self_play.py:101)                                          */
    int32_t max_moves;        /* config.MAX_MOVES = 70 (config.py:9)                           */
    double  temperature;      /* self_play_game(temperature=...)                               */
    int32_t opponent_mode;    /* 1: only red plies are stored (self_play.py:234)               */
    int32_t planes_format;    /* XQ_PLANES_*                                                   */
    int32_t device;           /* HIP device ordinal                                            */
    int32_t reserved;
} xq_config;

int  xq_engine_create(const xq_config *cfg, xq_engine **out);
void xq_engine_destroy(xq_engine *e);
int  xq_engine_set_stream(xq_engine *e, void *hip_stream);

/* counts ** (1/temperature) as the HOST's numpy evaluates it, for counts 0..n-1 (self_play.py:230);
 * NULL/0 = identity (temperature 1.0). */
int  xq_engine_set_pow_table(xq_engine *e, const double *table_host, int n);

/* Start G fresh games (ChineseChess.reset, chess_env.py:14-67).  seeds_host[g] seeds game g's
 * private MT19937 stream exactly as np.random.seed(seed) would (self_play.py:242 draws one
 * double per ply from it). */
int  xq_engine_new_games(xq_engine *e, const uint32_t *seeds_host);

/* Optional compact policy layout for XQ_EVAL_LOGITS_*: eval_a rows hold n_columns logits and
 * map_host[move] (int16[8100]) gives the column of a move, -1 for moves that can never be legal
 * (n_columns is the row stride: it may exceed 8100 when rows are padded, e.g. to xq_policy_fc_bf16's
 * multiple of 192).  NULL restores the reference layout (8100 columns, column = move index;
 * neural_network.py:160). */
int  xq_engine_set_logit_columns(xq_engine *e, const int16_t *map_host, int n_columns);

/* ---- opt-in search extensions with NO counterpart in the reference (BASELINE config C5) ----
 * per-ply temperature (call between plies; table as in xq_engine_set_pow_table) */
int  xq_engine_set_temperature(xq_engine *e, double temperature, const double *table_host, int n);
/* Dirichlet root noise: root priors become (1-eps) P + eps Dir(alpha); eps = 0 (default) disables */
int  xq_engine_set_root_noise(xq_engine *e, double alpha, double epsilon, uint64_t seed);
/* Tree reuse: the subtree of the played move becomes the next ply's tree (visit counts, values and
 * priors carried over; the reference builds a fresh tree every ply, self_play.py:98).  0 (default)
 * disables.  Re-sizes the node arena to a whole game's expansions; call before new_games/set_roots. */
int  xq_engine_set_tree_reuse(xq_engine *e, int enable);
/* Virtual loss: a round's simulations count as pending losses along their paths and spread over up
 * to leaf_batch distinct leaves (the reference's rounds all reach one leaf, Appendix A10).  The
 * evaluator then sees leaf_batch rows per game: row = game * leaf_batch + slot, for the input planes,
 * the logits / values it returns and xq_engine_priors_ptr / values_ptr; unused slots are ignored.
 * Not available to the host-side read_leaves / write_priors path.  Call before new_games/set_roots. */
int  xq_engine_set_virtual_loss(xq_engine *e, int enable);
int  xq_engine_leaf_slots(xq_engine *e);          /* evaluator rows per game: 1, or leaf_batch */
/* per game: nodes in its arena after the last search (1 + the children of every expanded node; the arena of a
 * fresh tree holds 1 + rounds * 128 at most) and the pending virtual-loss visits left in it (0 after a search) */
int  xq_engine_tree_stats(xq_engine *e, int32_t *n_nodes_host /*[G]*/, int32_t *pending_visits_host /*[G]*/);
/* priors stored on the root's children after the root was expanded, [G][128] */
int  xq_engine_read_root_priors(xq_engine *e, float *priors_host);

/* Replace the per-game uniform streams: uniforms_host[g][ply] is the double np.random.choice
 * consumes at that ply (self_play.py:242).  Lets the host mirror draw from NumPy's GLOBAL stream
 * exactly as the reference does (Appendix A14).  Call after xq_engine_new_games. */
int  xq_engine_set_uniforms(xq_engine *e, const double *uniforms_host /*[G][70]*/);

/* Replace the root states (MCTS.search on caller-provided envs; _copy_env semantics of
 * self_play.py:156-175: board, player, move_count, winner, king caches, no_capture_count). */
int  xq_engine_set_roots(xq_engine *e, const int8_t *boards_host, const int32_t *state_host /*[G][XQ_STATE_WORDS]*/);

/* One leaf-batch round r (0-based) of MCTS.search for every game (self_play.py:103-148):
 *  - if r > 0, first consumes the evaluator output of round r-1 (expand + backups),
 *  - runs the round's simulations (select / make_move / terminal backups) on the frozen tree,
 *  - leaves at most one pending leaf per game: planes, packed board, legal moves, multiplicity.
 * eval_a_dev / eval_v_dev per eval_kind; ignored for r == 0.  planes_dev may be NULL. */
int  xq_engine_search_round(xq_engine *e, int round, int eval_kind, const void *eval_a_dev,
                            const void *eval_v_dev, void *planes_dev);

/* Built-in exact evaluator ("HashNet") on the pending leaves; fills the
 * engine's own priors/values buffers (use XQ_EVAL_PRIORS with xq_engine_priors_ptr/values_ptr). */
int  xq_engine_eval_hashnet(xq_engine *e, int salt);

/* Consume the last round's evaluator output; afterwards root visit counts are final
 * (the return value of MCTS.search, self_play.py:151-154). */
int  xq_engine_end_search(xq_engine *e, int eval_kind, const void *eval_a_dev, const void *eval_v_dev);

/* One ply of self_play_game after the search (self_play.py:219-256): visit counts -> pi,
 * sample record, np.random.choice, make_move, next root.  Call after xq_engine_end_search. */
int  xq_engine_play_move(xq_engine *e);

/* z assignment (self_play.py:259-310) for all games. */
int  xq_engine_finalize(xq_engine *e);

/* number of games still running (synchronises the stream) */
int  xq_engine_active_games(xq_engine *e, int32_t *n_active_host);
/* The same count without blocking (the loop of self_play.py:203 polls `done` every ply; a device-side engine should not stop
 * for it): _post enqueues the count behind everything enqueued so far, _poll sets *ready to 1 and *n_active_host to the
 * newest posted count once it has arrived, else *ready = 0. */
int  xq_engine_active_games_post(xq_engine *e);
int  xq_engine_active_games_poll(xq_engine *e, int32_t *n_active_host, int32_t *ready);

/* device pointers of engine-owned buffers (valid for the engine's lifetime) */
void *xq_engine_priors_ptr(xq_engine *e);      /* float32 [G][128]  */
void *xq_engine_values_ptr(xq_engine *e);      /* float64 [G]       */
int   xq_engine_rounds_per_move(xq_engine *e); /* ceil(sims / leaf_batch) */

/* ---- read-back (each synchronises the stream) ---- */
/* pending leaves after xq_engine_search_round: rows for predict_batch (self_play.py:139-143) */
int  xq_engine_read_leaves(xq_engine *e, int8_t *boards_host /*[G][90]*/, int32_t *player_host,
                           uint16_t *moves_host /*[G][128]*/, int32_t *n_moves_host,
                           int32_t *mult_host /* 0 = no pending leaf */);
int  xq_engine_write_priors(xq_engine *e, const float *priors_host /*[G][128]*/, const double *values_host);
/* root children after xq_engine_end_search */
int  xq_engine_read_root_visits(xq_engine *e, uint16_t *moves_host /*[G][128]*/, int32_t *visits_host,
                                int32_t *n_child_host);
/* The search tree of one game as it stands - after xq_engine_end_search: the finished tree of the ply - node by node in
 * creation order: the fields of the reference's MCTSNode (/root/reference/self_play.py:19-28: move, visit_count,
 * value_sum, prior_prob; a node's children are the nodes [first_child, first_child + n_child) in legal-move order, its
 * parent the node that lists it).  *root_host = the root's index (0 unless tree reuse kept a subtree).  Copies
 * min(cap, *n_nodes_host) entries; any of the array pointers may be NULL. */
int  xq_engine_read_tree(xq_engine *e, int game, int cap, int32_t *n_nodes_host, int32_t *root_host, uint32_t *visit_count_host,
                         double *value_sum_host, float *prior_host, uint16_t *move_host, uint16_t *first_child_host,
                         uint8_t *n_child_host);
/* per-game outcome: winner (0 for None, self_play.py:259), reason code/side/count, plies, samples, error
 * (error 1 = np.random.choice would raise ValueError: NaN probabilities, sims <= leaf_batch) */
int  xq_engine_read_games(xq_engine *e, int32_t *winner_host, int32_t *reason_host, int32_t *reason_side_host,
                          int32_t *reason_count_host, int32_t *n_plies_host, int32_t *n_samples_host,
                          int32_t *error_host);
/* per-sample records, [G][70] leading dims; counts = root visit counts in moves order
 * (pi = counts**(1/T) / sum is formed by the host mirror with numpy, as the reference does) */
int  xq_engine_read_samples(xq_engine *e, int8_t *boards_host /*[G][70][90]*/, int8_t *player_host /*[G][70]*/,
                            uint8_t *n_moves_host /*[G][70]*/, uint16_t *moves_host /*[G][70][128]*/,
                            uint16_t *counts_host /*[G][70][128]*/, double *z_host /*[G][70]*/,
                            uint16_t *chosen_host /*[G][70]*/, double *step_reward_host /*[G][70]*/);
/* device-resident fixed-size sample records for the RCCL all-gather (one per [game][ply]),
 * record layout: xq_sample_record below.  Returns the device pointer and byte size. */
typedef struct {
    uint32_t board[12];       /* nibble-packed board (4 bit / square)                     */
    double   z;
    int8_t   player;
    uint8_t  n_moves;
    uint8_t  valid;
    uint8_t  pad;
    uint16_t chosen;
    uint16_t pad2;
    uint16_t moves[XQ_MAX_MOVES];
    uint16_t counts[XQ_MAX_MOVES];
} xq_sample_record;
int  xq_engine_pack_samples(xq_engine *e, void *records_dev /* xq_sample_record[G][70] */);

/* Result-identical work elimination (off until asked for; the reference has no counterpart: it rebuilds its tree
 * every ply, self_play.py:98, and so evaluates every new root a second time).  With it on, xq_engine_play_move
 * carries the played child's network evaluation over as the next root's and leaves the tree as round 0 would
 * (root.visit_count = first batch size, children unvisited); round 0 of the next ply then has nothing to evaluate
 * for that game.  With row compaction (below) the empty round costs no network time by itself; without it the
 * caller may skip round 0 - tree kernel and network forward - of a ply whenever xq_engine_roots_not_ready reports
 * 0 games needing it (fresh games always need it).  Games, visit counts, pi and z are bit-identical to the default
 * path with a deterministic, row-independent evaluator.  Not combinable with tree reuse, virtual loss, root noise
 * or opponent mode; call before xq_engine_new_games. */
int  xq_engine_set_root_eval_carry(xq_engine *e, int enable);
int  xq_engine_roots_not_ready(xq_engine *e, int32_t *n_host);

/* Evaluator row compaction (off until asked for: row = slot, every slot is evaluated, what the reference does with
 * its per-game batches, self_play.py:139-143).  When on, every xq_engine_search_round is followed by a numbering of
 * the slots that hold a pending leaf, in slot order: the evaluator reads the planes of row r at slot row_src[r],
 * writes logits / value to row r and stops at *row_count rows; consume reads a leaf's outputs from its row.
 * xq_engine_row_map hands out the two DEVICE pointers (NULL while compaction is off); xq_tower_nhwc_bf16,
 * xq_policy_fc_bf16 and xq_value_head_bf16 take them.  Games that are over, rounds that ended on terminal leaves,
 * roots carried over (above) and empty virtual-loss slots then cost no network time, with no host round trip.
 * Results do not depend on the numbering (every evaluator kernel is row-independent); evaluators that fill
 * xq_engine_priors_ptr (XQ_EVAL_PRIORS) stay indexed by slot.
 * While compaction is on, xq_engine_search_round / xq_engine_end_search refuse XQ_EVAL_LOGITS_* output from a
 * caller that has not fetched xq_engine_row_map since compaction was last switched (logits laid out by slot would
 * be read by row: wrong priors without an error otherwise).
 * xq_engine_read_row_history: rows of the last min(cap, n_rounds, XQ_ROW_HISTORY) search rounds, oldest first, into
 * rows_host[0 ..) - the engine keeps the last XQ_ROW_HISTORY rounds only - and the rounds launched since the count
 * was last reset.  xq_engine_read_leaf_rows: row of every slot (-1 = no pending leaf). */
#define XQ_ROW_HISTORY 65536
int  xq_engine_set_row_compaction(xq_engine *e, int enable);
int  xq_engine_row_map(xq_engine *e, const int32_t **row_src_dev, const int32_t **row_count_dev);
int  xq_engine_read_row_history(xq_engine *e, int32_t *rows_host, int cap, int64_t *n_rounds, int reset);
int  xq_engine_read_leaf_rows(xq_engine *e, int32_t *rows_host /* [G * leaf_slots] */);

/* Leaf dedupe (off until asked for; needs row compaction): pending leaves of one round that are the same position -
 * same board and same side to move, which is everything encode_board reads (neural_network.py:128-146) - share ONE
 * evaluator row; every slot of the group reads its logits and value from that row (the row of the group's lowest
 * slot; positions are compared in full, no hash is trusted).  The reference has no counterpart: each worker
 * evaluates every leaf of its own game (self_play.py:137-143), although all games of an epoch start from the same
 * position and, with the same weights, share their first plies with many others.  Result-identical whenever the
 * evaluator's output for a position depends on nothing but the position (not on the row, the batch or the launch);
 * the caller vouches for that by switching it on (true of xq_tower_nhwc_bf16 + xq_policy_fc_bf16 +
 * xq_value_head_bf16).  Row numbering stays deterministic.  Returns XQ_E_INVALID without row compaction;
 * switching compaction off switches this off. */
int  xq_engine_set_leaf_dedupe(xq_engine *e, int enable);

/* Evaluation cache (off until asked for): the evaluator's answer for a position - the priors of its legal moves and its
 * value - is kept in a table in HBM (2^log2_entries entries of 576 B, 10 <= log2_entries <= 24; 0 switches it off) for
 * two plies, and a later pending leaf that is the same position (board and side to move compared in full) takes it from
 * there: no evaluator row.  The reference has no counterpart: it rebuilds its tree every ply (self_play.py:98) and
 * evaluates again what the ply before expanded below the move that was played, and its workers share nothing
 * (self_play.py:137-143).  Result-identical whenever the evaluator's output for a position depends on nothing else -
 * the caller vouches for that, as for the leaf dedupe (true of xq_tower_nhwc_bf16 + xq_policy_fc_bf16 +
 * xq_value_head_bf16: one fixed fp32 chain per output element whatever the row and the launch).  An entry is used from
 * the launch after the one that filled it; xq_engine_new_games / set_roots / refill_begin age every entry out (new
 * weights must come with one of them).  Works with XQ_EVAL_LOGITS_* output; priors handed in by slot are not cached.
 * A negative log2_entries switches the cache to VERIFY mode with 2^-log2_entries entries: a leaf the cache could answer
 * is evaluated all the same and its priors, value and move count are compared with the entry's (a self-check of the
 * "depends on the position only" promise).  xq_engine_eval_cache_stats: leaves answered (verify mode: compared), entries
 * filled, and mismatches found in verify mode, since the last reset. */
int  xq_engine_set_eval_cache(xq_engine *e, int log2_entries);
int  xq_engine_eval_cache_stats(xq_engine *e, uint64_t *hits_fills_mismatches_host /* [3] */, int reset);

/* ---- refill: `total` games through the engine's G concurrent slots, a finished game's slot being restarted
 * on the next unplayed game at once — what the reference's pool does by construction (imap_unordered hands a
 * worker its next game as soon as one ends, self_play.py:404-408).  A game's result depends only on
 * seeds_host[id]: it is the game xq_engine_new_games would play for that seed, whatever slot and ply it
 * starts at.  Protocol: refill_begin; then per ply { search rounds as usual; xq_engine_play_move;
 * xq_engine_refill_step } until *active_host reads 0.  refill_step retires every finished game into
 * records_dev[id][0..69] (z table applied, self_play.py:259-310) and its outcome into the table
 * refill_read_games returns (same fields as xq_engine_read_games, indexed by game id 0..total-1), and
 * restarts the slot.  active_host may be NULL (no host synchronisation in that step).  total >= G; not
 * available in opponent mode (slots are at different plies). */
int  xq_engine_refill_begin(xq_engine *e, const uint32_t *seeds_host /*[total]*/, int total);
int  xq_engine_refill_step(xq_engine *e, void *records_dev /* xq_sample_record[total][70] */, int32_t *active_host);
int  xq_engine_refill_read_games(xq_engine *e, int32_t *winner, int32_t *reason, int32_t *reason_side,
                                 int32_t *reason_count, int32_t *n_plies, int32_t *n_samples, int32_t *error);
/* game id every slot holds right now (-1: the slot has retired, no unplayed game was left for it).  Which of two
 * slots that finish in the same step gets which id is not specified (an atomic counter deals them); a game's
 * result does not depend on it. */
int  xq_engine_refill_read_slots(xq_engine *e, int32_t *slot_game_host /*[G]*/);

/* ---- network: fused 3x3 convolution of the residual tower (neural_network.py:54,181-187 with the
 * eval-mode BatchNorm folded into weights and bias):
 *     y = relu?( conv3x3(x, w) + bias [+ residual] )
 * x [n_boards][10][9][c_in] bf16 (c_in = 16 or 128), w [9 taps][128 out][c_in] bf16,
 * bias float32[128], residual / y [n_boards][10][9][128] bf16; all device pointers; residual may
 * be NULL; y may alias residual but not x.  Launches on `hip_stream`. */
int  xq_conv3x3_nhwc_bf16(void *hip_stream, const void *x_dev, const void *w_dev, const void *bias_dev,
                          const void *residual_dev, void *y_dev, int n_boards, int c_in, int relu);

/* The two 1x1 head convolutions (policy 128->32, value 128->8; neural_network.py:34,42,61,66) with
 * folded BN and ReLU in one pass: x [n][10][9][128] bf16, w [64][128] bf16 (rows 0..31 policy,
 * 32..39 value, rest zero), bias float32[64] -> policy_out [n][90][32] bf16, value_out [n][90][8]
 * bf16 (the (h, w, c) order the permuted FC weights expect). */
int  xq_heads_nhwc_bf16(void *hip_stream, const void *x_dev, const void *w_dev, const void *bias_dev,
                        void *policy_out_dev, void *value_out_dev, int n_boards);

/* The whole convolutional trunk in one launch (neural_network.py:54-66,181-187, BN folded):
 * conv3x3(16->128)+ReLU, n_blocks residual blocks, both 1x1 heads + ReLU; activations never leave
 * LDS between layers.  planes [n][10][9][16] bf16, w1 [9][128][16] bf16, wt [2*n_blocks][9][128][128]
 * bf16, bias float32 [1 + 2*n_blocks][128] (conv1 first), wh / bh as in xq_heads_nhwc_bf16;
 * outputs as xq_heads_nhwc_bf16.  Same arithmetic as the per-layer calls (bf16 storage between
 * layers, fp32 accumulation in the same order), except that the skip connection is added in fp32
 * before the single bf16 rounding of a block's output (one rounding fewer per block).
 * row_src_dev / n_rows_dev (both optional, device int32: xq_engine_row_map): board b reads planes row row_src[b],
 * outputs go to row b, boards at or beyond *n_rows are skipped. */
int  xq_tower_nhwc_bf16(void *hip_stream, const void *planes_dev, const void *w1_dev, const void *wt_dev,
                        const void *bias_dev, const void *wh_dev, const void *bh_dev, void *policy_out_dev,
                        void *value_out_dev, int n_boards, int n_blocks, const void *row_src_dev, const void *n_rows_dev);

/* The policy head's fully-connected layer (neural_network.py:39,64: nn.Linear(32*10*9, 8100) applied to the
 * flattened policy-conv activations): logits[m][n] = bias[n] + sum_k act[m][k] * w[n][k].
 * act [n_rows][k] bf16 (the (h, w, c)-ordered output of xq_tower_nhwc_bf16 / xq_heads_nhwc_bf16), w [n_cols][k]
 * bf16 (nn.Linear weight layout, input columns permuted to (h, w, c)), bias float32[n_cols], logits
 * [n_rows][n_cols] bf16; all device pointers.  k % 64 == 0 and n_cols % 192 == 0 (pad w / bias with zero rows;
 * the caller may also DROP rows of w: the search gathers legal-move logits only, neural_network.py:148-169,
 * so a column no legal move can index is a dead output — see xq_engine_set_logit_columns).  fp32 accumulation
 * in a fixed order: results do not depend on the launch, the tile position or n_rows.
 * n_rows_dev (optional, device int32: xq_engine_row_map): only rows below it are computed. */
int  xq_policy_fc_bf16(void *hip_stream, const void *act_dev, const void *w_dev, const void *bias_dev, void *logits_dev,
                       int n_rows, int n_cols, int k, const void *n_rows_dev);

/* The value head behind its 1x1 convolution (neural_network.py:43-45,66-69): values[m] = tanh(fc2(relu(fc1(hv[m])))).
 * hv [n_rows][720] bf16 ((h, w, c) order, as xq_tower_nhwc_bf16 writes it) with at least 32 readable bytes behind
 * the last row; w1 [128][736] bf16 = value_fc1.weight with its input columns permuted to (h, w, c) and 16 zero
 * columns appended; b1, w2 float32[128]; b2 float32[1]; values bf16[n_rows].  The hidden layer stays in fp32.
 * n_rows_dev (optional, device int32): only rows below it are computed. */
int  xq_value_head_bf16(void *hip_stream, const void *hv_dev, const void *w1_dev, const void *b1_dev, const void *w2_dev,
                        const void *b2_dev, void *values_dev, int n_rows, const void *n_rows_dev);

/* ------------------------------------------------------------------------------------------
 * Replay buffer: device-resident mirror of trainer.py's ReplayBuffer
 * (trainer.py:22-44: deque(maxlen) of samples, push appends a game's samples in order) and of the
 * batch formation in Trainer.train_network (trainer.py:309-321: encode_board(board, 1) for every
 * sampled board, float32 reward as the value target).  Records are xq_sample_record.
 * ------------------------------------------------------------------------------------------ */
const char *xq_replay_last_error(void);
int     xq_replay_create(int device, int64_t capacity, void **out);
void    xq_replay_destroy(void *rb);
int64_t xq_replay_size(void *rb);                       /* len(buffer) */
/* push every valid record of records_dev [n_games][70] in game / ply order (drop-oldest) */
int     xq_replay_push_records(void *rb, void *hip_stream, const void *records_dev, int n_games, int64_t *n_pushed);
/* states float32 [batch][15][10][9], targets float32 [batch][1] for deque indices idx_host[] */
int     xq_replay_encode_batch(void *rb, void *hip_stream, const int64_t *idx_host, int batch, void *states_dev,
                               void *targets_dev);
/* compatibility view: the records at deque indices idx_host[] copied to host memory */
int     xq_replay_read_records(void *rb, void *hip_stream, const int64_t *idx_host, int batch, void *records_host);

/* ---- measurement: HIP events recorded on the engine's stream around the tree-kernel launches: enable = 0 off, 1 every
 * launch, N > 1 every N-th xq_engine_search_round (an event pair costs the GPU a few microseconds; every
 * xq_engine_play_move is timed either way).  _read: sums and counts of the launches that were timed ---- */
int  xq_engine_profile(xq_engine *e, int enable);
int  xq_engine_profile_read(xq_engine *e, double *search_ms_total, int64_t *search_launches,
                            double *play_ms_total, int64_t *play_launches);

#ifdef __cplusplus
}
#endif
#endif
