/*
 * xq_oracle.h — CPU restatement (plain C) of the reference self-play hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it, and only as the checker / reported
 * baseline.  The shipped path is the HIP library (chinesechessai_amd/csrc) and must
 * never call into this file.
 *
 * Parity status: PINNED.  Every function below follows the executed behaviour of
 * hpy666666/ChineseChessAI (file:line cited per function in xq_oracle.c) and is
 * checked bit-for-bit against golden vectors captured by running the unmodified
 * reference Python in the build container (oracle/gen_golden.py -> tests/golden/).
 */
#ifndef XQ_ORACLE_H
#define XQ_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XQO_MAX_MOVES   128      /* legal moves per position (max seen 68)          */
#define XQO_HIST_CAP    2048     /* plies of history kept per env                   */
#define XQO_WINNER_NONE 2        /* Python `None`                                   */
#define XQO_NO_KING     (-1)     /* king cache == None                              */

/* end_reason enum; the adapter formats the reference's f-strings from (code, side, count).
 * chess_env.py:297,359,366,373,381,389,397,404 */
enum {
    XQO_R_NONE = 0,
    XQO_R_KING_CAPTURED = 1,   /* "{mover}吃掉对方将帅"          side = mover            */
    XQO_R_CHECKMATE = 2,       /* "将死{loser}"                  side = side to move     */
    XQO_R_REPETITION = 3,      /* "三次重复局面判和"                                     */
    XQO_R_FIFTY = 4,           /* "50回合无吃子判和"                                     */
    XQO_R_STALEMATE = 5,       /* "困毙{loser}"                  side = side to move     */
    XQO_R_PERP_CHECK = 6,      /* "长将判负({side to move})"                             */
    XQO_R_PERP_CHASE = 7,      /* dead in the reference (chess_env.py:674)               */
    XQO_R_MOVE_CAP = 8         /* "超过{move_count}步判和"       count = move_count      */
};

typedef struct {
    int8_t  board[90];          /* row-major 10x9, codes +-1..7 (config.py:66-74)     */
    int32_t current_player;     /* +1 red, -1 black                                    */
    int32_t move_count;
    int32_t winner;             /* 1, -1, 0 or XQO_WINNER_NONE                         */
    int32_t end_reason, end_side, end_count;
    int32_t red_king, black_king;   /* cached square r*9+c or XQO_NO_KING             */
    int32_t no_capture_count;
    int32_t consecutive_checks;
    int32_t n_hist;                 /* len(position_history)                           */
    int32_t n_check;                /* len(check_history) (settable separately: tests) */
    uint8_t pos_hist[XQO_HIST_CAP][91];  /* board bytes + player byte, exact compare  */
    uint8_t check_hist[XQO_HIST_CAP];
} xqo_env;

/* Evaluator = the duck-typed `network.predict_batch` (neural_network.py:96-126).
 * rows may contain duplicates exactly as the reference passes them (self_play.py:139-143).
 * priors[row][j] is the prior of moves[row][j] as an np.float32; values[row] a Python float. */
typedef int (*xqo_eval_fn)(void *ctx, int nrows, const int8_t *boards /*[n][90]*/,
                           const int32_t *players, const uint16_t *moves /*[n][128]*/,
                           const int32_t *nmoves, float *priors /*[n][128]*/, double *values);

typedef struct {
    xqo_eval_fn fn;
    void *ctx;
} xqo_evaluator;

/* ---- rules ---- */
xqo_env *xqo_env_new(void);
void     xqo_env_free(xqo_env *e);
void     xqo_reset(xqo_env *e);
void     xqo_copy_min(xqo_env *dst, const xqo_env *src);   /* MCTS._copy_env */
int      xqo_legal_moves(xqo_env *e, uint16_t *out);        /* move = from*90+to */
int      xqo_is_in_check(xqo_env *e, int player);
int      xqo_are_kings_facing(const xqo_env *e);
int      xqo_is_move_suicide(xqo_env *e, int from, int to);
int      xqo_make_move(xqo_env *e, int move, double *reward, int *is_check);  /* returns done */
double   xqo_position_change(const xqo_env *e, int from, int to);
int      xqo_is_protected(xqo_env *e, int r, int c, int player);              /* chess_env.py:576-596 */
int      xqo_threatened_pieces(xqo_env *e, int player, uint16_t *out);        /* chess_env.py:550-574; from*90+to pairs */
int      xqo_check_checkmate(xqo_env *e);                                     /* chess_env.py:614-628 */
int      xqo_check_stalemate(xqo_env *e);                                     /* chess_env.py:630-644 */
int      xqo_check_draw_by_repetition(const xqo_env *e);                      /* chess_env.py:598-605 */

/* ---- search ---- */
/* returns number of root children; out_moves/out_visits in insertion (legal-move) order. */
/* the finished search tree, node by node in creation order (node 0 = root); arrays of `cap` entries, *n_nodes = nodes made */
typedef struct {
    int cap;
    int32_t *n_nodes;
    int32_t *parent; uint16_t *move; int32_t *visit_count; double *value_sum; float *prior;
    int32_t *first_child; int32_t *n_child;
} xqo_tree_dump;
int xqo_mcts_search_tree(const xqo_env *env, int sims, const xqo_evaluator *ev,
                         uint16_t *out_moves, int32_t *out_visits, const xqo_tree_dump *dump);
int xqo_mcts_search(const xqo_env *env, int sims, const xqo_evaluator *ev,
                    uint16_t *out_moves, int32_t *out_visits);
/* PUCT score exactly as NumPy>=2 evaluates self_play.py:51-52 (float32, stepwise). */
float xqo_puct_score(double value_sum, int visit, float prior, int parent_visit);

/* ---- sampling ---- */
void   xqo_mt_seed(uint32_t *mt624, int *idx, uint32_t seed);    /* np.random.seed(int)  */
double xqo_mt_double(uint32_t *mt624, int *idx);                  /* random_sample()      */
double xqo_np_sum(const double *a, int n);                        /* np.add.reduce f64    */
/* np.random.choice(n, p=p) given the uniform double already drawn. -1 = ValueError */
int    xqo_choice_from_uniform(const double *p, int n, double u);

/* ---- built-in exact evaluator (SURVEY Appendix B "HashNet") ---- */
uint32_t xqo_crc32(uint32_t crc, const uint8_t *buf, int len);
int xqo_hashnet_eval(void *ctx, int nrows, const int8_t *boards, const int32_t *players,
                     const uint16_t *moves, const int32_t *nmoves, float *priors, double *values);

/* ---- driver: self_play_game (self_play.py:178-312) ---- */
typedef struct {
    int32_t n_samples, n_plies, winner, end_reason, end_side, end_count, error;
    /* per stored sample (<= 70) */
    int8_t   s_board[70][90];
    int32_t  s_player[70];
    int32_t  s_nmoves[70];
    uint16_t s_moves[70][XQO_MAX_MOVES];
    double   s_probs[70][XQO_MAX_MOVES];
    double   s_z[70];
    /* per ply trace (for parity tests) */
    int32_t  t_move[70];
    int32_t  t_nchild[70];
    uint16_t t_moves[70][XQO_MAX_MOVES];
    int32_t  t_visits[70][XQO_MAX_MOVES];
    double   t_reward[70];
} xqo_game;

/* pow_table: optional table t[c] = float(c) ** (1.0/temperature) for c in [0, sims]
 * (NULL -> C pow()).  eval_black NULL -> self-play mode. */
int xqo_self_play_game(const xqo_evaluator *eval_red, const xqo_evaluator *eval_black,
                       double temperature, int sims, int max_moves, uint32_t seed,
                       const double *pow_table, xqo_game *out);

/* z table only (self_play.py:266-310) */
double xqo_z_value(int winner, int player, int game_length, int has_reward, double step_reward);

#ifdef __cplusplus
}
#endif
#endif
