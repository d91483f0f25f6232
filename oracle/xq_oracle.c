/*
 * xq_oracle.c — CPU restatement of the reference self-play hot path (rules -> MCTS -> driver).
 *
 * TEST INFRASTRUCTURE ONLY (see xq_oracle.h).  Parity: PINNED against golden vectors captured
 * from the unmodified reference (tests/golden/, made by oracle/gen_golden.py).
 *
 * Every function cites the reference lines it restates (paths relative to the reference repo).
 * The restatement is deliberately literal — including the behaviours listed in SURVEY.md
 * Appendix A (generators keyed on current_player, stale king caches, repetition key mismatch,
 * frozen tree inside a batch of 8 ...).  Do not "fix" anything here.
 *
 * Build: gcc -O2 -ffp-contract=off (no fast-math: fp64/fp32 results are part of the contract).
 */
#include "xq_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ROWS 10
#define COLS 9
enum { KING = 1, ADVISOR = 2, BISHOP = 3, KNIGHT = 4, ROOK = 5, CANNON = 6, PAWN = 7 };

static inline int on_board(int r, int c) { return r >= 0 && r < ROWS && c >= 0 && c < COLS; }

/* ------------------------------------------------------------------ reset / copy */

/* chess_env.py:14-67 */
void xqo_reset(xqo_env *e)
{
    static const int8_t back[9] = { ROOK, KNIGHT, BISHOP, ADVISOR, KING, ADVISOR, BISHOP, KNIGHT, ROOK };
    memset(e->board, 0, sizeof e->board);
    for (int c = 0; c < 9; c++) {
        e->board[9 * 9 + c] = back[c];
        e->board[0 * 9 + c] = (int8_t)-back[c];
    }
    e->board[7 * 9 + 1] = e->board[7 * 9 + 7] = CANNON;
    e->board[2 * 9 + 1] = e->board[2 * 9 + 7] = -CANNON;
    for (int c = 0; c < 9; c += 2) {
        e->board[6 * 9 + c] = PAWN;
        e->board[3 * 9 + c] = -PAWN;
    }
    e->red_king = 9 * 9 + 4;
    e->black_king = 0 * 9 + 4;
    e->current_player = 1;
    e->move_count = 0;
    e->winner = XQO_WINNER_NONE;
    e->end_reason = XQO_R_NONE;
    e->end_side = 0;
    e->end_count = 0;
    e->no_capture_count = 0;
    e->consecutive_checks = 0;
    e->n_hist = 0;
    e->n_check = 0;
}

xqo_env *xqo_env_new(void)
{
    xqo_env *e = (xqo_env *)malloc(sizeof(xqo_env));
    if (e) xqo_reset(e);
    return e;
}

void xqo_env_free(xqo_env *e) { free(e); }

/* self_play.py:156-175 — new env (reset), then overwrite board, player, move_count, winner,
 * king caches, no_capture_count.  Histories stay empty; consecutive_checks stays 0;
 * end_reason stays None. */
void xqo_copy_min(xqo_env *dst, const xqo_env *src)
{
    memcpy(dst->board, src->board, 90);
    dst->current_player = src->current_player;
    dst->move_count = src->move_count;
    dst->winner = src->winner;
    dst->red_king = src->red_king;
    dst->black_king = src->black_king;
    dst->no_capture_count = src->no_capture_count;
    dst->consecutive_checks = 0;
    dst->end_reason = XQO_R_NONE;
    dst->end_side = 0;
    dst->end_count = 0;
    dst->n_hist = 0;
    dst->n_check = 0;
}

/* ------------------------------------------------------------------ pseudo-move generators
 * Each writes (r,c) targets in the reference's emission order, unfiltered (may be off-board).
 * King/advisor/bishop/pawn key on e->current_player, NOT on the piece colour (Appendix A1). */

typedef struct { int8_t r, c; } rc_t;

/* chess_env.py:123-138 */
static int gen_king(const xqo_env *e, int r, int c, rc_t *out)
{
    static const int d[4][2] = { {0, 1}, {0, -1}, {1, 0}, {-1, 0} };
    int lo = (e->current_player == 1) ? 7 : 0, hi = (e->current_player == 1) ? 10 : 3, n = 0;
    for (int i = 0; i < 4; i++) {
        int nr = r + d[i][0], nc = c + d[i][1];
        if (lo <= nr && nr < hi && 3 <= nc && nc < 6) { out[n].r = (int8_t)nr; out[n].c = (int8_t)nc; n++; }
    }
    return n;
}

/* chess_env.py:140-154 */
static int gen_advisor(const xqo_env *e, int r, int c, rc_t *out)
{
    static const int d[4][2] = { {1, 1}, {1, -1}, {-1, 1}, {-1, -1} };
    int lo = (e->current_player == 1) ? 7 : 0, hi = (e->current_player == 1) ? 10 : 3, n = 0;
    for (int i = 0; i < 4; i++) {
        int nr = r + d[i][0], nc = c + d[i][1];
        if (lo <= nr && nr < hi && 3 <= nc && nc < 6) { out[n].r = (int8_t)nr; out[n].c = (int8_t)nc; n++; }
    }
    return n;
}

/* chess_env.py:156-176 */
static int gen_bishop(const xqo_env *e, int r, int c, rc_t *out)
{
    static const int d[4][2] = { {2, 2}, {2, -2}, {-2, 2}, {-2, -2} };
    int river = (e->current_player == 1) ? 5 : 4, n = 0;
    for (int i = 0; i < 4; i++) {
        int nr = r + d[i][0], nc = c + d[i][1];
        if (!on_board(nr, nc)) continue;
        if (e->current_player == 1 && nr < river) continue;
        if (e->current_player == -1 && nr >= river) continue;
        int br = r + d[i][0] / 2, bc = c + d[i][1] / 2;
        if (e->board[br * 9 + bc] == 0) { out[n].r = (int8_t)nr; out[n].c = (int8_t)nc; n++; }
    }
    return n;
}

/* chess_env.py:178-197 — leg must be on board and empty; the target is NOT bounds-checked here */
static int gen_knight(const xqo_env *e, int r, int c, rc_t *out)
{
    static const int o[8][4] = { {2, 1, 1, 0}, {2, -1, 1, 0}, {-2, 1, -1, 0}, {-2, -1, -1, 0},
                                 {1, 2, 0, 1}, {-1, 2, 0, 1}, {1, -2, 0, -1}, {-1, -2, 0, -1} };
    int n = 0;
    for (int i = 0; i < 8; i++) {
        int br = r + o[i][2], bc = c + o[i][3];
        if (on_board(br, bc) && e->board[br * 9 + bc] == 0) {
            out[n].r = (int8_t)(r + o[i][0]); out[n].c = (int8_t)(c + o[i][1]); n++;
        }
    }
    return n;
}

/* chess_env.py:199-213 */
static int gen_rook(const xqo_env *e, int r, int c, rc_t *out)
{
    static const int d[4][2] = { {0, 1}, {0, -1}, {1, 0}, {-1, 0} };
    int n = 0;
    for (int i = 0; i < 4; i++)
        for (int step = 1; step < 10; step++) {
            int nr = r + d[i][0] * step, nc = c + d[i][1] * step;
            if (!on_board(nr, nc)) break;
            out[n].r = (int8_t)nr; out[n].c = (int8_t)nc; n++;
            if (e->board[nr * 9 + nc] != 0) break;
        }
    return n;
}

/* chess_env.py:215-235 */
static int gen_cannon(const xqo_env *e, int r, int c, rc_t *out)
{
    static const int d[4][2] = { {0, 1}, {0, -1}, {1, 0}, {-1, 0} };
    int n = 0;
    for (int i = 0; i < 4; i++) {
        int jumped = 0;
        for (int step = 1; step < 10; step++) {
            int nr = r + d[i][0] * step, nc = c + d[i][1] * step;
            if (!on_board(nr, nc)) break;
            if (e->board[nr * 9 + nc] == 0) {
                if (!jumped) { out[n].r = (int8_t)nr; out[n].c = (int8_t)nc; n++; }
            } else {
                if (!jumped) jumped = 1;
                else { out[n].r = (int8_t)nr; out[n].c = (int8_t)nc; n++; break; }
            }
        }
    }
    return n;
}

/* chess_env.py:237-251 */
static int gen_pawn(const xqo_env *e, int r, int c, rc_t *out)
{
    int n = 0;
    if (e->current_player == 1) {
        out[n].r = (int8_t)(r - 1); out[n].c = (int8_t)c; n++;
        if (r < 5) {
            out[n].r = (int8_t)r; out[n].c = (int8_t)(c - 1); n++;
            out[n].r = (int8_t)r; out[n].c = (int8_t)(c + 1); n++;
        }
    } else {
        out[n].r = (int8_t)(r + 1); out[n].c = (int8_t)c; n++;
        if (r >= 5) {
            out[n].r = (int8_t)r; out[n].c = (int8_t)(c - 1); n++;
            out[n].r = (int8_t)r; out[n].c = (int8_t)(c + 1); n++;
        }
    }
    return n;
}

static int gen_piece(const xqo_env *e, int r, int c, int type, rc_t *out)
{
    switch (type) {
    case KING:    return gen_king(e, r, c, out);
    case ADVISOR: return gen_advisor(e, r, c, out);
    case BISHOP:  return gen_bishop(e, r, c, out);
    case KNIGHT:  return gen_knight(e, r, c, out);
    case ROOK:    return gen_rook(e, r, c, out);
    case CANNON:  return gen_cannon(e, r, c, out);
    case PAWN:    return gen_pawn(e, r, c, out);
    default:      return 0;
    }
}

/* ------------------------------------------------------------------ check / facing / suicide */

/* chess_env.py:506-548 — regenerates every enemy piece's pseudo-moves with the CURRENT player's
 * generator rules and looks for the cached king square. */
int xqo_is_in_check(xqo_env *e, int player)
{
    int king = (player == 1) ? e->red_king : e->black_king;
    if (king == XQO_NO_KING) return 0;
    int kr = king / 9, kc = king % 9;
    rc_t mv[20];
    for (int r = 0; r < ROWS; r++)
        for (int c = 0; c < COLS; c++) {
            int piece = e->board[r * 9 + c];
            if (piece * player < 0) {
                int n = gen_piece(e, r, c, abs(piece), mv);
                for (int i = 0; i < n; i++)
                    if (mv[i].r == kr && mv[i].c == kc) return 1;
            }
        }
    return 0;
}

/* chess_env.py:466-495 — uses the caches only */
int xqo_are_kings_facing(const xqo_env *e)
{
    if (e->red_king == XQO_NO_KING || e->black_king == XQO_NO_KING) return 0;
    int rr = e->red_king / 9, rcol = e->red_king % 9, br = e->black_king / 9, bc = e->black_king % 9;
    if (rcol != bc) return 0;
    int lo = rr < br ? rr : br, hi = rr < br ? br : rr;
    for (int r = lo + 1; r < hi; r++)
        if (e->board[r * 9 + rcol] != 0) return 0;
    return 1;
}

/* chess_env.py:431-464 — the king cache follows only a MOVING king (Appendix A5) */
int xqo_is_move_suicide(xqo_env *e, int from, int to)
{
    int8_t backup[90];
    memcpy(backup, e->board, 90);
    int brk = e->red_king, bbk = e->black_king;
    int8_t moving = e->board[from];
    e->board[to] = moving;
    e->board[from] = 0;
    if (moving == KING) e->red_king = to;
    else if (moving == -KING) e->black_king = to;
    int in_check = xqo_is_in_check(e, e->current_player);
    int facing = xqo_are_kings_facing(e);
    memcpy(e->board, backup, 90);
    e->red_king = brk;
    e->black_king = bbk;
    return in_check || facing;
}

/* chess_env.py:76-88 + 90-121: squares row-major; per piece the generator order; filters
 * in-bounds, not-own, not-suicide keep the order. */
int xqo_legal_moves(xqo_env *e, uint16_t *out)
{
    int n = 0;
    rc_t mv[20];
    for (int r = 0; r < ROWS; r++)
        for (int c = 0; c < COLS; c++) {
            int piece = e->board[r * 9 + c];
            if (piece * e->current_player > 0) {
                int k = gen_piece(e, r, c, abs(piece), mv);
                for (int i = 0; i < k; i++) {
                    int tr = mv[i].r, tc = mv[i].c;
                    if (!on_board(tr, tc)) continue;
                    int target = e->board[tr * 9 + tc];
                    if (target * e->current_player <= 0)
                        if (!xqo_is_move_suicide(e, r * 9 + c, tr * 9 + tc)) {
                            if (n < XQO_MAX_MOVES) out[n] = (uint16_t)((r * 9 + c) * 90 + tr * 9 + tc);
                            n++;
                        }
                }
            }
        }
    return n;
}

/* chess_env.py:90-121 for one piece: generator order, in-bounds / not-own / not-suicide filters.
 * out[i] = from*90+to. */
static int piece_moves(xqo_env *e, int r, int c, int piece, uint16_t *out)
{
    rc_t mv[20];
    int n = 0, k = gen_piece(e, r, c, abs(piece), mv);
    for (int i = 0; i < k; i++) {
        int tr = mv[i].r, tc = mv[i].c;
        if (!on_board(tr, tc)) continue;
        int target = e->board[tr * 9 + tc];
        if (target * e->current_player <= 0)
            if (!xqo_is_move_suicide(e, r * 9 + c, tr * 9 + tc))
                out[n++] = (uint16_t)((r * 9 + c) * 90 + tr * 9 + tc);
    }
    return n;
}

/* chess_env.py:576-596 — is the piece of `player` on (r, c) protected?  Scans player's pieces with
 * current_player temporarily = player and looks for a move ONTO (r, c) in _get_piece_moves' output.
 * (That output never contains a square held by the mover's own side — the :116 filter — so in the
 * executed reference this is always False; restated literally, not simplified.) */
int xqo_is_protected(xqo_env *e, int r, int c, int player)
{
    uint16_t mv[20];
    for (int pr = 0; pr < ROWS; pr++)
        for (int pc = 0; pc < COLS; pc++) {
            int piece = e->board[pr * 9 + pc];
            if (piece * player > 0) {
                int original = e->current_player;
                e->current_player = player;
                int n = piece_moves(e, pr, pc, piece, mv);
                e->current_player = original;
                for (int i = 0; i < n; i++)
                    if (mv[i] % 90 == r * 9 + c) return 1;
            }
        }
    return 0;
}

/* chess_env.py:550-574 — (attacker square, victim square) pairs `player` threatens: every move of a piece
 * of `player` (generated with current_player temporarily = player) that captures an enemy piece other
 * than the king and whose victim is not protected.  out[i] = from*90+to, in scan order.  Called twice per
 * make_move (:262, :344); only the second result is kept (chase_history, :345) and nothing reads it (:674). */
int xqo_threatened_pieces(xqo_env *e, int player, uint16_t *out)
{
    uint16_t mv[20];
    int n_out = 0;
    for (int r = 0; r < ROWS; r++)
        for (int c = 0; c < COLS; c++) {
            int piece = e->board[r * 9 + c];
            if (piece * player > 0) {
                int original = e->current_player;
                e->current_player = player;
                int n = piece_moves(e, r, c, piece, mv);
                e->current_player = original;
                for (int i = 0; i < n; i++) {
                    int to = mv[i] % 90, target = e->board[to];
                    if (target * player < 0 && abs(target) != KING)
                        if (!xqo_is_protected(e, to / 9, to % 9, -player)) {
                            if (n_out < XQO_MAX_MOVES) out[n_out] = mv[i];
                            n_out++;
                        }
                }
            }
        }
    return n_out;
}

/* chess_env.py:614-628 / 630-644 */
int xqo_check_checkmate(xqo_env *e)
{
    uint16_t mv[XQO_MAX_MOVES];
    if (xqo_legal_moves(e, mv) > 0) return 0;
    return xqo_is_in_check(e, e->current_player) ? 1 : 0;
}

int xqo_check_stalemate(xqo_env *e)
{
    uint16_t mv[XQO_MAX_MOVES];
    if (xqo_legal_moves(e, mv) > 0) return 0;
    return xqo_is_in_check(e, e->current_player) ? 0 : 1;
}

/* ------------------------------------------------------------------ make_move */

/* chess_env.py:683-737 */
double xqo_position_change(const xqo_env *e, int from, int to)
{
    int fr = from / 9, fc = from % 9, tr = to / 9, tc = to % 9;
    int type = abs(e->board[to]);
    double score = 0;
    int advance = (e->current_player == 1) ? fr - tr : tr - fr;
    if (advance > 0) {
        if (type == PAWN) score += advance * 2.0;
        else if (type == ROOK || type == CANNON) score += advance * 1.5;
        else if (type == KNIGHT) score += advance * 1.0;
    }
    if (tc >= 3 && tc <= 5) {
        score += 1.5;
        if (tr >= 3 && tr <= 6) score += 1.0;
    }
    if (type == PAWN) {
        if (e->current_player == 1 && tr < 5) score += 3.0;
        else if (e->current_player == -1 && tr >= 5) score += 3.0;
    }
    int ok = (e->current_player == 1) ? e->black_king : e->red_king;
    if (ok != XQO_NO_KING) {
        int kr = ok / 9, kc = ok % 9;
        int od = abs(fr - kr) + abs(fc - kc), nd = abs(tr - kr) + abs(tc - kc);
        if (nd < od) score += (od - nd) * 0.5;
    }
    return score;
}

static void push_history(xqo_env *e, int is_check)
{
    /* chess_env.py:497-504: key = board bytes + (0 if current_player == 1 else 1), recorded
     * BEFORE the side switch (Appendix A7).  Python compares 64-bit hashes of these bytes;
     * the oracle compares the bytes. */
    if (e->n_hist < XQO_HIST_CAP) {
        memcpy(e->pos_hist[e->n_hist], e->board, 90);
        e->pos_hist[e->n_hist][90] = (uint8_t)(e->current_player == 1 ? 0 : 1);
    }
    e->n_hist++;
    if (e->n_check < XQO_HIST_CAP) e->check_hist[e->n_check] = (uint8_t)is_check;
    e->n_check++;
}

/* chess_env.py:598-605 */
int xqo_check_draw_by_repetition(const xqo_env *e);
static int check_repetition(const xqo_env *e) { return xqo_check_draw_by_repetition(e); }
int xqo_check_draw_by_repetition(const xqo_env *e)
{
    uint8_t key[91];
    memcpy(key, e->board, 90);
    key[90] = (uint8_t)(e->current_player == 1 ? 0 : 1);
    int count = 0, n = e->n_hist < XQO_HIST_CAP ? e->n_hist : XQO_HIST_CAP;
    for (int i = 0; i < n; i++)
        if (memcmp(e->pos_hist[i], key, 91) == 0) count++;
    return count >= 3;
}

/* chess_env.py:646-662 */
static int check_perpetual_check(const xqo_env *e)
{
    int n = e->n_check < XQO_HIST_CAP ? e->n_check : XQO_HIST_CAP;
    if (n < 12) return 0;
    int cnt = 0;
    for (int i = n - 12; i < n; i++) cnt += e->check_hist[i] ? 1 : 0;
    return cnt >= 10;
}

/* chess_env.py:253-406.  _get_threatened_pieces (262, 344) feeds only chase_history, which no
 * live code reads (674) — not restated.  Returns done; *reward as a Python number. */
int xqo_make_move(xqo_env *e, int move, double *reward_out, int *is_check_out)
{
    int from = move / 90, to = move % 90;
    int captured = e->board[to];
    int moving = e->board[from];
    e->board[to] = (int8_t)moving;
    e->board[from] = 0;

    if (moving == KING) e->red_king = to;                 /* :271-274 */
    else if (moving == -KING) e->black_king = to;
    if (captured == KING) e->red_king = XQO_NO_KING;      /* :276-279 */
    else if (captured == -KING) e->black_king = XQO_NO_KING;

    if (captured != 0) e->no_capture_count = 0;           /* :282-285 */
    else e->no_capture_count += 1;

    double reward = 0;
    int done = 0;
    if (abs(captured) == KING) {                          /* :292-297 */
        e->winner = e->current_player;
        reward = 100;
        done = 1;
        e->end_reason = XQO_R_KING_CAPTURED;
        e->end_side = e->current_player;
    } else if (captured != 0) {                           /* :300-314 */
        double base;
        switch (abs(captured)) {
        case ROOK: base = 9; break;
        case CANNON: base = 4.5; break;
        case KNIGHT: base = 4; break;
        case BISHOP: base = 2; break;
        case ADVISOR: base = 2; break;
        case PAWN: base = 1; break;
        default: base = 0; break;
        }
        reward = base * 2.0;
        if (abs(captured) == ADVISOR || abs(captured) == BISHOP) reward += 3.0;
    }

    int is_checking = xqo_is_in_check(e, -e->current_player);   /* :317 */
    if (!done && is_checking) {                                  /* :318-327 */
        if (e->consecutive_checks == 0) reward += 15.0;
        else if (e->consecutive_checks == 1) reward += 10.0;
        else if (e->consecutive_checks == 2) reward += 5.0;
        e->consecutive_checks += 1;
    } else {                                                     /* :328-335 */
        e->consecutive_checks = 0;
        if (captured == 0 && !done) {
            double pr = xqo_position_change(e, from, to);
            reward += pr * 0.01;
        }
    }

    push_history(e, is_checking);                                /* :338-345 */

    e->current_player *= -1;                                     /* :348-349 */
    e->move_count += 1;

    if (!done) {                                                 /* :352-397 */
        uint16_t tmp[XQO_MAX_MOVES];
        int nlegal = xqo_legal_moves(e, tmp);      /* _check_checkmate and _check_stalemate both
                                                      call get_legal_moves on the same state */
        int in_chk = (nlegal == 0) ? xqo_is_in_check(e, e->current_player) : 0;
        if (nlegal == 0 && in_chk) {                             /* :354-359 */
            done = 1; reward = 200;
            e->winner = -e->current_player;
            e->end_reason = XQO_R_CHECKMATE; e->end_side = e->current_player;
        } else if (check_repetition(e)) {                        /* :362-366 */
            done = 1; reward = 0; e->winner = 0;
            e->end_reason = XQO_R_REPETITION; e->end_side = 0;
        } else if (e->no_capture_count >= 100) {                 /* :369-373, 612 */
            done = 1; reward = 0; e->winner = 0;
            e->end_reason = XQO_R_FIFTY; e->end_side = 0;
        } else if (nlegal == 0 && !in_chk) {                     /* :376-381 */
            done = 1; reward = 100;
            e->winner = -e->current_player;
            e->end_reason = XQO_R_STALEMATE; e->end_side = e->current_player;
        } else if (check_perpetual_check(e)) {                   /* :384-389 */
            done = 1; reward = -10;
            e->winner = -e->current_player;
            e->end_reason = XQO_R_PERP_CHECK; e->end_side = e->current_player;
        }
        /* :392 _check_perpetual_chase() returns False unconditionally (:674) */
    }
    if (!done && e->move_count >= 70) {                          /* :400-404 (literal 70) */
        done = 1; reward = -2; e->winner = 0;
        e->end_reason = XQO_R_MOVE_CAP; e->end_side = 0; e->end_count = e->move_count;
    }
    if (reward_out) *reward_out = reward;
    if (is_check_out) *is_check_out = is_checking;
    return done;
}

/* ------------------------------------------------------------------ MCTS (self_play.py:19-154) */

typedef struct {
    int parent;          /* -1 root */
    int first_child;     /* index of first child, children contiguous in insertion order */
    int n_child;
    int visit_count;
    double value_sum;
    float prior;
    uint16_t move;
} node_t;

typedef struct {
    node_t *nodes;
    int n, cap;
} tree_t;

static int tree_new_node(tree_t *t, int parent, uint16_t move, float prior)
{
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nodes = (node_t *)realloc(t->nodes, (size_t)t->cap * sizeof(node_t));
    }
    node_t *nd = &t->nodes[t->n];
    nd->parent = parent; nd->first_child = -1; nd->n_child = 0;
    nd->visit_count = 0; nd->value_sum = 0; nd->prior = prior; nd->move = move;
    return t->n++;
}

/* self_play.py:51-52 under NumPy >= 2 (NEP 50): prior is np.float32, Python scalars are weak,
 * so every step is rounded to float32: f32(Q) + ((f32(1.5)*P) * f32(sqrt(N))) / f32(1+n). */
float xqo_puct_score(double value_sum, int visit, float prior, int parent_visit)
{
    volatile float q = (visit == 0) ? 0.0f : (float)(value_sum / (double)visit);   /* :30-34 */
    volatile float t = 1.5f * prior;
    t = t * (float)sqrt((double)parent_visit);
    t = t / (float)(1 + visit);
    volatile float s = q + t;
    return s;
}

/* self_play.py:40-59 — strict '>' keeps the FIRST maximum in insertion order */
static int select_child(const tree_t *t, int ni)
{
    const node_t *nd = &t->nodes[ni];
    int best = -1;
    float best_score = -INFINITY;
    for (int i = 0; i < nd->n_child; i++) {
        const node_t *ch = &t->nodes[nd->first_child + i];
        float s = xqo_puct_score(ch->value_sum, ch->visit_count, ch->prior, nd->visit_count);
        if (s > best_score) { best_score = s; best = nd->first_child + i; }
    }
    return best;
}

/* self_play.py:70-80 */
static void node_update(tree_t *t, int ni, double value)
{
    while (ni >= 0) {
        t->nodes[ni].visit_count += 1;
        t->nodes[ni].value_sum += value;
        value = -value;
        ni = t->nodes[ni].parent;
    }
}

/* self_play.py:89-154; `dump` (may be NULL): the finished tree, node by node in creation order (self_play.py:19-28) */
static int mcts_search_impl(const xqo_env *env, int sims, const xqo_evaluator *ev,
                            uint16_t *out_moves, int32_t *out_visits, const xqo_tree_dump *dump)
{
    tree_t t = { 0, 0, 0 };
    int root = tree_new_node(&t, -1, 0, 0.0f);
    xqo_env *se = (xqo_env *)malloc(sizeof(xqo_env));

    int leaf_nodes[8];
    static const int BATCH = 8;
    int8_t *l_boards = (int8_t *)malloc(8 * 90);
    int32_t l_players[8], l_nmoves[8];
    uint16_t *l_moves = (uint16_t *)malloc(8 * XQO_MAX_MOVES * sizeof(uint16_t));
    float *l_priors = (float *)malloc(8 * XQO_MAX_MOVES * sizeof(float));
    double l_values[8];
    int rc = 0;

    for (int batch_start = 0; batch_start < sims && rc == 0; batch_start += BATCH) {
        int batch_end = batch_start + BATCH < sims ? batch_start + BATCH : sims;
        int batch_count = batch_end - batch_start, n_leaf = 0;
        for (int s = 0; s < batch_count; s++) {
            int node = root;
            xqo_copy_min(se, env);
            while (t.nodes[node].n_child != 0) {                         /* :117-119 */
                node = select_child(&t, node);
                xqo_make_move(se, t.nodes[node].move, 0, 0);
            }
            int cp = se->current_player;
            uint16_t *lm = l_moves + n_leaf * XQO_MAX_MOVES;
            int nl = xqo_legal_moves(se, lm);                             /* :123 */
            if (nl == 0 || se->winner != XQO_WINNER_NONE) {               /* :126-135 */
                double value;
                if (se->winner == cp) value = 1;
                else if (se->winner == -cp) value = -1;
                else value = 0;
                node_update(&t, node, value);
            } else {                                                      /* :138-139 */
                leaf_nodes[n_leaf] = node;
                memcpy(l_boards + n_leaf * 90, se->board, 90);
                l_players[n_leaf] = cp;
                l_nmoves[n_leaf] = nl;
                n_leaf++;
            }
        }
        if (n_leaf > 0) {                                                 /* :142-148 */
            if (ev->fn(ev->ctx, n_leaf, l_boards, l_players, l_moves, l_nmoves, l_priors, l_values) != 0) {
                rc = -1;
                break;
            }
            for (int i = 0; i < n_leaf; i++) {
                int nd = leaf_nodes[i];
                if (t.nodes[nd].n_child == 0) {   /* expand adds only missing moves (:66-68): a
                                                     node queued several times is expanded once */
                    int first = t.n;
                    for (int j = 0; j < l_nmoves[i]; j++)
                        tree_new_node(&t, nd, l_moves[i * XQO_MAX_MOVES + j], l_priors[i * XQO_MAX_MOVES + j]);
                    t.nodes[nd].first_child = first;
                    t.nodes[nd].n_child = l_nmoves[i];
                }
                node_update(&t, nd, l_values[i]);
            }
        }
    }

    int n = 0;
    if (rc == 0) {                                                        /* :151-154 */
        n = t.nodes[root].n_child;
        for (int i = 0; i < n; i++) {
            out_moves[i] = t.nodes[t.nodes[root].first_child + i].move;
            out_visits[i] = t.nodes[t.nodes[root].first_child + i].visit_count;
        }
    } else n = rc;
    if (rc == 0 && dump) {
        *dump->n_nodes = t.n;
        for (int i = 0; i < t.n && i < dump->cap; i++) {
            const node_t *nd = &t.nodes[i];
            dump->parent[i] = nd->parent; dump->move[i] = (uint16_t)nd->move; dump->visit_count[i] = nd->visit_count;
            dump->value_sum[i] = nd->value_sum; dump->prior[i] = nd->prior;
            dump->first_child[i] = nd->first_child; dump->n_child[i] = nd->n_child;
        }
    }
    free(t.nodes); free(se); free(l_boards); free(l_moves); free(l_priors);
    return n;
}

int xqo_mcts_search(const xqo_env *env, int sims, const xqo_evaluator *ev,
                    uint16_t *out_moves, int32_t *out_visits)
{
    return mcts_search_impl(env, sims, ev, out_moves, out_visits, 0);
}

/* the same search, and every MCTSNode of its tree (self_play.py:19-28: parent, move, visit_count, value_sum,
 * prior_prob; children = [first_child, first_child + n_child) in legal-move order) */
int xqo_mcts_search_tree(const xqo_env *env, int sims, const xqo_evaluator *ev,
                         uint16_t *out_moves, int32_t *out_visits, const xqo_tree_dump *dump)
{
    return mcts_search_impl(env, sims, ev, out_moves, out_visits, dump);
}

/* ------------------------------------------------------------------ NumPy sampling restated */

/* np.random.seed(int) -> mt19937_seed (init_genrand) */
void xqo_mt_seed(uint32_t *mt, int *idx, uint32_t seed)
{
    mt[0] = seed;
    for (int i = 1; i < 624; i++)
        mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    *idx = 624;
}

static uint32_t mt_next(uint32_t *mt, int *idx)
{
    if (*idx >= 624) {
        int i;
        for (i = 0; i < 624 - 397; i++) {
            uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
            mt[i] = mt[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; i < 623; i++) {
            uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
            mt[i] = mt[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        *idx = 0;
    }
    uint32_t y = mt[(*idx)++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* RandomState.random_sample(): 53-bit double */
double xqo_mt_double(uint32_t *mt, int *idx)
{
    uint32_t a = mt_next(mt, idx) >> 5, b = mt_next(mt, idx) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* NumPy float64 add.reduce over a contiguous 1-D array = pairwise summation
 * (numpy/_core/src/umath/loops_utils.h.src, PW_BLOCKSIZE 128, 8 partial sums). */
static double pairwise_sum(const double *a, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        int i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

double xqo_np_sum(const double *a, int n) { return pairwise_sum(a, n); }

/* RandomState.choice(n, p=p), size=None, replace=True: cdf = p.cumsum(); cdf /= cdf[-1];
 * idx = cdf.searchsorted(u, side='right').  NaN in p -> ValueError (-1). */
int xqo_choice_from_uniform(const double *p, int n, double u)
{
    double cdf[XQO_MAX_MOVES];
    double acc = 0;
    for (int i = 0; i < n; i++) {
        if (isnan(p[i])) return -1;
        acc += p[i];
        cdf[i] = acc;
    }
    double last = cdf[n - 1];
    int lo = 0, hi = n;            /* searchsorted right: first i with u < cdf[i] */
    for (int i = 0; i < n; i++) cdf[i] = cdf[i] / last;
    while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* ------------------------------------------------------------------ HashNet evaluator
 * h0 = crc32(board.tobytes() + bytes([player & 0xff]));
 * per move h = crc32(bytes(move), h0); prior = np.float32(((h>>8)%64+1)/1024);
 * value = ((h0>>4)%65-32)/64. */
static uint32_t crc_table[256];
static int crc_ready = 0;

uint32_t xqo_crc32(uint32_t crc, const uint8_t *buf, int len)
{
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[i] = c;
        }
        crc_ready = 1;
    }
    crc = ~crc;
    for (int i = 0; i < len; i++) crc = crc_table[(crc ^ buf[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}

int xqo_hashnet_eval(void *ctx, int nrows, const int8_t *boards, const int32_t *players,
                     const uint16_t *moves, const int32_t *nmoves, float *priors, double *values)
{
    (void)ctx;
    for (int i = 0; i < nrows; i++) {
        uint8_t buf[91];
        memcpy(buf, boards + i * 90, 90);
        buf[90] = (uint8_t)(players[i] & 0xff);
        uint32_t h0 = xqo_crc32(0, buf, 91);
        for (int j = 0; j < nmoves[i]; j++) {
            int mv = moves[i * XQO_MAX_MOVES + j];
            int from = mv / 90, to = mv % 90;
            uint8_t mb[4] = { (uint8_t)(from / 9), (uint8_t)(from % 9), (uint8_t)(to / 9), (uint8_t)(to % 9) };
            uint32_t h = xqo_crc32(h0, mb, 4);
            priors[i * XQO_MAX_MOVES + j] = (float)((double)((h >> 8) % 64 + 1) / 1024.0);
        }
        values[i] = ((double)((h0 >> 4) % 65) - 32.0) / 64.0;
    }
    return 0;
}

/* ------------------------------------------------------------------ driver */

/* self_play.py:266-310 */
double xqo_z_value(int winner, int player, int game_length, int has_reward, double step_reward)
{
    double final_reward;
    if (winner == 0) {
        if (game_length >= 60) final_reward = (player == 1) ? -0.15 : 0.05;
        else final_reward = (player == 1) ? -0.1 : 0.1;
    } else if (winner == player) {
        double bonus;
        if (game_length <= 30) bonus = 0.5;
        else if (game_length <= 50) bonus = 0.3;
        else if (game_length <= 70) bonus = 0.1;
        else bonus = 0.0;
        final_reward = 1.0 + bonus;
    } else {
        final_reward = (game_length >= 60) ? -1.2 : -1.0;
    }
    double immediate = has_reward ? step_reward : 0.0;
    volatile double scaled = immediate * 0.01;
    return final_reward + scaled;
}

/* self_play.py:178-312 */
int xqo_self_play_game(const xqo_evaluator *eval_red, const xqo_evaluator *eval_black,
                       double temperature, int sims, int max_moves, uint32_t seed,
                       const double *pow_table, xqo_game *out)
{
    xqo_env *env = xqo_env_new();
    uint32_t mt[624];
    int mti;
    xqo_mt_seed(mt, &mti, seed);
    memset(out, 0, sizeof *out);

    double step_rewards[70];
    int n_plies = 0, n_samples = 0;
    int s_players[70];
    uint16_t moves[XQO_MAX_MOVES], legal[XQO_MAX_MOVES];
    int32_t visits[XQO_MAX_MOVES];
    double probs[XQO_MAX_MOVES];

    for (int move_num = 0; move_num < max_moves && move_num < 70; move_num++) {
        int player = env->current_player;
        if (xqo_legal_moves(env, legal) == 0) break;                           /* :205-208 */
        const xqo_evaluator *ev = (player == 1 || !eval_black) ? eval_red : eval_black;   /* :211 */
        int n = xqo_mcts_search(env, sims, ev, moves, visits);                 /* :214 */
        if (n < 0) { out->error = 2; break; }
        if (n == 0) break;                                                     /* :216-217 */

        if (temperature < 0.01) {                                              /* :224-227 */
            int am = 0;
            for (int i = 1; i < n; i++) if (visits[i] > visits[am]) am = i;
            for (int i = 0; i < n; i++) probs[i] = 0.0;
            probs[am] = 1;
        } else {                                                               /* :230-231 */
            double inv = 1.0 / temperature;
            for (int i = 0; i < n; i++)
                probs[i] = pow_table ? pow_table[visits[i]] : pow((double)visits[i], inv);
            double sum = xqo_np_sum(probs, n);
            for (int i = 0; i < n; i++) probs[i] = probs[i] / sum;
        }

        if (player == 1 || !eval_black) {                                      /* :234-239 */
            memcpy(out->s_board[n_samples], env->board, 90);
            out->s_player[n_samples] = player;
            s_players[n_samples] = player;
            out->s_nmoves[n_samples] = n;
            memcpy(out->s_moves[n_samples], moves, (size_t)n * sizeof(uint16_t));
            memcpy(out->s_probs[n_samples], probs, (size_t)n * sizeof(double));
            n_samples++;
        }

        double u = xqo_mt_double(mt, &mti);                                    /* :242 */
        int idx = xqo_choice_from_uniform(probs, n, u);
        if (idx < 0) { out->error = 1; break; }       /* ValueError: probabilities contain NaN */
        if (idx >= n) idx = n - 1;

        out->t_move[n_plies] = moves[idx];
        out->t_nchild[n_plies] = n;
        memcpy(out->t_moves[n_plies], moves, (size_t)n * sizeof(uint16_t));
        memcpy(out->t_visits[n_plies], visits, (size_t)n * sizeof(int32_t));

        double reward;
        int done = xqo_make_move(env, moves[idx], &reward, 0);                 /* :246 */
        step_rewards[n_plies] = reward;                                        /* :249 */
        out->t_reward[n_plies] = reward;
        n_plies++;
        if (done) break;                                                       /* :255-256 */
    }

    int winner = (env->winner == XQO_WINNER_NONE) ? 0 : env->winner;           /* :259 */
    out->winner = winner;
    out->end_reason = env->end_reason;                                         /* NONE -> "未知原因" */
    out->end_side = env->end_side;
    out->end_count = env->end_count;
    out->n_plies = n_plies;
    out->n_samples = n_samples;
    for (int i = 0; i < n_samples; i++)                                        /* :266-310 */
        out->s_z[i] = xqo_z_value(winner, s_players[i], n_samples, i < n_plies, i < n_plies ? step_rewards[i] : 0.0);
    xqo_env_free(env);
    return out->error;
}
