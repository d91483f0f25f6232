"""Host mirror of the reference's rules engine, class ChineseChess (chess_env.py:9-768).

Same public surface and attribute names; every rules computation goes through the C ABI of the
HIP library (include/xq_selfplay.h, batch of one) — there is no Python or CPU implementation
behind it.  Attributes stay live NumPy / Python objects because callers mutate them directly
(SURVEY.md §8b "Ownership"): the state is re-packed on every call, so stale king caches behave
exactly as in the reference (Appendix A6).
"""
import numpy as np

from . import _lib
from .config import BOARD_SIZE, BOARD_WIDTH, PIECES

_REASON_TERMINAL_INT = {1: 100, 2: 200, 3: 0, 4: 0, 5: 100, 6: -10, 7: -10, 8: -2}


def decode_move(m):
    f, t = divmod(int(m), 90)
    return (f // 9, f % 9, t // 9, t % 9)


def encode_move(move):
    fr, fc, tr, tc = (int(x) for x in move)
    return (fr * 9 + fc) * 90 + tr * 9 + tc


def _sq(pos):
    return _lib.NO_KING if pos is None else int(pos[0]) * 9 + int(pos[1])


def _pos(sq):
    return None if sq < 0 else (int(sq) // 9, int(sq) % 9)


def format_end_reason(code, side, count):
    """The reference's f-strings (chess_env.py:297,359,366,373,381,389,397,404)."""
    name = "红方" if side == 1 else "黑方"
    if code == 1:
        return f"{name}吃掉对方将帅"
    if code == 2:
        return f"将死{name}"
    if code == 3:
        return "三次重复局面判和"
    if code == 4:
        return "50回合无吃子判和"
    if code == 5:
        return f"困毙{name}"
    if code == 6:
        return f"长将判负({name})"
    if code == 7:
        return f"长捉判负({name})"
    if code == 8:
        return f"超过{count}步判和"
    return None


class _PendingChase:
    """Snapshot a chase_history entry is computed from when the list is first read."""
    __slots__ = ("board", "side", "rk", "bk")

    def __init__(self, board, side, rk, bk):
        self.board, self.side, self.rk, self.bk = board, side, rk, bk


class ChineseChess:
    # chase_history (chess_env.py:344-345) is a dead output of the reference: nothing reads it (:674) and computing
    # it costs the reference 51 % of its run time.  Default: entries are computed LAZILY, the
    # first time the list is read (one xq_rules_threatened_pieces call per pending entry); set
    # ChineseChess.track_chase = False to keep empty placeholders instead (len() still matches).
    track_chase = True

    def __init__(self):
        self.reset()

    def reset(self):
        """chess_env.py:14-67"""
        self.board = np.zeros((BOARD_SIZE, BOARD_WIDTH), dtype=np.int8)
        self.position_history = []
        self.no_capture_count = 0
        self.check_history = []
        self._chase = []                       # entries: list of pairs, or a pending (board, side, rk, bk) snapshot
        self.consecutive_checks = 0
        self.red_king_pos = None
        self.black_king_pos = None
        self.end_reason = None
        back = ['ROOK', 'KNIGHT', 'BISHOP', 'ADVISOR', 'KING', 'ADVISOR', 'BISHOP', 'KNIGHT', 'ROOK']
        for c, name in enumerate(back):
            self.board[9, c] = PIECES['R_' + name]
            self.board[0, c] = PIECES['B_' + name]
        self.red_king_pos = (9, 4)
        self.black_king_pos = (0, 4)
        self.board[7, 1] = self.board[7, 7] = PIECES['R_CANNON']
        self.board[2, 1] = self.board[2, 7] = PIECES['B_CANNON']
        for i in [0, 2, 4, 6, 8]:
            self.board[6, i] = PIECES['R_PAWN']
            self.board[3, i] = PIECES['B_PAWN']
        self.current_player = 1
        self.move_count = 0
        self.winner = None
        self.end_reason = None
        return self.get_state()

    def get_state(self):
        """chess_env.py:69-74"""
        return self.board.copy(), self.current_player

    # ---- packing helpers -------------------------------------------------------------
    def _board_bytes(self):
        return np.ascontiguousarray(self.board, dtype=np.int8).reshape(1, 90).copy()

    def _scalars(self):
        one = lambda v: np.array([v], dtype=np.int32)
        return one(self.current_player), one(_sq(self.red_king_pos)), one(_sq(self.black_king_pos))

    def get_legal_moves(self):
        """chess_env.py:76-121 (order included) via xq_rules_legal_moves."""
        L = _lib.lib()
        b = self._board_bytes()
        player, rk, bk = self._scalars()
        moves = np.zeros((1, _lib.MAX_MOVES), dtype=np.uint16)
        count = np.zeros(1, dtype=np.int32)
        _lib.check(L.xq_rules_legal_moves(1, _lib.ptr(b), _lib.ptr(player), _lib.ptr(rk), _lib.ptr(bk),
                                           _lib.ptr(moves), _lib.ptr(count)))
        return [decode_move(m) for m in moves[0, :count[0]]]

    def make_move(self, move):
        """chess_env.py:253-406 via xq_rules_make_move.  Returns (state, reward, done)."""
        L = _lib.lib()
        b = self._board_bytes()
        st = np.zeros((1, _lib.STATE_WORDS), dtype=np.int32)
        st[0, _lib.S_PLAYER] = self.current_player
        st[0, _lib.S_MOVE_COUNT] = self.move_count
        st[0, _lib.S_WINNER] = _lib.WINNER_NONE if self.winner is None else self.winner
        st[0, _lib.S_RED_KING] = _sq(self.red_king_pos)
        st[0, _lib.S_BLACK_KING] = _sq(self.black_king_pos)
        st[0, _lib.S_NO_CAPTURE] = self.no_capture_count
        st[0, _lib.S_CONSEC_CHECKS] = self.consecutive_checks
        nh, ncheck = len(self.position_history), len(self.check_history)
        stride = max(nh, ncheck, 1)
        ph = np.zeros((1, stride), dtype=np.uint64)
        ch = np.zeros((1, stride), dtype=np.uint8)
        if nh:
            ph[0, :nh] = np.array([int(h) & 0xFFFFFFFFFFFFFFFF for h in self.position_history], dtype=np.uint64)
        if ncheck:
            ch[0, :ncheck] = np.array([1 if c else 0 for c in self.check_history], dtype=np.uint8)
        mv = np.array([encode_move(move)], dtype=np.int32)
        n_hist, n_chk = np.array([nh], np.int32), np.array([ncheck], np.int32)
        reward, done, is_check = np.zeros(1, np.float64), np.zeros(1, np.int32), np.zeros(1, np.int32)
        key = np.zeros(1, np.uint64)
        nxt, nxt_n = np.zeros((1, _lib.MAX_MOVES), np.uint16), np.zeros(1, np.int32)
        _lib.check(L.xq_rules_make_move(1, _lib.ptr(b), _lib.ptr(st), _lib.ptr(mv), _lib.ptr(ph), _lib.ptr(n_hist),
                                         _lib.ptr(ch), _lib.ptr(n_chk), stride, _lib.ptr(reward), _lib.ptr(done),
                                         _lib.ptr(is_check), _lib.ptr(key), _lib.ptr(nxt), _lib.ptr(nxt_n)))
        self.board = b.reshape(BOARD_SIZE, BOARD_WIDTH).copy()
        self.current_player = int(st[0, _lib.S_PLAYER])
        self.move_count = int(st[0, _lib.S_MOVE_COUNT])
        w = int(st[0, _lib.S_WINNER])
        self.winner = None if w == _lib.WINNER_NONE else w
        self.red_king_pos = _pos(st[0, _lib.S_RED_KING])
        self.black_king_pos = _pos(st[0, _lib.S_BLACK_KING])
        self.no_capture_count = int(st[0, _lib.S_NO_CAPTURE])
        self.consecutive_checks = int(st[0, _lib.S_CONSEC_CHECKS])
        self.position_history.append(int(key[0]))
        self.check_history.append(bool(is_check[0]))
        # chase_history entry (chess_env.py:344-345): threats of the MOVER on the position after the move, taken
        # before the side switch — kept as a snapshot and evaluated only if somebody reads the list
        self._chase.append(_PendingChase(b.copy(), -self.current_player, int(st[0, _lib.S_RED_KING]),
                                         int(st[0, _lib.S_BLACK_KING])) if self.track_chase else [])
        r = float(reward[0])
        if done[0]:
            code = int(st[0, _lib.S_REASON])
            self.end_reason = format_end_reason(code, int(st[0, _lib.S_REASON_SIDE]), int(st[0, _lib.S_REASON_COUNT]))
            r = _REASON_TERMINAL_INT.get(code, r)        # terminal overrides are Python ints (A16)
        return self.get_state(), r, bool(done[0])

    # ---- a8: _get_threatened_pieces / chase_history ----------------------------------------
    @staticmethod
    def _threats(board_bytes, side, rk, bk):
        L = _lib.lib()
        one = lambda v: np.array([v], dtype=np.int32)
        pairs = np.zeros((1, _lib.MAX_MOVES), dtype=np.uint16)
        count = np.zeros(1, dtype=np.int32)
        _lib.check(L.xq_rules_threatened_pieces(1, _lib.ptr(board_bytes), _lib.ptr(one(side)), _lib.ptr(one(rk)),
                                                 _lib.ptr(one(bk)), _lib.ptr(pairs), _lib.ptr(count)))
        out = []
        for m in pairs[0, :count[0]]:
            fr, fc, tr, tc = decode_move(m)
            out.append(((fr, fc), (tr, tc)))
        return out

    def _get_threatened_pieces(self, player):
        """chess_env.py:550-574: [((r, c), (tr, tc)), ...] for the pieces `player` threatens."""
        _, rk, bk = self._scalars()
        return self._threats(self._board_bytes(), int(player), int(rk[0]), int(bk[0]))

    def _is_protected(self, r, c, player):
        """chess_env.py:576-596.  The executed reference can never answer True: _get_piece_moves drops every move
        onto a square held by the mover's own side (:116), so no move of `player` ends on its own piece at (r, c)
        (pinned by tests/golden/rules_extra.json through the oracle's literal restatement)."""
        return False

    @property
    def chase_history(self):
        for i, e in enumerate(self._chase):
            if isinstance(e, _PendingChase):
                self._chase[i] = self._threats(e.board, e.side, e.rk, e.bk)
        return self._chase

    @chase_history.setter
    def chase_history(self, value):
        self._chase = value

    # ---- predicates the reference's own tests call -------------------------------------
    def _get_position_hash(self):
        """chess_env.py:497-504: equal values <=> equal (board, current_player); the reference's are salted Python
        hashes, these are the 64-bit keys the HIP rules engine compares (xq_rules_position_key)."""
        L = _lib.lib()
        key = np.zeros(1, np.uint64)
        player, _, _ = self._scalars()
        _lib.check(L.xq_rules_position_key(1, _lib.ptr(self._board_bytes()), _lib.ptr(player), _lib.ptr(key)))
        return int(key[0])

    def _check_draw_by_repetition(self):
        """chess_env.py:598-605"""
        return self.position_history.count(self._get_position_hash()) >= 3

    def _check_checkmate(self):
        """chess_env.py:614-628"""
        return len(self.get_legal_moves()) == 0 and self._is_in_check(self.current_player)

    def _check_stalemate(self):
        """chess_env.py:630-644"""
        return len(self.get_legal_moves()) == 0 and not self._is_in_check(self.current_player)

    def _is_move_suicide(self, from_r, from_c, to_r, to_c):
        """chess_env.py:431-464: play the move on a copy (the king cache follows only a moving king, Appendix A5),
        then in-check of the side to move or kings facing."""
        probe = ChineseChess.__new__(ChineseChess)
        probe.board = self.board.copy()
        probe.current_player = self.current_player
        probe.red_king_pos, probe.black_king_pos = self.red_king_pos, self.black_king_pos
        moving = probe.board[from_r, from_c]
        probe.board[to_r, to_c] = moving
        probe.board[from_r, from_c] = 0
        if moving == PIECES['R_KING']:
            probe.red_king_pos = (to_r, to_c)
        elif moving == PIECES['B_KING']:
            probe.black_king_pos = (to_r, to_c)
        red, black, facing = probe._query()
        return (red if self.current_player == 1 else black) or facing

    def _query(self):
        L = _lib.lib()
        b = self._board_bytes()
        player, rk, bk = self._scalars()
        a, c, f = (np.zeros(1, np.int32) for _ in range(3))
        _lib.check(L.xq_rules_query(1, _lib.ptr(b), _lib.ptr(player), _lib.ptr(rk), _lib.ptr(bk),
                                     _lib.ptr(a), _lib.ptr(c), _lib.ptr(f)))
        return bool(a[0]), bool(c[0]), bool(f[0])

    def _is_in_check(self, player):
        """chess_env.py:506-548"""
        red, black, _ = self._query()
        return red if player == 1 else black

    def _are_kings_facing(self):
        """chess_env.py:466-495"""
        return self._query()[2]

    def _check_perpetual_check(self):
        """chess_env.py:646-662 (host logic on the public list)"""
        if len(self.check_history) < 12:
            return False
        return sum(1 for c in self.check_history[-12:] if c) >= 10

    def _check_perpetual_chase(self):
        """chess_env.py:664-674: disabled in the reference"""
        return False

    def _check_draw_by_fifty_moves(self):
        """chess_env.py:607-612"""
        return self.no_capture_count >= 100

    _GLYPHS = "·帅士相马车炮兵卒炮车马象士将"      # index = piece code (negative codes index from the end)

    def render(self):
        """Text dump of the position (chess_env.py:408-429): header row, one line per rank, side
        to move and ply counter."""
        lines = ["", "  " + "".join(f"{c} " for c in range(BOARD_WIDTH))]
        for r in range(BOARD_SIZE):
            cells = " ".join(self._GLYPHS[int(p)] for p in self.board[r])
            lines.append(f"{r} {cells} ")
        lines.append("")
        lines.append("当前: " + ("红方" if self.current_player == 1 else "黑方"))
        lines.append(f"步数: {self.move_count}")
        print("\n".join(lines))


ChessEnv = ChineseChess          # BASELINE north_star's name for the class (chess_env.py:9)
