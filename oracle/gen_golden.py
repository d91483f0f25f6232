#!/usr/bin/env python3
"""Golden-vector generator: runs the UNMODIFIED reference (imported from /root/reference) and
writes small data fixtures to tests/golden/.  Runs only in the build container; the reference
never travels.  Nothing from the reference's source is copied — only inputs and the outputs the
executed code produced.

    python oracle/gen_golden.py [rules|edge|known|puct|sampler|ztable|search|net|all] ...

Harness-only monkeypatching (the reference functions under test always run unmodified):
  * search fixtures wrap MCTS.search / ChineseChess.make_move with pass-through recorders;
  * z-table fixtures replace the MCTS *object* by a scripted policy so that self_play_game's own
    z-assignment code sees wins, losses and draws of chosen lengths.
"""
import json
import os
import random
import sys
import zlib

REF = os.environ.get("XQ_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
os.environ.setdefault("OMP_NUM_THREADS", "1")

import numpy as np  # noqa: E402

import io  # noqa: E402
import contextlib  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    import chess_env  # noqa: E402
    import self_play  # noqa: E402

# (XQ_GOLDEN_OUT: tests/test_golden_regen_cpu.py regenerates into a scratch directory and compares bytes)
OUT = os.environ.get("XQ_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)

WINNER_NONE = 2
NO_KING = -1


def enc(mv):
    fr, fc, tr, tc = (int(x) for x in mv)
    return (fr * 9 + fc) * 90 + tr * 9 + tc


def sq(pos):
    return NO_KING if pos is None else int(pos[0]) * 9 + int(pos[1])


def reason_code(s):
    """end_reason string -> (code, side, count); codes as in include/xq_selfplay.h."""
    if s is None:
        return (0, 0, 0)
    side = lambda name: 1 if name == "红方" else -1
    if s.endswith("吃掉对方将帅"):
        return (1, side(s[:2]), 0)
    if s.startswith("将死"):
        return (2, side(s[2:4]), 0)
    if s == "三次重复局面判和":
        return (3, 0, 0)
    if s == "50回合无吃子判和":
        return (4, 0, 0)
    if s.startswith("困毙"):
        return (5, side(s[2:4]), 0)
    if s.startswith("长将判负"):
        return (6, side(s[5:7]), 0)
    if s.startswith("长捉判负"):
        return (7, side(s[5:7]), 0)
    if s.startswith("超过") and s.endswith("步判和"):
        return (8, 0, int(s[2:-3]))
    if s == "未知原因":
        return (0, 0, 0)
    raise ValueError(s)


def state_row(env):
    return dict(board=env.board.astype(np.int8).reshape(90).copy(), player=int(env.current_player),
                move_count=int(env.move_count), red_king=sq(env.red_king_pos),
                black_king=sq(env.black_king_pos), no_capture=int(env.no_capture_count),
                consecutive_checks=int(env.consecutive_checks),
                winner=WINNER_NONE if env.winner is None else int(env.winner))


# --------------------------------------------------------------------------- G1/G2 rules
def gen_rules(n_games=48, seed=20251205):
    rng = random.Random(seed)
    rows = []
    for g in range(n_games):
        env = chess_env.ChineseChess()
        # game styles: plain random / capture-greedy / long (continue after done, Appendix A9)
        style = g % 3
        limit = 70 if style == 0 else (110 if style == 1 else 220)
        for ply in range(limit):
            before = state_row(env)
            legal = env.get_legal_moves()
            if not legal:
                break
            caps = [m for m in legal if env.board[m[2], m[3]] != 0]
            if style >= 1 and caps and rng.random() < 0.7:
                mv = rng.choice(caps)
            else:
                mv = rng.choice(legal)
            _, reward, done = env.make_move(mv)
            code, side, cnt = reason_code(env.end_reason)
            rows.append(dict(game=g, ply=ply, **before,
                             legal=[enc(m) for m in legal], move=enc(mv),
                             reward=float(reward), done=bool(done),
                             winner_after=WINNER_NONE if env.winner is None else int(env.winner),
                             reason=code, reason_side=side, reason_count=cnt,
                             cc_after=int(env.consecutive_checks), nc_after=int(env.no_capture_count),
                             is_check=bool(env.check_history[-1]),
                             rk_after=sq(env.red_king_pos), bk_after=sq(env.black_king_pos),
                             n_hist=len(env.position_history)))
            if done and style == 0:
                break
            if env.red_king_pos is None or env.black_king_pos is None:
                break
    n = len(rows)
    legal = np.zeros((n, 128), np.uint16)
    nlegal = np.zeros(n, np.int32)
    for i, r in enumerate(rows):
        nlegal[i] = len(r["legal"])
        legal[i, :nlegal[i]] = r["legal"]
    pack = dict(legal=legal, nlegal=nlegal)
    for k in ("game", "ply", "player", "move_count", "red_king", "black_king", "no_capture",
              "consecutive_checks", "winner", "move", "winner_after", "reason", "reason_side",
              "reason_count", "cc_after", "nc_after", "rk_after", "bk_after", "n_hist"):
        pack[k] = np.array([r[k] for r in rows], np.int32)
    pack["board"] = np.stack([r["board"] for r in rows]).astype(np.int8)
    pack["reward"] = np.array([r["reward"] for r in rows], np.float64)
    pack["done"] = np.array([r["done"] for r in rows], np.uint8)
    pack["is_check"] = np.array([r["is_check"] for r in rows], np.uint8)
    np.savez_compressed(os.path.join(OUT, "rules_random.npz"), **pack)
    stats = dict(rows=n, done=int(pack["done"].sum()),
                 reasons={int(k): int((pack["reason"][pack["done"] == 1] == k).sum()) for k in range(9)},
                 checks=int(pack["is_check"].sum()), max_legal=int(nlegal.max()))
    print("rules_random:", stats)


# --------------------------------------------------------------------------- edge boards
def P(name):
    from config import PIECES
    return PIECES[name]


def edge_boards():
    """Hand-built positions: the boards of the reference's own unit tests (test_kings_facing.py,
    test_reward_system.py, test_fixes.py style) and the Appendix-A cases.  Each entry:
    (name, {square: piece}, player, red_king_cache, black_king_cache)."""
    K, A, B, N, R, Cn, Pw = 1, 2, 3, 4, 5, 6, 7
    cases = []
    # test_kings_facing.py boards (caches stay at the reset squares (9,4)/(0,4): stale, A6)
    cases.append(("kf1_direct_stale", {(2, 4): -K, (8, 4): K}, 1, (9, 4), (0, 4)))
    cases.append(("kf2_blocker_stale", {(2, 4): -K, (5, 4): R, (8, 4): K}, 1, (9, 4), (0, 4)))
    cases.append(("kf3_cannon_stale", {(2, 4): -K, (5, 4): Cn, (8, 4): K}, 1, (9, 4), (0, 4)))
    cases.append(("kf4_diffcol_stale", {(2, 3): -K, (8, 4): K}, 1, (9, 4), (0, 4)))
    cases.append(("kf5_arrow", {(0, 4): -K, (4, 4): -R, (9, 4): K}, -1, (9, 4), (0, 4)))
    # same boards with correct caches
    cases.append(("kf1_direct", {(2, 4): -K, (8, 4): K}, 1, (8, 4), (2, 4)))
    cases.append(("kf3_cannon", {(2, 4): -K, (5, 4): Cn, (8, 4): K}, 1, (8, 4), (2, 4)))
    cases.append(("kf3_cannon_black", {(2, 4): -K, (5, 4): -Cn, (8, 4): K}, -1, (8, 4), (2, 4)))
    # test_reward_system.py boards
    cases.append(("rw1_cannon_takes_king", {(0, 4): -K, (0, 1): Cn, (9, 4): K}, 1, (9, 4), (0, 4)))
    cases.append(("rw2_rook_takes_rook", {(0, 4): -K, (9, 4): K, (0, 0): -R, (0, 8): R}, 1, (9, 4), (0, 4)))
    cases.append(("rw3_rook_check", {(0, 4): -K, (9, 4): K, (2, 4): R}, 1, (9, 4), (0, 4)))
    cases.append(("rw4_suicide", {(9, 4): K, (9, 3): A, (0, 4): -K, (7, 4): -R}, 1, (9, 4), (0, 4)))
    # Appendix A1: pawn quirk in the self-check filter
    cases.append(("a1_pawn_front", {(8, 4): K, (7, 4): -Pw, (0, 3): -K}, 1, (8, 4), (0, 3)))
    cases.append(("a1_pawn_behind", {(8, 4): K, (9, 4): -Pw, (0, 3): -K}, 1, (8, 4), (0, 3)))
    cases.append(("a1_pawn_side", {(8, 4): K, (8, 3): -Pw, (0, 3): -K}, 1, (8, 4), (0, 3)))
    cases.append(("a1_pawn_front_blk", {(8, 4): K, (7, 4): -Pw, (0, 3): -K}, -1, (8, 4), (0, 3)))
    cases.append(("a1_black_pawn_quirk", {(1, 4): -K, (2, 4): Pw, (9, 3): K}, -1, (9, 3), (1, 4)))
    cases.append(("a1_black_pawn_above", {(1, 4): -K, (0, 4): Pw, (9, 3): K}, -1, (9, 3), (1, 4)))
    # Appendix A5: king capture filtered by the stale enemy-king cache
    cases.append(("a5_sole_blocker", {(0, 4): -K, (1, 4): Pw, (9, 4): K}, 1, (9, 4), (0, 4)))
    cases.append(("a5_two_blockers", {(0, 4): -K, (1, 4): Pw, (6, 4): Pw, (9, 4): K}, 1, (9, 4), (0, 4)))
    # knights / bishops / advisors / cannons near edges
    cases.append(("knight_corner", {(0, 0): N, (9, 4): K, (0, 4): -K, (1, 0): -Pw}, 1, (9, 4), (0, 4)))
    cases.append(("knight_legs", {(5, 4): N, (4, 4): Pw, (5, 5): -Pw, (9, 4): K, (0, 3): -K}, 1, (9, 4), (0, 3)))
    cases.append(("bishop_river", {(5, 2): B, (7, 4): B, (9, 4): K, (0, 3): -K, (6, 3): -Pw}, 1, (9, 4), (0, 3)))
    cases.append(("bishop_black", {(4, 2): -B, (2, 4): -B, (9, 3): K, (0, 4): -K, (3, 3): Pw}, -1, (9, 3), (0, 4)))
    cases.append(("advisors", {(8, 4): A, (9, 3): A, (9, 4): K, (0, 3): -K}, 1, (9, 4), (0, 3)))
    cases.append(("cannon_screens", {(5, 0): Cn, (5, 3): -Pw, (5, 6): -R, (2, 0): Pw, (0, 0): -N, (9, 4): K, (0, 3): -K}, 1, (9, 4), (0, 3)))
    cases.append(("rook_cannon_check", {(0, 4): -K, (9, 3): K, (5, 4): R, (3, 4): -A}, -1, (9, 3), (0, 4)))
    cases.append(("double_cannon", {(0, 4): -K, (0, 3): -A, (0, 5): -A, (3, 4): Cn, (5, 4): Cn, (9, 3): K}, -1, (9, 3), (0, 4)))
    cases.append(("stalemate_like", {(0, 3): -K, (2, 3): Pw, (1, 5): R, (9, 4): K}, -1, (9, 4), (0, 3)))
    cases.append(("no_king_cache", {(0, 4): -K, (9, 4): K, (5, 5): R}, 1, None, (0, 4)))
    cases.append(("many_moves", {(4, 4): R, (5, 3): R, (4, 1): Cn, (6, 6): Cn, (5, 5): N, (3, 2): N, (9, 4): K, (0, 3): -K,
                                 (6, 0): Pw, (4, 8): Pw, (3, 6): Pw}, 1, (9, 4), (0, 3)))
    return cases


def gen_edge():
    out = []
    for name, pieces, player, rk, bk in edge_boards():
        env = chess_env.ChineseChess()
        env.board = np.zeros((10, 9), dtype=np.int8)
        for (r, c), p in pieces.items():
            env.board[r, c] = p
        env.current_player = player
        env.red_king_pos = rk
        env.black_king_pos = bk
        legal = env.get_legal_moves()
        rec = dict(name=name, board=env.board.reshape(90).tolist(), player=player,
                   red_king=sq(rk), black_king=sq(bk),
                   legal=[enc(m) for m in legal],
                   in_check_red=bool(env._is_in_check(1)), in_check_black=bool(env._is_in_check(-1)),
                   facing=bool(env._are_kings_facing()))
        # make_move outcome of every legal move from this position (fresh env each time)
        mm = []
        for mv in legal:
            e2 = chess_env.ChineseChess()
            e2.board = env.board.copy()
            e2.current_player = player
            e2.red_king_pos, e2.black_king_pos = rk, bk
            _, reward, done = e2.make_move(mv)
            code, side, cnt = reason_code(e2.end_reason)
            mm.append(dict(move=enc(mv), reward=float(reward), done=bool(done),
                           winner=WINNER_NONE if e2.winner is None else int(e2.winner),
                           reason=code, reason_side=side, is_check=bool(e2.check_history[-1]),
                           rk=sq(e2.red_king_pos), bk=sq(e2.black_king_pos)))
        rec["moves"] = mm
        out.append(rec)
    with open(os.path.join(OUT, "rules_edge.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("rules_edge:", len(out), "boards")


# --------------------------------------------------------------------------- G3 known answers
def gen_known():
    env = chess_env.ChineseChess()
    init = [enc(m) for m in env.get_legal_moves()]
    line = [(7, 7, 7, 4), (0, 1, 2, 0), (7, 4, 3, 4), (2, 7, 7, 7), (7, 1, 5, 1), (2, 1, 2, 7), (5, 1, 5, 4)]
    rewards, dones = [], []
    for mv in line:
        _, r, d = env.make_move(mv)
        rewards.append(float(r))
        dones.append(bool(d))
    # perpetual-check predicate on injected histories (test_perpetual_rules.py:20-50)
    perp = []
    for hist in ([True, False] * 6, [True] * 11 + [False], [True] * 10 + [False, False],
                 [False, False] + [True] * 9 + [False], [True] * 11, [False] * 3 + [True] * 10):
        e = chess_env.ChineseChess()
        e.check_history = list(hist)
        perp.append(dict(hist=[int(x) for x in hist], result=bool(e._check_perpetual_check())))
    with open(os.path.join(OUT, "known.json"), "w") as f:
        json.dump(dict(initial_moves=init, mate_line=[enc(m) for m in line], mate_rewards=rewards,
                       mate_dones=dones, mate_winner=int(env.winner),
                       mate_reason=reason_code(env.end_reason), perpetual=perp), f)
    print("known: init", len(init), "mate rewards", rewards, env.winner, env.end_reason)


# --------------------------------------------------------------------------- G5 PUCT selection
def gen_puct(n=1500, seed=7):
    rng = np.random.RandomState(seed)
    rows = []
    for i in range(n):
        nc = int(rng.randint(2, 60))
        parent = self_play.MCTSNode()
        parent.visit_count = int(rng.choice([1, 2, 3, 8, 9, 16, 17, 24, 42, 50, 97, 200]))
        vis, wsum, pri = [], [], []
        for j in range(nc):
            ch = self_play.MCTSNode(parent=parent, move=(0, 0, 0, j), prior_prob=np.float32(int(rng.randint(1, 65)) / 1024))
            if rng.rand() < 0.5:
                ch.visit_count = int(rng.choice([1, 2, 8, 16, 3, 5]))
                # value sums as the driver builds them: sequences of +-k/64 floats or ints
                ch.value_sum = float(int(rng.randint(-32, 33)) / 64 * ch.visit_count) if rng.rand() < 0.8 else int(rng.randint(-2, 3))
            parent.children[ch.move] = ch
            vis.append(ch.visit_count)
            wsum.append(float(ch.value_sum))
            pri.append(float(ch.prior_prob))
        mv, _ = parent.select_child()
        rows.append(dict(N=parent.visit_count, visits=vis, wsum=wsum, prior=pri, best=int(mv[3])))
    # real-valued priors (softmax-like) too
    for i in range(500):
        nc = int(rng.randint(2, 60))
        parent = self_play.MCTSNode()
        parent.visit_count = int(rng.randint(1, 200))
        logits = rng.randn(nc).astype(np.float32)
        p = np.exp(logits - logits.max())
        p = p / p.sum()
        vis, wsum, pri = [], [], []
        for j in range(nc):
            ch = self_play.MCTSNode(parent=parent, move=(0, 0, 0, j), prior_prob=p[j])
            if rng.rand() < 0.5:
                ch.visit_count = int(rng.randint(1, 30))
                ch.value_sum = float(np.float32(rng.uniform(-1, 1))) * ch.visit_count
            parent.children[ch.move] = ch
            vis.append(ch.visit_count)
            wsum.append(float(ch.value_sum))
            pri.append(float(ch.prior_prob))
        mv, _ = parent.select_child()
        rows.append(dict(N=parent.visit_count, visits=vis, wsum=wsum, prior=pri, best=int(mv[3])))
    with open(os.path.join(OUT, "puct.json"), "w") as f:
        json.dump(rows, f, separators=(",", ":"))
    print("puct:", len(rows))


# --------------------------------------------------------------------------- G6 sampler
def gen_sampler(seed=11):
    rng = np.random.RandomState(seed)
    cases = []
    for i in range(300):
        n = int(rng.randint(1, 70))
        S = int(rng.choice([15, 16, 24, 50, 200]))
        T = float(rng.choice([1.0, 1.0, 0.5, 0.1, 0.3]))
        # visit vectors shaped like search output: a few non-zero children
        counts = np.zeros(n, dtype=np.int64)
        k = int(rng.randint(1, min(n, 8) + 1))
        idx = rng.choice(n, size=k, replace=False)
        counts[idx] = rng.randint(1, 9, size=k) * int(rng.choice([1, 1, 8]))
        c = counts ** (1.0 / T)
        p = c / c.sum()
        s = int(rng.randint(0, 2 ** 31 - 1))
        np.random.seed(s)
        draws = [int(np.random.choice(n, p=p)) for _ in range(3)]
        np.random.seed(s)
        us = [float(np.random.random_sample()) for _ in range(3)]
        cases.append(dict(seed=s, counts=counts.tolist(), T=T, pow=[float(x) for x in c],
                          p=[float(x) for x in p], draws=draws, uniforms=us))
    sums = []
    for i in range(200):
        n = int(rng.randint(1, 129))
        a = rng.rand(n) * (10.0 ** rng.randint(-3, 20))
        sums.append(dict(a=[float(x) for x in a], s=float(a.sum())))
    np.random.seed(0)
    first = [float(np.random.random_sample()) for _ in range(3)]
    with open(os.path.join(OUT, "sampler.json"), "w") as f:
        json.dump(dict(cases=cases, sums=sums, seed0_first3=first), f, separators=(",", ":"))
    print("sampler:", len(cases), "cases;", first)


# --------------------------------------------------------------------------- HashNet
class HashNet:
    """Exact dyadic evaluator; salt distinguishes a second 'network'."""

    def __init__(self, salt=0):
        self.salt = bytes([salt]) if salt else b""

    def predict_batch(self, rows):
        out = []
        for board, player, legal in rows:
            h0 = zlib.crc32(board.tobytes() + bytes([player & 0xff]) + self.salt)
            probs = {}
            for mv in legal:
                h = zlib.crc32(bytes(mv), h0)
                probs[mv] = np.float32(((h >> 8) % 64 + 1) / 1024)
            out.append((probs, ((h0 >> 4) % 65 - 32) / 64))
        return out


def play_recorded(seed, sims, temperature=1.0, opponent=False):
    """self_play_game (unmodified) with pass-through recorders around search and make_move."""
    rec = dict(visits=[], moves=[], rewards=[])
    real_env = {}
    orig_search = self_play.MCTS.search
    orig_make = chess_env.ChineseChess.make_move

    def search(self, env, num_simulations=None):
        real_env["id"] = id(env)
        vc = orig_search(self, env, num_simulations)
        rec["visits"].append([(enc(m), int(v)) for m, v in vc.items()])
        return vc

    def make_move(self, move):
        res = orig_make(self, move)
        if real_env.get("id") == id(self):
            rec["moves"].append(enc(move))
            rec["rewards"].append(float(res[1]))
        return res

    self_play.MCTS.search = search
    chess_env.ChineseChess.make_move = make_move
    try:
        np.random.seed(seed)
        err = None
        try:
            data, winner, reason = self_play.self_play_game(
                HashNet(), temperature=temperature, num_simulations=sims,
                opponent_network=HashNet(1) if opponent else None)
        except ValueError as ex:
            err = str(ex)
            data, winner, reason = [], 0, None
    finally:
        self_play.MCTS.search = orig_search
        chess_env.ChineseChess.make_move = orig_make
    code, side, cnt = reason_code(reason)
    crc = 0
    for b, _, _ in data:
        crc = zlib.crc32(b.tobytes(), crc)
    return dict(seed=seed, sims=sims, T=temperature, opponent=opponent, error=err,
                winner=int(winner), reason=code, reason_side=side, reason_count=cnt,
                n_samples=len(data), visits=rec["visits"], moves=rec["moves"], rewards=rec["rewards"],
                z=[float(z) for _, _, z in data],
                pi=[[float(p) for p in d.values()] for _, d, _ in data],
                pi_moves=[[enc(m) for m in d.keys()] for _, d, _ in data],
                boards_crc=crc)


def _search_job(args):
    return play_recorded(*args)


def gen_search(which="fast"):
    import multiprocessing as mp
    jobs = []
    if which in ("fast", "all"):
        jobs += [(s, 15, 1.0, False) for s in (0, 1, 2, 3)]
        jobs += [(s, 16, 1.0, False) for s in (0, 1)]
        jobs += [(s, 24, 1.0, False) for s in (0, 1)]
        jobs += [(5, 24, 0.5, False), (6, 24, 0.1, False), (7, 16, 0.001, False)]
        jobs += [(8, 24, 1.0, True), (9, 15, 0.5, True)]
        jobs += [(0, 8, 1.0, False)]                       # S<=8 -> ValueError (NaN probabilities)
        jobs += [(s, 50, 1.0, False) for s in (0, 1, 2, 3)]
        name = "search_hashnet.json"
    if which == "slow":
        jobs += [(0, 200, 1.0, False), (1, 100, 1.0, False)]
        name = "search_hashnet_slow.json"
    with mp.Pool(min(7, len(jobs))) as pool:
        res = pool.map(_search_job, jobs, chunksize=1)
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(res, f, separators=(",", ":"))
    for r in res:
        print("search:", r["seed"], r["sims"], r["T"], r["opponent"], "->", r["winner"], r["reason"],
              r["n_samples"], "%08x" % r["boards_crc"], r["error"])


# --------------------------------------------------------------------------- G7 z-table
class ScriptedMCTS:
    """Stands in for the MCTS object only; self_play_game's own code assigns z."""
    script = []

    def __init__(self, network, num_simulations=None):
        pass

    def search(self, env, num_simulations=None):
        if env.move_count < len(self.script):
            return {self.script[env.move_count]: 1}
        legal = env.get_legal_moves()
        return {legal[0]: 1}


def gen_ztable():
    mate_red = [(7, 7, 7, 4), (0, 1, 2, 0), (7, 4, 3, 4), (2, 7, 7, 7), (7, 1, 5, 1), (2, 1, 2, 7), (5, 1, 5, 4)]
    shuffle = [(9, 0, 8, 0), (0, 0, 1, 0), (8, 0, 9, 0), (1, 0, 0, 0)]

    def mirror(m):
        return (9 - m[0], 8 - m[1], 9 - m[2], 8 - m[3])

    # black mates: red wastes a tempo with a rook step, then colours are swapped
    mate_black = [(9, 8, 8, 8)] + [mirror(m) for m in mate_red]
    scripts = []
    for k in (0, 6, 7, 11, 12, 13, 14, 15):
        scripts.append(("red_win_%d" % k, shuffle * k + mate_red))
        scripts.append(("black_win_%d" % k, shuffle * k + mate_black))
    scripts.append(("draw_cap", shuffle * 18))
    out = []
    orig = self_play.MCTS
    self_play.MCTS = ScriptedMCTS
    try:
        for name, sc in scripts:
            ScriptedMCTS.script = sc
            np.random.seed(0)
            rec = dict(rewards=[])
            orig_make = chess_env.ChineseChess.make_move

            def make_move(self, move, _o=orig_make, _r=rec):
                res = _o(self, move)
                _r["rewards"].append(float(res[1]))
                return res

            chess_env.ChineseChess.make_move = make_move
            try:
                data, winner, reason = self_play.self_play_game(None, num_simulations=1)
            finally:
                chess_env.ChineseChess.make_move = orig_make
            # player of sample i: red on even plies (self-play mode stores every ply)
            out.append(dict(name=name, winner=int(winner), reason=reason_code(reason), length=len(data),
                            players=[1 if i % 2 == 0 else -1 for i in range(len(data))],
                            step_rewards=rec["rewards"][:len(data)], z=[float(z) for _, _, z in data],
                            script=[enc(m) for m in sc]))
            print("ztable:", name, winner, reason, len(data))
    finally:
        self_play.MCTS = orig
    with open(os.path.join(OUT, "ztable.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


# --------------------------------------------------------------------------- G8 net
def gen_net():
    import torch
    with contextlib.redirect_stdout(io.StringIO()):
        import neural_network
    torch.manual_seed(0)
    net = neural_network.ChessNet()
    net.eval()
    # a second copy with non-trivial BatchNorm statistics so that BN folding is exercised
    torch.manual_seed(1)
    net2 = neural_network.ChessNet()
    with torch.no_grad():
        for m in net2.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.8, 1.2)
                m.bias.uniform_(-0.1, 0.1)
    net2.eval()
    rng = random.Random(3)
    rows = []
    env = chess_env.ChineseChess()
    for i in range(16):
        for _ in range(rng.randint(0, 6)):
            lm = env.get_legal_moves()
            if not lm:
                break
            env.make_move(rng.choice(lm))
        b, p = env.get_state()
        rows.append((b, p, env.get_legal_moves()))
    pack = dict(boards=np.stack([r[0] for r in rows]).astype(np.int8),
                players=np.array([r[1] for r in rows], np.int32))
    legal = np.zeros((16, 128), np.uint16)
    nlegal = np.zeros(16, np.int32)
    for i, r in enumerate(rows):
        nlegal[i] = len(r[2])
        legal[i, :nlegal[i]] = [enc(m) for m in r[2]]
    pack["legal"], pack["nlegal"] = legal, nlegal
    for tag, n in (("a", net), ("b", net2)):
        res = n.predict_batch(rows)
        pri = np.zeros((16, 128), np.float32)
        val = np.zeros(16, np.float64)
        for i, (d, v) in enumerate(res):
            pri[i, :len(d)] = list(d.values())
            val[i] = v
        x = torch.from_numpy(np.stack([n.encode_board(b, p) for b, p, _ in rows]))
        with torch.no_grad():
            logits, _ = n(x)
        lg = np.zeros((16, 128), np.float32)
        for i in range(16):
            idx = [(m // 90) * 90 + (m % 90) for m in legal[i, :nlegal[i]]]
            lg[i, :nlegal[i]] = logits[i].numpy()[idx]
        pack["priors_" + tag], pack["values_" + tag], pack["logits_" + tag] = pri, val, lg
        sd = n.state_dict()
        pack["wsum_" + tag] = np.array([float(sd[k].double().sum()) for k in sorted(sd) if sd[k].dtype.is_floating_point])
    pack["planes"] = np.stack([net.encode_board(b, p) for b, p, _ in rows]).astype(np.float32)
    pack["n_params"] = np.array([sum(p.numel() for p in net.parameters())])
    np.savez_compressed(os.path.join(OUT, "net.npz"), **pack)
    print("net: params", int(pack["n_params"][0]), "values", pack["values_a"][:4])


# --------------------------------------------------------------------------- round 2: rules extras
def gen_rules_extra(seed=424242):
    """a8 (_get_threatened_pieces / chase_history, chess_env.py:550-596,344-345), the repetition draw with an
    injected position_history (chess_env.py:598-605: unreachable in legal play, Appendix A7) and the private
    predicates the reference's own tests call (_check_checkmate, _check_stalemate, _is_move_suicide)."""
    rng = random.Random(seed)
    out = dict(chase=[], threats=[], repetition=[], predicates=[])
    # chase_history per ply of seeded capture-heavy games + direct _get_threatened_pieces for both sides
    for g in range(4):
        env = chess_env.ChineseChess()
        plies = []
        for ply in range(60):
            before = state_row(env)
            legal = env.get_legal_moves()
            if not legal:
                break
            caps = [m for m in legal if env.board[m[2], m[3]] != 0]
            mv = rng.choice(caps) if (caps and rng.random() < 0.6) else rng.choice(legal)
            _, _, done = env.make_move(mv)
            plies.append(dict(board=before["board"].tolist(), player=before["player"], red_king=before["red_king"],
                              black_king=before["black_king"], move=enc(mv),
                              chase=[enc((a[0], a[1], b[0], b[1])) for a, b in env.chase_history[-1]]))
            if ply % 7 == 3:
                st = state_row(env)
                out["threats"].append(dict(board=st["board"].tolist(), player=st["player"], red_king=st["red_king"],
                                           black_king=st["black_king"],
                                           red=[enc((a[0], a[1], b[0], b[1])) for a, b in env._get_threatened_pieces(1)],
                                           black=[enc((a[0], a[1], b[0], b[1])) for a, b in env._get_threatened_pieces(-1)],
                                           current_player_after=int(env.current_player)))
            if done or env.red_king_pos is None or env.black_king_pos is None:
                break
        out["chase"].append(plies)
    # repetition: inject k copies of the hash the test will compute after the move (board after, NEXT player)
    for g in range(6):
        env = chess_env.ChineseChess()
        for _ in range(rng.randint(0, 12)):
            legal = env.get_legal_moves()
            env.make_move(rng.choice(legal))
        legal = env.get_legal_moves()
        quiet = [m for m in legal if env.board[m[2], m[3]] == 0] or legal
        mv = rng.choice(quiet)
        for k in (2, 3, 4):
            e2 = chess_env.ChineseChess()
            e2.board = env.board.copy(); e2.current_player = env.current_player; e2.move_count = env.move_count
            e2.red_king_pos, e2.black_king_pos = env.red_king_pos, env.black_king_pos
            e2.no_capture_count = env.no_capture_count
            # the position after the move, as the repetition test will hash it (next player to move)
            probe = chess_env.ChineseChess()
            probe.board = env.board.copy()
            probe.board[mv[2], mv[3]] = probe.board[mv[0], mv[1]]; probe.board[mv[0], mv[1]] = 0
            probe.current_player = -env.current_player
            h = probe._get_position_hash()
            e2.position_history = [h] * k
            before = state_row(e2)
            _, reward, done = e2.make_move(mv)
            code, side, cnt = reason_code(e2.end_reason)
            out["repetition"].append(dict(board=before["board"].tolist(), player=before["player"], move_count=before["move_count"],
                                          red_king=before["red_king"], black_king=before["black_king"],
                                          no_capture=before["no_capture"], move=enc(mv), copies=k,
                                          key_board=probe.board.reshape(90).astype(int).tolist(), key_player=int(probe.current_player),
                                          reward=float(reward), done=bool(done),
                                          winner=WINNER_NONE if e2.winner is None else int(e2.winner),
                                          reason=code, n_hist_after=len(e2.position_history),
                                          repetition_now=bool(e2._check_draw_by_repetition())))
    # predicates on the edge boards and on random positions
    boards = []
    for name, pieces, player, rk, bk in edge_boards():
        env = chess_env.ChineseChess()
        env.board = np.zeros((10, 9), dtype=np.int8)
        for (r, c), pc in pieces.items():
            env.board[r, c] = pc
        env.current_player = player
        env.red_king_pos, env.black_king_pos = rk, bk
        boards.append((name, env))
    for g in range(12):
        env = chess_env.ChineseChess()
        for _ in range(rng.randint(5, 60)):
            legal = env.get_legal_moves()
            if not legal or env.red_king_pos is None or env.black_king_pos is None:
                break
            env.make_move(rng.choice(legal))
        if env.red_king_pos is not None and env.black_king_pos is not None:
            boards.append(("random%d" % g, env))
    for name, env in boards:
        st = state_row(env)
        own = [(r, c) for r in range(10) for c in range(9) if env.board[r, c] * env.current_player > 0]
        cand = []
        for _ in range(30):
            if not own:
                break
            fr, fc = rng.choice(own)
            tr, tc = rng.randrange(10), rng.randrange(9)
            if (tr, tc) != (fr, fc):
                cand.append((fr, fc, tr, tc))
        cand += env.get_legal_moves()[:10]
        out["predicates"].append(dict(name=name, board=st["board"].tolist(), player=st["player"], red_king=st["red_king"],
                                      black_king=st["black_king"], checkmate=bool(env._check_checkmate()),
                                      stalemate=bool(env._check_stalemate()),
                                      suicide=[[enc(m), bool(env._is_move_suicide(*m))] for m in cand]))
    with open(os.path.join(OUT, "rules_extra.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("rules_extra: chase plies %d, threat probes %d (non-empty %d), repetition cases %d (draws %d), predicate boards %d (mates %d, stalemates %d)" % (
        sum(len(p) for p in out["chase"]), len(out["threats"]), sum(1 for t in out["threats"] if t["red"] or t["black"]),
        len(out["repetition"]), sum(1 for r in out["repetition"] if r["reason"] == 3), len(out["predicates"]),
        sum(p["checkmate"] for p in out["predicates"]), sum(p["stalemate"] for p in out["predicates"])))


# --------------------------------------------------------------------------- round 2: trainer-side consumers
def gen_trainer_io(seed=99):
    """(f-1) trainer.ReplayBuffer push / sample traces and the batch formation of trainer.py:313-321, (f-3)
    data/best_games.pkl written by Trainer._save_best_games and the structure of Trainer.save_model's checkpoint.
    trainer.py imports tensorboard at module level (not installed here): a stub SummaryWriter goes into
    sys.modules first; every reference function under test runs unmodified (the Trainer methods are called on a
    small stand-in object, so no 98 MB network is built or saved)."""
    import pickle
    import tempfile
    import types
    import torch
    stub = types.ModuleType("torch.utils.tensorboard")
    stub.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["torch.utils.tensorboard"] = stub
    with contextlib.redirect_stdout(io.StringIO()):
        import trainer
        import neural_network
    rng = np.random.RandomState(seed)
    # synthetic games in self_play_game's tuple format
    games = []
    for g in range(9):
        env = chess_env.ChineseChess()
        gd = []
        for ply in range(int(rng.randint(3, 14))):
            legal = env.get_legal_moves()
            pr = rng.dirichlet(np.ones(len(legal)))
            gd.append((env.board.copy(), {m: np.float64(p) for m, p in zip(legal, pr)}, float(np.round(rng.uniform(-1.2, 1.5), 6))))
            env.make_move(legal[int(rng.randint(len(legal)))])
        games.append(gd)
    buf = trainer.ReplayBuffer(max_size=40)                 # small capacity: the trace wraps around several times
    net = neural_network.ChessNet.__new__(neural_network.ChessNet)      # encode_board reads no weights
    trace = []
    for gi, gd in enumerate(games):
        buf.push(gd)
        if gi % 2 == 1:
            np.random.seed(1000 + gi)
            bs = min(8, len(buf))
            boards, probs, rewards = buf.sample(bs)
            states = np.stack([net.encode_board(b, 1) for b in boards]).astype(np.float32)     # trainer.py:313-317
            targets = torch.FloatTensor(rewards).unsqueeze(1).numpy()                            # trainer.py:319
            trace.append(dict(after_game=gi, size=len(buf), batch=bs, seed=1000 + gi,
                              boards=np.stack([np.asarray(b, np.int8).reshape(90) for b in boards]),
                              rewards=np.array(rewards, np.float64), states_bits=np.packbits(states.astype(np.uint8)),
                              targets=targets.astype(np.float32),
                              first_probs=[[enc(m), float(p)] for m, p in list(probs[0].items())]))
    pack = dict(n_games=np.array([len(games)]), capacity=np.array([40]))
    for gi, gd in enumerate(games):
        pack["g%d_boards" % gi] = np.stack([np.asarray(b, np.int8).reshape(90) for b, _, _ in gd])
        pack["g%d_z" % gi] = np.array([z for _, _, z in gd], np.float64)
        pack["g%d_nmoves" % gi] = np.array([len(p) for _, p, _ in gd], np.int32)
        pack["g%d_moves" % gi] = np.concatenate([np.array([enc(m) for m in p], np.int32) for _, p, _ in gd])
        pack["g%d_probs" % gi] = np.concatenate([np.array(list(p.values()), np.float64) for _, p, _ in gd])
    for ti, t in enumerate(trace):
        for k in ("boards", "rewards", "states_bits", "targets"):
            pack["t%d_%s" % (ti, k)] = t[k]
        pack["t%d_meta" % ti] = np.array([t["after_game"], t["size"], t["batch"], t["seed"]], np.int64)
        pack["t%d_first_probs" % ti] = np.array(t["first_probs"], np.float64)
    pack["n_traces"] = np.array([len(trace)])
    np.savez_compressed(os.path.join(OUT, "trainer_io.npz"), **pack)

    # best_games.pkl through the reference's own writer, from two reference games with the exact evaluator
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        best = []
        for seed_g, sims in ((2, 50), (0, 15)):
            np.random.seed(seed_g)
            with contextlib.redirect_stdout(io.StringIO()):
                gd, winner, reason = self_play.self_play_game(HashNet(), temperature=1.0, num_simulations=sims)
            best.append((gd, winner, len(gd), "训练"))
        fake = types.SimpleNamespace(total_games=200)
        with contextlib.redirect_stdout(io.StringIO()):
            trainer.Trainer._save_best_games(fake, best[:1])
            fake.total_games = 300
            trainer.Trainer._save_best_games(fake, best[1:])
        with open(os.path.join(tmp, "data", "best_games.pkl"), "rb") as f:
            blob = f.read()
        # checkpoint structure of Trainer.save_model (the file itself is ~280 MB: only its shape is recorded)
        small = types.SimpleNamespace(total_games=1000, training_steps=7)
        with contextlib.redirect_stdout(io.StringIO()):
            small.network = neural_network.ChessNet()
        small.optimizer = torch.optim.Adam(small.network.parameters(), lr=1e-3)
        with contextlib.redirect_stdout(io.StringIO()):
            trainer.Trainer.save_model(small)
        files = sorted(os.listdir(os.path.join(tmp, "models")))
        ck = torch.load(os.path.join(tmp, "models", "latest.pt"), map_location="cpu")
        struct = dict(files=files, keys=sorted(ck.keys()), total_games=int(ck["total_games"]), training_steps=int(ck["training_steps"]),
                      model={k: [list(v.shape), str(v.dtype)] for k, v in ck["model_state_dict"].items()},
                      optimizer_keys=sorted(ck["optimizer_state_dict"].keys()),
                      param_group_keys=sorted(ck["optimizer_state_dict"]["param_groups"][0].keys()),
                      n_param_ids=len(ck["optimizer_state_dict"]["param_groups"][0]["params"]))
    finally:
        os.chdir(cwd)
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(OUT, "best_games_ref.pkl"), "wb") as f:
        f.write(blob)
    with open(os.path.join(OUT, "checkpoint_struct.json"), "w") as f:
        json.dump(struct, f, separators=(",", ":"))
    games_ref = pickle.loads(blob)
    print("trainer_io: %d games pushed, %d sample traces; best_games_ref.pkl %d bytes, %d games (%s plies); checkpoint files %s" % (
        len(games), len(trace), len(blob), len(games_ref), [g["moves"] for g in games_ref], files))


if __name__ == "__main__":
    what = sys.argv[1:] or ["all"]
    if "all" in what:
        what = ["rules", "edge", "known", "puct", "sampler", "ztable", "net", "search"]
    for w in what:
        {"rules": gen_rules, "edge": gen_edge, "known": gen_known, "puct": gen_puct,
         "sampler": gen_sampler, "ztable": gen_ztable, "net": gen_net,
         "search": lambda: gen_search("fast"), "search_slow": lambda: gen_search("slow"),
         "rules_extra": gen_rules_extra, "trainer_io": gen_trainer_io}[w]()
