"""CPU suite: the C oracle (oracle/xq_oracle.c) against golden vectors captured from the
unmodified reference (tests/golden/, generator: oracle/gen_golden.py).  Bit-exact."""
import json
import os
import struct

import numpy as np
import pytest

from oracle import xq_oracle as xo


def bits(x):
    return struct.pack("<d", float(x))


@pytest.fixture(scope="module")
def rules(golden_dir):
    return np.load(os.path.join(golden_dir, "rules_random.npz"))


def test_rules_random_trajectories(rules):
    """G1+G2: legal-move lists (order included) and every make_move output along 48 seeded games,
    including plies played after `done` (Appendix A9)."""
    d = rules
    env = xo.OracleEnv()
    n = len(d["game"])
    cur = -1
    for i in range(n):
        if d["game"][i] != cur:
            cur = d["game"][i]
            env.reset()
        e = env.e
        assert np.array_equal(env.board().reshape(90), d["board"][i]), i
        assert e.current_player == d["player"][i]
        assert e.move_count == d["move_count"][i]
        assert e.red_king == d["red_king"][i] and e.black_king == d["black_king"][i]
        assert e.no_capture_count == d["no_capture"][i]
        assert e.consecutive_checks == d["consecutive_checks"][i]
        assert e.winner == d["winner"][i]
        legal = env.legal_moves()
        assert legal == d["legal"][i][: d["nlegal"][i]].tolist(), i
        reward, done, chk = env.make_move(d["move"][i])
        assert bits(reward) == bits(d["reward"][i]), (i, reward, d["reward"][i])
        assert done == bool(d["done"][i]), i
        assert chk == bool(d["is_check"][i]), i
        e = env.e
        assert e.winner == d["winner_after"][i], i
        if done:
            assert e.end_reason == d["reason"][i], i
            if d["reason"][i] in (1, 2, 5, 6):
                assert e.end_side == d["reason_side"][i], i
            if d["reason"][i] == 8:
                assert e.end_count == d["reason_count"][i], i
        assert e.consecutive_checks == d["cc_after"][i]
        assert e.no_capture_count == d["nc_after"][i]
        assert e.red_king == d["rk_after"][i] and e.black_king == d["bk_after"][i]
        assert e.n_hist == d["n_hist"][i]


def test_rules_edge_boards(golden_dir):
    """Hand-built boards: the reference's own unit-test positions (stale king caches included)
    and the Appendix-A quirk cases."""
    cases = json.load(open(os.path.join(golden_dir, "rules_edge.json")))
    assert len(cases) >= 30
    L = xo.lib()
    for c in cases:
        env = xo.OracleEnv()
        env.set_state(c["board"], c["player"], red_king=c["red_king"], black_king=c["black_king"])
        assert env.legal_moves() == c["legal"], c["name"]
        assert bool(L.xqo_is_in_check(env.p, 1)) == c["in_check_red"], c["name"]
        assert bool(L.xqo_is_in_check(env.p, -1)) == c["in_check_black"], c["name"]
        assert bool(L.xqo_are_kings_facing(env.p)) == c["facing"], c["name"]
        for m in c["moves"]:
            env.set_state(c["board"], c["player"], red_king=c["red_king"], black_king=c["black_king"])
            reward, done, chk = env.make_move(m["move"])
            assert bits(reward) == bits(m["reward"]), (c["name"], m)
            assert done == m["done"] and chk == m["is_check"], (c["name"], m)
            assert env.e.winner == m["winner"], (c["name"], m)
            if done:
                assert env.e.end_reason == m["reason"], (c["name"], m)
            assert env.e.red_king == m["rk"] and env.e.black_king == m["bk"]


def test_known_answers(golden_dir):
    """G3: 44 initial moves in order; the 7-ply double-cannon mate; 10-of-12 perpetual check."""
    k = json.load(open(os.path.join(golden_dir, "known.json")))
    env = xo.OracleEnv()
    assert env.legal_moves() == k["initial_moves"]
    assert len(k["initial_moves"]) == 44
    assert xo.decode_move(k["initial_moves"][0]) == (6, 0, 5, 0)
    assert xo.decode_move(k["initial_moves"][-1]) == (9, 8, 7, 8)
    for mv, r, d in zip(k["mate_line"], k["mate_rewards"], k["mate_dones"]):
        reward, done, _ = env.make_move(mv)
        assert bits(reward) == bits(r) and done == d
    assert env.e.winner == 1 and env.e.end_reason == xo.R_CHECKMATE and env.e.end_side == -1
    assert k["mate_rewards"] == [0.03, 0.025, 2.0, 0.1, 0.04, 0.0, 200.0]


def test_perpetual_check_injected(golden_dir):
    """test_perpetual_rules.py:20-50 style: the predicate on an injected check_history is
    observable through make_move: a quiet move with >= 10 checks in the last 12 entries
    (the new entry included) ends the game with reason 6."""
    k = json.load(open(os.path.join(golden_dir, "known.json")))
    for case in k["perpetual"]:
        hist = case["hist"]
        # inject all but the last entry, then play a quiet first move whose is_check matches it
        if hist[-1] != 0:
            continue
        env = xo.OracleEnv()
        for i, h in enumerate(hist[:-1]):
            env.e.check_hist[i] = h
        env.e.n_check = len(hist) - 1
        reward, done, chk = env.make_move(xo.encode_move((6, 0, 5, 0)))
        assert chk is False
        assert done == case["result"], case
        if done:
            assert env.e.end_reason == xo.R_PERP_CHECK and reward == -10
            # A4: the side that just moved (red) wins; the side to move is named
            assert env.e.winner == 1 and env.e.end_side == -1
