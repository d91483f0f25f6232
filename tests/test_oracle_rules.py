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


def test_round2_rules_extras_vs_reference(golden_dir):
    """Round 2 fixtures (oracle/gen_golden.py rules_extra, generated by the unmodified reference): a8
    _get_threatened_pieces per ply (= chase_history entries) and for both sides on probe positions, the
    repetition draw with an injected position_history (chess_env.py:598-605), _check_checkmate /
    _check_stalemate / _is_move_suicide on the edge boards and random positions."""
    d = json.load(open(os.path.join(golden_dir, "rules_extra.json")))
    n_chase = 0
    for game in d["chase"]:
        for p in game:
            env = xo.OracleEnv()
            env.set_state(p["board"], p["player"], red_king=p["red_king"], black_king=p["black_king"])
            env.make_move(p["move"])
            # threats_after is taken BEFORE the side switch: by the mover (chess_env.py:344-348)
            assert env.threatened_pieces(p["player"]) == p["chase"], p
            n_chase += len(p["chase"])
    assert n_chase > 100
    for t in d["threats"]:
        env = xo.OracleEnv()
        env.set_state(t["board"], t["player"], red_king=t["red_king"], black_king=t["black_king"])
        assert env.threatened_pieces(1) == t["red"] and env.threatened_pieces(-1) == t["black"]
        assert env.e.current_player == t["current_player_after"]            # the temporary side switch is undone
    draws = 0
    for r in d["repetition"]:
        env = xo.OracleEnv()
        env.set_state(r["board"], r["player"], move_count=r["move_count"], red_king=r["red_king"],
                      black_king=r["black_king"], no_capture=r["no_capture"])
        env.inject_position_history(r["key_board"], r["key_player"], r["copies"])
        reward, done, _ = env.make_move(r["move"])
        w = env.e.winner
        assert (float(reward), bool(done), int(w), int(env.e.end_reason)) == (r["reward"], r["done"], r["winner"], r["reason"]), r
        assert env.e.n_hist == r["n_hist_after"] and env.check_draw_by_repetition() == r["repetition_now"]
        draws += r["reason"] == 3
        assert (r["reason"] == 3) == (r["copies"] >= 3)
    assert draws == 12
    mates = stales = 0
    for p in d["predicates"]:
        env = xo.OracleEnv()
        env.set_state(p["board"], p["player"], red_king=p["red_king"], black_king=p["black_king"])
        assert env.check_checkmate() == p["checkmate"] and env.check_stalemate() == p["stalemate"], p["name"]
        for mv, expect in p["suicide"]:
            assert env.is_move_suicide(mv) == expect, (p["name"], mv)
        mates += p["checkmate"]; stales += p["stalemate"]
    assert mates >= 1 and stales >= 1
