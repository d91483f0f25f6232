"""CPU suite: oracle search / sampling / driver vs golden vectors from the reference.  Bit-exact."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from oracle import xq_oracle as xo


def bits(x):
    return struct.pack("<d", float(x))


def test_puct_selection(golden_dir):
    """G5: MCTSNode.select_child (self_play.py:40-59) on 2000 hand-built nodes with near-ties:
    float32 stepwise arithmetic, first maximum wins."""
    rows = json.load(open(os.path.join(golden_dir, "puct.json")))
    L = xo.lib()
    for r in rows:
        best, best_s = -1, -np.inf
        for j, (v, w, p) in enumerate(zip(r["visits"], r["wsum"], r["prior"])):
            s = L.xqo_puct_score(w, v, p, r["N"])
            if s > best_s:
                best, best_s = j, s
        assert best == r["best"], r


def test_sampler(golden_dir):
    """G6: MT19937 stream, np.random.choice(n, p), float64 pairwise sum, counts**(1/T)/sum."""
    g = json.load(open(os.path.join(golden_dir, "sampler.json")))
    L = xo.lib()
    mt = (C.c_uint32 * 624)()
    idx = C.c_int()
    L.xqo_mt_seed(mt, C.byref(idx), 0)
    assert [L.xqo_mt_double(mt, C.byref(idx)) for _ in range(3)] == g["seed0_first3"]
    assert g["seed0_first3"][0] == 0.5488135039273248
    for s in g["sums"]:
        a = np.array(s["a"], np.float64)
        got = L.xqo_np_sum(a.ctypes.data_as(C.POINTER(C.c_double)), len(a))
        assert bits(got) == bits(s["s"]), len(a)
    for c in g["cases"]:
        L.xqo_mt_seed(mt, C.byref(idx), c["seed"])
        us = [L.xqo_mt_double(mt, C.byref(idx)) for _ in range(3)]
        assert us == c["uniforms"]
        # counts ** (1/T) / sum restated with the fixture's pow table
        pw = np.array(c["pow"], np.float64)
        ssum = L.xqo_np_sum(pw.ctypes.data_as(C.POINTER(C.c_double)), len(pw))
        p = pw / ssum
        assert [bits(x) for x in p] == [bits(x) for x in c["p"]]
        for u, d in zip(us, c["draws"]):
            got = L.xqo_choice_from_uniform(p.ctypes.data_as(C.POINTER(C.c_double)), len(p), u)
            assert got == d, c["seed"]
        if c["T"] in (1.0, 0.5):     # exact powers: C pow() must agree with NumPy's
            inv = 1.0 / c["T"]
            assert [bits(float(x) ** inv) for x in c["counts"]] == [bits(x) for x in c["pow"]]


def test_z_table(golden_dir):
    """G7: z assignment of self_play_game (self_play.py:266-310) for wins / losses / draws of
    lengths 7..70, with the 0.01 x immediate-reward term."""
    games = json.load(open(os.path.join(golden_dir, "ztable.json")))
    L = xo.lib()
    seen = set()
    for g in games:
        for i, (pl, z) in enumerate(zip(g["players"], g["z"])):
            has = i < len(g["step_rewards"])
            got = L.xqo_z_value(g["winner"], pl, g["length"], int(has), g["step_rewards"][i] if has else 0.0)
            assert bits(got) == bits(z), (g["name"], i)
        seen.add((g["winner"], g["length"] >= 60, g["length"] <= 30, g["length"] <= 50))
    assert len(seen) >= 7


def _check_game(r, pow_table=None):
    rc, g = xo.self_play_game(r["seed"], r["sims"], temperature=r["T"],
                              eval_black=_salted(1) if r["opponent"] else None, pow_table=pow_table)
    if r["error"]:
        assert rc == 1 and "NaN" in r["error"]
        return
    assert rc == 0
    assert g.n_plies == len(r["moves"])
    for ply in range(g.n_plies):
        mv = [int(g.t_moves[ply][j]) for j in range(g.t_nchild[ply])]
        vs = [int(g.t_visits[ply][j]) for j in range(g.t_nchild[ply])]
        assert mv == [m for m, _ in r["visits"][ply]], (r["seed"], r["sims"], ply)
        assert vs == [v for _, v in r["visits"][ply]], (r["seed"], r["sims"], ply)
        assert g.t_move[ply] == r["moves"][ply], (r["seed"], r["sims"], ply)
        assert bits(g.t_reward[ply]) == bits(r["rewards"][ply])
    assert g.winner == r["winner"] and g.end_reason == r["reason"]
    assert g.n_samples == r["n_samples"]
    for i in range(g.n_samples):
        assert bits(g.s_z[i]) == bits(r["z"][i]), i
        assert [int(g.s_moves[i][j]) for j in range(g.s_nmoves[i])] == r["pi_moves"][i]
        assert [bits(g.s_probs[i][j]) for j in range(g.s_nmoves[i])] == [bits(x) for x in r["pi"][i]], i
    import zlib
    crc = 0
    for i in range(g.n_samples):
        crc = zlib.crc32(bytes(bytearray(np.frombuffer(g.s_board[i], dtype=np.uint8))), crc)
    assert crc == r["boards_crc"]


_keep = []


def _salted(salt):
    """HashNet with a salt byte (second 'network' of the arena mode), as a Python callback."""
    import zlib

    def fn(ctx, nrows, boards, players, moves, nmoves, priors, values):
        for i in range(nrows):
            b = bytes(bytearray((boards[i * 90 + k] & 0xff) for k in range(90)))
            h0 = zlib.crc32(b + bytes([players[i] & 0xff]) + bytes([salt]))
            for j in range(nmoves[i]):
                m = moves[i * 128 + j]
                f, t = divmod(m, 90)
                h = zlib.crc32(bytes([f // 9, f % 9, t // 9, t % 9]), h0)
                priors[i * 128 + j] = ((h >> 8) % 64 + 1) / 1024
            values[i] = ((h0 >> 4) % 65 - 32) / 64
        return 0

    cb = xo.EVAL_FN(fn)
    _keep.append(cb)
    return xo.Evaluator(cb, None)


def _pow_table(T, sims):
    if T < 0.01:
        return None
    return np.arange(sims + 1, dtype=np.int64) ** (1.0 / T)


def test_selfplay_games_hashnet(golden_dir):
    """G4: whole games through the reference's self_play_game with the exact HashNet evaluator:
    per-ply root visit dicts, sampled moves, rewards, pi, z, outcome, board CRC."""
    path = os.path.join(golden_dir, "search_hashnet.json")
    games = json.load(open(path))
    assert len(games) >= 16
    for r in games:
        _check_game(r, _pow_table(r["T"], r["sims"]))
    crcs = {(r["seed"], r["sims"]): "%08x" % r["boards_crc"] for r in games if r["T"] == 1.0 and not r["opponent"]}
    # SURVEY.md §8c G4 published CRCs for S=50
    assert crcs[(0, 50)] == "ce133970" and crcs[(1, 50)] == "d60ebdc7"


def test_selfplay_games_hashnet_slow(golden_dir):
    path = os.path.join(golden_dir, "search_hashnet_slow.json")
    if not os.path.exists(path):
        pytest.skip("slow fixture not generated")
    for r in json.load(open(path)):
        _check_game(r, _pow_table(r["T"], r["sims"]))
