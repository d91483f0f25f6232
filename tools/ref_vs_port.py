"""Build-container only (needs /root/reference): wall time of the reference's own Python self_play_game against
the C oracle port driving the SAME network object on CPU torch, identical seeds (the
ratio that translates bench.py's cpu_baseline, kind "port", into reference-equivalent time).
usage: python tools/ref_vs_port.py [sims] [n_games]"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("XQ_REFERENCE", "/root/reference"))
sys.dont_write_bytecode = True
import numpy as np
import torch

with contextlib.redirect_stdout(io.StringIO()):
    import neural_network as ref_nn
    import self_play as ref_sp
from chinesechessai_amd.chess_env import decode_move
from oracle import xq_oracle as xo

sims = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n_games = int(sys.argv[2]) if len(sys.argv) > 2 else 2
threads = int(os.environ.get("OMP_NUM_THREADS", "8"))
torch.set_num_threads(threads)
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    net = ref_nn.ChessNet()
net.eval()


def fn(ctx, nrows, boards, players, moves, nmoves, priors, values):
    rows = []
    for i in range(nrows):
        b = np.array([boards[i * 90 + k] for k in range(90)], dtype=np.int8).reshape(10, 9)
        rows.append((b, int(players[i]), [decode_move(moves[i * 128 + j]) for j in range(nmoves[i])]))
    for i, (d, v) in enumerate(net.predict_batch(rows)):
        for j, p in enumerate(d.values()):
            priors[i * 128 + j] = float(p)
        values[i] = float(v)
    return 0


ev = xo.Evaluator(xo.EVAL_FN(fn), None)
t_ref = t_port = 0.0
same = 0
for seed in range(n_games):
    np.random.seed(seed)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        gd, winner, reason = ref_sp.self_play_game(net, temperature=1.0, num_simulations=sims)
    t_ref += time.time() - t0
    t0 = time.time()
    rc, og = xo.self_play_game(seed, sims, eval_red=ev)
    t_port += time.time() - t0
    ok = rc == 0 and og.n_samples == len(gd) and all(
        np.array_equal(np.frombuffer(og.s_board[i], dtype=np.int8).reshape(10, 9), gd[i][0]) for i in range(len(gd)))
    same += bool(ok)
    print("seed %d: reference %.1f s (%d plies), port %.1f s, same game: %s" % (seed, t_ref, len(gd), t_port, ok), flush=True)
print("sims %d, %d games, %d torch threads: reference %.2f s/game, C port + same net %.2f s/game, ratio %.1fx; identical games %d/%d" % (
    sims, n_games, threads, t_ref / n_games, t_port / n_games, t_ref / t_port, same, n_games))
