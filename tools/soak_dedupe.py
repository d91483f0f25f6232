"""Soak of the leaf dedupe (run on the GPU box): random batch sizes (including sizes that are no multiple of 16), with and
without virtual loss (several pending slots per game), random seeds; after EVERY search round the device's row assignment is
compared with a host grouping of the planes the search kernel wrote: two slots share a row if and only if their planes are
equal, rows are numbered in the order of each group's lowest slot.  The table is filled by waves of all CUs at once
(dedupe_insert in k_search_round), so this is the test of its cross-XCD visibility protocol."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chinesechessai_amd import _lib
from chinesechessai_amd.engine import SelfPlayEngine, TorchNetEvaluator
from chinesechessai_amd.neural_network import ChessNet

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(0)
net = ChessNet(num_blocks=1).eval().cuda()
t0 = time.time()
runs = rounds = shared = 0
last = t0
while time.time() - t0 < budget:
    vl = bool(rng.integers(0, 3) == 0)
    G = int(rng.choice([int(rng.integers(1, 200)), int(rng.integers(200, 6000)), 16384]))
    if vl:
        G = min(G, 2048)
    S = int(rng.choice([16, 24, 50]))
    plies = int(rng.integers(3, 12))
    ev = TorchNetEvaluator(net, eval_cache=False)      # (row accounting of the dedupe alone)
    eng = SelfPlayEngine(G, sims=S, planes_format=ev.planes_format, max_moves=plies + 1)
    if vl:
        eng.set_virtual_loss(True)
    ev.bind(eng)
    assert eng.leaf_dedupe
    n = eng.n_rows
    # seeds from a small pool: games with the same seed stay identical for the whole game, others separate
    eng.new_games(rng.integers(0, max(2, G // int(rng.integers(1, 8))), size=G).astype(np.uint32))
    kind, a, v = _lib.EVAL_PRIORS, None, None
    for ply in range(plies):
        for r in range(eng.rounds):
            _lib.check(eng.L.xq_engine_search_round(eng.h, r, kind, a, v, ev.planes_ptr()))
            rows = eng.leaf_rows()
            planes = ev.storage.reshape(n, -1).view(torch.int16).cpu().numpy()
            pending = np.nonzero(rows >= 0)[0]
            first = {}
            for sl in pending:
                first.setdefault(planes[sl].tobytes(), int(sl))
            reps = sorted(first.values())
            row_of = {sl: i for i, sl in enumerate(reps)}
            want = np.full(n, -1, np.int32)
            for sl in pending:
                want[sl] = row_of[first[planes[sl].tobytes()]]
            assert np.array_equal(rows, want), ("row assignment differs", G, vl, S, ply, r)
            assert eng.row_history()[0][-1] == len(reps)
            rounds += 1
            shared += len(pending) - len(reps)
            kind, a, v = ev.evaluate(eng)
        _lib.check(eng.L.xq_engine_end_search(eng.h, kind, a, v))
        _lib.check(eng.L.xq_engine_play_move(eng.h))
        kind, a, v = _lib.EVAL_PRIORS, None, None
    eng.close()
    runs += 1
    if time.time() - last > 60:                                # (a silent GPU command is taken to be hung after 7 minutes)
        last = time.time()
        print("  ... %d batches, %d rounds" % (runs, rounds), flush=True)
print("leaf dedupe soak: %d batches, %d search rounds, every row assignment equal to the host grouping; %d leaves shared a row"
      % (runs, rounds, shared))
