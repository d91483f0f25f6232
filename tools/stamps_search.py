"""Where a k_search_round wave spends its cycles (probes library: XQ_TOWER_PROBES=1).  Plays a few plies of the bench
workload, then stamps one round >= 1 of a mid-game ply and prints the mean cycles per phase over all games whose wave ran
the whole path (run on the GPU box).  usage: stamps_search.py [plies=6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chinesechessai_amd import _lib
from chinesechessai_amd.engine import SelfPlayEngine, TorchNetEvaluator
from chinesechessai_amd.neural_network import ChessNet

plies = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.manual_seed(0)
net = ChessNet(num_blocks=6).eval().cuda()
G = 16384
ev = TorchNetEvaluator(net, eval_cache=False)
eng = SelfPlayEngine(G, sims=50, planes_format=ev.planes_format)
eng._auto_carry(ev, None)
eng._bind(ev)
eng.new_games(np.arange(G, dtype=np.uint32))
L = eng.L
buf = torch.zeros(G * 16, dtype=torch.int64, device="cuda")
names = ["start -> game record", "consume_eval", "root board", "descent", "leaf flags", "replay", "make_move", "record + planes", "dedupe insert"]
for ply in range(plies):
    kind, a, v = _lib.EVAL_PRIORS, None, None
    pp = ev.planes_ptr()
    for r in range(eng.rounds):
        stamp = ply == plies - 1 and r in (1, 3, 5)
        if stamp:
            buf.zero_()
            torch.cuda.synchronize()
            assert L.xq_engine_set_search_stamps(buf.data_ptr()) == 0, "not a probes library"
        _lib.check(L.xq_engine_search_round(eng.h, r, kind, a, v, pp))
        if stamp:
            torch.cuda.synchronize()
            L.xq_engine_set_search_stamps(None)
            s = buf.cpu().numpy().reshape(G, 16).astype(np.int64)
            full = (s[:, :10] > 0).all(axis=1)
            d = np.diff(s[full, :10], axis=1)
            print("ply %d round %d: %d of %d waves ran the whole path; mean cycles per phase (s_memtime = 100 MHz ticks? see total):" % (ply, r, full.sum(), G))
            for n, m, md in zip(names, d.mean(axis=0), np.median(d, axis=0)):
                print("   %-22s mean %8.0f  median %8.0f" % (n, m, md))
            print("   %-22s mean %8.0f" % ("whole wave", (s[full, 9] - s[full, 0]).mean()))
        kind, a, v = ev.evaluate(eng)
    _lib.check(L.xq_engine_end_search(eng.h, kind, a, v))
    _lib.check(L.xq_engine_play_move(eng.h))
torch.cuda.synchronize()
eng.close()
