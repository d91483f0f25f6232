"""Upper bound of what bit-packed input planes could buy k_search_round: time it with and without the 2,880-B
plane rows (the run without them evaluates stale planes: timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chinesechessai_amd import _lib
from chinesechessai_amd.engine import SelfPlayEngine, TorchNetEvaluator
from chinesechessai_amd.neural_network import ChessNet
torch.manual_seed(0)
net = ChessNet(num_blocks=6).eval().cuda()
G = 16384
for with_planes in (True, False, True, False):
    ev = TorchNetEvaluator(net)
    eng = SelfPlayEngine(G, sims=50, planes_format=ev.planes_format, max_moves=6)
    if not with_planes:
        real = ev.planes_ptr
        ev.planes_ptr = lambda: None
    ev.bind(eng)
    eng.new_games(np.arange(G, dtype=np.uint32))
    eng.profile(True)
    for ply in range(6):
        eng.search(ev)
        _lib.check(eng.L.xq_engine_play_move(eng.h))
    torch.cuda.synchronize()
    p = eng.profile_read()
    print("planes %s: k_search_round %.1f us avg over %d launches" % (with_planes, p["search_ms"] / p["search_launches"] * 1e3, p["search_launches"]))
    eng.close()
