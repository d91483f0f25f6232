"""Is a per-ply HIP graph worth building?  One ply of the bench workload (7 x {k_search_round, k_assign_rows, trunk, policy FC,
value head} + k_end_search + k_play_move; leaf dedupe off, because its round tag is still a kernel argument) captured with
torch.cuda.CUDAGraph on the engine's stream and replayed, against the same plies launched one by one; alternating, one
process.  usage (GPU box): probe_graph.py [plies=66]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chinesechessai_amd import _lib
from chinesechessai_amd.engine import SelfPlayEngine, TorchNetEvaluator
from chinesechessai_amd.neural_network import ChessNet

plies = int(sys.argv[1]) if len(sys.argv) > 1 else 66
torch.manual_seed(0)
net = ChessNet(num_blocks=6).eval().cuda()
G = 16384
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    ev = TorchNetEvaluator(net, leaf_dedupe=False)
    eng = SelfPlayEngine(G, sims=50, planes_format=ev.planes_format, stream=side.cuda_stream)
    eng._auto_carry(ev, None)
    eng._bind(ev)

    def one_ply():
        eng.search(ev)
        _lib.check(eng.L.xq_engine_play_move(eng.h))

    def fresh():
        eng.new_games(np.arange(G, dtype=np.uint32))
        one_ply(); one_ply()                     # (ply 0 has its round 0; from ply 1 on every ply is the same launch sequence)
        torch.cuda.synchronize()

    fresh()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        one_ply()
    torch.cuda.synchronize()
    for rep in range(3):
        for mode in ("launches", "graph"):
            fresh()
            t0 = time.time()
            for _ in range(plies):
                if mode == "graph":
                    g.replay()
                else:
                    one_ply()
            torch.cuda.synchronize()
            dt = time.time() - t0
            print("%-8s %d plies: %.2f ms per ply" % (mode, plies, dt / plies * 1e3), flush=True)
eng.close()
