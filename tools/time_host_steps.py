"""Where the non-kernel time of a step goes: host-side cost of xq_engine_new_games (MT19937 seeding + 70 doubles
per game on the host, H2D copy), pack_samples, the active-games poll."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chinesechessai_amd.engine import SelfPlayEngine, HashNetEvaluator
G = 16384
eng = SelfPlayEngine(G, sims=50)
seeds = np.arange(G, dtype=np.uint32)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time(); eng.new_games(seeds); torch.cuda.synchronize(); t1 = time.time()
    print("new_games: %.1f ms" % ((t1 - t0) * 1e3))
t0 = time.time(); n = eng.active_games(); t1 = time.time()
print("active_games poll: %.3f ms" % ((t1 - t0) * 1e3))
rec = torch.zeros(G * 70 * 576, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t0 = time.time(); eng.pack_samples(rec.data_ptr()); torch.cuda.synchronize(); t1 = time.time()
print("pack_samples: %.2f ms" % ((t1 - t0) * 1e3))
