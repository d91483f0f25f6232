"""Evaluation cache in VERIFY mode on the bench workload (run on the GPU box): every leaf the cache could answer is
evaluated all the same and compared with the entry.  usage: verify_eval_cache.py [plies=8] [G=16384] [peaked] [refill]
peaked: the policy head of the seeded random-init network x 256 (bench.py's stand-in for a trained network: the cache then
answers a third of the leaves instead of 2 %); refill: 2 G games through the G slots (play_refill)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chinesechessai_amd import _lib
from chinesechessai_amd.engine import SelfPlayEngine, TorchNetEvaluator
from chinesechessai_amd.neural_network import ChessNet

plies = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
torch.manual_seed(0)
net = ChessNet(num_blocks=6).eval().cuda()
peaked, refill = "peaked" in sys.argv[3:], "refill" in sys.argv[3:]
if peaked:
    with torch.no_grad():
        net.policy_fc.weight.mul_(256.0)
        net.policy_fc.bias.mul_(256.0)
print("plies %d, G %d%s%s" % (plies, G, ", peaked priors" if peaked else "", ", refill mode (2 G games)" if refill else ""), flush=True)
seeds = np.arange(G, dtype=np.uint32)
for dedupe, carry in ((True, True), (False, False), (True, False)):
    res = []
    for rep in range(2):
        ev = TorchNetEvaluator(net, leaf_dedupe=dedupe, eval_cache="verify")
        eng = SelfPlayEngine(G, sims=50, planes_format=ev.planes_format, max_moves=plies)
        if not carry:
            eng.set_root_eval_carry(False)
        if refill:
            rec = torch.zeros(2 * G * _lib.MAX_PLIES * _lib.SAMPLE_RECORD_BYTES, dtype=torch.uint8, device="cuda")
            out, _ = eng.play_refill(ev, np.arange(2 * G, dtype=np.uint32), rec.data_ptr())
            b = type("R", (), {"chosen": rec.cpu().numpy(), "s_counts": np.asarray(out["n_plies"])})()
            del rec
        else:
            b = eng.play(ev, seeds)
        print("dedupe %s carry %s rep %d: compared %d, fills %d, MISMATCHES %d" % ((dedupe, carry, rep) + eng.eval_cache_stats()), flush=True)
        eng.close()
        res.append(b)
    print("   run-to-run equal:", np.array_equal(res[0].chosen, res[1].chosen) and np.array_equal(res[0].s_counts, res[1].s_counts))
