"""Soak test of the trunk kernel (run on the GPU box): many launches on fresh buffers and changing
sizes, a 16x16x32 build (36 / 39 = 2 / 4 boards per workgroup, -1 = the library's choice) against the 32x32x16 build (or
against another build: third argument) and against itself — any difference beyond bf16
rounding noise would be a synchronisation bug (both kernels are deterministic)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chinesechessai_amd import _lib
from chinesechessai_amd.neural_network import ChessNet, InferenceNet

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
va = int(sys.argv[2]) if len(sys.argv) > 2 else -1       # build under test (run twice) ...
vb = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # ... and the build it is compared with
t0, it, worst, nondet = time.time(), 0, 0.0, 0
last_print = t0
gen = torch.Generator().manual_seed(1)
while time.time() - t0 < budget:
    blocks = int(torch.randint(0, 7, (1,), generator=gen))
    G = int(torch.randint(1, 3000, (1,), generator=gen))
    torch.manual_seed(it)
    net = ChessNet(num_blocks=blocks).eval()
    inet = InferenceNet(net.cuda(), fused_tower=False)
    planes = torch.zeros(G, 10, 9, 16, device="cuda", dtype=torch.bfloat16)
    planes[..., :15] = (torch.rand(G, 10, 9, 15, device="cuda") < 0.15).to(torch.bfloat16)
    outs = []
    for variant in (va, vb, va):
        L.xq_tower_set_variant(variant)
        P = torch.empty(G, 2880, device="cuda", dtype=torch.bfloat16)
        V = torch.empty(G, 720, device="cuda", dtype=torch.bfloat16)
        _lib.check(L.xq_tower_nhwc_bf16(st, planes.data_ptr(), inet.hip_w[0].data_ptr(), inet.hip_wt.data_ptr(),
                                        inet.hip_bt.data_ptr(), inet.hip_hw.data_ptr(), inet.hip_hb.data_ptr(),
                                        P.data_ptr(), V.data_ptr(), G, blocks, None, None))
        outs.append((P, V))
    torch.cuda.synchronize()
    if not (torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])):
        nondet += 1
    scale = max(1.0, outs[1][0].float().abs().max().item())
    err = max((outs[0][0].float() - outs[1][0].float()).abs().max().item(),
              (outs[0][1].float() - outs[1][1].float()).abs().max().item()) / scale
    worst = max(worst, err)
    assert err <= 2 ** -6, (it, blocks, G, err)
    it += 1
    if time.time() - last_print > 60:                      # (a silent GPU command is taken to be hung after 7 minutes)
        last_print = time.time()
        print("  ... %d nets, worst %.4g, mismatches %d" % (it, worst, nondet), flush=True)
L.xq_tower_set_variant(-1)
print("soak (build %d vs build %d): %d nets, worst relative difference between the two builds %.4g, run-to-run mismatches %d"
      % (va, vb, it, worst, nondet))
assert nondet == 0
