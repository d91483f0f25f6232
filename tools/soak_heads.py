"""Run-to-run determinism + accuracy soak of the hand-written FC kernels (csrc/xq_policy.hip) and of the
whole InferenceNet forward: every launch on the same inputs must give bit-identical outputs; against an
fp32 torch evaluation of the same bf16 operands the error must stay within one bf16 rounding."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chinesechessai_amd import _lib
from chinesechessai_amd.neural_network import ChessNet, InferenceNet

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
bad = 0
for blocks, G, mode in ((1, 37, "reachable"), (2, 300, "all"), (6, 16384, "reachable"), (1, 1, "reachable"), (2, 4097, "reachable")):
    net = ChessNet(num_blocks=blocks).eval().cuda()
    inet = InferenceNet(net, policy_columns=mode)
    planes = torch.zeros(G, 10, 9, 16, device="cuda", dtype=torch.bfloat16)
    planes[..., :15] = (torch.rand(G, 10, 9, 15, device="cuda") < 0.15).to(torch.bfloat16)
    x = planes.permute(0, 3, 1, 2)
    hp, hv = inet.trunk_hip(x)
    hp0, hv0 = hp.clone(), hv.clone()
    lg0, v0 = inet._fc(hp0, hv0, G, None, None)
    lg0, v0 = lg0.clone(), v0.clone()
    torch.cuda.synchronize()
    # accuracy of the FC kernels on their own
    ref = (hp0.float() @ inet.pfw.float().t() + inet.hip_pfb).float()
    err = (lg0.float() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    h1 = torch.relu(hv0.float() @ inet.hip_v1w.float()[:, :720].t() + inet.hip_v1b)
    vref = torch.tanh(h1 @ inet.hip_v2w + inet.hip_v2b)
    verr = (v0.float() - vref).abs().max().item()
    n_fc = n_tr = 0
    for it in range(20):
        lg, v = inet._fc(hp0, hv0, G, None, None)
        if not (torch.equal(lg, lg0) and torch.equal(v, v0)):
            n_fc += 1
        hp2, hv2 = inet.trunk_hip(x)
        if not (torch.equal(hp2, hp0) and torch.equal(hv2, hv0)):
            n_tr += 1
    torch.cuda.synchronize()
    print("blocks %d G %5d %-9s: policy FC rel err %.3g, value abs err %.3g; mismatching repeats: FC %d/20, trunk %d/20" % (
        blocks, G, mode, err, verr, n_fc, n_tr))
    bad += n_fc + n_tr + (err > 2 ** -7) + (verr > 2 ** -7)
assert bad == 0
print("soak ok")
