"""Compare a trunk build with k_tower16b<NB = 2> bit for bit, twice, over block counts and batch sizes (run on the GPU
box): usage compare_trunk_builds.py VARIANT   (39 = 4 boards per workgroup, 0 = 32x32x16, 50 = k_tower1w in a probes build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chinesechessai_amd import _lib
from chinesechessai_amd.neural_network import ChessNet, InferenceNet

L = _lib.lib()
v = int(sys.argv[1])
st = torch.cuda.current_stream().cuda_stream
for blocks in (0, 1, 2, 6):
    torch.manual_seed(10 + blocks)
    inet = InferenceNet(ChessNet(num_blocks=blocks).eval().cuda())
    for G in (1, 2, 4, 5, 37, 1024):
        planes = torch.zeros(G, 10, 9, 16, device="cuda", dtype=torch.bfloat16)
        planes[..., :15] = (torch.rand(G, 10, 9, 15, device="cuda") < 0.15).to(torch.bfloat16)
        outs = {}
        for var in (36, v, v):
            L.xq_tower_set_variant(var)
            P = torch.full((G, 2880), 9.0, device="cuda", dtype=torch.bfloat16)
            V = torch.full((G, 720), 9.0, device="cuda", dtype=torch.bfloat16)
            rc = L.xq_tower_nhwc_bf16(st, planes.data_ptr(), inet.hip_w[0].data_ptr(), inet.hip_wt.data_ptr(), inet.hip_bt.data_ptr(),
                                      inet.hip_hw.data_ptr(), inet.hip_hb.data_ptr(), P.data_ptr(), V.data_ptr(), G, blocks, None, None)
            torch.cuda.synchronize()
            outs.setdefault(var, []).append((rc, P.float(), V.float()))
        ref = outs[36][0]
        a, b = outs[v]
        dP = (a[1] - ref[1]).abs()
        dV = (a[2] - ref[2]).abs()
        bad_boards = (dP.max(dim=1).values > 0).nonzero().flatten().tolist()[:8]
        print("blocks %d G %4d: rc %d  policy max diff %.3g (%d elems differ, boards %s)  value max diff %.3g  run-to-run equal %s" % (
            blocks, G, a[0], dP.max().item(), int((dP > 0).sum()), bad_boards, dV.max().item(),
            torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])), flush=True)
L.xq_tower_set_variant(-1)
