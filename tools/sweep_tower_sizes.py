"""Launch-size sweep of the two trunk builds (run on the GPU box): k_tower16b with 2 boards per workgroup (two
workgroups per CU, variant 36) against 4 boards per workgroup (one per CU, variant 39).  Decides the size threshold of
the automatic selection in xq_tower.hip (launch_tower).  Back-to-back launches (the chip at its sustained clock) and single launches
separated by a synchronisation + 2 ms of idle (boost clock)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chinesechessai_amd import _lib
from chinesechessai_amd.neural_network import ChessNet, InferenceNet

L = _lib.lib()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
VB = int(sys.argv[2]) if len(sys.argv) > 2 else 39      # the build compared with variant 36
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
inet = InferenceNet(ChessNet(num_blocks=blocks).eval().cuda())
for G in (64, 256, 512, 1024, 1536, 2048, 3072, 4096, 8192, 16384):
    planes = torch.zeros(G, 10, 9, 16, device="cuda", dtype=torch.bfloat16)
    planes[..., :15] = (torch.rand(G, 10, 9, 15, device="cuda") < 0.15).to(torch.bfloat16)
    P = torch.empty(G, 2880, device="cuda", dtype=torch.bfloat16)
    V = torch.empty(G, 720, device="cuda", dtype=torch.bfloat16)
    args = (st, planes.data_ptr(), inet.hip_w[0].data_ptr(), inet.hip_wt.data_ptr(), inet.hip_bt.data_ptr(),
            inet.hip_hw.data_ptr(), inet.hip_hb.data_ptr(), P.data_ptr(), V.data_ptr(), G, blocks)
    res = {}
    for rep in range(2):
        for variant in (36, VB):
            L.xq_tower_set_variant(variant)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = max(20, 20 * 4096 // G)
            for _ in range(3):
                L.xq_tower_nhwc_bf16(*args, None, None)
            e0.record()
            for _ in range(n):
                L.xq_tower_nhwc_bf16(*args, None, None)
            e1.record()
            torch.cuda.synchronize()
            b2b = e0.elapsed_time(e1) / n
            single = []
            for _ in range(10):
                time.sleep(0.002)
                e0.record()
                L.xq_tower_nhwc_bf16(*args, None, None)
                e1.record()
                torch.cuda.synchronize()
                single.append(e0.elapsed_time(e1))
            res.setdefault(variant, []).append((b2b, sorted(single)[len(single) // 2]))
    print("G=%6d  16b: back-to-back %.4f / %.4f ms, single %.4f ms | 16s: back-to-back %.4f / %.4f ms, single %.4f ms | 16s/16b %.3f"
          % (G, res[36][0][0], res[36][1][0], res[36][1][1], res[VB][0][0], res[VB][1][0], res[VB][1][1],
             min(x[0] for x in res[VB]) / min(x[0] for x in res[36])), flush=True)
L.xq_tower_set_variant(-1)
