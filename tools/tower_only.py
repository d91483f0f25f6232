"""Launch only the trunk kernel (for rocprofv3 --pmc / --kernel-trace passes on the GPU box).
usage: tower_only.py G blocks variant [launches] [stamp-entry 0/1]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chinesechessai_amd import _lib
from chinesechessai_amd.neural_network import ChessNet, InferenceNet

L = _lib.lib()
G, blocks, variant = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
use_stamp = len(sys.argv) > 5 and sys.argv[5] == "1"
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
inet = InferenceNet(ChessNet(num_blocks=blocks).eval().cuda())
planes = torch.zeros(G, 10, 9, 16, device="cuda", dtype=torch.bfloat16)
planes[..., :15] = (torch.rand(G, 10, 9, 15, device="cuda") < 0.15).to(torch.bfloat16)
P = torch.empty(G, 2880, device="cuda", dtype=torch.bfloat16)
V = torch.empty(G, 720, device="cuda", dtype=torch.bfloat16)
args = (st, planes.data_ptr(), inet.hip_w[0].data_ptr(), inet.hip_wt.data_ptr(), inet.hip_bt.data_ptr(),
        inet.hip_hw.data_ptr(), inet.hip_hb.data_ptr(), P.data_ptr(), V.data_ptr(), G, blocks)
L.xq_tower_set_variant(variant)
if use_stamp:
    fn = L.xq_tower_debug_stamps
    fn.argtypes = [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p]
    stamps = torch.zeros(((G + 1) // 2) * 64, dtype=torch.int64, device="cuda")
    for _ in range(n):
        fn(*args, stamps.data_ptr())
else:
    for _ in range(n):
        L.xq_tower_nhwc_bf16(*args, None, None)
torch.cuda.synchronize()
print("done", G, blocks, variant, n)
