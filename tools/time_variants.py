"""Time trunk builds through the product entry point (no stamps), interleaved in one process, and check that each
gives the bits of the default build (run on the GPU box).  usage: time_variants.py V1 V2 ... [G=16384 via XQ_PROBE_G]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chinesechessai_amd import _lib
from chinesechessai_amd.neural_network import ChessNet, InferenceNet

L = _lib.lib()
G, blocks = int(os.environ.get("XQ_PROBE_G", "16384")), 6
variants = [int(v) for v in sys.argv[1:]] or [36, 39, 36, 39]
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
inet = InferenceNet(ChessNet(num_blocks=blocks).eval().cuda())
planes = torch.zeros(G, 10, 9, 16, device="cuda", dtype=torch.bfloat16)
planes[..., :15] = (torch.rand(G, 10, 9, 15, device="cuda") < 0.15).to(torch.bfloat16)
fl = 2.0 * G * 90 * (16 * 9 * 128 + 2 * blocks * 128 * 9 * 128 + 128 * 40)


def run(v, P, V):
    L.xq_tower_set_variant(v)
    return L.xq_tower_nhwc_bf16(st, planes.data_ptr(), inet.hip_w[0].data_ptr(), inet.hip_wt.data_ptr(), inet.hip_bt.data_ptr(),
                                inet.hip_hw.data_ptr(), inet.hip_hb.data_ptr(), P.data_ptr(), V.data_ptr(), G, blocks, None, None)


P0 = torch.empty(G, 2880, device="cuda", dtype=torch.bfloat16)
V0 = torch.empty(G, 720, device="cuda", dtype=torch.bfloat16)
_lib.check(run(36, P0, V0))
for v in variants:
    P = torch.empty(G, 2880, device="cuda", dtype=torch.bfloat16)
    V = torch.empty(G, 720, device="cuda", dtype=torch.bfloat16)
    if run(v, P, V) != 0:
        print("variant %d: not in this library" % v)
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        run(v, P, V)
    e0.record()
    for _ in range(20):
        run(v, P, V)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    same = torch.equal(P, P0) and torch.equal(V, V0)
    print("variant %2d: %.3f ms  %.1f TFLOP/s  %s" % (v, ms, fl / ms / 1e9, "== default build" if same else "differs from the default build"), flush=True)
L.xq_tower_set_variant(-1)
