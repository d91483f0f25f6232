"""Phase stamps of k_tower1wa (variant 60, stamped build; run on the GPU box): cycles per phase of a workgroup, the clock
it held, the launch's wall time; then the product entry point against the default build, interleaved in one process.
usage: stamps_1wa.py [G=16384] [blocks=6]
Stamp slots: 0 start, 1 input convolution done, 2 its epilogue + barrier, then per block b: 3+4b = taps 1..8 of the first
convolution, 4+4b = its epilogue under tap 0 of the second, 5+4b = taps 1..8 of the second, 6+4b = its epilogue under tap 0
of the next block's first (last block: the plain epilogue), 59 = behind the assembly body, 60 = heads' MFMAs, 61 = end."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chinesechessai_amd import _lib
from chinesechessai_amd.neural_network import ChessNet, InferenceNet

L = _lib.lib()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
inet = InferenceNet(ChessNet(num_blocks=blocks).eval().cuda())
planes = torch.zeros(G, 10, 9, 16, device="cuda", dtype=torch.bfloat16)
planes[..., :15] = (torch.rand(G, 10, 9, 15, device="cuda") < 0.15).to(torch.bfloat16)
P = torch.empty(G, 2880, device="cuda", dtype=torch.bfloat16)
V = torch.empty(G, 720, device="cuda", dtype=torch.bfloat16)
args = (st, planes.data_ptr(), inet.hip_w[0].data_ptr(), inet.hip_wt.data_ptr(), inet.hip_bt.data_ptr(),
        inet.hip_hw.data_ptr(), inet.hip_hb.data_ptr(), P.data_ptr(), V.data_ptr(), G, blocks)
fl = 2.0 * G * 90 * (16 * 9 * 128 + 2 * blocks * 128 * 9 * 128 + 128 * 40)


def timeit(fn, it=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


nwg = (G + 3) // 4
stamps = torch.zeros(((G + 1) // 2) * 64, dtype=torch.int64, device="cuda")
ABL_NAMES = {61: "drains read VGPRs, not accumulators", 62: "no stores", 63: "no drain", 64: "no weight DMA", 65: "no barriers",
             66: "no drain, DMA, barriers", 67: "Winograd probe: 2 extra VALU per MFMA", 68: "3 extra VALU per MFMA", 69: "4 extra VALU per MFMA",
             70: "3 extra VALU per MFMA, 3.5 x the weight DMA", 71: "4 extra VALU per MFMA, 3.5 x the weight DMA",
             72: "46 of the 48 MFMAs per K-step (pixel tile 5 without 2 of its 8 channel tiles)", 73: "44 of the 48 MFMAs per K-step"}
extra = [int(v) for v in os.environ.get("XQ_1WA_ABL", "").split(",") if v]
for variant in [60] + extra + [39]:
    L.xq_tower_set_variant(variant)
    if L.xq_tower_debug_stamps(*args, stamps.data_ptr()) != 0:
        print("variant %d: no stamped build" % variant)
        continue
    ms = timeit(lambda: L.xq_tower_debug_stamps(*args, stamps.data_ptr()), it=10)
    s = stamps.cpu().numpy().reshape(-1, 64)[:nwg].astype(np.float64)
    s = s[s[:, 61] > 0]
    tot = s[:, 61] - s[:, 0]
    rt = s[:, 63] - s[:, 62]
    print("variant %d stamped: %.3f ms; workgroup %d cycles (median), clock %.3f GHz, workgroup wall %.1f us, launch wall %.1f us" % (
        variant, ms, np.median(tot), np.median(tot / rt * 0.1), np.median(rt) / 100.0, (s[:, 63].max() - s[:, 62].min()) / 100.0))
    d = lambda a, b: int(np.median(s[:, a] - s[:, b]))
    if variant in ABL_NAMES:
        print("  (timing-only body: %s)" % ABL_NAMES[variant])
    if variant >= 60:
        print("  input conv %d, its epilogue + first DMA %d" % (d(1, 0), d(2, 1)))
        prev = 2
        for b in range(blocks):
            k = 3 + 4 * b
            print("  block %d: [tap 0 +] taps 1..8 %d | epilogue under tap 0 %d | taps 1..8 %d | %s %d" % (
                b, d(k, prev), d(k + 1, k), d(k + 2, k + 1), "epilogue under tap 0" if b < blocks - 1 else "last epilogue", d(k + 3, k + 2)))
            prev = k + 3
        print("  behind the body %d, heads %d, stores %d" % (d(59, prev), d(60, 59), d(61, 60)))
        if s[:, 40].max() > 0 and blocks >= 2:
            kx1, kx2 = 3 + 4 * (blocks - 1), 3 + 4 * (blocks - 2) + 2
            print("  inside the last block's first boundary (last weight stage + drains 0, 1 | drain-0 rest, group (0,0) + drain 1 | 3 groups + drain 2 | 5 groups + drain 3 | 7 groups, prefetch): %d | %d | %d | %d | %d" % (
                d(40, kx1), d(41, 40), d(42, 41), d(43, 42), d(kx1 + 1, 43)))
            print("  inside the last-but-one block's second boundary: %d | %d | %d | %d | %d" % (
                d(45, kx2), d(46, 45), d(47, 46), d(48, 47), d(kx2 + 1, 48)))
    else:
        nl = 2 * blocks
        print("  input conv %d + %d | main loops %s | epilogues %s | heads %d + %d" % (
            d(1, 0), d(2, 1), [d(3 + 2 * l, 2 + 2 * l) for l in range(nl)], [d(4 + 2 * l, 3 + 2 * l) for l in range(nl)],
            d(60, 2 + 2 * nl), d(61, 60)))
P0 = torch.empty_like(P)
V0 = torch.empty_like(V)
L.xq_tower_set_variant(39)
L.xq_tower_nhwc_bf16(st, planes.data_ptr(), *args[2:7], P0.data_ptr(), V0.data_ptr(), G, blocks, None, None)
for variant in (39, 60, 39, 60, 36, 60):
    L.xq_tower_set_variant(variant)
    ms = timeit(lambda: L.xq_tower_nhwc_bf16(*args, None, None))
    print("variant %2d: %.3f ms  %.1f TFLOP/s = %.3f of 2.5 PFLOP/s  %s" % (
        variant, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0, "== 4-board default" if torch.equal(P, P0) and torch.equal(V, V0) else "DIFFERS"), flush=True)
L.xq_tower_set_variant(-1)
