"""Trunk builds interleaved in one process through the stamp entry point (run on the GPU box): wall time, workgroup
cycles, the clock the chip holds, and (XQ_PROBE_PHASES=1) the cycles of every phase.  usage: probe_tiles.py V1 V2 ...
(36 / 39 = k_tower16b with 2 / 4 boards per workgroup, 0 = 32x32x16; 50, 30, 31, 41, 43 need a library built with
XQ_TOWER_PROBES=1)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
# the ablation / option builds exist only in a library compiled with XQ_TOWER_PROBES=1 (build it in the container:
# XQ_TOWER_PROBES=1 python -c "from chinesechessai_amd import _lib; _lib.build()", the .so travels to the GPU box)
from chinesechessai_amd import _lib
from chinesechessai_amd.neural_network import ChessNet, InferenceNet

L = _lib.lib()
G, blocks = int(os.environ.get("XQ_PROBE_G", "16384")), 6
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
fn = L.xq_tower_debug_stamps
fn.argtypes = [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p]
stamps = torch.zeros((G + 1) // 2 * 64, dtype=torch.int64, device="cuda")
names = {36: "k_tower16b, 2 boards per workgroup", 39: "k_tower16b, 4 boards per workgroup (lock-step)", 0: "k_tower (32x32x16)",
         50: "k_tower1w: one wave per SIMD, 128 x 96 tile, 4 boards/WG",
         51: "k_tower1w, no stage barriers (probe)", 52: "k_tower1w, no barriers, no vmcnt waits (probe)", 53: "k_tower1w, no barriers / waits / DMA (probe)",
         30: "k_tower16b<2>, no weight refills (probe)", 31: "k_tower16b<2>, no stage barriers (probe)",
         41: "k_tower16b<4>, NO stage barriers (probe)", 43: "k_tower16b<4>, no barriers, no refills (probe)"}
for variant in [int(v) for v in sys.argv[1:]] or (36, 39, 50, 36, 39, 50):
    L.xq_tower_set_variant(variant)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if fn(*args, stamps.data_ptr()) != 0:
        print("%-36s not in this library (timing probes need a build with XQ_TOWER_PROBES=1)" % names[variant])
        continue
    for _ in range(3):
        fn(*args, stamps.data_ptr())
    e0.record()
    for _ in range(20):
        fn(*args, stamps.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 64).astype(np.float64)
    s = s[s[:, 61] > 0]
    tot, rt = s[:, 61] - s[:, 0], s[:, 63] - s[:, 62]
    ms = e0.elapsed_time(e1) / 20
    print("%-52s %.3f ms = %.3f of 2.5 PFLOP/s, workgroup %d cycles, clock %.3f GHz" % (
        names.get(variant, "variant %d" % variant), ms, fl / ms / 1e9 / 2500.0, np.median(tot), np.median(tot / rt * 0.1)), flush=True)
    if os.environ.get("XQ_PROBE_PHASES"):
        nl = 2 * blocks
        main = [int(np.median(s[:, 3 + 2 * l] - s[:, 2 + 2 * l])) for l in range(nl)]
        epi = [int(np.median(s[:, 4 + 2 * l] - s[:, 3 + 2 * l])) for l in range(nl)]
        print("    input conv %d + epilogue %d | main loops %s | epilogues %s | heads %d + stores %d" % (
            np.median(s[:, 1] - s[:, 0]), np.median(s[:, 2] - s[:, 1]), main, epi, np.median(s[:, 60] - s[:, 2 + 2 * nl]),
            np.median(s[:, 61] - s[:, 60])))
    if os.environ.get("XQ_PROBE_PHASES") and variant in (50, 51, 52) and s[:, 30].max() > 0:
        for layer in (2, 3):
            f = s[:, 30 + 6 * (layer - 2):30 + 6 * (layer - 2) + 6]
            d = np.median(np.diff(f, axis=1), axis=0)
            print("    epilogue of layer %d: pair 0 %d, pair 1 %d, pairs 2 + 3 %d, wait for the bias loads %d, skip MFMAs %d cycles" % (
                layer, d[0], d[1], d[2], d[3], d[4]))
    stamps.zero_()
L.xq_tower_set_variant(-1)
