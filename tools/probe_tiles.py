"""Power probe (run on the GPU box): k_tower16s against a build whose last two waves leave out their sixth pixel tile
(23 instead of 24 pixel tiles of MFMA work per 4 boards; wrong results, timing only).  The critical path does not
change (the other six waves still run six tiles), so any gain is the chip answering less matrix work per board with
a higher clock."""
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
fn = L.xq_tower_debug_stamps
fn.argtypes = [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p]
stamps = torch.zeros((G + 1) // 2 * 64, dtype=torch.int64, device="cuda")
names = {33: "k_tower16b<PAIR>, skip connection on the VALU",
         29: "k_tower16b<PAIR>, 4 boards in lock-step", 9: "k_tower16b, 4 boards in lock-step",
         30: "k_tower16b<PAIR>, no weight refills", 31: "k_tower16b<PAIR>, no stage barriers", 32: "k_tower16b<PAIR>, one filler per MFMA gap",
         2: "k_tower16b", 8: "k_tower16b, 16-byte epilogue stores", 10: "k_tower16s", 24: "k_tower16s, 16-byte epilogue stores", 25: "k_tower16s, 16-byte stores, s_setprio 3", 11: "k_tower16s, 23 of 24 pixel tiles", 20: "k_tower16s, s_setprio 3 in epilogues",}
for variant in [int(v) for v in sys.argv[1:]] or (2, 10, 11, 2, 10, 11, 10, 11):
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
    print("%-36s %.3f ms, workgroup %d cycles, clock %.3f GHz" % (names[variant], e0.elapsed_time(e1) / 20, np.median(tot),
                                                                 np.median(tot / rt * 0.1)), flush=True)
    if variant in (10, 20, 24, 25) and s[:, 30].max() > 0:
        for grp, o in (("lead", 0), ("lag", 12)):
            for layer in (2, 3):
                f = s[:, 30 + o + 6 * (layer - 2):30 + o + 6 * (layer - 2) + 6]
                d = np.median(np.diff(f[:, :5], axis=1), axis=0)
                print("    %s group, layer %d: first half %d (issued after %d), wait at #18 %d, second half %d, wait at #19 %d cycles"
                      % (grp, layer, d[0], np.median(f[:, 5] - f[:, 0]), d[1], d[2], d[3]))
    if s[:, 54].max() > 0:      # slots 54 / 55 / 56: lead group's arrival at, release from stage barrier 8 of layer 2, arrival at 9; 57..59: lag group
        print("    layer 2, stage barrier 8: lead group waits %d, lag group waits %d cycles; arrival to next arrival: lead %d, lag %d; "
              "lead arrives %d cycles before lag" % (np.median(s[:, 55] - s[:, 54]), np.median(s[:, 58] - s[:, 57]),
                                                      np.median(s[:, 56] - s[:, 54]), np.median(s[:, 59] - s[:, 57]),
                                                      np.median(s[:, 57] - s[:, 54])))
    stamps.zero_()
L.xq_tower_set_variant(36)
