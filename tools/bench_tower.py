"""Micro-benchmark + phase stamps of the single-launch trunk kernel (run on the GPU box)."""
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


fl = 2.0 * G * 90 * (16 * 9 * 128 + 2 * blocks * 128 * 9 * 128 + 128 * 40)
NAMES = {0: "k_tower (32x32x16)", 36: "k_tower16b, 2 boards per workgroup", 39: "k_tower16b, 4 boards per workgroup",
         50: "k_tower1w (one wave per SIMD; probes build)"}
for variant in (0, 36, 39):
    L.xq_tower_set_variant(variant)
    ms = timeit(lambda: L.xq_tower_nhwc_bf16(*args, None, None))
    print("%s G=%d blocks=%d: %.3f ms  %.1f TFLOP/s" % (NAMES[variant], G, blocks, ms, fl / ms / 1e9))
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 36
L.xq_tower_set_variant(variant)
print("stamps: variant %d" % variant)

fn = L.xq_tower_debug_stamps
fn.argtypes = [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p]
nwg = (G + 1) // 2
stamps = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
if variant in (36, 39) and os.environ.get("XQ_BT_SHORT") is None:
    # ablation builds (results are wrong on purpose; probes library): what the weight refills / stage barriers cost
    abl = ((36, "stamped build"), (30, "no weight refills"), (31, "no stage barriers"), (36, "stamped build again")) if variant == 36 else (
        (39, "stamped build"), (41, "no stage barriers"), (43, "no stage barriers, no weight refills"), (39, "stamped build again"))
    for v, name in abl:
        L.xq_tower_set_variant(v)
        if fn(*args, stamps.data_ptr()) != 0:
            print("%s, %s: not in this library (build with XQ_TOWER_PROBES=1)" % (NAMES[variant], name))
            continue
        ms = timeit(lambda: fn(*args, stamps.data_ptr()), it=10)
        print("%s, %s: %.3f ms" % (NAMES[variant], name, ms))
    L.xq_tower_set_variant(variant)
for _ in range(2):
    fn(*args, stamps.data_ptr())
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(nwg, 64).astype(np.float64)
s = s[s[:, 61] > 0]                                  # (the 4-boards-per-workgroup build fills only half of the rows)
nwg = len(s)
nl = 2 * blocks
tot = s[:, 61] - s[:, 0]
rt = s[:, 63] - s[:, 62]
main = [np.median(s[:, 3 + 2 * l] - s[:, 2 + 2 * l]) for l in range(nl)]
epi = [np.median(s[:, 4 + 2 * l] - s[:, 3 + 2 * l]) for l in range(nl)]
print("input conv %d + epilogue %d | main loops %s | epilogues %s | heads %d + stores %d | WG total %d cycles" % (
    np.median(s[:, 1] - s[:, 0]), np.median(s[:, 2] - s[:, 1]), [int(v) for v in main], [int(v) for v in epi],
    np.median(s[:, 60] - s[:, 2 + 2 * nl]), np.median(s[:, 61] - s[:, 60]), np.median(tot)))
print("clock %.3f GHz, kernel wall %.1f us, WG wall median %.1f us; ideal MFMA cycles per WG-pair per layer: %d" % (
    np.median(tot / rt * 0.1), (s[:, 63].max() - s[:, 62].min()) / 100.0, np.median(rt) / 100.0, 2 * 18 * 24 * 32))

# per start-order group: do the early layers stay slow in later rounds? (lockstep of co-resident workgroups)
order = np.argsort(s[:, 62])
grp = np.array_split(order, 16)
print("start-order group: start us | main loop L0, L2, L4, L6, L11 | WG total")
t0 = s[:, 62].min()
for gi, idx in enumerate(grp):
    m = lambda l: int(np.median(s[idx, 3 + 2 * l] - s[idx, 2 + 2 * l]))
    print("  %2d: %7.1f..%7.1f | %6d %6d %6d %6d %6d | %d" % (
        gi, (s[idx, 62].min() - t0) / 100.0, (s[idx, 62].max() - t0) / 100.0, m(0), m(2), m(4), m(6), m(nl - 1),
        int(np.median(tot[idx]))))

# matrix-pipe ceiling under the power cap: same instruction and tile, 2 waves / SIMD, with and without
# the conv's LDS fragment reads and weight stream
probe = L.xq_mfma_probe
probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
seed = torch.randint(0, 2 ** 31 - 1, (64,), dtype=torch.int32, device="cuda")
wts = torch.randint(-2 ** 31, 2 ** 31 - 1, (216 * 16384 // 4,), dtype=torch.int32, device="cuda")
outp = torch.zeros(4, device="cuda")
for mode, name in ((0, "32x32x16, registers only"), (16, "16x16x32, registers only"), (0, "32x32x16, registers only"),
                   (16, "16x16x32, registers only"), (1, "32x32x16 + LDS fragment reads"),
                   (2, "32x32x16 + LDS reads + LDS-DMA weight stream")):
    iters = 20000
    ms = timeit(lambda: probe(st, seed.data_ptr(), wts.data_ptr(), outp.data_ptr(), 512, iters, mode), it=3)
    fl_p = 512 * 4 * iters * 24 * 2.0 * 32 * 32 * 16
    o = outp.cpu().numpy()
    print("MFMA probe, %s: %.3f ms  %.1f TFLOP/s, clock %.3f GHz, %.1f cycles per MFMA per SIMD" % (
        name, ms, fl_p / ms / 1e9, o[1] / o[2] * 0.1, o[1] / (iters * 24 * 2.0)))
