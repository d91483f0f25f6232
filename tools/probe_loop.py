"""Steady-state loop probes (run on the GPU box; library built with XQ_TOWER_PROBES=1): the trunk's main loop without
barriers / epilogues as a function of the wave tile - mode 20: 64 x 96 per wave, two waves per SIMD (the product
kernel's tile, 2 boards per workgroup); mode 21: 128 x 96 per wave, ONE wave per SIMD (4 boards per workgroup, 14
fragment reads per 48 MFMAs, half the weight stream per board)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chinesechessai_amd import _lib

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
probe = L.xq_mfma_probe
probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
seed = torch.randint(0, 2 ** 31 - 1, (64,), dtype=torch.int32, device="cuda")
wts = torch.randint(-2 ** 31, 2 ** 31 - 1, (216 * 16384 // 4,), dtype=torch.int32, device="cuda")
outp = torch.zeros(4, device="cuda")
taps = 108                       # 12 layers x 9 taps: one workgroup's main-loop work
for mode, name, nwg, mt in ((20, "64 x 96 tile, 2 waves/SIMD, 2 boards/WG", 8192, 4), (21, "128 x 96 tile, 1 wave/SIMD, 4 boards/WG", 4096, 8),
                            (22, "128 x 96, 1 wave/SIMD, accumulators on fixed AGPRs", 4096, 8),
                            (23, "  ... without the weight DMA", 4096, 8), (24, "  ... one read / DMA piece per MFMA gap", 4096, 8),
                            (25, "  ... per-gap issue, no DMA", 4096, 8)) * 2:
    if probe(st, seed.data_ptr(), wts.data_ptr(), outp.data_ptr(), nwg, taps, mode) != 0:
        print("mode %d: not in this library (XQ_TOWER_PROBES=1 build needed)" % mode)
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        probe(st, seed.data_ptr(), wts.data_ptr(), outp.data_ptr(), nwg, taps, mode)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    o = outp.cpu().numpy()
    fl = nwg * 4.0 * taps * 4 * mt * 6 * 2.0 * 16 * 16 * 32
    print("%-52s %.3f ms  %.1f TFLOP/s = %.3f of 2.5 PFLOP/s, workgroup %d cycles, clock %.3f GHz" % (
        name, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0, o[1], o[1] / o[2] * 0.1), flush=True)
