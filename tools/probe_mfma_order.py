"""Bare-MFMA probe (run on the GPU box): does it matter which operand stays constant over consecutive MFMAs?"""
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
iters = 20000
for mode, name in ((16, "pixel tile outer (B constant over 4 MFMAs)"), (17, "weight tile outer (A constant over 6 MFMAs)")) * 3:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    probe(st, seed.data_ptr(), wts.data_ptr(), outp.data_ptr(), 512, iters, mode)
    e0.record()
    for _ in range(3):
        probe(st, seed.data_ptr(), wts.data_ptr(), outp.data_ptr(), 512, iters, mode)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    o = outp.cpu().numpy()
    fl = 512 * 4 * iters * 48 * 2.0 * 16 * 16 * 32
    print("16x16x32, %s: %.3f ms  %.1f TFLOP/s, clock %.3f GHz" % (name, ms, fl / ms / 1e9, o[1] / o[2] * 0.1), flush=True)
