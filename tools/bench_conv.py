"""Micro-benchmark + check of the fused conv kernel against torch (run on the GPU box)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from chinesechessai_amd import _lib

L = _lib.lib()
torch.manual_seed(0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
st = torch.cuda.current_stream().cuda_stream


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


for cin in (128, 16):
    x = torch.relu(torch.randn(G, 10, 9, cin, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(128, cin, 3, 3, device="cuda") * (1.0 / (3 * cin ** 0.5))).bfloat16()
    b = torch.randn(128, device="cuda") * 0.1
    r = torch.relu(torch.randn(G, 10, 9, 128, device="cuda") * 0.5).bfloat16()
    wk = w.permute(2, 3, 0, 1).reshape(9, 128, cin).contiguous()
    y = torch.empty(G, 10, 9, 128, device="cuda", dtype=torch.bfloat16)
    n = min(G, 512)
    fl = 2.0 * G * 90 * 128 * 9 * cin
    for variant in (1, 2):
        L.xq_conv3x3_set_variant(variant)
        for use_res in (False, True):
            y.zero_()
            _lib.check(L.xq_conv3x3_nhwc_bf16(st, x.data_ptr(), wk.data_ptr(), b.data_ptr(),
                                              r.data_ptr() if use_res else None, y.data_ptr(), G, cin, 1))
            torch.cuda.synchronize()
            ref = F.conv2d(x[:n].permute(0, 3, 1, 2).float(), w.float(), b, padding=1).bfloat16().float()
            if use_res:
                ref = ref + r[:n].permute(0, 3, 1, 2).float()
            ref = torch.relu(ref).permute(0, 2, 3, 1)
            err = (y[:n].float() - ref).abs().max().item()
            tail = (y[G - 3:].float() - torch.relu(
                F.conv2d(x[G - 3:].permute(0, 3, 1, 2).float(), w.float(), b, padding=1).bfloat16().float()
                + (r[G - 3:].permute(0, 3, 1, 2).float() if use_res else 0)).permute(0, 2, 3, 1)).abs().max().item()
            print("variant=%d cin=%d res=%d max_abs_err=%.4g tail_err=%.4g" % (variant, cin, use_res, err, tail))
        ms = timeit(lambda: L.xq_conv3x3_nhwc_bf16(st, x.data_ptr(), wk.data_ptr(), b.data_ptr(), r.data_ptr(),
                                                    y.data_ptr(), G, cin, 1))
        print("variant=%d cin=%d G=%d: %.3f ms  %.1f TFLOP/s" % (variant, cin, G, ms, fl / ms / 1e9))
    xt = x.permute(0, 3, 1, 2)
    wt = w.contiguous(memory_format=torch.channels_last)
    bb = b.bfloat16()
    ms2 = timeit(lambda: F.conv2d(xt, wt, bb, padding=1))
    print("   torch conv2d+bias: %.3f ms  %.1f TFLOP/s" % (ms2, fl / ms2 / 1e9))

# ---- phase stamps (diagnostic builds) ----
cin = 128
x = torch.relu(torch.randn(G, 10, 9, cin, device="cuda") * 0.5).bfloat16()
w = (torch.randn(128, cin, 3, 3, device="cuda") * (1.0 / (3 * cin ** 0.5))).bfloat16()
wk = w.permute(2, 3, 0, 1).reshape(9, 128, cin).contiguous()
b = torch.randn(128, device="cuda") * 0.1
r = torch.relu(torch.randn(G, 10, 9, 128, device="cuda") * 0.5).bfloat16()
y = torch.empty(G, 10, 9, 128, device="cuda", dtype=torch.bfloat16)
fn = L.xq_conv3x3_debug_stamps
fn.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]
for variant, nb, nst in ((1, 2, 18), (2, 4, 9)):
    for ablate in (0, 1, 2, 3):
        nwg = (G + nb - 1) // nb
        stamps = torch.zeros(nwg * 32, dtype=torch.int64, device="cuda")
        for _ in range(2):
            fn(variant, ablate, st, x.data_ptr(), wk.data_ptr(), b.data_ptr(), r.data_ptr(), y.data_ptr(), G, 1,
               stamps.data_ptr())
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(nwg, 32).astype(np.float64)
        stages = s[:, 1 + nst] - s[:, 1]
        rt = s[:, 25] - s[:, 24]
        tot = s[:, 23] - s[:, 0]
        print("variant %d ablate %d: prologue %d | %d stages %d | acc->LDS %d | barrier %d | rows->global %d | WG total %d "
              "cycles, clock %.3f GHz, kernel wall %.1f us" % (
                  variant, ablate, np.median(s[:, 1] - s[:, 0]), nst, np.median(stages), np.median(s[:, 21] - s[:, 20]),
                  np.median(s[:, 22] - s[:, 21]), np.median(s[:, 23] - s[:, 22]), np.median(tot),
                  np.median(tot / rt * 0.1), (s[:, 25].max() - s[:, 24].min()) / 100.0))
