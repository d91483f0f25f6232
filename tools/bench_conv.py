"""Micro-benchmark + check of the fused conv kernel against torch (run on the GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from chinesechessai_amd import _lib

L = _lib.lib()
torch.manual_seed(0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for cin in (128, 16):
    x = (torch.randn(G, 10, 9, cin, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(128, cin, 3, 3, device="cuda") * (1.0 / (3 * cin ** 0.5))).bfloat16()
    b = torch.randn(128, device="cuda") * 0.1
    r = (torch.randn(G, 10, 9, 128, device="cuda") * 0.5).bfloat16()
    wk = w.permute(2, 3, 0, 1).reshape(9, 128, cin).contiguous()
    y = torch.empty(G, 10, 9, 128, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    for use_res in (False, True):
        _lib.check(L.xq_conv3x3_nhwc_bf16(st, x.data_ptr(), wk.data_ptr(), b.data_ptr(), r.data_ptr() if use_res else None,
                                          y.data_ptr(), G, cin, 1))
        torch.cuda.synchronize()
        n = min(G, 512)
        ref = F.conv2d(x[:n].permute(0, 3, 1, 2).float(), w.float(), b, padding=1)
        ref = ref.bfloat16().float()           # the kernel rounds conv+bias to bf16 before the residual add
        if use_res:
            ref = ref + r[:n].permute(0, 3, 1, 2).float()
        ref = torch.relu(ref).permute(0, 2, 3, 1)
        err = (y[:n].float() - ref).abs().max().item()
        print("cin=%d res=%d max_abs_err=%.4g (ref max %.3g)" % (cin, use_res, err, ref.abs().max().item()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        L.xq_conv3x3_nhwc_bf16(st, x.data_ptr(), wk.data_ptr(), b.data_ptr(), r.data_ptr(), y.data_ptr(), G, cin, 1)
    e0.record()
    it = 20
    for _ in range(it):
        L.xq_conv3x3_nhwc_bf16(st, x.data_ptr(), wk.data_ptr(), b.data_ptr(), r.data_ptr(), y.data_ptr(), G, cin, 1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    fl = 2.0 * G * 90 * 128 * 9 * cin
    print("cin=%d G=%d: %.3f ms  %.1f TFLOP/s" % (cin, G, ms, fl / ms / 1e9))
    # torch/MIOpen reference timing (conv only, no epilogue)
    xt = x.permute(0, 3, 1, 2)
    wt = w.contiguous(memory_format=torch.channels_last)
    bb = b.bfloat16()
    for _ in range(3):
        F.conv2d(xt, wt, bb, padding=1)
    e0.record()
    for _ in range(it):
        F.conv2d(xt, wt, bb, padding=1)
    e1.record()
    torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / it
    print("   torch conv2d+bias: %.3f ms  %.1f TFLOP/s" % (ms2, fl / ms2 / 1e9))

# ---- phase stamps (diagnostic build) ----
import ctypes as C
cin = 128
x = (torch.randn(G, 10, 9, cin, device="cuda") * 0.5).bfloat16()
w = (torch.randn(128, cin, 3, 3, device="cuda") * (1.0 / (3 * cin ** 0.5))).bfloat16()
wk = w.permute(2, 3, 0, 1).reshape(9, 128, cin).contiguous()
b = torch.randn(128, device="cuda") * 0.1
r = (torch.randn(G, 10, 9, 128, device="cuda") * 0.5).bfloat16()
y = torch.empty(G, 10, 9, 128, device="cuda", dtype=torch.bfloat16)
nwg = (G + 3) // 4
stamps = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
fn = L.xq_conv3x3_debug_stamps
fn.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]
for _ in range(2):
    fn(st, x.data_ptr(), wk.data_ptr(), b.data_ptr(), r.data_ptr(), y.data_ptr(), G, 1, stamps.data_ptr())
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(nwg, 16)
import numpy as np
d = np.diff(s[:, :13], axis=1).astype(np.float64)
names = ["prologue(load+barrier)"] + ["tap%d" % t for t in range(9)] + ["epi: acc->LDS + barrier", "epi: rows->global"]
print("phase medians in s_memtime ticks (100 MHz? see total):")
for i, n in enumerate(names):
    print("  %-26s median %8.0f  p90 %8.0f" % (n, np.median(d[:, i]), np.percentile(d[:, i], 90)))
tot = (s[:, 12] - s[:, 0]).astype(np.float64)
print("  total per WG median %.0f ticks; kernel span %.0f ticks; sum/CU estimate" % (np.median(tot), s[:, 12].max() - s[:, 0].min()))
