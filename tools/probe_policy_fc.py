"""k_policy_fc against k_policy_fc1w and against its two timing-only bodies (probes library: XQ_TOWER_PROBES=1), interleaved in one process (run on the
GPU box): what the MFMA stream alone takes, what the operand delivery alone takes, and the kernel.  usage: probe_policy_fc.py [M=16384]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chinesechessai_amd import _lib

L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N, K = 2304, 2880
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
act = (torch.randn(M, K, device="cuda") * (torch.rand(M, K, device="cuda") < 0.5)).bfloat16()      # post-ReLU-like: half zeros
w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
fl = 2.0 * M * N * K


def timeit(fn, it=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def variant(v):
    def fn():
        L.xq_policy_fc_set_variant(v)
        r = L.xq_policy_fc_bf16(st, act.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, None)
        L.xq_policy_fc_set_variant(-1)
        return r
    return fn


cases = (("k_policy_fc (8 waves, HIP)", variant(0)), ("k_policy_fc1w (one wave per SIMD, asm body)", variant(1)),
         ("k_policy_fc1w, one barrier, pieces 12 + 2", variant(2)), ("k_policy_fc1w, two barriers, pieces 5 + 9", variant(5)), ("k_policy_fc1w, no operand DMA behind the prologue", variant(3)),
         ("k_policy_fc1w, no MFMAs", variant(4)),
         ("k_policy_fc1w, one barrier, pieces 14 + 0", variant(6)),
         ("kernel", lambda: L.xq_policy_fc_bf16(st, act.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, None)),
         ("no operand DMA behind the first two stages", lambda: L.xq_policy_fc_debug(1, st, act.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K)),
         ("no MFMAs", lambda: L.xq_policy_fc_debug(2, st, act.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K)))
for rep in range(2):
    for name, fn in cases:
        if fn() != 0:
            print("%s: not in this library (XQ_TOWER_PROBES=1)" % name)
            continue
        us = timeit(fn)
        print("%-52s %7.1f us  %6.1f TFLOP/s equivalent = %.3f of 2.5 PFLOP/s; operand stream %.2f TB/s" % (
            name, us, fl / us / 1e6, fl / us / 1e6 / 2500.0, (M / 256) * (N / 192) * (K / 64) * 57344 / us / 1e6), flush=True)
