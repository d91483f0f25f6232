"""Phase stamps of k_policy_fc1w (probes library: XQ_TOWER_PROBES=1; run on the GPU box): per workgroup shader cycles of
prologue (operands of the first K-stage landed), K loop and epilogue, the shader clock the chip held (cycles / 100 MHz ticks),
and how the workgroups spread over the launch.  usage: stamps_policy_fc.py [M=16384]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chinesechessai_amd import _lib

L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N, K = 2304, 2880
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
act = (torch.randn(M, K, device="cuda") * (torch.rand(M, K, device="cuda") < 0.5)).bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ntiles = ((M + 255) // 256) * (N // 192)
stamps = torch.zeros(ntiles * 32, dtype=torch.int64, device="cuda")
for nodma in (0, 1):
    for rep in range(6):          # the last of a back-to-back series is the one reported (clock settled)
        rc = L.xq_policy_fc_debug_stamps(nodma, st, act.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, stamps.data_ptr())
        if rc != 0:
            sys.exit("not in this library (XQ_TOWER_PROBES=1)")
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(ntiles, 4, 8)[:, 0, :]          # wave 0 of every workgroup
    pro, loop, epi, tot = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 3] - s[:, 0]
    ticks = s[:, 5] - s[:, 4]
    ghz = tot.sum() / ticks.sum() * 0.1
    t0 = s[:, 4].min()
    nm = 90 * 48
    print("%s: %d workgroups; cycles per workgroup: prologue %.0f  K loop %.0f (%.2f per MFMA)  epilogue %.0f  total %.0f; clock %.2f GHz; "
          "a workgroup takes %.1f us; first start -> last end %.1f us; starts at (us, deciles) %s" % (
              "no operand DMA behind the prologue" if nodma else "k_policy_fc1w", ntiles, pro.mean(), loop.mean(), loop.mean() / nm, epi.mean(),
              tot.mean(), ghz, ticks.mean() / 100.0, (s[:, 5].max() - t0) / 100.0,
              np.round(np.percentile((s[:, 4] - t0) / 100.0, [0, 10, 30, 50, 70, 90, 100]), 1).tolist()), flush=True)
