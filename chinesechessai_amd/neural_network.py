"""Policy/value network of the reference (neural_network.py:12-187) under PyTorch-ROCm.

`ChessNet` keeps the reference's parameter names so that a reference `state_dict` loads unchanged
(keys conv1, bn1, res_blocks.{i}.{conv1,bn1,conv2,bn2}, policy_conv, policy_bn, policy_fc,
value_conv, value_bn, value_fc1, value_fc2); `num_blocks` generalises the hard-coded 4
(neural_network.py:29-31) for the 6- and 20-block BASELINE configs.

`InferenceNet` is the leaf evaluator the engine drives: eval-mode BatchNorm folded into the
convolutions, bf16 (or fp32) weights, channels-last activations, input planes written directly by
the search kernel.  The network is the only MFMA user on the path; everything else is integer work.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import BOARD_SIZE, BOARD_WIDTH


class ResidualBlock(nn.Module):
    """neural_network.py:172-187"""

    def __init__(self, num_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(num_channels, num_channels, kernel_size=3, padding=1)
        self.bn1 = nn.BatchNorm2d(num_channels)
        self.conv2 = nn.Conv2d(num_channels, num_channels, kernel_size=3, padding=1)
        self.bn2 = nn.BatchNorm2d(num_channels)

    def forward(self, x):
        residual = x
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out = out + residual
        return F.relu(out)


class ChessNet(nn.Module):
    """neural_network.py:12-169 (same module construction order, hence the same default init
    under the same torch seed)."""

    def __init__(self, num_channels=128, num_blocks=4):
        super().__init__()
        self.num_channels = num_channels
        self.num_blocks = num_blocks
        self.conv1 = nn.Conv2d(15, num_channels, kernel_size=3, padding=1)
        self.bn1 = nn.BatchNorm2d(num_channels)
        self.res_blocks = nn.ModuleList([ResidualBlock(num_channels) for _ in range(num_blocks)])
        self.policy_conv = nn.Conv2d(num_channels, 32, kernel_size=1)
        self.policy_bn = nn.BatchNorm2d(32)
        self.policy_fc = nn.Linear(32 * BOARD_SIZE * BOARD_WIDTH, BOARD_SIZE * BOARD_WIDTH * 90)
        self.value_conv = nn.Conv2d(num_channels, 8, kernel_size=1)
        self.value_bn = nn.BatchNorm2d(8)
        self.value_fc1 = nn.Linear(8 * BOARD_SIZE * BOARD_WIDTH, 128)
        self.value_fc2 = nn.Linear(128, 1)

    def forward(self, x):
        """neural_network.py:47-71"""
        x = F.relu(self.bn1(self.conv1(x)))
        for blk in self.res_blocks:
            x = blk(x)
        policy = F.relu(self.policy_bn(self.policy_conv(x)))
        policy = policy.reshape(policy.size(0), -1)
        policy = self.policy_fc(policy)
        value = F.relu(self.value_bn(self.value_conv(x)))
        value = value.reshape(value.size(0), -1)
        value = F.relu(self.value_fc1(value))
        value = torch.tanh(self.value_fc2(value))
        return policy, value

    # ---- reference-compatible host-side helpers (duck-typed evaluator surface) ----------
    @staticmethod
    def encode_board(board, current_player):
        """neural_network.py:128-146"""
        encoded = np.zeros((15, BOARD_SIZE, BOARD_WIDTH), dtype=np.float32)
        for i in range(1, 8):
            encoded[i - 1] = (board == i).astype(np.float32)
            encoded[i + 6] = (board == -i).astype(np.float32)
        encoded[14] = np.ones((BOARD_SIZE, BOARD_WIDTH)) * (current_player == 1)
        return encoded

    @staticmethod
    def _logits_to_move_probs(logits, legal_moves):
        """neural_network.py:148-169"""
        if len(legal_moves) == 0:
            return {}
        idx = [(m[0] * BOARD_WIDTH + m[1]) * 90 + (m[2] * BOARD_WIDTH + m[3]) for m in legal_moves]
        probs = np.array([logits[i] for i in idx])
        probs = np.exp(probs - np.max(probs))
        probs = probs / np.sum(probs)
        return {move: prob for move, prob in zip(legal_moves, probs)}

    def predict_batch(self, rows):
        """neural_network.py:96-126"""
        if len(rows) == 0:
            return []
        dev = next(self.parameters()).device
        states = np.array([self.encode_board(b, p) for b, p, _ in rows])
        x = torch.from_numpy(states).to(dev)
        with torch.no_grad():
            logits, values = self.forward(x)
        logits = logits.float().cpu().numpy()
        values = values.float().cpu()
        return [(self._logits_to_move_probs(logits[i], legal), values[i].item())
                for i, (_, _, legal) in enumerate(rows)]

    def predict(self, board, current_player, legal_moves):
        """neural_network.py:73-94"""
        return self.predict_batch([(board, current_player, legal_moves)])[0]


def reachable_policy_columns():
    """Policy-head columns (neural_network.py:160: idx = from*90 + to) that can ever be a legal
    move: the union, over both sides and every source square, of the reference generators' targets
    on an EMPTY board (chess_env.py:123-251; blockers only remove targets, cannon captures stay on
    rook lines).  Returns (sorted column indices, int16 map move -> compact column or -1).
    The other ~5,700 of the 8,100 logits are never read by the search (the engine gathers legal
    moves only), so a policy FC restricted to these rows of its weight is result-identical."""
    cols = set()
    for r in range(10):
        for c in range(9):
            tg = set()
            for k in range(9):                                   # rook / cannon lines
                if k != c:
                    tg.add((r, k))
            for k in range(10):
                if k != r:
                    tg.add((k, c))
            for dr, dc in ((2, 1), (2, -1), (-2, 1), (-2, -1), (1, 2), (-1, 2), (1, -2), (-1, -2)):   # knight
                tg.add((r + dr, c + dc))
            for side in (1, -1):
                lo, hi = (7, 10) if side == 1 else (0, 3)
                for dr, dc in ((0, 1), (0, -1), (1, 0), (-1, 0), (1, 1), (1, -1), (-1, 1), (-1, -1)):   # king, advisor
                    if lo <= r + dr < hi and 3 <= c + dc < 6:
                        tg.add((r + dr, c + dc))
                for dr, dc in ((2, 2), (2, -2), (-2, 2), (-2, -2)):                                       # bishop
                    nr = r + dr
                    if (side == 1 and nr >= 5) or (side == -1 and nr < 4):
                        tg.add((nr, c + dc))
                tg.add((r - 1, c) if side == 1 else (r + 1, c))                                           # pawn
                if (side == 1 and r < 5) or (side == -1 and r >= 5):
                    tg.add((r, c - 1))
                    tg.add((r, c + 1))
            for tr, tc in tg:
                if 0 <= tr < 10 and 0 <= tc < 9:
                    cols.add((r * 9 + c) * 90 + tr * 9 + tc)
    cols = sorted(cols)
    cmap = np.full(8100, -1, dtype=np.int16)
    cmap[cols] = np.arange(len(cols), dtype=np.int16)
    return np.array(cols, dtype=np.int64), cmap


def _fold_bn(conv, bn):
    """eval-mode BatchNorm folded into the preceding convolution (running stats, foldable per
    SURVEY.md §8a a12)."""
    w = conv.weight.detach().double()
    b = conv.bias.detach().double() if conv.bias is not None else torch.zeros(w.shape[0], dtype=torch.float64)
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    w = w * scale.view(-1, 1, 1, 1)
    b = (b - bn.running_mean.detach().double()) * scale + bn.bias.detach().double()
    return w, b


class InferenceNet(nn.Module):
    """Folded, channels-last, low-precision forward of a ChessNet for the engine.

    Input : planes tensor the search kernel wrote — logical shape [G, C_in, 10, 9] with
            C_in = 15 (NCHW) or 16 (channels-last storage, channel 15 zero).
    Output: (logits [G, n_policy], values [G]) in `dtype`.  With the default policy_columns="reachable" the
            logits are NOT in the reference's from*90+to order: n_policy = 2,304 compact columns (2,294 real ones,
            padded), and `column_map[from*90+to]` gives a move's column (-1: never legal).  policy_columns="all"
            keeps the reference's 8,100 columns in order (padded to 8,256 on the hand-written path).  Callers that
            index logits by move must go through column_map (the engine does: xq_engine_set_logit_columns).

    The hand-written kernels cover dtype=bfloat16, c_in=16 (channels-last), 128 channels on a CUDA device; any
    other configuration raises ValueError unless allow_library_fallback=True asks for PyTorch's library kernels.
    row_src / n_rows (device pointers from SelfPlayEngine.row_map(), hand-written fused path only): evaluator row
    compaction - row r reads the planes of x[row_src[r]], writes outputs to row r, rows >= *n_rows are skipped.
    """

    def __init__(self, net, dtype=torch.bfloat16, c_in=16, device="cuda", policy_columns="reachable", fused_tower=True,
                 allow_library_fallback=False):
        super().__init__()
        self.dtype = dtype
        self.fused_tower = fused_tower                 # one launch for the whole trunk (csrc/xq_tower.hip)
        self.c_in = c_in
        # "reachable" (default): the policy FC computes only the 2,294 columns a legal move can index (the search
        # gathers legal moves only, neural_network.py:148-169: every other logit is a dead output);
        # "all": the reference's 8,100 columns (padded to the kernel's multiple of 192)
        self.policy_columns = policy_columns
        convs = []
        w, b = _fold_bn(net.conv1, net.bn1)
        if c_in == 16:
            w = F.pad(w, (0, 0, 0, 0, 0, 1))
        convs.append((w, b))
        for blk in net.res_blocks:
            convs.append(_fold_bn(blk.conv1, blk.bn1))
            convs.append(_fold_bn(blk.conv2, blk.bn2))
        self.n_blocks = len(net.res_blocks)
        cl = torch.channels_last
        self.cw = nn.ParameterList([nn.Parameter(w.to(device=device, dtype=dtype).contiguous(memory_format=cl),
                                                 requires_grad=False) for w, _ in convs])
        self.cb = nn.ParameterList([nn.Parameter(b.to(device=device, dtype=dtype), requires_grad=False)
                                    for _, b in convs])
        # hand-written fused conv path: weights as [tap][cout][cin] bf16, bias fp32
        self.use_hip_conv = (dtype == torch.bfloat16 and c_in == 16 and str(device).startswith("cuda")
                             and net.conv1.out_channels == 128)
        if not self.use_hip_conv and not allow_library_fallback:
            # the hand-written kernels cover bf16, channels-last 16-plane input, 128 channels; anything else
            # would run on library kernels (MIOpen / hipBLASLt): only on request, never silently
            raise ValueError("InferenceNet: no hand-written HIP path for dtype=%s, c_in=%d, channels=%d, device=%s; "
                             "pass allow_library_fallback=True to run this configuration on PyTorch's library kernels "
                             "(parity / debugging runs only)" % (dtype, c_in, net.conv1.out_channels, device))
        self._buf = None
        self._hbuf = None
        self.tower_events = None
        # evaluator row compaction needs the row map inside the kernels: the single-launch hand-written path only
        self.supports_row_map = bool(self.use_hip_conv and fused_tower)
        if self.use_hip_conv:
            self.hip_w = [w.permute(2, 3, 0, 1).reshape(9, w.shape[0], w.shape[1]).to(device=device, dtype=dtype).contiguous()
                          for w, _ in convs]
            self.hip_b = [b.to(device=device, dtype=torch.float32).contiguous() for _, b in convs]
            # the same tensors as one weight stream / one bias table for the single-launch trunk
            self.hip_wt = (torch.stack(self.hip_w[1:]).contiguous() if self.n_blocks else
                           torch.zeros((0, 9, 128, 128), dtype=dtype, device=device))
            self.hip_bt = torch.stack(self.hip_b).contiguous()
        # heads: the two 1x1 convolutions share one GEMM (32 + 8 output channels)
        pw, pb = _fold_bn(net.policy_conv, net.policy_bn)
        vw, vb = _fold_bn(net.value_conv, net.value_bn)
        hw = torch.cat([pw, vw], 0)
        hb = torch.cat([pb, vb], 0)
        if self.use_hip_conv:
            hw64 = torch.zeros((64, hw.shape[1]), dtype=torch.float64)
            hw64[:40] = hw.reshape(40, -1)
            hb64 = torch.zeros(64, dtype=torch.float64)
            hb64[:40] = hb
            self.hip_hw = hw64.to(device=device, dtype=dtype).contiguous()
            self.hip_hb = hb64.to(device=device, dtype=torch.float32).contiguous()
        self.hw = nn.Parameter(hw.to(device=device, dtype=dtype).contiguous(memory_format=cl), requires_grad=False)
        self.hb = nn.Parameter(hb.to(device=device, dtype=dtype), requires_grad=False)
        # policy FC consumes the NHWC-flattened activation: permute its input columns once
        # from (c, h, w) order (neural_network.py:62) to (h, w, c)
        fcw = net.policy_fc.weight.detach().cpu().view(-1, 32, 90).permute(0, 2, 1).reshape(-1, 2880)
        pfb = net.policy_fc.bias.detach().cpu()
        if policy_columns == "reachable":
            cols, cmap = reachable_policy_columns()
        elif policy_columns == "all":
            cols, cmap = np.arange(8100, dtype=np.int64), None
        else:
            raise ValueError("policy_columns must be 'reachable' or 'all'")
        # rows padded to the hand-written GEMM's column tile (192; zero weights, zero bias); the library path of
        # the reference layout keeps exactly 8,100 columns
        pad = (-len(cols)) % 192 if (self.use_hip_conv or cmap is not None) else 0
        if pad and cmap is None:
            cmap = np.arange(8100, dtype=np.int16)           # identity map: only the row stride differs
        idx = torch.from_numpy(cols)
        fcw, pfb = fcw[idx], pfb[idx]
        if pad:
            fcw = torch.cat([fcw, torch.zeros((pad, 2880), dtype=fcw.dtype)], 0)
            pfb = torch.cat([pfb, torch.zeros(pad, dtype=pfb.dtype)], 0)
        self.column_map = cmap
        self.n_policy = fcw.shape[0]
        self.n_policy_real = len(cols)
        self.pfw = nn.Parameter(fcw.to(device=device, dtype=dtype).contiguous(), requires_grad=False)
        self.pfb = nn.Parameter(pfb.to(device=device, dtype=dtype), requires_grad=False)
        v1 = net.value_fc1.weight.detach().cpu().view(-1, 8, 90).permute(0, 2, 1).reshape(-1, 720)
        self.v1w = nn.Parameter(v1.to(device=device, dtype=dtype).contiguous(), requires_grad=False)
        self.v1b = nn.Parameter(net.value_fc1.bias.detach().to(device=device, dtype=dtype), requires_grad=False)
        self.v2w = nn.Parameter(net.value_fc2.weight.detach().to(device=device, dtype=dtype), requires_grad=False)
        self.v2b = nn.Parameter(net.value_fc2.bias.detach().to(device=device, dtype=dtype), requires_grad=False)
        if self.use_hip_conv:
            # hand-written FC kernels (csrc/xq_policy.hip): fp32 biases, value fc1 padded to K = 736
            self.hip_pfb = pfb.to(device=device, dtype=torch.float32).contiguous()
            v1p = torch.zeros((128, 736), dtype=v1.dtype)
            v1p[:, :720] = v1
            self.hip_v1w = v1p.to(device=device, dtype=dtype).contiguous()
            self.hip_v1b = net.value_fc1.bias.detach().to(device=device, dtype=torch.float32).contiguous()
            self.hip_v2w = net.value_fc2.weight.detach().reshape(128).to(device=device, dtype=torch.float32).contiguous()
            self.hip_v2b = net.value_fc2.bias.detach().reshape(1).to(device=device, dtype=torch.float32).contiguous()

    def folded_weights(self):
        """The tensors the hand-written kernels read (folded, re-laid, in kernel dtype): what must be equal on every rank
        of a sharded run (distributed.weights_equal_across_ranks)."""
        if not self.use_hip_conv:
            return [p.data for p in self.parameters()]
        return [self.hip_w[0], self.hip_wt, self.hip_bt, self.hip_hw, self.hip_hb, self.pfw.data, self.hip_pfb,
                self.hip_v1w, self.hip_v1b, self.hip_v2w, self.hip_v2b]

    def _tower_hip(self, x):
        """Residual tower on the hand-written fused conv kernel (csrc/xq_conv.hip): one launch per
        convolution, bias / residual / ReLU in its epilogue, NHWC bf16 throughout."""
        from . import _lib
        L = _lib.lib()
        g = x.shape[0]
        stream = torch.cuda.current_stream().cuda_stream
        xin = x.permute(0, 2, 3, 1)                       # NHWC view of the channels-last storage
        assert xin.is_contiguous() and xin.shape[-1] == 16
        if self._buf is None or self._buf[0].shape[0] != g:
            self._buf = [torch.empty((g, 10, 9, 128), dtype=torch.bfloat16, device=x.device) for _ in range(3)]
        a, b, c = self._buf

        def conv(src, dst, i, res, cin):
            _lib.check(L.xq_conv3x3_nhwc_bf16(stream, src.data_ptr(), self.hip_w[i].data_ptr(), self.hip_b[i].data_ptr(),
                                              res.data_ptr() if res is not None else None, dst.data_ptr(), g, cin, 1))
        conv(xin, a, 0, None, 16)
        cur, t1, t2 = a, b, c
        ev = None
        if self.tower_events is not None:                 # bench.py: HIP events around the 128-ch convs
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for i in range(self.n_blocks):
            conv(cur, t1, 1 + 2 * i, None, 128)
            conv(t1, t2, 2 + 2 * i, cur, 128)             # relu(conv + bias + residual)
            cur, t2 = t2, cur
        if ev is not None:
            ev[1].record()
            self.tower_events.append(ev)
        return cur.permute(0, 3, 1, 2)                    # logical NCHW, channels-last strides

    def _head_buffers(self, g, device):
        if self._hbuf is None or self._hbuf[0].shape[0] != g:
            flat = torch.zeros(g * 720 + 64, dtype=torch.bfloat16, device=device)   # slack: xq_value_head_bf16 reads 32 B past a row
            self._hbuf = (torch.empty((g, 2880), dtype=torch.bfloat16, device=device), flat[:g * 720].view(g, 720), flat)
        return self._hbuf[0], self._hbuf[1]

    def trunk_hip(self, x, row_src=None, n_rows=None):
        """planes -> (policy-head activations [G, 2880], value-head activations [G, 720]) in one
        launch of k_tower: activations stay in LDS across all layers."""
        from . import _lib
        g = x.shape[0]
        xin = x.permute(0, 2, 3, 1)
        assert xin.is_contiguous() and xin.shape[-1] == 16
        hp, hv = self._head_buffers(g, x.device)
        ev = None
        if self.tower_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        _lib.check(_lib.lib().xq_tower_nhwc_bf16(torch.cuda.current_stream().cuda_stream, xin.data_ptr(),
                                                 self.hip_w[0].data_ptr(), self.hip_wt.data_ptr(), self.hip_bt.data_ptr(),
                                                 self.hip_hw.data_ptr(), self.hip_hb.data_ptr(), hp.data_ptr(),
                                                 hv.data_ptr(), g, self.n_blocks, row_src, n_rows))
        if ev is not None:
            ev[1].record()
            self.tower_events.append(ev)
        return hp, hv

    @torch.no_grad()
    def forward(self, x, out_logits=None, out_values=None, row_src=None, n_rows=None):
        if self.use_hip_conv and x.is_cuda and self.fused_tower:
            hp, hv = self.trunk_hip(x, row_src, n_rows)
            return self._fc(hp, hv, x.shape[0], out_logits, out_values, n_rows)
        if row_src is not None or n_rows is not None:
            raise ValueError("InferenceNet: a row map needs the single-launch hand-written path (supports_row_map)")
        if self.use_hip_conv and x.is_cuda:
            x = self._tower_hip(x)
        else:
            x = F.relu(F.conv2d(x, self.cw[0], self.cb[0], padding=1))
            for i in range(self.n_blocks):
                y = F.relu(F.conv2d(x, self.cw[1 + 2 * i], self.cb[1 + 2 * i], padding=1))
                y = F.conv2d(y, self.cw[2 + 2 * i], self.cb[2 + 2 * i], padding=1)
                x = F.relu(y + x)
        g = x.shape[0]
        if self.use_hip_conv and x.is_cuda:
            # fused heads kernel: both 1x1 convs + ReLU, outputs already in the FC input layouts
            from . import _lib
            hp, hv = self._head_buffers(g, x.device)
            xin = x.permute(0, 2, 3, 1)
            assert xin.is_contiguous()
            _lib.check(_lib.lib().xq_heads_nhwc_bf16(torch.cuda.current_stream().cuda_stream, xin.data_ptr(),
                                                     self.hip_hw.data_ptr(), self.hip_hb.data_ptr(), hp.data_ptr(),
                                                     hv.data_ptr(), g))
        else:
            h = F.relu(F.conv2d(x, self.hw, self.hb))              # [G, 40, 10, 9] channels-last
            h = h.permute(0, 2, 3, 1)                               # [G, 10, 9, 40] view
            hp = h[..., :32].reshape(g, 2880)
            hv = h[..., 32:].reshape(g, 720)
        return self._fc(hp, hv, g, out_logits, out_values)

    def _fc(self, hp, hv, g, out_logits, out_values, n_rows=None):
        if self.use_hip_conv and hp.is_cuda:
            # hand-written FC kernels (csrc/xq_policy.hip): fixed accumulation order, no library GEMM
            from . import _lib
            L = _lib.lib()
            st = torch.cuda.current_stream().cuda_stream
            if out_logits is None:
                out_logits = torch.empty((g, self.n_policy), dtype=torch.bfloat16, device=hp.device)
            if out_values is None:
                out_values = torch.empty((g,), dtype=torch.bfloat16, device=hp.device)
            assert out_logits.is_contiguous() and out_logits.shape == (g, self.n_policy) and out_logits.dtype == torch.bfloat16
            assert hp.is_contiguous() and hv.is_contiguous() and out_values.is_contiguous()
            _lib.check(L.xq_policy_fc_bf16(st, hp.data_ptr(), self.pfw.data_ptr(), self.hip_pfb.data_ptr(),
                                           out_logits.data_ptr(), g, self.n_policy, 2880, n_rows))
            _lib.check(L.xq_value_head_bf16(st, hv.data_ptr(), self.hip_v1w.data_ptr(), self.hip_v1b.data_ptr(),
                                            self.hip_v2w.data_ptr(), self.hip_v2b.data_ptr(), out_values.data_ptr(), g,
                                            n_rows))
            return out_logits, out_values
        if out_logits is not None:
            policy = torch.addmm(self.pfb, hp, self.pfw.t(), out=out_logits)   # no extra copy of the logits
        else:
            policy = F.linear(hp, self.pfw, self.pfb)
        v = F.relu(F.linear(hv, self.v1w, self.v1b))
        v = torch.tanh(F.linear(v, self.v2w, self.v2b)).reshape(g)
        if out_values is not None:
            out_values.copy_(v)
            v = out_values
        return policy, v
