#!/usr/bin/env python3
"""gen_policy_fc1w.py - generator + checker of the assembly body of k_policy_fc1w (csrc/xq_policy.hip): the policy FC
(neural_network.py:39,64: Linear 2880 -> 8100, here on the columns the caller keeps) as a one-wave-per-SIMD MFMA GEMM.

    python tools/gen_policy_fc1w.py               # writes chinesechessai_amd/csrc/xq_policy_fc1w_body.inc after checking it
    python tools/gen_policy_fc1w.py --check       # regenerates in memory, checks, compares with the committed file
    python tools/gen_policy_fc1w.py --ablations   # + xq_policy_fc1w_abl.inc (timing-only / stamped bodies, not committed)

Why: round 4 measured k_policy_fc (8 waves, 64 x 96 wave tiles, two 56-KB K-stages, hipcc's schedule) at 183 us per 16,384
rows with its MFMA stream ALONE - no operand DMA - at 162 us and its operand delivery alone at 143 us
(profiles/r04_probe_policy_fc_ablations.txt).  A first one-wave body (K-stages of 32 through a ring of five, not kept) ran its
MFMA stream at 16.25 cycles per MFMA but took 23 with the DMA beside it: a K-stage of 32 fetches 64-B half cache lines, and
the L2 -> LDS path moved those at 11.5 TB/s (delivery alone 172 us) where whole 128-B lines arrive at 18 TB/s (108 us).  Hence:

* 4 waves, wave tile 96 columns x 128 rows = 6 x 8 MFMA tiles on a[0:191] (the trunk kernel's tile), weights as the A operand;
* K-stages of 64 (56 KB: 256 activation rows + 192 weight rows x 128 B), every DMA piece = 8 rows x 128 B = whole cache lines,
  two ring slots (at LDS 0 and 65,536; the slot of an address toggles with one XOR); a stage is two K-steps of 48 MFMAs;
* two barriers per stage ("early" placement, the one emitted): K-step (d, 0) loads the fragments of (d, 1) one per MFMA and, 10
  MFMAs later, frees slot d % 2 with lgkmcnt(0) + s_barrier; the 14 pieces per wave of stage d + 2 follow at once, one every 4
  MFMAs, and must have landed at the barrier of K-step (d + 1, 1), 66-118 MFMAs later - a counted vmcnt that finds them there
  (16.95 cycles per MFMA with the DMA, 16.4 without; the one-barrier forms, whose window is 22 MFMAs shorter, took 19.3);
  two instructions per piece (M0 in front of the preceding MFMA); a stage past the end is fetched from empty buffers;
* the next K-step's 14 fragments double-buffered in registers;
* LDS rows are 128 B with the 16-byte chunk index XOR-ed by (row >> 1) & 7 (activations) resp. the same function of the MFMA
  row a weight row feeds: conflict-free ds_read_b128; the DMA lands lane l at byte 16 l of its piece, so the swizzle is in what
  each lane fetches;
* weight row n of a 32-column group feeds MFMA row 8 (i >> 2) + 4 (tile & 1) + (i & 3): a lane then holds 8 CONSECUTIVE
  output columns of its row in a tile pair - 16-byte stores (16 rows x 64 B per store instruction instead of 16 x 32 B).
Every output element is the same fp32 chain as in k_policy_fc (K ascending in 32-wide MFMA steps, bias added after the last
step): the two kernels agree to the bit (tests/test_gpu_round4.py).  Measured (profiles/r04c_probe_policy_fc1w.txt,
r04c_policy_fc1w_stamps.txt, r04c_ab_policy_fc.txt): 152-156 us against 175-200 for k_policy_fc in one process - and a
self-play step 0.6 % LONGER, because the trunk kernel around it then holds a lower clock; k_policy_fc stays the default.

The script checks the stream symbolically before writing it: ring slots by stage and "published by a barrier", fragment
registers by (stage, half, operand, tile), every accumulator's K-step sequence, slot refills only behind a barrier that every
read of the old stage had completed at, exact waits, the slot each toggled address register points at.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "chinesechessai_amd", "csrc", "xq_policy_fc1w_body.inc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_tower1wa import Ins, insert_lgkm_waits, as_c_string, CheckError      # noqa: E402

SLOT_STRIDE = 0x10000
STAGE_BYTES = 57344              # 256 activation rows + 192 weight rows, 128 B each
W_OFF = 32768
LDS_BYTES = SLOT_STRIDE + STAGE_BYTES
NA, NB = 6, 8                    # weight (A operand) / activation (B operand) tiles per wave
NPIECE = 14                      # per wave and stage: 8 activation pieces (8 rows x 128 B), 6 weight pieces
# DMA pieces of stage d + 2: behind these MFMAs (1-based) of K-step (d, 1) and of K-step (d + 1, 0)
# one-barrier forms (the barrier of K-step (d, 1) frees slot d % 2 AND publishes stage d + 1): (MFMAs of (d, 1), MFMAs of (d + 1, 0), None)
# two-barrier forms: K-step (d, 0) loads the fragments of (d, 1) one per MFMA and frees slot d % 2 with a barrier of its own
# behind MFMA b1 - the pieces of stage d + 2 start there, 22 MFMAs earlier: (MFMAs of (d, 0) behind b1, MFMAs of (d, 1), b1)
DMA_AT = {"spread": ([5, 10, 15, 20, 25, 30, 35, 40, 45], [4, 9, 14, 19, 24], None),
          "front": ([4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48], [4, 8], None),
          "front14": ([4, 7, 10, 13, 16, 19, 22, 25, 28, 31, 34, 37, 40, 43], [], None),
          "early": ([28, 32, 36, 40, 44, 48], [4, 8, 12, 16, 20, 24, 28, 32], 26),
          "early5": ([27, 32, 37, 42, 47], [4, 9, 14, 19, 24, 29, 34, 39, 44], 25)}
PLACEMENT = "early"

# VGPRs: fragment slot i of buffer b = v[4 (14 b + i) ..]; i < 6: weight tile, else activation tile
V_BIAS = 112                     # 24: bias[at][0..3]
V_STORE = 136                    # 8: byte offset of (row of tile bt, this lane's first column) in the output
V_FA = 144                       # 2: weight fragment LDS address for K half 0 / 1 (slot toggled behind every use)
V_FB = 146                       # 2: activation fragment LDS address
V_DMA = 148                      # 14: source offset of this lane in piece j
V_T = 162                        # temporaries 162..177
V_LAST = 177
S_RA, S_RW = 36, 40              # buffer resources: activations, weights
S_OUT = 44                       # s[44:45]
S_KOFF = 46                      # source offset (bytes along K) of the next stage to fetch
S_KMAX = 47
S_CNT = 48
S_LD = 49                        # 14: LDS offset of piece j inside a ring slot
S_SRC = 63                       # clamped source offset of the stage being fetched
S_SLOT = 64                      # LDS base of the slot being filled
S_T = 65                         # 65, 66
S_N2 = 67
S_MASK = 68                      # 8 x 2: rows of tile bt that exist
S_STAMP = 84                     # stamped bodies only: 6 x 2
S_AB, S_WB = 96, 97              # bytes of the activation / weight tile (the buffers' num_records while there is something to fetch)
S_LAST = 97


def FRAG(b, i):
    return "v[%d:%d]" % (4 * (14 * b + i), 4 * (14 * b + i) + 3)


def a_off(at):
    """LDS byte offset of weight tile `at` against the lane's base: MFMA row i of the tile reads weight row
    32 (at >> 1) + 4 (at & 1) + 8 (i >> 2) + (i & 3) of the wave's 96"""
    return (32 * (at >> 1) + 4 * (at & 1)) * 128


class Gen:
    def __init__(self, placement=PLACEMENT):
        self.ins = []
        self.placement = placement

    def add(self, text, kind="salu", **m):
        self.ins.append(Ins(text, kind, **m))

    def frag_loads(self, step):
        """the 14 fragment loads of K-step `step` (stage step >> 1, half step & 1) -> register buffer step & 1"""
        d, h = step >> 1, step & 1
        out = []
        for which, n, base, off in (("A", NA, V_FA + h, a_off), ("B", NB, V_FB + h, lambda bt: bt * 2048)):
            for i in range(n):
                idx = i if which == "A" else NA + i
                out.append(Ins("ds_read_b128 %s, v%d offset:%d" % (FRAG(h, idx), base, (W_OFF if which == "A" else 0) + off(i)), "ldsr",
                               dst=14 * h + idx, stage=d, half=h, which=which, idx=i, basereg=base))
        return out

    def toggle(self, h):
        """the address registers of K half h now point at the other ring slot"""
        self.add("v_xor_b32 v%d, 0x%x, v%d" % (V_FA + h, SLOT_STRIDE, V_FA + h), "toggle", reg=V_FA + h)
        self.add("v_xor_b32 v%d, 0x%x, v%d" % (V_FB + h, SLOT_STRIDE, V_FB + h), "toggle", reg=V_FB + h)

    def dma_m0(self, stage, j, first_of_all=False):
        if j == 0:
            if not first_of_all:
                self.add("s_xor_b32 s%d, s%d, 0x%x" % (S_SLOT, S_SLOT, SLOT_STRIDE), "slot")
            # a stage past the end (the uniform loop body asks for two of them) is fetched from empty buffers: every lane out
            # of range, no traffic, nothing for the epilogue to wait for
            self.add("s_min_u32 s%d, s%d, s%d" % (S_SRC, S_KOFF, S_KMAX))
            self.add("s_cmp_le_u32 s%d, s%d" % (S_KOFF, S_KMAX))
            self.add("s_cselect_b32 s%d, s%d, 0" % (S_RA + 2, S_AB))
            self.add("s_cselect_b32 s%d, s%d, 0" % (S_RW + 2, S_WB))
            self.add("s_add_u32 s%d, s%d, 128" % (S_KOFF, S_KOFF))
        self.add("s_add_u32 m0, s%d, s%d" % (S_LD + j, S_SLOT), "m0", stage=stage, piece=j)

    def dma_load(self, stage, j):
        rs = S_RA if j < 8 else S_RW
        self.add("buffer_load_dwordx4 v%d, s[%d:%d], s%d offen lds" % (V_DMA + j, rs, rs + 3, S_SRC), "dma", stage=stage, piece=j)

    def mfma(self, step, at, bt):
        b = step & 1
        t = 4 * (bt * NA + at)
        self.add("v_mfma_f32_16x16x32_bf16 a[%d:%d], %s, %s, a[%d:%d]" % (t, t + 3, FRAG(b, at), FRAG(b, NA + bt), t, t + 3), "mfma",
                 tile=bt * NA + at, a=14 * b + at, b=14 * b + NA + bt, step=step, at=at, bt=bt, want=None, first=False)

    def kstep(self, step):
        """48 MFMAs of K-step (d, h) from register buffer h.  Behind odd MFMAs 3.. the fragment loads of the next K-step, then
        the toggle of the address registers they used.  h = 1: behind MFMA 2 every LDS read of this wave has returned (the last
        reads of slot d % 2 among them), the wave's own pieces of stage d + 1 have landed, s_barrier: stage d + 1 is published
        and slot d % 2 is free for stage d + 2, whose first pieces follow; h = 0: the rest of the pieces of stage d + 1"""
        d, h = step >> 1, step & 1
        fill = self.frag_loads(step + 1)
        first, second, b1 = DMA_AT[self.placement]
        if b1 is None:
            dq = {k: (d + 2, j) for j, k in enumerate(first)} if h == 1 else {k: (d + 1, len(first) + j) for j, k in enumerate(second)}
        else:
            dq = {k: (d + 2, j) for j, k in enumerate(first)} if h == 0 else {k: (d + 2, len(first) + j) for j, k in enumerate(second)}
            assert all(k > b1 for k in first)
        k = 0
        toggled = False
        for bt in range(NB):
            for at in range(NA):
                k += 1
                if k in dq:
                    self.dma_m0(*dq[k])
                self.mfma(step, at, bt)
                if k == 2 and h == 1:
                    if b1 is None:
                        self.add("s_waitcnt lgkmcnt(0)", "lgkm0")
                    self.add("s_waitcnt vmcnt(?)", "vmwait", landed=d + 1)
                    self.add("s_barrier", "barrier", stage=d)
                if h == 0 and k == b1:
                    self.add("s_waitcnt lgkmcnt(0)", "lgkm0")
                    self.add("s_barrier", "barrier", stage=d)
                if k in dq:
                    self.dma_load(*dq[k])
                if k > 2 and (k % 2 == 1 or (h == 0 and b1 is not None)) and fill:
                    self.ins.append(fill.pop(0))
                elif not fill and not toggled and k % 2 == 0 and k != b1:
                    self.toggle((step + 1) & 1)
                    toggled = True
        assert not fill and toggled


def head():
    """operands: %0 s[2] activations of this tile's first row, %1 s bytes of them, %2 s[2] weights of this tile's first column,
    %3 s bytes, %4 s[2] out (element [first row][first column]), %5 s wave, %6 s K * 2, %7 s stages (K / 64), %8 s N * 2, %9 s
    rows of this tile that exist, %10 s[2] bias of the first column; %11 v weight fragment address (slot 0, K half 0), %12 v
    activation fragment address, %13 v source offset of this lane inside an activation piece, %14 v output offset of (row of
    tile 0, first column of this lane), %15 v row of this lane in tile 0 (relative to the tile), %16 v bias offset of this lane,
    %17 s[2] stamps (stamped bodies), %18 v source offset of this lane inside a weight piece"""
    g = Gen()
    g.add("; ==== k_policy_fc1w body (generated by tools/gen_policy_fc1w.py - do not edit) ====", "comment")
    g.add("s_mov_b64 s[%d:%d], %%0" % (S_RA, S_RA + 1))
    g.add("s_and_b32 s%d, s%d, 0xffff" % (S_RA + 1, S_RA + 1))
    g.add("s_mov_b32 s%d, %%1" % S_AB)
    g.add("s_mov_b32 s%d, 0x00020000" % (S_RA + 3))
    g.add("s_mov_b64 s[%d:%d], %%2" % (S_RW, S_RW + 1))
    g.add("s_and_b32 s%d, s%d, 0xffff" % (S_RW + 1, S_RW + 1))
    g.add("s_mov_b32 s%d, %%3" % S_WB)
    g.add("s_mov_b32 s%d, 0x00020000" % (S_RW + 3))
    g.add("s_mov_b64 s[%d:%d], %%4" % (S_OUT, S_OUT + 1))
    g.add("s_mov_b32 s%d, %%8" % S_N2)
    # bias of this lane's 8 columns in every tile pair (tile at = 2 p + e: columns 32 p + 8 q + 4 e ..)
    for at in range(NA):
        g.add("global_load_dwordx4 v[%d:%d], %%16, %%10 offset:%d" % (V_BIAS + 4 * at, V_BIAS + 4 * at + 3, (at >> 1) * 128 + (at & 1) * 16), "gload")
    # pieces of this wave: p = wave + 4 j (8 rows each); activations j < 8, weights j >= 8
    g.add("s_lshl_b32 s%d, %%5, 3" % S_T)                                  # wave * 8 rows
    g.add("s_mul_i32 s%d, s%d, %%6" % (S_T, S_T))                           # ... * K2
    g.add("s_lshl_b32 s%d, %%6, 5" % (S_T + 1))                             # 32 rows * K2: from piece j to piece j + 1
    for j in range(NPIECE):
        if j == 8:
            g.add("s_lshl_b32 s%d, %%5, 3" % S_T)
            g.add("s_mul_i32 s%d, s%d, %%6" % (S_T, S_T))
        g.add("v_add_u32 v%d, s%d, %s" % (V_DMA + j, S_T, "%13" if j < 8 else "%18"), "valu")
        g.add("s_add_u32 s%d, s%d, s%d" % (S_T, S_T, S_T + 1))
    g.add("s_lshl_b32 s%d, %%5, 10" % S_T)                                 # wave * 1024
    for j in range(NPIECE):
        g.add("s_add_u32 s%d, s%d, 0x%x" % (S_LD + j, S_T, (j * 4096) if j < 8 else W_OFF + (j - 8) * 4096))
    g.add("s_mov_b32 s%d, 0" % S_KOFF)
    g.add("s_mov_b32 s%d, 0" % S_SLOT)
    g.add("s_sub_u32 s%d, %%7, 1" % S_KMAX)
    g.add("s_lshl_b32 s%d, s%d, 7" % (S_KMAX, S_KMAX))                    # (stages - 1) * 128: fetches past the end re-read the last stage
    g.add("s_mov_b32 s%d, %%7" % S_CNT)
    return g


def prologue(placement):
    """stage 0 and the part of stage 1 that K-step (-1, 1) would have issued; under their flight: fragment addresses, output
    offsets and row masks, accumulators = 0; then stage 0's first fragments"""
    g = Gen(placement)
    g.add("; ---- stage 0 and the head of stage 1 on their way", "comment")
    for j in range(NPIECE):
        g.dma_m0(0, j, first_of_all=True)
        g.add("s_nop 0")
        g.dma_load(0, j)
    for j in range(len(DMA_AT[placement][0]) if DMA_AT[placement][2] is None else NPIECE):
        g.dma_m0(1, j)
        g.add("s_nop 0")
        g.dma_load(1, j)
    for h in range(2):
        g.add("v_xor_b32 v%d, 0x%x, %%11" % (V_FA + h, 64 * h), "valu")
        g.add("v_xor_b32 v%d, 0x%x, %%12" % (V_FB + h, 64 * h), "valu")
    g.add("s_lshl_b32 s%d, s%d, 4" % (S_T, S_N2))                          # 16 rows * N2
    g.add("v_mov_b32 v%d, %%14" % V_STORE, "valu")
    g.add("v_mov_b32 v%d, %%15" % V_T, "valu")
    for bt in range(NB):
        g.add("v_cmp_gt_i32 s[%d:%d], %%9, v%d" % (S_MASK + 2 * bt, S_MASK + 2 * bt + 1, V_T), "valu")
        if bt + 1 < NB:
            g.add("v_add_u32 v%d, s%d, v%d" % (V_STORE + bt + 1, S_T, V_STORE + bt), "valu")
            g.add("v_add_u32 v%d, 16, v%d" % (V_T, V_T), "valu")
    for t in range(NA * NB * 4):
        g.add("v_accvgpr_write_b32 a%d, 0" % t, "valu")
    g.add("s_waitcnt vmcnt(?)", "vmwait", landed=0)
    g.add("s_barrier", "barrier", stage=-1)
    g.ins += g.frag_loads(0)
    g.toggle(0)
    return g


def body(stage, placement):
    g = Gen(placement)
    g.kstep(2 * stage)
    g.kstep(2 * stage + 1)
    return g


def epilogue():
    """+ bias, bf16, 16-byte stores (lane: one row, 8 consecutive columns of a tile pair), rows that do not exist masked out"""
    g = Gen()
    g.add("; ---- epilogue", "comment")
    g.add("s_waitcnt vmcnt(0)", "vm0")            # (the fetches past the end: no DMA may outlive the workgroup's LDS)
    g.add("s_nop 15")
    g.add("s_nop 15")                                # the last MFMAs -> v_accvgpr_read
    n = 0
    for bt in range(NB):
        g.add("s_mov_b64 exec, s[%d:%d]" % (S_MASK + 2 * bt, S_MASK + 2 * bt + 1))
        for p in range(NA // 2):
            r = V_T + 8 * (n & 1)
            n += 1
            for e in range(2):
                at = 2 * p + e
                t = 4 * (bt * NA + at)
                for i in range(4):
                    g.add("v_accvgpr_read_b32 v%d, a%d" % (r + 4 * e + i, t + i), "valu", accread=(bt * NA + at, i))
            for i in range(8):
                g.add("v_add_f32 v%d, v%d, v%d" % (r + i, r + i, V_BIAS + 8 * p + i), "valu")
            for i in range(4):
                g.add("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (r + i, r + 2 * i, r + 2 * i + 1), "valu")
            g.add("global_store_dwordx4 v%d, v[%d:%d], s[%d:%d] offset:%d" % (V_STORE + bt, r, r + 3, S_OUT, S_OUT + 1, p * 64), "gstore")
    g.add("s_mov_b64 exec, -1")               # (the stores drain behind the wave's end)
    return g


def check(linear, nst):
    """symbolic run of prologue + nst stages + epilogue"""
    def fail(i, msg):
        ctx = "\n".join("   %s%s" % (">> " if j == i else "   ", linear[j].text) for j in range(max(0, i - 5), min(len(linear), i + 3)))
        raise CheckError("instruction %d: %s\n%s" % (i, msg, ctx))
    ring = {k: [None, False, []] for k in range(2)}          # slot -> [stage, published, lds op indices of its reads]
    regslot = {V_FA: 0, V_FA + 1: 0, V_FB: 0, V_FB + 1: 0}     # slot each fragment address register points at
    fill_slot = 0                                             # S_SLOT
    m0 = None
    frag = {}
    hist = {t: [] for t in range(NA * NB)}
    vq = []
    nl = retired = 0
    bar_retired = -1
    vdone = 0
    for i, ins in enumerate(linear):
        k, m = ins.kind, ins.m
        if k == "lgkmwait":
            retired = max(retired, nl - m["n"])
        elif k == "lgkm0":
            retired = nl
        elif k == "toggle":
            regslot[m["reg"]] ^= 1
        elif k == "slot":
            fill_slot ^= 1
        elif k == "m0":
            m0 = (m["stage"], m["piece"], fill_slot, i)
        elif k == "ldsr":
            s, sl = m["stage"], m["stage"] % 2
            if regslot[m["basereg"]] != sl:
                fail(i, "address register v%d points at slot %d, stage %d is in slot %d" % (m["basereg"], regslot[m["basereg"]], s, sl))
            if ring[sl][0] != s or not ring[sl][1]:
                fail(i, "fragment read of stage %d from slot %d holding %s (published %s)" % (s, sl, ring[sl][0], ring[sl][1]))
            ring[sl][2].append(nl)
            frag[m["dst"]] = ((s, m["half"], m["which"], m["idx"]), nl)
            nl += 1
        elif k == "mfma":
            s, h = m["step"] >> 1, m["step"] & 1
            fa, fb = frag.get(m["a"]), frag.get(m["b"])
            if fa is None or fa[0] != (s, h, "A", m["at"]) or fb is None or fb[0] != (s, h, "B", m["bt"]):
                fail(i, "MFMA of K-step (%d, %d) tile (%d, %d) reads %s, %s" % (s, h, m["at"], m["bt"], fa, fb))
            if fa[1] >= retired or fb[1] >= retired:
                fail(i, "MFMA operand not waited for")
            hist[m["tile"]].append(m["step"])
        elif k == "dma":
            s, sl = m["stage"], m["stage"] % 2
            if m0 is None or m0[:2] != (s, m["piece"]) or m0[2] != sl:
                fail(i, "M0 holds %s, the piece is (%d, %d) for slot %d" % (m0, s, m["piece"], sl))
            if not any(x.kind in ("mfma", "salu") and x.text.startswith(("v_mfma", "s_nop")) for x in linear[m0[3] + 1:i]):
                fail(i, "no wait state between the M0 write and the DMA")
            old = ring[sl][0]
            if old != s:
                if old is not None:
                    if old + 2 != s:
                        fail(i, "slot %d: stage %d replaces stage %d" % (sl, s, old))
                    if any(r > bar_retired for r in ring[sl][2]):
                        fail(i, "slot %d refilled while a read of stage %d was not complete at the last barrier" % (sl, old))
                ring[sl] = [s, False, []]
            vq.append(s)
            m0 = None
        elif k == "gload":
            vq.append(-1)
        elif k == "vm0":
            vdone = len(vq)
        elif k == "vmwait":
            X = m["landed"]
            idx = max([j for j, x in enumerate(vq) if x <= X], default=-1)
            n = len(vq) - 1 - idx
            if n > 63:
                fail(i, "vmcnt %d does not fit" % n)
            ins.text = "s_waitcnt vmcnt(%d)" % n
            m["n"] = n
            vdone = max(vdone, len(vq) - n)
        elif k == "barrier":
            done = vq[:vdone]
            for sl in range(2):
                s = ring[sl][0]
                if s is not None and done.count(s) >= NPIECE:
                    ring[sl][1] = True
            bar_retired = retired - 1
        elif k == "valu" and "accread" in m:
            t, _ = m["accread"]
            if hist[t] != list(range(2 * nst)):
                fail(i, "tile %d read with K-steps %s..." % (t, hist[t][:4]))
    for t in range(NA * NB):
        if hist[t] != list(range(2 * nst)):
            raise CheckError("tile %d: %d K-steps accumulated" % (t, len(hist[t])))
    return True


# timing-only bodies (wrong results; -DXQ_TOWER_PROBES=1 library): name -> (DMA placement, what is dropped / added)
ABLATIONS = (("STAMPED", PLACEMENT, "stamps"), ("NODMA_STAMPED", PLACEMENT, "dma+stamps"), ("FRONT", "front", None), ("EVEN", "early5", None),
             ("NODMA", PLACEMENT, "dma"), ("NOMFMA", PLACEMENT, "mfma"), ("NOMFMA_NOLDS", "front14", None))


def linear_stream(nst, placement=PLACEMENT):
    """prologue + nst loop body instances + epilogue as one instruction list with the LDS waits in place (what check() runs)"""
    pro = prologue(placement)
    bodies = [body(d, placement) for d in range(nst)]
    epi = epilogue()
    linear = list(pro.ins)
    for b in bodies:
        linear += b.ins
    linear += epi.ins
    return pro, bodies, epi, insert_lgkm_waits(linear)


def generate(name="XQ_FC1W_BODY", placement=PLACEMENT, drop=None):
    nst = 4
    pro, bodies, epi, linear = linear_stream(nst, placement)
    check(linear, nst)

    # the loop body is emitted once: every instance must be the same text
    def section(first, last):
        i0 = next(i for i, x in enumerate(linear) if x is first)
        i1 = next(i for i, x in enumerate(linear) if x is last)
        while i0 > 0 and linear[i0 - 1].kind == "lgkmwait":
            i0 -= 1
        return linear[i0:i1 + 1]
    secs = [[x.text for x in section(b.ins[0], b.ins[-1])] for b in bodies]
    for k in range(1, nst):
        if secs[k] != secs[0]:
            for a, b in zip(secs[0], secs[k]):
                if a != b:
                    raise CheckError("loop body instances 0 and %d differ: %s | %s" % (k, a, b))
            raise CheckError("loop body instances differ in length")
    t0 = secs[0]
    lines = [x.text for x in head().ins] + [x.text for x in section(pro.ins[0], pro.ins[-1])]
    lines.append("XQFC1W_LOOP_%=:")
    lines += t0
    lines.append("s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT))
    lines.append("s_cmp_gt_u32 s%d, 0" % S_CNT)
    lines.append("s_cbranch_scc1 XQFC1W_LOOP_%=")
    lines += [x.text for x in section(epi.ins[0], epi.ins[-1])]
    if drop and drop.endswith("stamps"):
        # shader cycles at: start, stage 0 landed, loop done, all stored; 100 MHz ticks start -> end.  Every lane of wave w stores
        # the same 6 x 8 bytes at %17 + 64 w
        def memtime(k, real=False):
            return ["s_mem%stime s[%d:%d]" % ("real" if real else "", S_STAMP + 2 * k, S_STAMP + 2 * k + 1), "s_waitcnt lgkmcnt(0)"]
        i_loop = lines.index("XQFC1W_LOOP_%=:")
        i_epi = lines.index("; ---- epilogue")
        tail = ["s_waitcnt vmcnt(0)"] + memtime(3) + memtime(5, True)
        tail += ["s_lshl_b32 s%d, %%5, 6" % S_T, "v_mov_b32 v%d, s%d" % (V_T, S_T)]
        for k in range(6):
            tail += ["v_mov_b32 v%d, s%d" % (V_T + 2, S_STAMP + 2 * k), "v_mov_b32 v%d, s%d" % (V_T + 3, S_STAMP + 2 * k + 1),
                     "global_store_dwordx2 v%d, v[%d:%d], %%17 offset:%d" % (V_T, V_T + 2, V_T + 3, 8 * k)]
        tail += ["s_waitcnt vmcnt(0)"]
        lines = (lines[:1] + memtime(0) + memtime(4, True) + lines[1:i_loop] + memtime(1) + lines[i_loop:i_epi] + memtime(2) + lines[i_epi:] + tail)
        drop = drop.split("+")[0] if "+" in drop else "stamps"
    if drop == "dma":                                  # no operand DMA behind the prologue: the MFMA stream and its fragment reads alone
        i0 = lines.index("XQFC1W_LOOP_%=:")
        lines = lines[:i0] + ["s_nop 0" if l.startswith("buffer_load_dwordx4") else l for l in lines[i0:]]
    elif drop in ("mfma", "mfma_lds"):                 # no MFMAs: the operand delivery and the fragment reads alone
        lines = [l for l in lines if not l.startswith("v_mfma")]
        if drop == "mfma_lds":                         # ... and no fragment reads
            lines = [l for l in lines if not l.startswith("ds_read")]
    if drop or name != "XQ_FC1W_BODY":
        return as_c_string(lines, name)
    text = ("// xq_policy_fc1w_body.inc - GENERATED by tools/gen_policy_fc1w.py; do not edit.\n"
            "// %d instructions, %d MFMAs per loop body = one K-stage of 64; a %d-stage instance passed the symbolic check\n" % (
                sum(1 for l in lines if not l.startswith(";") and not l.endswith(":")), sum(1 for l in t0 if l.startswith("v_mfma")), nst))
    text += as_c_string(lines, "XQ_FC1W_BODY")
    text += "#define XQ_FC1W_LDS_BYTES %d\n#define XQ_FC1W_V_LAST %d\n#define XQ_FC1W_S_FIRST %d\n#define XQ_FC1W_S_LAST %d\n" % (
        LDS_BYTES, V_LAST, S_RA, S_LAST)
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--ablations", action="store_true", help="also write the timing-only bodies (xq_policy_fc1w_abl.inc, not committed)")
    args = ap.parse_args()
    if args.ablations:
        abl = "// xq_policy_fc1w_abl.inc - GENERATED by tools/gen_policy_fc1w.py --ablations; timing-only bodies, wrong results.\n"
        for nm, placement, drop in ABLATIONS:
            abl += generate("XQ_FC1W_BODY_" + nm, placement, drop)
        open(OUT.replace("_body.inc", "_abl.inc"), "w").write(abl)
    text = generate()
    if args.check:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("xq_policy_fc1w_body.inc is stale: run python tools/gen_policy_fc1w.py")
            return 1
        print("ok")
        return 0
    open(OUT, "w").write(text)
    print("wrote", OUT, len(text), "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
