#!/usr/bin/env python3
"""gen_tower1wa.py - generator + checker of the hand-written gfx950 layer body of k_tower1wa (csrc/xq_tower1wa.hpp).

    python tools/gen_tower1wa.py            # writes chinesechessai_amd/csrc/xq_tower1wa_body.inc after checking it
    python tools/gen_tower1wa.py --check    # regenerates in memory, checks, compares with the committed file

k_tower1wa is the one-wave-per-SIMD trunk kernel (4 boards per 256-thread workgroup, wave tile 128 channels x 96
pixels = 8 x 6 MFMA tiles of v_mfma_f32_16x16x32_bf16, 192 accumulator registers on a[0:191]) whose residual tower -
every layer's main loop, epilogue, weight DMA, stage barriers and the loop over the blocks - is ONE asm statement
that the compiler cannot interleave with.  What the assembly does that the HIP form (k_tower1w, round 3) could not:

  * a layer's epilogue runs UNDER the next layer's first tap.  The K-steps of a layer are ordered (tap, ks) with ks =
    the input channel group of 32, which is the output channel pair J = ks of the layer before.  As soon as pair J has
    been drained (accumulators -> bf16 -> ReLU -> LDS rows in place) and its accumulators hold the next layer's bias,
    the MFMAs of the next layer's tap 0 whose OUTPUT tiles are in a drained pair and whose INPUT group is a stored pair
    can issue: group (P, K) = 12 MFMAs is ready after drain max(P, K).  Every accumulator still sees its K-steps in the
    order (tap 0: ks 0, 1, 2, 3), (tap 1: ...) - the fp32 sums are the ones k_tower16b forms, bit for bit - but the
    drain's ~100 VALU + 18 LDS instructions per pair issue in the gaps of those MFMAs;
  * every s_waitcnt is an exact count (inserted by this script) and every hazard distance is checked.

The script is also the kernel's checker: it runs the instruction stream of a 3-block tower through a symbolic model
(LDS rows by (layer, pair) of the data they hold, ring slots by stage, fragment registers by what was loaded into
them, accumulators by the ordered list of products added) and fails if an MFMA would read a fragment that is not the
(layer, tap, ks, tile) it should be, an accumulator's history is out of order, an LDS row is overwritten before its
last reader was issued or read before its writer, a ring slot is refilled while a wave may still read it, a register
is overwritten before its last consumer issued, or an MFMA result is read too early.

Reference arithmetic: /root/reference/neural_network.py:181-187 (ResidualBlock.forward), eval-mode BatchNorm folded.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "chinesechessai_amd", "csrc", "xq_tower1wa_body.inc")

# ---- LDS image (bytes; the same as k_tower1w plus the junk rows) -----------------------------------------------------------
WBUF = 16384                      # one weight stage [128 cout][64 cin] bf16
ACT_BYTES = 90 * 256
ACT0 = 4 * WBUF
ZROW = ACT0 + 4 * ACT_BYTES       # 157,696
BIAS = ZROW + 256                 # [2][128] f32
JUNK = 159744                     # 1 KB: where lanes of pixel tile 5 that hold no pixel store (156 * 1024)
LDS_TOTAL = JUNK + 1024

# ---- register plan ------------------------------------------------------------------------------------------------------
# fragment pool: 30 slots of 4 VGPRs.  Main loop: buffer b (0 / 1), weight fragment m = slot 14 b + m, activation
# fragment n = slot 14 b + 8 + n.  Skewed tap 0: activation fragment (K, n) = slot 6 K + n, weight fragments of the
# groups in slots 24..29, three groups in turn.
SLOT_REG = [4 * s for s in range(28)] + [246, 250]
def SLOT(s):
    return "v[%d:%d]" % (SLOT_REG[s], SLOT_REG[s] + 3)
V_TAB0 = 112                       # first register filled from the lane table (TAB_DWORDS dwords, in this order)
V_TA = 112                         # ta[tap * 6 + nt]: activation fragment address of (tap, pixel tile), K-step 0
V_A0, V_A1 = 166, 167              # weight fragment address, K-step parity 0 / 1
V_SB = 168                         # sb[nt]: this lane's 16-byte chunk of pixel nt * 16 + r, channel pair 0 (pair J: ^ (J << 5))
V_SBW5 = 174                       # store address of tile 5 (in JUNK for lanes without a pixel)
V_LBQ = 175                        # BIAS + (lane >> 4) * 32
V_SEL = 176                        # skip-connection selectors, two fragments
V_WSRC = 184                       # weight DMA source offset of this lane: even / odd pieces
V_L16 = 186                        # lane * 16 (bias DMA source offset)
V_WSRC4 = (184, 185, 187, 245)     # weight DMA source offset of piece j of this wave: V_WSRC[j & 1] + the piece's offset in a stage (set up in the head)
TAB_DWORDS = 76
V_XT = 188                         # 6 address temporaries
V_RD = (194, 202)                  # 2 x 8 accumulator read-outs
V_PK = (210, 214)                  # 2 x 4 packed results
V_XF = 218                         # 6 x 4: block input x of a pair (skip connection)
V_ST = 242                         # stamps: 242..244
V_LAST = 253
A_BIAS = 192                       # a[192:223]: the next layer's bias row in accumulator layout, 4 registers per weight tile mt
A_LAST = 223

S_RSRC = 36                        # s[36:39] buffer resource of the weight stream
S_BLK = 40                         # byte offset of the current block's first stage in the weight stream
S_POFF = 45                        # [4] source offset of this wave's piece j inside a stage
S_LDST = 49                        # [4] LDS offset of this wave's piece j inside a ring slot
S_CNT = 57                         # blocks left (this one included)
S_BIASP = 58                       # s[58:59] global address of the next bias row to fetch
S_WAVE = 60
S_T = 61                           # temporaries s61, s62
S_STG = 63                         # source offset of the stage whose pieces are being issued
S_MAXB = 53                        # source offset of the LAST stage of the tower (clamp of the fetches that run past its end)
S_TIME = 64                        # s[64:65] s_memtime
S_STAMP = 66                       # s[66:67] stamp slot 3 + 4 blk of this workgroup
S_STAMP0 = 68                      # s[68:69] stamp slot 0 of this workgroup (fine stamps inside a transition)
S_LAST = 69
BLOCK_BYTES = 18 * 32768           # weight bytes of a block = 36 stages


# timing-only ablations (wrong results; bodies for the probes library, never checked, never in the product):
# "noaccread" drains read VGPRs instead of accumulators, "nostore" no ds_write in the drains, "nodrain" no drain at all,
# "nodma" no weight DMA, "nobar" no s_barrier
ABL = set()


class Ins:
    __slots__ = ("text", "kind", "m")

    def __init__(self, text, kind, **m):
        self.text, self.kind, self.m = text, kind, m


TILE = lambda mt, n: mt * 6 + n


class Emitter:
    """instruction list of one code section; stage numbers are LOCAL to the current block (0..35, the skew / tail of the
    odd layer reach 36..40 = the next block's 0..4), layers are 'E' (even: first convolution of a block) / 'O'"""

    def __init__(self, stamps):
        self.ins = []
        self.stamps = stamps
        self.nlabel = 0

    def add(self, text, kind="salu", **m):
        self.ins.append(Ins(text, kind, **m))

    def comment(self, s):
        self.add("; " + s, "comment")

    NLABEL = [0]
    NVALU = [0]
    NDMA = [0]

    def label(self):
        Emitter.NLABEL[0] += 1
        return "XQ1WA_%d_%%=" % Emitter.NLABEL[0]      # %= : unique per asm statement instance

    # ---- primitives ------------------------------------------------------------------------------------------------------
    def mfma(self, tile, a, b, want, first=False):
        """acc[tile] += A x B.  a / b: fragment slot numbers, or 'sel0' / 'sel1' / 'xf<n>'.  want: what the model expects:
        (layer tag, tap, ks) or ('skip', layer tag).  first: the tile's first MFMA of a layer - C is the layer's bias row
        (a[A_BIAS + 4 mt : +3], loaded once per layer) instead of the tile itself: acc[tile] = A x B + bias."""
        def reg(x):
            if isinstance(x, int):
                return SLOT(x)
            if x.startswith("sel"):
                o = int(x[3:])
                return "v[%d:%d]" % (V_SEL + 4 * o, V_SEL + 4 * o + 3)
            n = int(x[2:])
            return "v[%d:%d]" % (V_XF + 4 * n, V_XF + 4 * n + 3)
        # work-removal probes (timing only): "drop2" / "drop4" leave out the convolution MFMAs of 2 / 4 of the 8 channel tiles of
        # pixel tile 5 - 46 / 44 MFMAs per K-step instead of 48: what shedding the padding rows' work would be worth under the
        # power cap
        if want[0] != "skip" and tile % 6 == 5 and (("drop2" in ABL and tile // 6 >= 6) or ("drop4" in ABL and tile // 6 >= 4)):
            return
        t = 4 * tile
        c = A_BIAS + 4 * (tile // 6) if first else t
        self.add("v_mfma_f32_16x16x32_bf16 a[%d:%d], %s, %s, a[%d:%d]" % (t, t + 3, reg(a), reg(b), c, c + 3), "mfma",
                 tile=tile, a=a, b=b, want=want, first=first)
        # Winograd go / no-go probe (timing only): k independent f32 adds behind every MFMA = the lane-local input / output
        # transforms F(2x2, 3x3) would put beside it
        for fl in ABL:
            if fl.startswith("valu"):
                for k in range(int(fl[4:])):
                    r = V_RD[0] + (Emitter.NVALU[0] % 24)
                    Emitter.NVALU[0] += 1
                    self.add("v_add_f32 v%d, v%d, v%d" % (r, r, V_SEL + (k & 7)), "valu", wr=[r])

    def load_a(self, slot, lstage, kk, mt, want):
        """weight fragment mt of local stage lstage, K-step parity kk -> slot"""
        rs = lstage & 3
        self.add("ds_read_b128 %s, v%d offset:%d" % (SLOT(slot), V_A1 if kk else V_A0, rs * WBUF + mt * 2048), "ldsr",
                 dst=slot, src=("ring", lstage, kk, mt), want=want)

    def load_b(self, slot, tap, ks, n, want):
        """activation fragment of pixel tile n for (tap, ks) -> slot"""
        src = V_TA + tap * 6 + n
        if ks:
            self.add("v_xor_b32 v%d, %d, v%d" % (V_XT + n, ks << 5, src), "valu", wr=[V_XT + n])
            src = V_XT + n
        self.add("ds_read_b128 %s, v%d offset:%d" % (SLOT(slot), src, n * 4096), "ldsr", dst=slot, src=("act", ks, tap, n), want=want)

    def dma_piece(self, lstage, j, clamp=False):
        """1-KB piece j of this wave of local stage lstage -> its ring slot.  Two instructions per piece: the LDS destination
        into M0 - hoisted in front of the MFMA that precedes the piece, which is the wait state the M0 write needs - and the
        load; the stage's source offset is computed once per stage (piece 0), the piece's own offset sits in the lane's
        VGPR offset (V_WSRC4[j])."""
        rs = lstage & 3
        imm = (lstage >> 1) * 32768 + (lstage & 1) * 128
        m0 = Ins("s_add_u32 m0, s%d, 0x%x" % (S_LDST + j, rs * WBUF), "salu")
        k = len(self.ins) - 1
        while k >= 0 and self.ins[k].kind != "mfma" and "m0" not in self.ins[k].text and self.ins[k].kind not in ("label", "barrier"):
            k -= 1
        hoisted = k >= 0 and self.ins[k].kind == "mfma"
        if hoisted:
            self.ins.insert(k, m0)
        else:
            self.ins.append(m0)
        if j == 0:
            self.add("s_add_u32 s%d, s%d, 0x%x" % (S_STG, S_BLK, imm))
            if clamp:
                self.add("s_min_u32 s%d, s%d, s%d" % (S_STG, S_STG, S_MAXB))
        elif not hoisted:
            self.add("s_nop 0")
        S_Tx = S_STG
        if "dma35" in ABL:
            # Winograd probe: the transformed weight set (16 positions, 512 KB per layer) streamed once per 2 boards instead of
            # 288 KB once per 4: 3.5 x the pieces
            Emitter.NDMA[0] += 1
            for _ in range(3 if Emitter.NDMA[0] & 1 else 2):
                self.add("buffer_load_dwordx4 v%d, s[%d:%d], s%d offen lds" % (V_WSRC4[j], S_RSRC, S_RSRC + 3, S_STG), "dma_extra")
        if "nodma" in ABL:
            self.add("s_nop 0", "dma", lstage=lstage, piece=j)
        else:
            self.add("buffer_load_dwordx4 v%d, s[%d:%d], s%d offen lds" % (V_WSRC4[j], S_RSRC, S_RSRC + 3, S_STG), "dma", lstage=lstage, piece=j)

    def bias_dma(self, slot, lrow, skip_if_last_block=False):
        """wave 1, lanes 0..31: the next bias row -> bias slot `slot` (lrow: tag of the layer whose bias it is)"""
        lab = self.label()
        if skip_if_last_block:
            self.add("s_cmp_eq_u32 s%d, 1" % S_CNT)
            self.add("s_cbranch_scc1 " + lab)
        self.add("s_cmp_lg_u32 s%d, 1" % S_WAVE)
        self.add("s_cbranch_scc1 " + lab)
        self.add("s_mov_b32 m0, 0x%x" % (BIAS + slot * 512))
        self.add("s_mov_b64 exec, 0xffffffff")
        self.add("global_load_lds_dwordx4 v%d, s[%d:%d]" % (V_L16, S_BIASP, S_BIASP + 1), "dma_bias", slot=slot, lrow=lrow)
        self.add("s_mov_b64 exec, -1")
        self.add(lab + ":", "label")
        self.add("s_add_u32 s%d, s%d, 512" % (S_BIASP, S_BIASP))
        self.add("s_addc_u32 s%d, s%d, 0" % (S_BIASP + 1, S_BIASP + 1))

    def barrier(self, landed, note):
        """own pieces of local stage `landed` and of everything older have arrived (exact vmcnt filled in later), then the
        workgroup barrier"""
        self.add("s_waitcnt vmcnt(?)", "vmwait", landed=landed)
        self.add("s_nop 0" if "nobar" in ABL else "s_barrier", "barrier", note=note)

    def stamp(self, k, absolute=False):
        """stamped bodies: s_memtime -> stamp slot 3 + 4 blk + k of the workgroup (absolute: slot k)"""
        if not self.stamps:
            return
        self.add("s_memtime s[%d:%d]" % (S_TIME, S_TIME + 1), "smem")
        self.add("s_waitcnt lgkmcnt(0)", "lgkm0")
        self.add("s_mov_b64 exec, 1")
        self.add("v_mov_b32 v%d, s%d" % (V_ST, S_TIME), "valu", wr=[V_ST])
        self.add("v_mov_b32 v%d, s%d" % (V_ST + 1, S_TIME + 1), "valu", wr=[V_ST + 1])
        self.add("v_mov_b32 v%d, 0" % (V_ST + 2), "valu", wr=[V_ST + 2])
        base = S_STAMP0 if absolute else S_STAMP
        self.add("global_store_dwordx2 v%d, v[%d:%d], s[%d:%d] offset:%d" % (V_ST + 2, V_ST, V_ST + 1, base, base + 1, k * 8), "gstore")
        self.add("s_mov_b64 exec, -1")

    # ---- one regular tap: 4 K-steps = 2 weight stages ------------------------------------------------------------------------
    def tap_regular(self, lt, lbase, tap, first_load, prefetch_next, clamp_from=None, ks_stop=4):
        """K-steps (tap, 0..3) of the layer tagged lt (first local stage lbase) in k_tower1w's issue order: per pixel tile n
        8 MFMAs with one filler per MFMA gap - the next K-step's activation fragment of tile n, one or two of its weight
        fragments, a DMA piece.  first_load: nothing prefetched this tap's first K-step; prefetch_next: the last K-step
        prefetches (tap + 1, 0)."""
        if first_load:
            for mt in range(8):
                self.load_a(mt, lbase + 2 * tap, 0, mt, (lt, tap, 0, mt))
            for n in range(6):
                self.load_b(8 + n, tap, 0, n, (lt, tap, 0, n))
        for ks in range(ks_stop):
            cur, nxt = ks & 1, (ks & 1) ^ 1
            sl, kk = ks >> 1, ks & 1
            p = 2 * tap + sl
            last = ks == 3 and not prefetch_next
            for n in range(6):
                def lda(m):
                    if kk == 0:
                        self.load_a(14 * nxt + m, lbase + p, 1, m, (lt, tap, ks + 1, m))
                    elif ks == 1:
                        self.load_a(14 * nxt + m, lbase + p + 1, 0, m, (lt, tap, ks + 1, m))
                    else:
                        self.load_a(14 * nxt + m, lbase + p + 1, 0, m, (lt, tap + 1, 0, m))
                m0 = m1 = -1
                if kk == 0:
                    if n < 2:
                        m0, m1 = 2 * n, 2 * n + 1
                    else:
                        m0 = n + 2
                elif n >= 1:
                    if n < 4:
                        m0, m1 = 2 * n - 2, 2 * n - 1
                    else:
                        m0 = n + 2
                if last:
                    m0 = m1 = -1
                bslot = 14 * cur + 8 + n
                self.mfma(TILE(0, n), 14 * cur + 0, bslot, (lt, tap, ks))
                if not last:
                    if ks < 3:
                        self.load_b(14 * nxt + 8 + n, tap, ks + 1, n, (lt, tap, ks + 1, n))
                    else:
                        self.load_b(14 * nxt + 8 + n, tap + 1, 0, n, (lt, tap + 1, 0, n))
                self.mfma(TILE(1, n), 14 * cur + 1, bslot, (lt, tap, ks))
                if m0 >= 0:
                    lda(m0)
                self.mfma(TILE(2, n), 14 * cur + 2, bslot, (lt, tap, ks))
                if m1 >= 0:
                    lda(m1)
                self.mfma(TILE(3, n), 14 * cur + 3, bslot, (lt, tap, ks))
                if kk == 1 and 1 <= n <= 4:
                    ls = lbase + p + 3
                    self.dma_piece(ls, n - 1, clamp_from is not None and ls >= clamp_from)
                for mt in range(4, 8):
                    self.mfma(TILE(mt, n), 14 * cur + mt, bslot, (lt, tap, ks))
                if kk == 1 and n == 0:
                    # B_p: stage p + 1 has landed for every wave, every wave has left stage p - 1 (its slot is refilled next)
                    self.barrier(lbase + p + 1, "B%d" % p)

    # ---- drain of one channel pair ----------------------------------------------------------------------------------------
    def drain_ops(self, J, rx, bias_slot, final):
        """filler list of pair J's drain: per pixel tile 8 accumulator reads, 4 conversions, 4 ReLUs, [the block input x of
        the rows about to be overwritten,] one 16-byte store, two bias loads straight into the accumulators"""
        ops = []
        for n in range(6):
            st = n & 1
            ops += [("accread", J, n, k, st) for k in range(8)]
            ops += [("cvt", J, n, k, st) for k in range(4)]
            ops += [("relu", J, n, k, st) for k in range(4)]
            if rx:
                ops.append(("xread", J, n))
            ops.append(("store", J, n, st))
        return ops

    def emit_op(self, op, lt, lt_next):
        kind = op[0]
        if "nodrain" in ABL and kind != "biasrow":
            return
        if kind == "xread":
            _, J, n = op
            a = V_SB + n
            if J:
                self.add("v_xor_b32 v%d, %d, v%d" % (V_XT + n, J << 5, V_SB + n), "valu", wr=[V_XT + n])
                a = V_XT + n
            self.add("ds_read_b128 v[%d:%d], v%d" % (V_XF + 4 * n, V_XF + 4 * n + 3, a), "ldsr", dst="xf%d" % n, src=("actrow", J, n),
                     want=("XIN", lt, J, n))
        elif kind == "accread":
            _, J, n, k, st = op
            tile = TILE(2 * J + (k >> 2), n)
            if "noaccread" in ABL:
                self.add("v_mov_b32 v%d, v%d" % (V_RD[st] + k, V_SEL + (k & 3)), "valu", wr=[V_RD[st] + k], accread=(tile, k & 3), layer=lt)
            else:
                self.add("v_accvgpr_read_b32 v%d, a%d" % (V_RD[st] + k, 4 * tile + (k & 3)), "valu", wr=[V_RD[st] + k], accread=(tile, k & 3), layer=lt)
        elif kind == "cvt":
            _, J, n, k, st = op
            self.add("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (V_PK[st] + k, V_RD[st] + 2 * k, V_RD[st] + 2 * k + 1), "valu",
                     wr=[V_PK[st] + k], rd=[V_RD[st] + 2 * k, V_RD[st] + 2 * k + 1], cvt=(J, n, k))
        elif kind == "relu":
            _, J, n, k, st = op
            self.add("v_pk_max_i16 v%d, v%d, 0" % (V_PK[st] + k, V_PK[st] + k), "valu", wr=[V_PK[st] + k], rd=[V_PK[st] + k], relu=(J, n, k))
        elif kind == "store":
            _, J, n, st = op
            a = V_SBW5 if n == 5 else V_SB + n
            if J:
                self.add("v_xor_b32 v%d, %d, v%d" % (V_XT + n, J << 5, a), "valu", wr=[V_XT + n])
                a = V_XT + n
            if "nostore" in ABL:
                self.add("s_nop 0", "ldsw_skipped", store=(lt, J, n), data=V_PK[st])
            else:
                self.add("ds_write_b128 v%d, v[%d:%d]" % (a, V_PK[st], V_PK[st] + 3), "ldsw", store=(lt, J, n), data=V_PK[st])
        elif kind == "biasrow":
            mt, bslot = op[1], op[2]
            self.add("ds_read_b128 a[%d:%d], v%d offset:%d" % (A_BIAS + 4 * mt, A_BIAS + 4 * mt + 3, V_LBQ, bslot * 512 + (mt >> 1) * 128 + (mt & 1) * 16),
                     "ldsr", dst="bias%d" % mt, src=("bias", bslot), want=("BIASROW", lt_next, mt))
        else:
            raise ValueError(kind)

    # ---- epilogue of layer lt under tap 0 of the next layer -------------------------------------------------------------------
    def last_stage(self, lt, lbase, pair_major, fillers):
        """The layer's last weight stage = K-steps (tap 8, ks 2) and (tap 8, ks 3).  tap_regular(tap 8, ks_stop=2) left ks 2's
        fragments in buffer 0; ks 3's go to buffer 1 here (the regular prefetch, moved to the head of the region).
        pair_major: output pair by output pair - P0: ks 2, ks 3; P1: ks 2, ks 3; ... - every accumulator still sees ks 2 before
        ks 3, and pair 0 is complete after 24 of the 96 MFMAs, pair 1 after 48, pair 2 after 72: `fillers` (the next layer's bias
        row loads, drain 0, drain 1 in this order) issue two per MFMA from MFMA 36 on - a drain may start 24 instructions behind
        its pair's last MFMA - under the MFMAs of the later pairs.  Carries the stage's barrier and DMA pieces like the second
        K-step of every stage.  Returns the fillers it did not place."""
        tap, p = 8, 17
        fillers = list(fillers)
        k3 = [("b", n) for n in range(6)] + [("a", m) for m in range(8)]          # ks 3's loads, in the order of their first use
        dma = [(lbase + p + 3, j) for j in range(4)]
        order = []
        if pair_major:
            for P in range(4):
                for ks in (2, 3):
                    for n in range(6):
                        for o in range(2):
                            order.append((ks, 2 * P + o, n))
        else:
            for ks in (2, 3):
                for n in range(6):
                    for mt in range(8):
                        order.append((ks, mt, n))
        for k, (ks, mt, n) in enumerate(order, 1):
            buf = 14 * (ks & 1)
            self.mfma(TILE(mt, n), buf + mt, buf + 8 + n, (lt, tap, ks))
            if k3:
                kind, i = k3.pop(0)
                if kind == "b":
                    self.load_b(14 + 8 + i, tap, 3, i, (lt, tap, 3, i))
                else:
                    self.load_a(14 + i, lbase + p, 1, i, (lt, tap, 3, i))
            if k == 4:
                self.barrier(lbase + p + 1, "B%d" % p)              # (the place it has in the regular order: behind the first MFMAs)
            elif 16 <= k <= 28 and k % 4 == 0 and dma:
                ls, j = dma.pop(0)
                self.dma_piece(ls, j, clamp=not pair_major)
            elif pair_major and k > 36:
                for _ in range(2):
                    if fillers:
                        f = fillers.pop(0)
                        self.emit_op(f, lt, None if f[0] != "biasrow" else f[3])
        assert not dma and not k3
        return fillers

    def skew(self, lt, lt_next, lbase_next, rx, bias_slot, bias_fetch_slot, bias_fetch_tag, bias_skip_last, stamp_k, lead_in=None, fine=None):
        """drain of layer lt (pairs 0..3) with tap 0 of layer lt_next (first local stage lbase_next) issued as its operands
        become ready.  MFMA group (P, K) = output pair P x input group K, 12 MFMAs, ready after drain max(P, K); inside a
        pair the input groups ascend, so every accumulator sees (tap 0: ks 0, 1, 2, 3) in order.  Segment J = the groups
        that drain J makes ready; its MFMAs issue while drain J + 1 is emitted around them.
        Fragment slots: activation fragment (K, n) = slot 6 K + n (resident until its last group), weight fragments of
        group i = slots 24 + 2 (i mod 3)."""
        self.comment("---- epilogue of layer %s under tap 0 of layer %s" % (lt, lt_next))
        head = [("biasrow", mt, bias_slot, lt_next) for mt in range(8)] + self.drain_ops(0, rx, bias_slot, False)
        d1 = self.drain_ops(1, rx, bias_slot, False)
        if lead_in is not None:
            # the layer's last stage pair-major, with the bias row loads, drain 0 and the head of drain 1 under its MFMAs
            n0 = len(head)
            left = self.last_stage(lt, lead_in, True, head + d1)
            used = n0 + len(d1) - len(left)
            head, d1 = (head[used:], d1) if used < n0 else ([], d1[used - n0:])
        # B_E: every wave has left the main loop (its last stage's slot may be refilled: the DMA of stage 3 follows) and the
        # next layer's stages 0 and 1 have landed
        self.barrier(lbase_next + 1, "BE")
        if fine is not None:
            self.stamp(fine, absolute=True)           # fine stamps (stamped bodies): behind the last stage, then behind every segment
        segs = [[(0, 0)], [(1, 0), (0, 1), (1, 1)], [(2, 0), (2, 1), (0, 2), (1, 2), (2, 2)],
                [(3, 0), (3, 1), (3, 2), (0, 3), (1, 3), (2, 3), (3, 3)]]
        order = [g for sg in segs for g in sg]
        abuf = {g: 24 + 2 * (i % 3) for i, g in enumerate(order)}
        loaded_b = set()
        loads_of, mfmas_of = {}, {}
        for g in order:
            P, K = g
            lo = []
            if K not in loaded_b:
                loaded_b.add(K)
                lo += [("ldb", K, n) for n in range(6)]
            lo += [("lda", P, K, o, abuf[g] + o) for o in range(2)]
            loads_of[g] = lo
            # a tile's first MFMA of the layer takes the bias row as C: the selector MFMA behind an even layer, else group (P, 0)
            mfmas_of[g] = [("mfma", TILE(2 * P + o, n), abuf[g] + o, 6 * K + n, (lt_next, 0, K), (not rx) and K == 0)
                           for n in range(6) for o in range(2)]

        def emit_f(f):
            if f[0] == "ldb":
                _, K, n = f
                self.load_b(6 * K + n, 0, K, n, (lt_next, 0, K, n))
            elif f[0] == "lda":
                _, P, K, o, sl = f
                self.load_a(sl, lbase_next + (K >> 1), K & 1, 2 * P + o, (lt_next, 0, K, 2 * P + o))
            elif f[0] == "pfa":
                self.load_a(f[1], lbase_next + 2, 0, f[1], (lt_next, 1, 0, f[1]))
            elif f[0] == "pfb":
                self.load_b(8 + f[1], 1, 0, f[1], (lt_next, 1, 0, f[1]))
            elif f[0] == "dma":
                self.dma_piece(f[1], f[2])
            elif f[0] == "biasdma":
                self.bias_dma(bias_fetch_slot, bias_fetch_tag, bias_skip_last)
            else:
                self.emit_op(f, lt, lt_next)

        def emit_m(m):
            if m[0] == "mfma":
                self.mfma(m[1], m[2], m[3], m[4], first=m[5])
            else:
                self.mfma(m[1], "sel%d" % m[2], "xf%d" % m[3], ("skip", lt_next), first=True)

        # the next layer's bias row into a[A_BIAS ..] (8 loads instead of one per tile: the tiles' first MFMAs take it as C) and
        # phase A: the drain of pair 0 - what the layer's last K-step did not take under its MFMAs
        for f in head:
            emit_f(f)
        dma3 = [("dma", lbase_next + 3, j) for j in range(4)]       # the next layer's stage 3: under the first MFMAs
        for J in range(4):
            if fine is not None and J > 0:
                self.stamp(fine + J, absolute=True)
            groups = segs[J]
            # the segment's matrix items: [selector MFMAs of pair J,] then its groups; the loads of a group are emitted
            # behind the MFMAs of the item in front of it (the first item's: in front of the segment)
            items = []
            if rx:
                items.append(("sel", [("sel", TILE(2 * J + o, n), o, n) for n in range(6) for o in range(2)], []))
            for g in groups:
                items.append((g, mfmas_of[g], loads_of[g]))
            for f in items[0][2]:
                emit_f(f)
            if J < 3:
                low = d1 if J == 0 else self.drain_ops(J + 1, rx, bias_slot, False)    # low-priority fillers: the next pair's drain
                if J < 2:                                                  # ... and two pieces of the stage-3 DMA
                    h = len(low) // 3
                    low = low[:h] + [dma3.pop(0)] + low[h:2 * h] + [dma3.pop(0)] + low[2 * h:]
                n_m = sum(len(it[1]) for it in items)
                n_la = sum(len(it[2]) for it in items[1:])
                per_gap = -(-(len(low) + n_la) // n_m)
                for k, (name, mf, _) in enumerate(items):
                    la = list(items[k + 1][2]) if k + 1 < len(items) else []
                    for m in mf:
                        emit_m(m)
                        for _ in range(per_gap):
                            if la:
                                emit_f(la.pop(0))
                            elif low:
                                emit_f(low.pop(0))
                    for f in la:
                        emit_f(f)
                for f in low:
                    emit_f(f)
            else:
                # last segment, one filler per MFMA gap: look-ahead loads first; behind B_X (placed behind group (3, 1): every
                # wave has left the next layer's stage 0 = input groups 0 and 1, and its stage 2 has landed) the DMA of stage 4
                # and the bias fetch, then the prefetch of (tap 1, ks 0) into buffer 0 = slots 0..13: these hold the activation
                # fragments of input groups 0, 1 (free behind group (3, 1)) and two of group 2 (free behind (3, 2), the group
                # right behind B_X: the two loads into slots 12, 13 come last)
                low = []
                for k, (name, mf, _) in enumerate(items):
                    la = list(items[k + 1][2]) if k + 1 < len(items) else []
                    for m in mf:
                        emit_m(m)
                        if la:
                            emit_f(la.pop(0))
                        elif low:
                            emit_f(low.pop(0))
                    for f in la:
                        emit_f(f)
                    if name == (3, 1):
                        self.barrier(lbase_next + 2, "BX")
                        low = [("dma", lbase_next + 4, 0), ("dma", lbase_next + 4, 1), ("biasdma",), ("dma", lbase_next + 4, 2),
                               ("dma", lbase_next + 4, 3)] + [("pfa", m) for m in range(8)] + [("pfb", n) for n in range(6)]
                assert not low, low
        self.stamp(stamp_k)

    # ---- the transition between two layers: last tap of layer lt, its epilogue, first tap of layer lt_next ------------------------
    def transition(self, lt, lt_next, lbase, lbase_next, rx, bias_slot, bias_fetch_slot, bias_fetch_tag, bias_skip_last, stamp_k,
                   final=False, lead=20, fine=None):
        """Tap 8 of layer lt is issued in two halves by output channel: h1 = weight tiles 0..3 (pairs 0, 1) over ks 0..3, then
        h2 = weight tiles 4..7 - every accumulator still sees (tap 8: ks 0, 1, 2, 3).  Pairs 0 and 1 are complete behind h1, so
        their drains D(0), D(1) issue in the gaps of h2's 96 MFMAs; D(2) and D(3) issue under the first MFMA groups of the next
        layer's tap 0 (group (P, K) = output pair P x input group K, ready once pairs P and K are drained).  The drains' ~430
        VALU and ~60 LDS instructions then sit beside ~200 MFMAs instead of none.
        A list scheduler places everything: the MFMA sequence is fixed; each fragment load is emitted `lead` MFMAs ahead of its
        first use, not before the data exists (a barrier published the stage / the drain stored the rows), into a free slot;
        fillers are spread evenly over the MFMA range they are anchored to."""
        self.comment("---- transition: tap 8 of layer %s, its epilogue, tap 0 of layer %s" % (lt, lt_next))
        M = []                                   # matrix sequence: ("mfma", tile, Aid, Bid, want, first) | ("sel", tile, o, n) | ("bar", name, landed)
        tap = 8
        def A8(ks, mt): return ("A", lt, tap, ks, mt)
        def B8(ks, n, h): return ("B", lt, tap, ks, n, h)
        def An(K, mt): return ("A", lt_next, 0, K, mt)
        def Bn(K, n, seg): return ("B", lt_next, 0, K, n, seg)
        marks = {}
        for h, mts in ((1, range(0, 4)), (2, range(4, 8))):
            for ks in range(4):
                for n in range(6):
                    for mt in mts:
                        M.append(("mfma", TILE(mt, n), A8(ks, mt), B8(ks, n, h), (lt, tap, ks), False))
                    if h == 1 and ks == 1 and n == 0:
                        M.append(("bar", "B16", lbase + 17))           # stage 17 has landed; every wave has left stage 15
                if h == 2 and ks == 1:
                    M.append(("bar", "BH", None if final else lbase_next))      # every wave has left stage 16 (its slot: the next layer's stage 2)
                    marks["h2mid"] = len([m for m in M if m[0] not in ("bar", "stamp")])
                    if rx:
                        M += [("sel", TILE(o, n), o, n) for n in range(6) for o in range(2)]     # pair 0: x + bias (D(0) is done)
        n_h = len([m for m in M if m[0] not in ("bar", "stamp")])
        if fine is not None:
            # fine stamps (absolute slots fine .. fine + 4): behind h1, h2, the first, the second group segment (the section's own
            # stamp follows the third)
            k = [i for i, m in enumerate(M) if m[0] == "mfma"][95]
            M.insert(k + 1, ("stamp", fine))
            M.append(("stamp", fine + 1))
        segs = [[(0, 0), (1, 0), (0, 1), (1, 1)], [(2, 0), (2, 1), (0, 2), (1, 2), (2, 2)], [(3, 0), (3, 1), (3, 2), (0, 3), (1, 3), (2, 3), (3, 3)]]
        seg_start = []
        if not final:
            M.append(("bar", "BE", lbase_next + 1))                     # every wave has left stage 17; the next layer's stages 0, 1 have landed
            for si, sg in enumerate(segs):
                if fine is not None and si > 0:
                    M.append(("stamp", fine + 1 + si))
                seg_start.append(len([m for m in M if m[0] not in ("bar", "stamp")]))
                if rx:
                    J = si + 1
                    M += [("sel", TILE(2 * J + o, n), o, n) for n in range(6) for o in range(2)]
                for (P, K) in sg:
                    M += [("mfma", TILE(2 * P + o, n), An(K, 2 * P + o), Bn(K, n, si), (lt_next, 0, K), (not rx) and K == 0)
                          for n in range(6) for o in range(2)]
                    if (P, K) == (3, 1):
                        M.append(("bar", "BX", lbase_next + 2))         # every wave has left the next layer's stage 0; its stage 2 has landed
                        marks["bx"] = len([m for m in M if m[0] not in ("bar", "stamp")])
        nm = len([m for m in M if m[0] not in ("bar", "stamp")])
        # gap index of every matrix item; barrier positions
        gap_of_bar = {}
        g = 0
        for m in M:
            if m[0] == "bar":
                gap_of_bar[m[1]] = g
            elif m[0] != "stamp":
                g += 1
        # ---- fillers anchored to gap ranges [s, e)
        fill_at = {}
        def spread(ops, s0, e0):
            n = len(ops)
            for k, op in enumerate(ops):
                fill_at.setdefault(s0 + (k * (e0 - s0)) // max(n, 1), []).append(op)
            return e0
        h2s = 96
        mid = marks["h2mid"]                                             # first gap behind h2's ks 1
        d_end = {}
        spread([("biasrow", mt, bias_slot) for mt in range(8)] if not final else [], 0, 8)
        if rx:
            d_end[0] = spread(self.drain_ops(0, rx, bias_slot, final), h2s, mid - 2)
            d_end[1] = spread(self.drain_ops(1, rx, bias_slot, final), mid + 12, n_h)
        else:
            d_end[0] = spread(self.drain_ops(0, rx, bias_slot, final), h2s, mid)
            d_end[1] = spread(self.drain_ops(1, rx, bias_slot, final), mid, n_h)
        if not final:
            sel = 12 if rx else 0
            d_end[2] = spread(self.drain_ops(2, rx, bias_slot, final), seg_start[0] + sel, seg_start[1])
            d_end[3] = spread(self.drain_ops(3, rx, bias_slot, final), seg_start[1] + sel, seg_start[2])
            # weight DMA: stage 1' behind B16 (slot of stage 15), 2' behind BH (slot of 16), 3' behind BE (slot of 17), 4' behind BX
            spread([("dma", lbase + 19, j) for j in range(4)], gap_of_bar["B16"] + 2, gap_of_bar["B16"] + 40)
            spread([("dma", lbase + 20, j) for j in range(4)], gap_of_bar["BH"] + 2, gap_of_bar["BH"] + 40)
            spread([("dma", lbase + 21, j) for j in range(4)], gap_of_bar["BE"] + 2, gap_of_bar["BE"] + 40)
            spread([("dma", lbase + 22, j) for j in range(4)] + [("biasdma",)], gap_of_bar["BX"] + 1, gap_of_bar["BX"] + 30)
        # ---- fragments: first / last use, earliest load position
        first_use, last_use = {}, {}
        g = 0
        for m in M:
            if m[0] == "mfma":
                for fid in (m[2], m[3]):
                    first_use.setdefault(fid, g)
                    last_use[fid] = g
            if m[0] not in ("bar", "stamp"):
                g += 1
        earliest = {}
        for fid in first_use:
            e0 = 0
            if fid[0] == "A" and fid[1] == lt and fid[3] >= 2:
                e0 = gap_of_bar["B16"]                                   # stage 17
            if fid[1] == lt_next and not final:
                e0 = gap_of_bar["BE"]
                if fid[0] == "B":
                    e0 = max(e0, d_end[fid[3]])                          # the rows of input group K are stored
            earliest[fid] = e0
        # what tap 7's last K-step prefetched: (tap 8, ks 0) into slots 0..13
        slot_of = {}
        free = list(range(30))
        for mt in range(4):
            slot_of[A8(0, mt)] = mt
        for n in range(6):
            slot_of[B8(0, n, 1)] = 8 + n
        for sl in list(range(4)) + list(range(8, 14)):
            free.remove(sl)
        load_at = {}
        for fid in first_use:
            if fid in slot_of:
                continue
            load_at.setdefault(max(earliest[fid], first_use[fid] - lead), []).append(fid)
        # the next tap's first K-step into buffer 0 (slots 0..13), behind BX
        pf = []
        if not final:
            pf = [("pfa", m) for m in range(8)] + [("pfb", n) for n in range(6)]
            spread(pf, gap_of_bar["BX"] + 31, nm - 6)

        def emit_load(fid):
            # the slot freed longest ago; fragments still alive behind B_X stay out of slots 0..13, which must be free for the
            # next tap's prefetch at the end
            cand = [x for x in free if x >= 14] if (not final and last_use[fid] >= gap_of_bar["BX"] + 31) else free
            assert cand, ("out of fragment slots", fid, g)
            sl = cand[0]
            free.remove(sl)
            slot_of[fid] = sl
            if fid[0] == "A":
                _, L, tp, ks, mt = fid
                base = lbase if L == lt else lbase_next
                self.load_a(sl, base + 2 * tp + (ks >> 1), ks & 1, mt, (L, tp, ks, mt))
            else:
                _, L, tp, ks, n, _h = fid
                self.load_b(sl, tp, ks, n, (L, tp, ks, n))

        def emit_filler(f):
            if f[0] == "dma":
                self.dma_piece(f[1], f[2])
            elif f[0] == "biasdma":
                self.bias_dma(bias_fetch_slot, bias_fetch_tag, bias_skip_last)
            elif f[0] == "pfa":
                assert f[1] in free, ("prefetch slot busy", f, sorted(free))
                free.remove(f[1])
                self.load_a(f[1], lbase_next + 2, 0, f[1], (lt_next, 1, 0, f[1]))
            elif f[0] == "pfb":
                assert 8 + f[1] in free, ("prefetch slot busy", f, sorted(free))
                free.remove(8 + f[1])
                self.load_b(8 + f[1], 1, 0, f[1], (lt_next, 1, 0, f[1]))
            else:
                self.emit_op(f, lt, lt_next)

        g = 0
        for m in M:
            if m[0] == "bar":
                self.barrier(m[2], m[1])
                continue
            if m[0] == "stamp":
                self.stamp(m[1], absolute=True)
                continue
            for fid in load_at.get(g, []):
                emit_load(fid)
            if m[0] == "mfma":
                assert m[2] in slot_of and m[3] in slot_of, ("fragment not loaded", m, g)
                self.mfma(m[1], slot_of[m[2]], slot_of[m[3]], m[4], first=m[5])
                for fid in (m[2], m[3]):
                    if last_use[fid] == g:
                        free.append(slot_of[fid])
            else:
                self.mfma(m[1], "sel%d" % m[2], "xf%d" % m[3], ("skip", lt_next), first=True)
            for f in fill_at.get(g, []):
                emit_filler(f)
            g += 1
        for gg in sorted(k for k in fill_at if k >= g):
            for f in fill_at[gg]:
                emit_filler(f)
        if final:
            for J in (2, 3):
                for op in self.drain_ops(J, False, 0, True):
                    self.emit_op(op, lt, None)
            self.barrier(None, "BF")
            self.add("s_waitcnt lgkmcnt(0)", "lgkm0")
        self.stamp(stamp_k)

    def final_drain(self, lt, stamp_k, lead_in=None):
        self.comment("---- epilogue of the last layer")
        if lead_in is not None:
            self.last_stage(lt, lead_in, False, [])
        self.barrier(None, "BF")
        for J in range(4):
            for op in self.drain_ops(J, False, 0, True):
                self.emit_op(op, lt, None)
        self.add("s_waitcnt lgkmcnt(0)", "lgkm0")
        self.stamp(stamp_k)


# =============================================================================================================================
# program = sections; text emission; symbolic check
# =============================================================================================================================
def sec_head(stamps):
    """register set-up: the lane table, the weight stream's buffer resource, the per-wave piece offsets.
    asm operands: %0 s[2] lane table (this lane's row is at tid * TAB_DWORDS * 4), %1 s[2] weight stream, %2 s bytes of it,
    %3 s[2] address of bias row 2 (tower layer 1), %4 s blocks, %5 s wave, %6 s[2] stamp slot 3 of this workgroup, %7 v tid"""
    e = Emitter(stamps)
    e.comment("==== k_tower1wa layer body (generated by tools/gen_tower1wa.py - do not edit) ====")
    e.add("s_mov_b32 s%d, %%4" % S_CNT)
    e.add("s_mov_b32 s%d, %%5" % S_WAVE)
    e.add("s_mov_b64 s[%d:%d], %%3" % (S_BIASP, S_BIASP + 1))
    e.add("s_mov_b64 s[%d:%d], %%6" % (S_STAMP, S_STAMP + 1))
    e.add("s_sub_u32 s%d, s%d, 24" % (S_STAMP0, S_STAMP))
    e.add("s_subb_u32 s%d, s%d, 0" % (S_STAMP0 + 1, S_STAMP + 1))
    e.add("s_mov_b64 s[%d:%d], %%1" % (S_RSRC, S_RSRC + 1))
    e.add("s_and_b32 s%d, s%d, 0xffff" % (S_RSRC + 1, S_RSRC + 1))
    e.add("s_mov_b32 s%d, %%2" % (S_RSRC + 2))
    e.add("s_mov_b32 s%d, 0x00020000" % (S_RSRC + 3))
    e.add("v_mul_u32_u24 v%d, %d, %%7" % (V_XT, TAB_DWORDS * 4), "valu", wr=[V_XT])
    for i in range(TAB_DWORDS // 4):
        e.add("global_load_dwordx4 v[%d:%d], v%d, %%0 offset:%d" % (V_TAB0 + 4 * i, V_TAB0 + 4 * i + 3, V_XT, 16 * i), "gload")
    # piece j of this wave: P = 4 wave + j; source offset inside a stage = ((P>>3) 64 + ((P>>2)&1) 32 + (P&1) 16 + ((P>>1)&1) 4) 256,
    # LDS offset inside a ring slot = P 1024
    e.add("s_mov_b32 s%d, 0" % S_BLK)
    for j in range(4):
        # wave 0..3: P >> 3 = wave >> 1, (P >> 2) & 1 = wave & 1, P & 1 = j & 1, (P >> 1) & 1 = j >> 1
        e.add("s_lshr_b32 s%d, s%d, 1" % (S_T, S_WAVE))
        e.add("s_lshl_b32 s%d, s%d, 14" % (S_T, S_T))                         # (wave >> 1) * 64 * 256
        e.add("s_and_b32 s%d, s%d, 1" % (S_T + 1, S_WAVE))
        e.add("s_lshl_b32 s%d, s%d, 13" % (S_T + 1, S_T + 1))                 # (wave & 1) * 32 * 256
        e.add("s_add_u32 s%d, s%d, s%d" % (S_T, S_T, S_T + 1))
        e.add("s_add_u32 s%d, s%d, 0x%x" % (S_POFF + j, S_T, ((j & 1) * 16 + (j >> 1) * 4) * 256))
        e.add("s_lshl_b32 s%d, s%d, 12" % (S_T, S_WAVE))
        e.add("s_add_u32 s%d, s%d, 0x%x" % (S_LDST + j, S_T, j * 1024))
    e.add("s_waitcnt vmcnt(0)", "vm0")
    # piece j's source offset: the lane's (even / odd piece) + the piece's place in a stage; pieces 2, 3 first (they read
    # V_WSRC[0 / 1], which pieces 0, 1 then overwrite in place)
    for j in (2, 3, 0, 1):
        e.add("v_add_u32 v%d, s%d, v%d" % (V_WSRC4[j], S_POFF + j, V_WSRC + (j & 1)), "valu", wr=[V_WSRC4[j]])
    # (clamp) the last stage of the tower: local stage 35 of the last block = blocks * BLOCK_BYTES - 32768 + 128
    e.add("s_mul_i32 s%d, s%d, 0x%x" % (S_MAXB, S_CNT, BLOCK_BYTES))
    e.add("s_sub_u32 s%d, s%d, 0x%x" % (S_MAXB, S_MAXB, 32768 - 128))
    return e


def sec_pro(stamps):
    """Round 5: the INPUT convolution's epilogue (HIP code until round 4: 4.8 k cycles of its own, followed by an unskewed tap 0
    of layer 0 with cold fragment loads, 7.5 k) is the same drain as every odd layer's - accumulators -> bf16 -> ReLU -> LDS
    rows, the next layer's bias row as the C operand of its tiles' first MFMAs - so it runs under tap 0 of layer 0 like every
    other boundary (Emitter.skew with lt = -1, no lead-in: the input convolution has no weight stage of this ring to run its
    head under).  The HIP code in front of the statement now ends behind the input convolution's MFMAs: the accumulators hold
    conv + bias[0], the weight DMA of stages 0..2 is in flight (the head's vmcnt(0) covers this wave's pieces, the skew's
    first barrier the other waves'), bias row 1 (layer 0) sits in bias slot 1."""
    e = Emitter(stamps)
    e.skew(-1, 0, 0, False, 1, 0, 1, False, 3)
    return e


# Two schedules of the layer boundary were built and measured:
#   "skew"        taps 1..8 regular, then the epilogue under tap 0 of the next layer (Emitter.skew)            <- the product
#   "transition"  taps 1..7 regular, then tap 8 in two halves by output channel with the drains of pairs 0, 1 under the
#                 second half, drains 2, 3 under the next layer's first groups (Emitter.transition): more fragment loads,
#                 same wall time
SCHEDULE = os.environ.get("XQ_1WA_SCHEDULE", "skew")


def sec_even(stamps, blk):
    e = Emitter(stamps)
    L = 2 * blk
    e.add("s_waitcnt lgkmcnt(0)", "lgkm0")                 # the loop head: one known state of the LDS queue for both ways in
    if SCHEDULE == "skew":
        e.comment("---- first convolution of the block: taps 1..8 (the last weight stage opens the epilogue)")
        for tap in range(1, 9):
            e.tap_regular(L, 0, tap, False, tap < 8, ks_stop=2 if tap == 8 else 4)
        e.stamp(0)
        # its epilogue (with the block input x for the skip connection) under tap 0 of the second convolution; the bias row of
        # the next block's first convolution is fetched here (none behind the last block)
        e.skew(L, L + 1, 18, True, 0, 1, L + 2, True, 1, lead_in=0, fine=40 if stamps else None)
    else:
        e.comment("---- first convolution of the block: taps 1..7")
        for tap in range(1, 8):
            e.tap_regular(L, 0, tap, False, True)
        e.stamp(0)
        e.transition(L, L + 1, 0, 18, True, 0, 1, L + 2, True, 1, fine=40 if stamps else None)
    return e


def sec_odd(stamps, blk):
    e = Emitter(stamps)
    L = 2 * blk + 1
    if SCHEDULE == "skew":
        e.comment("---- second convolution of the block: taps 1..8 (the last weight stage opens the epilogue)")
        for tap in range(1, 9):
            e.tap_regular(L, 18, tap, False, tap < 8, clamp_from=36, ks_stop=2 if tap == 8 else 4)
    else:
        e.comment("---- second convolution of the block: taps 1..7")
        for tap in range(1, 8):
            e.tap_regular(L, 18, tap, False, True, clamp_from=36)
    e.stamp(2)
    return e


def sec_x2(stamps, blk):
    e = Emitter(stamps)
    L = 2 * blk + 1
    if SCHEDULE == "skew":
        e.skew(L, L + 1, 36, False, 1, 0, L + 2, False, 3, lead_in=18, fine=45 if stamps else None)
    else:
        e.transition(L, L + 1, 18, 36, False, 1, 0, L + 2, False, 3, fine=45 if stamps else None)
    return e


def sec_fin(stamps, blk):
    e = Emitter(stamps)
    L = 2 * blk + 1
    if SCHEDULE == "skew":
        e.final_drain(L, 3, lead_in=18)
    else:
        e.transition(L, None, 18, None, False, 1, 0, None, False, 3, final=True)
    return e


LGKM_WINDOW = 8       # an s_waitcnt costs an issue slot even when it does not stall: one wait covers the MFMAs of this window


def insert_lgkm_waits(linear):
    """exact s_waitcnt lgkmcnt(n) in front of every MFMA that consumes the result of an LDS read still in flight.
    LDS operations of a wave complete in order; n = operations issued behind the one waited for.  A wait also covers the
    operands of the next LGKM_WINDOW MFMAs as far as their loads have been issued already (they were issued about as long
    ago as this one's: waiting for them here costs nothing and saves the later waits).  Returns the new list; 'lgkm0'
    instructions wait for everything."""
    out = []
    nl = 0                    # LDS / SMEM operations issued
    retired = 0               # operations known complete: the first `retired` ones
    pend = {}                 # register key -> index of the read that will write it

    def keys_of(ins):
        return [("f", ins.m["a"]), ("f", ins.m["b"]), ("bias", ins.m["tile"] // 6) if ins.m.get("first") else ("acc", ins.m["tile"])]

    for pos, ins in enumerate(linear):
        k = ins.kind
        if k == "mfma":
            mine = max([pend[x] for x in keys_of(ins) if x in pend and pend[x] >= retired], default=None)
            if mine is not None:
                want = mine
                seen = 0
                for nxt in linear[pos + 1:]:
                    if nxt.kind in ("barrier", "label", "lgkm0", "blk"):
                        break
                    if nxt.kind == "mfma":
                        seen += 1
                        if seen > LGKM_WINDOW:
                            break
                        # operands whose loads are already in the queue (a load emitted between here and there re-defines the
                        # register: then the register's pending index changes and this look-ahead must not count the old one)
                        for x in keys_of(nxt):
                            if x in pend and retired <= pend[x] < nl and not redefined(linear, pos, nxt, x):
                                want = max(want, pend[x])
                n = min(nl - 1 - want, 15)
                out.append(Ins("s_waitcnt lgkmcnt(%d)" % n, "lgkmwait", n=n))
                retired = max(retired, nl - n)
        elif k == "lgkm0":
            retired = nl
        elif k == "smem":
            nl += 1
        if k == "ldsr":
            d = ins.m["dst"]
            key = ("acc", int(d[3:])) if isinstance(d, str) and d.startswith("acc") else (
                ("bias", int(d[4:])) if isinstance(d, str) and d.startswith("bias") else ("f", d))
            pend[key] = nl
            nl += 1
        elif k == "ldsw":
            nl += 1
        out.append(ins)
    return out


def redefined(linear, pos, upto, key):
    """is register `key` loaded again between linear[pos] and the instruction object `upto`?"""
    for x in linear[pos + 1:]:
        if x is upto:
            return False
        if x.kind == "ldsr":
            d = x.m["dst"]
            k2 = ("acc", int(d[3:])) if isinstance(d, str) and d.startswith("acc") else (
                ("bias", int(d[4:])) if isinstance(d, str) and d.startswith("bias") else ("f", d))
            if k2 == key:
                return True
    return False


class CheckError(Exception):
    pass


def check_and_fill(linear, nblocks, verify=True):
    """symbolic run of the linearized program (one wave's view; every wave runs the same stream, wave 1 also fetches the
    bias rows).  Fills in the vmcnt of every stage barrier.  Raises CheckError on the first violation."""
    def fail(i, msg):
        ctx = "\n".join("   %s%s" % (">> " if j == i else "   ", linear[j].text) for j in range(max(0, i - 6), min(len(linear), i + 3)))
        raise CheckError("instruction %d: %s\n%s" % (i, msg, ctx))

    if not verify:
        # ablation bodies: only the vmcnt of the barriers is filled in (from the unablated DMA bookkeeping)
        vq = []
        blk = 0
        for ins in linear:
            if ins.kind == "blk":
                blk = ins.m["blk"]
            elif ins.kind == "dma":
                vq.append(36 * blk + ins.m["lstage"])
            elif ins.kind == "gstore":
                vq.append(-1)
            elif ins.kind == "vmwait":
                X = ins.m["landed"]
                idx = -1 if X is not None else len(vq) - 1
                if X is not None:
                    idx = max([j for j, x in enumerate(vq) if 0 <= x <= 36 * blk + X], default=-1)
                ins.m["n"] = min(len(vq) - 1 - idx, 63)
                ins.text = "s_waitcnt vmcnt(%d)" % ins.m["n"]
        return True

    nlayers = 2 * nblocks
    # LDS activation rows: content[J][n] = layer whose output is stored there (-1 = the input convolution's, -2 = nothing yet:
    # the statement starts behind the input convolution's MFMAs, its epilogue is the first drain)
    act = [[-2] * 6 for _ in range(4)]
    # ring: slot -> (gstage, landed_for_all).  Stages 0..2 were issued by the HIP code (4 pieces per wave each); the head's
    # vmcnt(0) covers this wave's pieces, the first barrier publishes everybody's
    ring = {0: [0, False], 1: [1, False], 2: [2, False], 3: [None, True]}
    ring_reads = {}            # gstage -> list of lds op indices of reads issued
    bias_slot = {0: [None, True, []], 1: [0, True, []]}      # slot -> [layer whose bias, landed for all, read op indices]
    frag = {}                  # slot / 'xf<n>' -> (tag, lds op index)
    acc = {t: {"hist": [("BIAS", -1), ("INPUT",)], "read": set(), "last_mfma": -10 ** 9, "pending": None} for t in range(48)}
    biasreg = {}               # weight tile mt -> (layer whose bias row a[A_BIAS + 4 mt ..] holds, lds op index)
    rd = {}                    # vgpr -> (tile, comp, layer)
    pk = {}                    # vgpr -> ("cvt" | "relu", tile, comps, layer)
    nl, retired = 0, 0         # LDS queue
    vq = [("w", gs, j, -1) for gs in range(3) for j in range(4)]    # vm queue entries: ("w", gstage, piece) | ("bias", slot) | ("st",)
    vdone = 0                  # entries known complete (prefix), this wave's view WITHOUT the bias entries
    vq_nb = []                 # the same queue without bias entries (what waves 0, 2, 3 see)
    vdone_nb = 12              # (the head's vmcnt(0))
    vdone_w1 = 12              # wave 1's view (with its bias entries)
    last_barrier = -1
    stores_since = {}          # bias slot read bookkeeping
    blk = 0
    expected = {-1: [("BIAS", -1), ("INPUT",)]}
    for L in range(nlayers):
        h = [("BIAS", L)] + ([("skip", L)] if L & 1 else []) + [(L, tap, ks) for tap in range(9) for ks in range(4)]
        expected[L] = h

    def lds_retire_to(idx):
        nonlocal retired
        retired = max(retired, idx + 1)

    for i, ins in enumerate(linear):
        k, m = ins.kind, ins.m
        if k == "blk":
            blk = m["blk"]
            continue
        g0 = 36 * blk
        if k == "lgkmwait":
            retired = max(retired, nl - m["n"])
        elif k == "lgkm0":
            retired = nl
        elif k == "smem":
            nl += 1
        elif k == "ldsr":
            src, dst, want = m["src"], m["dst"], m["want"]
            if src[0] == "ring":
                _, lst, kk, mt = src
                gs = g0 + lst
                sl = gs & 3
                if ring[sl][0] != gs:
                    fail(i, "weight read of stage %d from slot %d which holds stage %s" % (gs, sl, ring[sl][0]))
                if not ring[sl][1]:
                    fail(i, "weight read of stage %d before a barrier published it" % gs)
                L, tap, ks, mt2 = want
                if (gs, kk, mt) != (18 * L + 2 * tap + (ks >> 1), ks & 1, mt2):
                    fail(i, "weight fragment address (stage %d, kk %d, mt %d) is not %s" % (gs, kk, mt, (want,)))
                ring_reads.setdefault(gs, []).append(nl)
                frag[dst] = (("W",) + want, nl)
            elif src[0] == "act":
                _, ks, tap, n = src
                L, tap2, ks2, n2 = want
                if (ks, tap, n) != (ks2, tap2, n2):
                    fail(i, "activation fragment address mismatch")
                if any(act[ks][nn] != L - 1 for nn in range(6)):
                    fail(i, "activation read of group %d for layer %d: rows hold %s" % (ks, L, act[ks]))
                frag[dst] = (("X",) + want, nl)
            elif src[0] == "actrow":
                _, J, n = src
                _, L, J2, n2 = want
                if act[J][n] != L - 1:
                    fail(i, "x read of (pair %d, tile %d) for layer %d: row holds %d" % (J, n, L, act[J][n]))
                frag[dst] = (want, nl)
            elif src[0] == "bias":
                _, bs = src
                _, Lb, mt = want
                if bias_slot[bs][0] != Lb or not bias_slot[bs][1]:
                    fail(i, "bias read of layer %d from slot %d which holds %s (landed %s)" % (Lb, bs, bias_slot[bs][0], bias_slot[bs][1]))
                # the row it replaces must not be needed any more: every tile of weight tile mt has had its first MFMA
                old = biasreg.get(mt)
                if old is not None and any(acc[TILE(mt, n)]["hist"][0] != ("BIAS", old[0]) for n in range(6)):
                    fail(i, "bias row of weight tile %d (layer %d) replaced before every tile took it" % (mt, old[0]))
                biasreg[mt] = (Lb, nl)
                bias_slot[bs][2].append(nl)
            nl += 1
        elif k == "ldsw":
            L, J, n = m["store"]
            # data: four packed registers of the right tiles
            for kk in range(4):
                want = ("relu", TILE(2 * J + (kk >> 1), n), (2 * (kk & 1), 2 * (kk & 1) + 1), L)
                if pk.get(m["data"] + kk) != want:
                    fail(i, "store data v%d is %s, not %s" % (m["data"] + kk, pk.get(m["data"] + kk), want))
            if act[J][n] != L - 1:
                fail(i, "store of layer %d over rows that hold %d" % (L, act[J][n]))
            act[J][n] = L
            nl += 1
        elif k == "mfma":
            tile, want = m["tile"], m["want"]
            a = acc[tile]
            if m.get("first"):
                Ln = want[1] if want[0] == "skip" else want[0]
                br = biasreg.get(tile // 6)
                if br is None or br[0] != Ln:
                    fail(i, "first MFMA of tile %d for layer %d: bias registers hold %s" % (tile, Ln, br))
                if br[1] >= retired:
                    fail(i, "first MFMA of tile %d: bias row not waited for" % tile)
                if a["read"] != {0, 1, 2, 3}:
                    fail(i, "first MFMA of tile %d overwrites an accumulator that was not drained (%s)" % (tile, a["read"]))
                a["hist"], a["read"] = [("BIAS", Ln)], set()
            elif a["read"]:
                fail(i, "MFMA accumulates onto tile %d after it was (partly) read out" % tile)
            if want[0] == "skip":
                L = want[1]
                if frag.get(m["b"], (None,))[0] != ("XIN", L - 1, tile // 12, tile % 6):
                    fail(i, "selector MFMA on tile %d reads %s" % (tile, frag.get(m["b"])))
                if frag[m["b"]][1] >= retired:
                    fail(i, "selector MFMA operand not waited for")
                if m["a"] != "sel%d" % ((tile // 6) & 1):
                    fail(i, "wrong selector")
                a["hist"].append(("skip", L))
            else:
                L, tap, ks = want
                fa, fb = frag.get(m["a"]), frag.get(m["b"])
                if fa is None or fa[0] != ("W", L, tap, ks, tile // 6):
                    fail(i, "MFMA (layer %d tap %d ks %d tile %d) reads weight fragment %s" % (L, tap, ks, tile, fa))
                if fb is None or fb[0] != ("X", L, tap, ks, tile % 6):
                    fail(i, "MFMA (layer %d tap %d ks %d tile %d) reads activation fragment %s" % (L, tap, ks, tile, fb))
                if fa[1] >= retired or fb[1] >= retired:
                    fail(i, "MFMA operand not waited for (queue: %d issued, %d retired; operands %d, %d)" % (nl, retired, fa[1], fb[1]))
                a["hist"].append(want)
            if a["hist"] != expected[a["hist"][0][1]][:len(a["hist"])]:
                fail(i, "tile %d accumulates out of order: ... %s" % (tile, a["hist"][-3:]))
            if i - a["last_mfma"] < 8 and a["last_mfma"] > 0:
                fail(i, "two MFMAs on tile %d only %d instructions apart" % (tile, i - a["last_mfma"]))
            a["last_mfma"] = i
        elif k == "valu":
            if "accread" in m:
                tile, comp = m["accread"]
                L = m["layer"]
                a = acc[tile]
                if a["hist"] != expected[L]:
                    fail(i, "tile %d read for layer %d with %d of %d terms" % (tile, L, len(a["hist"]), len(expected[L])))
                if i - a["last_mfma"] < 24:
                    fail(i, "accumulator of tile %d read %d instructions behind its last MFMA" % (tile, i - a["last_mfma"]))
                a["read"].add(comp)
                rd[m["wr"][0]] = (tile, comp, L)
            elif "cvt" in m:
                r0, r1 = rd.get(m["rd"][0]), rd.get(m["rd"][1])
                if not r0 or not r1 or r0[0] != r1[0] or r0[2] != r1[2] or (r0[1], r1[1]) not in ((0, 1), (2, 3)):
                    fail(i, "conversion of %s, %s" % (r0, r1))
                pk[m["wr"][0]] = ("cvt", r0[0], (r0[1], r1[1]), r0[2])
            elif "relu" in m:
                p0 = pk.get(m["rd"][0])
                if not p0 or p0[0] != "cvt":
                    fail(i, "ReLU of %s" % (p0,))
                pk[m["wr"][0]] = ("relu",) + p0[1:]
            for r in m.get("wr", []):
                if "accread" not in m and "cvt" not in m and "relu" not in m:
                    rd.pop(r, None)
                    pk.pop(r, None)
        elif k == "dma":
            gs = g0 + m["lstage"]
            if gs >= 18 * nlayers:
                gs_eff = 18 * nlayers - 1          # clamped refetch of the last stage (never read)
            sl = gs & 3
            old = ring[sl][0]
            if old is not None and old != gs:
                if gs < 18 * nlayers or True:
                    # every read of the old stage must have been retired before the last barrier
                    for r in ring_reads.get(old, []):
                        if r > ring.get(("bar", "retired"), -1):
                            fail(i, "slot %d refilled with stage %d while a read of stage %d (lds op %d) was not complete at the last barrier" % (sl, gs, old, r))
                    if old + 4 != gs:
                        fail(i, "slot %d: stage %d replaces stage %d" % (sl, gs, old))
            if old != gs:
                ring[sl] = [gs, False]
            vq.append(("w", gs, m["piece"], i))
        elif k == "dma_bias":
            bs = m["slot"]
            for r in bias_slot[bs][2]:
                if r > ring.get(("bar", "retired"), -1):
                    fail(i, "bias slot %d refetched while a read (lds op %d) was not complete at the last barrier" % (bs, r))
            bias_slot[bs] = [m["lrow"], False, []]
            vq.append(("bias", bs, 0, i))
        elif k == "gstore":
            vq.append(("st", 0, 0, i))
        elif k == "vm0":
            vdone_nb = len([x for x in vq if x[0] != "bias"])
            vdone_w1 = len(vq)
        elif k == "vmwait":
            X = m["landed"]
            nb = [x for x in vq if x[0] != "bias"]
            if X is None:
                n = 0                                   # the last barrier: everything
                idx = len(nb) - 1
            else:
                gX = g0 + X
                idx = max([j for j, x in enumerate(nb) if x[0] == "w" and x[1] <= gX], default=-1)
                n = len(nb) - 1 - idx
            if n > 63:
                fail(i, "vmcnt %d" % n)
            ins.text = "s_waitcnt vmcnt(%d)" % n
            m["n"] = n
            vdone_nb = max(vdone_nb, len(nb) - n)
            vdone_w1 = max(vdone_w1, len(vq) - n)
        elif k == "barrier":
            # what this wave has waited for, every wave has: published
            nb = [x for x in vq if x[0] != "bias"]
            for x in nb[:vdone_nb]:
                if x[0] == "w":
                    sl = x[1] & 3
                    if ring[sl][0] == x[1]:
                        # landed once all 4 pieces of the stage are within the waited prefix
                        pieces = [y for y in nb[:vdone_nb] if y[0] == "w" and y[1] == x[1]]
                        if len(pieces) >= 4:
                            ring[sl][1] = True
            for x in vq[:vdone_w1]:
                if x[0] == "bias" and bias_slot[x[1]][0] is not None:
                    bias_slot[x[1]][1] = True
            ring[("bar", "retired")] = retired - 1      # LDS ops with index <= this were complete when the wave arrived
            last_barrier = i
    # the end: every tile drained for the last layer
    for t in range(48):
        if acc[t]["read"] != {0, 1, 2, 3}:
            raise CheckError("tile %d not drained at the end" % t)
    for J in range(4):
        if act[J] != [nlayers - 1] * 6:
            raise CheckError("activation rows at the end: %s" % act)
    return True


def build(stamps, nblocks_check=3):
    """sections of the emitted text and the linearized 3-block instance that is checked"""
    head, pro = sec_head(stamps), sec_pro(stamps)
    even = [sec_even(stamps, b) for b in range(nblocks_check)]
    odd = [sec_odd(stamps, b) for b in range(nblocks_check)]
    x2 = [sec_x2(stamps, b) for b in range(nblocks_check)]
    fin = sec_fin(stamps, nblocks_check - 1)
    # linear instance
    linear = [Ins("", "blk", blk=0)] + list(pro.ins)
    for b in range(nblocks_check):
        linear += [Ins("", "blk", blk=b)] + even[b].ins + odd[b].ins
        linear += x2[b].ins if b < nblocks_check - 1 else fin.ins
    linear = insert_lgkm_waits(linear)
    check_and_fill(linear, nblocks_check, verify=not ABL)
    # split the linear list back into sections by identity of the first / last instruction objects
    def section(first, last):
        i0 = next(i for i, x in enumerate(linear) if x is first)
        i1 = next(i for i, x in enumerate(linear) if x is last)
        while i0 > 0 and linear[i0 - 1].kind == "lgkmwait":
            i0 -= 1
        return linear[i0:i1 + 1]
    S = {"pro": section(pro.ins[0], pro.ins[-1])}
    for b in range(nblocks_check):
        S["even%d" % b] = section(even[b].ins[0], even[b].ins[-1])
        S["odd%d" % b] = section(odd[b].ins[0], odd[b].ins[-1])
        if b < nblocks_check - 1:
            S["x2_%d" % b] = section(x2[b].ins[0], x2[b].ins[-1])
    S["fin"] = section(fin.ins[0], fin.ins[-1])
    # the loop body is emitted once: every block's instance must be the same text
    def texts(sec):
        import re
        return [re.sub(r"XQ1WA_\d+_", "XQ1WA_N_", x.text) for x in sec if x.kind != "comment"]
    def unify(names):
        """the block instances of a section may differ in a barrier's vmcnt only (stamped bodies: the stamp stores of the code in
        front of the loop head differ between the first and the later ways in): the emitted one takes the smallest count - a
        stricter wait is always valid"""
        secs = [[x for x in S[nm] if x.kind != "comment"] for nm in names]
        if any(len(sec) != len(secs[0]) for sec in secs):
            raise CheckError("section %s: instances differ in length" % names[0])
        for row in zip(*secs):
            tx = [texts([x])[0] for x in row]
            if len(set(tx)) > 1:
                if not all(x.kind == "vmwait" for x in row):
                    raise CheckError("section %s differs between blocks: %s" % (names[0], tx))
                row[0].text = "s_waitcnt vmcnt(%d)" % min(x.m["n"] for x in row)
    unify(["even%d" % b for b in range(nblocks_check)])
    unify(["odd%d" % b for b in range(nblocks_check)])
    unify(["x2_%d" % b for b in range(nblocks_check - 1)])
    return head, S, linear


def render(stamps):
    head, S, linear = build(stamps)
    lines = []
    def put(sec):
        for x in sec:
            lines.append(x.text)
    put(head.ins)
    put(S["pro"])
    lines.append("XQ1WA_LOOP_%=:")
    put(S["even0"])
    put(S["odd0"])
    lines.append("s_cmp_eq_u32 s%d, 1" % S_CNT)
    lines.append("s_cbranch_scc1 XQ1WA_FIN_%=")
    put(S["x2_0"])
    lines.append("s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT))
    lines.append("s_add_u32 s%d, s%d, 0x%x" % (S_BLK, S_BLK, BLOCK_BYTES))
    if stamps:
        lines.append("s_add_u32 s%d, s%d, 32" % (S_STAMP, S_STAMP))
        lines.append("s_addc_u32 s%d, s%d, 0" % (S_STAMP + 1, S_STAMP + 1))
    lines.append("s_branch XQ1WA_LOOP_%=")
    lines.append("XQ1WA_FIN_%=:")
    put(S["fin"])
    stats = {"instructions": sum(1 for l in lines if not l.startswith(";") and not l.endswith(":")),
             "mfma": sum(1 for l in lines if l.startswith("v_mfma")),
             "lgkm_waits": sum(1 for l in lines if l.startswith("s_waitcnt lgkmcnt")),
             "checked": len(linear)}
    return lines, stats


def as_c_string(lines, name):
    out = ["#define %s \\" % name]
    for l in lines:
        out.append('    "%s\\n\\t" \\' % l.replace("\\", "\\\\").replace('"', '\\"'))
    out.append('    ""')
    return "\n".join(out) + "\n"


def generate():
    parts = ["// xq_tower1wa_body.inc - GENERATED by tools/gen_tower1wa.py (python tools/gen_tower1wa.py); do not edit.\n"
             "// The residual tower of k_tower1wa as one asm statement: see the generator for the schedule and the checks it passed.\n"]
    allstats = {}
    for stamps, name in ((False, "XQ_1WA_BODY"), (True, "XQ_1WA_BODY_STAMPED")):
        lines, stats = render(stamps)
        allstats[name] = stats
        parts.append("// %s: %d instructions (%d MFMAs per loop body + first tap + last epilogue), %d exact lgkmcnt waits; %d instructions of a 3-block instance checked\n"
                     % (name, stats["instructions"], stats["mfma"], stats["lgkm_waits"], stats["checked"]))
        parts.append(as_c_string(lines, name))
    consts = ("#define XQ_1WA_LDS_BYTES %d\n#define XQ_1WA_JUNK %d\n#define XQ_1WA_TAB_DWORDS %d\n#define XQ_1WA_V_LAST %d\n#define XQ_1WA_A_LAST %d\n#define XQ_1WA_S_FIRST %d\n#define XQ_1WA_S_LAST %d\n"
              % (LDS_TOTAL, JUNK, TAB_DWORDS, V_LAST, A_LAST, S_RSRC, S_LAST))
    parts.append(consts)
    return "".join(parts), allstats


ABLATIONS = [("noaccread",), ("nostore",), ("nodrain",), ("nodma",), ("nobar",), ("nodrain", "nodma", "nobar"),
             ("valu2",), ("valu3",), ("valu4",), ("valu3", "dma35"), ("valu4", "dma35"), ("drop2",), ("drop4",)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="regenerate in memory and compare with the committed file")
    ap.add_argument("--dump", default="", help="write the plain assembly text of the unstamped body to this file")
    ap.add_argument("--ablations", action="store_true",
                    help="write csrc/xq_tower1wa_abl.inc: stamped timing-only bodies (wrong results) for a -DXQ_TOWER_PROBES=1 library")
    args = ap.parse_args()
    if args.ablations:
        parts = ["// xq_tower1wa_abl.inc - GENERATED by tools/gen_tower1wa.py --ablations: timing-only bodies, wrong results (probes builds only)\n"]
        for k, flags in enumerate(ABLATIONS, 1):
            ABL.clear()
            ABL.update(flags)
            lines, _ = render(True)
            parts.append("// ablation %d: %s\n" % (k, ", ".join(flags)))
            parts.append(as_c_string(lines, "XQ_1WA_BODY_ABL%d" % k))
        ABL.clear()
        parts.append("#define XQ_1WA_N_ABL %d\n" % len(ABLATIONS))
        open(OUT.replace("_body.inc", "_abl.inc"), "w").write("".join(parts))
        print("wrote", OUT.replace("_body.inc", "_abl.inc"))
        return 0
    text, stats = generate()
    if args.dump:
        lines, _ = render(False)
        open(args.dump, "w").write("\n".join(lines) + "\n")
    if args.check:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("xq_tower1wa_body.inc is stale: run python tools/gen_tower1wa.py")
            return 1
        print("ok", stats)
        return 0
    open(OUT, "w").write(text)
    print("wrote", OUT, stats)
    return 0


if __name__ == "__main__":
    sys.exit(main())
