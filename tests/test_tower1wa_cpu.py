"""CPU suite for the hand-written trunk kernel k_tower1wa: the committed assembly body is what the generator emits and
passes its symbolic checker; the checker really rejects broken schedules; the compiled kernel keeps the compiler out of
the accumulator registers.  (Parity of the kernel itself: tests/test_gpu_parity.py, bit for bit against k_tower16b.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_committed_body_is_the_generators_and_passes_the_symbolic_check():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_tower1wa.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


def _linear(g):
    n = 3
    pro = g.sec_pro(False)
    even = [g.sec_even(False, b) for b in range(n)]
    odd = [g.sec_odd(False, b) for b in range(n)]
    x2 = [g.sec_x2(False, b) for b in range(n)]
    fin = g.sec_fin(False, n - 1)
    lin = [g.Ins("", "blk", blk=0)] + list(pro.ins)
    for b in range(n):
        lin += [g.Ins("", "blk", blk=b)] + even[b].ins + odd[b].ins
        lin += x2[b].ins if b < n - 1 else fin.ins
    return g.insert_lgkm_waits(lin)


def test_the_symbolic_checker_rejects_broken_schedules():
    """mutations of the instruction stream that would give wrong results or races on the GPU: each must be caught"""
    import gen_tower1wa as g

    def drop_wait(l):
        del l[[i for i, x in enumerate(l) if x.kind == "lgkmwait"][40]]

    def early_store(l):
        st = [i for i, x in enumerate(l) if x.kind == "ldsw"][30]
        l.insert(st - 400, l.pop(st))

    def k_order(l):
        a = [i for i, x in enumerate(l) if x.kind == "mfma" and x.m["want"] == (1, 0, 0) and x.m["tile"] == 0][0]
        b = [i for i, x in enumerate(l) if x.kind == "mfma" and x.m["want"] == (1, 0, 1) and x.m["tile"] == 0][0]
        l[a], l[b] = l[b], l[a]

    def no_bx(l):
        j = [k for k, x in enumerate(l) if x.kind == "barrier" and x.m["note"] == "BX"][0]
        del l[j]
        del l[j - 1]

    def dma_two_barriers_early(l):
        bars = [i for i, x in enumerate(l) if x.kind == "barrier"]
        d = next(i for i in range(bars[40], len(l)) if l[i].kind == "dma")
        l.insert(bars[38] - 1, l.pop(d))

    def early_accumulator_read(l):
        i = [i for i, x in enumerate(l) if x.kind == "valu" and "accread" in x.m][0]
        j = [k for k, x in enumerate(l) if x.kind == "valu" and "accread" in x.m and x.m["accread"][0] == 41][0]
        l.insert(i, l.pop(j))

    lin = _linear(g)
    assert g.check_and_fill(lin, 3)
    for mut in (drop_wait, early_store, k_order, no_bx, dma_two_barriers_early, early_accumulator_read):
        lin = _linear(g)
        mut(lin)
        with pytest.raises(g.CheckError):
            g.check_and_fill(lin, 3)


def test_compiled_kernel_keeps_the_compiler_out_of_the_accumulators():
    """tools/scan_tower1wa_isa.py on a fresh compile: no compiler-generated use of a0..a223 while the accumulators are live,
    no VALU write right in front of an asm MFMA that reads it, no scratch, <= 512 registers"""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_tower1wa_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
