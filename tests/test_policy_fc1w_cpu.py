"""CPU tests of the generator behind k_policy_fc1w's assembly body (tools/gen_policy_fc1w.py): the committed body is what the
generator emits today, every DMA placement it knows passes its symbolic check, and the check rejects broken schedules - a
slot refilled before the barrier that frees it, a fragment read of a stage that was never published, a missing slot toggle,
an MFMA on the wrong fragment, a missing K-step."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_policy_fc1w as G                      # noqa: E402


def test_committed_body_is_the_generators_output():
    assert open(G.OUT).read() == G.generate()


@pytest.mark.parametrize("placement", sorted(G.DMA_AT))
def test_every_placement_passes_the_check(placement):
    _, _, _, linear = G.linear_stream(3, placement)
    assert G.check(linear, 3)


def _stream():
    return G.linear_stream(3)[3]


def test_check_rejects_a_refill_in_front_of_the_freeing_barrier():
    lin = _stream()
    # first DMA piece of stage 2 (slot 0) and its M0 write moved in front of the barrier of K-step (0, 0)
    i_dma = next(i for i, x in enumerate(lin) if x.kind == "dma" and x.m["stage"] == 2)
    i_m0 = max(i for i in range(i_dma) if lin[i].kind == "m0")
    i_bar = max(i for i in range(i_m0) if lin[i].kind == "barrier")
    i_lg = i_bar - 1
    assert lin[i_lg].kind == "lgkm0"
    moved = [lin[j] for j in range(len(lin)) if lin[j].kind == "slot" and i_bar < j < i_dma] + [lin[i_m0], G.Ins("s_nop 0", "salu"), lin[i_dma]]
    rest = [x for x in lin if not any(x is y for y in moved)]
    k = next(i for i, x in enumerate(rest) if x is lin[i_lg])
    with pytest.raises(G.CheckError, match="refilled while a read"):
        G.check(rest[:k] + moved + rest[k:], 3)


def test_check_rejects_a_read_of_an_unpublished_stage():
    lin = _stream()
    # without the vmcnt wait in front of the barrier of K-step (0, 1), stage 1 is not known to have landed
    i = next(i for i, x in enumerate(lin) if x.kind == "vmwait" and x.m["landed"] == 1)
    with pytest.raises(G.CheckError, match="published False"):
        G.check(lin[:i] + lin[i + 1:], 3)


def test_check_rejects_a_missing_toggle():
    lin = _stream()
    i = next(i for i, x in enumerate(lin) if x.kind == "toggle")
    with pytest.raises(G.CheckError, match="points at slot"):
        G.check(lin[:i] + lin[i + 1:], 3)


def test_check_rejects_a_wrong_fragment_and_a_missing_k_step():
    lin = _stream()
    i = next(i for i, x in enumerate(lin) if x.kind == "mfma" and x.m["step"] == 2)
    bad = G.Ins(lin[i].text, "mfma", **dict(lin[i].m, a=lin[i].m["a"] ^ 1))
    with pytest.raises(G.CheckError, match="reads"):
        G.check(lin[:i] + [bad] + lin[i + 1:], 3)
    with pytest.raises(G.CheckError, match="K-steps"):
        G.check(lin[:i] + lin[i + 1:], 3)
