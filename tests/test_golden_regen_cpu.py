"""Regenerate-and-compare: when the reference is present (/root/reference — the build container only), re-run the
committed generator `oracle/gen_golden.py` against the UNMODIFIED reference into a scratch directory and require the
fixtures under tests/golden/ byte for byte.  This is what pins the oracle: the fixtures are outputs of
the reference itself, and anybody with the reference can see that they still are.  Skipped where the reference does not
exist (the GPU box).  The search fixtures (whole reference games, minutes of CPU) are regenerated only with
XQ_REGEN_SEARCH=1; everything else takes well under a minute."""
import os
import pickle
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("XQ_REFERENCE", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "chess_env.py")), reason="reference not present")

FAST = {
    "rules": ["rules_random.npz"], "edge": ["rules_edge.json"], "known": ["known.json"], "puct": ["puct.json"],
    "sampler": ["sampler.json"], "ztable": ["ztable.json"], "rules_extra": ["rules_extra.json"], "net": ["net.npz"],
    "trainer_io": ["trainer_io.npz", "checkpoint_struct.json", "best_games_ref.pkl"],
}


def _start(what, out):
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, XQ_GOLDEN_OUT=str(out), XQ_REFERENCE=REF, PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="1")
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), what], env=env,
                            stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)


def _regen(what, out):
    p = _start(what, out)
    _, err = p.communicate(timeout=900)
    assert p.returncode == 0, err.decode()[-2000:]


@pytest.fixture(scope="module")
def regenerated(tmp_path_factory):
    """every fast generator at once, one process and one scratch directory each (the slowest takes under a minute)"""
    base = tmp_path_factory.mktemp("golden_regen")
    procs = {w: _start(w, os.path.join(base, w)) for w in sorted(FAST)}
    for w, p in procs.items():
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0, (w, err.decode()[-2000:])
    return base


def _same_file(a, b):
    with open(a, "rb") as fa, open(b, "rb") as fb:
        return fa.read() == fb.read()


@pytest.mark.parametrize("what", sorted(FAST))
def test_fixture_regenerates_byte_identically(what, regenerated):
    for name in FAST[what]:
        new, old = os.path.join(regenerated, what, name), os.path.join(GOLDEN, name)
        assert os.path.exists(new), name
        if name == "best_games_ref.pkl":
            # the reference's _save_best_games stamps every entry with the wall clock (trainer.py:468-502): everything else equal
            a, b = pickle.load(open(new, "rb")), pickle.load(open(old, "rb"))
            assert len(a) == len(b)
            for x, y in zip(a, b):
                x, y = dict(x), dict(y)
                x.pop("timestamp"), y.pop("timestamp")
                assert pickle.dumps(x) == pickle.dumps(y)
        else:
            assert _same_file(new, old), "%s: the reference no longer produces the committed fixture" % name


@pytest.mark.skipif(os.environ.get("XQ_REGEN_SEARCH", "0") != "1", reason="whole reference games: minutes; set XQ_REGEN_SEARCH=1")
def test_search_fixture_regenerates_byte_identically(tmp_path):
    _regen("search", tmp_path)
    assert _same_file(os.path.join(tmp_path, "search_hashnet.json"), os.path.join(GOLDEN, "search_hashnet.json"))
