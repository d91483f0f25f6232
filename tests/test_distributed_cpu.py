"""N>1 path on CPU: world_size-2 and -3 gloo runs of the sharding + all-gather of sample records."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from chinesechessai_amd import distributed as xd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_ranges_cover_and_seeds_are_sharding_independent():
    for n in (1, 5, 8, 16384, 16385):
        for w in (1, 2, 3, 8):
            spans = [xd.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
            seeds = np.concatenate([xd.game_seeds(7, n, r, w) for r in range(w)])
            assert np.array_equal(seeds, xd.game_seeds(7, n, 0, 1))


def test_record_layout_matches_c_struct():
    import re
    hdr = open(os.path.join(ROOT, "include", "xq_selfplay.h")).read()
    assert "uint32_t board[12]" in hdr and "uint16_t counts[XQ_MAX_MOVES]" in hdr
    assert xd.RECORD_DTYPE.itemsize == 576
    assert xd.RECORD_DTYPE.fields["z"][1] == 48 and xd.RECORD_DTYPE.fields["moves"][1] == 64
    src = open(os.path.join(ROOT, "chinesechessai_amd", "csrc", "xq_engine.hip")).read()
    assert re.search(r"static_assert\(sizeof\(xq_sample_record\) == 576", src)


@pytest.mark.parametrize("world", [2, 3])
def test_all_gather_records_gloo(world):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "DIST_OK world=%d" % world in p.stdout
