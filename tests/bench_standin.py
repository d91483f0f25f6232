"""Test-only stand-in for the HIP engine in bench.py's CPU rehearsal (XQ_BENCH_BACKEND=gloo):
plays each rank's shard with the CPU oracle (shortened games) and returns the sample records in
the production wire format, so the launcher / sharding / all-gather / timing code of bench.py runs
unchanged on a machine without a GPU.  Never imported by the product path."""
import numpy as np
import torch

from tests.dist_worker import oracle_records


def make_step(n_games, sims):
    def play(seeds):
        assert len(seeds) == n_games
        rec = oracle_records(seeds, sims, n_games)
        return torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy())
    return play
