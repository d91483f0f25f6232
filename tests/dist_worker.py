"""world_size-N gloo worker for tests/test_distributed_cpu.py: shards game indices, builds each
rank's sample records from ORACLE games (CPU stand-in for the HIP engine, test-only), all-gathers
them with the production gather code and checks the result against a single-rank run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from chinesechessai_amd import distributed as xd
from oracle import xq_oracle as xo


def pack_board(board90):
    code = np.where(board90 > 0, board90, np.where(board90 < 0, 7 - board90, 0)).astype(np.uint32)
    w = np.zeros(12, np.uint32)
    for s in range(90):
        w[s // 8] |= code[s] << np.uint32(4 * (s % 8))
    return w


def oracle_records(seeds, sims, n_pad):
    rec = np.zeros((n_pad, 70), dtype=xd.RECORD_DTYPE)
    for k, seed in enumerate(seeds):
        rc, g = xo.self_play_game(int(seed), sims, max_moves=6)
        assert rc == 0
        for i in range(g.n_samples):
            r = rec[k, i]
            r["board"] = pack_board(np.frombuffer(g.s_board[i], dtype=np.int8))
            r["z"] = g.s_z[i]
            r["player"] = g.s_player[i]
            n = g.s_nmoves[i]
            r["n_moves"] = n
            r["valid"] = 1
            r["chosen"] = g.t_move[i]
            r["moves"][:n] = g.s_moves[i][:n]
            r["counts"][:n] = g.t_visits[i][:n]
    return rec


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    num_games, sims, base = 5, 16, 40
    lo, hi = xd.shard_range(num_games, rank, world)
    n_pad = xd.shard_range(num_games, 0, world)[1]
    seeds = xd.game_seeds(base, num_games, rank, world)
    assert len(seeds) == hi - lo
    local = oracle_records(seeds, sims, n_pad)
    t = torch.from_numpy(np.frombuffer(local.tobytes(), dtype=np.uint8).copy())
    out = xd.all_gather_records(t)
    allrec = xd.records_to_numpy(out).reshape(world, n_pad, 70)
    # single-rank truth
    truth = oracle_records(xd.game_seeds(base, num_games, 0, 1), sims, num_games)
    k = 0
    for r in range(world):
        a, b = xd.shard_range(num_games, r, world)
        for j in range(b - a):
            assert allrec[r, j].tobytes() == truth[k].tobytes(), (r, j)
            k += 1
        for j in range(b - a, n_pad):
            assert not allrec[r, j]["valid"].any()
    assert k == num_games
    board, pi, z = xd.record_to_sample(allrec[0, 0, 0])
    assert board.shape == (10, 9) and abs(sum(pi.values()) - 1) < 1e-12 and len(pi) == 44
    # the overlapped form bench.py uses (RecordGather: two buffers in turn, async gathers): five epochs of distinct
    # data; every epoch's gathered tensor must equal the synchronous gather of the same bytes, also when it is only
    # waited for two epochs later
    pipe = xd.RecordGather(t.numel(), "cpu")
    want, outs = [], []
    for epoch in range(5):
        local_e = torch.roll(t, epoch * 7 + rank) ^ epoch
        want.append(xd.all_gather_records(local_e))
        buf = pipe.next_buffer()                       # waits for the gather of epoch - 2
        if epoch >= 2:
            assert torch.equal(pipe.out[epoch % 2], want[epoch - 2]), epoch
        buf.copy_(local_e)
        pipe.launch()
    last = pipe.drain()
    assert torch.equal(last, want[4]) and torch.equal(pipe.out[1], want[3])
    dist.barrier()
    if rank == 0:
        print("DIST_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
