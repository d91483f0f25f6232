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
    check_broadcast_weights(rank, world)
    dist.barrier()
    if rank == 0:
        print("DIST_OK world=%d" % world)
    dist.destroy_process_group()


def check_broadcast_weights(rank, world):
    """SURVEY.md 8(e) "identical weights on every rank": every rank starts from DIFFERENT weights (and BatchNorm
    statistics) and must end bit-equal to rank 0, whose own tensors must not change; the shard a rank then plays is
    the single-rank run's (seeds base + g, checked above); ranks that disagree on the architecture all raise."""
    import hashlib
    from chinesechessai_amd.neural_network import ChessNet
    torch.manual_seed(100 + rank)
    net = ChessNet(num_blocks=1).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.num_batches_tracked.fill_(7 + rank)

    def digest(sd):
        h = hashlib.sha256()
        for k, t in sd.items():
            h.update(k.encode())
            h.update(t.detach().cpu().contiguous().numpy().tobytes())
        return h.digest()

    before = digest(net.state_dict())
    tensors = [t for t in net.state_dict().values() if t.is_floating_point()]
    assert xd.weights_equal_across_ranks(tensors) == (world == 1)          # the check bench.py prints as weights_equal
    nbytes = xd.broadcast_weights(net, src=0)
    after = digest(net.state_dict())
    assert xd.weights_equal_across_ranks(tensors)
    if world > 1:                                                           # one flipped mantissa bit on one rank is seen
        keep = tensors[3].view(-1)[5].item()
        if rank == world - 1:
            tensors[3].view(-1)[5] = float(np.nextafter(np.float32(keep), np.float32(9)))
        assert not xd.weights_equal_across_ranks(tensors)
        tensors[3].view(-1)[5] = keep
        assert xd.weights_equal_across_ranks(tensors)
    assert nbytes == sum(t.numel() * (4 if t.is_floating_point() else 8) for t in net.state_dict().values())
    mine = torch.tensor(list(after), dtype=torch.uint8)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    assert all(torch.equal(e, every[0]) for e in every), "ranks differ after broadcast_weights"
    if rank == 0:
        assert after == before                         # the source's tensors are untouched
    else:
        assert after != before                         # (this rank really started from other weights)
    assert int(net.bn1.num_batches_tracked) == 7       # integer buffers travel too
    # a state_dict works as well as a module, and from another source rank
    sd = {k: v.clone() + (rank if v.is_floating_point() else 0) for k, v in net.state_dict().items()}
    xd.broadcast_weights(sd, src=world - 1)
    want = net.state_dict()["conv1.weight"] + (world - 1)
    assert torch.equal(sd["conv1.weight"], want)
    # architecture mismatch: every rank raises, nobody hangs in the payload broadcast
    other = ChessNet(num_blocks=1 if rank == 0 else 2)
    try:
        xd.broadcast_weights(other, src=0)
        raised = False
    except ValueError:
        raised = True
    assert raised


if __name__ == "__main__":
    main()
