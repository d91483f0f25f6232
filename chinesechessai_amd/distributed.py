"""Multi-GPU self-play: games shard across ranks, one process per GPU, no exchange during play;
one RCCL all-gather of fixed-size sample records at the end of the epoch.

The reference's only parallelism is a 4-process pool with pickled results
(self_play.py:368-469); here `torch.distributed` (backend "nccl" = RCCL over xGMI, "gloo" in the
CPU tests) gathers `xq_sample_record[G][70]` shards with all_gather_into_tensor — a direct
all-gather: every rank's shard crosses each of the 7 xGMI links once.
"""
import numpy as np

from . import _lib
from .chess_env import decode_move

RECORD_BYTES = _lib.SAMPLE_RECORD_BYTES           # sizeof(xq_sample_record) = 576
RECORD_DTYPE = np.dtype([("board", "<u4", (12,)), ("z", "<f8"), ("player", "i1"), ("n_moves", "u1"),
                         ("valid", "u1"), ("pad", "u1"), ("chosen", "<u2"), ("pad2", "<u2"),
                         ("moves", "<u2", (128,)), ("counts", "<u2", (128,))])
assert RECORD_DTYPE.itemsize == RECORD_BYTES


def shard_range(num_games, rank, world_size):
    """Contiguous block of game indices for `rank`; the remainder goes to the low ranks."""
    base, rem = divmod(num_games, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def game_seeds(base_seed, num_games, rank, world_size):
    """seed_g = base + g: independent of the sharding, so a W-rank run plays
    exactly the games a 1-rank run plays."""
    lo, hi = shard_range(num_games, rank, world_size)
    return (np.uint64(base_seed) + np.arange(lo, hi, dtype=np.uint64)).astype(np.uint32)


def all_gather_records(local, group=None):
    """local: uint8 tensor [n_local_records * RECORD_BYTES] (device tensor under RCCL, CPU tensor
    under gloo).  Every rank must contribute the same number of records (pad with valid=0).
    Returns the concatenation over ranks in rank order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


class RecordGather:
    """The epoch-end all-gather, overlapped with the next epoch's play (what a training loop does with it: the
    samples of epoch k travel while epoch k + 1 is being played).  Two local buffers in turn: `next_buffer()`
    hands out the one whose gather (two epochs ago) has been waited for, `launch()` starts the gather of the
    buffer just filled (async_op: RCCL's stream first waits for the work already queued on the current stream,
    i.e. for pack_samples), `drain()` waits for everything in flight and returns the newest gathered tensor.
    Every byte is gathered exactly as all_gather_records does; only the waiting moves."""

    def __init__(self, nbytes, device, group=None):
        import torch
        self.group = group
        self.buf = [torch.zeros(nbytes, dtype=torch.uint8, device=device) for _ in range(2)]
        self.out = [None, None]
        self.work = [None, None]
        self.k = 0

    def next_buffer(self):
        i = self.k % 2
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        return self.buf[i]

    def launch(self):
        import torch
        import torch.distributed as dist
        i = self.k % 2
        world = dist.get_world_size(self.group)
        if self.out[i] is None:
            self.out[i] = torch.empty(world * self.buf[i].numel(), dtype=torch.uint8, device=self.buf[i].device)
        self.work[i] = dist.all_gather_into_tensor(self.out[i], self.buf[i], group=self.group, async_op=True)
        self.k += 1

    def drain(self):
        for i in range(2):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None
        return self.out[(self.k - 1) % 2] if self.k else None


def records_to_numpy(t):
    return np.frombuffer(t.detach().cpu().numpy().tobytes(), dtype=RECORD_DTYPE)


def unpack_board(words):
    """nibble-packed board (12 x uint32) -> int8[10, 9]"""
    w = np.asarray(words, dtype=np.uint32)
    s = np.arange(90)
    code = (w[s // 8] >> (4 * (s % 8)).astype(np.uint32)) & 15
    code = code.astype(np.int16)
    return np.where(code <= 7, code, 7 - code).astype(np.int8).reshape(10, 9)


def record_to_sample(rec, temperature=1.0):
    """One valid record -> the reference's (board, {move: prob}, z) tuple (self_play.py:234-239,310)."""
    n = int(rec["n_moves"])
    counts = rec["counts"][:n].astype(np.int64)
    if temperature < 0.01:
        probs = np.zeros(n)
        probs[np.argmax(counts)] = 1
    else:
        c = counts ** (1.0 / temperature)
        probs = c / c.sum()
    moves = [decode_move(m) for m in rec["moves"][:n]]
    return unpack_board(rec["board"]), {m: p for m, p in zip(moves, probs)}, float(rec["z"])


def play_sharded(make_evaluator, num_games, sims, base_seed=0, temperature=1.0, group=None, gather=True):
    """Each rank plays its shard on its own GPU, then all ranks all-gather the sample records.
    Returns (engine results of the local shard, gathered uint8 tensor or None)."""
    import torch
    import torch.distributed as dist
    from .engine import SelfPlayEngine
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if num_games < world:
        # every rank sees the same arguments, so every rank raises: nobody is left waiting in the all-gather
        raise ValueError("play_sharded: %d games cannot be sharded over %d ranks (each rank needs at least one)"
                         % (num_games, world))
    lo, hi = shard_range(num_games, rank, world)
    n_local = hi - lo
    n_pad = shard_range(num_games, 0, world)[1]               # largest shard
    ev = make_evaluator()
    eng = SelfPlayEngine(n_local, sims=sims, temperature=temperature,
                         planes_format=getattr(ev, "planes_format", _lib.PLANES_NONE),
                         device=torch.cuda.current_device(),
                         stream=torch.cuda.current_stream().cuda_stream)
    eng.play(ev, game_seeds(base_seed, num_games, rank, world), read=False)
    local = torch.zeros(n_pad * _lib.MAX_PLIES * RECORD_BYTES, dtype=torch.uint8, device="cuda")
    eng.pack_samples(local.data_ptr())
    gathered = all_gather_records(local, group) if gather else None
    outcomes = eng.read_game_outcomes()
    eng.close()
    return outcomes, gathered
