"""Multi-GPU self-play: games shard across ranks, one process per GPU, no exchange during play;
one RCCL all-gather of fixed-size sample records at the end of the epoch, one broadcast of the weights at its
start.

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


def broadcast_weights(net_or_state_dict, src=0, group=None, device=None):
    """Every rank ends up with rank `src`'s weights, bit for bit: the counterpart of the reference shipping the
    state_dict to every worker with its task (self_play.py:386,394, rebuilt at :337-339) - here once per epoch
    instead of once per game.  All floating-point tensors of the state_dict (parameters and BatchNorm statistics,
    in state_dict order) travel as ONE flat float32 buffer (24.6 M values = 98.5 MB for the reference net; fp32
    because InferenceNet folds BatchNorm from the fp32 values), integer buffers (num_batches_tracked) as one flat
    int64 buffer: two broadcasts in all.  Tensors are overwritten in place; a leaf evaluator must be (re)built
    from them afterwards (InferenceNet folds and re-lays the weights when it is constructed).
    `device`: where the flat buffers live - default "cuda" under nccl (RCCL), "cpu" under gloo.
    Returns the number of bytes broadcast."""
    import torch
    import torch.distributed as dist
    sd = net_or_state_dict.state_dict() if hasattr(net_or_state_dict, "state_dict") else net_or_state_dict
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    fl = [t for t in sd.values() if torch.is_tensor(t) and t.is_floating_point()]
    it = [t for t in sd.values() if torch.is_tensor(t) and not t.is_floating_point()]
    # every rank must hold the same architecture: compare the layout before any payload moves
    sig = torch.tensor([len(fl), sum(t.numel() for t in fl), len(it), sum(t.numel() for t in it)], dtype=torch.int64, device=device)
    ref = sig.clone()
    dist.broadcast(ref, src, group=group)
    ok = torch.tensor([1 if torch.equal(ref, sig) else 0], dtype=torch.int64, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)     # every rank raises, or none: nobody is left in a broadcast
    if int(ok.item()) == 0:
        raise ValueError("broadcast_weights: state_dict layouts differ between ranks (this rank %s, rank %d %s)"
                         % (sig.tolist(), src, ref.tolist()))
    nbytes = 0
    for tensors, dtype in ((fl, torch.float32), (it, torch.int64)):
        if not tensors:
            continue
        flat = torch.cat([t.detach().reshape(-1).to(device=device, dtype=dtype) for t in tensors])
        dist.broadcast(flat, src, group=group)
        nbytes += flat.numel() * flat.element_size()
        off = 0
        with torch.no_grad():
            for t in tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape).to(dtype=t.dtype))
                off += n
    return nbytes


def weights_digest(tensors):
    """int64[3] digest of a list of tensors' BIT patterns (count, wrap-around sum, position-weighted wrap-around sum of
    the 16-bit words): equal tensors give equal digests on every device; cheap enough to run after every broadcast."""
    import torch
    dev = tensors[0].device
    d = torch.zeros(3, dtype=torch.int64, device=dev)
    base = 0
    for t in tensors:
        w = t.detach().contiguous().view(-1).view(torch.int16).to(torch.int64) & 0xFFFF
        n = w.numel()
        pos = (torch.arange(base, base + n, dtype=torch.int64, device=dev) % 65521) + 1
        d[0] += n
        d[1] += w.sum()
        d[2] += (w * pos).sum()                       # (int64 wrap-around is fine: a digest, and the same on every rank)
        base += n
    return d


def weights_equal_across_ranks(tensors, group=None, device=None):
    """True on every rank iff every rank holds bit-identical `tensors` (e.g. InferenceNet.folded_weights() after
    broadcast_weights): MIN and MAX all-reduce of the digest agree."""
    import torch
    import torch.distributed as dist
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    d = weights_digest(tensors).to(device)
    lo, hi = d.clone(), d.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))


def play_sharded(make_evaluator, num_games, sims, base_seed=0, temperature=1.0, group=None, gather=True, network=None):
    """Each rank plays its shard on its own GPU, then all ranks all-gather the sample records.
    `network` (a ChessNet on this rank's GPU): rank 0's weights are broadcast to every rank first
    (broadcast_weights: what parallel_self_play's per-task state_dict does, self_play.py:386-394) and every rank
    builds its leaf evaluator from them; `make_evaluator` may then be None.  Without `network` the caller's
    `make_evaluator()` is trusted to produce identical evaluators on all ranks.
    Returns (engine results of the local shard, gathered uint8 tensor or None)."""
    import torch
    import torch.distributed as dist
    from .engine import SelfPlayEngine, TorchNetEvaluator
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if num_games < world:
        # every rank sees the same arguments, so every rank raises: nobody is left waiting in the all-gather
        raise ValueError("play_sharded: %d games cannot be sharded over %d ranks (each rank needs at least one)"
                         % (num_games, world))
    lo, hi = shard_range(num_games, rank, world)
    n_local = hi - lo
    n_pad = shard_range(num_games, 0, world)[1]               # largest shard
    if network is not None:
        broadcast_weights(network, src=0, group=group)
        ev = make_evaluator() if make_evaluator is not None else TorchNetEvaluator(network)
    else:
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
